"""Diagnostic: run one DIS-MF / DIS-SF training step with every torch.empty() buffer pre-filled with NaN
(torch.utils.deterministic.fill_uninitialized_memory) and report which outputs / parameter gradients turn NaN, i.e.
where the step reads memory it never wrote.    python scripts/diag/find_uninit.py [multi_frame|single_frame] [H W]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

torch.use_deterministic_algorithms(True, warn_only=True)
torch.utils.deterministic.fill_uninitialized_memory = True

from depthinspace_amd import synth
from depthinspace_amd.model import multi_frame_networks, multi_frame_worker, single_frame_worker, networks
from depthinspace_amd.trainer import FlatAdam


def main():
    arch = sys.argv[1] if len(sys.argv) > 1 else 'multi_frame'
    H = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    W = int(sys.argv[3]) if len(sys.argv) > 3 else 64
    args = argparse.Namespace(use_pseudo_gt=False, lcn_radius=5, track_length=4, data_type='synthetic', architecture=arch,
                              epochs=1, warmup_epochs=150, train_batch_size=1, max_disp=128)
    settings = synth.make_settings(H, W)
    torch.manual_seed(0)
    if arch == 'multi_frame':
        w = multi_frame_worker.Worker(args, settings=settings)
        net = multi_frame_networks.FuseNet((H, W), settings.K, settings.baseline).cuda()
    else:
        w = single_frame_worker.Worker(args, settings=settings)
        net = networks.DispDecoder(channels_in=2, max_disp=128, imsizes=w.imsizes,
                                   act_dtype=torch.bfloat16 if os.environ.get('DIS_ACT_DTYPE') == 'bf16' else torch.float32).cuda()
    w.build_losses()
    w.current_epoch = 2
    opt = FlatAdam(net.parameters(), lr=1e-4)
    batch = {k: torch.from_numpy(v) for k, v in synth.make_batch(settings, 1, 4, seed=1234).items()}
    ref = None
    for it in range(2):
        w.copy_data(batch, device=w.train_device, requires_grad=False, train=True)
        opt.zero_grad()
        flow = w.read_optical_flow(True)
        out = w.net_forward(net, flow)
        losses = w.loss_forward(out, True, flow)
        sum(losses).backward()
        torch.cuda.synchronize()
        print('iteration', it, 'losses', [float(l) for l in losses])
        bad = 0
        for name, p in net.named_parameters():
            g = p.grad if p.grad is not None else None
            if g is None:
                continue
            nn = int(torch.isnan(g).sum())
            if nn:
                bad += 1
                print(f'  NaN in grad of {name}: {nn} of {g.numel()}')
        outs = out if isinstance(out, (list, tuple)) else [out]
        for i, o in enumerate(outs):
            if bool(torch.isnan(o).any()):
                print(f'  NaN in output {i}')
        print('  parameters with NaN gradients:', bad)
        if ref is None:
            ref = opt.flat_g.clone()
        else:
            d = (opt.flat_g - ref).abs()
            print('  max |g - g_first| / max|g| =', float(d.max()) / float(ref.abs().max()))


if __name__ == '__main__':
    main()
