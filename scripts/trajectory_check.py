"""HIP path vs the CPU oracle over several consecutive Adam steps (same parameters, same batch): loss trajectory, disparity
and parameter drift per step.      python scripts/trajectory_check.py multi_frame|single_frame [steps]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from depthinspace_amd import synth
from depthinspace_amd.model import multi_frame_networks, multi_frame_worker, single_frame_worker, networks
from depthinspace_amd.trainer import FlatAdam
from oracle import dis_oracle as O


def main():
    arch = sys.argv[1] if len(sys.argv) > 1 else 'single_frame'
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    H = W = 64
    args = argparse.Namespace(use_pseudo_gt=False, lcn_radius=5, track_length=4, data_type='synthetic', architecture=arch,
                              epochs=1, warmup_epochs=150, train_batch_size=1, max_disp=128)
    settings = synth.make_settings(H, W)
    batch = synth.make_batch(settings, 1, 4, seed=77, scene='bumps')
    mf = arch == 'multi_frame'
    params = O.init_params(O.mf_param_shapes() if mf else O.sf_param_shapes(), seed=3)
    if mf:
        w = multi_frame_worker.Worker(args, settings=settings)
        net = multi_frame_networks.FuseNet((H, W), settings.K, settings.baseline)
    else:
        w = single_frame_worker.Worker(args, settings=settings)
        net = networks.DispDecoder(channels_in=2, max_disp=128, imsizes=w.imsizes)
    net.load_state_dict({k: v.detach() for k, v in params.items()})
    net = net.cuda()
    w.build_losses()
    w.current_epoch = 2
    opt = FlatAdam(net.parameters(), lr=1e-4)
    ctx = O.StepContext(settings)
    st = {'step': 0, 'm': {}, 'v': {}}
    tb = {k: torch.from_numpy(v) for k, v in batch.items()}
    named = dict(net.named_parameters())
    for it in range(steps):
        errs, out = w.train_step(net, opt, tb)
        torch.cuda.synchronize()
        if mf:
            idx, idxq = net.last_knn_index
            O.CONV3D_FORCE = {'core': idx.cpu(), 'quarter': idxq.cpu()}
        r = O.train_step(ctx, arch, params, tb, adam_state=st, epoch=2)
        o_h = (out[0] if isinstance(out, (list, tuple)) else out).detach().cpu()
        o_r = (r['out'][0] if isinstance(r['out'], (list, tuple)) else r['out']).detach()
        lh = float(sum(float(e) for e in errs))
        lr_ = float(sum(float(v) for v in r['vals']))
        pd = max(float((named[k].detach().cpu() - params[k].detach()).abs().max()) for k in params)
        print(f'step {it}: loss hip {lh:.6f} oracle {lr_:.6f} (rel {abs(lh - lr_) / abs(lr_):.1e}); disparity L1 '
              f'{float((o_h - o_r).abs().mean()):.2e} max {float((o_h - o_r).abs().max()):.2e}; max parameter difference after the step {pd:.2e}',
              flush=True)


if __name__ == '__main__':
    main()
