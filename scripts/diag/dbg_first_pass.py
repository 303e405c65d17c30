"""Diagnostic: does the FIRST forward/backward of a fresh process differ from the following ones (DIS-MF, 64x64)?  Runs groups of
concurrent fresh processes; each evaluates the same step 3 times and reports, per parameter family, pass 1 and pass 2 against pass 3."""
import argparse
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


def child(tag, mode):
    import torch
    from depthinspace_amd import synth, ops
    from depthinspace_amd.model import multi_frame_networks, multi_frame_worker
    from depthinspace_amd.trainer import FlatAdam
    H = W = 64
    args = argparse.Namespace(use_pseudo_gt=False, lcn_radius=5, track_length=4, data_type='synthetic',
                              architecture='multi_frame', epochs=1, warmup_epochs=150, train_batch_size=1, max_disp=128)
    settings = synth.make_settings(H, W)
    torch.manual_seed(0)
    w = multi_frame_worker.Worker(args, settings=settings)
    net = multi_frame_networks.FuseNet((H, W), settings.K, settings.baseline).cuda()
    w.build_losses()
    w.current_epoch = 2
    opt = FlatAdam(net.parameters(), lr=1e-4)
    batch = {k: torch.from_numpy(v) for k, v in synth.make_batch(settings, 1, 4, seed=1234).items()}
    names = [n for n, _ in net.named_parameters()]
    gs, outs = [], []
    for it in range(3):
        w.copy_data(batch, device=w.train_device, requires_grad=False, train=True)
        opt.zero_grad()
        flow = w.read_optical_flow(True)
        out = w.net_forward(net, flow)
        sum(w.loss_forward(out, True, flow)).backward()
        if mode == 'sync':
            torch.cuda.synchronize()
        gs.append(opt.flat_g.clone())
        outs.append(out.detach().clone())
    torch.cuda.synchronize()
    gm = float(gs[2].abs().max())
    msg = []
    for it in (0, 1):
        fam = {}
        for n_, p_, off in zip(names, opt.params, opt.offsets):
            sl = slice(off, off + p_.numel())
            e = float((gs[it][sl] - gs[2][sl]).abs().max()) / gm
            key = 'c3.w' if n_.endswith('conv3d_1.w') or n_.endswith('conv3d_2.w') else ('c3.mlp' if 'conv3d' in n_ else 'other')
            fam[key] = max(fam.get(key, 0.0), e)
        msg.append(f'pass{it + 1}: ' + ' '.join(f'{k}={v:.1e}' for k, v in sorted(fam.items())) +
                   f' out={float((outs[it] - outs[2]).abs().max()):.1e}')
    print(tag, mode, ' | '.join(msg), flush=True)


if __name__ == '__main__':
    if len(sys.argv) > 1 and sys.argv[1] == 'child':
        child(sys.argv[2], sys.argv[3])
    else:
        for rnd in range(4):
            ps = [subprocess.Popen([sys.executable, __file__, 'child', f'r{rnd}p{k}', 'sync' if rnd % 2 else 'nosync'],
                                   stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True) for k in range(3)]
            for p in ps:
                print(p.communicate()[0].strip())
