"""DIS-SF with bf16 activation storage against the fp32 path over a short training run: same initial parameters, same batches
(8 synthetic 'bumps' tracks at 128x108, cycled), Adam lr 1e-4.  Prints the total loss of both runs every 10 steps and the final
full-resolution disparity difference.      python scripts/bf16_convergence.py [steps]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from depthinspace_amd import synth
from depthinspace_amd.model import single_frame_worker, networks
from depthinspace_amd.trainer import FlatAdam


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 120
    H, W = 128, 108
    args = argparse.Namespace(use_pseudo_gt=False, lcn_radius=5, track_length=4, data_type='synthetic',
                              architecture='single_frame', epochs=1, warmup_epochs=150, train_batch_size=2, max_disp=128)
    settings = synth.make_settings(H, W)
    w = single_frame_worker.Worker(args, settings=settings)
    w.build_losses()
    w.current_epoch = 2
    batches = [{k: torch.from_numpy(v) for k, v in synth.make_batch(settings, 2, 4, seed=100 + i, scene='bumps').items()}
               for i in range(4)]
    curves, finals = {}, {}
    for name, dt in (('fp32', torch.float32), ('bf16', torch.bfloat16)):
        torch.manual_seed(0)
        net = networks.DispDecoder(channels_in=2, max_disp=128, imsizes=w.imsizes, act_dtype=dt).cuda()
        opt = FlatAdam(net.parameters(), lr=1e-4)
        tot = []
        for it in range(steps):
            errs, _ = w.train_step(net, opt, batches[it % len(batches)])
            tot.append(float(sum(float(e) for e in errs)))
            del errs, _
        curves[name] = np.array(tot)
        with torch.no_grad():
            w.copy_data(batches[0], device=w.train_device, requires_grad=False, train=False)
            out = w.net_forward(net, None)
        finals[name] = (out[0] if isinstance(out, (list, tuple)) else out).float().cpu()
    print('step   fp32_loss   bf16_loss   rel_diff')
    for it in list(range(0, steps, 10)) + [steps - 1]:
        a, b = curves['fp32'][it], curves['bf16'][it]
        print(f'{it:4d}   {a:9.5f}   {b:9.5f}   {abs(a - b) / abs(a):8.2e}')
    m10 = lambda c: float(c[-10:].mean())
    print(f'mean of the last 10 steps: fp32 {m10(curves["fp32"]):.5f}  bf16 {m10(curves["bf16"]):.5f};  first step: '
          f'fp32 {curves["fp32"][0]:.5f}  bf16 {curves["bf16"][0]:.5f}')
    d = (finals['fp32'] - finals['bf16']).abs()
    print(f'disparity after {steps} steps (batch 0, scale 0): mean |fp32 - bf16| = {float(d.mean()):.4f} px, max {float(d.max()):.3f} px, '
          f'mean disparity {float(finals["fp32"].mean()):.2f} px')


if __name__ == '__main__':
    main()
