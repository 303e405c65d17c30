"""Training sanity at the benchmark configuration: N steps of the captured step (trainer.GraphedStep, the object bench.py times) on
fresh synthetic batches, full resolution.  Prints the loss every few steps.    python scripts/training_sanity.py multi_frame|single_frame [bf16] [steps]"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from depthinspace_amd import synth
from depthinspace_amd.model import multi_frame_networks, multi_frame_worker, single_frame_worker, networks
from depthinspace_amd.trainer import FlatAdam, GraphedStep


def main():
    arch = sys.argv[1] if len(sys.argv) > 1 else 'multi_frame'
    bf = len(sys.argv) > 2 and sys.argv[2] == 'bf16'
    steps = int(sys.argv[-1]) if sys.argv[-1].isdigit() else 40
    H, W = 512, 432
    bs = 4 if arch == 'multi_frame' else 8
    args = argparse.Namespace(use_pseudo_gt=False, lcn_radius=5, track_length=4, data_type='synthetic', architecture=arch,
                              epochs=1, warmup_epochs=150, train_batch_size=bs, max_disp=128)
    settings = synth.make_settings(H, W)
    torch.manual_seed(0)
    if arch == 'multi_frame':
        w = multi_frame_worker.Worker(args, settings=settings)
        net = multi_frame_networks.FuseNet((H, W), settings.K, settings.baseline).cuda()
    else:
        w = single_frame_worker.Worker(args, settings=settings)
        net = networks.DispDecoder(channels_in=2, max_disp=128, imsizes=w.imsizes,
                                   act_dtype=torch.bfloat16 if bf else torch.float32).cuda()
    w.build_losses()
    w.current_epoch = 2
    opt = FlatAdam(net.parameters(), lr=1e-4)
    pool = [{k: torch.from_numpy(v) for k, v in synth.make_batch(settings, bs, 4, seed=500 + i, scene='bumps').items()}
            for i in range(6)]
    g = GraphedStep(w, net, opt, pool[0], use_graph=True, warmup=1)
    curve = []
    t0 = time.time()
    for it in range(steps):
        g.run(pool[it % len(pool)])
        if it % 4 == 3 or it == 0:
            curve.append((it, float(sum(g.losses()))))
    torch.cuda.synchronize()
    dt = time.time() - t0
    print(f'{arch}{" bf16" if bf else ""} bs={bs} {H}x{W}: mode {g.mode}, {steps} steps in {dt:.1f} s incl. capture and host batch copies; '
          f'adam steps {opt.step_count}')
    print('  total loss: ' + ', '.join(f'{it}: {v:.4f}' for it, v in curve))
    assert all(np.isfinite(v) for _, v in curve) and curve[-1][1] < curve[0][1]


if __name__ == '__main__':
    main()
