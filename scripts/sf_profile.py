#!/usr/bin/env python
"""Per-call HIP-event timing of one eager DIS-SF bs=8 step, grouped by entry point and integer arguments."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import argparse, torch
import bench as B
from depthinspace_amd import synth, lib
from depthinspace_amd.model import single_frame_worker, networks
from depthinspace_amd.trainer import FlatAdam
dev = torch.device('cuda', 0)
settings = synth.make_settings(B.H, B.W)
bs = int(sys.argv[1]) if len(sys.argv) > 1 else 8
bf16 = len(sys.argv) > 2 and sys.argv[2] == 'bf16'   # DispNetS with bf16 activation storage (BASELINE config 2)
worker = single_frame_worker.Worker(B.make_args(bs, 'single_frame'), settings=settings, train_device=str(dev))
net = networks.DispDecoder(channels_in=2, max_disp=128, imsizes=worker.imsizes,
                           **({'act_dtype': torch.bfloat16} if bf16 else {})).to(dev)
worker.build_losses(device=dev)
worker.current_epoch = 2
opt = FlatAdam(net.parameters(), lr=1e-4)
batch = B.make_device_batch(settings, bs, 1234, dev)
for _ in range(2):
    worker.train_step(net, opt, batch)
lib.profile_start()
worker.train_step(net, opt, batch)
rec = lib.profile_stop()
agg = {}
for name, ia, ms, tag, _ in rec:
    name = name + ("@" + tag if tag else "")
    k = (name, ia)
    agg.setdefault(k, [0, 0.0]); agg[k][0] += 1; agg[k][1] += ms
tot = sum(v[1] for v in agg.values())
print('total kernel ms', tot)
for (name, ia), (n, ms) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:110]:
    print(f'{ms:8.3f} ms x{n:2d} {name} {ia}')
