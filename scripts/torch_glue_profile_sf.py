#!/usr/bin/env python
"""Which tensors the small ATen kernels (adds, fills, copies) of a DIS-SF step work on (torch.profiler over one eager
step, grouped by op and input shapes).   python scripts/torch_glue_profile_sf.py"""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench as B
from depthinspace_amd import synth
from depthinspace_amd.model import single_frame_worker, networks
from depthinspace_amd.trainer import FlatAdam

dev = torch.device('cuda', 0)
settings = synth.make_settings(B.H, B.W)
worker = single_frame_worker.Worker(B.make_args(8, 'single_frame'), settings=settings, train_device=str(dev))
kw = {"act_dtype": torch.bfloat16} if (len(sys.argv) > 1 and sys.argv[1] == "bf16") else {}
net = networks.DispDecoder(channels_in=2, max_disp=128, imsizes=worker.imsizes, **kw).to(dev)
worker.build_losses(device=dev)
worker.current_epoch = 2
opt = FlatAdam(net.parameters(), lr=1e-4)
batch = B.make_device_batch(settings, 8, 1234, dev)
for _ in range(2):
    worker.train_step(net, opt, batch)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    worker.train_step(net, opt, batch)
    torch.cuda.synchronize()
agg, tim = collections.Counter(), collections.Counter()
for ev in prof.key_averages(group_by_input_shape=True):
    dt = getattr(ev, 'self_device_time_total', 0) or getattr(ev, 'self_cuda_time_total', 0)
    if not ev.key.startswith('aten::') or dt <= 0:
        continue
    k = (ev.key, str(ev.input_shapes)[:100])
    agg[k] += ev.count
    tim[k] += dt
for (name, where), c in sorted(agg.items(), key=lambda kv: -tim[kv[0]])[:40]:
    print(f'{c:4d} x {name:22s} {tim[(name, where)]:8.0f} us  {where}')
