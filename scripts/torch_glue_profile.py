#!/usr/bin/env python
"""Which tensors the small ATen kernels (adds, fills, copies) of a DIS-MF step work on: torch.profiler over one eager
step, ATen ops with a device kernel grouped by (op, input shapes).   python scripts/torch_glue_profile.py"""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

sys.argv = ['bench.py']
from depthinspace_amd import synth, ops
from depthinspace_amd.model import multi_frame_networks, multi_frame_worker
from depthinspace_amd.trainer import FlatAdam

dev = torch.device('cuda', 0)
settings = synth.make_settings(bench.H, bench.W)
torch.manual_seed(0)
net = multi_frame_networks.FuseNet(imsize=(bench.H, bench.W), K=settings.K, baseline=settings.baseline, track_length=bench.TL,
                                   max_disp=128).to(dev)
worker = multi_frame_worker.Worker(bench.make_args(4), settings=settings, train_device=str(dev))
worker.build_losses(device=dev)
worker.current_epoch = 2
opt = FlatAdam(net.parameters(), lr=1e-4, world_size=1)
batch = bench.make_device_batch(settings, 4, 1234, dev)


def step():
    worker.copy_data(batch, device=dev, requires_grad=False, train=True)
    opt.zero_grad()
    flow = worker.read_optical_flow(train=True)
    out = worker.net_forward(net, flow)
    errs = worker.loss_forward(out, True, flow)
    tot = getattr(errs, 'total', None)
    (tot if tot is not None else sum(errs)).backward()
    opt.step()


for _ in range(2):
    step()
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    step()
    torch.cuda.synchronize()
agg = collections.Counter()
tim = collections.Counter()
for ev in prof.key_averages(group_by_input_shape=True):
    dt = getattr(ev, 'self_device_time_total', 0) or getattr(ev, 'self_cuda_time_total', 0)
    if not ev.key.startswith('aten::') or dt <= 0:
        continue
    k = (ev.key, str(ev.input_shapes)[:90])
    agg[k] += ev.count
    tim[k] += dt
print('ops.begin_step:', hasattr(ops, 'begin_step'))
for (name, where), c in sorted(agg.items(), key=lambda kv: -tim[kv[0]])[:60]:
    print(f'{c:4d} x {name:22s} {tim[(name, where)]:8.0f} us  {where}')
