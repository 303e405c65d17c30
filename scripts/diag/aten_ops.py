"""Which ATen kernels (torch glue between the C-ABI launches) one eager DIS-MF step issues: name, count, device time."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench
from depthinspace_amd import synth
from depthinspace_amd.model import multi_frame_networks, multi_frame_worker
from depthinspace_amd.trainer import FlatAdam, GraphedStep

dev = torch.device('cuda:0')
H, W, TL = 512, 432, 4
settings = synth.make_settings(H, W)
torch.manual_seed(0)
net = multi_frame_networks.FuseNet(imsize=(H, W), K=settings.K, baseline=settings.baseline, track_length=TL, max_disp=128).to(dev)
worker = multi_frame_worker.Worker(bench.make_args(4), settings=settings, train_device=str(dev))
worker.build_losses(device=dev)
worker.current_epoch = 2
opt = FlatAdam(net.parameters(), lr=1e-4, world_size=1)
batch = bench.make_device_batch(settings, 4, 1234, dev)
stepper = GraphedStep(worker, net, opt, batch, use_graph=False, warmup=1)
for _ in range(2):
    stepper.run()
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=False) as prof:
    stepper.run()
    torch.cuda.synchronize()
rows = [(e.key, e.count, e.device_time_total if hasattr(e, 'device_time_total') else e.cuda_time_total) for e in prof.key_averages()
        if e.key.startswith('aten::')]
rows.sort(key=lambda r: -r[1])
for k, c, t in rows[:30]:
    print(f'{k:40s} x{c:4d}  {t:9.1f} us')
