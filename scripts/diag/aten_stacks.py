"""Where the ATen kernels of one eager DIS-MF step come from: device kernels launched by aten ops, grouped by the innermost
repository source line on the Python stack (torch.profiler with_stack).  usage: python scripts/diag/aten_stacks.py [mf | sf | sf_bf16]"""
import os, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench
from depthinspace_amd import synth
from depthinspace_amd.model import multi_frame_networks, multi_frame_worker
from depthinspace_amd.trainer import FlatAdam, GraphedStep

dev = torch.device('cuda:0')
ARCH = sys.argv[1] if len(sys.argv) > 1 else 'mf'      # mf | sf | sf_bf16
H, W, TL = 512, 432, 4
settings = synth.make_settings(H, W)
torch.manual_seed(0)
if ARCH == 'mf':
    net = multi_frame_networks.FuseNet(imsize=(H, W), K=settings.K, baseline=settings.baseline, track_length=TL, max_disp=128).to(dev)
    worker = multi_frame_worker.Worker(bench.make_args(4), settings=settings, train_device=str(dev))
    BS = 4
else:
    from depthinspace_amd.model import single_frame_worker, networks
    BS = 8
    worker = single_frame_worker.Worker(bench.make_args(BS, 'single_frame'), settings=settings, train_device=str(dev))
    kw = dict(act_dtype=torch.bfloat16) if ARCH == 'sf_bf16' else {}
    net = networks.DispDecoder(channels_in=2, max_disp=128, imsizes=worker.imsizes, **kw).to(dev)
worker.build_losses(device=dev)
worker.current_epoch = 2
opt = FlatAdam(net.parameters(), lr=1e-4, world_size=1)
batch = bench.make_device_batch(settings, BS, 1234, dev)
stepper = GraphedStep(worker, net, opt, batch, use_graph=False, warmup=1)
for _ in range(2):
    stepper.run()
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    stepper.run()
    torch.cuda.synchronize()
rows = [(e.key, e.count, e.self_device_time_total) for e in prof.key_averages() if e.key.startswith('aten::') and e.self_device_time_total > 0]
rows.sort(key=lambda r: -r[2])
print('aten ops with device time in one eager step:')
for k, c, t in rows[:25]:
    print(f'  {k:36s} x{c:4d}  {t:9.1f} us')
import traceback, functools
LOG = collections.defaultdict(lambda: [0, 0])


def _where():
    fr = [f for f in traceback.extract_stack()[:-2] if 'depthinspace_amd' in f.filename or f.filename.endswith('bench.py')]
    return ' <- '.join(f'{os.path.basename(f.filename)}:{f.lineno}' for f in fr[-3:][::-1]) or '?'


def _wrap(owner, name, tag, sizer):
    orig = getattr(owner, name)

    @functools.wraps(orig)
    def w(*a, **k):
        out = orig(*a, **k)
        try:
            n = sizer(a, k, out)
        except Exception:
            n = -1
        if n != 0:
            e = LOG[(tag, _where())]
            e[0] += 1
            e[1] += max(n, 0)
        return out
    setattr(owner, name, w)


def _is_dev(t):
    return isinstance(t, torch.Tensor) and t.is_cuda


_wrap(torch.Tensor, 'copy_', 'copy_', lambda a, k, o: a[0].numel() * a[0].element_size() if _is_dev(a[0]) else 0)
_wrap(torch.Tensor, 'zero_', 'zero_', lambda a, k, o: a[0].numel() * a[0].element_size() if _is_dev(a[0]) else 0)
_wrap(torch.Tensor, 'fill_', 'fill_', lambda a, k, o: a[0].numel() * a[0].element_size() if _is_dev(a[0]) else 0)
_wrap(torch.Tensor, 'clone', 'clone', lambda a, k, o: o.numel() * o.element_size() if _is_dev(o) else 0)
_wrap(torch.Tensor, 'contiguous', 'contiguous', lambda a, k, o: (o.numel() * o.element_size()) if (_is_dev(o) and o.data_ptr() != a[0].data_ptr()) else 0)
_wrap(torch.Tensor, 'to', 'to', lambda a, k, o: (o.numel() * o.element_size()) if (_is_dev(o) and o.data_ptr() != a[0].data_ptr()) else 0)
_wrap(torch.Tensor, 'float', 'float', lambda a, k, o: (o.numel() * o.element_size()) if (_is_dev(o) and o.data_ptr() != a[0].data_ptr()) else 0)
for meth in ('add', 'add_', 'mul', 'mul_', 'sub', 'div', '__add__', '__mul__', '__sub__', '__truediv__', '__radd__', '__rmul__', 'sum', 'mean',
             'bfloat16', 'permute_copy'):
    if hasattr(torch.Tensor, meth):
        _wrap(torch.Tensor, meth, meth, lambda a, k, o: o.numel() * o.element_size() if _is_dev(o) else 0)
for fn in ('zeros', 'zeros_like', 'ones', 'ones_like', 'full', 'cat', 'stack', 'tensor', 'as_tensor'):
    _wrap(torch, fn, fn, lambda a, k, o: o.numel() * o.element_size() if _is_dev(o) else 0)
stepper.run()
torch.cuda.synchronize()
print('python-level copy / fill sources of one eager step (count, bytes):')
for (tag, where), (cnt, nb) in sorted(LOG.items(), key=lambda kv: -kv[1][1]):
    print(f'{tag:12s} x{cnt:3d} {nb / 1e6:10.3f} MB  {where}')
