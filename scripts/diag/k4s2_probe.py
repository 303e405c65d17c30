#!/usr/bin/env python
"""Does the general (DispNetS) family serve FuseNet's 4x4 stride-2 down convolution (Block2D3D.conv2_1, 32 -> 32) - forward and the
4-class input gradient in one launch - and how long does it take beside the fp32-MFMA kernels the step uses now?"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from depthinspace_amd import ops, lib
import torch.nn.functional as F
torch.manual_seed(0)
n, h, w, c = 16, 256, 216, 32
x = torch.randn(n, h, w, c, device='cuda')
wt = torch.randn(c, c, 4, 4, device='cuda') * 0.05
b = torch.randn(c, device='cuda') * 0.1
ho, wo = h // 2, w // 2
gy = torch.randn(n, ho, wo, c, device='cuda')


def timeit(f, reps=50):
    for _ in range(5):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


ref = F.conv2d(x.permute(0, 3, 1, 2).double(), wt.double(), b.double(), stride=2, padding=1).permute(0, 2, 3, 1)
y0 = torch.empty(n, ho, wo, c, device='cuda')
st = torch.zeros(2 * n, dtype=torch.float64, device='cuda')


def cur_fwd():
    ops._conv_fwd_any(x, wt, c, 0, b, y0, st, n, h, w, c, c, 4, 2, 1, ops.ACT_NONE)


cur_fwd()
print('current fwd  err', float((y0.double() - ref).abs().max()), 'us', timeit(cur_fwd))
y2 = torch.empty_like(y0)
f2 = lambda: lib.call('dis_conv2d_fwd_k4s2_f16x2', x, wt, b, y2, st, n, h, w, ops.ACT_NONE)
f2()
print('k4s2 f16x2 fwd err', float((y2.double() - ref).abs().max()), 'us', timeit(f2))
y1 = torch.empty_like(y0)
try:
    f = lambda: ops._convg_run(ops.CONVG_CONV, x, wt, b, y1, n, h, w, c, c, ho, wo, c, c, 4, 2, 1, ops.ACT_NONE)
    f()
    print('convg   fwd  err', float((y1.double() - ref).abs().max()), 'us', timeit(f), 'kernel', lib.last_kernel() if hasattr(lib, 'last_kernel') else '')
except Exception as e:
    print('convg fwd failed:', e)
# input gradient
xr = x.permute(0, 3, 1, 2).double().requires_grad_(True)
F.conv2d(xr, wt.double(), None, stride=2, padding=1).backward(gy.permute(0, 3, 1, 2).double())
gref = xr.grad.permute(0, 2, 3, 1)
g0 = torch.empty_like(x)
ws = torch.empty(16 * c * c, dtype=torch.float32, device='cuda')
cur_d = lambda: lib.call('dis_conv2d_dgrad_strided', gy, wt, g0, ws, n, h, w, c, c, 4, 2, 1, 0)
cur_d()
print('current dgrad err', float((g0.double() - gref).abs().max()), 'us', timeit(cur_d))
g2 = torch.empty_like(x)
fd = lambda: lib.call('dis_conv2d_dgrad_k4s2_f16x2', gy, wt, g2, n, h, w, 0)
fd()
print('k4s2 f16x2 dgrad err', float((g2.double() - gref).abs().max()), 'us', timeit(fd))
g1 = torch.empty_like(x)
try:
    f = lambda: ops._convg_run(ops.CONVG_CONV_DGRAD, gy, wt, None, g1, n, ho, wo, c, c, h, w, c, c, 4, 2, 1, ops.ACT_NONE)
    f()
    print('convg   dgrad err', float((g1.double() - gref).abs().max()), 'us', timeit(f))
except Exception as e:
    print('convg dgrad failed:', e)
