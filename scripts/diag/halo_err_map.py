#!/usr/bin/env python
"""Where does convh2_kernel's error against fp64 sit?  7x7 32->32 conv, halo vs streaming form: per-row / per-column / per-channel
error maxima and the error of a run with the second operand plane zeroed out of the comparison (diagnostic)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, torch.nn.functional as F
from depthinspace_amd import ops
g = torch.Generator().manual_seed(5)
n, cin, cout, k, h, w = 2, 32, 32, int(sys.argv[1]) if len(sys.argv) > 1 else 7, 48, 48
x = torch.randn(n, cin, h, w, generator=g)
wt = torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5
b = torch.zeros(cout)
ref = F.conv2d(x.double(), wt.double(), None, padding=k // 2)
def run(minhw):
    os.environ['DIS_CONVG_HALO_MIN'] = str(minhw)
    y = ops.convg(x.permute(0, 2, 3, 1).contiguous().cuda(), wt.cuda(), b.cuda(), 1, k // 2, ops.ACT_NONE)
    return y.permute(0, 3, 1, 2).double().cpu()
for name, m in (('halo', 1), ('stream', 1 << 40)):
    e = (run(m) - ref).abs() / ref.abs().max()
    print(name, 'max %.2e mean %.2e' % (e.max(), e.mean()))
    print('  rows   ', ' '.join('%.0f' % (v * 1e8) for v in e.amax(dim=(0, 1, 3))))
    print('  cols   ', ' '.join('%.0f' % (v * 1e8) for v in e.amax(dim=(0, 1, 2))))
    print('  chans  ', ' '.join('%.0f' % (v * 1e8) for v in e.amax(dim=(0, 2, 3))))
    print('  images ', ' '.join('%.0f' % (v * 1e8) for v in e.amax(dim=(1, 2, 3))))

# the arithmetic both kernels are meant to do, with exact accumulation: x 2^sx = h1 + h2, w 2^sw = g1 + g2, y = (h1 g1 + h1 g2 + h2 g1)
def split(v, e):
    s = v.double() * 2.0 ** e
    h1 = s.half().double()
    h2 = (s - h1).half().double()
    return h1, h2
import math
def sexp(m):
    return 14 - math.frexp(m)[1] + 1
em = torch.zeros_like(ref)
ew = sexp(float(wt.abs().max()))
g1, g2 = split(wt, ew)
for i in range(n):
    ex = sexp(float(x[i].abs().max()))
    h1, h2 = split(x[i:i + 1], ex)
    em[i:i + 1] = (F.conv2d(h1, g1, None, padding=k // 2) + F.conv2d(h1, g2, None, padding=k // 2) + F.conv2d(h2, g1, None, padding=k // 2)) * 2.0 ** -(ex + ew)
sc = ref.abs().max()
print('emulation (exact accumulation) vs fp64: max %.2e mean %.2e' % (((em - ref).abs() / sc).max(), ((em - ref).abs() / sc).mean()))
for name, m in (('halo', 1), ('stream', 1 << 40)):
    e = (run(m) - em).abs() / sc
    print(name, 'vs emulation: max %.2e mean %.2e' % (e.max(), e.mean()))
