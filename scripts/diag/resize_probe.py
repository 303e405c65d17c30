#!/usr/bin/env python
"""resize_nhwc forward / backward: the tiled kernels against the grid-stride kernels they replace (reached through tensors that are
not 16-byte aligned), at the two shapes of a DIS-MF step.   python scripts/diag/resize_probe.py > gpurun_out/resize_probe.txt"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from depthinspace_amd import ops
L = ops.lib


def unaligned(shape):
    buf = torch.randn(int(torch.tensor(shape).prod()) + 1, device='cuda')
    return buf[1:].view(shape)


def t_of(fn, reps=50):
    for _ in range(10):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


for n, hin, win in ((16, 128, 108), (16, 256, 216)):
    ho, wo, c = 2 * hin, 2 * win, 32
    mb = (n * hin * win * c + n * ho * wo * c) * 4 / 1e6
    for name, mk in (('tiled', lambda s: torch.randn(s, device='cuda')), ('grid-stride', unaligned)):
        x, y = mk((n, hin, win, c)), mk((n, ho, wo, c))
        tf = t_of(lambda: L.call('dis_resize_bilinear_nhwc_fwd', x, y, n, hin, win, ho, wo, c, 1))
        tb = t_of(lambda: L.call('dis_resize_bilinear_nhwc_bwd', y, x, n, hin, win, ho, wo, c, 1))
        print(f'{n}x{hin}x{win} -> {ho}x{wo} x{c} {name}: fwd {tf:.1f} us ({mb/tf:.2f} TB/s of in + out), bwd {tb:.1f} us ({mb/tb:.2f} TB/s)')
