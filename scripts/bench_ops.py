#!/usr/bin/env python
"""Micro-benchmark of single C-ABI entry points at the DIS-MF bs=4 shapes (HIP events, median of N launches).
    python scripts/bench_ops.py [filter]
"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from depthinspace_amd import lib, ops

dev = 'cuda'


def timeit(fn, n=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ts.sort()
    return ts[len(ts) // 2]


def conv_case(name, n, h, w, cin, cout, k, stride, pad, act, stats):
    x = torch.randn(n, h, w, cin, device=dev)
    wt = torch.randn(cout, cin, k, k, device=dev) * 0.05
    b = torch.randn(cout, device=dev)
    ho, wo = (h + 2 * pad - k) // stride + 1, (w + 2 * pad - k) // stride + 1
    y = torch.empty(n, ho, wo, cout, device=dev)
    wp = ops._pack_w(wt, cin, 0)
    st = torch.zeros(2 * n, dtype=torch.float64, device=dev) if stats else None
    fl = 2.0 * n * ho * wo * cin * cout * k * k
    t = timeit(lambda: lib.call('dis_conv2d_fwd', x, wp, b, y, st, n, h, w, cin, cout, k, stride, pad, act))
    print(f'{name:34s} fwd   {t*1e3:8.1f} us  {fl/t/1e9:7.1f} TFLOP/s')
    if (cin, cout, k, stride) == (32, 32, 3, 1):
        pk = torch.empty(9 * 3 * 4 * 32 * 8, dtype=torch.int16, device=dev)
        lib.call('dis_conv2d_pack_weights_bf16x3', wt, pk, 32, 32, 3, 0)
        y2 = torch.empty_like(y)
        st2 = torch.zeros(2 * n, dtype=torch.float64, device=dev) if stats else None
        t = timeit(lambda: lib.call('dis_conv2d_fwd_bf16x3', x, pk, b, y2, st2, n, h, w, cin, cout, k, stride, pad, act))
        ref = torch.nn.functional.conv2d(x.permute(0, 3, 1, 2).double(), wt.double(), b.double(), padding=pad)
        ref = torch.nn.functional.selu(ref) if act == 1 else ref
        ref = ref.permute(0, 2, 3, 1)
        e32 = float((y.double() - ref).abs().max() / ref.abs().max())
        ex3 = float((y2.double() - ref).abs().max() / ref.abs().max())
        print(f'{name:34s} fwd3  {t*1e3:8.1f} us  {fl/t/1e9:7.1f} TFLOP/s-equivalent   max rel err vs fp64: fp32-MFMA {e32:.2e}  bf16x3 {ex3:.2e}')
    gy = torch.randn_like(y)
    gw = torch.empty_like(wt); gb = torch.empty(cout, device=dev)
    wsz = lib.fn('dis_conv2d_wgrad_workspace')(cin, cout, k, stride)
    ws = torch.empty(wsz, device=dev)
    t = timeit(lambda: lib.call('dis_conv2d_wgrad', x, gy, gw, gb, ws, n, h, w, cin, cin, cout, k, stride, pad))
    print(f'{name:34s} wgrad {t*1e3:8.1f} us  {fl/t/1e9:7.1f} TFLOP/s')
    if (cin, cout, k, stride) == (32, 32, 3, 1):
        gw2 = torch.empty_like(wt); gb2 = torch.empty(cout, device=dev)
        t = timeit(lambda: lib.call('dis_conv2d_wgrad_bf16x3', x, gy, gw2, gb2, ws, n, h, w, cin, cin, cout, k, stride, pad))
        xr = x.permute(0, 3, 1, 2).double().requires_grad_(False)
        wr = wt.double().clone().requires_grad_(True)
        br = b.double().clone().requires_grad_(True)
        torch.nn.functional.conv2d(xr, wr, br, padding=pad).backward(gy.permute(0, 3, 1, 2).double())
        e32 = float((gw.double() - wr.grad).abs().max() / wr.grad.abs().max())
        ex3 = float((gw2.double() - wr.grad).abs().max() / wr.grad.abs().max())
        eb = float((gb2.double() - br.grad).abs().max() / br.grad.abs().max())
        print(f'{name:34s} wgrd3 {t*1e3:8.1f} us  {fl/t/1e9:7.1f} TFLOP/s-equivalent   max rel err vs fp64: fp32-MFMA {e32:.2e}  bf16x3 {ex3:.2e}  bias {eb:.2e}')


def main():
    flt = sys.argv[1] if len(sys.argv) > 1 else ''
    cases = [
        ('c32x32 k3 full (ref_res)', 16, 512, 432, 32, 32, 3, 1, 1, 1, True),
        ('c32x32 k3 core (res/conv1_x)', 16, 256, 216, 32, 32, 3, 1, 1, 1, True),
        ('c16x16 k3 full (amb_res)', 16, 512, 432, 16, 16, 3, 1, 1, 1, True),
        ('c96x32 k3 core (conv_fuse)', 16, 256, 216, 96, 32, 3, 1, 1, 0, True),
        ('c48x32 k3 full (ref_conv)', 16, 512, 432, 48, 32, 3, 1, 1, 1, False),
        ('c128x32 k1 core (conv_mf)', 16, 256, 216, 128, 32, 1, 1, 0, 0, True),
        ('c32x32 k4s2 core (conv2_1)', 16, 256, 216, 32, 32, 4, 2, 1, 1, True),
        ('c32x16 k3 full (final_conv)', 16, 512, 432, 32, 16, 3, 1, 1, 1, False),
    ]
    for c in cases:
        if flt in c[0]:
            conv_case(*c)


if __name__ == '__main__':
    main()
