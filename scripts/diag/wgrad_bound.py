#!/usr/bin/env python
"""What does conv_wgrad_f16x2_kernel<32, 32> wait for?  Knock-out builds (scripts/diag/build_conv_variants.sh: w_ko_*), one process,
interleaved rounds, random and zero data, in-kernel clock - the weight-gradient twin of scripts/diag/conv_bound.py.
    VARIANTS=pf2,w_ko_mfma,... python scripts/diag/wgrad_bound.py [n h w]"""
import ctypes, os, sys
import numpy as np
import torch

root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
n, h, w = (int(v) for v in sys.argv[1:4]) if len(sys.argv) >= 4 else (16, 256, 216)
P = ctypes.c_void_p
order = os.environ.get('VARIANTS', 'pf2,w_ko_mfma,w_ko_load,w_ko_both,w_ko_split').split(',')
libs = {}
for v in order:
    f = os.path.join(root, 'build_variants', f'libf2_{v}.so')
    if os.path.exists(f):
        libs[v] = ctypes.CDLL(f)
        libs[v].dis_conv2d_wgrad_workspace.restype = ctypes.c_long
torch.manual_seed(1)
c = 32
xr, gr = torch.randn(n, h, w, c, device='cuda'), torch.randn(n, h, w, c, device='cuda')
xz, gz = torch.zeros_like(xr), torch.zeros_like(gr)
gw, gb = torch.empty(c, c, 3, 3, device='cuda'), torch.empty(c, device='cuda')
first = next(iter(libs.values()))
ws = torch.empty(first.dis_conv2d_wgrad_workspace(c, c, 3, 1), dtype=torch.float32, device='cuda')
flops = 2.0 * n * h * w * c * c * 9
nbytes = 2.0 * n * h * w * c * 4


def call(L, x, g):
    r = L.dis_conv2d_wgrad_bf16x3(P(x.data_ptr()), P(g.data_ptr()), P(gw.data_ptr()), P(gb.data_ptr()), P(ws.data_ptr()), n, h, w, c, c, c,
                                  3, 1, 1, P(0))
    assert r == 0, r


def clock(L):
    buf = np.zeros(512, dtype=np.uint64)
    assert L.dis_debug_f2_clk(buf.ctypes.data_as(P)) == 0
    cc = buf.reshape(256, 2).astype(np.float64)
    ok = cc[:, 1] > 0
    return float(np.median(cc[ok, 0] / cc[ok, 1]) * 100.0), float(np.median(cc[ok, 0]))


def time_arm(L, x, g, reps):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        call(L, x, g)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


print(f'# conv_wgrad_f16x2_kernel<32,32> + slab reduce, {n}x{h}x{w}: {flops/1e9:.2f} GFLOP, {nbytes/1e6:.1f} MB of x + gy per launch')
for dname, x, g in (('random', xr, gr), ('zeros', xz, gz)):
    for _ in range(3):
        for v, L in libs.items():
            time_arm(L, x, g, 30)
    res, clk = {v: [] for v in libs}, {v: [] for v in libs}
    for rnd in range(7):
        for v, L in libs.items():
            res[v].append(time_arm(L, x, g, 40))
            clk[v].append(clock(L))
    print(f'\n## {dname} data')
    print('| arm | us/call median (min) | TB/s of x + gy | TFLOP/s fp32-equivalent | in-kernel clock MHz | workgroup cycles |')
    print('|---|---|---|---|---|---|')
    for v in libs:
        t = np.array(res[v])
        print(f'| {v} | {np.median(t):.1f} ({t.min():.1f}) | {nbytes/np.median(t)/1e6:.2f} | {flops/np.median(t)/1e6:.1f} | '
              f'{np.median([q[0] for q in clk[v]]):.0f} | {np.median([q[1] for q in clk[v]]):.0f} |')
    sys.stdout.flush()
