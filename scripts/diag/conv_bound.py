#!/usr/bin/env python
"""What does conv_f16x2_kernel<32, 32> wait for?  Knock-out builds of the kernel (scripts/diag/build_conv_variants.sh), all loaded
into ONE process and timed in interleaved rounds on the same tensors (cdna_hip_programming.md rule 24), random and zero-filled
data, with the in-kernel clock of every arm (s_memtime / s_memrealtime, MI355X_MICROARCH.md DVFS give-back item 6).

    python scripts/diag/conv_bound.py [n h w] > gpurun_out/conv_bound.txt
"""
import ctypes, glob, os, sys
import numpy as np
import torch

root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
n, h, w = (int(v) for v in sys.argv[1:4]) if len(sys.argv) >= 4 else (16, 256, 216)
P = ctypes.c_void_p
dev = 'cuda'
order = ['base', 'ko_store', 'ko_load', 'ko_mem', 'ko_mfma', 'ko_split', 'ko_epi', 'ko_compute', 'ko_mem_mfma']
if os.environ.get('VARIANTS'):   # VARIANTS=base,wait1,... : these arms only; arms that are not knock-outs are checked against `base`
    order = os.environ['VARIANTS'].split(',')
libs = {}
for v in order:
    f = os.path.join(root, 'build_variants', f'libf2_{v}.so')
    if os.path.exists(f):
        libs[v] = ctypes.CDLL(f)
torch.manual_seed(1)
xr = torch.randn(n, h, w, 32, device=dev)
xz = torch.zeros(n, h, w, 32, device=dev)
wt = torch.randn(32, 32, 3, 3, device=dev) * 0.05
b = torch.randn(32, device=dev)
y = torch.empty(n, h, w, 32, device=dev)
st = torch.zeros(2 * n, dtype=torch.float64, device=dev)
# a streaming copy of the same bytes for scale (read x, write y): what HBM gives a pure pass on this box
flops = 2.0 * n * h * w * 32 * 32 * 9
nbytes = 2.0 * n * h * w * 32 * 4


def call(L, x, act, stats):
    r = L.dis_conv2d_fwd_bf16x3_oihw(P(x.data_ptr()), P(wt.data_ptr()), 0, 32, 32, 0, P(b.data_ptr()), P(y.data_ptr()),
                                     P(st.data_ptr() if stats else 0), n, h, w, 32, 32, 3, 1, 1, act, P(0))
    assert r == 0, r


def clock(L):
    buf = np.zeros(512, dtype=np.uint64)
    assert L.dis_debug_f2_clk(buf.ctypes.data_as(P)) == 0
    c = buf.reshape(256, 2).astype(np.float64)
    ok = c[:, 1] > 0
    return float(np.median(c[ok, 0] / c[ok, 1]) * 100.0), float(np.median(c[ok, 0]))   # MHz, shader cycles of a workgroup


def time_arm(L, x, act, stats, reps):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        call(L, x, act, stats)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


ref = None
for v, L in libs.items():
    if v.startswith('ko_'):
        continue
    y.zero_()
    call(L, xr, 1, True)
    torch.cuda.synchronize()
    if ref is None:
        ref = y.clone()
    else:
        print(f'# {v}: max |y - y_base| = {float((y - ref).abs().max()):.3e} (|y| max {float(ref.abs().max()):.3f})')
print(f'# conv_f16x2_kernel<32,32> {n}x{h}x{w}: {flops/1e9:.2f} GFLOP, {nbytes/1e6:.1f} MB in + out per launch; tiles/CU '
      f'{n*((h+15)//16)*((w+15)//16)/256:.1f}')
for label, act, stats in (('plain (input gradient form: no activation, no statistics)', 0, False), ('SELU + GroupNorm statistics (forward form)', 1, True)):
    for dname, x in (('random', xr), ('zeros', xz)):
        # hold the clock where a training step holds it, then interleaved rounds
        for _ in range(3):
            for v, L in libs.items():
                time_arm(L, x, act, stats, 30)
        res = {v: [] for v in libs}
        clk = {v: [] for v in libs}
        for rnd in range(7):
            for v, L in libs.items():
                res[v].append(time_arm(L, x, act, stats, 40))
                clk[v].append(clock(L))
        print(f'\n## {label}, {dname} data')
        print('| arm | us/launch median (min) | TB/s of in+out | TFLOP/s fp32-equivalent | in-kernel clock MHz | workgroup cycles |')
        print('|---|---|---|---|---|---|')
        for v in libs:
            t = np.array(res[v])
            mhz = np.median([c[0] for c in clk[v]])
            cyc = np.median([c[1] for c in clk[v]])
            print(f'| {v} | {np.median(t):.1f} ({t.min():.1f}) | {nbytes/np.median(t)/1e6:.2f} | {flops/np.median(t)/1e6:.1f} | {mhz:.0f} | {cyc:.0f} |')
        sys.stdout.flush()
# a plain copy kernel of the same bytes, for scale
for _ in range(20):
    y.copy_(xr)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50):
    y.copy_(xr)
e1.record()
torch.cuda.synchronize()
t = e0.elapsed_time(e1) * 1e3 / 50
print(f'\n# torch copy of the same tensor: {t:.1f} us = {nbytes/t/1e6:.2f} TB/s')
