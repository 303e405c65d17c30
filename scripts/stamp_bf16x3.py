#!/usr/bin/env python
"""Phase split of conv_bf16x3_kernel from in-kernel s_memtime stamps (diagnostic build: `make -C depthinspace_amd/csrc stamp`).
    python scripts/stamp_bf16x3.py [n h w]
Prints, per phase, the median over waves of the cycles spent per tile, and the share of the kernel.
"""
import ctypes, os, sys
import numpy as np
import torch

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
L = ctypes.CDLL(os.path.join(root, 'depthinspace_amd', os.environ.get('BX_STAMP_LIB', 'libdis_hip_stamp.so')))
n, h, w = (int(v) for v in sys.argv[1:4]) if len(sys.argv) >= 4 else (6, 512, 432)
dev = 'cuda'
x = torch.randn(n, h, w, 32, device=dev)
wt = torch.randn(32, 32, 3, 3, device=dev) * 0.05
b = torch.randn(32, device=dev)
y = torch.empty(n, h, w, 32, device=dev)
st = torch.zeros(2 * n, dtype=torch.float64, device=dev)
pk = torch.empty(9 * 3 * 4 * 32 * 8, dtype=torch.int16, device=dev)
P = ctypes.c_void_p
L.dis_conv2d_pack_weights_bf16x3(P(wt.data_ptr()), P(pk.data_ptr()), 32, 32, 3, 0, P(0))


F2 = len(sys.argv) > 4 and sys.argv[4] == 'f16x2'   # python scripts/stamp_bf16x3.py n h w f16x2: the two-term fp16 kernel


def run(act, stats):
    if F2:
        return L.dis_conv2d_fwd_bf16x3_oihw(P(x.data_ptr()), P(wt.data_ptr()), 0, 32, 32, 0, P(b.data_ptr()), P(y.data_ptr()),
                                            P(st.data_ptr() if stats else 0), n, h, w, 32, 32, 3, 1, 1, act, P(0))
    return L.dis_conv2d_fwd_bf16x3(P(x.data_ptr()), P(pk.data_ptr()), P(b.data_ptr()), P(y.data_ptr()),
                                   P(st.data_ptr() if stats else 0), n, h, w, 32, 32, 3, 1, 1, act, P(0))


names = ['tile setup (first: weight copy)', 'barrier A (skew)', 'stage (wait halo+split+ds_write)', 'barrier B',
         'next-tile coordinates', 'MFMA loop + halo issue + deferred epilogue', 'hand-over', 'last epilogue + stats']
if F2:
    names = ['tile setup (first: weight split)', 'prep (wait halo, values, wave max)', 'barrier A (skew)',
             'stage (scale + split + ds_write)', 'barrier B', 'MFMA loop + halo issue + deferred epilogue',
             'hand-over (descale + bias)', 'last epilogue + stats']
for act, stats, label in ((1, True, 'fwd: SELU + GN stats'), (0, False, 'dgrad: no act, no stats'),
                          (0x100, False, 'dgrad accumulating into y')):
    for _ in range(200):  # hold the clock where a training step holds it
        assert run(act, stats) == 0
    torch.cuda.synchronize()
    buf = np.zeros(256 * 8 * 8, dtype=np.uint64)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); run(act, stats); e1.record()
    torch.cuda.synchronize()
    assert (L.dis_debug_f2_stamps if F2 else L.dis_debug_bx_stamps)(buf.ctypes.data_as(P)) == 0
    s = buf.reshape(256, 8, 8).astype(np.float64)
    tiles = n * ((h + 15) // 16) * ((w + 15) // 16)
    per_cu = tiles / float(os.environ.get('BX_GRID', 256))
    tot = s.sum(axis=2)
    print(f'== {label}: {n}x{h}x{w}, {tiles} tiles ({per_cu:.1f}/CU), launch {e0.elapsed_time(e1)*1e3:.1f} us, '
          f'median wave {np.median(tot):.0f} memtime ticks (100 MHz x {np.median(tot)/(e0.elapsed_time(e1)*1e3)/100:.2f})')
    for k, nm in enumerate(names):
        v = s[:, :, k]
        print(f'  {nm:38s} median {np.median(v)/per_cu:9.1f} ticks/tile   share {v.sum()/tot.sum()*100:5.1f} %   '
              f'(wave min {v.min()/per_cu:8.1f} max {v.max()/per_cu:8.1f})')
