"""Phase stamps of conv_bwd_fused_kernel (diagnostic build: scripts/diag/fb_variant.sh stamp -DFB_STAMP; DIS_HIP_LIB=build_variants/
libdis_hip_stamp.so): per-wave s_memtime sums (100 MHz ticks -> shader cycles via the clock estimate) of a tile's phases, plain form,
16 x 256 x 216 x 32."""
import ctypes, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from depthinspace_amd import ops
L = ops.lib
n, h, w, c = 16, 256, 216, 32
g_ = torch.Generator().manual_seed(0)
gq = torch.randn(n, h, w, c, generator=g_).cuda(); x = torch.randn(n, h, w, c, generator=g_).cuda()
wt = (torch.randn(c, c, 3, 3, generator=g_) * 0.05).cuda()
gx = torch.empty_like(x); gw = torch.empty(c, c, 3, 3, device='cuda'); gb = torch.empty(c, device='cuda')
ws = torch.empty(L.fn('dis_conv2d_bwd_fused_workspace')(c), dtype=torch.float32, device='cuda')
for _ in range(3):
    assert L.call_try('dis_conv2d_bwd_fused_f16x2', gq, None, None, 0, None, wt, c, c, wt.stride(0), gx, 0, None, None, None, x, None, None, None,
                      1e-5, gw, gb, ws, n, h, w, c, 0)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
L.call_try('dis_conv2d_bwd_fused_f16x2', gq, None, None, 0, None, wt, c, c, wt.stride(0), gx, 0, None, None, None, x, None, None, None, 1e-5, gw, gb, ws, n, h, w, c, 0)
e1.record(); torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * (256 * 4 * 12))()
lib = L.load()
lib.dis_debug_fb_stamps.argtypes = [ctypes.c_void_p]
assert lib.dis_debug_fb_stamps(ctypes.cast(buf, ctypes.c_void_p)) == 0
a = np.array(buf, dtype=np.float64).reshape(256, 4, 12)
tiles = n * (h // 16) * ((w + 15) // 16) / 256.0
names = ['loop control + flush of the channel sums', 'TOP: final values + maxima (halo, x strip)', 'barrier A', 'staging (halo + x tile split, LDS writes), next loads issued',
         'barrier B', 'D: 216 input-gradient products', 'W: epilogue + 216 dW products', 'after the loop: slab / bias',
         '  (staging: halo items split + written)', '  (staging: x strip split + written)',
         'prologue: first tile requested (once)', 'prologue: weights -> fp16 planes in LDS (once)']
print(f'launch {e0.elapsed_time(e1) * 1e3:.1f} us, {tiles:.1f} tiles per workgroup; s_memtime ticks (100 MHz) per tile and wave, median over workgroups')
tot = 0.0
for k in range(12):
    v = np.median(a[:, :, k]) / (tiles if k not in (7, 10, 11) else 1.0)
    tot += np.median(a[:, :, k]) if k < 8 else 0.0
    print(f'  {names[k]:40s} {v:9.1f} ticks = {v * 10:.0f} ns' + (' per tile' if k not in (7, 10, 11) else ' once'))
print(f'  sum {tot:.0f} ticks = {tot / 100:.1f} us per workgroup')
