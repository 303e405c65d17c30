"""Fixed cost per launch vs per-tile cost of the 32->32 3x3 conv kernels: time the launch at several batch sizes (tiles per
workgroup = n * 224 / 256 at 256x216) and fit a line.   python scripts/diag/conv_fixed_cost.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from depthinspace_amd import lib

h, w = 256, 216
wt = (torch.randn(32, 32, 3, 3) * 0.05).cuda()
b = torch.randn(32).cuda()
for split in (1, 0):
    lib.fn('dis_set_conv_split')(split)
    for act, stats, label in ((1, True, 'SELU+stats'), (1, False, 'SELU only '), (0, True, 'stats only'), (0, False, 'plain     ')):
        pts = []
        for n in (8, 16, 32, 64):
            x = torch.randn(n, h, w, 32, device='cuda')
            y = torch.empty_like(x)
            st = torch.zeros(2 * n, dtype=torch.float64, device='cuda')
            def run():
                lib.call('dis_conv2d_fwd_bf16x3_oihw', x, wt, 0, 32, 32, 0, b, y, st if stats else None, n, h, w, 32, 32, 3, 1, 1, act)
            for _ in range(100):
                run()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(50):
                run()
            e1.record()
            torch.cuda.synchronize()
            pts.append((n * 224 / 256.0, e0.elapsed_time(e1) / 50 * 1e3))
        t = np.array(pts)
        k, c = np.polyfit(t[:, 0], t[:, 1], 1)
        print(('f16x2 ' if split else 'bf16x3'), label, 'us per launch:', [round(v, 1) for v in t[:, 1]], ' per tile %.2f us, fixed %.1f us' % (k, c))
