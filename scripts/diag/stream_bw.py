"""What streaming (read-once / write-once) kernels reach on this box: a plain device copy, torch's elementwise add, and the GroupNorm
forward apply (dis_gn_apply) at the shapes of the DIS-MF step - the yardstick for the HBM-bound elementwise families of the step
(gn_*, resize, feature warps).  GB/s = (bytes read + bytes written) / time."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from depthinspace_amd import ops
L = ops.lib


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3   # us


for name, (n, h, w, c) in {'quarter 16x128x108x32': (16, 128, 108, 32), 'core 16x256x216x32': (16, 256, 216, 32),
                           'full 16x512x432x16': (16, 512, 432, 16), 'wf 16x256x216x128': (16, 256, 216, 128)}.items():
    x = torch.randn(n, h, w, c, device='cuda')
    y = torch.empty_like(x)
    r = torch.randn(n, h, w, c, device='cuda')
    nbytes = x.numel() * 4
    t_copy = timed(lambda: y.copy_(x))
    t_add = timed(lambda: torch.add(x, r, out=y))
    line = f'{name:24s} {nbytes / 1e6:7.1f} MB  copy {t_copy:7.1f} us = {2 * nbytes / t_copy / 1e3:6.0f} GB/s   add {t_add:7.1f} us = {3 * nbytes / t_add / 1e3:6.0f} GB/s'
    if c <= 64:
        stats = torch.zeros(2 * n, dtype=torch.float64, device='cuda')
        m = float(h * w * c)
        stats[0::2] = 0.1 * m
        stats[1::2] = 1.5 * m
        gam, bet = torch.ones(c, device='cuda'), torch.zeros(c, device='cuda')
        t_gn = timed(lambda: L.call('dis_gn_apply', x, stats, gam, bet, None, y, n, h * w, c, 0, 1e-5))
        t_gnr = timed(lambda: L.call('dis_gn_apply', x, stats, gam, bet, r, y, n, h * w, c, ops.ACT_SELU, 1e-5))
        line += f'   gn_apply {t_gn:7.1f} us = {2 * nbytes / t_gn / 1e3:6.0f} GB/s   gn_apply+res+selu {t_gnr:7.1f} us = {3 * nbytes / t_gnr / 1e3:6.0f} GB/s'
    print(line, flush=True)
