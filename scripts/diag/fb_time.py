import os, sys, torch
sys.path.insert(0, '/root/repo')
from depthinspace_amd import ops
L = ops.lib
n, h, w, c = 16, 256, 216, 32
g_ = torch.Generator().manual_seed(0)
gq = torch.randn(n, h, w, c, generator=g_).cuda(); x = torch.randn(n, h, w, c, generator=g_).cuda()
wt = (torch.randn(c, c, 3, 3, generator=g_) * 0.05).cuda()
gx = torch.empty_like(x); gw = torch.empty(c, c, 3, 3, device='cuda'); gb = torch.empty(c, device='cuda')
ws = torch.empty(L.fn('dis_conv2d_bwd_fused_workspace')(c), dtype=torch.float32, device='cuda')
def run():
    L.call_try('dis_conv2d_bwd_fused_f16x2', gq, None, None, 0, None, wt, c, c, wt.stride(0), gx, 0, None, None, None, x, None, None, None, 1e-5, gw, gb, ws, n, h, w, c, 0)
for _ in range(5): run()
ts = []
for r in range(7):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): run()
    e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) * 100)
print('plain fused launch + reduce: %.1f us' % sorted(ts)[3])
