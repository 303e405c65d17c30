"""Does the class-ordered Conv3D backward (latency-bound launches, 1.5 waves per SIMD) run in the shadow of the 3x3 conv kernels
when both are in flight on two streams?  Prints sequential vs concurrent time of the same two launch sequences."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from depthinspace_amd import ops

tl, bs, C = 4, 4, 32
g = torch.Generator(device='cuda').manual_seed(1)


def c3_setup(h, w, stride):
    yy, xx = torch.meshgrid(torch.arange(h, device='cuda', dtype=torch.float32), torch.arange(w, device='cuda', dtype=torch.float32),
                            indexing='ij')
    geom = torch.empty(tl, bs, h, w, tl, 4, device='cuda')
    z = 1.0 + 0.3 * torch.rand(tl, bs, h, w, tl, device='cuda', generator=g)
    geom[..., 0] = (xx[None, None, :, :, None] / w - 0.5) * z
    geom[..., 1] = (yy[None, None, :, :, None] / w - 0.5) * z
    geom[..., 2] = z
    geom[..., 3] = 1.0
    wf = torch.randn(tl, bs, h, w, tl, C, device='cuda', generator=g)
    ps = [torch.randn(s, device='cuda', generator=g) * 0.3 for s in ((16, 3), (16,), (32, 16), (32,), (32, 32))]
    idx = ops.conv3d_select(geom, stride)
    ho, wo = idx.shape[2:4]
    y = torch.empty((tl, bs, ho, wo, C), device='cuda')
    agg = torch.empty_like(y)
    args = (geom, wf, *ps, idx)
    ops.lib.call('dis_conv3d_knn_fwd_agg', *args, y, agg, tl, bs, h, w, stride)
    gy = torch.randn(y.shape, device='cuda', generator=g)
    accd = torch.empty(ops.lib.fn('dis_conv3d_knn_bwd_det_workspace')(tl, bs, h, w, stride), device='cuda')
    gw, gp = torch.zeros_like(wf), torch.empty(1632, device='cuda')
    return lambda: ops.lib.call('dis_conv3d_knn_bwd_det', *args, y, agg, gy, gw, gp, accd, tl, bs, h, w, stride)


c3a, c3b = c3_setup(256, 216, 2), c3_setup(128, 108, 1)
x = torch.randn(16, 256, 216, 32, device='cuda', generator=g)
wt = torch.randn(32, 32, 3, 3, device='cuda', generator=g) * 0.05
b = torch.zeros(32, device='cuda')


def convs(n=24):
    for _ in range(n):
        ops.conv2d(x, wt, b, 1, 1, ops.ACT_SELU)


def c3s(n=4):
    for _ in range(n):
        c3a(); c3b()


side = torch.cuda.Stream()
for name, fn in (('convs only', lambda: convs()), ('conv3d only', lambda: c3s()), ('sequential', lambda: (convs(), c3s())),
                 ('two streams', None)):
    for rep in range(3):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        if fn is not None:
            with torch.no_grad():
                fn()
        else:
            with torch.no_grad():
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    c3s()
                convs()
                torch.cuda.current_stream().wait_stream(side)
        e1.record()
        torch.cuda.synchronize()
    print(f'{name:12s} {e0.elapsed_time(e1):.3f} ms')
