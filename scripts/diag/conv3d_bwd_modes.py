"""Conv3D backward at the two shapes of the DIS-MF bs=4 step: float-atomic scatter vs class-ordered plain read-modify-write
(dis_conv3d_knn_bwd_det).  Prints ms per call and the distance between the two feature / parameter gradients."""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from depthinspace_amd import ops

libc = ctypes.CDLL(None)


def run(h, w, stride, reps=10, modes=('atomic', 'agg', 'det', 'det2')):
    tl, bs, C = 4, 4, 32
    g = torch.Generator(device='cuda').manual_seed(1)
    yy, xx = torch.meshgrid(torch.arange(h, device='cuda', dtype=torch.float32), torch.arange(w, device='cuda', dtype=torch.float32),
                            indexing='ij')
    geom = torch.empty(tl, bs, h, w, tl, 4, device='cuda')
    z = 1.0 + 0.3 * torch.rand(tl, bs, h, w, tl, device='cuda', generator=g)
    geom[..., 0] = (xx[None, None, :, :, None] / w - 0.5) * z
    geom[..., 1] = (yy[None, None, :, :, None] / w - 0.5) * z
    geom[..., 2] = z
    geom[..., 3] = (torch.rand(tl, bs, h, w, tl, device='cuda', generator=g) > 0.1).float()
    wf = torch.randn(tl, bs, h, w, tl, C, device='cuda', generator=g)
    ps = [torch.randn(s, device='cuda', generator=g) * 0.3 for s in ((16, 3), (16,), (32, 16), (32,), (32, 32))]
    idx = ops.conv3d_select(geom, stride)
    ho, wo = idx.shape[2:4]
    y = torch.empty((tl, bs, ho, wo, C), device='cuda')
    args = (geom, wf, *ps, idx)
    agg = torch.empty_like(y)
    ops.lib.call('dis_conv3d_knn_fwd_agg', *args, y, agg, tl, bs, h, w, stride)
    y0 = torch.empty_like(y)
    ops.lib.call('dis_conv3d_knn_fwd', *args, y0, tl, bs, h, w, stride)
    assert torch.equal(y, y0)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        ops.lib.call('dis_conv3d_knn_fwd_agg', *args, y, agg, tl, bs, h, w, stride)
    e1.record()
    torch.cuda.synchronize()
    print(f'h={h} w={w} stride={stride} forward (+ aggregate kept): {e0.elapsed_time(e1) / reps:.3f} ms per call', flush=True)
    gy = torch.randn(y.shape, device='cuda', generator=g)
    acc = torch.empty(ops.lib.fn('dis_conv3d_knn_bwd_workspace')(), device='cuda')
    accd = torch.empty(ops.lib.fn('dis_conv3d_knn_bwd_det_workspace')(tl, bs, h, w, stride), device='cuda')
    base = torch.randn(wf.shape, device='cuda', generator=g)

    def bwd(mode, gw, gp):
        if mode == 'atomic':
            ops.lib.call('dis_conv3d_knn_bwd', *args, y, gy, gw, gp, acc, tl, bs, h, w, stride)
        elif mode == 'agg':
            ops.lib.call('dis_conv3d_knn_bwd_agg', *args, y, agg, gy, gw, gp, accd, tl, bs, h, w, stride)
        else:
            ops.lib.call('dis_conv3d_knn_bwd_det', *args, y, agg, gy, gw, gp, accd, tl, bs, h, w, stride)

    res = {}
    for mode in modes:
        gw, gp = base.clone(), torch.empty(1632, device='cuda')
        bwd(mode, gw, gp)
        res[mode] = (gw, gp.clone())
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            bwd(mode, gw, gp)
        e1.record()
        torch.cuda.synchronize()
        print(f'h={h} w={w} stride={stride} {mode}: {e0.elapsed_time(e1) / reps:.3f} ms per call', flush=True)
    if len(modes) < 4:
        return
    sc = float((res['atomic'][0] - base).abs().max())
    print('   feature gradient: max |det - atomic| / scale =', float((res['det'][0] - res['atomic'][0]).abs().max()) / sc,
          ' parameters:', float((res['det'][1] - res['atomic'][1]).abs().max() / res['atomic'][1].abs().max()),
          ' agg:', float((res['agg'][0] - res['atomic'][0]).abs().max()) / sc,
          float((res['agg'][1] - res['atomic'][1]).abs().max() / res['atomic'][1].abs().max()),
          ' det repeats bitwise:', torch.equal(res['det'][0], res['det2'][0]) and torch.equal(res['det'][1], res['det2'][1]))




def stamps(mode='det', h=128, w=108, stride=1):
    """phase stamps of block C3_STAMP, wave 0, first group (library built with -DC3_STAMP=<block>; DIS_HIP_LIB points at it)"""
    import numpy as np
    from depthinspace_amd import lib
    dll = lib.load()
    run(h, w, stride, reps=1, modes=(mode,))
    torch.cuda.synchronize()
    out = (ctypes.c_ulonglong * 64)()
    dll.dis_c3_stamps(out)
    t = np.array(list(out), dtype=np.int64)
    t0 = t[0]
    names = {0: 'start', 1: 'prologue done', 2: 'ids+centre -> rows', 3: 'geometry issued, gy/y/agg', 4: 'first rows issued, D written',
             5: 'dagg + dW done', 6: 'group done', 7: 'loop done'}
    for k in (0, 1, 2, 3, 4, 5):
        print(f'  {names[k]:34s} {t[k] - t0:8d}')
    for n in range(9):
        a = t[8 + 4 * n: 12 + 4 * n] - t0
        print(f'  nb {n}: top {a[0]:8d}  fetch-issued {a[1]:8d}  mlp {a[2]:8d}  grads {a[3]:8d}')
    print(f'  {names[6]:34s} {t[6] - t0:8d}\n  {names[7]:34s} {t[7] - t0:8d}')


if __name__ == '__main__':
    if len(sys.argv) > 1 and sys.argv[1] == '--stamps':
        stamps(*(sys.argv[2:3] or ['det']))
    else:
        run(256, 216, 2)
        run(128, 108, 1)
