#!/usr/bin/env python
"""gather_warped_feat forward / CSR backward: tiled kernels (default) against the grid-stride kernels (DIS_GATHER_TILED=0, read per call),
bit comparison and timing at the two resolutions of a DIS-MF step.   python scripts/diag/gather_probe.py > gpurun_out/gather_probe.txt"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from depthinspace_amd import ops
L = ops.lib


def t_of(fn, reps=30):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


tl, bs, c = 4, 4, 32
for h, w in ((256, 216), (128, 108)):
    g = torch.Generator().manual_seed(h)
    feat = torch.randn(tl, bs, h, w, c, generator=g).cuda()
    # smooth flows of a few pixels, like the step's
    base = torch.randn(tl * tl, bs, h // 16 + 2, w // 16 + 2, 2, generator=g) * 6
    flows = torch.nn.functional.interpolate(base.permute(0, 1, 4, 2, 3).reshape(-1, 2, h // 16 + 2, w // 16 + 2), size=(h, w),
                                            mode='bilinear', align_corners=True).reshape(tl * tl, bs, 2, h, w).permute(0, 1, 3, 4, 2).contiguous().cuda()
    go = torch.randn(tl, bs, h, w, tl, c, generator=g).cuda()
    csr = ops.gather_csr(flows)
    res = {}
    for mode in ('1', '0'):
        os.environ['DIS_GATHER_TILED'] = mode
        out = torch.empty(tl, bs, h, w, tl, c, device='cuda')
        gf = torch.empty(tl, bs, h, w, c, device='cuda')
        tf = t_of(lambda: L.call('dis_gather_warped_feat_fwd', feat, flows, out, tl, bs, h, w, c))
        tb = t_of(lambda: L.call('dis_gather_warped_feat_bwd_csr', go, csr, None, gf, tl, bs, h, w, c))
        res[mode] = (out.clone(), gf.clone())
        print(f'{h}x{w} tiled={mode}: fwd {tf:.1f} us, bwd_csr {tb:.1f} us')
    print('   bit-identical:', torch.equal(res['1'][0], res['0'][0]), torch.equal(res['1'][1], res['0'][1]))
os.environ.pop('DIS_GATHER_TILED')
