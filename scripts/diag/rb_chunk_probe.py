#!/usr/bin/env python
"""Would sample-group launches keep a ResNetBlock's tensors in the 256 MB Infinity Cache?  GroupNorm(1 group) is per sample, so a
group of samples can run through conv -> GN -> conv -> GN + residual (and its backward) independently of the others.  Times the
block's forward + backward on the whole batch against the same work in groups of 8 / 4 / 2 samples (eager launches, HIP events)."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from depthinspace_amd import ops
from depthinspace_amd.model import multi_frame_networks as mfn
torch.manual_seed(0)
for (n, h, w, c) in ((16, 256, 216, 32), (16, 512, 432, 32)):
    blk1, blk2 = mfn.ResNetBlock(c).cuda(), mfn.ResNetBlock(c).cuda()
    for p in list(blk1.parameters()) + list(blk2.parameters()):
        if p.dim() > 1:
            torch.nn.init.normal_(p, std=0.05)
    x = torch.randn(n, h, w, c, device='cuda')
    go = torch.randn(n, h, w, c, device='cuda')

    def run(group):
        for i in range(0, n, group):
            ops.begin_step('cuda:0')
            xi = x[i:i + group].clone().requires_grad_(True)
            y = blk2(blk1(xi))
            y.backward(go[i:i + group])

    for group in (16, 8, 4, 2):
        for _ in range(3):
            run(group)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            run(group)
        e1.record()
        torch.cuda.synchronize()
        print(f'{n}x{h}x{w}x{c}: groups of {group:2d}: {e0.elapsed_time(e1) / 5:.3f} ms per 2 ResNetBlocks fwd + bwd', flush=True)
