"""Under GPU sharing (start two copies): the backward of the GroupNorm-on-load pair (ops._Conv2dGnIn, GN sums form) repeated on
fixed inputs - every output must repeat bit for bit (fixed summation orders).  Prints mismatch counts per output and shape."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from depthinspace_amd import ops

torch.manual_seed(0)
reps = int(os.environ.get('REPS', 150))
for (n, h, w, c) in ((16, 256, 216, 32), (16, 512, 432, 16), (16, 128, 108, 32), (16, 512, 432, 32)):
    x = torch.randn(n, h, w, c, device='cuda')
    wt = (torch.randn(c, c, 3, 3, device='cuda') * 0.05)
    gamma, beta = torch.rand(c, device='cuda') + 0.5, torch.randn(c, device='cuda') * 0.1
    gy = torch.randn(n, h, w, c, device='cuda')
    st = torch.stack([x.double().sum(dim=(1, 2, 3)), (x.double() ** 2).sum(dim=(1, 2, 3))], 1).reshape(-1).contiguous()
    slots = ops.lib.fn('dis_conv2d_gnsums_slots')()
    outs = None
    bad = {}
    for r in range(reps):
        gnorm, gx = torch.empty_like(x), torch.empty_like(x)
        gg, gb = torch.empty(c, device='cuda'), torch.empty(c, device='cuda')
        ab = torch.zeros(n * slots * 2 * c, dtype=torch.float64, device='cuda')
        ops.lib.call('dis_conv2d_dgrad_bf16x3_gnsums', gy, wt, c, c, wt.stride(0), gnorm, x, ab, n, h, w, c, c, 1)
        coef = torch.empty(n * (c + 2) + 4 * n * c + 2, dtype=torch.float32, device='cuda')
        ops.lib.call('dis_gn_bwd_from_sums', gnorm, x, st, gamma, ab, slots, gx, gg, gb, coef, n, h * w, c, 1e-5, ops.ACT_SELU)
        cur = {'gnorm': gnorm, 'ab': ab, 'gx': gx, 'gg': gg, 'gb': gb}
        if outs is None:
            outs = {k: v.clone() for k, v in cur.items()}
        else:
            for k, v in cur.items():
                if not torch.equal(v, outs[k]):
                    bad[k] = bad.get(k, 0) + 1
    torch.cuda.synchronize()
    print(f'pid {os.getpid()} shape {(n, h, w, c)}: {reps} repetitions, outputs that differed from the first run: {bad}', flush=True)
