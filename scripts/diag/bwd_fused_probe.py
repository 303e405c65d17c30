"""Same-process A/B of dis_conv2d_bwd_fused_f16x2 against the two launches it replaces (input gradient + weight gradient, each
with its slab-reduce), per form, at FuseNet's core resolution (16 x 256 x 216 x 32) - HIP events, interleaved rounds.

    python scripts/diag/bwd_fused_probe.py [reps] [h w]
"""
import os
import sys
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from depthinspace_amd import ops

L = ops.lib
S = ops.ACT_SELU


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    h, w = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (256, 216)
    n, c = 16, 32
    g_ = torch.Generator().manual_seed(0)
    rnd = lambda *s: torch.randn(*s, generator=g_).cuda()
    q0 = rnd(n, h, w, c)
    gq = rnd(n, h, w, c)
    wt = (rnd(c, c, 3, 3) * 0.05).contiguous()
    x = rnd(n, h, w, c)
    xs = F.selu(x)
    ab_other = rnd(n, h, w, c)
    slots = L.fn('dis_conv2d_gnsums_slots')()
    gamma = (torch.rand(c, generator=g_) + 0.5).cuda()
    wsz = L.fn('dis_conv2d_wgrad_workspace')(c, c, 3, 1)
    xst = torch.stack([x.double().sum(dim=(1, 2, 3)), (x.double() ** 2).sum(dim=(1, 2, 3))], 1).reshape(-1).contiguous()
    xgam, xbet = (torch.rand(c, generator=g_) + 0.5).cuda(), (torch.randn(c, generator=g_) * 0.1).cuda()
    forms = {
        # name: (coef?, in_act, accum, ab_x, act_y, x, xgn)
        'plain (conv_fuse slices b, c)': (False, 0, False, None, None, x, False),
        'plain_act (conv4)': (False, S, False, None, None, x, False),
        'coef_sums_xgn (ResNetBlock conv2)': (True, 0, False, x, None, x, True),
        'coef_act_sums_xgn (conv1_2 / conv2_2)': (True, S, False, x, None, x, True),
        'coef_act_accum (res1 / ref_res1 conv1)': (True, S, True, None, None, x, False),
        'two_consumer (conv1_1)': (True, S, True, ab_other, None, x, False),
        'chain (res2/3, ref_res2/3 conv1)': (True, S, True, ab_other, xs, xs, False),
    }
    rows = []
    for name, (cf, in_act, accum, ab_x, act_y, xx, xgn) in forms.items():
        q = F.selu(q0) if in_act else q0
        coef = None
        if cf:
            st = torch.stack([q.double().sum(dim=(1, 2, 3)), (q.double() ** 2).sum(dim=(1, 2, 3))], 1).reshape(-1).contiguous()
            ab0 = torch.zeros(n, slots, 2, c, dtype=torch.float64, device='cuda')
            ab0[:, 0, 0] = gq.double().sum(dim=(1, 2))
            ab0[:, 0, 1] = (gq.double() * q.double()).sum(dim=(1, 2))
            coef = torch.empty(n * (c + 2) + 4 * n * c + 2, dtype=torch.float32, device='cuda')
            L.call('dis_gn_bwd_coef', st, gamma, ab0, slots, coef, torch.empty(c, device='cuda'), torch.empty(c, device='cuda'),
                   torch.zeros(2, dtype=torch.int32, device='cuda'), n, h * w, c, 1e-5)
        gx = torch.zeros(n, h, w, c, device='cuda')
        gpre = torch.empty_like(gq)
        ab = torch.zeros(n * slots * 2 * c, dtype=torch.float64, device='cuda') if ab_x is not None else None
        gw, gb = torch.empty(c, c, 3, 3, device='cuda'), torch.empty(c, device='cuda')
        ws = torch.empty(max(wsz, L.fn('dis_conv2d_bwd_fused_workspace')(c)), dtype=torch.float32, device='cuda')

        def old():
            if cf:
                L.call('dis_conv2d_dgrad_f16x2_gnb', gq, q, coef, in_act, gpre, wt, c, c, wt.stride(0), gx, 1 if accum else 0, ab_x, act_y,
                       ab, n, h, w, c)
                gp = gpre
            elif in_act:
                L.call('dis_conv2d_dgrad_bf16x3_act', gq, q, in_act, wt, c, c, wt.stride(0), gx, n, h, w, c, c, 1, 0)
                gp = None
            else:
                L.call('dis_conv2d_fwd_bf16x3_oihw', gq, wt, 1, c, c, wt.stride(0), None, gx, None, n, h, w, c, c, 3, 1, 1, 0)
                gp = gq
            if xgn:
                L.call('dis_conv2d_wgrad_bf16x3_gn', xx, xst, xgam, xbet, 1e-5, gp, gw, gb, ws, n, h, w, c, c, c, 3, 1, 1)
            elif gp is None:
                L.call('dis_conv2d_wgrad_bf16x3_act', xx, gq, q, in_act, gw, gb, ws, n, h, w, c, c, c, 3, 1, 1)
            else:
                L.call('dis_conv2d_wgrad_bf16x3', xx, gp, gw, gb, ws, n, h, w, c, c, c, 3, 1, 1)

        def new():
            ok = L.call_try('dis_conv2d_bwd_fused_f16x2', gq, q if (cf or in_act) else None, coef, in_act, None, wt, c, c, wt.stride(0), gx,
                            1 if accum else 0, ab_x, act_y, ab, xx, xst if xgn else None, xgam if xgn else None, xbet if xgn else None,
                            1e-5, gw, gb, ws, n, h, w, c, 0)
            assert ok

        t = {'old': [], 'new': []}
        for r in range(reps + 2):
            for k, fn in (('old', old), ('new', new)):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(5):
                    fn()
                e1.record()
                torch.cuda.synchronize()
                if r >= 2:
                    t[k].append(e0.elapsed_time(e1) / 5 * 1e3)
        o, nw = sorted(t['old'])[len(t['old']) // 2], sorted(t['new'])[len(t['new']) // 2]
        rows.append((name, o, nw))
        print(f'{name:48s} two launches {o:7.1f} us   fused {nw:7.1f} us   ratio {nw / o:5.2f}', flush=True)
    return rows


if __name__ == '__main__':
    main()
