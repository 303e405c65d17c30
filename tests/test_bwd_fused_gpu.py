"""dis_conv2d_bwd_fused_f16x2 (round 6, csrc/conv_bwd_fused.hip): the input gradient and the weight gradient of a 3x3 conv 32 -> 32 in
ONE launch, against the two launches it replaces and against fp64.

Reference semantics: torch.nn.Conv2d's backward inside ResNetBlock / Block2D3D (/root/reference/model/multi_frame_networks.py:338-345,
514-542).  Bars: gx BIT-identical to the unfused input-gradient launch (same split, same per-tile scales, same accumulation order);
grad_w / grad_b within 1e-6 of the largest entry of the fp64 result (the bar of the kernel replaced, checked next to it); the
GroupNorm-backward channel sums within 1e-6 relative (other partial-sum order: 4 waves x 4 rows instead of 8 x 2)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

FORMS = ['plain', 'plain_accum', 'plain_act', 'coef', 'coef_act', 'coef_sums_xgn', 'coef_act_sums_xgn', 'coef_sums_xgn_store',
         'coef_act_accum', 'two_consumer', 'chain']


def _fp64_wgrad(x, gpre):
    xn = x.permute(0, 3, 1, 2).double()
    gn = gpre.permute(0, 3, 1, 2).double()
    gw = torch.nn.grad.conv2d_weight(xn, (gn.shape[1], xn.shape[1], 3, 3), gn, padding=1)
    return gw, gn.sum(dim=(0, 2, 3))


@pytest.mark.parametrize('form', FORMS)
@pytest.mark.parametrize('n,h,w', [(3, 37, 29), (2, 64, 48), (2, 16, 250), (1, 20, 20)])
def test_bwd_fused_matches_the_two_launches(form, n, h, w):
    from depthinspace_amd import ops
    L = ops.lib
    if L.fn('dis_get_conv_split')() != 1:
        pytest.skip('two-term fp16 kernels only')
    c = 32
    S = ops.ACT_SELU
    g_ = torch.Generator().manual_seed(1000 + 7 * h + w + len(form))
    rnd = lambda *s: torch.randn(*s, generator=g_).cuda()
    coef_form = form.startswith(('coef', 'two_consumer', 'chain'))
    in_act = S if form in ('plain_act', 'coef_act', 'coef_act_sums_xgn', 'coef_act_accum', 'two_consumer', 'chain') else 0
    accum = form in ('plain_accum', 'coef_act_accum', 'two_consumer', 'chain')
    sums = 'sums' in form or form in ('two_consumer', 'chain')
    xgn = 'xgn' in form
    store = form.endswith('store')
    q = rnd(n, h, w, c)
    if in_act:
        q = F.selu(q)
    gq = rnd(n, h, w, c) * (1.0 + 3.0 * torch.rand(n, 1, 1, 1, generator=g_).cuda())   # per-sample magnitudes differ: the running scales move
    wt = (rnd(c, c, 3, 3) * 0.05).contiguous()
    x = rnd(n, h, w, c) * 2.0 + 0.3
    if form == 'chain':
        x = F.selu(x)          # x = SELU(GroupNorm(x2) + res): the activation output the result is multiplied with
    slots = L.fn('dis_conv2d_gnsums_slots')()
    base = rnd(n, h, w, c)
    # ---- the operand: coefficients of a GroupNorm backward, or gy itself
    coef = None
    if coef_form:
        gamma = (torch.rand(c, generator=g_) + 0.5).cuda()
        st = torch.stack([q.double().sum(dim=(1, 2, 3)), (q.double() ** 2).sum(dim=(1, 2, 3))], 1).reshape(-1).contiguous()
        ab0 = torch.zeros(n, slots, 2, c, dtype=torch.float64, device='cuda')
        ab0[:, 0, 0] = gq.double().sum(dim=(1, 2))
        ab0[:, 0, 1] = (gq.double() * q.double()).sum(dim=(1, 2))
        coef = torch.empty(n * (c + 2) + 4 * n * c + 2, dtype=torch.float32, device='cuda')
        gg, gb_ = torch.empty(c, device='cuda'), torch.empty(c, device='cuda')
        L.call('dis_gn_bwd_coef', st, gamma, ab0, slots, coef, gg, gb_, torch.zeros(2, dtype=torch.int32, device='cuda'), n, h * w, c, 1e-5)
    # ---- epilogue operands
    ab_x = act_y = None
    if form in ('coef_sums_xgn', 'coef_act_sums_xgn', 'coef_sums_xgn_store'):
        ab_x = x                       # conv2d_gn_in: the GroupNorm input of the sums IS the conv's input
    elif form == 'two_consumer':
        ab_x = rnd(n, h, w, c)         # the GroupNorm input in front of x (x = GroupNorm(ab_x) with two consumers)
    elif form == 'chain':
        ab_x = rnd(n, h, w, c)
        act_y = x
    xg = None
    if xgn:
        xst = torch.stack([x.double().sum(dim=(1, 2, 3)), (x.double() ** 2).sum(dim=(1, 2, 3))], 1).reshape(-1).contiguous()
        xgam, xbet = (torch.rand(c, generator=g_) + 0.5).cuda(), (torch.randn(c, generator=g_) * 0.1).cuda()
        xg = (xst, xgam, xbet)
    # ---- the two launches replaced
    gx_ref = base.clone()
    ab_ref = torch.zeros(n * slots * 2 * c, dtype=torch.float64, device='cuda') if sums else None
    if coef_form:
        gpre_ref = torch.empty_like(gq)
        ok = L.call_try('dis_conv2d_dgrad_f16x2_gnb', gq, q, coef, in_act, gpre_ref, wt, c, c, wt.stride(0), gx_ref, 1 if accum else 0,
                        ab_x, act_y, ab_ref, n, h, w, c)
        assert ok
    else:
        if in_act:
            gpre_ref = gq * torch.where(q > 0, torch.full_like(q, 1.0507009873554804934193349852946),
                                        q + 1.0507009873554804934193349852946 * 1.6732632423543772848170429916717)
            L.call('dis_conv2d_dgrad_bf16x3_act', gq, q, in_act, wt, c, c, wt.stride(0), gx_ref, n, h, w, c, c, 1, 1 if accum else 0)
        else:
            gpre_ref = gq
            L.call('dis_conv2d_fwd_bf16x3_oihw', gq, wt, 1, c, c, wt.stride(0), None, gx_ref, None, n, h, w, c, c, 3, 1, 1,
                   ops.CONV_ACCUM if accum else 0)
    wsz = L.fn('dis_conv2d_wgrad_workspace')(c, c, 3, 1)
    gw_ref, gb_ref = torch.empty(c, c, 3, 3, device='cuda'), torch.empty(c, device='cuda')
    ws = torch.empty(wsz, dtype=torch.float32, device='cuda')
    if xgn:
        L.call('dis_conv2d_wgrad_bf16x3_gn', x, xg[0], xg[1], xg[2], 1e-5, gpre_ref, gw_ref, gb_ref, ws, n, h, w, c, c, c, 3, 1, 1)
    else:
        L.call('dis_conv2d_wgrad_bf16x3', x, gpre_ref, gw_ref, gb_ref, ws, n, h, w, c, c, c, 3, 1, 1)
    # ---- fp64
    x_eff = x
    if xgn:
        mean = (xg[0].view(n, 2)[:, 0] / (h * w * c)).view(n, 1, 1, 1)
        var = (xg[0].view(n, 2)[:, 1] / (h * w * c)).view(n, 1, 1, 1) - mean ** 2
        x_eff = ((x.double() - mean) / torch.sqrt(var + 1e-5) * xg[1].double() + xg[2].double())
    gw64, gb64 = _fp64_wgrad(x_eff, gpre_ref)
    # ---- the fused launch
    gx = base.clone()
    gpre = torch.full_like(gq, float('nan')) if store else None
    ab = torch.zeros(n * slots * 2 * c, dtype=torch.float64, device='cuda') if sums else None
    gw, gb = torch.full((c, c, 3, 3), float('nan'), device='cuda'), torch.full((c,), float('nan'), device='cuda')
    ws2 = torch.empty(L.fn('dis_conv2d_bwd_fused_workspace')(c), dtype=torch.float32, device='cuda')
    ok = L.call_try('dis_conv2d_bwd_fused_f16x2', gq, q if (coef_form or in_act) else None, coef, in_act, gpre, wt, c, c, wt.stride(0), gx,
                    1 if accum else 0, ab_x, act_y, ab, x, xg[0] if xgn else None, xg[1] if xgn else None, xg[2] if xgn else None, 1e-5,
                    gw, gb, ws2, n, h, w, c, 0)
    if not ok:
        pytest.skip('no instance for this form in this build')
    torch.cuda.synchronize()
    assert torch.equal(gx, gx_ref), float((gx - gx_ref).abs().max())
    if store:
        assert torch.equal(gpre, gpre_ref)
    if sums:
        got, ref = ab.view(n, slots, 2, c).sum(dim=1), ab_ref.view(n, slots, 2, c).sum(dim=1)
        assert float((got - ref).abs().max()) <= 1e-6 * float(ref.abs().max()), float((got - ref).abs().max())
    sw, sb = float(gw64.abs().max()), float(gb64.abs().max())
    e_new, e_old = float((gw.double() - gw64).abs().max()) / sw, float((gw_ref.double() - gw64).abs().max()) / sw
    b_new, b_old = float((gb.double() - gb64).abs().max()) / sb, float((gb_ref.double() - gb64).abs().max()) / sb
    assert e_new < 1e-6, (e_new, e_old)
    assert b_new < 1e-6, (b_new, b_old)
    print(form, (n, h, w), 'grad_w err / largest: fused %.2e, two launches %.2e; grad_b %.2e / %.2e' % (e_new, e_old, b_new, b_old))


def test_bwd_fused_is_reproducible_and_handles_extreme_ranges():
    """a 1e4 outlier in one sample, a 1e-6 sample, zeros in another: the running dW exponent moves, nothing overflows, and the launch
    repeats bit for bit (fixed summation orders, no atomics)"""
    from depthinspace_amd import ops
    L = ops.lib
    if L.fn('dis_get_conv_split')() != 1:
        pytest.skip('two-term fp16 kernels only')
    n, h, w, c = 5, 48, 40, 32
    g_ = torch.Generator().manual_seed(77)
    gy = torch.randn(n, h, w, c, generator=g_).cuda()
    x = torch.randn(n, h, w, c, generator=g_).cuda()
    gy[1] *= 1e-6
    x[2] = 0.0
    gy[3, 7, 9, 5] = 1e4
    x[4, 30, 2, 11] = -3e3
    gy[0, :16, :16] = 0.0
    wt = (torch.randn(c, c, 3, 3, generator=g_) * 0.05).cuda()
    wsz = L.fn('dis_conv2d_bwd_fused_workspace')(c)
    outs = []
    for rep in range(3):
        gx = torch.empty_like(x)
        gw, gb = torch.empty(c, c, 3, 3, device='cuda'), torch.empty(c, device='cuda')
        ws = torch.empty(wsz, dtype=torch.float32, device='cuda')
        if not L.call_try('dis_conv2d_bwd_fused_f16x2', gy, None, None, 0, None, wt, c, c, wt.stride(0), gx, 0, None, None, None, x, None,
                          None, None, 1e-5, gw, gb, ws, n, h, w, c, 0):
            pytest.skip('no instance')
        outs.append((gx, gw, gb))
    torch.cuda.synchronize()
    for o in outs[1:]:
        assert all(torch.equal(a_, b_) for a_, b_ in zip(o, outs[0]))
    gw64, gb64 = _fp64_wgrad(x, gy)
    assert bool(torch.isfinite(outs[0][1]).all())
    assert float((outs[0][1].double() - gw64).abs().max()) < 1e-6 * float(gw64.abs().max())
    assert float((outs[0][2].double() - gb64).abs().max()) < 1e-6 * float(gb64.abs().max())


def test_bwd_fused_writes_a_slice_of_a_wider_weight_gradient():
    """grad_w_row_stride: the slab reduce writes the (32, 32, 3, 3) slice of a (32, 96, 3, 3) gradient in place (conv2d_multi's
    conv over a channel concatenation): the slice equals the contiguous result bit for bit, the rest of the tensor is untouched."""
    from depthinspace_amd import ops
    L = ops.lib
    if L.fn('dis_get_conv_split')() != 1:
        pytest.skip('two-term fp16 kernels only')
    n, h, w, c = 2, 40, 56, 32
    g_ = torch.Generator().manual_seed(5)
    gy = torch.randn(n, h, w, c, generator=g_).cuda()
    x = torch.randn(n, h, w, c, generator=g_).cuda()
    wt = (torch.randn(c, c, 3, 3, generator=g_) * 0.05).cuda()
    wsz = L.fn('dis_conv2d_bwd_fused_workspace')(c)

    def run(gw, stride):
        gx, gb = torch.empty_like(x), torch.empty(c, device='cuda')
        ws = torch.empty(wsz, dtype=torch.float32, device='cuda')
        ok = L.call_try('dis_conv2d_bwd_fused_f16x2', gy, None, None, 0, None, wt, c, c, wt.stride(0), gx, 0, None, None, None, x, None,
                        None, None, 1e-5, gw, gb, ws, n, h, w, c, stride)
        torch.cuda.synchronize()
        return ok, gx, gb

    ref = torch.empty(c, c, 3, 3, device='cuda')
    ok, gx0, gb0 = run(ref, 0)
    if not ok:
        pytest.skip('no instance')
    wide = torch.full((c, 3 * c, 3, 3), 7.0, device='cuda')
    sl = wide[:, c:2 * c]
    ok, gx1, gb1 = run(sl, sl.stride(0))
    assert ok and torch.equal(gx0, gx1) and torch.equal(gb0, gb1)
    assert torch.equal(sl, ref)
    assert bool((wide[:, :c] == 7.0).all()) and bool((wide[:, 2 * c:] == 7.0).all())
    # a pitch that is not whole input channels / narrower than the slice is refused
    for bad in (c * 9 + 1, c * 9 - 9):
        with pytest.raises(L.DisHipError):
            run(ref, bad)
