"""HIP network operators (MFMA conv, GroupNorm, resize, warps, geometry, Conv3D, head, Adam) through the
C ABI vs plain PyTorch fp32 CPU references / the oracle."""
import os
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from oracle import dis_oracle as O


def relerr(a, b):
    a = a.detach().cpu().double()
    b = b.detach().cpu().double()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def nhwc(x):
    return x.permute(0, 2, 3, 1).contiguous()


def nchw(x):
    return x.permute(0, 3, 1, 2).contiguous()


CONV_SHAPES = [
    # cin, cout, k, stride, pad, cin_pad, h, w
    (4, 16, 4, 2, 1, 4, 24, 40),
    (1, 16, 3, 1, 1, 4, 19, 23),
    (16, 16, 3, 1, 1, 16, 20, 36),
    (16, 32, 3, 1, 1, 16, 17, 33),
    (32, 16, 3, 1, 1, 32, 16, 16),
    (32, 32, 3, 1, 1, 32, 27, 45),
    (48, 32, 3, 1, 1, 48, 16, 24),
    (96, 32, 3, 1, 1, 96, 18, 20),
    (128, 32, 1, 1, 0, 128, 12, 28),
    (32, 32, 4, 2, 1, 32, 24, 40),
]


@pytest.mark.parametrize('cin,cout,k,stride,pad,cin_pad,h,w', CONV_SHAPES)
@pytest.mark.parametrize('act', [0, 1])
def test_conv2d_fwd_bwd(cin, cout, k, stride, pad, cin_pad, h, w, act):
    from depthinspace_amd import ops
    g = torch.Generator().manual_seed(cin * 100 + cout + k)
    n = 3
    x = torch.randn(n, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5
    b = torch.randn(cout, generator=g) * 0.1
    xr = x.clone().requires_grad_(True)
    wr = wt.clone().requires_grad_(True)
    br = b.clone().requires_grad_(True)
    y = F.conv2d(F.pad(xr, (pad,) * 4), wr, br, stride=stride)
    if act == 1:
        y = F.selu(y)
    go = torch.randn(y.shape, generator=g)
    y.backward(go)

    xp = torch.zeros(n, cin_pad, h, w)
    xp[:, :cin] = x
    dgrad = cin_pad == cin and cin >= 16   # the 4-channel stem convs only ever see network inputs
    xd = nhwc(xp).cuda().requires_grad_(dgrad)
    wd = wt.cuda().requires_grad_(True)
    bd = b.cuda().requires_grad_(True)
    yd, stats = ops.conv2d(xd, wd, bd, stride, pad, act, want_stats=True, need_dgrad=dgrad)
    yd.backward(nhwc(go).cuda())
    assert relerr(nchw(yd), y) < 2e-6
    # fused GroupNorm statistics of the output
    # (they are the sums of the kernel's OWN outputs - tight; against the fp32 reference's sums the allowance is what the 2e-6 output
    # bar leaves of a sum of numel outputs with errors of random sign: the two-term 4 x 4 stride-2 kernel, K = 512, measured 2e-4 on 7680)
    st = stats.view(n, 2).cpu()
    yo = nchw(yd).detach().double().cpu()
    assert torch.allclose(st[:, 0], yo.sum(dim=(1, 2, 3)), rtol=1e-6, atol=2e-5)
    assert torch.allclose(st[:, 1], (yo ** 2).sum(dim=(1, 2, 3)), rtol=1e-6, atol=2e-5)
    slack = 1e-4 + 2e-6 * float(y.abs().max()) * (y[0].numel() ** 0.5)
    assert torch.allclose(st[:, 0], y.double().sum(dim=(1, 2, 3)), rtol=1e-6, atol=slack)
    assert torch.allclose(st[:, 1], (y.double() ** 2).sum(dim=(1, 2, 3)), rtol=1e-6, atol=slack * 2 * float(y.abs().max()))
    assert relerr(wd.grad, wr.grad) < 1e-5
    assert relerr(bd.grad, br.grad) < 1e-5
    if dgrad:
        assert relerr(nchw(xd.grad), xr.grad) < 1e-5


@pytest.mark.parametrize('cin,cout', [(32, 48), (32, 96), (32, 128)])
def test_conv2d_dgrad_shapes(cin, cout):
    """forward kernels used only as input-gradient kernels (cin/cout swapped)"""
    from depthinspace_amd import ops
    g = torch.Generator().manual_seed(7)
    k = 1 if cout == 128 else 3
    x = torch.randn(2, cin, 14, 18, generator=g)
    wt = torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5
    y = F.conv2d(x, wt, None, padding=k // 2)
    yd, _ = ops.conv2d(nhwc(x).cuda(), wt.cuda(), None, 1, k // 2, 0)
    assert relerr(nchw(yd), y) < 2e-6


def test_conv2d_unsupported_shape_is_loud():
    from depthinspace_amd import ops, lib
    x = torch.zeros(1, 8, 8, 20).cuda()
    w = torch.zeros(16, 20, 3, 3).cuda()
    with pytest.raises(lib.DisHipError):
        ops.conv2d(x, w, None, 1, 1, 0)


@pytest.mark.parametrize('c', [16, 32])
@pytest.mark.parametrize('act,res', [(0, False), (1, True), (1, False)])
def test_group_norm(c, act, res):
    from depthinspace_amd import ops
    g = torch.Generator().manual_seed(c + act)
    n, h, w = 3, 11, 13
    x = torch.randn(n, c, h, w, generator=g) * 2 + 0.5
    gam = 1 + 0.2 * torch.randn(c, generator=g)
    bet = 0.2 * torch.randn(c, generator=g)
    r = torch.randn(n, c, h, w, generator=g)
    xr, gr, br, rr = [t.clone().requires_grad_(True) for t in (x, gam, bet, r)]
    y = F.group_norm(xr, 1, gr, br)
    if res:
        y = y + rr
    if act:
        y = F.selu(y)
    go = torch.randn(y.shape, generator=g)
    y.backward(go)
    xd = nhwc(x).cuda().requires_grad_(True)
    gd = gam.cuda().requires_grad_(True)
    bd = bet.cuda().requires_grad_(True)
    rd = nhwc(r).cuda().requires_grad_(True) if res else None
    yd = ops.group_norm(xd, gd, bd, None, rd, act)
    yd.backward(nhwc(go).cuda())
    assert relerr(nchw(yd), y) < 2e-6
    assert relerr(nchw(xd.grad), xr.grad) < 2e-5
    assert relerr(gd.grad, gr.grad) < 1e-5
    assert relerr(bd.grad, br.grad) < 1e-5
    if res:
        assert relerr(nchw(rd.grad), rr.grad) < 1e-6


@pytest.mark.parametrize('n,h,w', [(3, 19, 23), (2, 32, 48)])
def test_conv1x1_scaled_input_matches_torch(n, h, w):
    """ops.conv2d_scaled_in: the 128 -> 32 1x1 conv over the four mask-weighted slots of Block2D3D (conv_mf, reference
    model/multi_frame_networks.py:406-411: `wf * (mask / mean(mask))` then Conv2d(4C, C, 1)), multiplier per (pixel, 32-channel slot)
    applied inside the kernels: output, statistics, input gradient, weight and bias gradient (the weight gradient takes all 128
    channels in one workgroup since round 3) against torch on the host; ragged tiles."""
    from depthinspace_amd import ops
    g = torch.Generator().manual_seed(n * 100 + h)
    x = torch.randn(n, h, w, 128, generator=g)
    sc = torch.rand(n, h, w, 4, generator=g) * 2
    wt = torch.randn(32, 128, 1, 1, generator=g) / 128 ** 0.5
    b = torch.randn(32, generator=g) * 0.1
    go = torch.randn(n, h, w, 32, generator=g)
    xr, wr, br = x.clone().requires_grad_(True), wt.clone().requires_grad_(True), b.clone().requires_grad_(True)
    xs = (xr.view(n, h, w, 4, 32) * sc.unsqueeze(-1)).view(n, h, w, 128)
    yr = F.conv2d(xs.permute(0, 3, 1, 2), wr, br).permute(0, 2, 3, 1)
    yr.backward(go)
    xd, wd, bd = x.cuda().requires_grad_(True), wt.cuda().requires_grad_(True), b.cuda().requires_grad_(True)
    yd, st = ops.conv2d_scaled_in(xd, sc.cuda(), wd, bd, 1, 0, want_stats=True)
    yd.backward(go.cuda())
    assert relerr(yd, yr) < 2e-6
    s_ref = torch.stack([yr.double().sum(dim=(1, 2, 3)), (yr.double() ** 2).sum(dim=(1, 2, 3))], dim=1).reshape(-1)
    assert torch.allclose(st.cpu(), s_ref.detach(), rtol=1e-6, atol=1e-4)
    assert relerr(xd.grad, xr.grad) < 5e-6
    assert relerr(wd.grad, wr.grad) < 5e-6
    assert relerr(bd.grad, br.grad) < 5e-6


@pytest.mark.parametrize('c', [16, 32])
@pytest.mark.parametrize('act', [0, 1])
@pytest.mark.parametrize('n,h,w', [(3, 37, 29), (2, 64, 48)])
def test_conv_with_group_norm_on_load_equals_the_two_pass_form(c, act, n, h, w):
    """ops.conv2d_gn_in (GroupNorm applied by the consuming conv while it stages its input: the normalised tensor never
    exists) against the two-pass form it replaces, group_norm -> conv2d, on the same values: outputs, statistics and EVERY
    gradient (input of the pair, GroupNorm scale / shift, conv weight / bias) must be bit-identical - the staging computes
    x * (rstd * gamma) + (beta - rstd * gamma * mean) exactly as gn_apply does, padding stays zero (ragged 37 x 29 tiles
    included).  And both against torch (reference chain: model/multi_frame_networks.py:338-345,514-542)."""
    from depthinspace_amd import ops, lib
    assert ops.gn_fusable(c, c, 3, 1)
    g = torch.Generator().manual_seed(100 * c + act + h)
    x = torch.randn(n, c, h, w, generator=g) * 1.5 + 0.3
    gam = 1 + 0.2 * torch.randn(c, generator=g)
    bet = 0.2 * torch.randn(c, generator=g)
    wt = torch.randn(c, c, 3, 3, generator=g) / (3 * c ** 0.5)
    b = 0.1 * torch.randn(c, generator=g)
    go = torch.randn(n, c, h, w, generator=g)
    # producer: x = SELU(pre); the pair's input gradient is handed back as the PRE-activation gradient (in_act)
    res = []
    for fused in (False, True):
        xd = F.selu(nhwc(x).cuda()).contiguous().requires_grad_(True)
        gd, bd, wd, bbd = [t.cuda().requires_grad_(True) for t in (gam, bet, wt, b)]
        st = torch.zeros(2 * n, dtype=torch.float64, device='cuda')
        lib.call('dis_gn_stats', xd.detach(), st, n, h * w * c)
        if fused:
            y, yst = ops.conv2d_gn_in(xd, st, gd, bd, wd, bbd, 1, act, want_stats=True, gy_is_pre=False, in_act=ops.ACT_SELU)
        else:
            nrm = ops.group_norm(xd, gd, bd, stats=st, in_act=ops.ACT_SELU)
            y, yst = ops.conv2d(nrm, wd, bbd, 1, 1, act, want_stats=True)
        y.backward(nhwc(go).cuda())
        res.append((y.detach(), yst.clone(), xd.grad, gd.grad, bd.grad, wd.grad, bbd.grad))
    for a_, b_, name in zip(res[0], res[1], ('y', 'stats', 'gx', 'ggamma', 'gbeta', 'gw', 'gb')):
        if name in ('gx', 'ggamma', 'gbeta') and ops.GN_SUMS:
            # the fused pair's GroupNorm backward works from the per-(sample, channel) sums of its input-gradient epilogue
            # (dis_gn_bwd_from_sums): the same quantities, factored differently - equal to rounding, not bit for bit
            assert relerr(a_, b_) < 2e-5, (name, relerr(a_, b_))
        elif name in ('gw', 'gb') and ops.BWD_FUSED and c == 32:
            # round 6: the two-pass form's conv backward is the fused input- / weight-gradient launch, the pair's is the stand-alone
            # weight-gradient kernel - the same sums in another order (both within 2e-7 of fp64, tests/test_bwd_fused_gpu.py)
            assert relerr(a_, b_) < 2e-6, (name, relerr(a_, b_))
        else:
            assert torch.equal(a_, b_), (name, float((a_ - b_).abs().max()))
    # torch on the host
    xr = F.selu(nchw(nhwc(x))).detach().requires_grad_(True)
    gr, br, wr, bbr = [t.clone().requires_grad_(True) for t in (gam, bet, wt, b)]
    yr = F.conv2d(F.group_norm(xr, 1, gr, br), wr, bbr, padding=1)
    if act:
        yr = F.selu(yr)
    yr.backward(go)
    y, _, gx, gg, gb_, gw, gbb = res[1]
    assert relerr(nchw(y), yr) < 5e-6
    # gx is the gradient wrt the producer's PRE-activation: torch's gradient wrt xr times SELU'(xr)
    sel = torch.where(xr.detach() > 0, torch.full_like(xr, 1.0507009873554805), xr.detach() + 1.0507009873554805 * 1.6732632423543772)
    assert relerr(nchw(gx), xr.grad * sel) < 5e-5
    assert relerr(gg, gr.grad) < 5e-5 and relerr(gb_, br.grad) < 5e-5
    assert relerr(gw, wr.grad) < 5e-5 and relerr(gbb, bbr.grad) < 5e-5


@pytest.mark.parametrize('ac', [True, False])
@pytest.mark.parametrize('hin,win,ho,wo', [(8, 10, 16, 20), (9, 7, 18, 14), (16, 20, 8, 10), (5, 6, 13, 11)])
def test_resize(ac, hin, win, ho, wo):
    from depthinspace_amd import ops
    g = torch.Generator().manual_seed(hin * win)
    x = torch.randn(2, 8, hin, win, generator=g)
    xr = x.clone().requires_grad_(True)
    y = F.interpolate(xr, size=(ho, wo), mode='bilinear', align_corners=ac)
    go = torch.randn(y.shape, generator=g)
    y.backward(go)
    xd = nhwc(x).cuda().requires_grad_(True)
    yd = ops.resize_nhwc(xd, (ho, wo), ac)
    yd.backward(nhwc(go).cuda())
    assert relerr(nchw(yd), y) < 1e-6
    assert relerr(nchw(xd.grad), xr.grad) < 1e-5
    xp = x.cuda().requires_grad_(True)
    yp = ops.resize_planar(xp, (ho, wo), ac)
    yp.backward(go.cuda())
    assert relerr(yp, y) < 1e-6
    assert relerr(xp.grad, xr.grad) < 1e-5


@pytest.mark.parametrize('ac', [True, False])
@pytest.mark.parametrize('c', [32, 16])
@pytest.mark.parametrize('hin,win,ho,wo', [(27, 45, 54, 90), (16, 70, 32, 140), (30, 38, 15, 19), (9, 13, 27, 39), (7, 5, 40, 33),
                                           (13, 67, 21, 100), (1, 3, 2, 6)])
def test_resize_nhwc_tiled(ac, c, hin, win, ho, wo):
    """the tiled resize kernels (round 5: a workgroup owns an 8-row tile, per-workgroup weight tables in the backward) against torch and,
    bit for bit, against the grid-stride kernels they replace (reached through a tensor that is not 16-byte aligned); 2 x upsampling
    with ragged tiles, downsampling, 3 x and 6 x upsampling (more than five contributing destination indices: the general loop)"""
    from depthinspace_amd import ops
    L = ops.lib
    g = torch.Generator().manual_seed(hin * win + c)
    n = 3
    x = torch.randn(n, c, hin, win, generator=g)
    xr = x.clone().requires_grad_(True)
    y = F.interpolate(xr, size=(ho, wo), mode='bilinear', align_corners=ac)
    go = torch.randn(y.shape, generator=g)
    y.backward(go)
    xd, gd = nhwc(x).cuda(), nhwc(go).cuda()
    yd = torch.full((n, ho, wo, c), float('nan'), device='cuda')
    gx = torch.full((n, hin, win, c), float('nan'), device='cuda')
    L.call('dis_resize_bilinear_nhwc_fwd', xd, yd, n, hin, win, ho, wo, c, int(ac))
    L.call('dis_resize_bilinear_nhwc_bwd', gd, gx, n, hin, win, ho, wo, c, int(ac))
    # (source positions near 66 carry 8e-6 of fp32 rounding: the weights of torch's kernel and of this one may differ by that much)
    assert relerr(nchw(yd), y) < 5e-6
    assert relerr(nchw(gx), xr.grad) < 2e-5

    def unaligned(t):
        buf = torch.empty(t.numel() + 1, device='cuda')
        v = buf[1:].view(t.shape)
        v.copy_(t)
        assert v.data_ptr() % 16 == 4
        return v
    xu, gu = unaligned(xd), unaligned(gd)
    yu, gxu = unaligned(torch.full_like(yd, float('nan'))), unaligned(torch.full_like(gx, float('nan')))
    L.call('dis_resize_bilinear_nhwc_fwd', xu, yu, n, hin, win, ho, wo, c, int(ac))
    L.call('dis_resize_bilinear_nhwc_bwd', gu, gxu, n, hin, win, ho, wo, c, int(ac))
    assert torch.equal(yu, yd)
    assert torch.equal(gxu, gx)


def test_resize_flow_scale(golden_dir):
    from depthinspace_amd import ops
    G = np.load(os.path.join(golden_dir, 'ops.npz'))
    fl = torch.from_numpy(G['warp_flow'])
    y = ops.resize_planar(fl.cuda(), (10, 12), True, flow_scale=(12 / 24.0, 10 / 20.0))
    assert relerr(y, torch.from_numpy(G['resize_flow_out'])) < 1e-6
    x = torch.from_numpy(G['warp_x'])
    y = ops.resize_planar(x.cuda(), (10, 12), True)
    assert relerr(y, torch.from_numpy(G['resize_out'])) < 1e-6


def _stack_flows(flow, tl, bs, h, w):
    fl = torch.zeros(tl * tl, bs, 2, h, w)
    for i in range(tl):
        for j in range(tl):
            if i != j:
                fl[i * tl + j] = flow[f'flow_{i}{j}']
    return fl


def test_gather_warped_feat():
    from depthinspace_amd import ops
    g = torch.Generator().manual_seed(5)
    tl, bs, c, h, w = 4, 2, 32, 12, 15
    feat = torch.randn(tl, bs, c, h, w, generator=g)
    flow = {f'flow_{i}{j}': torch.randn(bs, 2, h, w, generator=g) * 3 for i in range(tl) for j in range(tl) if i != j}
    fr = feat.clone().requires_grad_(True)
    ref = torch.stack([O._gather_warped_feat(fr, flow, t, tl) for t in range(tl)], 0)  # (tl,slot,bs,c,h,w)
    go = torch.randn(ref.shape, generator=g)
    ref.backward(go)
    fl = _stack_flows(flow, tl, bs, h, w).permute(0, 1, 3, 4, 2).contiguous().cuda()
    fd = feat.permute(0, 1, 3, 4, 2).contiguous().cuda().requires_grad_(True)
    out = ops.gather_warped_feat(fd, fl)  # (tl,bs,h,w,slot,c)
    out.backward(go.permute(0, 2, 4, 5, 1, 3).contiguous().cuda())
    assert relerr(out.permute(0, 4, 1, 5, 2, 3), ref) < 2e-6
    assert relerr(fd.grad.permute(0, 1, 4, 2, 3), fr.grad) < 1e-5
    # CSR (atomic-free) backward: same gradient, bitwise reproducible from call to call
    god = go.permute(0, 2, 4, 5, 1, 3).contiguous().cuda()
    grads = []
    for _ in range(2):
        csr = ops.gather_csr(fl)
        f2 = feat.permute(0, 1, 3, 4, 2).contiguous().cuda().requires_grad_(True)
        ops.gather_warped_feat(f2, fl, csr).backward(god)
        grads.append(f2.grad.clone())
    assert relerr(grads[0].permute(0, 1, 4, 2, 3), fr.grad) < 1e-5
    assert torch.equal(grads[0], grads[1])


@pytest.mark.parametrize('tl,bs,c,h,w', [(4, 2, 32, 37, 45), (4, 1, 16, 16, 70), (3, 2, 32, 9, 33), (2, 3, 64, 21, 8)])
def test_gather_warped_feat_tiled_kernels(tl, bs, c, h, w):
    """the tiled feature-warp kernels (round 5: 8 x 32 pixel tiles, the targets of a tile on one XCD, several rows / CSR entries in flight)
    against the grid-stride kernels they replace (DIS_GATHER_TILED=0, read per call): forward and CSR backward bit for bit - ragged
    tiles, 16 / 32 / 64 channels, 2 - 4 frames, large flows that leave the image."""
    from depthinspace_amd import ops
    L = ops.lib
    g = torch.Generator().manual_seed(tl * 100 + c + h)
    feat = torch.randn(tl, bs, h, w, c, generator=g).cuda()
    flows = (torch.randn(tl * tl, bs, h, w, 2, generator=g) * 5).cuda()
    go = torch.randn(tl, bs, h, w, tl, c, generator=g).cuda()
    init = torch.randn(tl, bs, h, w, c, generator=g).cuda()
    csr = ops.gather_csr(flows)
    res = []
    for mode in ('1', '0'):
        os.environ['DIS_GATHER_TILED'] = mode
        try:
            out = torch.full((tl, bs, h, w, tl, c), float('nan'), device='cuda')
            L.call('dis_gather_warped_feat_fwd', feat, flows, out, tl, bs, h, w, c)
            gf = torch.full((tl, bs, h, w, c), float('nan'), device='cuda')
            L.call('dis_gather_warped_feat_bwd_csr', go, csr, None, gf, tl, bs, h, w, c)
            gf2 = torch.full((tl, bs, h, w, c), float('nan'), device='cuda')
            L.call('dis_gather_warped_feat_bwd_csr', go, csr, init, gf2, tl, bs, h, w, c)
            res.append((out, gf, gf2))
        finally:
            os.environ.pop('DIS_GATHER_TILED')
    for a_, b_ in zip(res[0], res[1]):
        assert torch.equal(a_, b_)
    assert not torch.isnan(res[0][0]).any() and not torch.isnan(res[0][1]).any()
    assert relerr(res[0][2], res[0][1] + init) < 1e-6


@pytest.mark.parametrize('cfg', [(64, 48, 2, 8, True, {}), (128, 128, 1, 4321, False, dict(scene='bumps', motion=1.5)),
                                 (256, 216, 1, 5, False, {})])
def test_mf_geometry_masks_and_selection_bit_exact(cfg):
    """Index-class outputs of FuseNet's parameter-free part: the geometry pyramids, the fb masks and Conv3D's neighbour
    sets of the HIP path equal tests/bitexact.py (the rounding-exact numpy statement, itself equal to the oracle and to
    the reference's torch.topk output: tests/test_bitexact_cpu.py) BIT FOR BIT - no tolerance, no excluded near-ties.
    (64x48 -> 32x24 / 16x12 runs ATen's small-output resize arithmetic, 256x216 -> 128x108 / 64x54 the generic one.)"""
    from depthinspace_amd import ops, synth, lib
    from tests import bitexact as B
    H, W, bs, seed, rnd, kw = cfg
    st = synth.make_settings(H, W)
    b = synth.make_random_batch(st, bs, 4, seed=seed) if rnd else synth.make_batch(st, bs, 4, seed=seed, **kw)
    tb = {k: torch.from_numpy(v).transpose(0, 1).contiguous() if v.ndim > 2 else torch.from_numpy(v) for k, v in b.items()}
    tl = 4
    h, w = H // 2, W // 2
    bf = float(st.K[0, 0]) * st.baseline
    eq = np.array_equal
    # disparity -> depth -> core depth / core flows
    depth = ops.disp_to_depth(tb['primary_disp'].cuda(), bf)
    e_depth = B.disp_to_depth(tb['primary_disp'].numpy(), float(st.K[0, 0]), st.baseline)
    assert eq(depth.cpu().numpy(), e_depth)
    depth_core = ops.resize_planar(depth.view(tl, bs, H, W), (h, w), True)
    e_dc = B.resize_ac(e_depth, h, w)
    assert eq(depth_core.cpu().numpy(), e_dc[:, :, 0])
    flow = {k: v[0] for k, v in tb.items() if k.startswith('flow_')}
    ff = _stack_flows(flow, tl, bs, H, W).cuda()
    fc_p = ops.resize_planar(ff, (h, w), True, flow_scale=(float(w) / float(W), float(h) / float(H)))
    e_fc = {k: B.resize_flow(v.numpy(), h, w) for k, v in flow.items()}
    for i in range(tl):
        for j in range(tl):
            if i != j:
                assert eq(fc_p[i * tl + j].cpu().numpy(), e_fc[f'flow_{i}{j}']), (i, j)
    # geometry + masks
    fl = ops.planar_to_nhwc(fc_p.view(tl * tl * bs, 2, h, w)).view(tl * tl, bs, h, w, 2)
    Ki = lib.host_floats(np.linalg.inv(st.K).reshape(-1))
    geom = ops.mf_geometry(depth_core, tb['R'].cuda(), tb['t'].cuda(), fl, Ki, W // w, H // h)
    ex, em = B.mf_geometry(e_dc, O.mf_core_rays(st.K, H, W).numpy(), tb['R'].numpy(), tb['t'].numpy(), e_fc)
    gx = geom[..., :3].permute(0, 4, 1, 5, 2, 3).cpu().numpy()  # (tl,slot,bs,3,h,w)
    gm = geom[..., 3].permute(0, 4, 1, 2, 3).unsqueeze(3).cpu().numpy()  # (tl,slot,bs,1,h,w)
    assert eq(gm, em), float((gm != em).mean())
    assert eq(gx, ex), float((gx != ex).mean())
    # the oracle agrees too (it does on any host whose torch build rounds like the fixture host's)
    depth_o = O.disp_to_depth(tb['primary_disp'], float(st.K[0, 0]), st.baseline)
    wxyz, wmask = O.mf_geometry(O.resize_ac(depth_o, (h, w)), O.mf_core_rays(st.K, H, W), tb['R'], tb['t'],
                                O.resize_flow(flow, (h, w)))
    host_same = eq(wxyz.numpy(), ex)
    assert eq(wmask.numpy(), em) or not host_same
    if not host_same:
        print('note: this host rounds torch CPU kernels differently from the fixture host; oracle agreement skipped')
        assert relerr(torch.from_numpy(gx), wxyz) < 2e-6
    # quarter resolution
    hq, wq = (h + 2 - 3) // 2 + 1, (w + 2 - 3) // 2 + 1
    gq = ops.mf_geometry_resize(geom, (hq, wq))
    exq = B.resize_ac(ex, hq, wq)
    emq = (B.resize_ac(em, hq, wq) > 0.5).astype(np.float32)
    assert eq(gq[..., :3].permute(0, 4, 1, 5, 2, 3).cpu().numpy(), exq)
    assert eq(gq[..., 3].permute(0, 4, 1, 2, 3).unsqueeze(3).cpu().numpy(), emq)
    # Conv3D neighbour sets: torch.topk's ids in torch.topk's order
    assert eq(ops.conv3d_select(geom, 2).cpu().numpy(), B.conv3d_select(ex, em, 2))
    assert eq(ops.conv3d_select(gq, 1).cpu().numpy(), B.conv3d_select(exq, emq, 1))
    # slot weighting
    g = torch.Generator().manual_seed(1)
    wf = torch.randn(tl, tl, bs, 8, h, w, generator=g)
    wm = torch.from_numpy(em)
    ref = wf * wm / wm.mean(dim=1, keepdim=True)
    out = ops.mask_weight_slots(wf.permute(0, 2, 4, 5, 1, 3).contiguous().cuda(), geom)
    assert relerr(out.permute(0, 4, 1, 5, 2, 3), ref) < 1e-6


@pytest.mark.parametrize('stride', [1, 2])
def test_conv3d_golden(golden_dir, stride):
    from depthinspace_amd import ops
    G = np.load(os.path.join(golden_dir, 'ops.npz'))
    xyz, feat, mask = [torch.from_numpy(G[k]) for k in ('c3_xyz', 'c3_feat', 'c3_mask')]
    tl, bs, C, h, w = feat.shape
    p = O.init_params({k: v for k, v in O.mf_param_shapes().items() if k.startswith('blocks.0.conv3d_1')}, seed=5)
    pd = {k[len('blocks.0.conv3d_1.'):]: v.detach().cuda().requires_grad_(True) for k, v in p.items()}
    # one target: build (tl=4 targets) by repeating the same window set so the kernel's layout is exercised
    geom1 = torch.cat([xyz, mask], dim=2).permute(1, 3, 4, 0, 2)  # (bs,h,w,slot,4)
    geom = geom1.unsqueeze(0).expand(tl, -1, -1, -1, -1, -1).contiguous().cuda()
    wf1 = feat.permute(1, 3, 4, 0, 2)
    wf = wf1.unsqueeze(0).expand(tl, -1, -1, -1, -1, -1).contiguous().cuda().requires_grad_(True)
    idx = ops.conv3d_select(geom, stride)
    y = ops.conv3d_knn(geom, wf, pd['dense1.0.weight'], pd['dense1.0.bias'], pd['dense2.0.weight'],
                       pd['dense2.0.bias'], pd['w'], idx, stride)
    ho, wo = y.shape[2:4]
    yn = ops.group_norm(y.view(tl * bs, ho, wo, C), pd['bn.weight'], pd['bn.bias'])
    out = yn.view(tl, bs, ho, wo, C)
    go = torch.from_numpy(G[f'c3_s{stride}_go'])  # (bs,C,ho,wo)
    gfull = torch.zeros(tl, bs, ho, wo, C)
    gfull[1] = go.permute(0, 2, 3, 1)
    out.backward(gfull.cuda())
    ref = torch.from_numpy(G[f'c3_s{stride}_out'])
    assert relerr(out[1].permute(0, 3, 1, 2), ref) < 2e-5
    assert float(G[f'c3_s{stride}_margin_min']) > 0  # goldens have no top-k ties
    sel = np.sort(idx[1].cpu().numpy().astype(np.int16), axis=-1)
    assert np.array_equal(sel, G[f'c3_s{stride}_idx_sorted'])  # every neighbour set, exactly
    gfeat = torch.from_numpy(G[f'c3_s{stride}_gfeat'])  # (tl(slot),bs,C,h,w)
    assert relerr(wf.grad[1].permute(3, 0, 4, 1, 2), gfeat) < 5e-5
    for k_ in ('w', 'dense1.0.weight', 'dense1.0.bias', 'dense2.0.weight', 'dense2.0.bias', 'bn.weight', 'bn.bias'):
        assert relerr(pd[k_].grad, torch.from_numpy(G[f'c3_s{stride}_g:{k_}'])) < 1e-4, k_


@pytest.mark.parametrize('stride', [1, 2])
def test_conv3d_csr_feature_gradient(golden_dir, stride):
    """The deterministic (CSR gather) form of the Conv3D feature gradient equals the float-atomic scatter to rounding, in both
    modes (write / accumulate into a shared buffer), repeats bit for bit, and the golden gradient is met through it."""
    from depthinspace_amd import ops
    G = np.load(os.path.join(golden_dir, 'ops.npz'))
    xyz, feat, mask = [torch.from_numpy(G[k]) for k in ('c3_xyz', 'c3_feat', 'c3_mask')]
    tl, bs, C, h, w = feat.shape
    p = O.init_params({k: v for k, v in O.mf_param_shapes().items() if k.startswith('blocks.0.conv3d_1')}, seed=5)
    pd = {k[len('blocks.0.conv3d_1.'):]: v.detach().cuda() for k, v in p.items()}
    g = torch.Generator().manual_seed(3)
    geom1 = torch.cat([xyz, mask], dim=2).permute(1, 3, 4, 0, 2)
    geom = geom1.unsqueeze(0).expand(tl, -1, -1, -1, -1, -1).contiguous().cuda()
    wf = torch.randn(tl, bs, h, w, tl, C, generator=g).cuda()
    idx = ops.conv3d_select(geom, stride)
    ho, wo = idx.shape[2:4]
    y = torch.empty((tl, bs, ho, wo, C), device='cuda')
    args = (geom, wf, pd['dense1.0.weight'], pd['dense1.0.bias'], pd['dense2.0.weight'], pd['dense2.0.bias'], pd['w'], idx)
    ops.lib.call('dis_conv3d_knn_fwd', *args, y, tl, bs, h, w, stride)
    gy = torch.randn(y.shape, generator=g).cuda()
    base = torch.randn(wf.shape, generator=g).cuda()
    acc = torch.empty(ops.lib.fn('dis_conv3d_knn_bwd_workspace')(), device='cuda')
    csr = ops.conv3d_csr(idx, h, w, stride)
    stage = torch.empty(ops.lib.fn('dis_conv3d_knn_bwd_stage')(tl, bs, h, w, stride), device='cuda')
    # atomic scatter on top of `base`
    g_at, gp_at = base.clone(), torch.empty(1632, device='cuda')
    ops.lib.call('dis_conv3d_knn_bwd', *args, y, gy, g_at, gp_at, acc, tl, bs, h, w, stride)
    outs = []
    for rep in range(2):
        g_acc, gp = base.clone(), torch.empty(1632, device='cuda')
        ops.lib.call('dis_conv3d_knn_bwd_csr', *args, y, gy, g_acc, gp, acc, csr, stage, 1, tl, bs, h, w, stride)
        g_wr = torch.full(wf.shape, float('nan'), device='cuda')   # write mode must define every row
        ops.lib.call('dis_conv3d_knn_bwd_csr', *args, y, gy, g_wr, gp, acc, csr, stage, 0, tl, bs, h, w, stride)
        outs.append((g_acc, g_wr, gp))
    g_acc, g_wr, gp = outs[0]
    scale = float((g_at - base).abs().max())
    assert scale > 0 and float((g_acc - g_at).abs().max()) < 2e-6 * scale
    assert bool(torch.isfinite(g_wr).all()) and float((g_wr - (g_at - base)).abs().max()) < 4e-6 * scale
    assert relerr(gp, gp_at) < 1e-6
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])   # bitwise reproducible
    # every entry whose source pixel exists is in exactly one list
    nsrc = tl * bs * h * w * 4
    offs = csr[:nsrc + 1].cpu().numpy()
    ents = csr[2 * nsrc + 1: 2 * nsrc + 1 + offs[-1]].cpu().numpy()
    assert len(np.unique(ents)) == len(ents) and ents.max() < tl * bs * ho * wo * 9
    assert all(np.all(np.diff(ents[offs[d]:offs[d + 1]]) > 0) for d in np.flatnonzero(np.diff(offs) > 1)[:2000])


def test_gn_sums_epilogue_repeats_bitwise():
    """The channel sums an input-gradient launch leaves for the GroupNorm backward (fixed summation orders, no atomics) and what
    dis_gn_bwd_from_sums makes of them repeat bit for bit - at a shape with few tiles per workgroup, where a workgroup's last
    tile often belongs to another sample than its sums so far: two barrier-free flushes then follow each other at the end of
    the kernel, and before the barrier between them was added they raced (wrong / missing sums whenever wave timing shifted;
    found with two processes sharing the GPU, scripts/diag/share_gnin.py)."""
    from depthinspace_amd import ops
    if ops.lib.fn('dis_get_conv_split')() != 1:
        pytest.skip('channel sums: f16x2 kernels')
    g = torch.Generator().manual_seed(11)
    n, h, w, c = 16, 128, 108, 32
    x = torch.randn(n, h, w, c, generator=g).cuda()
    wt = (torch.randn(c, c, 3, 3, generator=g) * 0.05).cuda()
    gamma, beta = (torch.rand(c, generator=g) + 0.5).cuda(), (torch.randn(c, generator=g) * 0.1).cuda()
    gy = torch.randn(n, h, w, c, generator=g).cuda()
    st = torch.stack([x.double().sum(dim=(1, 2, 3)), (x.double() ** 2).sum(dim=(1, 2, 3))], 1).reshape(-1).contiguous()
    slots = ops.lib.fn('dis_conv2d_gnsums_slots')()
    first = None
    for rep in range(12):
        gnorm, gx = torch.empty_like(x), torch.empty_like(x)
        gg, gb = torch.empty(c, device='cuda'), torch.empty(c, device='cuda')
        ab = torch.zeros(n * slots * 2 * c, dtype=torch.float64, device='cuda')
        ops.lib.call('dis_conv2d_dgrad_bf16x3_gnsums', gy, wt, c, c, wt.stride(0), gnorm, x, ab, n, h, w, c, c, 1)
        coef = torch.empty(n * (c + 2) + 4 * n * c + 2, dtype=torch.float32, device='cuda')
        ops.lib.call('dis_gn_bwd_from_sums', gnorm, x, st, gamma, ab, slots, gx, gg, gb, coef, n, h * w, c, 1e-5, ops.ACT_SELU)
        cur = (ab, gx, gg, gb)
        if first is None:
            first = tuple(t.clone() for t in cur)
            # the sums are the sums: against a plain reduction of the kernel's own gradient output
            ref_a = gnorm.double().sum(dim=(1, 2))
            ref_b = (gnorm.double() * x.double()).sum(dim=(1, 2))
            got = ab.view(n, slots, 2, c).sum(dim=1)
            assert float((got[:, 0] - ref_a).abs().max()) < 1e-6 * float(ref_a.abs().max())
            assert float((got[:, 1] - ref_b).abs().max()) < 1e-6 * float(ref_b.abs().max())
        else:
            assert all(torch.equal(a_, b_) for a_, b_ in zip(cur, first)), rep


@pytest.mark.parametrize('c', [16, 32])
@pytest.mark.parametrize('in_act', [0, 1])
@pytest.mark.parametrize('form', ['plain', 'accum', 'sums', 'accum_sums', 'accum_sums_act'])
@pytest.mark.parametrize('n,h,w', [(3, 37, 29), (4, 64, 48), (2, 16, 250)])
def test_dgrad_with_group_norm_backward_on_load(c, in_act, form, n, h, w):
    """dis_conv2d_dgrad_f16x2_gnb (round 5): the GroupNorm backward's elementwise pass applied while the input-gradient launch of the
    conv in front of the GroupNorm stages its operand, against the two launches it replaces - dis_gn_bwd_apply_coef, then the plain /
    accumulating / channel-sum input-gradient launch on the materialised tensor: the stored pre-activation gradient (every pixel
    exactly once, ragged tiles and one-tile-high maps included), the input gradient and the channel sums of the epilogue forms must
    all be BIT-identical (same arithmetic, same summation orders).  dis_gn_bwd_coef against dis_gn_bwd_from_sums likewise
    (coefficients, grad_gamma, grad_beta).  Reference: GroupNorm(1, C) behind a conv, model/multi_frame_networks.py:338-345,514-542."""
    from depthinspace_amd import ops
    L = ops.lib
    if L.fn('dis_get_conv_split')() != 1:
        pytest.skip('two-term fp16 kernels only')
    if form == 'accum_sums_act' and in_act == 0:
        pytest.skip('the ResNetBlock-chain epilogue follows a SELU conv')
    g_ = torch.Generator().manual_seed(7 * c + in_act + h)
    q = torch.randn(n, h, w, c, generator=g_).cuda()
    if in_act:
        q = F.selu(q)
    gq = torch.randn(n, h, w, c, generator=g_).cuda()           # gradient wrt the GroupNorm's output
    wt = (torch.randn(c, c, 3, 3, generator=g_) * 0.05).cuda()
    gamma = (torch.rand(c, generator=g_) + 0.5).cuda()
    st = torch.stack([q.double().sum(dim=(1, 2, 3)), (q.double() ** 2).sum(dim=(1, 2, 3))], 1).reshape(-1).contiguous()
    slots = L.fn('dis_conv2d_gnsums_slots')()
    # the sums a producing launch would have left (any values do: one slot per sample carries them)
    ab0 = torch.zeros(n, slots, 2, c, dtype=torch.float64, device='cuda')
    ab0[:, 0, 0] = gq.double().sum(dim=(1, 2))
    ab0[:, 0, 1] = (gq.double() * q.double()).sum(dim=(1, 2))
    ncoef = n * (c + 2) + 4 * n * c + 2
    # --- the old composition
    gpre_ref = torch.empty_like(gq)
    gg_ref, gb_ref = torch.empty(c, device='cuda'), torch.empty(c, device='cuda')
    coef_ref = torch.empty(ncoef, dtype=torch.float32, device='cuda')
    L.call('dis_gn_bwd_from_sums', gq, q, st, gamma, ab0, slots, gpre_ref, gg_ref, gb_ref, coef_ref, n, h * w, c, 1e-5, in_act)
    # --- the coefficient kernel alone
    coef = torch.empty(ncoef, dtype=torch.float32, device='cuda')
    gg, gb = torch.empty(c, device='cuda'), torch.empty(c, device='cuda')
    counter = torch.zeros(2, dtype=torch.int32, device='cuda')
    for _ in range(2):   # (the counter re-arms itself: a second launch on the same word)
        L.call('dis_gn_bwd_coef', st, gamma, ab0, slots, coef, gg, gb, counter, n, h * w, c, 1e-5)
    assert torch.equal(coef[:n * (c + 2)], coef_ref[:n * (c + 2)]) and torch.equal(gg, gg_ref) and torch.equal(gb, gb_ref)
    assert int(counter[0]) == 0
    gpre2 = torch.empty_like(gq)
    L.call('dis_gn_bwd_apply_coef', gq, q, coef, gpre2, n, h * w, c, in_act)
    assert torch.equal(gpre2, gpre_ref)
    # --- input gradient of the materialised tensor, in the form under test
    accum = form.startswith('accum')
    sums = 'sums' in form
    act_y = F.selu(torch.randn(n, h, w, c, generator=g_)).cuda() if form == 'accum_sums_act' else None
    gn_x = torch.randn(n, h, w, c, generator=g_).cuda() if sums else None
    base = torch.randn(n, h, w, c, generator=g_).cuda()
    gx_ref = base.clone()
    ab_ref = torch.zeros(n * slots * 2 * c, dtype=torch.float64, device='cuda') if sums else None
    if form == 'plain' or form == 'accum':
        L.call('dis_conv2d_fwd_bf16x3_oihw', gpre_ref, wt, 1, c, c, wt.stride(0), None, gx_ref, None, n, h, w, c, c, 3, 1, 1,
               ops.CONV_ACCUM if accum else 0)
    elif form == 'sums':
        L.call('dis_conv2d_dgrad_bf16x3_gnsums', gpre_ref, wt, c, c, wt.stride(0), gx_ref, gn_x, ab_ref, n, h, w, c, c, 1)
    else:
        L.call('dis_conv2d_dgrad_bf16x3_gnsums_res', gpre_ref, wt, c, c, wt.stride(0), gx_ref, act_y, gn_x, ab_ref, n, h, w, c, c, 1)
    # --- the fused launch
    gx = base.clone()
    gpre = torch.full_like(gq, float('nan'))
    ab = torch.zeros(n * slots * 2 * c, dtype=torch.float64, device='cuda') if sums else None
    ok = L.call_try('dis_conv2d_dgrad_f16x2_gnb', gq, q, coef, in_act, gpre, wt, c, c, wt.stride(0), gx, 1 if accum else 0, gn_x,
                    act_y, ab, n, h, w, c)
    if not ok:
        pytest.skip('no instance for this form in this build')
    torch.cuda.synchronize()
    assert torch.equal(gpre, gpre_ref), float((gpre - gpre_ref).abs().max())
    assert torch.equal(gx, gx_ref), float((gx - gx_ref).abs().max())
    if sums:
        assert torch.equal(ab, ab_ref)


@pytest.mark.parametrize('c', [16, 32])
@pytest.mark.parametrize('n,h,w', [(3, 37, 29), (2, 64, 48)])
def test_resnet_chain_with_deferred_block_outputs(c, n, h, w):
    """ResNetBlock(defer_out=True): the block's output SELU(GroupNorm(x2) + x) is not written by a pass of its own but by the next
    block's first conv while it loads it (dis_conv2d_fwd_f16x2_gnres).  A chain of three blocks + a 32 -> 16 conv with and without the
    deferral: outputs, every block output tensor, and ALL gradients bit-identical (the staging repeats dis_gn_apply's arithmetic).
    Reference: model/multi_frame_networks.py:514-542."""
    from depthinspace_amd import ops
    from depthinspace_amd.model import multi_frame_networks as mfn
    if ops.lib.fn('dis_get_conv_split')() != 1:
        pytest.skip('two-term fp16 kernels only')
    torch.manual_seed(5 + c + h)
    blocks = [mfn.ResNetBlock(c).cuda() for _ in range(3)]
    tail = mfn.ConvParams(c, 16, 3).cuda()
    for b in blocks + [tail]:
        for p_ in b.parameters():
            if p_.dim() > 1:
                torch.nn.init.normal_(p_, std=0.08)
            else:
                torch.nn.init.normal_(p_, mean=0.5 if p_.dim() == 1 else 0.0, std=0.2)
    x0 = torch.randn(n, h, w, c).cuda()
    go = torch.randn(n, h, w, 16).cuda()
    res = []
    for defer in (False, True):
        ops.begin_step('cuda:0')
        for b in blocks + [tail]:
            for p_ in b.parameters():
                p_.grad = None
        x = x0.clone().requires_grad_(True)
        o1 = blocks[0](x, defer_out=defer)
        o2 = blocks[1](o1, defer_out=defer)
        o3 = blocks[2](o2, defer_out=defer and c == 32)
        y = ops.conv2d(o3, tail.weight, tail.bias, 1, 1, ops.ACT_SELU, gnres=getattr(o3, '_gn_res_src', None))[0]
        y.backward(go)
        torch.cuda.synchronize()
        assert not ops._GN_PENDING and not ops._GN_LAZY and not ops._GN_PRE
        res.append([t.detach().clone() for t in (y, o1, o2, o3, x.grad)] +
                   [p_.grad.clone() for b in blocks + [tail] for p_ in b.parameters()])
    for i, (a_, b_) in enumerate(zip(res[0], res[1])):
        assert torch.equal(a_, b_), (i, float((a_ - b_).abs().max()))


@pytest.mark.parametrize('c', [16, 32])
@pytest.mark.parametrize('act,res', [(1, True), (0, False)])
@pytest.mark.parametrize('n,hw', [(3, 37 * 29), (2, 64 * 48), (1, 5)])
def test_gn_bwd_res_sums(c, act, res, n, hw):
    """dis_gn_bwd_res_sums (round 5): g = gy SELU'(y) stored, per-(sample, channel) sums of g and g x in the slot layout of the conv
    epilogues (n, slots, 2, c) - against fp64; the sums repeat bit for bit."""
    from depthinspace_amd import ops
    L = ops.lib
    g_ = torch.Generator().manual_seed(c + hw + act)
    gy = torch.randn(n, hw, c, generator=g_).cuda()
    y = (torch.randn(n, hw, c, generator=g_) * 1.5).cuda()
    x = (torch.randn(n, hw, c, generator=g_) * 2 + 0.3).cuda()
    slots = L.fn('dis_conv2d_gnsums_slots')()
    outs = []
    for _ in range(2):
        ab = torch.full((n, slots, 2, c), float('nan'), dtype=torch.float64, device='cuda')
        gres = torch.full_like(gy, float('nan')) if res else None
        L.call('dis_gn_bwd_res_sums', gy, y if act else None, x, gres, ab, slots, n, hw, c, act)
        outs.append((ab.clone(), gres.clone() if res else None))
    assert torch.equal(outs[0][0], outs[1][0])
    yd = y.double()
    sel = torch.where(yd > 0, torch.full_like(yd, 1.0507009873554805), yd + 1.0507009873554805 * 1.6732632423543772)
    g = gy.double() * sel if act else gy.double()
    if res:
        assert relerr(outs[0][1], g) < 1e-6
    ab = outs[0][0].sum(dim=1)   # (n, 2, c)
    assert relerr(ab[:, 0], g.sum(dim=1)) < 2e-6
    assert relerr(ab[:, 1], (g * x.double()).sum(dim=1)) < 2e-6


@pytest.mark.parametrize('fuse_a', [True, False])
@pytest.mark.parametrize('n,h,w', [(3, 37, 29), (2, 64, 48)])
def test_conv_multi_group_norm_backward_from_a_join(fuse_a, n, h, w):
    """Block2D3D's tail (reference model/multi_frame_networks.py:338-345): out = SELU(GroupNorm(conv_fuse(cat(a, b, c))) + feat), whose
    output gradient arrives from a join (no conv epilogue left channel sums).  Round 5: one pass forms the residual gradient and the
    sums (dis_gn_bwd_res_sums), the elementwise pass rides on the first slice's input-gradient launch of conv_fuse
    (dis_conv2d_dgrad_f16x2_gnb) - against dis_gn_apply_bwd's reduce + apply launches (DIS_GN_RES_SUMS=0) and against torch."""
    from depthinspace_amd import ops
    if ops.lib.fn('dis_get_conv_split')() != 1:
        pytest.skip('two-term fp16 kernels only')
    C = 32
    g_ = torch.Generator().manual_seed(n + h)
    mk = lambda *sh, s=1.0: (torch.randn(*sh, generator=g_) * s)
    a0, b0, c0, feat0 = mk(n, h, w, C), mk(n, h, w, C), mk(n, h, w, C), mk(n, h, w, C)
    wt, bias = mk(C, 3 * C, 3, 3, s=0.06), mk(C, s=0.1)
    gam, bet = mk(C, s=0.2) + 0.8, mk(C, s=0.2)
    gam_a, bet_a = mk(C, s=0.2) + 0.9, mk(C, s=0.2)
    go = mk(n, h, w, C)
    runs = []
    for flag in (True, False):
        ops.GN_RES_SUMS = flag
        try:
            ops.begin_step('cuda:0')
            leaves = [t.clone().cuda().requires_grad_(True) for t in (a0, b0, c0, feat0, wt, bias, gam, bet, gam_a, bet_a)]
            a, b, c, feat, wd, bd, gd, btd, gad, bad = leaves
            if fuse_a:   # the first source is a GroupNorm input applied on load (conv1_2's GroupNorm), as in Block2D3D
                # (a conv in front so that the source is a conv output, as the token path requires of its producer)
                a_in, st_a = ops.conv2d(a, wd[:, :C].contiguous() * 0.5, None, 1, 1, ops.ACT_SELU, want_stats=True, gy_is_pre=True)
                f, st = ops.conv2d_multi((a_in, b, c), wd, bd, 1, ops.ACT_NONE, want_stats=True, gn0=(st_a, gad, bad, 1e-5, ops.ACT_SELU))
            else:
                f, st = ops.conv2d_multi((a, b, c), wd, bd, 1, ops.ACT_NONE, want_stats=True)
            out = ops.group_norm(f, gd, btd, stats=st, residual=feat, act=ops.ACT_SELU)
            out.backward(go.cuda())
            torch.cuda.synchronize()
            assert not ops._GN_LAZY and not ops._GN_PRE
            runs.append([out.detach().clone()] + [t.grad.clone() if t.grad is not None else None for t in leaves])
        finally:
            ops.GN_RES_SUMS = True
    for i, (p_, q_) in enumerate(zip(runs[0], runs[1])):
        if p_ is None:
            assert q_ is None
            continue
        assert relerr(p_, q_) < 2e-5, (i, relerr(p_, q_))
    if not fuse_a:   # torch reference of the plain form
        lv = [t.clone().double().requires_grad_(True) for t in (a0, b0, c0, feat0, wt, bias, gam, bet)]
        a, b, c, feat, wd, bd, gd, btd = lv
        xin = torch.cat([a, b, c], 3).permute(0, 3, 1, 2)
        f = F.conv2d(xin, wd, bd, padding=1)
        o = F.selu(F.group_norm(f, 1, gd, btd, 1e-5) + feat.permute(0, 3, 1, 2))
        o.backward(go.double().permute(0, 3, 1, 2))
        assert relerr(nchw(runs[0][0]), o) < 1e-5
        for i, t in enumerate(lv):
            assert relerr(runs[0][1 + i], t.grad) < 5e-5, i


@pytest.mark.parametrize('in_act', [0, 1])
@pytest.mark.parametrize('accum', [0, 1])
@pytest.mark.parametrize('n,h,w', [(3, 21, 37), (2, 64, 48)])
def test_dgrad1x1_with_group_norm_backward_on_load(in_act, accum, n, h, w):
    """dis_conv2d_dgrad1x1_scaled_gnb (round 5): the input gradient of the 1 x 1 multi-frame conv (128 -> 32, slot weights on its
    input) with the GroupNorm backward's elementwise pass applied while g is staged, against dis_gn_bwd_apply_coef followed by
    dis_conv2d_fwd_scaled on the materialised tensor: stored values and input gradient bit-identical (ragged tiles included).
    Reference: model/multi_frame_networks.py:406-413 (conv_mf + GroupNorm)."""
    from depthinspace_amd import ops
    L = ops.lib
    g_ = torch.Generator().manual_seed(31 + in_act + 2 * accum + h)
    c, cw = 32, 128
    q = torch.randn(n, h, w, c, generator=g_).cuda()
    if in_act:
        q = F.selu(q)
    gq = torch.randn(n, h, w, c, generator=g_).cuda()
    wt = (torch.randn(c, cw, 1, 1, generator=g_) * 0.1).cuda()
    ysc = torch.rand(n, h, w, cw // 32, generator=g_).cuda()
    coef = (torch.randn(n * (c + 2) + 4 * n * c + 2, generator=g_) * 0.5).cuda()
    base = torch.randn(n, h, w, cw, generator=g_).cuda()
    wp = ops._pack_w(wt, cw, 1)
    gpre_ref = torch.empty_like(gq)
    L.call('dis_gn_bwd_apply_coef', gq, q, coef, gpre_ref, n, h * w, c, in_act)
    gx_ref = base.clone()
    L.call('dis_conv2d_fwd_scaled', gpre_ref, None, wp, None, gx_ref, ysc, None, n, h, w, c, cw, 1, 1, 0,
           ops.ACT_NONE | (ops.CONV_ACCUM if accum else 0))
    gx = base.clone()
    gpre = torch.full_like(gq, float('nan'))
    L.call('dis_conv2d_dgrad1x1_scaled_gnb', gq, q, coef, in_act, gpre, wp, gx, ysc, n, h, w, c, cw, accum)
    torch.cuda.synchronize()
    assert torch.equal(gpre, gpre_ref), float((gpre - gpre_ref).abs().max())
    assert torch.equal(gx, gx_ref), float((gx - gx_ref).abs().max())


@pytest.mark.parametrize('in_act', [0, 1])
@pytest.mark.parametrize('n,h,w', [(3, 38, 30), (2, 64, 48), (16, 64, 54)])
def test_wgrad_k4s2_with_group_norm_backward_on_load(in_act, n, h, w):
    """dis_conv2d_wgrad_k4s2_f16x2_gnb (round 5): the weight gradient of FuseNet's 4 x 4 stride-2 conv with the GroupNorm backward's
    elementwise pass applied while gy is staged, against dis_gn_bwd_apply_coef + dis_conv2d_wgrad on the materialised tensor: the
    stored values, grad_w and grad_b bit-identical (ragged tiles, several tiles per workgroup).  Reference:
    model/multi_frame_networks.py:338-345 (conv2_1: Conv2d(k4, s2, p1) -> SELU -> GroupNorm)."""
    from depthinspace_amd import ops
    L = ops.lib
    if L.fn('dis_get_conv_split')() != 1:
        pytest.skip('two-term fp16 kernels only')
    g_ = torch.Generator().manual_seed(77 + in_act + h)
    c = 32
    ho, wo = h // 2, w // 2
    x = torch.randn(n, h, w, c, generator=g_).cuda()
    q = torch.randn(n, ho, wo, c, generator=g_).cuda()
    if in_act:
        q = F.selu(q)
    gq = torch.randn(n, ho, wo, c, generator=g_).cuda()
    coef = (torch.randn(n * (c + 2) + 4 * n * c + 2, generator=g_) * 0.5).cuda()
    wsz = L.fn('dis_conv2d_wgrad_workspace')(c, c, 4, 2)
    gpre_ref = torch.empty_like(gq)
    L.call('dis_gn_bwd_apply_coef', gq, q, coef, gpre_ref, n, ho * wo, c, in_act)
    gw_ref, gb_ref = torch.empty(c, c, 4, 4, device='cuda'), torch.empty(c, device='cuda')
    L.call('dis_conv2d_wgrad', x, gpre_ref, gw_ref, gb_ref, torch.empty(wsz, device='cuda'), n, h, w, c, c, c, 4, 2, 1)
    gpre = torch.full_like(gq, float('nan'))
    gw, gb = torch.empty(c, c, 4, 4, device='cuda'), torch.empty(c, device='cuda')
    L.call('dis_conv2d_wgrad_k4s2_f16x2_gnb', x, gq, q, coef, in_act, gpre, gw, gb, torch.empty(wsz, device='cuda'), n, h, w)
    torch.cuda.synchronize()
    assert torch.equal(gpre, gpre_ref), float((gpre - gpre_ref).abs().max())
    assert torch.equal(gw, gw_ref), float((gw - gw_ref).abs().max())
    assert torch.equal(gb, gb_ref), float((gb - gb_ref).abs().max())


@pytest.mark.parametrize('case', ['randn', 'outlier', 'tiny_corner'])
@pytest.mark.parametrize('n,h,w,act', [(3, 38, 30, 1), (2, 64, 48, 0), (16, 64, 56, 1), (1, 17, 70, 2)])
def test_conv_fwd_k4s2_f16x2(case, n, h, w, act):
    """dis_conv2d_fwd_k4s2_f16x2 (round 5, csrc/conv_k4s2.hip): FuseNet's 4 x 4 stride-2 pad-1 down convolution (32 -> 32) forward on
    the two-term fp16 kernel - wave-resident weight fragments, de-interleaved halo columns, one scale per 18 x 34 halo tile - against
    fp64 (bar: 1e-6 of the largest output, the bar of the 3 x 3 two-term kernels) beside the exact-fp32 MFMA kernel it replaces,
    with the GroupNorm statistics of the epilogue; ragged tiles, odd sizes, several tiles per workgroup, a 1e4 outlier pixel and a
    1e-6 corner (block scaling).  Reference: model/multi_frame_networks.py:338-345 (conv2_1: Conv2d(32, 32, 4, stride 2, pad 1) + SELU)."""
    from depthinspace_amd import ops
    L = ops.lib
    if L.fn('dis_get_conv_split')() != 1:
        pytest.skip('two-term fp16 kernels only')
    g_ = torch.Generator().manual_seed(n + h + w + act + len(case))
    c = 32
    x = torch.randn(n, c, h, w, generator=g_)
    if case == 'outlier':
        x[0, :, h // 2, w // 3] = 1e4
    elif case == 'tiny_corner':
        x[:, :, : h // 2, : w // 2] *= 1e-6
    wt = torch.randn(c, c, 4, 4, generator=g_) / (c * 16) ** 0.5
    b = torch.randn(c, generator=g_) * 0.1
    ref = F.conv2d(x.double(), wt.double(), b.double(), stride=2, padding=1)
    ref = {0: ref, 1: F.selu(ref), 2: F.relu(ref)}[act].permute(0, 2, 3, 1)
    ho, wo = ref.shape[1], ref.shape[2]
    xd, wd, bd = nhwc(x).cuda(), wt.cuda(), b.cuda()
    y = torch.full((n, ho, wo, c), float('nan'), device='cuda')
    st = torch.zeros(2 * n, dtype=torch.float64, device='cuda')
    L.call('dis_conv2d_fwd_k4s2_f16x2', xd, wd, bd, y, st, n, h, w, act)
    y0 = torch.empty_like(y)
    st0 = torch.zeros(2 * n, dtype=torch.float64, device='cuda')
    L.call('dis_conv2d_fwd', xd, ops._pack_w(wd, c, 0), bd, y0, st0, n, h, w, c, c, 4, 2, 1, act)
    torch.cuda.synchronize()
    scale = float(ref.abs().max())
    e2, e0 = float((y.double().cpu() - ref).abs().max()) / scale, float((y0.double().cpu() - ref).abs().max()) / scale
    print(f'{case} {n}x{h}x{w} act {act}: two-term {e2:.2e}, fp32 MFMA {e0:.2e}')
    assert e2 < 1e-6, (e2, e0)
    if case == 'tiny_corner':   # the small region keeps its own relative accuracy where whole tiles lie inside it (its tiles' own scale)
        sub, sref = y[:, : ho // 2 - 5, : wo // 2 - 9].double().cpu(), ref[:, : ho // 2 - 5, : wo // 2 - 9]
        if sub.numel() and act != 1:   # (SELU adds the bias-dominated offset: judge the linear cases)
            assert float((sub - sref).abs().max()) < 1e-6 * max(float(sref.abs().max()), 1e-30) + 1e-7 * float(b.abs().max())
    sums = torch.stack([y.double().sum(dim=(1, 2, 3)), (y.double() ** 2).sum(dim=(1, 2, 3))], 1).reshape(-1)
    assert float((st - sums).abs().max()) < 1e-5 * float(sums.abs().max())


@pytest.mark.parametrize('case', ['randn', 'outlier', 'tiny_corner'])
@pytest.mark.parametrize('n,h,w,accum', [(3, 38, 30, 0), (2, 64, 48, 1), (16, 64, 56, 0), (1, 18, 70, 1)])
def test_conv_dgrad_k4s2_f16x2(case, n, h, w, accum):
    """dis_conv2d_dgrad_k4s2_f16x2 (round 5): the input gradient of FuseNet's 4 x 4 stride-2 conv, all four parity classes from one gy
    halo tile in one launch on the two-term kernel, against fp64 (bar 1e-6 of the largest entry) beside the four fp32 parity launches
    it replaces (dis_conv2d_dgrad_strided); writing and accumulating, ragged tiles, block-scaling cases."""
    from depthinspace_amd import ops
    L = ops.lib
    if L.fn('dis_get_conv_split')() != 1:
        pytest.skip('two-term fp16 kernels only')
    g_ = torch.Generator().manual_seed(n + h + w + accum + len(case))
    c = 32
    ho, wo = h // 2, w // 2
    gy = torch.randn(n, c, ho, wo, generator=g_)
    if case == 'outlier':
        gy[0, :, ho // 2, wo // 3] = 1e4
    elif case == 'tiny_corner':
        gy[:, :, : ho // 2, : wo // 2] *= 1e-6
    wt = torch.randn(c, c, 4, 4, generator=g_) / (c * 4) ** 0.5
    base = torch.randn(n, h, w, c, generator=g_) if accum else torch.zeros(n, h, w, c)
    ref = F.conv_transpose2d(gy.double(), wt.double(), stride=2, padding=1).permute(0, 2, 3, 1) + base.double()
    gd, wd = nhwc(gy).cuda(), wt.cuda()
    gx = base.clone().cuda() if accum else torch.full((n, h, w, c), float('nan'), device='cuda')
    L.call('dis_conv2d_dgrad_k4s2_f16x2', gd, wd, gx, n, h, w, accum)
    gx0 = base.clone().cuda()
    L.call('dis_conv2d_dgrad_strided', gd, wd, gx0, torch.empty(16 * c * c, device='cuda'), n, h, w, c, c, 4, 2, 1, accum)
    torch.cuda.synchronize()
    scale = float(ref.abs().max())
    e2, e0 = float((gx.double().cpu() - ref).abs().max()) / scale, float((gx0.double().cpu() - ref).abs().max()) / scale
    print(f'{case} {n}x{h}x{w} accum {accum}: two-term {e2:.2e}, fp32 MFMA {e0:.2e}')
    assert e2 < 1e-6, (e2, e0)


@pytest.mark.parametrize('stride', [1, 2])
def test_conv3d_class_ordered_backward(golden_dir, stride):
    """dis_conv3d_knn_bwd_det (the default backward: class-ordered plain read-modify-write, aggregate read back from the forward)
    equals round 1-2's float-atomic kernel to rounding - feature gradient added on top of a base and all parameter gradients -,
    repeats bit for bit, and so does its one-launch float-atomic form up to the atomics' order."""
    from depthinspace_amd import ops
    G = np.load(os.path.join(golden_dir, 'ops.npz'))
    xyz, feat, mask = [torch.from_numpy(G[k]) for k in ('c3_xyz', 'c3_feat', 'c3_mask')]
    tl, bs, C, h, w = feat.shape
    p = O.init_params({k: v for k, v in O.mf_param_shapes().items() if k.startswith('blocks.0.conv3d_1')}, seed=5)
    pd = {k[len('blocks.0.conv3d_1.'):]: v.detach().cuda() for k, v in p.items()}
    g = torch.Generator().manual_seed(3)
    geom1 = torch.cat([xyz, mask], dim=2).permute(1, 3, 4, 0, 2)
    geom = geom1.unsqueeze(0).expand(tl, -1, -1, -1, -1, -1).contiguous().cuda()
    wf = torch.randn(tl, bs, h, w, tl, C, generator=g).cuda()
    idx = ops.conv3d_select(geom, stride)
    ho, wo = idx.shape[2:4]
    y, agg, y0 = [torch.empty((tl, bs, ho, wo, C), device='cuda') for _ in range(3)]
    args = (geom, wf, pd['dense1.0.weight'], pd['dense1.0.bias'], pd['dense2.0.weight'], pd['dense2.0.bias'], pd['w'], idx)
    ops.lib.call('dis_conv3d_knn_fwd_agg', *args, y, agg, tl, bs, h, w, stride)
    ops.lib.call('dis_conv3d_knn_fwd', *args, y0, tl, bs, h, w, stride)
    assert torch.equal(y, y0)
    gy = torch.randn(y.shape, generator=g).cuda()
    base = torch.randn(wf.shape, generator=g).cuda()
    acc = torch.empty(ops.lib.fn('dis_conv3d_knn_bwd_workspace')(), device='cuda')
    accd = torch.empty(ops.lib.fn('dis_conv3d_knn_bwd_det_workspace')(tl, bs, h, w, stride), device='cuda')
    g_at, gp_at = base.clone(), torch.empty(1632, device='cuda')
    ops.lib.call('dis_conv3d_knn_bwd', *args, y, gy, g_at, gp_at, acc, tl, bs, h, w, stride)
    outs = []
    for rep in range(2):
        g_d, gp_d = base.clone(), torch.empty(1632, device='cuda')
        ops.lib.call('dis_conv3d_knn_bwd_det', *args, y, agg, gy, g_d, gp_d, accd, tl, bs, h, w, stride)
        outs.append((g_d, gp_d))
    g_a, gp_a = base.clone(), torch.empty(1632, device='cuda')
    ops.lib.call('dis_conv3d_knn_bwd_agg', *args, y, agg, gy, g_a, gp_a, accd, tl, bs, h, w, stride)
    scale = float((g_at - base).abs().max())
    assert scale > 0
    for g_x, gp_x in (outs[0], (g_a, gp_a)):
        assert float((g_x - g_at).abs().max()) < 2e-6 * scale
        assert relerr(gp_x, gp_at) < 2e-6
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])   # bitwise reproducible


def test_disp_head():
    from depthinspace_amd import ops
    g = torch.Generator().manual_seed(2)
    n, c, h, w = 2, 16, 21, 19
    x = torch.randn(n, c, h, w, generator=g)
    wt = torch.randn(1, c, 3, 3, generator=g) * 0.2
    b = torch.randn(1, generator=g)
    xr, wr, br = [t.clone().requires_grad_(True) for t in (x, wt, b)]
    y = 128 * torch.sigmoid(F.conv2d(xr, wr, br, padding=1) - 3)
    go = torch.randn(y.shape, generator=g)
    y.backward(go)
    xd = nhwc(x).cuda().requires_grad_(True)
    wd, bd = wt.cuda().requires_grad_(True), b.cuda().requires_grad_(True)
    yd = ops.disp_head(xd, wd, bd, 128.0, 3.0)
    yd.backward(go.cuda())
    assert relerr(yd, y) < 2e-6
    assert relerr(nchw(xd.grad), xr.grad) < 1e-5
    assert relerr(wd.grad, wr.grad) < 1e-5
    assert relerr(bd.grad, br.grad) < 1e-5


@pytest.mark.parametrize('cin', [16, 32])
@pytest.mark.parametrize('n,h,w', [(2, 37, 45), (3, 16, 64), (1, 33, 130), (2, 5, 3)])
def test_disp_head_tiled_forward(cin, n, h, w):
    """head_fwd_tiled_kernel (round 5: a workgroup walks down a 16-row tile with three running sums per thread) against the grid-stride
    kernel it replaces (DIS_HEAD_TILED=0, read per call) - bit for bit, ragged tiles and maps smaller than a tile included - and torch."""
    from depthinspace_amd import ops
    L = ops.lib
    g = torch.Generator().manual_seed(cin + h)
    x = torch.randn(n, h, w, cin, generator=g).cuda()
    wt = (torch.randn(1, cin, 3, 3, generator=g) * 0.2).cuda()
    b = torch.randn(1, generator=g).cuda()
    ys = []
    for mode in ('1', '0'):
        os.environ['DIS_HEAD_TILED'] = mode
        try:
            y = torch.full((n, 1, h, w), float('nan'), device='cuda')
            L.call('dis_disp_head_fwd', x, wt, b, y, n, h, w, cin, 128.0, 0.5)
            ys.append(y)
        finally:
            os.environ.pop('DIS_HEAD_TILED')
    assert torch.equal(ys[0], ys[1])
    ref = 128.0 * torch.sigmoid(F.conv2d(nchw(x).double(), wt.double(), b.double(), padding=1) - 0.5)
    assert relerr(ys[0], ref) < 2e-6


def test_adam_matches_torch():
    from depthinspace_amd import ops
    g = torch.Generator().manual_seed(4)
    p = torch.randn(1024, generator=g)
    pr = p.clone().requires_grad_(True)
    opt = torch.optim.Adam([pr], lr=1e-4)
    pd = p.cuda()
    m = torch.zeros_like(pd)
    v = torch.zeros_like(pd)
    for step in range(1, 4):
        gr = torch.randn(1024, generator=g) * 10 ** (-step)
        pr.grad = gr.clone()
        opt.step()
        ops.adam_step(pd, gr.cuda(), m, v, step)
        assert float((pd.cpu() - pr.detach()).abs().max()) < 2e-7


def test_adam_graph_replay_matches_torch():
    """ONE captured optimiser step replayed k times == k steps of torch.optim.Adam: the step counter and the bias
    corrections live on the device (dis_adam_step_dev), so replay k applies step k's correction.  Also FlatAdam's
    state_dict round-trips through torch.optim.Adam's own layout."""
    from depthinspace_amd.trainer import FlatAdam
    g = torch.Generator().manual_seed(4)
    shapes = [(8, 4, 3, 3), (8,), (33,)]
    ps = [torch.randn(s_, generator=g) for s_ in shapes]
    ref = [p.clone().requires_grad_(True) for p in ps]
    topt = torch.optim.Adam(ref, lr=1e-3)
    mine = [torch.nn.Parameter(p.clone().cuda()) for p in ps]
    opt = FlatAdam(mine, lr=1e-3)
    grads = [[torch.randn(s_, generator=g) * 10 ** (-k) for s_ in shapes] for k in range(6)]
    gin = torch.zeros_like(opt.flat_g)   # static input of the captured step

    def load(k):
        gin[:opt.n].copy_(torch.cat([t.reshape(-1) for t in grads[k]]).cuda())

    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):  # one eager step (k = 0) before the capture, as in bench.py
        load(0)
        opt.flat_g.copy_(gin)
        opt.step()
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    load(1)
    with torch.cuda.graph(graph):
        opt.flat_g.copy_(gin)
        opt.step(all_reduce=False)
    for k in range(2, 5):   # the capture itself executed nothing: replays are steps 2, 3, 4 (k = 1 .. 3 of the data)
        load(k - 1)
        graph.replay()
    torch.cuda.synchronize()
    assert opt.step_count == 4
    for k in range(4):
        for r, gr in zip(ref, grads[k]):
            r.grad = gr.clone()
        topt.step()
    for a, b in zip(mine, ref):
        assert float((a.detach().cpu() - b.detach()).abs().max()) < 5e-7
    # state_dict in torch.optim.Adam's layout: loads into torch.optim.Adam and back
    sd = opt.state_dict()
    t2 = torch.optim.Adam([p.clone().requires_grad_(True) for p in ps], lr=1e-3)
    t2.load_state_dict({'state': {k: {kk: (vv.cpu() if torch.is_tensor(vv) else vv) for kk, vv in v.items()}
                                  for k, v in sd['state'].items()}, 'param_groups': sd['param_groups']})
    tsd = topt.state_dict()
    for i in range(len(shapes)):
        assert float((sd['state'][i]['exp_avg'].cpu() - tsd['state'][i]['exp_avg']).abs().max()) < 1e-7
        assert float(sd['state'][i]['step']) == float(tsd['state'][i]['step']) == 4.0
    opt2 = FlatAdam([torch.nn.Parameter(p.clone().cuda()) for p in ps], lr=1e-3)
    opt2.load_state_dict(tsd)   # a reference-format optimizer state (torch.optim.Adam.state_dict())
    assert opt2.step_count == 4
    assert float((opt2.exp_avg[:opt2.n] - opt.exp_avg[:opt.n]).abs().max()) < 1e-7
    assert float((opt2.exp_avg_sq[:opt2.n] - opt.exp_avg_sq[:opt.n]).abs().max()) < 1e-9


@pytest.mark.parametrize('h,w', [(37, 45), (64, 48)])
@pytest.mark.parametrize('cin,cout,cin_real', [(32, 32, 32), (16, 16, 16), (16, 32, 16), (32, 16, 32), (16, 16, 3)])
def test_conv_bf16x3_is_fp32_accurate(h, w, cin, cout, cin_real):
    """The 3x3 convolutions on the bf16 matrix cores (3-way operand split, 6 products, fp32 accumulate) must be as close
    to an fp64 reference as the exact-fp32 MFMA kernels are: forward, input gradient and weight gradient, for 16 / 32
    channels on either side and for a first layer whose x carries zero-padded extra channels (cin_real < cin).
    The two-term fp16 split (3 products, 22-bit operands; the default) is measured beside it with its own bar: within
    1e-6 of the largest entry (about 4x the exact-fp32 kernel's distance from fp64)."""
    from depthinspace_amd import lib, ops
    from tests.conftest import conv_split
    g = torch.Generator().manual_seed(h * 100 + w + cin + 3 * cout + cin_real)
    n = 3
    x = torch.randn(n, h, w, cin, generator=g)
    x[..., cin_real:] = 0
    x = x.cuda()
    wt = (torch.randn(cout, cin_real, 3, 3, generator=g) * 0.06).cuda()
    b = torch.randn(cout, generator=g).cuda()
    gy = torch.randn(n, h, w, cout, generator=g).cuda()
    xr = x[..., :cin_real].permute(0, 3, 1, 2).double().cpu().requires_grad_(True)
    wr = wt.double().cpu().requires_grad_(True)
    br = b.double().cpu().requires_grad_(True)
    yr = F.conv2d(xr, wr, br, padding=1)
    yr.backward(gy.permute(0, 3, 1, 2).double().cpu())
    res = {}
    for tag in ('fp32', 'bf16x3', 'f16x2'):
      with conv_split('f16x2' if tag == 'f16x2' else 'bf16x3'):
          y = torch.empty(n, h, w, cout, device='cuda')
          gx = torch.zeros_like(x)
          gw = torch.empty_like(wt)
          gb = torch.empty(cout, device='cuda')
          ws = torch.empty(lib.fn('dis_conv2d_wgrad_workspace')(cin, cout, 3, 1), device='cuda')
          if tag == 'fp32':
              lib.call('dis_conv2d_fwd', x, ops._pack_w(wt, cin, 0), b, y, None, n, h, w, cin, cout, 3, 1, 1, 0)
              if cin_real == cin:
                  lib.call('dis_conv2d_fwd', gy, ops._pack_w(wt, cin, 1), None, gx, None, n, h, w, cout, cin, 3, 1, 1, 0)
              lib.call('dis_conv2d_wgrad', x, gy, gw, gb, ws, n, h, w, cin, cin_real, cout, 3, 1, 1)
          else:
              lib.call('dis_conv2d_fwd_bf16x3_oihw', x, wt, 0, cout, cin_real, 0, b, y, None, n, h, w, cin, cout, 3, 1, 1, 0)
              if cin_real == cin:
                  lib.call('dis_conv2d_fwd_bf16x3_oihw', gy, wt, 1, cout, cin_real, 0, None, gx, None, n, h, w, cout, cin, 3, 1,
                           1, 0)
              lib.call('dis_conv2d_wgrad_bf16x3', x, gy, gw, gb, ws, n, h, w, cin, cin_real, cout, 3, 1, 1)
          egx = relerr(gx[..., :cin_real].permute(0, 3, 1, 2), xr.grad) if cin_real == cin else 0.0
          res[tag] = (relerr(y.permute(0, 3, 1, 2), yr), egx, relerr(gw, wr.grad), relerr(gb, br.grad))
    print('max rel err vs fp64 (y, gx, gw, gb): fp32-MFMA', res['fp32'], ' bf16x3', res['bf16x3'], ' f16x2', res['f16x2'])
    for e32, e3, e2 in zip(res['fp32'], res['bf16x3'], res['f16x2']):
        assert e3 < 3e-6, res
        assert e3 < 4 * e32 + 1e-7, res
        assert e2 < 1e-6, res


@pytest.mark.parametrize('case', ['outlier_pixel', 'tiny_tiles', 'dominant_weight'])
@pytest.mark.parametrize('cin,cout', [(32, 32), (16, 16)])
def test_conv_f16x2_dynamic_range(case, cin, cout):
    """The two-term fp16 split scales every halo TILE (and the weights of a launch) by a power of two, because fp16 has 5 exponent
    bits.  Inputs that stress that block scaling, against fp64, beside the three-term bf16 kernel (8 exponent bits, no scaling)
    on the same inputs; bar = 4 x the bf16x3 kernel's error (+ 2e-7 of the normalising magnitude):
      outlier_pixel   one pixel of 1e4 among O(1) values: the tile's scale is set by the outlier; errors are measured over ALL
                      outputs (of the largest entry) and over the outputs that do not see the outlier (of THEIR largest entry);
      tiny_tiles      a 32 x 32 corner of 1e-6 values beside O(1) tiles: the all-tiny tile (its halo included) must keep full
                      relative accuracy (own scale); a tile whose halo touches the O(1) region is scaled by that region - its tiny
                      entries keep an ABSOLUTE error of 2^-39 of the tile maximum (asserted: 1e-9 of the largest output), the
                      local relative figure is printed;
      dominant_weight one weight 1e3 x the rest: one weight scale per launch.
    Forward, input gradient (the same kernel on gy) and weight gradient (running scales over the tiles)."""
    from depthinspace_amd import lib
    from tests.conftest import conv_split
    g = torch.Generator().manual_seed(cin * 10 + cout + len(case))
    n, h, w = 2, 48, 64
    x = torch.randn(n, h, w, cin, generator=g)
    gy = torch.randn(n, h, w, cout, generator=g)
    wt = torch.randn(cout, cin, 3, 3, generator=g) * 0.06
    keep = torch.ones(n, h, w, dtype=torch.bool)     # outputs judged by the local metric
    if case == 'outlier_pixel':
        x[0, 20, 21, :] = 1e4
        gy[1, 7, 40, 3] = 1e4
        keep[0, 19:22, 20:23] = False
    elif case == 'tiny_tiles':
        x[:, :32, :32, :] *= 1e-6
        gy[:, :32, :32, :] *= 1e-6
    else:
        wt[5, 3, 1, 1] = 60.0
    x, gy, wt = x.cuda(), gy.cuda(), wt.cuda()
    b = torch.randn(cout, generator=g).cuda() * (1e-6 if case == 'tiny_tiles' else 1.0)
    xr = x.permute(0, 3, 1, 2).double().cpu().requires_grad_(True)
    wr = wt.double().cpu().requires_grad_(True)
    yr = F.conv2d(xr, wr, b.double().cpu(), padding=1)
    yr.backward(gy.permute(0, 3, 1, 2).double().cpu())
    yr, gxr, gwr = yr.detach().permute(0, 2, 3, 1), xr.grad.permute(0, 2, 3, 1), wr.grad
    out = {}
    for tag in ('bf16x3', 'f16x2'):
        with conv_split(tag):
            y = torch.empty(n, h, w, cout, device='cuda')
            gx = torch.zeros_like(x)
            gw = torch.empty_like(wt)
            gb = torch.empty(cout, device='cuda')
            ws = torch.empty(lib.fn('dis_conv2d_wgrad_workspace')(cin, cout, 3, 1), device='cuda')
            lib.call('dis_conv2d_fwd_bf16x3_oihw', x, wt, 0, cout, cin, 0, b, y, None, n, h, w, cin, cout, 3, 1, 1, 0)
            lib.call('dis_conv2d_fwd_bf16x3_oihw', gy, wt, 1, cout, cin, 0, None, gx, None, n, h, w, cout, cin, 3, 1, 1, 0)
            lib.call('dis_conv2d_wgrad_bf16x3', x, gy, gw, gb, ws, n, h, w, cin, cin, cout, 3, 1, 1)
            out[tag] = (y.double().cpu(), gx.double().cpu(), gw.double().cpu())

    def err(a, ref, sel=None):
        d, r = (a - ref).abs(), ref.abs()
        if sel is not None:
            d, r = d[sel], r[sel]
        return float(d.max() / (r.max() + 1e-300))
    for name, i, ref in (('y', 0, yr), ('gx', 1, gxr), ('gw', 2, gwr)):
        e3, e2 = err(out['bf16x3'][i], ref), err(out['f16x2'][i], ref)
        print(f'{case} {cin}->{cout} {name}: of the largest entry: bf16x3 {e3:.2e}  f16x2 {e2:.2e}')
        assert e2 < 4 * e3 + 2e-7, (case, name, e3, e2)
    if case == 'outlier_pixel':
        e3, e2 = err(out['bf16x3'][0], yr, keep), err(out['f16x2'][0], yr, keep)
        print(f'  outputs that do not see the outlier (same tile included): bf16x3 {e3:.2e}  f16x2 {e2:.2e}')
        assert e2 < 4 * e3 + 2e-7, (e3, e2)
    if case == 'tiny_tiles':
        own = torch.zeros(n, h, w, dtype=torch.bool)
        own[:, :15, :15] = True       # tile (0, 0) minus the row / column whose taps reach halo row / column 16 (still tiny here)
        mixed = torch.zeros(n, h, w, dtype=torch.bool)
        mixed[:, 16:31, :31] = True   # tiny outputs of tiles whose halo (row 32) lies in the O(1) region
        mixed[:, :31, 16:31] = True
        for name, i, ref in (('y', 0, yr), ('gx', 1, gxr)):
            e3, e2 = err(out['bf16x3'][i], ref, own), err(out['f16x2'][i], ref, own)
            print(f'  {name}, all-tiny tile, of ITS largest entry: bf16x3 {e3:.2e}  f16x2 {e2:.2e}')
            assert e2 < 4 * e3 + 2e-7, (name, e3, e2)
            m3, m2 = err(out['bf16x3'][i], ref, mixed), err(out['f16x2'][i], ref, mixed)
            a2 = float((out['f16x2'][i] - ref).abs()[mixed].max() / ref.abs().max())
            print(f'  {name}, tiny outputs of tiles scaled by an O(1) halo: local relative error bf16x3 {m3:.2e}  f16x2 {m2:.2e}; '
                  f'absolute, of the largest output: {a2:.2e}')
            assert a2 < 1e-9, a2


@pytest.mark.parametrize('case', ['randn', 'outlier_pixel', 'tiny_tiles'])
def test_conv_wgrad_k4s2_f16x2(case):
    """Weight gradient of FuseNet's 4 x 4 stride-2 down convolution (32 -> 32, Block2D3D conv2_1) on the two-term fp16 kernel (all
    16 taps in one workgroup, running block scales over the tiles) against fp64, beside the exact-fp32 MFMA kernel the three-term
    mode keeps for this shape; weight and bias gradient; bar 4 x the fp32 kernel's error + 2e-7 of the largest entry.
      outlier_pixel  one input pixel and one gradient value of 1e4 among O(1) values (the running scales drop there),
      tiny_tiles     a 32 x 32 corner of x and a 16 x 16 corner of gy at 1e-6 of the rest."""
    from depthinspace_amd import lib
    from tests.conftest import conv_split
    g = torch.Generator().manual_seed(40 + len(case))
    n, h, w, c = 3, 46, 70, 32
    ho, wo = (h + 2 - 4) // 2 + 1, (w + 2 - 4) // 2 + 1
    x = torch.randn(n, h, w, c, generator=g)
    gy = torch.randn(n, ho, wo, c, generator=g)
    if case == 'outlier_pixel':
        x[1, 20, 21, :] = 1e4
        gy[2, 7, 30, 3] = 1e4
    elif case == 'tiny_tiles':
        x[:, :32, :32, :] *= 1e-6
        gy[:, :16, :16, :] *= 1e-6
    xr = x.permute(0, 3, 1, 2).double()
    wr = torch.zeros(c, c, 4, 4, dtype=torch.float64, requires_grad=True)
    br = torch.zeros(c, dtype=torch.float64, requires_grad=True)
    F.conv2d(xr, wr, br, stride=2, padding=1).backward(gy.permute(0, 3, 1, 2).double())
    out = {}
    for tag in ('bf16x3', 'f16x2'):
        with conv_split(tag):
            gw = torch.empty(c, c, 4, 4, device='cuda')
            gb = torch.empty(c, device='cuda')
            ws = torch.empty(lib.fn('dis_conv2d_wgrad_workspace')(c, c, 4, 2), device='cuda')
            lib.profile_start()
            lib.call('dis_conv2d_wgrad', x.cuda(), gy.cuda(), gw, gb, ws, n, h, w, c, c, c, 4, 2, 1)
            tags = {t for (_, _, _, t, _) in lib.profile_stop()}
            assert any(('f16x2' in t) == (tag == 'f16x2') for t in tags), (tag, tags)
            out[tag] = (gw.double().cpu(), gb.double().cpu())

    def err(a, ref):
        return float((a - ref).abs().max() / (ref.abs().max() + 1e-300))
    for name, i, ref in (('gw', 0, wr.grad), ('gb', 1, br.grad)):
        e32, e2 = err(out['bf16x3'][i], ref), err(out['f16x2'][i], ref)
        print(f'{case} {name}: fp32 MFMA {e32:.2e}  f16x2 {e2:.2e}')
        assert e2 < 4 * e32 + 2e-7, (case, name, e32, e2)


@pytest.mark.parametrize('h,w,pad', [(27, 45, 1), (16, 16, 1), (8, 19, 0), (33, 64, 2)])
@pytest.mark.parametrize('act', [0, 1, 2])
@pytest.mark.parametrize('cin,cout', [(32, 32), (16, 16), (16, 32), (32, 16)])
@pytest.mark.parametrize('split', ['bf16x3', 'f16x2'])
def test_conv_bf16x3_variants_agree(h, w, pad, act, cin, cout, split):
    """Every form of the bf16x3 convolution kernel computes the same thing: 16 / 32 channels on either side, weights
    handed over packed or as OIHW (forward and input-gradient order), with and without GroupNorm statistics, writing or
    accumulating into y; ragged tile edges, a single tile, no padding, padding wider than the halo.  split = f16x2: the
    OIHW entry points run the two-term fp16 kernels (the packed form stays bf16x3: equal to 3e-6 instead of bit for bit)."""
    from depthinspace_amd import lib
    from tests.conftest import conv_split
    with conv_split(split):
        _variants_agree(h, w, pad, act, cin, cout, split)


def _variants_agree(h, w, pad, act, cin, cout, split):
    from depthinspace_amd import lib
    g = torch.Generator().manual_seed(h * 1000 + w * 10 + act + cin * 7 + cout)
    n = 3
    x = torch.randn(n, h, w, cin, generator=g).cuda()
    b = torch.randn(cout, generator=g).cuda()
    ho, wo = h + 2 * pad - 2, w + 2 * pad - 2
    for mode in (0, 1):
        # the module's weight: (cout, cin) for the forward conv, (cin, cout) for the conv whose input gradient this is
        wt = (torch.randn(cout, cin, 3, 3, generator=g) * 0.06).cuda() if mode == 0 else \
            (torch.randn(cin, cout, 3, 3, generator=g) * 0.06).cuda()
        pk = torch.empty(lib.fn('dis_conv2d_pack_bf16x3_size')(cin, cout), dtype=torch.int16, device='cuda')
        lib.call('dis_conv2d_pack_weights_bf16x3', wt, pk, cout, cin, 3, mode)
        y0 = torch.empty(n, ho, wo, cout, device='cuda')
        lib.call('dis_conv2d_fwd_bf16x3', x, pk, b, y0, None, n, h, w, cin, cout, 3, 1, pad, act)
        # fp64 reference of the same convolution (mode 1 = channels swapped, taps flipped)
        wr = wt.double().cpu() if mode == 0 else wt.double().cpu().transpose(0, 1).flip(2, 3)
        ref = F.conv2d(x.permute(0, 3, 1, 2).double().cpu(), wr, b.double().cpu(), padding=pad)
        ref = F.selu(ref) if act == 1 else (F.relu(ref) if act == 2 else ref)
        assert relerr(y0.permute(0, 3, 1, 2), ref) < 3e-6
        # OIHW weights split inside the kernel: bit-identical to the packed path
        y1 = torch.empty_like(y0)
        st = torch.zeros(2 * n, dtype=torch.float64, device='cuda')
        lib.call('dis_conv2d_fwd_bf16x3_oihw', x, wt, mode, wt.shape[0], wt.shape[1], 0, b, y1, st, n, h, w, cin, cout, 3, 1,
                 pad, act)
        if split == 'bf16x3' or act == 2:
            assert torch.equal(y0, y1)
        else:
            assert relerr(y1.permute(0, 3, 1, 2), ref) < 3e-6
        # statistics of what was written
        s_ref = torch.stack([y1.double().sum(dim=(1, 2, 3)), (y1.double() ** 2).sum(dim=(1, 2, 3))], dim=1).reshape(-1)
        assert torch.allclose(st, s_ref, rtol=1e-6, atol=1e-4)
        # accumulate mode: y = act(y_old + conv + bias)
        yold = torch.randn(n, ho, wo, cout, generator=g).cuda()
        y2 = yold.clone()
        lib.call('dis_conv2d_fwd_bf16x3_oihw', x, wt, mode, wt.shape[0], wt.shape[1], 0, b, y2, None, n, h, w, cin, cout, 3,
                 1, pad, act | 0x100)
        pre = F.conv2d(x.permute(0, 3, 1, 2).double().cpu(), wr, b.double().cpu(), padding=pad) + \
            yold.permute(0, 3, 1, 2).double().cpu()
        ref2 = F.selu(pre) if act == 1 else (F.relu(pre) if act == 2 else pre)
        assert relerr(y2.permute(0, 3, 1, 2), ref2) < 3e-6


def test_conv_bf16x3_zero_padded_input_channels():
    """First-layer case: x carries 16 channels of which the module's conv reads the first 5 (w_i < cin)."""
    from depthinspace_amd import lib
    g = torch.Generator().manual_seed(5)
    n, h, w = 2, 21, 35
    x = torch.randn(n, h, w, 16, generator=g).cuda()
    wt = (torch.randn(16, 5, 3, 3, generator=g) * 0.1).cuda()
    b = torch.randn(16, generator=g).cuda()
    y = torch.empty(n, h, w, 16, device='cuda')
    lib.call('dis_conv2d_fwd_bf16x3_oihw', x, wt, 0, 16, 5, 0, b, y, None, n, h, w, 16, 16, 3, 1, 1, 0)
    ref = F.conv2d(x[..., :5].permute(0, 3, 1, 2).double().cpu(), wt.double().cpu(), b.double().cpu(), padding=1)
    assert relerr(y.permute(0, 3, 1, 2), ref) < 3e-6


@pytest.mark.parametrize('cin,cout', [(32, 32), (16, 16), (16, 32), (32, 16)])
@pytest.mark.parametrize('act', [1, 2])
@pytest.mark.parametrize('split', ['bf16x3', 'f16x2'])
def test_conv_bf16x3_fused_activation_gradient(cin, cout, act, split):
    """Backward of conv -> activation with the activation gradient applied while gy is staged (dis_conv2d_dgrad_bf16x3_act,
    dis_conv2d_wgrad_bf16x3_act) is bit-identical to dis_act_bwd followed by the plain kernels (either operand split; the
    two-term kernels have SELU instances only - the ReLU form is DispNetS's, which runs the slice launches)."""
    from depthinspace_amd import lib
    from tests.conftest import conv_split
    if split == 'f16x2' and act == 2:
        pytest.skip('no ReLU instance of the two-term kernels')
    with conv_split(split):
        _fused_activation_gradient(cin, cout, act)


def _fused_activation_gradient(cin, cout, act):
    from depthinspace_amd import lib
    g = torch.Generator().manual_seed(cin * 100 + cout + act)
    n, h, w = 2, 29, 37
    x = torch.randn(n, h, w, cin, generator=g).cuda()
    wt = (torch.randn(cout, cin, 3, 3, generator=g) * 0.06).cuda()
    b = torch.randn(cout, generator=g).cuda()
    y = torch.empty(n, h, w, cout, device='cuda')
    lib.call('dis_conv2d_fwd_bf16x3_oihw', x, wt, 0, cout, cin, 0, b, y, None, n, h, w, cin, cout, 3, 1, 1, act)
    gy = torch.randn(n, h, w, cout, generator=g).cuda()
    gpre = torch.empty_like(gy)
    lib.call('dis_act_bwd', gy, y, gpre, act, gy.numel())
    ws = torch.empty(lib.fn('dis_conv2d_wgrad_workspace')(cin, cout, 3, 1), device='cuda')
    for accumulate in (0, 1):
        base = torch.randn(n, h, w, cin, generator=g).cuda()
        gx0, gx1 = base.clone(), base.clone()
        lib.call('dis_conv2d_fwd_bf16x3_oihw', gpre, wt, 1, cout, cin, 0, None, gx0, None, n, h, w, cout, cin, 3, 1, 1,
                 0x100 if accumulate else 0)
        lib.call('dis_conv2d_dgrad_bf16x3_act', gy, y, act, wt, cout, cin, 0, gx1, n, h, w, cout, cin, 1, accumulate)
        assert torch.equal(gx0, gx1)
    gw0, gw1 = torch.empty_like(wt), torch.empty_like(wt)
    gb0, gb1 = torch.empty(cout, device='cuda'), torch.empty(cout, device='cuda')
    lib.call('dis_conv2d_wgrad_bf16x3', x, gpre, gw0, gb0, ws, n, h, w, cin, cin, cout, 3, 1, 1)
    lib.call('dis_conv2d_wgrad_bf16x3_act', x, gy, y, act, gw1, gb1, ws, n, h, w, cin, cin, cout, 3, 1, 1)
    assert torch.equal(gw0, gw1) and torch.equal(gb0, gb1)


@pytest.mark.gpu
@pytest.mark.parametrize('k,stride,pad', [(3, 1, 1), (4, 2, 1)])
@pytest.mark.parametrize('n,h,w', [(2, 40, 56), (1, 33, 47), (3, 16, 130)])
@pytest.mark.parametrize('act', ['selu', 'relu'])
def test_conv_wgrad_act_on_load_equals_the_separate_pass(k, stride, pad, n, h, w, act):
    """dis_conv2d_wgrad_act (round 6; FuseNet's 4 -> 16 stems conv1 / amb_conv, reference model/multi_frame_networks.py:216-233): the
    weight / bias gradient for gy * act'(y) with the product formed while gy is staged - bit-identical to dis_act_bwd followed by
    dis_conv2d_wgrad (the same fp32 product, the same kernel behind it), and within 1e-6 of the largest entry of the fp64 result."""
    from depthinspace_amd import ops
    L = ops.lib
    a = ops.ACT_SELU if act == 'selu' else ops.ACT_RELU
    if (h + 2 * pad - k) // stride + 1 <= 0:
        pytest.skip('empty output')
    g_ = torch.Generator().manual_seed(100 * k + h + w)
    cin_pad, cin, cout = 4, 3, 16
    x = torch.randn(n, h, w, cin_pad, generator=g_)
    x[..., cin:] = 0.0
    ho, wo = (h + 2 * pad - k) // stride + 1, (w + 2 * pad - k) // stride + 1
    pre = torch.randn(n, ho, wo, cout, generator=g_)
    y = (torch.nn.functional.selu(pre) if act == 'selu' else torch.relu(pre)).cuda()
    gy = torch.randn(n, ho, wo, cout, generator=g_).cuda()
    x = x.cuda()
    wsz = L.fn('dis_conv2d_wgrad_workspace')(cin_pad, cout, k, stride)
    assert wsz > 0
    gw_a, gb_a = torch.full((cout, cin, k, k), float('nan'), device='cuda'), torch.full((cout,), float('nan'), device='cuda')
    L.call('dis_conv2d_wgrad_act', x, gy, y, a, gw_a, gb_a, torch.empty(wsz, device='cuda'), n, h, w, cin_pad, cin, cout, k, stride, pad)
    gpre = torch.empty_like(gy)
    L.call('dis_act_bwd', gy, y, gpre, a, gy.numel())
    gw_b, gb_b = torch.empty(cout, cin, k, k, device='cuda'), torch.empty(cout, device='cuda')
    L.call('dis_conv2d_wgrad', x, gpre, gw_b, gb_b, torch.empty(wsz, device='cuda'), n, h, w, cin_pad, cin, cout, k, stride, pad)
    torch.cuda.synchronize()
    assert torch.equal(gw_a, gw_b) and torch.equal(gb_a, gb_b)
    xn = x[..., :cin].permute(0, 3, 1, 2).double()
    gn = gpre.permute(0, 3, 1, 2).double()
    gw64 = torch.nn.grad.conv2d_weight(xn, (cout, cin, k, k), gn, stride=stride, padding=pad)
    assert float((gw_a.double() - gw64).abs().max()) < 1e-6 * float(gw64.abs().max())
    assert float((gb_a.double() - gn.sum(dim=(0, 2, 3))).abs().max()) < 1e-6 * float(gn.sum(dim=(0, 2, 3)).abs().max())
    # another shape: refused, the caller keeps the two launches
    assert not L.call_try('dis_conv2d_wgrad_act', x, gy, y, a, gw_a, gb_a, torch.empty(wsz, device='cuda'), n, h, w, cin_pad, cin, 32, k,
                          stride, pad)


@pytest.mark.gpu
@pytest.mark.parametrize('c', [1, 2, 3, 4, 5, 32])
@pytest.mark.parametrize('n,h,w', [(3, 17, 23), (64, 32, 27), (1, 1, 1)])
def test_planar_to_nhwc_all_channel_counts(c, n, h, w):
    """dis_planar_to_nhwc: the per-pixel kernel for 2 .. 4 channels (the flow fields) and the tiled transpose for the rest are the
    same permutation as torch's"""
    from depthinspace_amd import ops
    x = torch.randn(n, c, h, w, generator=torch.Generator().manual_seed(c * 100 + h)).cuda()
    y = ops.planar_to_nhwc(x)
    assert tuple(y.shape) == (n, h, w, c)
    assert torch.equal(y, x.permute(0, 2, 3, 1).contiguous())
    assert torch.equal(ops.nhwc_to_planar(y), x)


@pytest.mark.gpu
@pytest.mark.parametrize('tl,bs,c,h,w,slots', [(4, 2, 32, 37, 45, 256), (4, 1, 16, 16, 70, 256), (3, 2, 32, 24, 200, 8), (4, 4, 32, 64, 54, 256)])
def test_gather_bwd_with_group_norm_sums(tl, bs, c, h, w, slots):
    """dis_gather_warped_feat_bwd_csr_gnres (round 6): the feature warp's backward when its input IS y = SELU(GroupNorm(x2) + residual)
    and the launch completes the gradient wrt y - against the two launches it replaces: the stored g SELU'(y) is bit-identical to
    dis_gather_warped_feat_bwd_csr followed by dis_gn_bwd_res_sums' residual gradient, the channel sums agree to 1e-6 of the largest
    (other partial-sum partition); also with more tiles per image than slots (fp64 atomics into shared slots)."""
    from depthinspace_amd import ops
    L = ops.lib
    g = torch.Generator().manual_seed(tl * 100 + c + h)
    flows = (torch.randn(tl * tl, bs, h, w, 2, generator=g) * 5).cuda()
    go = torch.randn(tl, bs, h, w, tl, c, generator=g).cuda()
    init = torch.randn(tl, bs, h, w, c, generator=g).cuda()
    x2 = torch.randn(tl, bs, h, w, c, generator=g).cuda()
    y = torch.nn.functional.selu(torch.randn(tl, bs, h, w, c, generator=g)).cuda()
    csr = ops.gather_csr(flows)
    n = tl * bs
    # the two launches
    gf = init.clone()
    L.call('dis_gather_warped_feat_bwd_csr', go, csr, gf, gf, tl, bs, h, w, c)
    s_ref = int(L.fn('dis_conv2d_gnsums_slots')())
    ab_ref = torch.zeros(n * s_ref * 2 * c, dtype=torch.float64, device='cuda')
    gres_ref = torch.empty_like(gf)
    L.call('dis_gn_bwd_res_sums', gf, y, x2, gres_ref, ab_ref, s_ref, n, h * w, c, ops.ACT_SELU)
    # one launch
    gf1 = init.clone()
    ab = torch.zeros(n * slots * 2 * c, dtype=torch.float64, device='cuda')
    ok = L.call_try('dis_gather_warped_feat_bwd_csr_gnres', go, csr, gf1, gf1, y, x2, ab, slots, ops.ACT_SELU, tl, bs, h, w, c)
    if not ok:
        pytest.skip('no tiled instance for this shape')
    torch.cuda.synchronize()
    assert torch.equal(gf1, gres_ref)
    got, ref = ab.view(n, slots, 2, c).sum(dim=1), ab_ref.view(n, s_ref, 2, c).sum(dim=1)
    assert float((got - ref).abs().max()) <= 1e-6 * float(ref.abs().max()), float((got - ref).abs().max())
    # fixed order where every tile has its own slot: repeats bit for bit
    if slots >= ((h + 7) // 8) * ((w + 31) // 32):
        gf2 = init.clone()
        ab2 = torch.zeros_like(ab)
        L.call('dis_gather_warped_feat_bwd_csr_gnres', go, csr, gf2, gf2, y, x2, ab2, slots, ops.ACT_SELU, tl, bs, h, w, c)
        assert torch.equal(ab, ab2) and torch.equal(gf1, gf2)
