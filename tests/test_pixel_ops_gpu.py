"""HIP per-pixel operators (through the C ABI) vs the CPU oracle and the reference goldens."""
import os
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import dis_oracle as O


def dev(x):
    return torch.as_tensor(x).cuda()


def close(a, b, atol, rtol=0.0, what=''):
    a = a.detach().cpu().double()
    b = torch.as_tensor(b).detach().cpu().double()
    err = (a - b).abs()
    tol = atol + rtol * b.abs()
    assert bool((err <= tol).all()), f'{what}: max err {float(err.max()):.3e} (tol {atol:.1e}+{rtol:.1e}*|ref|)'


@pytest.fixture(scope='module')
def G(golden_dir):
    return np.load(os.path.join(golden_dir, 'ops.npz'))


def test_lcn_golden(G):
    from depthinspace_amd import ops
    l, s = ops.lcn(dev(G['lcn_x']))
    close(s, G['lcn_std'], 2e-5, what='lcn std')
    close(l, G['lcn_out'], 2e-4, what='lcn out')


def test_lcn_full_size():
    from depthinspace_amd import ops, synth
    st = synth.make_settings()
    b = synth.make_batch(st, 1, 4, seed=5, with_flow=False)
    x = torch.from_numpy(b['im0'][0])
    l, s = ops.lcn(x.cuda())
    ol, os_ = O.lcn(x)
    close(s, os_, 2e-5, what='std')
    close(l, ol, 3e-4, what='lcn')


@pytest.mark.parametrize('name', ['mse', 'sad', 'census_mse', 'census_sad'])
@pytest.mark.parametrize('blk,eps', [(9, 0.5), (5, 0.1)])
@pytest.mark.parametrize('via_multi', [True, False])
def test_photometric_golden(G, name, blk, eps, via_multi, monkeypatch):
    """(via_multi False = DIS_PHOTO_SINGLE_VIA_MULTI=0: the general single-estimate kernels stay pinned to the reference's values
    too, now that ops.photometric() routes the 9 x 9 census calls through the multi-estimate kernels by default)"""
    from depthinspace_amd import ops
    monkeypatch.setattr(ops, 'PHOTO_SINGLE_VIA_MULTI', via_multi)
    es = dev(G['ph_es']).requires_grad_(True)
    out = ops.photometric(es, dev(G['ph_ta']), blk, O.PHOTO_TYPES[name], eps)
    out.backward(dev(G['ph_go']))
    close(out, G[f'ph_{name}_{blk}_out'], 1e-5, 1e-5, what='fwd')
    close(es.grad, G[f'ph_{name}_{blk}_grad'], 2e-6, 2e-5, what='bwd')


def test_photometric_ragged_sizes():
    from depthinspace_amd import ops
    g = torch.Generator().manual_seed(3)
    for (n, c, h, w) in [(1, 1, 9, 9), (2, 2, 33, 41), (1, 1, 8, 70)]:
        es = torch.randn(n, c, h, w, generator=g)
        ta = torch.randn(n, c, h, w, generator=g)
        go = torch.rand(n, 1, h, w, generator=g)
        e1 = es.clone().requires_grad_(True)
        y = O.photometric(e1, ta, 9, 'census_sad', 0.5)
        y.backward(go)
        e2 = es.cuda().requires_grad_(True)
        y2 = ops.photometric(e2, ta.cuda(), 9, 3, 0.5)
        y2.backward(go.cuda())
        close(y2, y, 1e-5, 1e-5, what=f'fwd {n,c,h,w}')
        close(e2.grad, e1.grad, 2e-6, 2e-5, what=f'bwd {n,c,h,w}')


@pytest.mark.parametrize('type_id', [2, 3])
@pytest.mark.parametrize('s_n_h_w', [(4, 2, 40, 70), (1, 1, 9, 9), (3, 1, 8, 130), (2, 3, 33, 41), (4, 1, 64, 64)])
def test_photometric_multi_equals_per_estimate_calls(type_id, s_n_h_w):
    """dis_photometric_fwd_multi / _bwd_multi (S estimates against one target, the target's census terms shared): equal to S
    calls of the single-estimate kernels to rounding (fused multiply-adds; the rare near-zero-difference sign path corrects the
    sum instead of branching inside it); ragged tiles, images smaller than the window, borders; and against the oracle."""
    from depthinspace_amd import ops
    s, n, h, w = s_n_h_w
    g = torch.Generator().manual_seed(s * 1000 + n * 100 + h + w + type_id)
    ta = torch.randn(n, 1, h, w, generator=g)
    es = [torch.randn(n, 1, h, w, generator=g) for _ in range(s)]
    es[0][:, :, : h // 2] = ta[:, :, : h // 2]            # exact zeros of h(des) - h(dta): the sign path of census_sad
    go = [torch.rand(n, 1, h, w, generator=g) for _ in range(s)]
    assert ops.photometric_multi_ok(s, 1, 9, type_id)
    e_m = [e.cuda().requires_grad_(True) for e in es]
    outs = ops.photometric_multi(e_m, ta.cuda(), 9, type_id, 0.5)
    sum((o * g_.cuda()).sum() for o, g_ in zip(outs, go)).backward()
    name = {2: 'census_mse', 3: 'census_sad'}[type_id]
    for k in range(s):
        e1 = es[k].cuda().requires_grad_(True)
        y1 = ops._Photometric.apply(e1, ta.cuda(), 9, type_id, 0.5)   # (the general single-estimate kernels)
        y1.backward(go[k].cuda())
        close(outs[k], y1, 1e-7, 2e-6, what=f'fwd vs single {k}')   # (fused multiply-adds in the multi kernels: rounding level)
        close(e_m[k].grad, e1.grad, 1e-7, 1e-6, what=f'bwd vs single {k}')
        eo = es[k].clone().requires_grad_(True)
        yo = O.photometric(eo, ta, 9, name, 0.5)
        yo.backward(go[k])
        close(outs[k], yo, 1e-5, 1e-5, what=f'fwd vs oracle {k}')
        if k > 0:   # (estimate 0 has exact ties, where |.|' is a convention: the HIP kernels follow the reference's autograd, checked above against the single kernel)
            close(e_m[k].grad, eo.grad, 2e-6, 2e-5, what=f'bwd vs oracle {k}')


def test_photometric_rejects_bad_args():
    from depthinspace_amd import ops, lib
    x = torch.zeros(1, 1, 8, 8).cuda()
    with pytest.raises(lib.DisHipError):
        ops.photometric(x, x, 8, 3, 0.5)   # even block
    with pytest.raises(lib.DisHipError):
        ops.photometric(x, x, 9, 7, 0.5)   # unknown type
    with pytest.raises(RuntimeError):
        ops.photometric(torch.zeros(1, 1, 8, 8), torch.zeros(1, 1, 8, 8), 9, 3, 0.5)  # CPU tensors: no fallback


def test_pattern_loss_golden(G):
    from depthinspace_amd import ops
    pat = torch.from_numpy(G['pl_pat'])
    pat1 = torch.cat([pat] * 3, 1).mean(dim=1, keepdim=True).contiguous()
    disp = dev(G['pl_disp']).requires_grad_(True)
    proj = ops.pattern_warp(pat1.cuda(), disp)
    diff = ops.photometric(proj, dev(G['pl_im']), 9, 3, 0.5)
    val = ops.weighted_mean(diff, dev(G['pl_std']))
    val.backward()
    close(proj, G['pl_proj'], 2e-6, 1e-5, what='proj')
    close(val, float(G['pl_val']), 1e-6, 1e-5, what='val')
    close(disp.grad, G['pl_grad'], 1e-8, 2e-4, what='grad')


@pytest.mark.parametrize('s', [1, 3, 4])
def test_pattern_photo_loss_one_node_equals_the_separate_nodes(G, s):
    """ops.pattern_photo_loss_multi (pattern warp -> census loss of S estimates in one launch -> weighted means as ONE autograd node,
    no stack / cat of the S maps): the same launches as the separate nodes, so values and disparity gradients are bit-identical; and
    the golden vector of the single-estimate chain holds for estimate 0."""
    from depthinspace_amd import ops
    pat = torch.from_numpy(G['pl_pat'])
    pat1 = torch.cat([pat] * 3, 1).mean(dim=1, keepdim=True).contiguous().cuda()
    im, std = dev(G['pl_im']), dev(G['pl_std'])
    g = torch.Generator().manual_seed(s)
    d0 = torch.from_numpy(G['pl_disp'])
    base = [d0] + [d0 + 0.7 * k * torch.rand(d0.shape, generator=g) for k in range(1, s)]
    wts = [1.0, 0.5, 0.25, 0.125][:s]
    da = [b.cuda().requires_grad_(True) for b in base]
    va = ops.pattern_photo_loss_multi(pat1, da, im, std, 9, 3, 0.5)
    sum(v * w_ for v, w_ in zip(va, wts)).backward()
    db = [b.cuda().requires_grad_(True) for b in base]
    projs = [ops.pattern_warp(pat1, d) for d in db]
    vb = [ops.weighted_mean(x, std) for x in ops.photometric_multi(projs, im, 9, 3, 0.5)]
    sum(v * w_ for v, w_ in zip(vb, wts)).backward()
    for k in range(s):
        assert torch.equal(va[k], vb[k]), (k, float(va[k]), float(vb[k]))
        assert torch.equal(da[k].grad, db[k].grad), k
    close(va[0], float(G['pl_val']), 1e-6, 1e-5, what='val')
    close(da[0].grad, G['pl_grad'], 1e-8, 2e-4, what='grad')
    # an estimate whose value is not used gets a zero gradient
    dc = [b.cuda().requires_grad_(True) for b in base]
    vc = ops.pattern_photo_loss_multi(pat1, dc, im, std, 9, 3, 0.5)
    vc[0].backward()
    assert torch.equal(dc[0].grad, da[0].grad / 1.0) or float((dc[0].grad - da[0].grad).abs().max()) < 1e-12
    for k in range(1, s):
        assert dc[k].grad is None or float(dc[k].grad.abs().max()) == 0.0


def test_smooth_golden(G):
    from depthinspace_amd import ops
    disp = dev(G['sm_disp']).requires_grad_(True)
    val = ops.smooth_loss(disp, dev(G['sm_amb']))
    val.backward()
    close(val, float(G['sm_val']), 1e-7, 1e-5, what='val')
    close(disp.grad, G['sm_grad'], 1e-9, 1e-4, what='grad')


def test_d2d_golden(G):
    from depthinspace_amd import ops
    d = dev(G['d2d_in']).requires_grad_(True)
    y = ops.disp_to_depth(d, 0.025 * 435.2)
    close(y, G['d2d_out'], 0, 1e-6, what='d2d')
    d1 = torch.from_numpy(G['d2d_in']).requires_grad_(True)
    y1 = O.disp_to_depth(d1, 435.2, 0.025)
    go = torch.rand(y1.shape) * 1e-3
    y1.backward(go)
    y.backward(go.cuda())
    close(d.grad, d1.grad, 1e-12, 1e-5, what='d2d grad')


def test_l1_mean():
    from depthinspace_amd import ops
    g = torch.Generator().manual_seed(1)
    a = torch.randn(3, 1, 17, 19, generator=g)
    b = torch.randn(3, 1, 17, 19, generator=g)
    a1 = a.clone().requires_grad_(True)
    v1 = (a1 - b).abs().mean()
    v1.backward()
    a2 = a.cuda().requires_grad_(True)
    v2 = ops.l1_mean(a2, b.cuda())
    (v2 * 0.1).backward()
    close(v2, v1, 1e-7, 1e-6)
    close(a2.grad, a1.grad * 0.1, 1e-10, 1e-6)


@pytest.mark.parametrize('mode', ['mf', 'sf'])
def test_geo_loss_golden(G, mode):
    from depthinspace_amd import ops, synth, lib
    st = synth.make_settings(48, 56)
    b = synth.make_random_batch(st, 2, 4, seed=int(G['ge_seed']))
    tb = {k: torch.from_numpy(v).transpose(0, 1).contiguous() if v.ndim > 2 else torch.from_numpy(v) for k, v in b.items()}
    K = lib.host_floats(st.K.reshape(-1))
    Ki = lib.host_floats(np.linalg.inv(st.K).reshape(-1))
    bf = float(st.K[0, 0]) * st.baseline
    disp = dev(G['ge_disp']).requires_grad_(True)
    depth = ops.disp_to_depth(disp, bf)
    pdepth = ops.disp_to_depth(tb['primary_disp'].cuda(), bf)
    i, j = 0, 2
    c = lambda t: t.contiguous().cuda()
    f0, f1 = c(tb['flow_02'][0]), c(tb['flow_20'][0])
    a0, a1 = c(tb['ambient0'][i]), c(tb['ambient0'][j])
    R, t = tb['R'].cuda(), tb['t'].cuda()
    if mode == 'mf':
        l0, _ = ops.geo_loss_dir(depth[i], depth[j], f0, f1, a0, a1, pdepth[j], R[i], t[i], R[j], t[j], K, Ki, -1.0)
        l1, _ = ops.geo_loss_dir(depth[j], depth[i], f1, f0, a1, a0, pdepth[i], R[j], t[j], R[i], t[i], K, Ki, -1.0)
    else:
        l0, _ = ops.geo_loss_dir(depth[i], depth[j], f0, f1, a0, a1, None, R[i], t[i], R[j], t[j], K, Ki, 0.1)
        l1, _ = ops.geo_loss_dir(depth[j], depth[i], f1, f0, a1, a0, None, R[j], t[j], R[i], t[i], K, Ki, 0.1)
    val = l0 + l1
    val.backward()
    close(val, float(G[f'ge_{mode}_val']), 1e-7, 2e-5, what='val')
    ref = torch.from_numpy(G[f'ge_{mode}_grad'])
    scale = float(ref.abs().max())
    close(disp.grad, ref, 2e-5 * scale, 1e-4, what='grad')


@pytest.mark.parametrize('mode', ['mf', 'sf'])
@pytest.mark.parametrize('hw', [(48, 56), (130, 94)])
def test_geo_loss_all_terms_in_one_launch(G, mode, hw):
    """dis_geo_loss_fwd_multi / _bwd_multi (round 5): the 12 directional terms of a step in one launch each way against 12 calls of
    dis_geo_loss_fwd / _bwd - values and masks bit for bit (same pixel partition, same sums), the depth gradient to rounding (the
    multi-term backward adds both gradients with atomics); the golden pair (0, 2) of test_geo_loss_golden among them."""
    from depthinspace_amd import ops, synth, lib
    st = synth.make_settings(*hw)
    b = synth.make_random_batch(st, 2, 4, seed=int(G['ge_seed']))
    tb = {k: torch.from_numpy(v).transpose(0, 1).contiguous() if v.ndim > 2 else torch.from_numpy(v) for k, v in b.items()}
    K = lib.host_floats(st.K.reshape(-1))
    Ki = lib.host_floats(np.linalg.inv(st.K).reshape(-1))
    bf = float(st.K[0, 0]) * st.baseline
    tl = 4
    disp0 = dev(G['ge_disp']) if hw == (48, 56) else (tb['primary_disp'] * 1.03 + 0.4).cuda()
    pdepth = ops.disp_to_depth(tb['primary_disp'].cuda(), bf) if mode == 'mf' else None
    amb, R, t = tb['ambient0'].contiguous().cuda(), tb['R'].cuda(), tb['t'].cuda()
    flow = {k: v[0].contiguous().cuda() if v.dim() == 5 else v.contiguous().cuda() for k, v in tb.items() if k.startswith('flow_')}
    clamp = -1.0 if mode == 'mf' else 0.1
    pairs, flows = [], []
    for i in range(tl):
        for j in range(i + 1, tl):
            pairs += [(i, j), (j, i)]
            flows += [(flow[f'flow_{i}{j}'], flow[f'flow_{j}{i}']), (flow[f'flow_{j}{i}'], flow[f'flow_{i}{j}'])]
    gvec = torch.linspace(0.5, 1.5, len(pairs)).cuda()
    # one launch each way
    d1 = disp0.clone().requires_grad_(True)
    depth = ops.disp_to_depth(d1, bf)
    vals = ops.geo_loss_all(depth, amb, pdepth, R, t, K, Ki, clamp, pairs, flows)
    (vals * gvec).sum().backward()
    # one call per term
    d2 = disp0.clone().requires_grad_(True)
    depth2 = ops.disp_to_depth(d2, bf)
    singles = []
    for (i, j), (f0, f1) in zip(pairs, flows):
        v, _ = ops.geo_loss_dir(depth2[i], depth2[j], f0, f1, amb[i], amb[j], pdepth[j] if pdepth is not None else None, R[i], t[i], R[j],
                                t[j], K, Ki, clamp)
        singles.append(v)
    sv = torch.stack(singles)
    (sv * gvec).sum().backward()
    assert torch.equal(vals, sv), (vals, sv)
    # a subset of the terms (three frames: six terms) gives the same six values
    sub = [k for k, (i, j) in enumerate(pairs) if i < 3 and j < 3]
    v3 = ops.geo_loss_all(depth2.detach(), amb, pdepth, R, t, K, Ki, clamp, [pairs[k] for k in sub], [flows[k] for k in sub])
    assert torch.equal(v3, vals[sub])
    scale = float(d2.grad.abs().max())
    assert scale > 0 and float((d1.grad - d2.grad).abs().max()) < 2e-6 * scale
    if hw == (48, 56):   # the golden pair
        k = pairs.index((0, 2))
        close(vals[k] + vals[k + 1], float(G[f'ge_{mode}_val']), 1e-7, 2e-5, what='val')


@pytest.mark.parametrize('cfg', [(48, 56, 2, 3, True), (64, 64, 1, 1234, False), (130, 94, 1, 7, True)])
@pytest.mark.parametrize('mode', ['mf', 'sf'])
def test_geo_loss_masks_bit_exact(cfg, mode):
    """The fb / vc / rf masks of the flow-consistency losses (reference model/networks.py:584-595, 644-651) are index-class
    outputs: the HIP kernel's mask equals tests/bitexact.py (== the oracle, tests/test_bitexact_cpu.py) on EVERY pixel."""
    from depthinspace_amd import ops, synth, lib
    from tests import bitexact as B
    H, W, bs, seed, rnd = cfg
    st = synth.make_settings(H, W)
    b = (synth.make_random_batch if rnd else synth.make_batch)(st, bs, 4, seed=seed)
    tb = {k: torch.from_numpy(v).transpose(0, 1).contiguous() if v.ndim > 2 else torch.from_numpy(v) for k, v in b.items()}
    K = lib.host_floats(st.K.reshape(-1))
    Ki = lib.host_floats(np.linalg.inv(st.K).reshape(-1))
    bf = float(st.K[0, 0]) * st.baseline
    g = torch.Generator().manual_seed(5)
    disp = tb['disp0'] + 0.05 * torch.randn(tb['disp0'].shape, generator=g)
    depth = ops.disp_to_depth(disp.cuda(), bf)
    pdepth = ops.disp_to_depth(tb['primary_disp'].cuda(), bf)
    e_depth = B.disp_to_depth(disp.numpy(), float(st.K[0, 0]), st.baseline)
    e_pdepth = B.disp_to_depth(tb['primary_disp'].numpy(), float(st.K[0, 0]), st.baseline)
    assert np.array_equal(depth.cpu().numpy(), e_depth)
    ray = O.make_rays(st.K, H, W).numpy()
    c = lambda t: t.contiguous().cuda()
    R, t = tb['R'].cuda(), tb['t'].cuda()
    for i, j in ((0, 1), (2, 0), (3, 2)):
        f0, f1 = tb[f'flow_{i}{j}'][0], tb[f'flow_{j}{i}'][0]
        a0, a1 = tb['ambient0'][i], tb['ambient0'][j]
        _, mask = ops.geo_loss_dir(depth[i], depth[j], c(f0), c(f1), c(a0), c(a1), pdepth[j] if mode == 'mf' else None,
                                   R[i], t[i], R[j], t[j], K, Ki, -1.0 if mode == 'mf' else 0.1)
        m, _ = B.flow_consistency_mask(st.K, ray, e_depth[i], tb['R'][i].numpy(), tb['t'][i].numpy(), tb['R'][j].numpy(),
                                       tb['t'][j].numpy(), f0.numpy(), f1.numpy(), a0.numpy(), a1.numpy(),
                                       primary_depth1=e_pdepth[j] if mode == 'mf' else None)
        assert np.array_equal(mask.cpu().numpy(), m), (i, j, float((mask.cpu().numpy() != m).mean()))
        assert 0.02 < float(m.mean()) < 0.999 or not rnd  # the random batch exercises both outcomes
