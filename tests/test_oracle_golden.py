"""CPU: the oracle restatement against the golden vectors produced by the imported reference
(oracle/make_golden.py).  Sized to run in about a minute on 8 cores."""
import os
import numpy as np
import pytest
import torch

from oracle import dis_oracle as O
from depthinspace_amd import synth


@pytest.fixture(scope='module')
def G(golden_dir):
    return np.load(os.path.join(golden_dir, 'ops.npz'))


def t(x):
    return torch.from_numpy(np.asarray(x))


def test_lcn(G):
    l, s = O.lcn(t(G['lcn_x']))
    assert torch.equal(l, t(G['lcn_out'])) and torch.equal(s, t(G['lcn_std']))


@pytest.mark.parametrize('name', ['mse', 'sad', 'census_mse', 'census_sad'])
def test_photometric(G, name):
    for blk, eps in ((9, 0.5), (5, 0.1)):
        es = t(G['ph_es']).clone().requires_grad_(True)
        y = O.photometric(es, t(G['ph_ta']), blk, name, eps)
        y.backward(t(G['ph_go']))
        assert torch.allclose(y, t(G[f'ph_{name}_{blk}_out']), rtol=1e-5, atol=1e-5)
        assert torch.allclose(es.grad, t(G[f'ph_{name}_{blk}_grad']), rtol=1e-4, atol=1e-6)


def test_pattern_smooth_d2d_warp_resize(G):
    pat = torch.cat([t(G['pl_pat'])] * 3, 1).mean(dim=1, keepdim=True)
    d = t(G['pl_disp']).clone().requires_grad_(True)
    v, proj = O.pattern_loss(pat, d, t(G['pl_im']), t(G['pl_std']))
    v.backward()
    assert abs(float(v) - float(G['pl_val'])) < 1e-6
    assert torch.allclose(proj, t(G['pl_proj']), atol=1e-6)
    assert torch.allclose(d.grad, t(G['pl_grad']), rtol=1e-4, atol=1e-8)
    d = t(G['sm_disp']).clone().requires_grad_(True)
    v = O.smooth_loss(d, t(G['sm_amb']))
    v.backward()
    assert abs(float(v) - float(G['sm_val'])) < 1e-7 and torch.allclose(d.grad, t(G['sm_grad']), atol=1e-9)
    assert torch.equal(O.disp_to_depth(t(G['d2d_in']), 435.2, 0.025), t(G['d2d_out']))
    assert torch.equal(O.warp(t(G['warp_x']), t(G['warp_flow'])), t(G['warp_out']))
    assert torch.equal(O.resize_ac(t(G['warp_x']), (10, 12)), t(G['resize_out']))
    assert torch.equal(O.resize_flow({'a': t(G['warp_flow'])}, (10, 12))['a'], t(G['resize_flow_out']))


@pytest.mark.parametrize('stride', [1, 2])
def test_conv3d(G, stride):
    p = O.init_params({k: v for k, v in O.mf_param_shapes().items() if k.startswith('blocks.0.conv3d_1')}, seed=5)
    f = t(G['c3_feat']).clone().requires_grad_(True)
    y, idx, key = O.conv3d_knn(p, 'blocks.0.conv3d_1', t(G['c3_xyz']), f, t(G['c3_mask']), stride, 4, return_index=True)
    y.backward(t(G[f'c3_s{stride}_go']))
    assert torch.allclose(y, t(G[f'c3_s{stride}_out']), rtol=1e-4, atol=1e-5)
    assert (np.sort(idx.numpy(), -1) == G[f'c3_s{stride}_idx_sorted']).all()
    ref = t(G[f'c3_s{stride}_gfeat'])
    assert float((f.grad - ref).abs().max()) < 1e-5 * float(ref.abs().max()) + 1e-7


@pytest.mark.parametrize('name', ['mf_64_bs1', 'mf_64_bs2_rnd', 'sf_64_bs1', 'mf_64_real_sgm', 'sf_64_real_sgm'])
def test_step_golden(golden_dir, name):
    """(i)-(v) of SURVEY.md section 8(c): data after copy_data, outputs, ordered loss terms, gradients, Adam."""
    Gs = np.load(os.path.join(golden_dir, name + '.npz'))
    arch = str(Gs['arch'])
    H, W, bs = int(Gs['H']), int(Gs['W']), int(Gs['bs'])
    settings = synth.make_settings(H, W, pattern=str(Gs['pattern']))
    mk = synth.make_random_batch if int(Gs['random_batch']) else synth.make_batch
    batch = mk(settings, bs, 4, seed=int(Gs['bseed']), with_pseudo_gt=bool(int(Gs['use_pseudo_gt'])))
    real_sgm = 'real_sgm' in Gs.files   # `real` data in the warm-up epochs: SGM disparities + the reference's recorded draws
    if real_sgm:
        batch['sgm_disp'] = Gs['sgm_disp']
        for k in range(4):
            if f'sgm_noise{k}' in Gs.files:
                batch[f'_sgm_noise{k}'] = Gs[f'sgm_noise{k}']
    shapes = O.mf_param_shapes() if arch == 'multi_frame' else O.sf_param_shapes()
    params = O.init_params(shapes, seed=int(Gs['pseed']))
    ctx = O.StepContext(settings)
    st = {'step': 0, 'm': {}, 'v': {}}
    res = O.train_step(ctx, arch, params, {k: t(v) for k, v in batch.items()}, adam_state=st, epoch=int(Gs['epoch']),
                       use_pseudo_gt=bool(int(Gs['use_pseudo_gt'])), data_type='real' if real_sgm else 'synthetic')
    assert len(res['vals']) == len(Gs['vals'])
    outs = res['out'] if isinstance(res['out'], (list, tuple)) else [res['out']]
    for i, o in enumerate(outs):
        assert float((o.detach() - t(Gs[f'out{i}'])).abs().max()) < 1e-5
    np.testing.assert_allclose([float(v) for v in res['vals']], Gs['vals'], rtol=1e-5, atol=1e-7)
    assert abs(float(res['data']['std0'].double().sum()) - float(Gs['std0_sum'])) < 1e-6 * abs(float(Gs['std0_sum']))
    keys = list(Gs['grad_keys'])
    for i, k in enumerate(keys):
        g = res['grads'][k]
        if bool(Gs['grad_none'][i]):
            assert g is None or float(g.abs().max()) == 0.0
            continue
        assert abs(float(g.double().norm()) - float(Gs['grad_l2'][i])) <= 1e-4 * float(Gs['grad_l2'][i]) + 1e-12, k
        if 'grad:' + k in Gs.files:
            ref = t(Gs['grad:' + k])
            assert float((g - ref).abs().max()) <= 2e-5 * float(ref.abs().max()) + 1e-12, k


def test_geometry_against_numpy_convention():
    """independent cross-check of unproject/project against the numpy statement of the same camera convention
    (reference co/geometry.py:515-545: xyz = (d*uvn - t) R ; uvd = K (R xyz + t))."""
    st = synth.make_settings(32, 40)
    b = synth.make_batch(st, 1, 4, seed=3)
    K = st.K.astype(np.float64)
    d = 0.025 * 435.2 / b['disp0'][0, 0, 0].astype(np.float64)
    R0, t0, R1, t1 = [b[k][0, i].astype(np.float64) for k, i in (('R', 0), ('t', 0), ('R', 1), ('t', 1))]
    u, v = np.meshgrid(np.arange(40), np.arange(32))
    uvn = np.stack([u, v, np.ones_like(u)], -1).reshape(-1, 3) @ np.linalg.inv(K).T
    xyz = (d.reshape(-1, 1) * uvn - t0) @ R0
    uvd = (xyz @ R1.T + t1) @ K.T
    ray = O.make_rays(st.K, 32, 40)
    uv, dd = O.project(O.unproject(t(b['disp0'][0:1, 0]) * 0 + t(d.astype(np.float32)).view(1, 1, 32, 40), ray,
                                   t(b['R'][0:1, 0]), t(b['t'][0:1, 0])), t(st.K), t(b['R'][0:1, 1]), t(b['t'][0:1, 1]))
    assert np.allclose(dd[0, :, 0].numpy(), uvd[:, 2], rtol=1e-5)
    assert np.allclose(uv[0].numpy(), uvd[:, :2] / uvd[:, 2:3], atol=2e-3)
    # exact rigid flow of the synthetic scene agrees with the projection
    assert np.allclose(b['flow_01'][0, 0, 0].reshape(-1), uvd[:, 0] / uvd[:, 2] - u.reshape(-1), atol=2e-3)


STEP_FIXTURES = ['mf_64_bs1', 'mf_64_bs2_rnd', 'mf_128_bs1', 'mf_128_bumps', 'mf_64_real_sgm', 'sf_64_bs1', 'sf_128_bs1_pgt',
                 'sf_128x108_bs1', 'sf_128_real_pgt', 'sf_64_real_sgm', 'mf_512x432_bs1']


def test_committed_fixtures_are_what_the_generator_writes(golden_dir):
    """Every step fixture carries the generator's current field set (`torch_threads`: ATen picks its CPU rounding chains by
    thread count for a few ops, so index-class parity is parity with the 8-thread reference run) ..."""
    for name in STEP_FIXTURES:
        Gs = np.load(os.path.join(golden_dir, name + '.npz'))
        assert 'torch_threads' in Gs.files and int(Gs['torch_threads']) == 8, name


def test_fixtures_regenerate_bit_for_bit(golden_dir, tmp_path):
    """... and regenerating them from the imported reference (a child process: the import recipe monkey-patches torch) reproduces
    the committed files array for array, the `real`-data SGM fixtures included: the reference draws their noise from torch's
    global generator, which the generator script seeds per case.  Needs /root/reference (this container only; the GPU box skips)."""
    import subprocess
    import sys
    if not os.path.isdir('/root/reference/model'):
        pytest.skip('reference checkout not present')
    names = ['sf_64_real_sgm', 'mf_64_real_sgm', 'sf_64_bs1', 'mf_64_bs1']
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, 'oracle', 'make_golden.py'), '--out', str(tmp_path)] + names,
                       capture_output=True, text=True, cwd=root, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    for name in names:
        new, old = np.load(os.path.join(str(tmp_path), name + '.npz')), np.load(os.path.join(golden_dir, name + '.npz'))
        assert sorted(new.files) == sorted(old.files), name
        for k in old.files:
            assert np.array_equal(new[k], old[k]), (name, k)
