"""CPU: tests/bitexact.py (the rounding-exact numpy statement of the index-class parts of the step, which the HIP kernels
follow operation for operation) equals the oracle - and through the goldens the reference itself - BIT FOR BIT:
FuseNet's geometry pyramids and fb masks, Conv3D's keys and top-9 (ids and order), the loss masks."""
import os
import numpy as np
import pytest
import torch

from oracle import dis_oracle as O
from depthinspace_amd import synth
from tests import bitexact as B


def _inputs(H, W, bs, seed, rnd, **kw):
    st = synth.make_settings(H, W)
    b = synth.make_random_batch(st, bs, 4, seed=seed) if rnd else synth.make_batch(st, bs, 4, seed=seed, **kw)
    tb = {k: torch.from_numpy(v).transpose(0, 1).contiguous() if v.ndim > 2 else torch.from_numpy(v) for k, v in b.items()}
    return st, tb


def _oracle_geometry(st, tb, H, W):
    h, w = H // 2, W // 2
    depth = O.disp_to_depth(tb['primary_disp'], float(st.K[0, 0]), st.baseline)
    depth_core = O.resize_ac(depth, (h, w))
    flow = {k: v[0] for k, v in tb.items() if k.startswith('flow_')}
    flow_core = O.resize_flow(flow, (h, w))
    ray = O.mf_core_rays(st.K, H, W)
    wxyz, wmask = O.mf_geometry(depth_core, ray, tb['R'], tb['t'], flow_core)
    return depth, depth_core, flow, flow_core, ray, wxyz, wmask


@pytest.mark.parametrize('cfg', [(64, 64, 1, 1234, False, {}), (64, 48, 2, 8, True, {}),
                                 (128, 128, 1, 4321, False, dict(scene='bumps', motion=1.5)),
                                 (256, 216, 1, 5, False, {})])
def test_geometry_keys_and_topk_bit_exact(cfg):
    H, W, bs, seed, rnd, kw = cfg
    st, tb = _inputs(H, W, bs, seed, rnd, **kw)
    h, w = H // 2, W // 2
    depth, depth_core, flow, flow_core, ray, wxyz, wmask = _oracle_geometry(st, tb, H, W)
    eq = np.array_equal
    assert eq(B.disp_to_depth(tb['primary_disp'].numpy(), float(st.K[0, 0]), st.baseline), depth.numpy())
    assert eq(B.resize_ac(depth.numpy(), h, w), depth_core.numpy())
    for k in flow:
        assert eq(B.resize_flow(flow[k].numpy(), h, w), flow_core[k].numpy()), k
    ex, em = B.mf_geometry(depth_core.numpy(), ray.numpy(), tb['R'].numpy(), tb['t'].numpy(),
                           {k: v.numpy() for k, v in flow_core.items()})
    assert eq(ex, wxyz.numpy()) and eq(em, wmask.numpy())
    hq, wq = (h + 2 - 3) // 2 + 1, (w + 2 - 3) // 2 + 1
    wxyz_q = O.resize_ac(wxyz, (hq, wq))
    wmask_q = (O.resize_ac(wmask, (hq, wq)) > 0.5).float()
    assert eq(B.resize_ac(wxyz.numpy(), hq, wq), wxyz_q.numpy())
    assert eq((B.resize_ac(wmask.numpy(), hq, wq) > 0.5).astype(np.float32), wmask_q.numpy())
    pp = O.init_params({k: v for k, v in O.mf_param_shapes().items() if k.startswith('blocks.0.conv3d_1')}, seed=5)
    for stride, X, M, hh, ww in ((2, wxyz, wmask, h, w), (1, wxyz_q, wmask_q, hq, wq)):
        sel = B.conv3d_select(X.numpy(), M.numpy(), stride)
        for ti in range(4):
            O.CONV3D_TAP = []
            with torch.no_grad():
                O.conv3d_knn(pp, 'blocks.0.conv3d_1', X[ti], torch.zeros(4, bs, 32, hh, ww), M[ti], stride, 4, target=ti)
            tap, O.CONV3D_TAP = O.CONV3D_TAP[0], None
            dist, valid = B.conv3d_keys(X[ti].numpy(), M[ti].numpy(), stride)
            okey = tap['key'].numpy()
            assert eq(np.where(valid > 0, dist, okey.max()).astype(np.float32), okey)   # keys, bit for bit
            assert eq(sel[ti], tap['idx'].numpy())                                      # torch.topk's ids, in its order


@pytest.mark.parametrize('cfg', [(48, 56, 2, 3, True), (64, 64, 1, 1234, False)])
def test_loss_masks_bit_exact(cfg):
    H, W, bs, seed, rnd = cfg
    st, tb = _inputs(H, W, bs, seed, rnd)
    K = torch.from_numpy(st.K)
    ray = O.make_rays(st.K, H, W)
    g = torch.Generator().manual_seed(5)
    disp = tb['disp0'] + 0.05 * torch.randn(tb['disp0'].shape, generator=g)
    depth = O.disp_to_depth(disp, float(st.K[0, 0]), st.baseline)
    pdepth = O.disp_to_depth(tb['primary_disp'], float(st.K[0, 0]), st.baseline)
    for i, j in ((0, 1), (2, 0), (3, 2)):
        for mf in (True, False):
            args = (K, ray, depth[i], depth[j], tb['R'][i], tb['t'][i], tb['R'][j], tb['t'][j], tb[f'flow_{i}{j}'][0],
                    tb[f'flow_{j}{i}'][0], tb['ambient0'][i], tb['ambient0'][j])
            _, mask = O.flow_consistency_dir(*args, primary_depth1=pdepth[j] if mf else None, clamp=None if mf else 0.1)
            n = [a.numpy() for a in args]
            m, d1 = B.flow_consistency_mask(st.K, n[1], n[2], n[4], n[5], n[6], n[7], n[8], n[9], n[10], n[11],
                                            primary_depth1=pdepth[j].numpy() if mf else None)
            assert np.array_equal(m, mask.numpy())
            _, d1o = O.project(O.unproject(depth[i], ray, tb['R'][i], tb['t'][i]), K, tb['R'][j], tb['t'][j])
            assert np.array_equal(d1.reshape(-1), d1o.numpy().reshape(-1))


@pytest.mark.parametrize('name', ['mf_64_bs1', 'mf_64_bs2_rnd', 'mf_128_bs1', 'mf_128_bumps'])
def test_emulated_selection_is_the_reference_topk(golden_dir, name):
    """end to end from the raw batch: bitexact's neighbour ids == the REFERENCE module's torch.topk output stored in the
    goldens (oracle/make_golden.py records it while the imported reference runs), every id in the same position"""
    G = np.load(os.path.join(golden_dir, name + '.npz'))
    H, W, bs = int(G['H']), int(G['W']), int(G['bs'])
    st, tb = _inputs(H, W, bs, int(G['bseed']), bool(int(G['random_batch'])),
                     **({} if int(G['random_batch']) else dict(scene=str(G['scene']), motion=float(G['motion']))))
    h, w = H // 2, W // 2
    depth = B.disp_to_depth(tb['primary_disp'].numpy(), float(st.K[0, 0]), st.baseline)
    depth_core = B.resize_ac(depth, h, w)
    flow_core = {k: B.resize_flow(v[0].numpy(), h, w) for k, v in tb.items() if k.startswith('flow_')}
    wxyz, wmask = B.mf_geometry(depth_core, O.mf_core_rays(st.K, H, W).numpy(), tb['R'].numpy(), tb['t'].numpy(), flow_core)
    assert np.array_equal(B.conv3d_select(wxyz, wmask, 2), G['knn_idx_core'])
    hq, wq = (h + 2 - 3) // 2 + 1, (w + 2 - 3) // 2 + 1
    wxyz_q = B.resize_ac(wxyz, hq, wq)
    wmask_q = (B.resize_ac(wmask, hq, wq) > 0.5).astype(np.float32)
    assert np.array_equal(B.conv3d_select(wxyz_q, wmask_q, 1), G['knn_idx_quarter'])
