"""Whole DIS-MF training step on the HIP path vs the reference goldens (tests/golden/mf_*.npz, produced by
oracle/make_golden.py from the imported reference) and vs the CPU oracle."""
import os
import argparse
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import dis_oracle as O


def make_args(arch, bs):
    return argparse.Namespace(use_pseudo_gt=False, lcn_radius=5, track_length=4, data_type='synthetic',
                              architecture=arch, epochs=1, warmup_epochs=150, train_batch_size=bs, max_disp=128)


def run_hip_step(G, force_reference_knn=True):
    from depthinspace_amd import synth
    from depthinspace_amd.model import multi_frame_networks, multi_frame_worker
    from depthinspace_amd.trainer import FlatAdam
    H, W, bs = int(G['H']), int(G['W']), int(G['bs'])
    settings = synth.make_settings(H, W)
    mk = synth.make_random_batch if int(G['random_batch']) else synth.make_batch
    batch = mk(settings, bs, 4, seed=int(G['bseed']))
    params = O.init_params(O.mf_param_shapes(), seed=int(G['pseed']))
    net = multi_frame_networks.FuseNet(imsize=(H, W), K=settings.K, baseline=settings.baseline, track_length=4,
                                       max_disp=128)
    net.load_state_dict({k: v.detach() for k, v in params.items()})
    net = net.cuda()
    if force_reference_knn:
        # Conv3D's top-9 is ill-conditioned in the reference itself (a 2e-7 input perturbation moves the
        # reference's own output by `ulp_sens_free`, see oracle/make_golden.py and DESIGN.md); arithmetic parity is
        # therefore pinned with the reference's neighbour sets, and the HIP selection is tested separately below.
        net.knn_index_override = (torch.from_numpy(G['knn_idx_core']).cuda(), torch.from_numpy(G['knn_idx_quarter']).cuda())
    w = multi_frame_worker.Worker(make_args('multi_frame', bs), settings=settings)
    w.build_losses()
    w.current_epoch = int(G['epoch'])
    opt = FlatAdam(net.parameters(), lr=1e-4)
    errs, out = w.train_step(net, opt, {k: torch.from_numpy(v) for k, v in batch.items()})
    torch.cuda.synchronize()
    return net, opt, errs, out


@pytest.mark.parametrize('name', ['mf_64_bs1', 'mf_64_bs2_rnd', 'mf_128_bs1'])
def test_mf_step_matches_reference(golden_dir, name):
    G = np.load(os.path.join(golden_dir, name + '.npz'))
    net, opt, errs, out = run_hip_step(G)
    # (ii) network output: disparity L1 vs reference < 1e-4 (north-star tolerance)
    ref_out = torch.from_numpy(G['out0'])
    l1 = float((out.detach().cpu() - ref_out).abs().mean())
    mx = float((out.detach().cpu() - ref_out).abs().max())
    assert l1 < 1e-4, (l1, mx)
    assert mx < 2e-3, mx
    # (iii) ordered loss terms
    vals = np.array([float(e.detach()) for e in errs])
    assert len(vals) == len(G['vals'])
    np.testing.assert_allclose(vals, G['vals'], rtol=2e-4, atol=2e-6)
    # (iv) gradients of every parameter
    keys = list(G['grad_keys'])
    named = dict(net.named_parameters())
    worst = 0.0
    for i, k in enumerate(keys):
        g = named[k].grad
        if bool(G['grad_none'][i]):
            assert float(g.abs().max()) == 0.0, k
            continue
        scale = float(G['grad_absmax'][i]) + 1e-20
        l2_ref = float(G['grad_l2'][i])
        l2 = float(g.double().norm())
        assert abs(l2 - l2_ref) <= 2e-3 * l2_ref + 1e-12, (k, l2, l2_ref)
        if 'grad:' + k in G.files:
            err = float((g.cpu() - torch.from_numpy(G['grad:' + k])).abs().max()) / scale
            worst = max(worst, err)
            assert err < 5e-3, (k, err)
    # (v) parameters after one Adam step, where stored
    for k in keys:
        if 'new:' + k in G.files:
            new_ref = torch.from_numpy(G['new:' + k])
            # Adam's first step moves every weight by ~lr*sign(g); sign flips of ~0 gradients allow 2*lr
            assert float((named[k].detach().cpu() - new_ref).abs().max()) <= 2.1e-4, k
    print(name, 'disp L1', l1, 'max', mx, 'worst grad rel err', worst)


@pytest.mark.parametrize('name', ['mf_64_bs1', 'mf_64_bs2_rnd', 'mf_128_bs1'])
def test_mf_free_running_knn_selection(golden_dir, name):
    """HIP neighbour selection on its own geometry vs the reference's: identical wherever the reference's top-9
    is well conditioned (relative gap between the 9th and 10th key > 1e-3); the free-running output stays within
    a small multiple of the reference's own sensitivity to a 1-ulp input perturbation."""
    G = np.load(os.path.join(golden_dir, name + '.npz'))
    net, opt, errs, out = run_hip_step(G, force_reference_knn=False)
    for tag, k in (('core', 0), ('quarter', 1)):
        mine = np.sort(net.last_knn_index[k].cpu().numpy(), axis=-1)
        ref = np.sort(G[f'knn_idx_{tag}'], axis=-1)
        same = (mine == ref).all(axis=-1)
        good = G[f'knn_margin_{tag}'] > 1e-3
        assert same[good].all(), (tag, float(same[good].mean()))
        print(name, tag, 'agreement overall', float(same.mean()), 'well-conditioned fraction', float(good.mean()))
    ref_out = torch.from_numpy(G['out0'])
    l1 = float((out.detach().cpu() - ref_out).abs().mean())
    mx = float((out.detach().cpu() - ref_out).abs().max())
    sens_l1, sens_max = [float(v) for v in G['ulp_sens_free']]
    print(name, 'free-running disp L1', l1, 'max', mx, '| reference 1-ulp sensitivity L1', sens_l1, 'max', sens_max)
    assert l1 < 10 * sens_l1 + 1e-4, (l1, sens_l1)
