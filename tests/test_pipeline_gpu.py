"""The rows either side of the hot path on the HIP path, end to end at 64x64:
on-disk .npz schema -> DIS-SF training through Worker.do (retrain, then resume from state.dict) -> presave of the
DIS-SF disparities -> DIS-MF training on them -> presave of the DIS-MF disparities -> DIS-FTSF with pseudo-GT."""
import argparse
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _args(arch, epochs, bs=2, pgt=False):
    return argparse.Namespace(use_pseudo_gt=pgt, lcn_radius=5, track_length=4, data_type='synthetic', architecture=arch,
                              epochs=epochs, warmup_epochs=150, train_batch_size=bs, max_disp=128)


def _oracle_params(net):
    return {k: v.detach().cpu().clone() for k, v in net.state_dict().items()}


def test_sf_mf_ftsf_pipeline(tmp_path):
    """The reference's three-stage flow through the REAL entry-point classes (Worker(data_root=...): its own
    get_train_set / get_test_sets on the on-disk tracks, device-side augmentation on, evaluation metrics, checkpoints)."""
    import json
    from oracle import dis_oracle as O
    from depthinspace_amd import synth
    from depthinspace_amd.data import dataset as D
    from depthinspace_amd.data.presave_disp import presave_disp
    from depthinspace_amd.model import networks, multi_frame_networks, single_frame_worker, multi_frame_worker
    from depthinspace_amd.trainer import FlatAdam
    H = W = 64
    settings = synth.make_settings(H, W)
    root = str(tmp_path / 'data')
    paths = D.write_synthetic_dataset(root, settings, 6, seed=50)
    assert sorted(os.listdir(paths[0])) == ['flow.npz', 'frames.npz']
    out = str(tmp_path / 'out')
    mk = dict(data_root=root, output_dir=out, num_workers=0, test_batch_size=1)

    # ---- DIS-SF: retrain 1 epoch, then resume to epoch 2 (state.dict / net_%04d.params / metrics.json layout)
    w = single_frame_worker.Worker(_args('single_frame', 1), **mk)
    assert w.device_aug and len(w.train_paths) == 4 and len(w.test_paths) == 1 and len(w.valid_paths) == 1
    net = networks.DispDecoder(channels_in=2, max_disp=128, imsizes=w.imsizes).cuda()
    w.do(net, FlatAdam(net.parameters(), lr=1e-4), cmd='retrain')
    exp = os.path.join(out, 'single_frame')
    assert os.path.exists(os.path.join(exp, 'state.dict')) and os.path.exists(os.path.join(exp, 'net_0000.params'))
    sd = torch.load(os.path.join(exp, 'net_0000.params'))
    assert len(sd) == 64 and 'disp_decoder.conv1.0.weight' in sd  # reference state_dict keys
    state = torch.load(os.path.join(exp, 'state.dict'), weights_only=False)
    assert set(state.keys()) == {'epoch', 'min_err', 'state_dict', 'optimizer', 'cpu_rng_state', 'gpu_rng_state'}
    assert set(state['optimizer'].keys()) == {'state', 'param_groups'}          # torch.optim.Adam's layout
    assert float(state['optimizer']['state'][0]['step']) == 2.0                 # 4 train tracks / bs 2
    w2 = single_frame_worker.Worker(_args('single_frame', 2), **mk)
    net2 = networks.DispDecoder(channels_in=2, max_disp=128, imsizes=w2.imsizes).cuda()
    opt2 = FlatAdam(net2.parameters(), lr=1e-4)
    w2.do(net2, opt2, cmd='resume')
    assert opt2.step_count == 4                                                 # the restored counter went on from 2
    assert os.path.exists(os.path.join(exp, 'net_0001.params'))
    m = json.load(open(os.path.join(exp, 'metrics.json')))
    assert set(m.keys()) == {'0', '1'} and len(m['1']['train']['loss']) == 11
    # evaluation metrics of test_epoch (reference co/metric.py via callback_test_*): DistanceMetric + OutlierFractionMetric
    t0 = m['1']['test']['0']
    for k in ('dist2_mean', 'dist2_std', 'dist2_median', 'dist2_q10', 'dist2_q90', 'dist2_min', 'dist2_max', 'of0.1', 'of0.5',
              'of1', 'of2', 'of5', 'loss'):
        assert k in t0, k
    assert 0.0 <= t0['of5'] <= t0['of1'] <= t0['of0.1'] <= 1.0 and t0['dist2_min'] <= t0['dist2_median'] <= t0['dist2_max']
    # retest of a stored epoch reproduces the stored metrics (deterministic evaluation)
    w3 = single_frame_worker.Worker(_args('single_frame', 2), **mk)
    net3 = networks.DispDecoder(channels_in=2, max_disp=128, imsizes=w3.imsizes).cuda()
    w3.do(net3, FlatAdam(net3.parameters(), lr=1e-4), cmd='retest', epoch=1)
    m2 = json.load(open(os.path.join(exp, 'metrics.json')))
    assert m2['1']['test']['0']['dist2_mean'] == pytest.approx(t0['dist2_mean'], rel=1e-6)
    # ---- presave DIS-SF -> primary_disp for DIS-MF; parity of the stored disparities with the oracle's no_grad forward
    assert presave_disp('single_frame', net2, root) == 6
    d = np.load(os.path.join(paths[0], 'single_frame_disp.npz'))['disp']
    assert d.shape == (4, 1, H, W) and np.isfinite(d).all() and d.min() >= 0 and d.max() <= 128
    fr = np.load(os.path.join(paths[0], 'frames.npz'))
    im = torch.from_numpy(fr['im'])
    lcn, _ = O.lcn(im)
    with torch.no_grad():
        ref_sf = O.sf_forward({k: v for k, v in _oracle_params(net2).items()}, torch.cat([lcn, im], 1))[0]
    assert float((torch.from_numpy(d) - ref_sf).abs().mean()) < 1e-4
    # ---- DIS-MF on the presaved disparities
    wm = multi_frame_worker.Worker(_args('multi_frame', 1), **mk)
    netm = multi_frame_networks.FuseNet(imsize=wm.imsizes[0], K=wm.K, baseline=wm.baseline, track_length=4, max_disp=128).cuda()
    wm.do(netm, FlatAdam(netm.parameters(), lr=1e-4), cmd='retrain')
    assert len(torch.load(os.path.join(out, 'multi_frame', 'net_0000.params'))) == 236
    assert presave_disp('multi_frame', netm, root) == 6
    # the stored multi-frame disparities of the LAST track vs the oracle on the neighbour sets the HIP run used
    last = sorted(paths)[-1]
    dm = np.load(os.path.join(last, 'multi_frame_disp.npz'))['disp']
    fr = np.load(os.path.join(last, 'frames.npz'))
    fl = np.load(os.path.join(last, 'flow.npz'))
    prim = torch.from_numpy(np.load(os.path.join(last, 'single_frame_disp.npz'))['disp'])
    im = torch.from_numpy(fr['im'])
    lcn, _ = O.lcn(im)
    st = D.load_settings(root)
    O.CONV3D_FORCE = {'core': netm.last_knn_index[0].cpu().long(), 'quarter': netm.last_knn_index[1].cpu().long()}
    try:
        with torch.no_grad():
            ref_mf = O.mf_forward(_oracle_params(netm), st.K, torch.cat([lcn, im], 1).unsqueeze(1),
                                  torch.from_numpy(fr['ambient']).unsqueeze(1), prim.unsqueeze(1),
                                  O.disp_to_depth(prim.unsqueeze(1), float(st.K[0, 0]), st.baseline),
                                  torch.from_numpy(fr['R']).unsqueeze(1), torch.from_numpy(fr['t']).unsqueeze(1),
                                  {k: torch.from_numpy(fl[k]) for k in fl.files})
    finally:
        O.CONV3D_FORCE = None
    assert float((torch.from_numpy(dm) - ref_mf[:, 0]).abs().mean()) < 1e-4
    # ---- DIS-FTSF: single-frame net with the multi-frame disparities as pseudo ground truth
    wf = single_frame_worker.Worker(_args('single_frame', 3, pgt=True), **mk)
    wf.do(net2, FlatAdam(net2.parameters(), lr=1e-4), cmd='resume')
    m = json.load(open(os.path.join(exp, 'metrics.json')))
    assert len(m['2']['train']['loss']) == 15 and all(np.isfinite(m['2']['train']['loss']))


def test_resume_from_reference_checkpoint(golden_dir):
    """A state.dict written by the REFERENCE (torch.save of {'epoch','min_err','state_dict','optimizer': torch.optim.Adam
    .state_dict(), 'cpu_rng_state'}, produced by oracle/make_golden.py while the imported reference runs) loads into the HIP
    path - network through the reference's state_dict keys, optimiser through torch.optim.Adam's layout - and the NEXT step
    equals the reference's own next step (loss terms, output, parameters): Adam's moments and bias correction carried over."""
    from tests.test_step_gpu import golden_batch, make_args
    from depthinspace_amd.model import multi_frame_networks, multi_frame_worker
    from depthinspace_amd.trainer import FlatAdam
    G = np.load(os.path.join(golden_dir, 'mf_64_bs2_rnd.npz'))
    state = torch.load(os.path.join(golden_dir, 'mf_64_bs2_rnd_ref_state.dict'), weights_only=False)
    H, W, bs = int(G['H']), int(G['W']), int(G['bs'])
    settings, batch = golden_batch(G)
    net = multi_frame_networks.FuseNet(imsize=(H, W), K=settings.K, baseline=settings.baseline, track_length=4, max_disp=128)
    cur = net.state_dict()
    cur.update(state['state_dict'])
    net.load_state_dict(cur)
    net = net.cuda()
    opt = FlatAdam(net.parameters(), lr=1e-4)
    opt.load_state_dict(state['optimizer'])
    assert opt.step_count == 1
    w = multi_frame_worker.Worker(make_args('multi_frame', bs), settings=settings)
    w.build_losses()
    w.current_epoch = int(G['epoch'])
    errs, out = w.train_step(net, opt, {k: torch.from_numpy(v) for k, v in batch.items()})
    torch.cuda.synchronize()
    assert opt.step_count == 2
    assert np.array_equal(net.last_knn_index[0].cpu().numpy(), G['knn_idx_core'])  # (geometry does not depend on weights)
    np.testing.assert_allclose(np.array([float(e.detach()) for e in errs]), G['step2_vals'], rtol=2e-4, atol=2e-6)
    assert float((out.detach().cpu() - torch.from_numpy(G['step2_out0'])).abs().mean()) < 1e-4
    named = dict(net.named_parameters())
    n, checked = 0, 0
    # second moments after the step (exp_avg_sq of the flat buffer, per parameter): where sqrt(v_hat) is well above eps the
    # update lr * m_hat / (sqrt(v_hat) + eps) is determined by the carried-over moments and the new gradient, so the
    # parameters must equal the reference's to fp32 rounding; entries with a ~0 second moment may differ by up to 2 lr
    v_of = {id(p): opt.exp_avg_sq[o:o + p.numel()].view(p.shape) for p, o in zip(opt.params, opt.offsets)}
    for k in G.files:
        if k.startswith('step2_new:') and k[10:] in named:
            p = named[k[10:]]
            d = (p.detach().cpu() - torch.from_numpy(G[k])).abs()
            rv = (v_of[id(p)] / (1.0 - 0.999 ** 2)).sqrt().cpu()
            sure = rv > max(0.05 * float(rv.max()), 1e-6)   # (gradient error <= 2e-4 of the largest entry moves the ratio by < 1e-2)
            checked += int(sure.sum())
            if bool(sure.any()):
                assert float(d[sure].max()) <= 1e-6, (k, float(d[sure].max()))
            assert float(d.max()) <= 2.1e-4, k
            n += 1
    assert n > 100 and checked > 1000, (n, checked)


def test_device_augmentation_distributions():
    """dis_augment (reference data/data_manipulation.py:114-195 with data/dataset.py:67-70): not comparable
    with numpy's generator draw for draw, so the DISTRIBUTIONS are checked, each effect isolated."""
    import torch.nn.functional as F
    from depthinspace_amd import ops
    g = torch.Generator().manual_seed(3)
    n, h, w = 6, 96, 80
    im = (torch.rand(n, 1, h, w, generator=g) * 0.6 + 0.2)
    amb = (torch.rand(n, 1, h, w, generator=g) * 0.5 + 0.25)
    imd, ambd = im.cuda(), amb.cuda()
    seed = torch.tensor([1234567], dtype=torch.int64).cuda()

    def run(params, s=seed):
        return ops.augment(imd, ambd, torch.tensor(params, dtype=torch.float32).cuda(), s)
    # identity: nothing switched on
    a, b = run([[0, 0, 0, 0, 0, -1]] * n)
    assert torch.equal(a, imd) and torch.equal(b, ambd)
    # blur only == cv2.GaussianBlur((5,5), sigma): separable exp(-x^2/2 sigma^2) kernel, BORDER_REFLECT_101
    sig = [0.2, 0.3, 0.35, 0.4, 0.45, 0.5]
    a, b = run([[1, s_, 0.7 - s_, 0, 0, -1] for s_ in sig])
    for i, s_ in enumerate(sig):
        for src, got, sg in ((im, a, s_), (amb, b, 0.7 - s_)):
            x = torch.arange(-2, 3, dtype=torch.float64)
            k = torch.exp(-(x * x) / (2 * sg * sg))
            k = (k / k.sum())
            k2 = (k[:, None] * k[None, :]).float().view(1, 1, 5, 5)
            ref = F.conv2d(F.pad(src[i:i + 1], (2, 2, 2, 2), mode='reflect'), k2)
            assert float((got[i:i + 1].cpu() - ref).abs().max()) < 2e-6
    # noise only: zero mean, std = amplitude / 255, image and ambient independent, different seeds differ, same seed repeats
    a, b = run([[0, 0, 0, 3.0, 1.5, -1]] * n)
    da, db = (a - imd).double(), (b - ambd).double()
    assert abs(float(da.mean())) < 2e-4 and abs(float(da.std()) * 255 - 3.0) < 0.05
    assert abs(float(db.mean())) < 2e-4 and abs(float(db.std()) * 255 - 1.5) < 0.03
    assert abs(float((da * db).mean()) / (float(da.std()) * float(db.std()))) < 0.02
    a2, _ = run([[0, 0, 0, 3.0, 1.5, -1]] * n)
    a3, _ = run([[0, 0, 0, 3.0, 1.5, -1]] * n, torch.tensor([99], dtype=torch.int64).cuda())
    assert torch.equal(a, a2) and not torch.equal(a, a3)
    kur = float(((da - da.mean()) ** 4).mean() / da.var() ** 2)
    assert 2.9 < kur < 3.1  # Gaussian
    # salt and pepper only: image pixels set to the image's own max / min at rate ~ratio each, ambient untouched
    ratio = 0.02
    a, b = run([[0, 0, 0, 0, 0, ratio]] * n)
    assert torch.equal(b, ambd)
    for i in range(n):
        ch = a[i] != imd[i]
        vals = a[i][ch]
        lo, hi = float(imd[i].min()), float(imd[i].max())
        assert bool(((vals == lo) | (vals == hi)).all())
        frac = float(ch.float().mean())
        assert 0.7 * 2 * ratio < frac < 1.3 * 2 * ratio
    # clipping
    a, b = run([[0, 0, 0, 200.0, 200.0, -1]] * n)
    assert float(a.min()) >= 0 and float(a.max()) <= 1 and float(b.min()) >= 0 and float(b.max()) <= 1
    # the host-side draws follow the reference's ranges
    p = ops.draw_augment_params(4000, np.random.RandomState(0))
    assert 0.45 < p[:, 0].mean() < 0.55 and p[p[:, 0] > 0, 1].min() >= 0.2 and p[p[:, 0] > 0, 1].max() <= 0.5
    assert p[:, 3].max() <= 3.0 and p[:, 4].max() <= 3.0 and 0.45 < (p[:, 5] >= 0).mean() < 0.55 and p[:, 5].max() <= 0.0005


def test_sgm_warmup_term():
    """real-data warm-up term (reference model/multi_frame_worker.py:168-173) vs the same expression in torch"""
    from depthinspace_amd import ops
    g = torch.Generator().manual_seed(8)
    o = (torch.rand(4, 2, 1, 40, 36, generator=g) * 60)
    sgm = (torch.rand(4, 2, 1, 40, 36, generator=g) * 60)
    noise = 1.5 * torch.randn(o.shape, generator=g)
    orf = o.clone().requires_grad_(True)
    valid = (sgm > 30).float()
    ref = torch.sum(torch.abs(orf - sgm + noise) * valid) / torch.sum(valid)
    (ref * 0.1).backward()
    od = o.cuda().requires_grad_(True)
    val = ops.sgm_l1(od, sgm.cuda(), noise.cuda(), 30.0)
    (val * 0.1).backward()
    assert abs(float(val) - float(ref)) < 1e-5 * float(ref)
    assert float((od.grad.cpu() - orf.grad).abs().max()) < 1e-9


@pytest.mark.parametrize('arch', ['single_frame', 'multi_frame'])
def test_worker_train_epoch_graph_equals_eager(tmp_path, arch):
    """Worker(use_graph=True) - the captured step behind DIS_TRAIN_GRAPH=1 - through the real loop (`do('retrain')`: loader, epoch
    statistics, checkpoint) lands on the same parameters and epoch losses as the eager loop on the same on-disk tracks."""
    import json
    from depthinspace_amd import synth
    from depthinspace_amd.data import dataset as D
    from depthinspace_amd.model import networks, multi_frame_networks, single_frame_worker, multi_frame_worker
    from depthinspace_amd.trainer import FlatAdam
    H = W = 64
    settings = synth.make_settings(H, W)
    root = str(tmp_path / 'data')
    D.write_synthetic_dataset(root, settings, 8, seed=60)
    if arch == 'multi_frame':   # DIS-MF reads the DIS-SF disparities of the tracks: any stored disparity will do here
        import numpy as np
        for p in sorted(os.listdir(root)):
            d = os.path.join(root, p)
            if os.path.isdir(d) and os.path.exists(os.path.join(d, 'frames.npz')):
                f = np.load(os.path.join(d, 'frames.npz'))
                np.savez(os.path.join(d, 'single_frame_disp.npz'), disp=f['disp'])
    res = {}
    for mode in ('eager', 'graph'):
        out = str(tmp_path / ('out_' + mode))
        mod = multi_frame_worker if arch == 'multi_frame' else single_frame_worker
        w = mod.Worker(_args(arch, 1), data_root=root, output_dir=out, num_workers=0, test_batch_size=1, use_graph=(mode == 'graph'))
        w.device_aug = False          # (the augmentation draws are host-RNG driven: off, so that both runs see the same batches)
        torch.manual_seed(5)
        if arch == 'multi_frame':
            net = multi_frame_networks.FuseNet(imsize=w.imsizes[0], K=w.K, baseline=w.baseline, track_length=4, max_disp=128).cuda()
        else:
            net = networks.DispDecoder(channels_in=2, max_disp=128, imsizes=w.imsizes).cuda()
        opt = FlatAdam(net.parameters(), lr=1e-4)
        w.do(net, opt, cmd='retrain')
        m = json.load(open(os.path.join(out, arch, 'metrics.json')))
        res[mode] = (opt.flat_p.clone(), m['0']['train']['loss'], opt.step_count)
    assert res['eager'][2] == res['graph'][2] > 0
    # (Adam turns the rounding noise of ~0 gradients into +-lr steps: a few elements may differ by 2 lr per step)
    d = (res['eager'][0] - res['graph'][0]).abs()
    assert float(d.max()) <= 2.1e-4 * res['eager'][2] and float(d.mean()) < 2e-6
    import numpy as np
    np.testing.assert_allclose(res['graph'][1], res['eager'][1], rtol=5e-3, atol=1e-5)
