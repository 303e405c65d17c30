"""Generate the committed golden vectors under tests/golden/ by running the REFERENCE itself on CPU.

Only runs in the build container (needs /root/reference).  The reference's Python is imported with
the import-time shims of SURVEY.md Appendix D (fake cv2/h5py/ext_cpu modules, .cuda() -> no-op); its
sources are never copied: what is committed is data (inputs by seed + recipe, outputs by value).

    python oracle/make_golden.py            # writes tests/golden/*.npz and prints oracle-vs-reference diffs

Every fixture stores the seeds needed to regenerate inputs and parameters with
`depthinspace_amd.synth` and `oracle.dis_oracle.init_params` (CPU generators, deterministic), plus the
reference's outputs.  The oracle restatement is checked against the same outputs here and again in
tests/test_oracle_golden.py.
"""
import os
import sys
import types
import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
GOLD = os.path.join(ROOT, 'tests', 'golden')


def import_reference():
    sys.path.insert(0, '/root/reference')
    torch.cuda.synchronize = lambda *a, **k: None
    torch.cuda.empty_cache = lambda *a, **k: None
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.nn.Module.cuda = lambda self, *a, **k: self
    for cls in (torch.Tensor, torch.nn.Module):
        orig = cls.to
        def make(o):
            def to(s, *a, **k):
                a = ['cpu' if isinstance(x, str) and 'cuda' in x else x for x in a]
                return o(s, *a, **k)
            return to
        setattr(cls, 'to', make(orig))
    for name in ('ext_cpu', 'ext_cuda', 'h5py'):
        sys.modules[name] = types.ModuleType(name)
    cv2 = types.ModuleType('cv2')
    cv2.INTER_NEAREST, cv2.INTER_LINEAR = 0, 1

    def _resize(a, dsize, interpolation=1):
        w, h = dsize
        H, W = a.shape[:2]
        if (h, w) == (H, W):
            return a.copy()
        assert interpolation == 0
        return a[np.floor(np.arange(h) * H / h).astype(int)][:, np.floor(np.arange(w) * W / w).astype(int)]
    cv2.resize = _resize
    sys.modules['cv2'] = cv2
    import matplotlib
    matplotlib.use('Agg')
    from model import ext_functions
    T = {0: 'mse', 1: 'sad', 2: 'census_mse', 3: 'census_sad'}
    sys.modules['ext_cpu'].photometric_loss_forward = \
        lambda es, ta, b, t, e: ext_functions.photometric_loss_pytorch(es, ta, b, T[t], e)

    def _bwd(es, ta, g, b, t, e):
        with torch.enable_grad():
            x = es.detach().clone().requires_grad_(True)
            y = ext_functions.photometric_loss_pytorch(x, ta.detach(), b, T[t], e)
            return torch.autograd.grad(y, x, g)[0]
    sys.modules['ext_cpu'].photometric_loss_backward = _bwd
    from model import networks, multi_frame_networks, single_frame_worker, multi_frame_worker
    return dict(ext=ext_functions, networks=networks, mfn=multi_frame_networks, sfw=single_frame_worker,
                mfw=multi_frame_worker)


def ref_worker(ref, arch, settings, epoch=0, use_pseudo_gt=False):
    """A reference Worker without the filesystem."""
    networks = ref['networks']
    mod = ref['mfw'] if arch == 'multi_frame' else ref['sfw']
    w = object.__new__(mod.Worker)
    w.track_length = 4
    w.current_epoch = epoch
    w.warmup_epochs = 150
    w.data_type = 'synthetic'
    w.use_pseudo_gt = use_pseudo_gt
    w.lcn_in = networks.LCN(5, 0.05)
    w.ref_pattern = settings.pattern
    w.disparity_loss = networks.DisparitySmoothLoss()
    H, W = settings.imsize
    pat = settings.pattern.mean(axis=2)
    pat = torch.from_numpy(pat[None][None].astype(np.float32))
    pat, _ = w.lcn_in(pat)
    w.patterns = [pat]
    pat3 = torch.cat([pat for _ in range(3)], dim=1)
    w.ph_losses = [networks.RectifiedPatternSimilarityLoss(H, W, pattern=pat3)]
    K = settings.K
    Ki = np.linalg.inv(K)
    cls = networks.Multi_Frame_Flow_Consistency_Loss if arch == 'multi_frame' else networks.Single_Frame_Flow_Consistency_Loss
    w.ge_losses = [cls(torch.from_numpy(K), torch.from_numpy(Ki), H, W, clamp=0.1)]
    w.d2ds = [networks.DispToDepth(float(K[0, 0]), float(settings.baseline))]
    return w


def to_torch_batch(batch):
    return {k: torch.from_numpy(v.copy()) for k, v in batch.items()}


def maxdiff(a, b):
    return float((a.detach() - b.detach()).abs().max())


def make_sgm_disp(batch, seed):
    """a stand-in for the SGM disparities of the `real` dataset: the primary disparity plus noise, shifted so that the
    values straddle the validity threshold (30) - the mask of the warm-up term is neither empty nor full at fixture size"""
    g = np.random.RandomState(seed)
    d = batch['primary_disp'].astype(np.float32)
    sgm = d - np.float32(np.median(d)) + np.float32(31.0) + g.normal(0, 2.0, d.shape).astype(np.float32)
    frac = float((sgm > 30).mean())
    assert 0.2 < frac < 0.9, frac
    return np.ascontiguousarray(sgm.astype(np.float32))


def run_step_case(ref, arch, size, bs, pseed, bseed, epoch=0, use_pseudo_gt=False, random_batch=False, full_grads=False,
                  pattern='default', scene='plane', motion=1.0, save_ckpt=False, real_sgm=False):
    from depthinspace_amd import synth
    from oracle import dis_oracle as O
    H, W = size
    settings = synth.make_settings(H, W, pattern=pattern)
    if random_batch:
        assert scene == 'plane' and motion == 1.0
        batch = synth.make_random_batch(settings, bs, 4, seed=bseed, with_pseudo_gt=use_pseudo_gt)
    else:
        batch = synth.make_batch(settings, bs, 4, seed=bseed, with_pseudo_gt=use_pseudo_gt, scene=scene, motion=motion)
    shapes = O.mf_param_shapes() if arch == 'multi_frame' else O.sf_param_shapes()
    params = O.init_params(shapes, seed=pseed)
    if real_sgm:
        batch['sgm_disp'] = make_sgm_disp(batch, bseed + 1)

    # ---- reference
    if arch == 'multi_frame':
        net = ref['mfn'].FuseNet(imsize=(H, W), K=settings.K, baseline=settings.baseline, track_length=4, max_disp=128)
    else:
        imsizes = [(H, W)]
        for _ in range(3):
            imsizes.append((imsizes[-1][0] // 2, imsizes[-1][1] // 2))
        net = ref['networks'].DispDecoder(channels_in=2, max_disp=128, imsizes=imsizes)
    sd = net.state_dict()
    assert sorted(sd.keys()) == sorted(shapes.keys()), set(sd.keys()) ^ set(shapes.keys())
    for k in sd:
        assert tuple(sd[k].shape) == tuple(shapes[k]), (k, sd[k].shape, shapes[k])
    net.load_state_dict({k: v.detach().clone() for k, v in params.items()})
    net.train()
    w = ref_worker(ref, arch, settings, epoch, use_pseudo_gt)
    if real_sgm:
        w.data_type = 'real'   # `real` data, epoch < warmup_epochs: the SGM warm-up term (multi_frame_worker.py:168-173)
    opt = torch.optim.Adam(net.parameters(), lr=1e-4)
    w.copy_data(to_torch_batch(batch), 'cpu', False, True)
    opt.zero_grad()
    flow = w.read_optical_flow(True)
    # record what the reference's own torch.topk calls return (Conv3D, multi_frame_networks.py:498): 32 calls per
    # forward = 4 blocks x (conv3d_1, conv3d_2) x 4 target frames, each (bs*ho*wo, 9, 1)
    ref_topk = []
    _topk = torch.topk

    def _rec_topk(*a, **k):
        r = _topk(*a, **k)
        ref_topk.append(r[1].detach().clone())
        return r
    torch.topk = _rec_topk
    try:
        out = w.net_forward(net, flow)
    finally:
        torch.topk = _topk
    # the SGM term draws its noise inside the loss expression (`1.5 * torch.randn(o.size()).cuda()`): the draws of the
    # reference's own run are recorded (a hook, as for torch.topk) and become inputs of the fixture
    sgm_draws = []
    _randn = torch.randn

    def _rec_randn(*a, **k):
        r = _randn(*a, **k)
        sgm_draws.append(r.detach().clone())
        return r
    torch.randn = _rec_randn
    # the reference's draw comes from torch's GLOBAL generator (no seed anywhere in its loss): seed it here, per case, so that a
    # regeneration reproduces the committed fixture bit for bit (the recorded draws travel as inputs either way)
    torch.manual_seed(1000 + bseed)
    try:
        vals = w.loss_forward(out, True, flow)
    finally:
        torch.randn = _randn
    assert len(sgm_draws) == ((1 if arch == 'multi_frame' else 4) if real_sgm else 0), len(sgm_draws)
    sum(vals).backward()
    ref_grads = {k: (p.grad.detach().clone() if p.grad is not None else None) for k, p in net.named_parameters()}
    opt.step()
    ref_new = {k: v.detach().clone() for k, v in net.state_dict().items()}
    ckpt = None
    if save_ckpt:
        # what the reference's Worker.train writes after an epoch (model/worker.py:376-402), here after this one step ...
        ckpt = {'epoch': epoch, 'min_err': {'simple': 1e9}, 'state_dict': {k: v.clone() for k, v in net.state_dict().items()},
                'optimizer': opt.state_dict(), 'cpu_rng_state': torch.get_rng_state()}
        import copy
        ckpt = copy.deepcopy(ckpt)
        # ... and what the reference computes in its NEXT step from that state (same batch): pins the resumed optimiser
        w.copy_data(to_torch_batch(batch), 'cpu', False, True)
        opt.zero_grad()
        flow2 = w.read_optical_flow(True)
        out2 = w.net_forward(net, flow2)
        vals2 = w.loss_forward(out2, True, flow2)
        sum(vals2).backward()
        opt.step()
        step2 = {'vals': np.array([float(v) for v in vals2], dtype=np.float64),
                 'out0': (out2[0] if isinstance(out2, (list, tuple)) else out2).detach().numpy().copy(),
                 'new': {k: v.detach().clone().numpy() for k, v in net.state_dict().items() if v.numel() <= 4096}}
    ref_data = {k: v.detach().clone() for k, v in w.data.items() if k in ('im0', 'std0')}
    outs = out if isinstance(out, (list, tuple)) else [out]

    # ---- oracle
    ctx = O.StepContext(settings)
    st = {'step': 0, 'm': {}, 'v': {}}
    O.CONV3D_TAP = [] if arch == 'multi_frame' else None
    obatch = to_torch_batch(batch)
    for k, dr in enumerate(sgm_draws):   # loader layout (bs, tl, ...), scaled as the expression scales it
        obatch[f'_sgm_noise{k}'] = (1.5 * dr).transpose(0, 1).contiguous()
    res = O.train_step(ctx, arch, params, obatch, adam_state=st, epoch=epoch, use_pseudo_gt=use_pseudo_gt,
                       data_type='real' if real_sgm else 'synthetic')
    tap, O.CONV3D_TAP = O.CONV3D_TAP, None
    o_outs = res['out'] if isinstance(res['out'], (list, tuple)) else [res['out']]
    rep = {'out': max(maxdiff(a, b) for a, b in zip(outs, o_outs)),
           'vals': max(abs(float(a) - float(b)) for a, b in zip(vals, res['vals'])),
           'im0': maxdiff(ref_data['im0'], res['data']['im0']), 'std0': maxdiff(ref_data['std0'], res['data']['std0'])}
    assert len(vals) == len(res['vals'])
    gd, gn = 0.0, 0.0
    for k, g in ref_grads.items():
        og = res['grads'][k]
        if g is None:
            assert og is None or float(og.abs().max()) == 0.0, k
            continue
        gd = max(gd, maxdiff(g, og) / (float(g.abs().max()) + 1e-12))
    rep['grad_rel'] = gd
    rep['adam'] = max(maxdiff(ref_new[k], params[k]) for k in params)
    print(f'[{arch} {H}x{W} bs={bs} epoch={epoch} pgt={use_pseudo_gt} rnd={random_batch}] oracle-vs-reference:', rep)

    fx = {'arch': arch, 'H': H, 'W': W, 'bs': bs, 'pseed': pseed, 'bseed': bseed, 'epoch': epoch,
          'use_pseudo_gt': int(use_pseudo_gt), 'random_batch': int(random_batch), 'pattern': pattern, 'scene': scene,
          'motion': float(motion), 'torch_threads': torch.get_num_threads(),
          'vals': np.array([float(v) for v in vals], dtype=np.float64)}
    if real_sgm:
        fx['real_sgm'] = 1
        fx['sgm_disp'] = batch['sgm_disp']
        for k, dr in enumerate(sgm_draws):
            fx[f'sgm_noise{k}'] = obatch[f'_sgm_noise{k}'].numpy()   # = 1.5 * the reference's k-th torch.randn draw
    for i, o in enumerate(outs):
        fx[f'out{i}'] = o.detach().numpy()
    fx['std0_sum'] = np.float64(ref_data['std0'].double().sum())
    fx['im0_lcn_sample'] = ref_data['im0'][:, :, 0, ::7, ::5].numpy()
    if arch == 'multi_frame':
        # Conv3D neighbour sets (identical in all 4 blocks: they depend on the geometry only) + top-k margins
        for lname, tag in (('conv3d_1', 'core'), ('conv3d_2', 'quarter')):
            per_block = []
            for b in range(4):
                calls = [c for c in tap if c['name'] == f'blocks.{b}.{lname}']
                assert [c['target'] for c in calls] == [0, 1, 2, 3]
                per_block.append(torch.stack([c['idx'] for c in calls], 0))  # (tl,bs,ho,wo,9)
            for b in range(1, 4):
                assert bool((torch.sort(per_block[b], -1)[0] == torch.sort(per_block[0], -1)[0]).all())
            fx[f'knn_idx_{tag}'] = per_block[0].numpy().astype(np.uint8)
            # ... and they ARE the reference module's own torch.topk output, element for element and in its order
            # (the checkpointed reference forward is not re-run here: 32 recorded calls, block-major, layer, target)
            assert len(ref_topk) == 32, len(ref_topk)
            li = 0 if lname == 'conv3d_1' else 1
            for b in range(4):
                for ti in range(4):
                    r = ref_topk[b * 8 + li * 4 + ti]
                    assert bool((r.view(per_block[b][ti].shape) == per_block[b][ti]).all()), (lname, b, ti)
            keysrt = torch.sort(torch.stack([c['key'] for c in tap if c['name'] == f'blocks.0.{lname}'], 0), -1)[0]
            k9, k10 = keysrt[..., 8].double(), keysrt[..., 9].double()
            fx[f'knn_margin_{tag}'] = ((k10 - k9) / torch.clamp(k10, min=1e-30)).float().numpy()
        # evidence: the reference's own sensitivity to a 2e-7 relative perturbation of one input
        # (top-k near-ties amplify rounding noise; see DESIGN.md "top-k conditioning")
        # (run with the post-Adam parameters the oracle holds at this point; the baseline is recomputed with them)
        def knn_sets(tp):
            out = {}
            for lname, tag in (('conv3d_1', 'core'), ('conv3d_2', 'quarter')):
                out[tag] = torch.stack([c['idx'] for c in tp if c['name'] == f'blocks.0.{lname}'], 0)
            return out
        b2 = to_torch_batch(batch)
        b2['primary_disp'] = b2['primary_disp'] * np.float32(1.0 + 2e-7)
        with torch.no_grad():
            d1 = O.copy_data(ctx, to_torch_batch(batch))
            O.CONV3D_TAP = []
            o1 = O.mf_net_forward(ctx, params, d1, O.read_optical_flow(d1, 4))
            base_sets, O.CONV3D_TAP = knn_sets(O.CONV3D_TAP), None
            d2 = O.copy_data(ctx, b2)
            o2 = O.mf_net_forward(ctx, params, d2, O.read_optical_flow(d2, 4))
            # same perturbation with the neighbour sets of the unperturbed run forced
            O.CONV3D_FORCE = base_sets
            o3 = O.mf_net_forward(ctx, params, d2, O.read_optical_flow(d2, 4))
            O.CONV3D_FORCE = None
        outs0 = [o1]
        fx['ulp_sens_free'] = np.array([float((o2 - outs0[0]).abs().mean()), float((o2 - outs0[0]).abs().max())])
        fx['ulp_sens_forced'] = np.array([float((o3 - outs0[0]).abs().mean()), float((o3 - outs0[0]).abs().max())])
        print('   reference sensitivity to 2e-7 input perturbation: free top-k L1/max', fx['ulp_sens_free'],
              ' forced neighbour sets L1/max', fx['ulp_sens_forced'],
              ' margin<1e-3 fraction core', float((fx['knn_margin_core'] < 1e-3).mean()),
              'quarter', float((fx['knn_margin_quarter'] < 1e-3).mean()))
    keys = sorted(ref_grads.keys())
    fx['grad_keys'] = np.array(keys)
    fx['grad_absmax'] = np.array([0.0 if ref_grads[k] is None else float(ref_grads[k].abs().max()) for k in keys])
    fx['grad_sum'] = np.array([0.0 if ref_grads[k] is None else float(ref_grads[k].double().sum()) for k in keys])
    fx['grad_l2'] = np.array([0.0 if ref_grads[k] is None else float(ref_grads[k].double().norm()) for k in keys])
    fx['grad_none'] = np.array([ref_grads[k] is None for k in keys])
    for k in keys:
        g = ref_grads[k]
        if g is None:
            continue
        if full_grads or g.numel() <= 4096:
            fx['grad:' + k] = g.numpy()
            fx['new:' + k] = ref_new[k].numpy()
    if ckpt is not None:
        fx['step2_vals'] = step2['vals']
        fx['step2_out0'] = step2['out0']
        for k, v in step2['new'].items():
            fx['step2_new:' + k] = v
        fx['_ckpt'] = ckpt
    return fx, rep


def op_goldens(ref):
    """Operator-level vectors from the reference modules (small, random inputs)."""
    from oracle import dis_oracle as O
    networks, mfn, ext = ref['networks'], ref['mfn'], ref['ext']
    g = torch.Generator().manual_seed(77)
    fx = {}
    rep = {}
    # LCN (a3)
    x = torch.rand(3, 1, 40, 36, generator=g)
    l, s = networks.LCN(5, 0.05)(x)
    fx['lcn_x'], fx['lcn_out'], fx['lcn_std'] = x.numpy(), l.numpy(), s.numpy()
    ol, os_ = O.lcn(x)
    rep['lcn'] = max(maxdiff(l, ol), maxdiff(s, os_))
    # photometric fwd/bwd, all four types (a17)
    es = torch.randn(2, 1, 24, 28, generator=g)
    ta = torch.randn(2, 1, 24, 28, generator=g)
    go = torch.rand(2, 1, 24, 28, generator=g)
    fx['ph_es'], fx['ph_ta'], fx['ph_go'] = es.numpy(), ta.numpy(), go.numpy()
    for name, code in O.PHOTO_TYPES.items():
        for blk, eps in ((9, 0.5), (5, 0.1)):
            e = es.clone().requires_grad_(True)
            y = ext.photometric_loss(e, ta, blk, name, eps)
            y.backward(go)
            fx[f'ph_{name}_{blk}_out'] = y.detach().numpy()
            fx[f'ph_{name}_{blk}_grad'] = e.grad.numpy()
            e2 = es.clone().requires_grad_(True)
            y2 = O.photometric(e2, ta, blk, name, eps)
            y2.backward(go)
            rep[f'ph_{name}_{blk}'] = max(maxdiff(y, y2), maxdiff(e.grad, e2.grad))
    # pattern loss (a16)
    H, W = 32, 40
    pat = torch.randn(1, 1, H, W, generator=g)
    disp = (torch.rand(2, 1, H, W, generator=g) * 12 - 2).requires_grad_(True)
    im = torch.randn(2, 1, H, W, generator=g)
    std = torch.rand(2, 1, H, W, generator=g) + 0.05
    mod = networks.RectifiedPatternSimilarityLoss(H, W, pattern=torch.cat([pat] * 3, 1))
    val, proj = mod(disp, im, std)
    val.backward()
    fx.update(pl_pat=pat.numpy(), pl_disp=disp.detach().numpy(), pl_im=im.numpy(), pl_std=std.numpy(),
              pl_val=np.float64(val), pl_proj=proj.detach().numpy(), pl_grad=disp.grad.numpy())
    d2 = disp.detach().clone().requires_grad_(True)
    v2, p2 = O.pattern_loss(torch.cat([pat] * 3, 1).mean(dim=1, keepdim=True), d2, im, std)
    v2.backward()
    rep['pattern_loss'] = max(abs(float(val) - float(v2)), maxdiff(proj, p2), maxdiff(disp.grad, d2.grad))
    # smoothness (a18)
    disp = (torch.rand(2, 1, H, W, generator=g) * 10).requires_grad_(True)
    amb = torch.rand(2, 1, H, W, generator=g) * 0.02
    val = networks.DisparitySmoothLoss()(disp, amb)
    val.backward()
    fx.update(sm_disp=disp.detach().numpy(), sm_amb=amb.numpy(), sm_val=np.float64(val), sm_grad=disp.grad.numpy())
    d2 = disp.detach().clone().requires_grad_(True)
    v2 = O.smooth_loss(d2, amb)
    v2.backward()
    rep['smooth'] = max(abs(float(val) - float(v2)), maxdiff(disp.grad, d2.grad))
    # disp->depth (a9)
    d = torch.randn(2, 1, 8, 8, generator=g) * 3
    fx['d2d_in'] = d.numpy()
    fx['d2d_out'] = networks.DispToDepth(435.2, 0.025)(d).numpy()
    rep['d2d'] = maxdiff(torch.from_numpy(fx['d2d_out']), O.disp_to_depth(d, 435.2, 0.025))
    # warp zeros (a11)
    x = torch.randn(2, 5, 20, 24, generator=g)
    fl = torch.randn(2, 2, 20, 24, generator=g) * 4
    fx['warp_x'], fx['warp_flow'] = x.numpy(), fl.numpy()
    fx['warp_out'] = mfn.warp(x, fl).numpy()
    rep['warp'] = maxdiff(torch.from_numpy(fx['warp_out']), O.warp(x, fl))
    # resize helpers
    y = mfn.resize_like(x, torch.zeros(1, 1, 10, 12))
    fx['resize_out'] = y.numpy()
    rep['resize'] = maxdiff(y, O.resize_ac(x, (10, 12)))
    fr = mfn.resize_flow_like({'a': fl}, torch.zeros(1, 1, 10, 12))['a']
    fx['resize_flow_out'] = fr.numpy()
    rep['resize_flow'] = maxdiff(fr, O.resize_flow({'a': fl}, (10, 12))['a'])
    # Conv3D (a14), stride 1 and 2
    C, tl, bs, h, w = 32, 4, 2, 12, 14
    p = O.init_params({k: v for k, v in O.mf_param_shapes().items() if k.startswith('blocks.0.conv3d_1')}, seed=5)
    xyz = torch.randn(tl, bs, 3, h, w, generator=g) * 0.05
    xyz[:, :, 2] += 3.0
    uu, vv = O.pixel_grid(h, w)
    xyz[:, :, 0] += (uu - w / 2) * 0.01
    xyz[:, :, 1] += (vv - h / 2) * 0.01
    feat = torch.randn(tl, bs, C, h, w, generator=g)
    mask = (torch.rand(tl, bs, 1, h, w, generator=g) > 0.2).float()
    mask[0] = 1
    fx.update(c3_xyz=xyz.numpy(), c3_feat=feat.numpy(), c3_mask=mask.numpy())
    for stride in (1, 2):
        m = mfn.Conv3D(channels_in=C, channels_out=C, tl=tl, stride=stride)
        m.load_state_dict({k[len('blocks.0.conv3d_1.'):]: v.detach().clone() for k, v in p.items()})
        f = feat.clone().requires_grad_(True)
        y = m(xyz, f, mask)
        go = torch.randn(y.shape, generator=g)
        y.backward(go)
        fx[f'c3_s{stride}_out'] = y.detach().numpy()
        fx[f'c3_s{stride}_go'] = go.numpy()
        fx[f'c3_s{stride}_gfeat'] = f.grad.numpy()
        for k_, v_ in m.named_parameters():
            fx[f'c3_s{stride}_g:{k_}'] = v_.grad.numpy()
        for v_ in p.values():
            v_.grad = None
        f2 = feat.clone().requires_grad_(True)
        y2, idx, key = O.conv3d_knn(p, 'blocks.0.conv3d_1', xyz, f2, mask, stride, tl, return_index=True)
        y2.backward(go)
        fx[f'c3_s{stride}_idx_sorted'] = np.sort(idx.numpy(), axis=-1).astype(np.int16)
        srt = np.sort(key.numpy(), axis=-1)
        fx[f'c3_s{stride}_margin_min'] = np.float64((srt[..., 9] - srt[..., 8]).min())
        rep[f'conv3d_s{stride}'] = max(maxdiff(y, y2), maxdiff(f.grad, f2.grad),
                                       max(maxdiff(v_.grad, p['blocks.0.conv3d_1.' + k_].grad) for k_, v_ in m.named_parameters()))
    # geometric losses (a19-a21) on a small physical scene + perturbation
    from depthinspace_amd import synth
    st = synth.make_settings(48, 56)
    b = synth.make_random_batch(st, 2, 4, seed=3)
    K = torch.from_numpy(st.K); Ki = torch.from_numpy(np.linalg.inv(st.K))
    tb = {k: torch.from_numpy(v).transpose(0, 1) if v.ndim > 2 else torch.from_numpy(v) for k, v in b.items()}
    d2d = networks.DispToDepth(float(st.K[0, 0]), st.baseline)
    disp_pert = tb['disp0'] + 0.05 * torch.randn(tb['disp0'].shape, generator=g)
    ray = O.make_rays(st.K, 48, 56)
    for nm, cls in (('mf', networks.Multi_Frame_Flow_Consistency_Loss), ('sf', networks.Single_Frame_Flow_Consistency_Loss)):
        mod = cls(K, Ki, 48, 56, clamp=0.1)
        dd = disp_pert.clone().requires_grad_(True)
        depth = d2d(dd)
        pdepth = d2d(tb['primary_disp'])
        i, j = 0, 2
        args = (depth[i], depth[j], tb['R'][i], tb['t'][i], tb['R'][j], tb['t'][j], tb['flow_02'][0], tb['flow_20'][0],
                tb['ambient0'][i], tb['ambient0'][j])
        if nm == 'mf':
            val = mod(*args, pdepth[i], pdepth[j])
        else:
            val = mod(*args)[0]
        val.backward()
        fx[f'ge_{nm}_val'] = np.float64(val)
        fx[f'ge_{nm}_grad'] = dd.grad.numpy()
        d3 = disp_pert.clone().requires_grad_(True)
        dep = O.disp_to_depth(d3, float(st.K[0, 0]), st.baseline)
        pdep = O.disp_to_depth(tb['primary_disp'], float(st.K[0, 0]), st.baseline)
        a2 = (K, ray, dep[i], dep[j], tb['R'][i], tb['t'][i], tb['R'][j], tb['t'][j], tb['flow_02'][0], tb['flow_20'][0],
              tb['ambient0'][i], tb['ambient0'][j])
        v2 = O.mf_flow_consistency(*a2, pdep[i], pdep[j]) if nm == 'mf' else O.sf_flow_consistency(*a2)[0]
        v2.backward()
        rep[f'ge_{nm}'] = max(abs(float(val) - float(v2)), maxdiff(dd.grad, d3.grad))
    fx['ge_disp'] = disp_pert.numpy()
    fx['ge_seed'] = 3
    # evaluation metrics of test_epoch / retest (reference co/metric.py:104-154, as the workers construct them,
    # model/multi_frame_worker.py:236-240): three add() calls of unequal length, values in disparity range
    import co.metric as RM
    met = RM.MultipleMetric(RM.DistanceMetric(vec_length=1),
                            RM.OutlierFractionMetric(vec_length=1, thresholds=[0.1, 0.5, 1, 2, 5]))
    for k in range(3):
        es = (torch.rand(1500 + 37 * k, 1, generator=g) * 8).numpy()
        gt = (es + (torch.randn(es.shape, generator=g) * (0.05 + 0.6 * k)).numpy()).astype(np.float32)
        met.add(es, gt)
        fx[f'met_es{k}'], fx[f'met_gt{k}'] = es, gt
    vals = met.get()
    fx['met_keys'] = np.array(list(vals.keys()))
    fx['met_vals'] = np.array([vals[k] for k in vals], dtype=np.float64)
    print('[ops] oracle-vs-reference:', rep)
    return fx, rep


CASES = [
    ('mf_64_bs1', dict(arch='multi_frame', size=(64, 64), bs=1, pseed=11, bseed=1234, epoch=0, full_grads=True)),
    ('mf_64_bs2_rnd', dict(arch='multi_frame', size=(64, 64), bs=2, pseed=12, bseed=99, epoch=2, random_batch=True,
                           save_ckpt=True)),
    ('mf_128_bs1', dict(arch='multi_frame', size=(128, 128), bs=1, pseed=13, bseed=1234, epoch=2)),
    ('sf_64_bs1', dict(arch='single_frame', size=(64, 64), bs=1, pseed=21, bseed=1234)),
    ('sf_128_bs1_pgt', dict(arch='single_frame', size=(128, 128), bs=1, pseed=22, bseed=1234, use_pseudo_gt=True)),
    # crop_like trims here (reference model/networks.py:242-263): W 108 -> 54 -> 27 -> 14 -> 7 -> 4 -> 2 -> 1, so upconv
    # outputs 28 -> 27 and 8 -> 7 (and 2 -> 1 at the deepest level) are cut exactly as at 512x432
    ('sf_128x108_bs1', dict(arch='single_frame', size=(128, 108), bs=1, pseed=23, bseed=77)),
    # non-degenerate scene: non-planar surface, 1.5x camera motion (few exactly tied top-k keys)
    ('mf_128_bumps', dict(arch='multi_frame', size=(128, 128), bs=1, pseed=14, bseed=4321, epoch=2, scene='bumps',
                          motion=1.5)),
    # BASELINE config 5: DIS-FTSF (pseudo-GT) on the real pattern, K_processed, baseline 0.0246
    ('sf_128_real_pgt', dict(arch='single_frame', size=(128, 128), bs=1, pseed=24, bseed=555, use_pseudo_gt=True,
                             pattern='real')),
    # `real` data during the warm-up epochs: the masked-L1 SGM term with the reference's own noise draws recorded
    # (model/multi_frame_worker.py:168-173: one draw; model/single_frame_worker.py:158-163: one per output scale)
    ('mf_64_real_sgm', dict(arch='multi_frame', size=(64, 64), bs=1, pseed=15, bseed=808, epoch=2, pattern='real',
                            real_sgm=True)),
    ('sf_64_real_sgm', dict(arch='single_frame', size=(64, 64), bs=1, pseed=25, bseed=809, epoch=2, pattern='real',
                            real_sgm=True)),
    # the metric's own size and bench.py's own sample (cpu_baseline / hip_first_step: batch seed 1234, init_params(seed=0),
    # epoch 2, bs=1): the reference's disparity, neighbour ids and loss terms at 512x432 (reference
    # model/multi_frame_worker.py:87-175); ~1 min per reference step on 8 threads, ~5 MB
    ('mf_512x432_bs1', dict(arch='multi_frame', size=(512, 432), bs=1, pseed=0, bseed=1234, epoch=2)),
]


def main():
    global GOLD
    only = sys.argv[1:]
    if '--out' in only:   # write somewhere else than tests/golden (tests/test_oracle_golden.py regenerates cases into a tmp dir)
        i = only.index('--out')
        GOLD = only[i + 1]
        only = only[:i] + only[i + 2:]
    os.makedirs(GOLD, exist_ok=True)
    torch.manual_seed(0)
    torch.set_num_threads(8)
    ref = import_reference()
    if not only or 'ops' in only:
        fx, rep = op_goldens(ref)
        np.savez_compressed(os.path.join(GOLD, 'ops.npz'), **fx)
    cases = CASES
    for name, kw in cases:
        if only and name not in only:
            continue
        fx, rep = run_step_case(ref, **kw)
        ckpt = fx.pop('_ckpt', None)
        if ckpt is not None:  # a reference-format state.dict (torch.save, like model/worker.py:387-389)
            torch.save(ckpt, os.path.join(GOLD, name + '_ref_state.dict'))
        np.savez_compressed(os.path.join(GOLD, name + '.npz'), **fx)
        print(name, 'written', os.path.getsize(os.path.join(GOLD, name + '.npz')) // 1024, 'KiB')


if __name__ == '__main__':
    main()
