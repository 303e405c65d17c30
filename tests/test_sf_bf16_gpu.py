"""BASELINE config 2: DispNetS with bf16 ACTIVATION STORAGE (csrc/conv_bf16.hip: nhwc feature maps in bf16, one bf16 x bf16
product per MAC, fp32 accumulation; parameters, their gradients, disparities and losses fp32).

Tolerances.  This mode is NOT the north star's parity path (disparity L1 < 1e-4 vs the fp32 CPU reference is met by the
fp32-result path, tests/test_sf_gpu.py); it trades accuracy for bytes and matrix-core products, so it is held to its own,
stated bounds:
  * operator level, EXACT-ARITHMETIC reference: against torch's fp32 convolution of the SAME bf16-rounded operands the
    outputs agree to bf16 output rounding (2^-8 relative, asserted 1.2 * 2^-8 of the largest entry) and the fp32 weight /
    bias gradients to 2e-3;
  * network level, vs the fp32 CPU oracle on fp32 inputs: full-resolution disparity L1 < 0.05 px on a 0..128 px range
    (measured on MI355X: 0.0096 px at scale 0, 3e-4 .. 4e-5 px at the coarser scales, random weights; every layer rounds
    its output to 8 significant bits), parameter-gradient cosine > 0.99 against the oracle's gradients (measured >= 0.9956);
  * whole step on a reference-generated fixture: loss terms within 1.5 % (measured <= 0.15 %)."""
import argparse
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from oracle import dis_oracle as O

BF = torch.bfloat16


def bfr(t):
    return t.to(BF).float()


CASES = [
    # cin_w, cin_mem, cout, k, stride, transposed, h, w, x_fp32
    (2, 4, 32, 7, 2, False, 40, 36, True),      # conv1.0: fp32 network input
    (32, 32, 32, 7, 1, False, 20, 18, False),
    (32, 32, 64, 5, 2, False, 20, 18, False),
    (64, 64, 64, 5, 1, False, 11, 13, False),
    (128, 128, 256, 3, 2, False, 9, 7, False),
    (129, 136, 64, 3, 1, False, 12, 10, False),  # iconv3: 129 real channels in a 136-lane buffer
    (17, 24, 16, 3, 1, False, 16, 12, False),    # iconv1
    (64, 64, 32, 3, 2, True, 8, 7, False),       # upconv + crop (15 x 13 of 16 x 14)
    (512, 512, 512, 3, 2, True, 2, 2, False),
    # maps of >= 1024 output positions run the halo form (convb_halo_kernel): every tap-packing / chunking / stride case
    (2, 4, 32, 7, 2, False, 72, 80, True),       # 4 taps per k-step (cin <= 8), fp32 input, stride-2 halo
    (16, 16, 16, 3, 1, False, 40, 36, False),    # 2 taps per k-step (cin <= 16)
    (17, 24, 16, 3, 1, False, 40, 36, False),    # one chunk, ragged channels
    (32, 32, 32, 7, 1, False, 36, 40, False),    # 49 k-steps in weight groups
    (32, 32, 64, 5, 2, False, 72, 70, False),
    (64, 64, 64, 5, 1, False, 33, 37, False),    # two chunks
    (129, 136, 64, 3, 1, False, 34, 38, False),  # five chunks, the last one 8 channels wide
    (64, 64, 32, 3, 2, True, 36, 34, False),     # transposed conv and its input gradient: four tap-parity phases
    (512, 512, 512, 3, 1, False, 32, 27, False),  # a deep layer on a small map (< 1024 positions): streaming, 128-channel stages, split-K
]


@pytest.mark.parametrize('cin_w,cin_mem,cout,k,stride,transposed,h,w,x_fp32', CASES)
def test_convb_matches_exact_arithmetic(cin_w, cin_mem, cout, k, stride, transposed, h, w, x_fp32):
    from depthinspace_amd import ops
    g = torch.Generator().manual_seed(cin_w * 7 + cout + k)
    n = 2
    x = torch.randn(n, cin_w, h, w, generator=g)
    if transposed:
        wt = torch.randn(cin_w, cout, k, k, generator=g) / (cin_w * k * k / 4) ** 0.5
        oh, ow = 2 * h - 1, 2 * w - 1   # crop_like trims the last row / column
    else:
        wt = torch.randn(cout, cin_w, k, k, generator=g) / (cin_w * k * k) ** 0.5
    b = torch.randn(cout, generator=g) * 0.1
    # exact-arithmetic reference on the values the kernel sees: bf16-rounded x (unless the input is fp32: the kernel rounds it
    # when it stages it) and bf16-rounded weights, fp32 products and sums
    xr = bfr(x).clone().requires_grad_(True)
    wr = bfr(wt).clone().requires_grad_(True)
    br = b.clone().requires_grad_(True)
    pad = (k - 1) // 2
    if transposed:
        y = F.relu(F.conv_transpose2d(xr, wr, br, stride=2, padding=1, output_padding=1)[:, :, :oh, :ow])
    else:
        y = F.relu(F.conv2d(xr, wr, br, stride=stride, padding=pad))
    go = bfr(torch.randn(y.shape, generator=g))
    y.backward(go)
    xp = torch.zeros(n, h, w, cin_mem)
    xp[..., :cin_w] = x.permute(0, 2, 3, 1)
    xd = (xp if x_fp32 else xp.to(BF)).cuda().requires_grad_(not x_fp32)
    wd, bd = wt.cuda().requires_grad_(True), b.cuda().requires_grad_(True)
    if transposed:
        yd = ops.convg_transposed(xd, wd, bd, (oh, ow), 1, ops.ACT_RELU, dtype=BF)
    else:
        yd = ops.convg(xd, wd, bd, stride, pad, ops.ACT_RELU, need_dgrad=not x_fp32, dtype=BF)
    assert yd.dtype == BF and tuple(yd.shape) == (n, y.shape[2], y.shape[3], cout)
    scale = float(y.abs().max())
    err = float((yd.float().cpu().permute(0, 3, 1, 2) - y.detach()).abs().max())
    assert err <= 1.2 * 2 ** -8 * scale, (err, scale)
    yd.backward(go.permute(0, 2, 3, 1).contiguous().to(BF).cuda())
    # the backward reference must use the ReLU mask of the kernel's own (bf16-rounded) output: identical except at |y| ~ 0
    ws = float(wr.grad.abs().max())
    assert float((wd.grad.cpu() - wr.grad).abs().max()) < 4e-3 * ws
    assert float((bd.grad.cpu() - br.grad).abs().max()) < 4e-3 * float(br.grad.abs().max())
    if not x_fp32:
        gx = xd.grad.float().cpu()[..., :cin_w].permute(0, 3, 1, 2)
        assert float((gx - xr.grad).abs().max()) < 1.5 * 2 ** -8 * float(xr.grad.abs().max()) + 4e-3 * float(xr.grad.abs().max())
        if cin_mem > cin_w:
            assert float(xd.grad.float()[..., cin_w:].abs().max()) == 0.0


def test_convb_large_tiles():
    """convb_fwd128_kernel<128, ., 8> (eight waves on a 256-pixel x 128-cout tile: deep layers whose launch fills the device without
    split-K - at bench scale the 32 x 27 maps) against the exact-arithmetic reference of test_convb_matches_exact_arithmetic:
    8 images of 32 x 27, 128 -> 1024 channels; forward, input gradient (1024 -> 128 runs the small tile: cout 128 x 27 tiles) and
    weight gradient; the forward launch must report the large tile."""
    from depthinspace_amd import ops, lib
    g = torch.Generator().manual_seed(21)
    n, cin, cout, h, w = 8, 128, 1024, 32, 27
    x = torch.randn(n, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, 3, 3, generator=g) / (cin * 9) ** 0.5
    b = torch.randn(cout, generator=g) * 0.1
    xr, wr, br = bfr(x).clone().requires_grad_(True), bfr(wt).clone().requires_grad_(True), b.clone().requires_grad_(True)
    y = F.relu(F.conv2d(xr, wr, br, padding=1))
    go = bfr(torch.randn(y.shape, generator=g))
    y.backward(go)
    xd = x.permute(0, 2, 3, 1).contiguous().to(BF).cuda().requires_grad_(True)
    wd, bd = wt.cuda().requires_grad_(True), b.cuda().requires_grad_(True)
    lib.profile_start()
    yd = ops.convg(xd, wd, bd, 1, 1, ops.ACT_RELU, dtype=BF)
    tags = {t for (name, _, _, t, _) in lib.profile_stop() if name == 'dis_convb_run'}
    assert tags and all('256 x 128 tiles' in t for t in tags), tags
    scale = float(y.abs().max())
    assert float((yd.float().cpu().permute(0, 3, 1, 2) - y.detach()).abs().max()) <= 1.2 * 2 ** -8 * scale
    yd.backward(go.permute(0, 2, 3, 1).contiguous().to(BF).cuda())
    assert float((wd.grad.cpu() - wr.grad).abs().max()) < 4e-3 * float(wr.grad.abs().max())
    gx = xd.grad.float().cpu().permute(0, 3, 1, 2)
    assert float((gx - xr.grad).abs().max()) < (1.5 * 2 ** -8 + 4e-3) * float(xr.grad.abs().max())


@pytest.mark.parametrize('H,W', [(64, 56), (128, 108)])
def test_dispnets_bf16_vs_fp32_oracle(H, W):
    from depthinspace_amd.model import networks
    params = O.init_params(O.sf_param_shapes(), seed=5)
    g = torch.Generator().manual_seed(11)
    x = torch.randn(3, 2, H, W, generator=g)
    outs = O.sf_forward(params, x)
    gos = [torch.randn(o.shape, generator=g) / o.numel() ** 0.5 for o in outs]
    sum((o * go).sum() for o, go in zip(outs, gos)).backward()
    imsizes = [(H, W)]
    for _ in range(3):
        imsizes.append((imsizes[-1][0] // 2, imsizes[-1][1] // 2))
    net = networks.DispDecoder(channels_in=2, max_disp=128, imsizes=imsizes, act_dtype=BF)
    net.load_state_dict({k: v.detach() for k, v in params.items()})
    net = net.cuda()
    outs_d = net(x.cuda())
    l1s = []
    for o, od in zip(outs, outs_d):
        assert od.dtype == torch.float32 and tuple(od.shape) == tuple(o.shape)
        l1s.append(float((od.detach().cpu() - o.detach()).abs().mean()))
    sum((od * go.cuda()).sum() for od, go in zip(outs_d, gos)).backward()
    cos = []
    for k, p in net.named_parameters():
        a, b = p.grad.cpu().double().reshape(-1), params[k].grad.double().reshape(-1)
        cos.append((float(a @ b / (a.norm() * b.norm() + 1e-300)), k))
    print('DispNetS bf16', H, W, 'disparity L1 per scale', l1s, 'worst gradient cosine', min(cos))
    assert max(l1s) < 0.05, l1s
    assert min(cos)[0] > 0.99, min(cos)


def test_sf_step_bf16_vs_reference_golden(golden_dir):
    """whole DIS-SF step with bf16 activation storage on the reference-generated fixture's inputs: loss terms within 1.5 %
    (terms below 1e-3 within 1e-4 absolute), disparity L1 < 0.05 px, and the Adam update moves the same way."""
    from depthinspace_amd import synth
    from depthinspace_amd.model import networks, single_frame_worker
    from depthinspace_amd.trainer import FlatAdam
    G = np.load(os.path.join(golden_dir, 'sf_128x108_bs1.npz'))
    H, W, bs = int(G['H']), int(G['W']), int(G['bs'])
    settings = synth.make_settings(H, W)
    batch = synth.make_batch(settings, bs, 4, seed=int(G['bseed']))
    params = O.init_params(O.sf_param_shapes(), seed=int(G['pseed']))
    args = argparse.Namespace(use_pseudo_gt=False, lcn_radius=5, track_length=4, data_type='synthetic',
                              architecture='single_frame', epochs=1, warmup_epochs=150, train_batch_size=bs, max_disp=128)
    w = single_frame_worker.Worker(args, settings=settings)
    w.build_losses()
    w.current_epoch = int(G['epoch'])
    net = networks.DispDecoder(channels_in=2, max_disp=128, imsizes=w.imsizes, act_dtype=BF)
    net.load_state_dict({k: v.detach() for k, v in params.items()})
    net = net.cuda()
    opt = FlatAdam(net.parameters(), lr=1e-4)
    errs, outs = w.train_step(net, opt, {k: torch.from_numpy(v) for k, v in batch.items()})
    torch.cuda.synchronize()
    l1 = [float((o.detach().cpu() - torch.from_numpy(G[f'out{i}'])).abs().mean()) for i, o in enumerate(outs)]
    vals = np.array([float(e.detach()) for e in errs])
    print('bf16 step: disparity L1', l1, 'loss terms', vals, 'reference', G['vals'])
    assert max(l1) < 0.05
    assert np.all(np.abs(vals - G['vals']) <= 0.015 * np.abs(G['vals']) + 1e-4)
    agree = tot = 0
    named = dict(net.named_parameters())
    for k in G.files:
        if k.startswith('new:'):
            d_ref = torch.from_numpy(G[k]) - params[k[4:]].detach()
            d = named[k[4:]].detach().cpu() - params[k[4:]].detach()
            m = d_ref.abs() > 5e-5          # entries whose Adam step is not a ~0-gradient coin flip
            agree += int(((d * d_ref) > 0)[m].sum())
            tot += int(m.sum())
    assert tot > 1000 and agree / tot > 0.9, (agree, tot)


def test_worker_trains_and_evaluates_in_bf16():
    """The entry-point classes with the bf16 network (train_val.py, DIS_ACT_DTYPE=bf16): Worker.train_step for a few steps
    (losses finite and close to the fp32 network's from the same initial parameters, parameters move), the same step as a
    hipGraph (trainer.GraphedStep), and a no_grad evaluation forward (float32 disparities)."""
    from depthinspace_amd import synth
    from depthinspace_amd.model import single_frame_worker, networks
    from depthinspace_amd.trainer import FlatAdam, GraphedStep
    args = argparse.Namespace(use_pseudo_gt=False, lcn_radius=5, track_length=4, data_type='synthetic',
                              architecture='single_frame', epochs=1, warmup_epochs=150, train_batch_size=1, max_disp=128)
    H, W = 128, 108
    settings = synth.make_settings(H, W)
    w = single_frame_worker.Worker(args, settings=settings)
    w.build_losses()
    w.current_epoch = 2
    batch = {k: torch.from_numpy(v) for k, v in synth.make_batch(settings, 1, 4, seed=5, scene='bumps').items()}
    losses = {}
    for name, dt in (('f32', torch.float32), ('bf16', BF)):
        torch.manual_seed(0)
        net = networks.DispDecoder(channels_in=2, max_disp=128, imsizes=w.imsizes, act_dtype=dt).cuda()
        opt = FlatAdam(net.parameters(), lr=1e-4)
        p0 = opt.flat_p.clone()
        ls = []
        for _ in range(3):
            errs, _ = w.train_step(net, opt, batch)
            ls.append([float(e) for e in errs])
        losses[name] = np.array(ls)
        assert np.isfinite(losses[name]).all() and float((opt.flat_p - p0).abs().max()) > 0
        if dt == BF:
            g = GraphedStep(w, net, opt, batch, use_graph=True, warmup=1)
            g.run()
            g.run()
            torch.cuda.synchronize()
            assert g.mode == 'graph' and opt.step_count == 3 + 2 and np.isfinite(g.losses()).all()   # (warm-up steps do not train)
            with torch.no_grad():
                w.copy_data(batch, device=w.train_device, requires_grad=False, train=False)
                out = w.net_forward(net, None)
            d = out[0] if isinstance(out, (list, tuple)) else out
            assert d.dtype == torch.float32 and bool(torch.isfinite(d).all())
    np.testing.assert_allclose(losses['bf16'], losses['f32'], rtol=0.03, atol=2e-4)


def test_batched_weight_packing_equals_per_call_packing():
    """One packing launch per step (ops._PackBatch: dis_convb_pack_record / dis_convb_pack_batch / DIS_CONVB_PREPACKED) must be
    invisible: three training steps of DIS-SF with bf16 activation storage at 64 x 56 with it (record in step 1, batch launch from
    step 2 on) and without it give the same gradients in every step when each step starts from the same optimizer state (to the
    run-to-run noise of the step's float atomics, ~1e-6 of the largest gradient; weights that are one Adam step stale move them by
    ~1e-2), and an inference forward after the optimizer step packs for itself (the batch launch's
    weights are stale by then)."""
    import argparse
    from depthinspace_amd import synth, ops
    from depthinspace_amd.model import single_frame_worker, networks
    from depthinspace_amd.trainer import FlatAdam
    H, W = 64, 56
    settings = synth.make_settings(H, W)
    batch = {k: torch.from_numpy(v) for k, v in synth.make_batch(settings, 2, 4, seed=5).items()}

    def run(enabled, reset=True, ref=None):
        """three training steps.  ref: the per-step optimizer states of the reference run - this run then STARTS every step from the
        reference's state (parameters, Adam moments, step counter), so that each step's gradient can be compared on its own: without
        that the float-atomic noise of step k (1e-6 of the largest gradient) is amplified by Adam on near-zero gradients and by bf16
        rounding boundaries into 1e-3 parameter differences two steps later, the size of the stale-weight error the test looks for
        (seen on the round-6 boxes with round 5's code as well)."""
        PB = ops._PackBatch
        if reset:
            PB.invalidate()
            PB.enabled, PB.cache, PB.state = enabled, {}, 'idle'
        args = argparse.Namespace(use_pseudo_gt=False, lcn_radius=5, track_length=4, data_type='synthetic',
                                  architecture='single_frame', epochs=1, warmup_epochs=150, train_batch_size=2, max_disp=128)
        worker = single_frame_worker.Worker(args, settings=settings, train_device='cuda:0')
        torch.manual_seed(3)
        net = networks.DispDecoder(channels_in=2, max_disp=128, imsizes=worker.imsizes, act_dtype=torch.bfloat16).cuda()
        worker.build_losses(device=torch.device('cuda', 0))
        worker.current_epoch = 2
        worker.device_aug = False   # (no random augmentation: the two runs see the same images)
        np.random.seed(0)
        opt = FlatAdam(net.parameters(), lr=1e-3)
        states, rec = [], []
        for k in range(3):
            if ref is not None and k > 0:
                for dst, src in zip((opt.flat_p, opt.exp_avg, opt.exp_avg_sq, opt.state_dev), ref[k - 1][1:]):
                    dst.copy_(src)
                ops.params_changed()   # (as a checkpoint load does; the step's own begin_step packs the new values)
            worker.train_step(net, opt, batch)
            states.append(PB.state)
            rec.append((opt.flat_g.clone(), opt.flat_p.clone(), opt.exp_avg.clone(), opt.exp_avg_sq.clone(), opt.state_dev.clone()))
        with torch.no_grad():
            worker.copy_data(batch, device=torch.device('cuda', 0), requires_grad=False, train=False)
            out = worker.net_forward(net, worker.read_optical_flow(train=False))
        torch.cuda.synchronize()
        o = out[0] if isinstance(out, (list, tuple)) else out
        return rec, o.detach().float().clone(), states

    def check_grads(rec, ref, what):
        # a weight that is one Adam step stale (1e-3 on weights of ~5e-2) moves the gradients by ~1e-2 of the largest entry; the
        # float-atomic noise of two identical steps is ~1e-6
        for k in range(3):
            scale = float(ref[k][0].abs().max())
            err = float((rec[k][0] - ref[k][0]).abs().max()) / scale
            assert err < 1e-4, (what, k, err)

    try:
        r0, o0, st0 = run(False)
        r1, o1, st1 = run(True, ref=r0)
        assert st1 == ['recording', 'ready', 'ready'], st1
        # the network of that run is gone (its weights were freed): the table recorded from it must not be replayed - the next
        # begin_step drops it and records the new network (raw pointers in the descriptor table)
        import gc
        gc.collect()
        torch.cuda.empty_cache()
        assert ops._PackBatch.dead or ops._PackBatch.state != 'ready'
        r2, o2, st2 = run(True, reset=False, ref=r0)
        assert st2 == ['recording', 'ready', 'ready'], st2
        assert len(ops._PackBatch.retired) > 0
    finally:
        PB = ops._PackBatch
        PB.invalidate()
        PB.enabled, PB.cache, PB.state = True, {}, 'idle'
        del PB.retired[:]
    check_grads(r1, r0, 'first run')
    check_grads(r2, r0, 'second run (table re-recorded)')
    # (the inference outputs come from parameters that differ by that noise; a bf16-stored activation that sits on a rounding boundary then
    # lands one bf16 ulp - 2^-8 relative - apart, and the largest difference over 1.8 M outputs finds such pixels: 2.3e-3 of the
    # largest output was seen in 2 of 6 runs of this file, scripts/diag/repeat_case.py, with a mean difference of 1.6e-4 of it - the
    # same figure every time: one early flip and its deterministic wake.)
    assert float((o1 - o0).abs().max()) < 1e-2 * float(o0.abs().max()), float((o1 - o0).abs().max())
    assert float((o1 - o0).abs().mean()) < 1e-3 * float(o0.abs().max()), float((o1 - o0).abs().mean())
