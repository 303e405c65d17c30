#!/usr/bin/env python
"""Benchmark of the hot path: DIS-MF training step (FuseNet fwd + losses + bwd + Adam), bs=4 per GPU,
512x432 default-pattern synthetic data, fp32, on N MI355X of one node (one process per GPU, RCCL).

    python bench.py --gpus 1 --steps 20 --warmup 5
    python bench.py --gpus N --steps K --warmup W      (no launcher: spawns its N ranks itself, one fresh process per GPU)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \\
        bench.py --gpus N --steps K --warmup W

Prints ONE JSON line on rank 0 (contract in the task description): BASELINE.json's metric
(training frames/s, frame = one 512x432 image of a 4-frame track), plus
  roofline     : the dominant kernel (MFMA fp32 32->32 3x3 implicit-GEMM conv) against the fp32 matrix peak,
                 duration measured live with HIP events around every launch of that kernel in one step;
  cpu_baseline : the CPU oracle (pure PyTorch restatement of the reference step) timed on the host cores on a
                 bounded sample (rank 0, N=1 only);
  disp_l1_vs_ref : mean |disparity(HIP) - disparity(reference)| on that sample's inputs (the metric's "disp L1 vs ref"): the
                 free-running HIP forward against the reference's own 512x432 output (tests/golden/mf_512x432_bs1.npz), with the
                 comparison against the CPU oracle of this host as secondary fields;
  ranks_seen / devices_seen / replicas_equal : at N > 1 every rank checks the world size, that no two ranks share a
                 device (RCCL) and, after the timed loop, that all parameter replicas are bit-identical.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np
import torch

H, W, TL = 512, 432, 4
PEAK_FP32_MFMA_TFLOPS = 157.3   # /opt/skills/guides/MI355X_MICROARCH.md, "Peak FP32 (matrix)"
PEAK_BF16_MFMA_TFLOPS = 2500.0  # same guide, "Peak BF16/FP16 MFMA ~2.5 PF dense"
PEAK_HBM_GBS = 8000.0           # same guide, HBM3E spec peak
# compulsory conv activation traffic 5.65 GB / sample forward, training = 3 x, bs = 4 samples; per-pixel
# path 1.03 GB / frame incl. Conv3D, 16 frames
SURVEY_8D_STEP_BYTES = 3 * 5.65e9 * 4 + 1.03e9 * 16
SURVEY_8D_STEP_BYTES_NOTE = 'SURVEY 8(d): convs 3 x 5.65 GB/sample x 4 samples = 67.8 GB + per-pixel path 1.03 GB/frame x 16 = 16.5 GB'
# DIS-SF (config 2, bs = 8 -> 32 images): SURVEY 8(d) / BASELINE.md section 2: compulsory conv activation traffic 0.198 GB / image
# forward in fp32 (half of it with bf16 activation storage), training = 3 x; per-pixel kernels 0.10 GB / frame (fp32 either way)
SURVEY_8D_SF_STEP_BYTES = {'f32': 3 * 0.198e9 * 32 + 0.10e9 * 32, 'bf16': 3 * 0.099e9 * 32 + 0.10e9 * 32}
SURVEY_8D_SF_STEP_BYTES_NOTE = ('SURVEY 8(d): convs 3 x 0.198 GB/image (fp32; 0.099 with bf16 activation storage) x 32 images + per-pixel '
                                'kernels 0.10 GB/frame x 32')


def make_args(bs, arch='multi_frame'):
    return argparse.Namespace(use_pseudo_gt=False, lcn_radius=5, track_length=TL, data_type='synthetic',
                              architecture=arch, epochs=1, warmup_epochs=150, train_batch_size=bs,
                              max_disp=128)


def make_device_batch(settings, bs, seed, dev):
    from depthinspace_amd import synth
    b = synth.make_batch(settings, bs, TL, seed=seed)
    out = {}
    stacked = np.zeros((bs, TL * TL, 2, H, W), np.float32)
    for k, v in b.items():
        if k.startswith('flow_'):
            i, j = int(k[5]), int(k[6])
            stacked[:, i * TL + j] = v[:, 0]
        else:
            out[k] = torch.from_numpy(v).to(dev)
    out['_flow_stacked'] = torch.from_numpy(stacked).to(dev)
    return out


def conv_flops(n, ho, wo, cin, cout, k):
    return 2.0 * n * ho * wo * cin * cout * k * k


# The entry points that launch the halo-resident 3x3 kernels (conv_f16x2_kernel / conv_bf16x3_kernel), keyed by C-ABI name:
# where (n, h, w) of the staged input and the padding sit among the call's int arguments (lib.call records them in order), and how
# many EXTRA activation-sized tensors the fused prologue (`in_extra`, at the input's positions) and epilogue (`out_extra`, at the
# output's positions) read besides x in / y out.  DIS_CONV_ACCUM = 0x100 in `act`: y = y_old + conv.
def _f(nhw, pad, in_extra=0, out_extra=0):
    return {'nhw': tuple(nhw), 'pad': pad, 'in_extra': in_extra, 'out_extra': out_extra}


CONV3X3_FORMS = {
    # (n, hin, win, cin, cout, k, stride, pad, act)
    'dis_conv2d_fwd_bf16x3': lambda ia, nptr: _f(ia[0:3], ia[7], 0, 1 if ia[8] & 0x100 else 0),
    # (mode, w_o, w_i, w_row_stride, n, hin, win, cin, cout, k, stride, pad, act)
    'dis_conv2d_fwd_bf16x3_oihw': lambda ia, nptr: _f(ia[4:7], ia[11], 0, 1 if ia[12] & 0x100 else 0),
    # (w_o, w_i, w_row_stride, n, hin, win, cin, cout, k, stride, pad, act): GroupNorm applied on load
    'dis_conv2d_fwd_bf16x3_gn': lambda ia, nptr: _f(ia[3:6], ia[10], 0, 1 if ia[11] & 0x100 else 0),
    # (act, w_o, w_i, w_row_stride, n, h, w, cin, cout, pad, accum): gy * act'(y) staged (+ y read), optional accumulate
    'dis_conv2d_dgrad_bf16x3_act': lambda ia, nptr: _f(ia[4:7], ia[9], 1, 1 if ia[10] else 0),
    # (w_o, w_i, w_row_stride, n, h, w, cin, cout, pad): + the GroupNorm input at the output's positions (channel sums)
    'dis_conv2d_dgrad_bf16x3_gnsums': lambda ia, nptr: _f(ia[3:6], ia[8], 0, 1),
    # the accumulating residual form: + old gradient + GroupNorm input (+ the activation output when the pattern has a SELU:
    # 6 tensor arguments instead of 5)
    'dis_conv2d_dgrad_bf16x3_gnsums_res': lambda ia, nptr: _f(ia[3:6], ia[8], 0, 3 if nptr >= 6 else 2),
    # activation-fused input gradient + GroupNorm input + activation output (final_conv; not accumulating)
    'dis_conv2d_dgrad_bf16x3_act_gnsums_res': lambda ia, nptr: _f(ia[3:6], ia[8], 1, 2),
    # (in_act, w_o, w_i, w_row_stride, accumulate, n, h, w, c): GroupNorm backward applied on load - the GroupNorm input read and the
    # pre-activation gradient written beside the operand (2 extra tensors on the input side); old gradient / GroupNorm input /
    # activation output of the epilogue forms on the output side (tensor arguments: 6, 8 with the channel sums, 9 with SELU')
    # (w_o, w_i, w_row_stride, n, h, w, cin, cout, act): the previous ResNetBlock's output formed on load - its residual read and the
    # output written beside the operand (2 extra tensors on the input side)
    'dis_conv2d_fwd_f16x2_gnres': lambda ia, nptr: _f(ia[3:6], 1, 2, 0),
    'dis_conv2d_dgrad_f16x2_gnb': lambda ia, nptr: _f(ia[5:8], 1, 2, (1 if ia[4] else 0) + (1 if nptr >= 8 else 0) + (1 if nptr >= 9 else 0)),
}


def knn_tie_check(tap, hip_sets, ulps=16.0):
    """Every Conv3D row (output pixel) whose free-running-oracle neighbour SET differs from the HIP path's, judged on the oracle's
    own keys.  The two selections of a row are compared as SORTED KEY VECTORS: the 9 smallest of the 36 keys are unique as values,
    only which of several equal-keyed candidates carries a value is a tie-break.  A key is a squared distance of plane
    coordinates, key = |p_nb - p_ctr|^2, so an error delta in a plane coordinate moves it by 2 sqrt(key) delta + delta^2.  With
    delta = `ulps` fp32 ulps of the largest plane coordinate (the K = 3 products of the view change, the bilinear warp and the
    division by z are ~10 roundings; how THIS host's CPU kernels round them is what differs from the fixture host, DESIGN.md
    section 4) a row is a TIE when its two sorted key vectors agree within that bound at the row's 9th-smallest key, and a NON-TIE
    otherwise: HIP picked a candidate whose key is distinctly larger than one it left out - a selection error.  The histogram of
    the relative gaps is reported so that the bound can be judged: a wrong neighbour shows gaps of percents, not 1e-4.
    non_tie_rows must be 0 for the comparison on the HIP neighbour sets to stand in for the free-running one.
    tap: oracle.CONV3D_TAP entries {'name', 'target', 'idx' (bs,ho,wo,9), 'key' (bs,ho,wo,36), 'plane_absmax'}; hip_sets: (core,
    quarter) uint8 (tl,bs,ho,wo,9)."""
    seen, rows, differ, non_tie, worst, worst_over_tol = set(), 0, 0, 0, 0.0, 0.0
    per = {}
    edges = [1e-6, 1e-5, 1e-4, 1e-3, 1e-2]
    hist = [0] * (len(edges) + 1)
    for e in tap:
        res = 0 if e['name'].endswith('conv3d_1') else 1   # conv3d_1: core -> quarter (stride 2); conv3d_2: quarter (stride 1)
        k = (res, e['target'])
        if k in seen or e['target'] is None:
            continue   # (the four blocks share a geometry: same keys, same selection)
        seen.add(k)
        key = e['key'].reshape(-1, e['key'].shape[-1]).double()
        io = e['idx'].reshape(-1, 9).long()
        ih = hip_sets[res][e['target']].reshape(-1, 9).long()
        so, sh = io.sort(dim=1).values, ih.sort(dim=1).values
        d_rows = (so != sh).any(dim=1)
        ko = key.gather(1, io).sort(dim=1).values
        kh = key.gather(1, ih).sort(dim=1).values
        k9 = ko[:, -1].clamp_min(1e-30)
        gap_abs = (kh - ko).abs().max(dim=1).values
        delta = ulps * 2.0 ** -24 * max(float(e.get('plane_absmax', 1.0)), 1e-3)
        tol = 2.0 * k9.sqrt() * delta + delta * delta
        nt = d_rows & (gap_abs > tol)
        rows += key.shape[0]
        differ += int(d_rows.sum())
        non_tie += int(nt.sum())
        if bool(d_rows.any()):
            rel = (gap_abs / k9)[d_rows]
            worst = max(worst, float(rel.max()))
            worst_over_tol = max(worst_over_tol, float((gap_abs / tol)[d_rows].max()))
            lo = 0.0
            for i, hi in enumerate(edges + [float('inf')]):
                hist[i] += int(((rel > lo) & (rel <= hi)).sum()) + (int((rel == 0).sum()) if i == 0 else 0)
                lo = hi
        name = ('core' if res == 0 else 'quarter')
        per.setdefault(name, [0, 0])
        per[name][0] += key.shape[0]
        per[name][1] += int(d_rows.sum())
    return {'conv3d_rows': rows, 'rows_whose_set_differs': differ, 'non_tie_rows': non_tie,
            'largest_relative_key_gap_of_a_differing_row': worst, 'largest_gap_over_bound': worst_over_tol,
            'bound': f'2 sqrt(k9) d + d^2, d = {ulps:g} fp32 ulps of the largest plane coordinate',
            # the smallest d (in ulps) under which every differing row still counts as a tie (the bound is ~linear in d)
            'ulps_needed': ulps * worst_over_tol,
            'relative_gap_histogram_of_differing_rows': dict(zip(['<=1e-6', '<=1e-5', '<=1e-4', '<=1e-3', '<=1e-2', '>1e-2'], hist)),
            'by_resolution_rows_and_differing': per, 'geometries_checked': len(seen), 'pass': non_tie == 0}


def cpu_baseline(arch='multi_frame', timed_steps=3, knn_sets=None):
    """Oracle training step on the host cores: 1 warm-up + `timed_steps` timed steps (Adam state carried along), bs=1
    (one 4-frame track), full resolution.  The bounded sample of BASELINE.md section 3; reported, never the target."""
    from depthinspace_amd import synth
    from oracle import dis_oracle as O
    settings = synth.make_settings(H, W)
    batch = synth.make_batch(settings, 1, TL, seed=1234)
    params = O.init_params(O.mf_param_shapes() if arch == 'multi_frame' else O.sf_param_shapes(), seed=0)
    ctx = O.StepContext(settings)
    st = {'step': 0, 'm': {}, 'v': {}}
    tb = {k: torch.from_numpy(v) for k, v in batch.items()}
    nthr = torch.get_num_threads()
    want = int(os.environ.get('DIS_CPU_BASELINE_THREADS', str(min(nthr, 32))))  # past ~32 threads the small convs slow down
    torch.set_num_threads(want)
    try:
        times = []
        first = None
        for i in range(1 + timed_steps):
            if i == 0 and arch == 'multi_frame' and knn_sets is not None:
                O.CONV3D_TAP = []   # the warm-up step records the oracle's own Conv3D keys and selections (tie analysis below)
            t0 = time.time()
            try:
                r = O.train_step(ctx, arch, params, tb, adam_state=st, epoch=2)
            finally:
                tap, O.CONV3D_TAP = O.CONV3D_TAP, None
            times.append(time.time() - t0)
            if i == 0:   # the warm-up step starts from init_params(seed=0): its disparity is the reference of `disp_l1_vs_ref`
                out = r['out'] if arch == 'multi_frame' else r['out'][0]
                first = {'out': out.detach().clone(), 'vals': [float(v.detach()) for v in r['vals']]}
                if tap:
                    first['tie_check'] = knn_tie_check(tap, knn_sets)
                del tap
    finally:
        torch.set_num_threads(nthr)
    dt = sum(times[1:]) / timed_steps
    tag = 'DIS-MF' if arch == 'multi_frame' else 'DIS-SF'
    if knn_sets is not None and arch == 'multi_frame':
        # (untimed) the oracle forward once more from the initial parameters, on the HIP path's Conv3D neighbour sets: the
        # oracle's own top-9-of-36 depends on how THIS host's BLAS rounds a K = 3 product at exact key ties, which is not
        # the rounding of the host the reference fixtures were generated on; the HIP selection is
        # pinned to the reference's by tests/test_step_gpu.py and tests/test_fullsize_gpu.py
        p0 = O.init_params(O.mf_param_shapes(), seed=0)
        O.CONV3D_FORCE = {'core': knn_sets[0].long(), 'quarter': knn_sets[1].long()}
        try:
            with torch.no_grad():
                data = O.copy_data(ctx, tb)
                first['out_forced'] = O.mf_net_forward(ctx, p0, data, O.read_optical_flow(data, TL)).detach().clone()
        finally:
            O.CONV3D_FORCE = None
    return {'value': TL / dt, 'unit': 'frames/s', 'cores': want, 'kind': 'port',
            'sample': f'CPU oracle training step, {tag} bs=1 (one 4-frame track) 512x432 fp32: 1 warm-up ({times[0]:.1f} s) + '
                      f'{timed_steps} timed steps (mean {dt:.1f} s), torch threads={want} of os.cpu_count()={os.cpu_count()}'}, first


def hip_first_step(arch, settings, dev, dtype='f32', scene='plane'):
    """The HIP path on cpu_baseline()'s inputs: bs=1, batch seed 1234, init_params(seed=0), epoch 2 - one forward + losses.
    Returns (disparity on the host, loss terms, Conv3D neighbour sets or None)."""
    from depthinspace_amd import synth
    from depthinspace_amd.model import multi_frame_networks, multi_frame_worker, single_frame_worker, networks
    from oracle import dis_oracle as O
    mf = arch == 'multi_frame'
    params = O.init_params(O.mf_param_shapes() if mf else O.sf_param_shapes(), seed=0)
    batch = synth.make_batch(settings, 1, TL, seed=1234, scene=scene)
    if mf:
        net = multi_frame_networks.FuseNet(imsize=(H, W), K=settings.K, baseline=settings.baseline, track_length=TL, max_disp=128)
        worker = multi_frame_worker.Worker(make_args(1), settings=settings, train_device=str(dev))
    else:
        worker = single_frame_worker.Worker(make_args(1, 'single_frame'), settings=settings, train_device=str(dev))
        kw = {'act_dtype': torch.bfloat16} if dtype == 'bf16' else {}
        net = networks.DispDecoder(channels_in=2, max_disp=128, imsizes=worker.imsizes, **kw)
    net.load_state_dict({k: v.detach() for k, v in params.items()})
    net = net.to(dev)
    worker.build_losses(device=dev)
    worker.current_epoch = 2
    with torch.no_grad():
        aug, worker.device_aug = worker.device_aug, False   # the oracle step has no augmentation either
        worker.copy_data({k: torch.from_numpy(v) for k, v in batch.items()}, device=dev, requires_grad=False, train=True)
        worker.device_aug = aug
        flow = worker.read_optical_flow(train=True)
        out = worker.net_forward(net, flow)
        errs = worker.loss_forward(out, True, flow)
    torch.cuda.synchronize()
    o = out if mf else out[0]
    sets = [t.cpu() for t in net.last_knn_index] if mf else None
    return o.detach().float().cpu(), [float(e) for e in errs], sets


def spawn_ranks(n):
    """Start `n` copies of this command line, one per GPU, with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT
    set; children are fresh interpreter processes (no exec of a GPU-initialised process, nothing inherited).  Rank 0's
    stdout (the JSON line) passes through.  Returns the largest exit code; a rank that dies takes the others with it."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY='0')
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    alive = list(procs)
    while alive:
        for p in list(alive):
            try:
                c = p.wait(timeout=0.5)
            except subprocess.TimeoutExpired:
                continue
            alive.remove(p)
            if c != 0:
                rc = max(rc, c if c > 0 else 1)
                for q in alive:   # exactly the children started above
                    q.terminate()
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--bs', type=int, default=None, help='tracks per GPU (default 4 for multi_frame, 8 for single_frame)')
    ap.add_argument('--arch', default='multi_frame', choices=['multi_frame', 'single_frame'],
                    help='multi_frame = BASELINE.json metric (config 3/4); single_frame = DIS-SF (config 2, run in fp32)')
    ap.add_argument('--dtype', default='f32', choices=['f32', 'bf16'],
                    help='single_frame only: bf16 = DispNetS with bf16 activation storage (BASELINE config 2; one bf16 product per '
                         'MAC, fp32 accumulate; parameters / disparities / losses fp32), reported under its own label')
    ap.add_argument('--no-graph', action='store_true', help='launch every kernel eagerly instead of one hipGraph')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--dump-calls', default=None, help='write the per-call HIP-event times of one eager step to this file')
    ap.add_argument('--no-eager-leg', action='store_true', help='skip the eager-launch timing of the same step')
    ap.add_argument('--backend', default='nccl', choices=['nccl', 'gloo'],
                    help='torch.distributed backend for --gpus > 1 (nccl = RCCL over xGMI; gloo only to exercise the '
                         'multi-rank code path on a box with fewer GPUs than ranks)')
    ap.add_argument('--no-extra-legs', action='store_true',
                    help='the default N = 1 DIS-MF run also times, as fresh child processes after everything of this run has been '
                         'measured, (1) the same step on the strict three-term bf16 split (DIS_CONV_SPLIT=bf16x3, >= 24-bit operands: '
                         '`strict_fp32_frames_per_s`) and (2) BASELINE config 2, DIS-SF bs=8, with bf16 activation storage and in '
                         'fp32 (`extra_dis_sf`); this flag skips them (profiling runs, child runs)')
    ap.add_argument('--with-sf', action='store_true', help='(accepted for compatibility: the DIS-SF legs are on by default)')
    ap.add_argument('--epoch', type=int, default=2, help='training epoch the step models (epoch<2 adds the L1 warm-up term)')
    args = ap.parse_args()
    if args.bs is None:
        args.bs = 4 if args.arch == 'multi_frame' else 8
    if args.dtype == 'bf16' and args.arch != 'single_frame':
        raise SystemExit('--dtype bf16 is the DIS-SF configuration (BASELINE config 2); DIS-MF runs fp32')

    if 'WORLD_SIZE' not in os.environ and args.gpus > 1:
        # bare `python bench.py --gpus N`: this process never touches a GPU; it starts the N ranks as fresh children
        # (the environment torch.distributed.run would give them), waits, and leaves with their worst exit code
        raise SystemExit(spawn_ranks(args.gpus))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        raise SystemExit(f'--gpus {args.gpus} but WORLD_SIZE={world}: launch one rank per GPU '
                         f'(python bench.py --gpus N spawns them itself; torch.distributed.run --nproc-per-node N works too)')
    ndev = torch.cuda.device_count()
    if local_rank >= ndev and args.backend == 'nccl':
        raise SystemExit(f'rank {rank}: local rank {local_rank} but only {ndev} GPU(s) visible')
    dev_idx = local_rank % max(ndev, 1)  # ranks share a device only in the gloo plumbing test
    torch.cuda.set_device(dev_idx)
    dev = torch.device('cuda', dev_idx)
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if args.backend == 'nccl':
            torch.distributed.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)
        else:
            torch.distributed.init_process_group('gloo', rank=rank, world_size=world)

    from depthinspace_amd import synth, lib, ops
    from depthinspace_amd.model import multi_frame_networks, multi_frame_worker, single_frame_worker, networks
    from depthinspace_amd.trainer import FlatAdam
    lib.check_all_symbols()

    # ---- the job checks its own shape: N ranks, one device each (RCCL), before anything is timed
    ranks_seen, devices_seen = [rank], [dev_idx]
    if world > 1:
        assert torch.distributed.get_world_size() == args.gpus, (torch.distributed.get_world_size(), args.gpus)
        me = torch.tensor([rank, dev_idx, torch.cuda.current_device()], dtype=torch.int64,
                          device=dev if args.backend == 'nccl' else 'cpu')
        got = [torch.zeros_like(me) for _ in range(world)]
        torch.distributed.all_gather(got, me)
        ranks_seen = sorted(int(g[0]) for g in got)
        devices_seen = [int(g[2]) for g in got]
        assert ranks_seen == list(range(world)), ranks_seen
        if args.backend == 'nccl':
            assert len(set(devices_seen)) == world, ('ranks share a device', devices_seen)

    settings = synth.make_settings(H, W)
    torch.manual_seed(0)  # identical initial weights on every rank
    mf = args.arch == 'multi_frame'
    if mf:
        net = multi_frame_networks.FuseNet(imsize=(H, W), K=settings.K, baseline=settings.baseline, track_length=TL,
                                           max_disp=128).to(dev)
        worker = multi_frame_worker.Worker(make_args(args.bs), settings=settings, train_device=str(dev))
    else:
        worker = single_frame_worker.Worker(make_args(args.bs, 'single_frame'), settings=settings,
                                            train_device=str(dev))
        if args.dtype == 'bf16':
            net = networks.DispDecoder(channels_in=2, max_disp=128, imsizes=worker.imsizes, act_dtype=torch.bfloat16).to(dev)
        else:
            net = networks.DispDecoder(channels_in=2, max_disp=128, imsizes=worker.imsizes).to(dev)
    worker.build_losses(device=dev)
    worker.current_epoch = args.epoch
    opt = FlatAdam(net.parameters(), lr=1e-4, world_size=world)
    batch = make_device_batch(settings, args.bs, 1234 + rank, dev)  # weak scaling: own tracks per rank

    from depthinspace_amd.trainer import GraphedStep
    # the product's own step object: Worker.train_epoch(use_graph=True) runs the same one (trainer.GraphedStep)
    # N = 1: the captured step.  N > 1: the eager step, whose gradient buckets are all-reduced on a communication stream
    # while the backward pass runs (trainer.FlatAdam); eager launch costs nothing here (same frames/s as the graph at N = 1,
    # `eager_launch_frames_per_s` below)
    want_graph = (not args.no_graph) and world == 1
    # strict: a capture failure raises - a line that says hip_graph must have timed the graph
    stepper = GraphedStep(worker, net, opt, batch, use_graph=want_graph, warmup=max(1, min(args.warmup, 2)), strict=True)

    def fwd_bwd():  # (roofline leg below)
        stepper._forward_loss().backward()

    def step():
        stepper.run()

    # ---- warm-up (the first graph-mode call runs eager warm-up steps, then captures)
    for _ in range(max(1, args.warmup)):
        step()
    use_graph = stepper.use_graph
    assert (stepper.mode == 'graph') == bool(use_graph), stepper.mode

    def barrier():
        if world > 1:
            torch.distributed.barrier()

    torch.cuda.synchronize()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
        dt = float(tt)
    nterms = stepper.nterms
    losses = stepper.losses()
    adam_steps = opt.step_count
    # ---- replicas identical after the timed steps: every rank's parameter checksum (fp64 sum and sum of squares of the flat
    # parameter buffer, and its step counter) gathered and compared bit for bit
    replicas_equal = None
    if world > 1:
        ck = torch.stack([opt.flat_p.double().sum(), (opt.flat_p.double() ** 2).sum(),
                          opt.state_dev[0].double()])
        if args.backend != 'nccl':
            ck = ck.cpu()
        cks = [torch.zeros_like(ck) for _ in range(world)]
        torch.distributed.all_gather(cks, ck)
        replicas_equal = all(bool(torch.equal(c, cks[0])) for c in cks)
        assert replicas_equal, ('parameter replicas diverged', [c.tolist() for c in cks])

    # ---- the same loop launched eagerly (what Worker.train_step does without DIS_TRAIN_GRAPH): reported beside the headline
    eager_fps = None
    if rank == 0 and world == 1 and use_graph and not args.no_eager_leg:
        est = GraphedStep(worker, net, opt, batch, use_graph=False)
        for _ in range(2):
            est.run()
        torch.cuda.synchronize()
        te = time.perf_counter()
        ne = max(3, min(10, args.steps))
        for _ in range(ne):
            est.run()
        torch.cuda.synchronize()
        eager_fps = args.bs * TL * ne / (time.perf_counter() - te)

    # ---- roofline leg: HIP events around every launch of the dominant kernel during one eager step
    roof = None
    hbm = None
    kernel_ms = None
    if rank == 0:
        snap = [t.clone() for t in (opt.flat_p, opt.exp_avg, opt.exp_avg_sq, opt.state_dev)]
        overlap, opt.overlap = opt.overlap, False   # rank-local leg: the gradient buckets must not start all-reduces here
        lib.profile_start()
        fwd_bwd()                    # (only rank 0 runs this leg)
        opt.step(all_reduce=False)
        rec = lib.profile_stop()
        opt.overlap = overlap
        for t, c in zip((opt.flat_p, opt.exp_avg, opt.exp_avg_sq, opt.state_dev), snap):
            t.copy_(c)               # the profiled extra step is undone: replicas stay identical
        if args.dump_calls:
            os.makedirs(os.path.dirname(os.path.abspath(args.dump_calls)), exist_ok=True)
            with open(args.dump_calls, 'w') as f:
                for name, ia, ms, tag, nptr in rec:
                    f.write('%s %.4f %s # %s\n' % (name, ms, ' '.join(str(v) for v in ia), tag))
        per = {}
        for name, ia, ms, tag, nptr in rec:
            # label = entry point -> the kernel family that served it (the `*_bf16x3*` entry points run the two-term fp16 kernels
            # by default; dis_last_kernel() reports what was launched)
            label = name + (' -> ' + tag if tag else '')
            per.setdefault(label, [0, 0.0])
            per[label][0] += 1
            per[label][1] += ms
        rec3 = [(name, ia, ms) for name, ia, ms, _, _ in rec]
        peak, peak_note = PEAK_FP32_MFMA_TFLOPS, 'fp32 MFMA dense peak'
        roof_bytes = None
        if mf:
            # EVERY launch of the dominant kernel template in one step: the 32 -> 32 instances of conv_f16x2_kernel (or of
            # conv_bf16x3_kernel under DIS_CONV_SPLIT=bf16x3), selected by the kernel the library reports, whatever entry point
            # asked for it: forward, GroupNorm-on-load forward, plain / activation-fused / accumulating input gradients and the
            # input gradients that leave the GroupNorm-backward channel sums (76 launches per step)
            f2 = lib.fn('dis_get_conv_split')() == 1
            dom = 'conv_f16x2_kernel<32,32>' if f2 else 'conv_bf16x3_kernel<32,32>'
            sel, roof_bytes, forms = [], 0.0, {}
            # round 6: the backward of the 3x3 32 -> 32 layers is ONE launch of conv_bwd_fused_kernel (input gradient + weight
            # gradient; the call also holds the 7-us slab reduce).  Algorithmic bytes: gy (+ q of the GroupNorm-backward / act'
            # forms) and x in, gx out, + the accumulated-into gradient, + the epilogue's GroupNorm input / activation output where
            # they are not x itself, + gpre where it is stored; flops: both products.
            fsel, fbytes, fflops, fforms = [], 0.0, 0.0, {}
            for (name, ia, ms, tag, nptr), ptrs in zip(rec, lib.last_profile_ptrs):
                if name != 'dis_conv2d_bwd_fused_f16x2' or not tag.startswith('conv_bwd_fused_kernel'):
                    continue
                g_, q_, coef_, _, gpre_, _, _, _, _, gx_, acc_, abx_, acty_, _, x_ = ptrs[:15]
                n_, h_, w_ = ia[5:8]
                units = 1 + (q_ is not None) + (gpre_ is not None) + 1 + (1 if ia[4] else 0) + 1 + \
                    (abx_ is not None and abx_ != x_) + (acty_ is not None and acty_ != x_)
                fsel.append(((n_, h_, w_), ms))
                fbytes += 4.0 * 32 * n_ * h_ * w_ * units
                fflops += 2.0 * conv_flops(n_, h_, w_, 32, 32, 3)
                key = 'dis_conv2d_bwd_fused_f16x2' + (' (GroupNorm backward on load)' if coef_ is not None else '')
                fforms[key] = fforms.get(key, 0) + 1
            for name, ia, ms, tag, nptr in rec:
                if tag != dom:
                    continue
                d = CONV3X3_FORMS[name](ia, nptr)
                n_, h_, w_ = d['nhw']
                ho_, wo_ = h_ + 2 * d['pad'] - 2, w_ + 2 * d['pad'] - 2
                sel.append(((n_, ho_, wo_), ms))
                # algorithmic bytes of THIS launch: x in, y out, plus every tensor its fused prologue / epilogue reads
                # (old y of an accumulating launch, the activation output of a fused act', the GroupNorm input / output of the
                # channel-sum forms); weights, bias and the per-sample statistics are negligible
                roof_bytes += 4.0 * 32 * (n_ * h_ * w_ * (1 + d['in_extra']) + n_ * ho_ * wo_ * (1 + d['out_extra']))
                forms[name] = forms.get(name, 0) + 1
            nprod = 3 if f2 else 6
            if f2:
                kname = ('conv_f16x2_kernel<32,32,...> (fp32 conv as 3 fp16 products per MAC on v_mfma_f32_16x16x32_f16: two-term operand '
                         'split with power-of-two block scaling, fp32 accumulate; TFLOP/s are fp32-equivalent algorithmic flops; ALL '
                         '32 -> 32 3x3 launches of a step: forward, GroupNorm-on-load forward, every input-gradient form)')
            else:
                kname = ('conv_bf16x3_kernel<32,32,...> (fp32 conv as 6 bf16 products per MAC on v_mfma_f32_16x16x32_bf16, fp32 '
                         'accumulate; TFLOP/s are fp32-equivalent algorithmic flops; all 32 -> 32 3x3 launches of a step)')
            peak = PEAK_BF16_MFMA_TFLOPS / nprod
            peak_note = f'bf16 / fp16 MFMA dense peak (2500 TFLOP/s) / {nprod} products per fp32 MAC'
            if not sel:  # DIS_CONV_BF16X3=0: the fp32-MFMA kernel
                sel = [(ia, ms) for name, ia, ms in rec3 if name == 'dis_conv2d_fwd' and ia[3:7] == (32, 32, 3, 1)]
                kname = 'conv_fwd_kernel<32,32,3,3,1> (fp32 MFMA 16x16x4)'
                peak, peak_note = PEAK_FP32_MFMA_TFLOPS, 'fp32 MFMA dense peak'
                roof_bytes = sum(2.0 * ia[0] * ia[1] * ia[2] * 32 * 4 for ia, _ in sel)
                forms = {'dis_conv2d_fwd': len(sel)}
            fl = sum(conv_flops(ia[0], ia[1], ia[2], 32, 32, 3) for ia, _ in sel)
            # the family with the larger share of the step is THE dominant kernel of the `roofline` object; the other one is reported
            # beside it (`roofline.other_family`)
            other_family = None
            def _fam(sel_, fl_, by_, kn_, forms_):
                tm_ = sum(ms for _, ms in sel_) * 1e-3
                return {'kernel': kn_, 'launches_per_step': len(sel_), 'ms_per_step': tm_ * 1e3, 'avg_launch_ms': tm_ * 1e3 / max(len(sel_), 1),
                        'achieved_gbs': by_ / tm_ / 1e9 if tm_ else None, 'frac_hbm': by_ / tm_ / 1e9 / PEAK_HBM_GBS if tm_ else None,
                        'achieved_tflops': fl_ / tm_ / 1e12 if tm_ else None, 'frac_mfma': fl_ / tm_ / 1e12 / (PEAK_BF16_MFMA_TFLOPS / nprod) if tm_ else None,
                        'launches_by_entry_point': forms_}
            if fsel:
                fname = ('conv_bwd_fused_kernel<32> (input gradient AND weight gradient of a 3x3 conv 32 -> 32 in one launch: two-term fp16 '
                         'operands, 3 products per MAC on v_mfma_f32_16x16x32_f16, fp32 accumulate; the gy halo in LDS feeds both products; '
                         'the measured call includes its slab-reduce launch)')
                if sum(ms for _, ms in fsel) >= sum(ms for _, ms in sel):
                    other_family = _fam(sel, fl, roof_bytes, kname, forms)
                    sel, fl, roof_bytes, kname, forms = fsel, fflops, fbytes, fname, fforms
                else:
                    other_family = _fam(fsel, fflops, fbytes, fname, fforms)
        else:
            # dis_convg_run int args: (mode, ldx, xoff, ldy, yoff, n, hin, win, cin, cin_w, hout, wout, cout, cout_w,
            # k, stride, pad, act): every launch of the streaming kernel family convg_fwd_kernel<BN>; algorithmic
            # flops use the real channel counts and the spatial size of the strided side
            sel = [(ia, ms) for name, ia, ms in rec3 if name == 'dis_convg_run']
            bf_mode = args.dtype == 'bf16'
            if bf_mode:
                # dis_convb_run int args: (mode, x_bf16, ldx, xoff, y_bf16, ldy, yoff, n, hin, win, cin, cin_w, hout, wout, cout,
                # cout_w, k, stride, pad, act) -> the layout gflops() reads: drop the two storage flags
                sel = [((ia[0],) + ia[2:4] + ia[5:], ms) for name, ia, ms in rec3 if name == 'dis_convb_run']
            if ops.BF16X3 and not bf_mode:  # layers with >= 32 input channels run convg3_fwd_kernel (bf16x3): the dominant kernel
                def sliced(ia):  # ... except the 3x3 stride-1 layers csrc/conv2d.hip:dis_bx_slices_ok() sends to the
                    # halo-resident kernel as 32-channel slice launches (iconv1/2/3, conv1b: forward and input gradient)
                    mode, n, hin, win, cin, cout, k, stride = ia[0], ia[5], ia[6], ia[7], ia[8], ia[12], ia[14], ia[15]
                    pairs = ((cin + 31) // 32) * ((cout + 31) // 32) * (7 if k == 7 else 1)
                    return mode in (0, 1) and k in (3, 7) and stride == 1 and pairs <= 10 and n * hin * win >= 400000
                sel = [(ia, ms) for ia, ms in sel if ia[8] >= 32 and not sliced(ia)]

            def gflops(ia):
                mode, n, hin, win, cw, hout, wout, cow, k = ia[0], ia[5], ia[6], ia[7], ia[9], ia[10], ia[11], ia[13], ia[14]
                hw = hout * wout if mode in (0, 3) else hin * win
                return 2.0 * n * hw * cw * cow * k * k
            fl = sum(gflops(ia) for ia, _ in sel)
            kname = 'convg_fwd_kernel<BN> (fp32 MFMA 16x16x4, all conv / dgrad / transposed-conv launches)'
            if bf_mode:
                kname = ('convb_halo_kernel / convb_fwd_kernel (bf16 activations x bf16 weights on v_mfma_f32_16x16x32_bf16, fp32 '
                         'accumulate: all conv / dgrad / transposed-conv launches of DispNetS; LDS halo tiles for maps of >= 1024 '
                         'positions, per-tap streaming below; the times include the per-call weight packing)')
                peak, peak_note = PEAK_BF16_MFMA_TFLOPS, 'bf16 MFMA dense peak'
            elif ops.BF16X3:
                kname = ('convg3_fwd_kernel<BN> (fp32 conv as 6 bf16 products per MAC on v_mfma_f32_16x16x32_bf16; all '
                         'conv / dgrad / transposed-conv launches with >= 32 input channels; TFLOP/s are fp32-equivalent '
                         'algorithmic flops)')
                peak, peak_note = PEAK_BF16_MFMA_TFLOPS / 6.0, 'bf16 MFMA dense peak (2500 TFLOP/s) / 6 products per fp32 MAC'
        tm = sum(ms for _, ms in sel) * 1e-3
        if sel:
            ach = fl / tm / 1e12
            traffic, tsrc = None, None
            tpath = os.path.join(ROOT, 'profiles', 'roofline_traffic.json' if mf else f'roofline_traffic_sf_{args.dtype}.json')
            if os.path.exists(tpath):
                # HBM bytes per launch from the PMC passes (separate rocprofv3 runs, see profiles/README.md)
                tj = json.load(open(tpath))
                traffic, tsrc = tj['hbm_bytes_per_launch'], tj['source']
                if mf and ('conv_bwd_fused' in kname) != ('conv_bwd_fused' in tj.get('kernel', '')):
                    traffic, tsrc = None, None   # (the committed PMC summary is of the other kernel family)
            # both roofs of the kernel (a 32 -> 32 3x3 fp32 conv is 72 flop per algorithmic byte; the balance point of
            # the n-product form is (2500 / n) TFLOP/s / 8 TB/s = 104 flop/B for n = 3: below it the binding roof is HBM)
            frac_mfma = ach / peak
            ach_gbs = (roof_bytes / tm / 1e9) if roof_bytes else None
            frac_hbm = (ach_gbs / PEAK_HBM_GBS) if ach_gbs else None
            hbm_bound = frac_hbm is not None and frac_hbm >= frac_mfma
            step_traffic = None
            if os.path.exists(tpath) and tj.get('step_hbm_bytes'):
                alg = SURVEY_8D_STEP_BYTES if mf else SURVEY_8D_SF_STEP_BYTES[args.dtype]
                step_traffic = {'pmc_bytes_per_step': tj['step_hbm_bytes'], 'algorithmic_bytes_per_step': alg,
                                'algorithmic_note': SURVEY_8D_STEP_BYTES_NOTE if mf else SURVEY_8D_SF_STEP_BYTES_NOTE,
                                'ratio': tj['step_hbm_bytes'] / alg,
                                'source': tj['source'], 'ms_per_step_live': dt / args.steps * 1e3,
                                'achieved_gbs': tj['step_hbm_bytes'] / (dt / args.steps) / 1e9,
                                'frac_of_hbm_peak': tj['step_hbm_bytes'] / (dt / args.steps) / 1e9 / PEAK_HBM_GBS,
                                'algorithmic_frac_of_hbm_peak': alg / (dt / args.steps) / 1e9 / PEAK_HBM_GBS}
            roof = {'bound': 'hbm' if hbm_bound else 'mfma', 'kernel': kname,
                    'achieved': ach_gbs if hbm_bound else ach, 'peak': PEAK_HBM_GBS if hbm_bound else peak,
                    'unit': 'GB/s' if hbm_bound else 'TFLOP/s', 'frac': frac_hbm if hbm_bound else frac_mfma,
                    'frac_mfma': frac_mfma, 'achieved_tflops': ach, 'peak_tflops': peak, 'peak_note': peak_note,
                    'frac_hbm': frac_hbm, 'achieved_gbs': ach_gbs, 'peak_gbs': PEAK_HBM_GBS,
                    'flop_per_algorithmic_byte': (fl / roof_bytes) if roof_bytes else None,
                    'balance_flop_per_byte': peak * 1e12 / (PEAK_HBM_GBS * 1e9),
                    'achieved_over_fp32_mfma_peak': ach / PEAK_FP32_MFMA_TFLOPS,
                    'traffic': traffic, 'traffic_unit': 'bytes/launch',
                    'traffic_source': tsrc,
                    'algorithmic_bytes_per_launch_avg': (roof_bytes / len(sel)) if roof_bytes else None,
                    'algorithmic_bytes_note': 'x in + y out + every activation-sized tensor the fused prologue / epilogue reads '
                                              '(accumulate, act\', GroupNorm channel-sum operands), per launch',
                    'launches_per_step': len(sel), 'launches_by_entry_point': forms if mf else None,
                    'avg_launch_ms': tm * 1e3 / len(sel), 'flop_per_launch_avg': fl / len(sel),
                    'share_of_step_kernel_time': tm / (sum(ms for _, _, ms in rec3) * 1e-3),
                    'other_family': other_family if mf else None,
                    'step_traffic': step_traffic}
        # ---- HBM-bound kernels of the per-pixel path (north star: achieved GB/s of the warp / loss kernels): algorithmic
        # bytes (SURVEY.md section 8(d) per-pixel table; every tensor read / written once) / HIP-event time of every launch
        def hbm_row(label, names, nbytes, note):
            sel_ = [(ia, ms) for name, ia, ms in rec3 if name in names]
            if not sel_:
                return None
            by = sum(nbytes(ia) for ia, _ in sel_)
            ms_ = sum(ms for _, ms in sel_)
            return {'kernel': label, 'bound': 'hbm', 'launches_per_step': len(sel_), 'ms_per_step': round(ms_, 4),
                    'algorithmic_bytes_per_step': by, 'achieved': by / (ms_ * 1e-3) / 1e9, 'peak': PEAK_HBM_GBS,
                    'unit': 'GB/s', 'frac': by / (ms_ * 1e-3) / 1e9 / PEAK_HBM_GBS, 'bytes_model': note}
        hbm = []
        if mf:
            # int args (tl, bs, h, w, c): out (tl,bs,h,w,tl,c) written, feat rows read once per slot, flows 8 B per warped slot
            hbm.append(hbm_row('gather_warped_feat_fwd (feature warp, 4 slots / px)', ('dis_gather_warped_feat_fwd',),
                               lambda ia: ia[0] * ia[1] * ia[2] * ia[3] * ia[0] * (8.0 * ia[4] + 8.0),
                               '(8C+8) B per pixel and slot, C = 32'))
            hbm.append(hbm_row('gather_warped_feat_bwd_csr (feature warp backward)', ('dis_gather_warped_feat_bwd_csr',),
                               lambda ia: ia[0] * ia[1] * ia[2] * ia[3] * ia[0] * (8.0 * ia[4] + 8.0),
                               '(8C+8) B per pixel and slot, C = 32'))
        # dis_gn_apply (n, hw, c, act): x read, y written (+ residual read); dis_gn_apply_bwd: two passes over gy, y/x -> gx
        hbm.append(hbm_row('gn_apply (GroupNorm forward apply)', ('dis_gn_apply',),
                           lambda ia: 8.0 * ia[0] * ia[1] * ia[2], '8 B per element (residual reads not counted)'))
        hbm.append(hbm_row('gn_apply_bwd (GroupNorm backward: reduce + apply)', ('dis_gn_apply_bwd',),
                           lambda ia: 24.0 * ia[0] * ia[1] * ia[2], '24 B per element (two passes)'))
        # dis_geo_loss_fwd (bs, h, w): 36 B / px; bwd ~20 B / px
        hbm.append(hbm_row('geo_loss_fwd (reprojection + 4 warps + masks + masked mean, per direction)',
                           ('dis_geo_loss_fwd',), lambda ia: 36.0 * ia[0] * ia[1] * ia[2], '36 B per pixel'))
        hbm.append(hbm_row('geo_loss_bwd', ('dis_geo_loss_bwd',), lambda ia: 20.0 * ia[0] * ia[1] * ia[2], '20 B per pixel'))
        hbm.append(hbm_row('photometric_fwd (census 9x9)', ('dis_photometric_fwd',),
                           lambda ia: 12.0 * ia[0] * ia[2] * ia[3], '12 B per pixel (VALU-bound: 81 taps x rsqrt)'))
        hbm.append(hbm_row('photometric_bwd', ('dis_photometric_bwd',),
                           lambda ia: 16.0 * ia[0] * ia[2] * ia[3], '16 B per pixel (VALU-bound)'))
        # the round-4 census kernels: int args (s, n, h, w, block, type); s estimates share one target image
        hbm.append(hbm_row('census_fwd_multi (census 9x9, s estimates against one image)', ('dis_photometric_fwd_multi',),
                           lambda ia: (8.0 * ia[0] + 4.0) * ia[1] * ia[2] * ia[3],
                           '8 B per pixel and estimate + 4 B per pixel for the shared image (VALU-bound: 81 taps x (s + 1) rsqrt)'))
        hbm.append(hbm_row('census_bwd_multi', ('dis_photometric_bwd_multi',),
                           lambda ia: (12.0 * ia[0] + 4.0) * ia[1] * ia[2] * ia[3],
                           '12 B per pixel and estimate + 4 B per pixel for the shared image (VALU-bound)'))
        hbm.append(hbm_row('lcn_fwd', ('dis_lcn_fwd',), lambda ia: 12.0 * ia[0] * ia[1] * ia[2], '12 B per pixel'))
        hbm = [r for r in hbm if r is not None]
        top = sorted(per.items(), key=lambda kv: -kv[1][1])[:12]
        kernel_ms = {k: {'calls': v[0], 'ms': round(v[1], 3)} for k, v in top}
    cpu = None
    l1_ref = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        # BASELINE metric, second half ("disp L1 vs ref"): the HIP forward on the oracle's bs=1 inputs and initial parameters
        # against the disparity of the oracle's first (warm-up) step; north-star bar 1e-4 px for the fp32 path
        hip_out, hip_vals, hip_sets = hip_first_step(args.arch, settings, dev, args.dtype)
        cpu, first = cpu_baseline(args.arch, knn_sets=hip_sets)
        d_free = (hip_out.reshape(-1) - first['out'].float().reshape(-1)).abs()
        bar = 1e-4 if args.dtype == 'f32' else 0.05
        free = {'value': float(d_free.mean()), 'max': float(d_free.max())}
        l1_ref = {'unit': 'px', 'bar': bar,
                  'free_running': free,
                  'loss_terms_max_abs_diff_free_running': max(abs(a - b) for a, b in zip(hip_vals, first['vals']))}
        if 'out_forced' in first:
            # DIS-MF: the oracle's own top-9-of-36 depends on how THIS host's CPU BLAS rounds at exact key ties (DESIGN.md section
            # 4).  The comparison on the HIP path's neighbour sets is accepted as `value` only if every row whose set differs is
            # such a tie on the oracle's own keys (tie_check.non_tie_rows == 0); otherwise `value` is the free-running figure.
            d = (hip_out.reshape(-1) - first['out_forced'].float().reshape(-1)).abs()
            tc = first.get('tie_check') or {'pass': False, 'error': 'no tie check'}
            same = (first['out'].float().reshape(-1) == first['out_forced'].float().reshape(-1))
            forced = {'value': float(d.mean()), 'max': float(d.max())}
            l1_ref.update({'on_hip_neighbour_sets': forced, 'tie_check': tc,
                           # pixels whose ORACLE value does not depend on the tie-breaks (free-running == forced, bit for bit)
                           'pixels_unaffected_by_ties': {'fraction': float(same.float().mean()),
                                                         'free_running_l1': float(d_free[same].mean()) if bool(same.any()) else None,
                                                         'free_running_max': float(d_free[same].max()) if bool(same.any()) else None}})
            chosen, kind = (forced, 'on_hip_neighbour_sets (every differing Conv3D row is a key tie)') if tc.get('pass') else \
                           (free, 'free_running (tie check failed or unavailable)')
        else:
            chosen, kind = free, 'free_running'
        if mf and args.dtype == 'f32':
            # the same comparison on a scene without lattice symmetry (synth scene 'bumps': a non-planar surface): same seeds, same
            # initial parameters - FREE-RUNNING, then with its own tie analysis and on the HIP path's neighbour sets.  (Round-4 review
            # item 6 expected the free-running figure to pass there; it does not on a host whose CPU kernels round the oracle's keys
            # differently from the fixture host: ~2.5 % of the Conv3D rows have two candidates within 1e-5 relative of each other on
            # ANY smooth scene, and which one the oracle picks is the host's rounding.  The HIP selection equals the reference's
            # torch.topk on the committed fixtures, tests/test_step_gpu.py incl. mf_128_bumps.)
            from depthinspace_amd import synth as _synth
            from oracle import dis_oracle as O
            hip_b, _, sets_b = hip_first_step(args.arch, settings, dev, args.dtype, scene='bumps')
            ctx_b = O.StepContext(settings)
            tb_b = {k: torch.from_numpy(v) for k, v in _synth.make_batch(settings, 1, TL, seed=1234, scene='bumps').items()}
            nthr_b = torch.get_num_threads()
            torch.set_num_threads(int(os.environ.get('DIS_CPU_BASELINE_THREADS', str(min(nthr_b, 32)))))
            try:
                p0_b = O.init_params(O.mf_param_shapes(), seed=0)
                with torch.no_grad():
                    data_b = O.copy_data(ctx_b, tb_b)
                    O.CONV3D_TAP = []
                    try:
                        ref_b = O.mf_net_forward(ctx_b, p0_b, data_b, O.read_optical_flow(data_b, TL)).detach().float()
                    finally:
                        tap_b, O.CONV3D_TAP = O.CONV3D_TAP, None
                    O.CONV3D_FORCE = {'core': sets_b[0].long(), 'quarter': sets_b[1].long()}
                    try:
                        forced_b = O.mf_net_forward(ctx_b, p0_b, data_b, O.read_optical_flow(data_b, TL)).detach().float()
                    finally:
                        O.CONV3D_FORCE = None
            finally:
                torch.set_num_threads(nthr_b)
            d_b = (hip_b.reshape(-1) - ref_b.reshape(-1)).abs()
            d_bf = (hip_b.reshape(-1) - forced_b.reshape(-1)).abs()
            tc_b = knn_tie_check(tap_b, sets_b) if tap_b else {'pass': False, 'error': 'no tap'}
            l1_ref['bumps_free_running'] = {
                'value': float(d_b.mean()), 'max': float(d_b.max()), 'pass': float(d_b.mean()) < bar,
                'scene': "synth.make_batch(scene='bumps'): non-planar surface, batch seed 1234, init_params(seed=0), bs=1, 512x432",
                'on_hip_neighbour_sets': {'value': float(d_bf.mean()), 'max': float(d_bf.max()), 'pass': float(d_bf.mean()) < bar},
                'tie_check': {k: tc_b.get(k) for k in ('conv3d_rows', 'rows_whose_set_differs', 'non_tie_rows', 'ulps_needed',
                                                       'largest_relative_key_gap_of_a_differing_row',
                                                       'relative_gap_histogram_of_differing_rows', 'pass', 'error') if k in tc_b}}
        l1_ref.update({'value': chosen['value'], 'max': chosen['max'], 'value_kind': kind, 'pass': chosen['value'] < bar,
                       'sample': 'bs=1 (one 4-frame track, batch seed 1234, init_params(seed=0)), 512x432: free-running HIP forward '
                                 'vs the CPU oracle of cpu_baseline on this host'})
        fx_path = os.path.join(ROOT, 'tests', 'golden', 'mf_512x432_bs1.npz')
        if mf and args.dtype == 'f32' and os.path.exists(fx_path):
            # the REFERENCE's own output at the metric's size (tests/golden/mf_512x432_bs1.npz, written by oracle/make_golden.py from
            # the imported reference on the same sample: batch seed 1234, init_params(seed=0), epoch 2): nothing forced, no tie
            # argument - `value` is this comparison; the oracle-on-this-host figures above stay as secondary fields
            fx = np.load(fx_path)
            assert (int(fx['H']), int(fx['W']), int(fx['bs']), int(fx['pseed']), int(fx['bseed']), int(fx['epoch'])) == (H, W, 1, 0, 1234, 2)
            d_fx = (hip_out.reshape(-1) - torch.from_numpy(fx['out0']).float().reshape(-1)).abs()
            ids_eq = bool(np.array_equal(hip_sets[0].numpy(), fx['knn_idx_core']) and np.array_equal(hip_sets[1].numpy(), fx['knn_idx_quarter']))
            rows_diff = int((np.sort(hip_sets[0].numpy(), -1) != np.sort(fx['knn_idx_core'], -1)).any(-1).sum() +
                            (np.sort(hip_sets[1].numpy(), -1) != np.sort(fx['knn_idx_quarter'], -1)).any(-1).sum())
            vals_fx = [float(v) for v in fx['vals']]
            l1_ref['oracle_on_this_host'] = {k: l1_ref[k] for k in ('value', 'max', 'value_kind', 'pass', 'sample')}
            l1_ref.update({'value': float(d_fx.mean()), 'max': float(d_fx.max()), 'pass': float(d_fx.mean()) < bar,
                           'value_kind': 'free-running vs reference fixture (512x432)',
                           'ids_equal_reference': ids_eq, 'conv3d_rows_whose_set_differs_from_reference': rows_diff,
                           'loss_terms_max_abs_diff_vs_reference': max(abs(a - b) for a, b in zip(hip_vals, vals_fx)),
                           'sample': 'bs=1 (one 4-frame track, batch seed 1234, init_params(seed=0), epoch 2), 512x432: free-running HIP '
                                     'forward vs the output of the reference itself (tests/golden/mf_512x432_bs1.npz, '
                                     'oracle/make_golden.py case mf_512x432_bs1, 8 torch threads)'})

    extra_sf, strict_leg = None, None
    if rank == 0 and world == 1 and mf and not args.no_extra_legs and args.dtype == 'f32':
        # fresh child processes (their own GPU context), after everything of this run has been measured
        import subprocess
        torch.cuda.synchronize()

        def child(extra, env_extra=None):
            cmd = [sys.executable, os.path.abspath(__file__), '--steps', str(max(5, args.steps // 2)), '--warmup', '3',
                   '--no-cpu-baseline', '--no-extra-legs', '--no-eager-leg'] + extra
            out = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                                 env=dict(os.environ, **(env_extra or {})))
            try:
                return json.loads(out.stdout.strip().splitlines()[-1]), out
            except Exception as e:   # the headline line must not depend on the extra legs
                return {'error': f'{type(e).__name__}: {e}', 'rc': out.returncode, 'stderr_tail': out.stderr[-400:]}, out

        # (1) the headline step with >= 24-bit conv operands: every 3x3 conv on the three-term bf16 split (6 products per MAC)
        d, _ = child([], {'DIS_CONV_SPLIT': 'bf16x3'})
        if 'error' in d:
            strict_leg = d
        else:
            strict_leg = {'value': d['value'], 'unit': d['unit'], 'ms_per_step': d['ms_per_step'], 'steps': d['steps'],
                          'conv_arithmetic': 'DIS_CONV_SPLIT=bf16x3: three-term bf16 operand split, 6 products per MAC, >= 24-bit '
                                             'operands (error vs fp64 <= the exact-fp32 MFMA kernel), otherwise the headline step',
                          'roofline': {k: (d.get('roofline') or {}).get(k) for k in
                                       ('bound', 'frac', 'frac_mfma', 'frac_hbm', 'achieved_tflops', 'achieved_gbs',
                                        'launches_per_step', 'avg_launch_ms')},
                          'command': 'DIS_CONV_SPLIT=bf16x3 python bench.py'}
        # (2) BASELINE config 2 (DIS-SF bs=8): bf16 activation storage, and the fp32 parity path
        extra_sf = {}
        for tag, extra in (('bf16_activation_storage', ['--dtype', 'bf16']), ('fp32', [])):
            d, _ = child(['--arch', 'single_frame'] + extra)
            if 'error' in d:
                extra_sf[tag] = d
            else:
                extra_sf[tag] = {'metric': d['metric'], 'value': d['value'], 'unit': d['unit'], 'ms_per_step': d['ms_per_step'],
                                 'dtype': d['dtype'], 'roofline_frac': (d.get('roofline') or {}).get('frac'),
                                 'roofline_bound': (d.get('roofline') or {}).get('bound'),
                                 'roofline_traffic': (d.get('roofline') or {}).get('traffic'),
                                 'step_traffic': (d.get('roofline') or {}).get('step_traffic'),
                                 'command': 'python bench.py --arch single_frame' + (' --dtype bf16' if extra else '')}
    if rank == 0:
        frames = world * args.bs * TL * args.steps
        res = {
            'metric': ('DIS-MF train frames/sec bs=4 default-pattern' if mf else
                       f'DIS-SF train frames/sec bs={args.bs} default-pattern ({"bf16 activation storage" if args.dtype == "bf16" else "fp32"})'),
            'value': frames / dt, 'unit': 'frames/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': dt / args.steps * 1e3,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': args.dtype,
            'dtype_note': ('fp32 tensors and results; the 3x3 16/32-channel convs multiply 22-bit two-term fp16 operands (config.conv_arithmetic); the >= 24-bit form of the same step is `strict_fp32_frames_per_s`' if (args.dtype == 'f32' and mf) else None),
            'data': 'synthetic',
            'config': {'workload': (f'DIS-MF (FuseNet)' if mf else 'DIS-SF (DispNetS)') +
                                   f' training step, bs={args.bs} per GPU x 4 frames, 512x432, '
                                   f'default-pattern synthetic, fwd+losses+bwd+Adam, epoch>={args.epoch}',
                       'global_batch': world * args.bs, 'parallelism': f'dp{world}', 'hip_graph': bool(use_graph),
                       'backend': (args.backend if world > 1 else None),
                       'conv_arithmetic': (('fp32 results everywhere; the 3x3 stride-1 convs with 16 / 32 channels (fwd, '
                                            'dgrad, wgrad) run as ' +
                                            ('f16x2 (two-term fp16 operand split with power-of-two block scaling, 3 products, '
                                             'fp32 accumulate: 22-bit operands, error vs fp64 < 1e-6 of the largest entry; '
                                             'DIS_CONV_SPLIT=bf16x3 selects the three-term bf16 split, 6 products, >= 24 bits; '
                                             if lib.fn('dis_get_conv_split')() == 1 else
                                             'bf16x3 (3-way bf16 operand split, 6 products, fp32 accumulate: error vs fp64 <= '
                                             'the exact-fp32 MFMA kernel; ') +
                                            'tests/test_net_ops_gpu.py), all other convs on v_mfma_f32_16x16x4_f32' if mf else
                                            'fp32 results everywhere; forward / input-gradient convs with >= 32 input '
                                            'channels run as bf16x3 (3-way bf16 operand split, 6 products, fp32 accumulate), '
                                            'the others and the streaming weight gradients on v_mfma_f32_16x16x4_f32')
                                           if ops.BF16X3 else 'v_mfma_f32_16x16x4_f32 (exact fp32 FMA chains)')
                       if args.dtype == 'f32' else
                       ('bf16 ACTIVATION STORAGE (BASELINE config 2): nhwc feature maps in bf16, forward / input-gradient / '
                        'transposed convs = one bf16 x bf16 product per MAC on v_mfma_f32_16x16x32_bf16 with fp32 accumulation, '
                        'weight gradients likewise (one-pass slice-pair kernel with transposing LDS reads; the first layer and ragged '
                        'shapes on fp32 MFMA from the bf16 values); parameters, their gradients, disparities '
                        'and losses fp32.  Not the parity path: disparity L1 vs the fp32 oracle 0.01 px (tests/test_sf_bf16_gpu.py)')},
            'roofline': roof, 'roofline_hbm_kernels': hbm, 'cpu_baseline': cpu, 'disp_l1_vs_ref': l1_ref,
            'ranks_seen': ranks_seen, 'devices_seen': devices_seen, 'replicas_equal': replicas_equal,
            'multi_gpu_measured': ('this line' if world > 1 and args.backend == 'nccl' else
                                   'unmeasured (no SCALE record with N > 1 exists yet)' if world == 1 else
                                   'gloo plumbing run, not a measurement'),
            'strict_fp32_frames_per_s': strict_leg, 'extra_dis_sf': extra_sf,
            'loss_terms': losses,
            'eager_launch_frames_per_s': eager_fps, 'adam_steps_taken': adam_steps, 'step_mode': stepper.mode,
            'kernel_ms_one_eager_step': kernel_ms,
        }
        print(json.dumps(res))
    if world > 1:
        barrier()   # rank 0 ran the per-call roofline leg on its own: leave together
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
