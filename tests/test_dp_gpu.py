"""Data parallelism through the real entry-point classes on the GPU: 2 ranks (gloo; both on the one GPU of the test box, the
collective's transport is not what is tested) run Worker.train_step with trainer.FlatAdam's bucketed, overlapped all-reduce.
The reduced gradient must equal the sum of the two ranks' single-process gradients (DDP semantics: per-rank masked-mean
losses, mean of the gradients - SURVEY.md section 8(e)), buckets must be in flight before backward returns, replicas must
stay identical.  (The single-GPU hipGraph form of the step is compared with the eager form in test_graphed_step_matches_eager.)"""
import argparse
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _args(arch, bs):
    return argparse.Namespace(use_pseudo_gt=False, lcn_radius=5, track_length=4, data_type='synthetic', architecture=arch,
                              epochs=1, warmup_epochs=150, train_batch_size=bs, max_disp=128)


NSTEPS = 5


def _rank(rank, world, port, arch, q, logdir):
    import faulthandler
    log = open(os.path.join(logdir, f'rank{rank}.log'), 'w')
    os.dup2(log.fileno(), 2)   # C++ / HIP runtime messages of this rank
    faulthandler.enable(file=log)
    faulthandler.dump_traceback_later(150, file=log)   # a hung collective leaves its stack behind for the parent to show

    def mark(msg):
        log.write(msg + '\n')
        log.flush()
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    os.environ['RANK'], os.environ['WORLD_SIZE'], os.environ['LOCAL_RANK'] = str(rank), str(world), '0'
    try:
        from depthinspace_amd.trainer import FlatAdam, GraphedStep, init_distributed
        assert init_distributed('gloo') == (rank, world, 0)
        from depthinspace_amd import synth
        from depthinspace_amd.model import multi_frame_networks, multi_frame_worker, single_frame_worker, networks
        H = W = 64
        settings = synth.make_settings(H, W)
        torch.manual_seed(0)
        if arch == 'multi_frame':
            w = multi_frame_worker.Worker(_args(arch, 1), settings=settings)
            net = multi_frame_networks.FuseNet((H, W), settings.K, settings.baseline).cuda()
        else:
            w = single_frame_worker.Worker(_args(arch, 1), settings=settings)
            net = networks.DispDecoder(channels_in=2, max_disp=128, imsizes=w.imsizes).cuda()
        assert (w.rank, w.world_size, w.is_main) == (rank, world, rank == 0)
        w.build_losses()
        w.current_epoch = 2
        opt = FlatAdam(net.parameters(), lr=1e-4, bucket_mb=(0.25 if arch == 'multi_frame' else 16.0))
        assert opt.world_size == world and opt.overlap and len(opt.buckets) >= 3
        batches = [{k: torch.from_numpy(v) for k, v in synth.make_batch(settings, 1, 4, seed=1234 + r).items()}
                   for r in range(world)]

        def local_grads():
            opt.overlap = False   # plain single-process backward passes, no collective
            gs = []
            for r in range(world):
                w.copy_data(batches[r], device=w.train_device, requires_grad=False, train=True)
                opt.zero_grad()
                flow = w.read_optical_flow(True)
                sum(w.loss_forward(w.net_forward(net, flow), True, flow)).backward()
                gs.append(opt.flat_g.clone())
            opt.overlap = True
            return gs

        def close(a, b):
            scale = float(b.abs().max()) + 1e-20
            return float((a - b).abs().max()) / scale

        res = {}
        early = []
        for step in range(NSTEPS):
            mark(f'step {step}: local gradients')
            gs = local_grads()
            mark(f'step {step}: dp train_step')
            opt.zero_grad()
            # ---- the product's step; peek at the bucket state between backward and the optimiser
            orig = opt.step
            def spy(all_reduce=True):
                early.append(sum(opt._reduced))
                return orig(all_reduce)
            opt.step = spy
            w.train_step(net, opt, batches[rank])
            opt.step = orig
            torch.cuda.synchronize()
            res[f'grad_err{step}'] = close(opt.flat_g, gs[0] + gs[1])
            if res[f'grad_err{step}'] > 1e-5:   # diagnostics: which parameters carry the difference
                want = gs[0] + gs[1]
                names = [n for n, _ in net.named_parameters()]
                rows = []
                for n_, p_, off in zip(names, opt.params, opt.offsets):
                    sl = slice(off, off + p_.numel())
                    e = float((opt.flat_g[sl] - want[sl]).abs().max())
                    rows.append((e / (float(want.abs().max()) + 1e-20), n_, float(want[sl].abs().max())))
                rows.sort(reverse=True)
                res[f'worst{step}'] = rows[:6]
        res['early'] = early
        res['nbuckets'] = len(opt.buckets)
        # identical replicas
        chk = torch.stack([opt.flat_p.double().sum(), opt.flat_p.double().abs().sum()])
        both = [torch.zeros_like(chk) for _ in range(world)]
        dist.all_gather(both, chk)
        res['replicas_equal'] = bool(torch.equal(both[0], both[1]))
        res['steps'] = opt.step_count
        # trainer.GraphedStep with world_size > 1 runs this same eager, overlapped step on static buffers
        gstep = GraphedStep(w, net, opt, batches[rank], use_graph=True)
        assert not gstep.use_graph and gstep.mode == 'eager-overlap'
        gstep.run()
        torch.cuda.synchronize()
        res['graph_steps'] = opt.step_count
        mark('done')
        faulthandler.cancel_dump_traceback_later()
        q.put((rank, res))
    except Exception as e:
        import traceback
        q.put((rank, {'error': traceback.format_exc()}))
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


@pytest.mark.parametrize('arch', ['multi_frame', 'single_frame'])
def test_two_rank_train_step(arch, tmp_path):
    world = 2
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rank, args=(r, world, port, arch, q, str(tmp_path))) for r in range(world)]
    for p in procs:
        p.start()
    res = {}
    try:
        while len(res) < world:
            try:
                r, v = q.get(timeout=240)
            except Exception:
                logs = {r: open(os.path.join(str(tmp_path), f'rank{r}.log')).read() for r in range(world)}
                raise AssertionError('a rank hung or died:\n' + '\n'.join(f'--- rank {r}\n{t[-3000:]}' for r, t in logs.items()))
            res[r] = v
            if 'error' in v:  # the other rank may be blocked in a collective: do not wait for it
                break
    finally:
        for p in procs:
            p.join(timeout=5 if any('error' in v for v in res.values()) else 120)
            if p.is_alive():
                p.terminate()
    for v in res.values():
        assert 'error' not in v, v['error']
    for r in range(world):
        assert 'error' not in res[r], res[r]['error']
        for k_, v_ in sorted(res[r].items()):
            print(f'rank {r} {k_}: {v_}')
        # reduced gradient == sum of the ranks' own gradients (float atomics in two scatter kernels: tolerance, not bits; measured
        # <= 3e-7), on EVERY step.  The two ranks of this test time-share one GPU; in round 2 that sharing made an evaluation of
        # the DIS-MF step come out 1e-4..1e-3 off once in ~10 steps and the bar here was a median.  Round 3 found the cause - a
        # compiler-formed v_pk_add_f32 losing one half's addend in lanes 48..63 under contention (scripts/diag/share_repro.hip,
        # DESIGN.md section 4) - and the library is now built without packed fp32 instructions: the per-step bar is back.
        errs = [res[r][f'grad_err{step}'] for step in range(NSTEPS)]
        assert max(errs) < 1e-5, (r, errs, res[r])
        # step 0 learns the notification pattern; from step 1 on (almost) every bucket is in flight before backward returns
        assert res[r]['early'][0] == 0 and all(e >= res[r]['nbuckets'] - 1 for e in res[r]['early'][1:]), res[r]
        assert res[r]['replicas_equal'] and res[r]['steps'] == NSTEPS
        assert res[r]['graph_steps'] == NSTEPS + 1, res[r]
    print(arch, res[0])


@pytest.mark.parametrize('arch', ['multi_frame', 'single_frame'])
def test_graphed_step_matches_eager(arch):
    """trainer.GraphedStep (the object bench.py and Worker.train_epoch(use_graph=True) run): three replays of the captured step
    land on the same parameters as three eager steps from the same state, with Adam's device-side step counter advancing."""
    from depthinspace_amd import synth
    from depthinspace_amd.model import multi_frame_networks, multi_frame_worker, single_frame_worker, networks
    from depthinspace_amd.trainer import FlatAdam, GraphedStep
    H = W = 64
    settings = synth.make_settings(H, W)
    torch.manual_seed(0)
    if arch == 'multi_frame':
        w = multi_frame_worker.Worker(_args(arch, 1), settings=settings)
        net = multi_frame_networks.FuseNet((H, W), settings.K, settings.baseline).cuda()
    else:
        w = single_frame_worker.Worker(_args(arch, 1), settings=settings)
        net = networks.DispDecoder(channels_in=2, max_disp=128, imsizes=w.imsizes).cuda()
    w.build_losses()
    w.current_epoch = 2
    opt = FlatAdam(net.parameters(), lr=1e-4)
    batch = {k: torch.from_numpy(v) for k, v in synth.make_batch(settings, 1, 4, seed=77).items()}
    state = (opt.flat_p, opt.exp_avg, opt.exp_avg_sq, opt.state_dev)
    snap = [t.clone() for t in state]
    eager = GraphedStep(w, net, opt, batch, use_graph=False)
    for _ in range(3):
        eager.run()
    torch.cuda.synchronize()
    p_eager, l_eager = opt.flat_p.clone(), eager.losses()
    assert opt.step_count == 3
    graphed = GraphedStep(w, net, opt, batch, use_graph=True, warmup=1)
    graphed.run()   # eager warm-up step + capture + first replay
    torch.cuda.synchronize()
    assert graphed.mode == 'graph'
    for t, c in zip(state, snap):
        t.copy_(c)
    for _ in range(3):
        graphed.run()
    torch.cuda.synchronize()
    assert opt.step_count == 3
    # (two float-atomic scatters in the step: run-to-run rounding noise, amplified by Adam's normalisation of ~0 gradients)
    assert float((opt.flat_p - p_eager).abs().max()) < 2.5e-4
    assert float((opt.flat_p - p_eager).abs().mean()) < 2e-6
    np.testing.assert_allclose(graphed.losses(), l_eager, rtol=2e-3, atol=1e-5)
