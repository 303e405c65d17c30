"""N>1 data-parallel plumbing on CPU: world_size-2 `gloo` run of the flat-gradient all-reduce that the GPU path
performs over RCCL (depthinspace_amd/trainer.py FlatAdam.all_reduce_grads), plus the rank-sharded synthetic batches
(weak scaling: every rank owns its own tracks, seed 1234+rank).  No HIP kernel runs here."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from depthinspace_amd.trainer import FlatAdam
        torch.manual_seed(0)  # identical replicas on every rank, as bench.py does
        params = [torch.nn.Parameter(torch.randn(3, 5)), torch.nn.Parameter(torch.randn(7)),
                  torch.nn.Parameter(torch.randn(2, 2, 3, 3))]
        opt = FlatAdam(params, lr=1e-4, world_size=world)
        # parameters and gradients are views of the two flat buffers
        assert params[0].data_ptr() == opt.flat_p.data_ptr()
        assert params[1].grad.data_ptr() == opt.flat_g[15:].data_ptr()
        opt.zero_grad()
        g = torch.Generator().manual_seed(100 + rank)
        local = [torch.randn(p.shape, generator=g) for p in params]
        for p, l in zip(params, local):
            p.grad.add_(l)  # autograd accumulates into the flat views the same way
        opt.all_reduce_grads()
        # expected: sum over ranks (the 1/world factor is applied inside the Adam kernel as grad_scale)
        exp = []
        for r in range(world):
            gr = torch.Generator().manual_seed(100 + r)
            exp.append(torch.cat([torch.randn(p.shape, generator=gr).reshape(-1) for p in params]))
        exp = sum(exp)
        ok = bool(torch.allclose(opt.flat_g[:opt.n], exp, rtol=0, atol=1e-6))
        # the optimiser step itself is HIP-only: on CPU tensors it must fail loudly, not fall back
        try:
            opt.step(all_reduce=False)
            loud = False
        except RuntimeError:
            loud = True
        q.put((rank, ok, loud))
    finally:
        dist.destroy_process_group()


def test_flat_gradient_allreduce_gloo_world2():
    world = 2
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(r[0] for r in res) == [0, 1]
    assert all(r[1] for r in res), res
    assert all(r[2] for r in res), res


def _bucket_worker(rank, world, port, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from depthinspace_amd.trainer import FlatAdam
        torch.manual_seed(0)
        def make():
            return torch.nn.Sequential(torch.nn.Linear(40, 64), torch.nn.Tanh(), torch.nn.Linear(64, 64), torch.nn.Tanh(),
                                       torch.nn.Linear(64, 64), torch.nn.Tanh(), torch.nn.Linear(64, 8))
        net = make()
        unused = torch.nn.Parameter(torch.randn(11))  # a parameter that never gets a gradient (FuseNet.upconv1/2)
        params = list(net.parameters()) + [unused]
        opt = FlatAdam(params, lr=1e-4, bucket_mb=0.008)   # ~2k floats per bucket: several buckets
        assert opt.overlap and len(opt.buckets) >= 3
        assert sorted(lo for lo, _ in opt.buckets)[0] == 0 and max(hi for _, hi in opt.buckets) == opt.n
        ok = True
        launched_early = []
        for step in range(4):
            xs = [torch.randn(16, 40, generator=torch.Generator().manual_seed(1000 * step + r)) for r in range(world)]
            opt.zero_grad()
            net(xs[rank]).pow(2).sum().backward()
            launched_early.append(sum(opt._reduced))   # buckets already in flight when backward returns
            opt.finish_grads()
            got = opt.flat_g[:opt.n].clone()
            exp = torch.zeros(opt.n)
            for r in range(world):   # what every rank computed, recomputed locally on a detached copy
                net2 = make()
                net2.load_state_dict(net.state_dict())
                net2(xs[r]).pow(2).sum().backward()
                exp[:opt.n - 11] += torch.cat([p.grad.reshape(-1) for p in net2.parameters()])
            ok = ok and bool(torch.allclose(got, exp, rtol=1e-5, atol=1e-6))
        # step 0 learns the notification counts (nothing launched during backward); later steps launch during backward
        ok = ok and launched_early[0] == 0 and all(n >= len(opt.buckets) - 1 for n in launched_early[1:])
        # a step that produces different notifications must fail loudly
        opt.zero_grad()
        loud = False
        try:
            (net(xs[rank]).pow(2).sum() + net(xs[rank]).sum()).backward()  # every parameter used twice -> still one hook call
            net[0].weight.grad.add_(1.0)
            opt._notify(net[0].weight)  # an extra contribution
        except RuntimeError:
            loud = True
        q.put((rank, ok, loud, launched_early))
    except Exception as e:  # report instead of leaving the parent waiting for the queue
        q.put((rank, False, False, repr(e)))
        raise
    finally:
        dist.destroy_process_group()


def test_bucketed_overlapped_allreduce_gloo_world2():
    """FlatAdam's bucketed all-reduce driven by gradient-ready notifications (what overlaps the collectives with the backward
    pass on the GPU path): buckets in reverse parameter order, launched while backward is still running, summed over ranks
    exactly like one flat all-reduce, parameters without a gradient handled, loud on a changed graph."""
    world = 2
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_bucket_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(r[1] for r in res), res
    assert all(r[2] for r in res), res


def test_shard_sampler_partitions_the_tracks():
    from depthinspace_amd.model.worker import ShardSampler, split_paths
    for n, world in ((64, 8), (37, 4), (5, 2)):
        for shuffle in (True, False):
            parts = [list(ShardSampler(n, r, world, shuffle, seed=42, epoch=3)) for r in range(world)]
            flat = [i for p in parts for i in p]
            assert len(set(flat)) == len(flat)  # disjoint
            if shuffle:
                assert len({len(p) for p in parts}) == 1 and len(flat) == n // world * world  # equal counts (collectives)
            else:
                assert sorted(flat) == list(range(n))  # evaluation covers every track exactly once
        assert list(ShardSampler(n, 0, world, True, seed=42, epoch=3)) != list(ShardSampler(n, 0, world, True, seed=42, epoch=4))
    paths = [f'{i:08d}' for i in range(9216)]
    tr, te, va = split_paths(paths, 'synthetic')   # reference model/worker.py:170-173
    assert (len(tr), len(te), len(va)) == (8192, 512, 512) and te[0] == '00000512' and tr[0] == '00001024'
    tr, te, va = split_paths(paths[:80], 'real')
    assert te == paths[4:80:8] and len(tr) == 70


def test_rank_sharded_batches_differ():
    from depthinspace_amd import synth
    s = synth.make_settings(32, 32)
    b0 = synth.make_batch(s, 1, 4, seed=1234 + 0)
    b1 = synth.make_batch(s, 1, 4, seed=1234 + 1)
    assert b0['im0'].shape == b1['im0'].shape == (1, 4, 1, 32, 32)
    assert float(np.abs(b0['R'] - b1['R']).max()) > 0
    assert float(np.abs(b0['im0'] - b1['im0']).max()) > 0


def _metric_worker(rank, world, port, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from depthinspace_amd.co import metric
        m = metric.MultipleMetric(metric.DistanceMetric(vec_length=1),
                                  metric.OutlierFractionMetric(vec_length=1, thresholds=[0.1, 0.5, 1]))
        if rank == 0:  # rank 1's test shard is EMPTY (fewer test tracks than ranks): it must still join the collectives
            g = torch.Generator().manual_seed(3)
            m.add(torch.randn(50, 1, generator=g), torch.zeros(50, 1))
            m.add(torch.randn(30, 1, generator=g), torch.zeros(30, 1))
        q.put((rank, m.get()))
    finally:
        dist.destroy_process_group()


def test_metrics_merge_with_an_empty_rank_gloo_world2():
    """callback_test_stop on a rank that never added a sample used to raise (torch.cat([]), counts None) while
    its peers blocked in all_gather: every rank must return the metrics of the whole set."""
    world = 2
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_metric_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=180) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    from depthinspace_amd.co import metric
    g = torch.Generator().manual_seed(3)
    d = torch.cat([torch.randn(50, 1, generator=g), torch.randn(30, 1, generator=g)])
    ref = metric.MultipleMetric(metric.DistanceMetric(vec_length=1),
                                metric.OutlierFractionMetric(vec_length=1, thresholds=[0.1, 0.5, 1]))
    ref.add(d, torch.zeros(80, 1))
    want = ref.get()
    for r in range(world):
        assert res[r].keys() == want.keys()
        for k in want:
            assert abs(res[r][k] - want[k]) < 1e-12, (r, k)
    # nothing anywhere: NaNs, not an exception
    e = metric.MultipleMetric(metric.DistanceMetric(vec_length=1), metric.OutlierFractionMetric(vec_length=1, thresholds=[1]))
    assert all(np.isnan(v) for v in e.get().values())
