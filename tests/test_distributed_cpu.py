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


def test_rank_sharded_batches_differ():
    from depthinspace_amd import synth
    s = synth.make_settings(32, 32)
    b0 = synth.make_batch(s, 1, 4, seed=1234 + 0)
    b1 = synth.make_batch(s, 1, 4, seed=1234 + 1)
    assert b0['im0'].shape == b1['im0'].shape == (1, 4, 1, 32, 32)
    assert float(np.abs(b0['R'] - b1['R']).max()) > 0
    assert float(np.abs(b0['im0'] - b1['im0']).max()) > 0
