"""Optimiser and data-parallel plumbing of the step: one flat fp32 parameter / gradient buffer, a fused Adam kernel over
it (dis_adam_step_dev: step counter on the device, so a captured step replays correctly) and the gradient all-reduce
over RCCL (torch.distributed 'nccl'), bucketed and overlapped with the backward pass.

The reference uses torch.optim.Adam(lr=1e-4) on a single GPU (train_val.py:55-56) and has no communication layer; data
parallelism here is batch sharding with mean-of-per-rank gradients: rank r owns its own tracks,
every rank holds full replicas of the parameters and of the Adam state.
"""
import os

import torch

from . import ops


def init_distributed(backend=None):
    """Join the process group torchrun describes (RANK / WORLD_SIZE / LOCAL_RANK / MASTER_*) and bind this process to its
    GPU.  Call it BEFORE the first HIP call of the process.  Returns (rank, world_size, local_rank); (0, 1, 0) and no process
    group when the process was not started by a launcher."""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29500')
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')  # the host driver only supports dmabuf IPC
        if torch.cuda.device_count() > 0:
            torch.cuda.set_device(local_rank % torch.cuda.device_count())
        if not torch.distributed.is_initialized():
            torch.distributed.init_process_group(backend or os.environ.get('DIS_DIST_BACKEND', 'nccl'), rank=rank,
                                                 world_size=world)
    return rank, world, local_rank


class AbiComm(object):
    """The C ABI's own gradient exchange (include/dis_hip.h: dis_allreduce_*; RCCL bound by the library itself) - what a host
    that is not PyTorch would use.  torch.distributed (any backend) only carries rank 0's 128-byte id to the other ranks.
    FlatAdam uses it instead of torch.distributed.all_reduce when DIS_ALLREDUCE=abi."""

    def __init__(self, rank=0, world_size=1, process_group=None, unique_id=None):
        import ctypes
        L = ops.lib
        self._ct = ctypes
        self.rank, self.world_size = rank, world_size
        if unique_id is None:
            buf = ctypes.create_string_buffer(128)
            if rank == 0:
                rc = L.fn('dis_allreduce_unique_id')(ctypes.cast(buf, ctypes.c_void_p))
                if rc != 0:
                    raise L.DisHipError(f'dis_allreduce_unique_id failed: {rc}')
            obj = [buf.raw]
            if world_size > 1:
                torch.distributed.broadcast_object_list(obj, src=0, group=process_group)
            unique_id = obj[0]
        assert len(unique_id) == 128
        self.unique_id = bytes(unique_id)
        idb = ctypes.create_string_buffer(self.unique_id, 128)
        h = ctypes.c_void_p()
        rc = L.fn('dis_allreduce_init')(ctypes.cast(ctypes.pointer(h), ctypes.c_void_p), ctypes.cast(idb, ctypes.c_void_p), world_size, rank)
        if rc != 0:
            raise L.DisHipError(f'dis_allreduce_init failed: {rc}')
        self.handle = h

    def all_reduce(self, t, average=False):
        """in place on a contiguous fp32 device tensor, asynchronous on the current stream"""
        assert t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()
        ops.lib.call('dis_allreduce_sum_f32', self.handle.value, t, t.numel(), 1 if average else 0)

    def close(self):
        if self.handle is not None and self.handle.value:
            ops.lib.fn('dis_allreduce_destroy')(self.handle)
            self.handle = None


class _Done(object):
    def wait(self):
        pass


class FlatAdam(object):
    """Adam with torch.optim.Adam's default hyper-parameters on a flattened view of `params`.

    All parameters (also the never-used `upconv1/2` of FuseNet, whose gradient stays zero: a zero gradient leaves an Adam
    parameter unchanged) are re-pointed into one contiguous buffer; gradients are accumulated by autograd - or written by
    the weight-gradient kernels themselves (ops._sink) - directly into the matching views of one flat gradient buffer.

    Data parallelism (world_size > 1): the flat gradient is cut into buckets of ~`bucket_mb` in REVERSE parameter order
    (the order the backward pass completes them).  Every finished parameter gradient notifies its bucket (ops._sink for the
    kernels that write the flat buffer, a post-accumulate hook for gradients autograd adds); a complete bucket is
    all-reduced at once on a communication stream while the backward pass goes on.  How many notifications complete a
    parameter is learnt in the first step (which reduces everything after backward, like overlap=False); a later step
    that notifies differently fails loudly.  step() waits for the collectives and runs the fused Adam kernel with
    grad_scale = 1/world_size (mean of the per-rank gradients).  Inside a hipGraph capture the hooks stay silent: the
    caller places all_reduce_grads() between captured segments (bench.py)."""

    def __init__(self, params, lr=1e-4, betas=(0.9, 0.999), eps=1e-8, world_size=None, process_group=None,
                 bucket_mb=16.0, overlap=True):
        self.params = [p for p in params]
        assert len(self.params) > 0
        dev = self.params[0].device
        n = sum(p.numel() for p in self.params)
        self.n = n
        pad = (n + 3) // 4 * 4
        self.flat_p = torch.zeros(pad, dtype=torch.float32, device=dev)
        self.flat_g = torch.zeros(pad, dtype=torch.float32, device=dev)
        self.exp_avg = torch.zeros(pad, dtype=torch.float32, device=dev)
        self.exp_avg_sq = torch.zeros(pad, dtype=torch.float32, device=dev)
        # {int steps taken, float 1-beta1^t, float sqrt(1-beta2^t), unused}: advanced by dis_adam_step_dev on the device
        self.state_dev = torch.zeros(4, dtype=torch.int32, device=dev)
        self.offsets = []
        off = 0
        with torch.no_grad():
            for p in self.params:
                k = p.numel()
                self.flat_p[off:off + k].copy_(p.detach().reshape(-1))
                p.data = self.flat_p[off:off + k].view(p.shape)
                p.grad = self.flat_g[off:off + k].view(p.shape)
                self.offsets.append(off)
                off += k
        self.lr, self.betas, self.eps = lr, betas, eps
        if world_size is None:
            world_size = torch.distributed.get_world_size(process_group) if torch.distributed.is_initialized() else 1
        self.world_size = world_size
        self.process_group = process_group
        self.overlap = bool(overlap) and world_size > 1
        # DIS_ALLREDUCE=abi: the exchange goes through the C ABI's dis_allreduce_* (the library's own RCCL binding) instead of
        # torch.distributed.all_reduce - same buckets, same streams
        self._abi = None
        if world_size > 1 and dev.type == 'cuda' and os.environ.get('DIS_ALLREDUCE', '') == 'abi':
            self._abi = AbiComm(torch.distributed.get_rank(process_group), world_size, process_group)
        if dev.type == 'cuda':
            ops.register_grad_sinks(self.params, self._notify if self.overlap else None)
        # ---- buckets (reverse parameter order)
        self.buckets = []          # [lo, hi) ranges of the flat buffer, in the order the backward pass finishes them
        self._bucket_of = {}       # param index -> bucket index
        if self.overlap:
            limit = int(bucket_mb * (1 << 20) / 4)
            hi = n
            cur = []
            for i in range(len(self.params) - 1, -1, -1):
                cur.append(i)
                lo = self.offsets[i]
                if hi - lo >= limit or i == 0:
                    for j in cur:
                        self._bucket_of[j] = len(self.buckets)
                    self.buckets.append((lo, hi))
                    hi, cur = lo, []
            self._index = {id(p): i for i, p in enumerate(self.params)}
            self._expected = None                      # notifications per parameter, learnt in the first step
            self._seen = [0] * len(self.params)
            self._works = []
            self._reduced = [False] * len(self.buckets)
            self._comm_stream = torch.cuda.Stream(device=dev) if dev.type == 'cuda' else None
            for p in self.params:
                p.register_post_accumulate_grad_hook(self._notify)

    # ------------------------------------------------------------------ gradient notifications -> bucketed all-reduce
    def _notify(self, p):
        if not self.overlap or (self.flat_g.is_cuda and torch.cuda.is_current_stream_capturing()):
            return
        i = self._index.get(id(p))
        if i is None:
            return
        self._seen[i] += 1
        if self._expected is None:
            return  # calibration step
        if self._seen[i] > self._expected[i]:
            raise RuntimeError(f'FlatAdam: parameter {i} received more gradient contributions than in the first step; '
                               f'the overlapped all-reduce needs a static graph (use overlap=False)')
        if self._seen[i] == self._expected[i]:
            b = self._bucket_of[i]
            self._pending[b] -= 1
            if self._pending[b] == 0:
                self._reduce_bucket(b)

    def _reduce_bucket(self, b):
        lo, hi = self.buckets[b]
        view = self.flat_g[lo:hi]
        if self._comm_stream is not None:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream())
            self._comm_stream.wait_event(ev)
            with torch.cuda.stream(self._comm_stream):
                if self._abi is not None:
                    self._abi.all_reduce(view)
                    w = _Done()
                else:
                    w = torch.distributed.all_reduce(view, group=self.process_group, async_op=True)
        else:
            w = torch.distributed.all_reduce(view, group=self.process_group, async_op=True)
        self._works.append(w)
        self._reduced[b] = True

    def _arm(self):
        if not self.overlap:
            return
        self._seen = [0] * len(self.params)
        self._works = []
        self._reduced = [False] * len(self.buckets)
        if self._expected is not None:
            self._pending = [0] * len(self.buckets)
            for i, e in enumerate(self._expected):
                if e > 0:
                    self._pending[self._bucket_of[i]] += 1

    def finish_grads(self):
        """every bucket reduced (SUM over ranks) and visible to the current stream"""
        if self.world_size <= 1:
            return
        if not self.overlap or (self.flat_g.is_cuda and torch.cuda.is_current_stream_capturing()):
            self.all_reduce_grads()
            return
        if self._expected is None:
            self._expected = list(self._seen)   # first step: learn, reduce everything now
            self.all_reduce_grads()
            return
        for i, (s, e) in enumerate(zip(self._seen, self._expected)):
            if s != e:
                raise RuntimeError(f'FlatAdam: parameter {i} got {s} gradient contributions, {e} in the first step')
        for b in range(len(self.buckets)):
            if not self._reduced[b]:      # buckets of parameters that never receive a gradient (upconv1/2): zeros
                self._reduce_bucket(b)
        if self._comm_stream is not None:
            with torch.cuda.stream(self._comm_stream):
                for w in self._works:
                    w.wait()
            torch.cuda.current_stream().wait_stream(self._comm_stream)
        else:
            for w in self._works:
                w.wait()
        self._works = []

    # ------------------------------------------------------------------ torch.optim-like surface
    def zero_grad(self, set_to_none=False):
        self.flat_g.zero_()
        ops.reset_grad_sinks()
        if self.flat_g.is_cuda:
            ops.begin_step(self.flat_g.device)
        self._arm()

    def all_reduce_grads(self):
        if self.world_size > 1:
            if self._abi is not None:
                self._abi.all_reduce(self.flat_g)
            else:
                torch.distributed.all_reduce(self.flat_g, group=self.process_group)

    @property
    def step_count(self):
        return int(self.state_dev[0])  # device -> host: only checkpointing / tests read it

    def step(self, all_reduce=True):
        # every GroupNorm token of this backward pass redeemed, every pre-reduced gradient picked up - checked BEFORE the update
        # (a stale table means gradients of this step are wrong; the next zero_grad() would be one update too late)
        ops.check_backward_complete()
        if all_reduce:
            self.finish_grads()
        ops.adam_step_dev(self.flat_p, self.flat_g, self.exp_avg, self.exp_avg_sq, self.state_dev, self.lr,
                          self.betas[0], self.betas[1], self.eps, 1.0 / self.world_size)

    def broadcast_parameters(self, src=0):
        """identical replicas: parameters, moments and the step counter of rank `src`"""
        if self.world_size > 1:
            for t in (self.flat_p, self.exp_avg, self.exp_avg_sq, self.state_dev):
                torch.distributed.broadcast(t, src, group=self.process_group)
            ops.params_changed()   # (weights packed by this step's batch launch are stale)

    # torch.optim.Adam's own (de)serialisation layout, so state.dict files are interchangeable with the reference's
    # (reference model/worker.py:342-364,376-402 saves optimizer.state_dict() of torch.optim.Adam)
    def state_dict(self):
        step = float(self.step_count)
        state = {}
        if step > 0:
            for i, (p, off) in enumerate(zip(self.params, self.offsets)):
                k = p.numel()
                state[i] = {'step': torch.tensor(step), 'exp_avg': self.exp_avg[off:off + k].view(p.shape).clone(),
                            'exp_avg_sq': self.exp_avg_sq[off:off + k].view(p.shape).clone()}
        group = {'lr': self.lr, 'betas': tuple(self.betas), 'eps': self.eps, 'weight_decay': 0, 'amsgrad': False,
                 'maximize': False, 'foreach': None, 'capturable': False, 'differentiable': False, 'fused': None,
                 'params': list(range(len(self.params)))}
        return {'state': state, 'param_groups': [group]}

    def load_state_dict(self, sd):
        if 'param_groups' not in sd:  # round-1 private layout: {'step', 'exp_avg', 'exp_avg_sq'} flat
            self._set_step(int(sd['step']))
            self.exp_avg[:self.n].copy_(sd['exp_avg'])
            self.exp_avg_sq[:self.n].copy_(sd['exp_avg_sq'])
            return
        groups = sd['param_groups']
        ids = [i for g in groups for i in g['params']]
        if len(ids) != len(self.params):
            raise ValueError(f'optimizer state has {len(ids)} parameters, this model has {len(self.params)}')
        g0 = groups[0]
        if g0.get('amsgrad') or g0.get('weight_decay', 0) or g0.get('maximize'):
            raise ValueError('only plain Adam (no amsgrad / weight decay / maximize) is supported')
        self.lr, self.betas, self.eps = float(g0['lr']), tuple(g0['betas']), float(g0['eps'])
        steps = set()
        self.exp_avg.zero_()
        self.exp_avg_sq.zero_()
        for pos, (p, off) in enumerate(zip(self.params, self.offsets)):
            st = sd['state'].get(ids[pos])
            if st is None:
                continue  # a parameter that never received a gradient (upconv1/2) has no state in torch.optim.Adam
            k = p.numel()
            if tuple(st['exp_avg'].shape) != tuple(p.shape):
                raise ValueError(f'optimizer state {ids[pos]}: shape {tuple(st["exp_avg"].shape)} != {tuple(p.shape)}')
            self.exp_avg[off:off + k].copy_(st['exp_avg'].reshape(-1))
            self.exp_avg_sq[off:off + k].copy_(st['exp_avg_sq'].reshape(-1))
            steps.add(int(float(st['step'])))
        if len(steps) > 1:
            raise ValueError(f'per-parameter step counts differ ({sorted(steps)}): not representable in the fused optimiser')
        self._set_step(steps.pop() if steps else 0)

    def _set_step(self, step):
        b1, b2 = self.betas
        st = torch.zeros(4, dtype=torch.int32)
        st[0] = int(step)
        st[1:3] = torch.tensor([1.0 - b1 ** step, (1.0 - b2 ** step) ** 0.5], dtype=torch.float32).view(torch.int32)
        self.state_dev.copy_(st)


class GraphedStep(object):
    """One training step of a stage worker (`copy_data` + LCN, net_forward, loss_forward, backward, gradient all-reduce,
    Adam: the body of the reference loop, model/worker.py:499-539) on static device buffers, eager or captured in hipGraphs
    and replayed per batch.  bench.py and Worker.train_epoch(use_graph=True) both run THIS object, so the benchmark times
    the product's own loop.

    use_graph=False: the eager step (what Worker.train_step does): with world_size > 1 FlatAdam's hooks all-reduce every
      gradient bucket on the communication stream while the backward pass is still running.
    use_graph=True, world_size == 1: one graph (forward, losses, backward, Adam).
    world_size > 1 always runs the eager form: it is the one whose collectives overlap the backward pass, and it costs
      nothing (at N = 1 the eager loop reaches the captured loop's frames/s, bench.py `eager_launch_frames_per_s`).
      Captured multi-rank forms (graph, all-reduce, Adam graph; and a two-graph backward with the tail bucket reduced
      underneath the second graph) were built in round 2 and REMOVED: DIS-MF's faulted on its second replay in the 2-rank
      test, and neither can be validated on RCCL with the single GPU of the development box.
    The set of loss terms may change with the epoch (epoch < 2 adds the L1 warm-up term, reference
    model/multi_frame_worker.py:160-165): key() changes and the step is re-captured.
    Batches are copied into static device buffers (run(batch)); `errs` of the last step are in `loss_buf[:nterms]`."""

    def __init__(self, worker, net, opt, example_batch, use_graph=True, warmup=2, strict=False):
        self.worker, self.net, self.opt = worker, net, opt
        self.strict = strict          # True: a failing capture raises instead of falling back to eager launches
        self.capture_error = None     # why the step runs eagerly although a graph was asked for
        self.dev = opt.flat_p.device
        self.static = {k: torch.as_tensor(v).to(self.dev).contiguous().clone() for k, v in example_batch.items()}
        self.loss_buf = torch.zeros(32, device=self.dev)
        self.nterms = 0
        self.world = opt.world_size
        self.use_graph = bool(use_graph) and self.world == 1
        self.warmup = warmup
        self._key = None
        self._graphs = None
        self.mode = 'eager-overlap' if (self.world > 1 and opt.overlap) else 'eager'

    # ---- pieces of the step
    def key(self):
        w = self.worker
        return (w.current_epoch < 2, w.current_epoch < w.warmup_epochs and w.data_type == 'real')

    def _forward_loss(self):
        w = self.worker
        w.copy_data(self.static, device=self.dev, requires_grad=False, train=True)
        self.opt.zero_grad()
        flow = w.read_optical_flow(train=True)
        out = w.net_forward(self.net, flow)
        errs = w.loss_forward(out, True, flow)
        if isinstance(errs, dict):
            errs = errs['errs']
        if not isinstance(errs, (list, tuple)):
            errs = [errs]
        self.nterms = len(errs)
        wv = getattr(errs, 'weighted', None)   # (ops.LossTerms: the terms as one vector, their sum as one autograd node)
        self.loss_buf[:len(errs)].copy_(wv.detach() if wv is not None else torch.stack([e.detach() for e in errs]))
        tot = getattr(errs, 'total', None)
        return tot if tot is not None else sum(errs)

    def _eager(self):
        self._forward_loss().backward()
        self.opt.step()

    def _capture(self):
        opt = self.opt
        # the warm-up steps (kernel attributes, allocator pools, autograd nodes on the capture-compatible stream) must not train:
        # parameters, moments and the step counter are put back afterwards, so a graphed run takes exactly the steps an eager one does
        state = (opt.flat_p, opt.exp_avg, opt.exp_avg_sq, opt.state_dev)
        snap = [t.clone() for t in state]
        side = torch.cuda.Stream(device=self.dev)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(max(1, self.warmup)):
                self._eager()
            for t, c in zip(state, snap):
                t.copy_(c)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        g1 = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g1):
            self._forward_loss().backward()
            opt.step(all_reduce=False)
        self._graphs, self.mode = (g1,), 'graph'

    # ---- public
    def run(self, batch=None):
        if batch is not None:
            for k, v in batch.items():
                self.static[k].copy_(torch.as_tensor(v), non_blocking=True)
        if not self.use_graph:
            self._eager()
            return
        if self._graphs is None or self._key != self.key():
            self._graphs, self._key = None, self.key()
            try:
                self._capture()
            except Exception as e:  # pragma: no cover
                if self.strict:
                    raise
                import logging
                self.capture_error = f'{type(e).__name__}: {e}'
                logging.warning(f'[GraphedStep] hipGraph capture failed ({self.capture_error}); running eagerly')
                self.use_graph, self._graphs = False, None
                self.mode = 'eager (capture failed)'
                torch.cuda.synchronize()
                self._eager()
                return
        self._graphs[0].replay()

    def losses(self):
        return [float(v) for v in self.loss_buf[:self.nterms].cpu()]
