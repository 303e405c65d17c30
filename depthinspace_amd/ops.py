"""torch.autograd wrappers over the C ABI (include/dis_hip.h).

Each Function allocates outputs/workspaces with torch (the caller owns all memory), launches the HIP
kernels on the current stream through `lib.call`, and wires the hand-written backward kernels.
No ATen compute op is used for the arithmetic of the step; torch is memory + autograd tape only.
"""
import os as _os_env
import torch
from . import lib

ACT_NONE, ACT_SELU, ACT_RELU = 0, 1, 2
CONV_ACCUM = 0x100
PHOTO_TYPES = {'mse': 0, 'sad': 1, 'census_mse': 2, 'census_sad': 3}


def _chk(*ts):
    for t in ts:
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError('depthinspace_amd ops need CUDA(HIP) tensors: the HIP path is the only path')
        if t.dtype != torch.float32:
            raise RuntimeError(f'expected float32, got {t.dtype}')
        if not t.is_contiguous():
            raise RuntimeError('expected a contiguous tensor')


def _c(t):
    return t if t.is_contiguous() else t.contiguous()


# Zeroed fp64 accumulators (GroupNorm statistics, loss sums) come from one arena per device that is cleared ONCE per
# step (begin_step, called by FlatAdam.zero_grad) instead of one tiny fill kernel per accumulator (~100 per step).
# A slice is handed out at most once between two clears; without begin_step (tests, inference) the arena simply runs
# out and plain torch.zeros takes over.
_ARENA = {}
_ARENA_DOUBLES = int(_os_env.environ.get('DIS_ARENA_DOUBLES', 3 << 22))   # 96 MiB (64 MiB sent 7 requests per DIS-MF step to torch.zeros): the per-workgroup channel-sum slots of the GroupNorm backward are 2 MiB per launch, ~30 per step


# gradient tensors whose producer already left the GroupNorm-backward sums (see _Conv2d.backward / _GroupNorm.backward):
# data_ptr -> (ab, slots)
_GN_PRE = {}
# GroupNorm backward applied ON LOAD by the input-gradient launch of the conv in front of the GroupNorm (round 5,
# dis_conv2d_dgrad_f16x2_gnb): the GroupNorm's backward runs only the coefficient kernel and returns a TOKEN in place of the gradient
# wrt its input - a one-element tensor expanded to the gradient's shape (no memory, an address no other tensor has); the conv's
# backward finds (g, q, coef, in_act) under the token's address.  token data_ptr -> (token, g, q, coef, in_act).  Only conv outputs
# that conv2d() / conv2d_gn_in() marked `_gn_lazy_ok` (their backward understands tokens, and the GroupNorm is their only consumer)
# are answered with one; begin_step() raises if a token was never redeemed.
_GN_LAZY = {}
GN_LAZY = _os_env.environ.get('DIS_GN_LAZY', '1') != '0'


# Forward twin of the tokens (round 5, dis_conv2d_fwd_f16x2_gnres): group_norm(..., residual, act=SELU, defer=True) does NOT run its
# elementwise pass; the output tensor exists but is unwritten, `_pending_gn` = (x2, stats, gamma, beta, residual, eps) hangs on it, and
# the 3x3 conv that consumes it first - conv2d() looks for the attribute - forms the values on load and stores them as a side
# output.  Every other reader must come after that conv (a ResNetBlock's own residual use does) or call realize() first;
# begin_step() raises if an output was never written.
_GN_PENDING = {}


def realize(t):
    """run the deferred GroupNorm pass of `t` now (a consumer without an on-load form)"""
    pend = getattr(t, '_pending_gn', None)
    if pend is not None:
        x2, stats, gamma, beta, res, eps = pend
        n, c = x2.shape[0], x2.shape[-1]
        lib.call('dis_gn_apply', x2, stats, gamma, beta, res, t, n, x2.numel() // (n * c), c, ACT_SELU, float(eps))
        t._pending_gn = None
        _GN_PENDING.pop(t.data_ptr(), None)
    return t


# Tokens come from a persistent per-device pool filled with NaN ONCE (unique addresses, no launch per step): arithmetic on a token
# that was not redeemed - autograd summing it with the gradient of a second consumer of the conv output - poisons the loss of the
# same step visibly.  _GN_TOKEN_FOR: uid of the conv node whose output a token stands for -> the token's address; that conv's
# backward raises if what it receives is not the token (see _gn_lazy_pop).
_TOKEN_POOL = {}
_TOKEN_POOL_SIZE = 4096
_GN_TOKEN_FOR = {}
import itertools as _it
_LAZY_UID = _it.count(1)


def _new_token(dev):
    pool = _TOKEN_POOL.get(dev)
    if pool is None:
        pool = _TOKEN_POOL[dev] = [torch.full((_TOKEN_POOL_SIZE,), float('nan'), dtype=torch.float32, device=dev), 0]
    if pool[1] >= _TOKEN_POOL_SIZE:   # (no begin_step between many backward passes: tests, ad-hoc use)
        return torch.full((1,), float('nan'), dtype=torch.float32, device=dev)
    i = pool[1]
    pool[1] = i + 1
    return pool[0][i:i + 1]


def _mark_lazy_ok(out):
    """`out` = (y, stats) of a conv node whose backward redeems GroupNorm tokens: a GroupNorm that is y's ONLY consumer may answer
    with one.  The mark is the node's uid (truthy); the node keeps it (the ctx of a torch.autograd.Function IS the output's grad_fn)."""
    uid = next(_LAZY_UID)
    out[0]._gn_lazy_ok = uid
    if out[0].grad_fn is not None:
        out[0].grad_fn.lazy_uid = uid


def _gn_lazy_defer(g, q, stats, gamma, ab, slots, gg, gb, n, hw, c, eps, in_act, uid=0):
    coef = torch.empty(n * (c + 2) + 4 * n * c + 2, dtype=torch.float32, device=g.device)
    lib.call('dis_gn_bwd_coef', stats, gamma, ab, slots, coef, gg, gb, _zeros_d(1, g.device), n, hw, c, eps)
    tok = _new_token(g.device)
    _GN_LAZY[tok.data_ptr()] = (tok, g, q, coef, in_act)
    if uid and uid is not True:
        _GN_TOKEN_FOR[uid] = tok.data_ptr()
    return tok.expand(g.shape)


def _gn_lazy_pop(gy, ctx=None):
    ent = _GN_LAZY.pop(gy.data_ptr(), None) if _GN_LAZY else None
    if _GN_TOKEN_FOR:
        exp = _GN_TOKEN_FOR.pop(getattr(ctx, 'lazy_uid', None), None)
        if exp is not None and (ent is None or ent[0].data_ptr() != exp):
            if ent is not None:
                _GN_LAZY[gy.data_ptr()] = ent
            raise RuntimeError('ops: GroupNorm token was summed with another gradient: the conv output has a second consumer '
                               '(only an output whose ONLY consumer is the GroupNorm may be answered with a token)')
    return ent


def check_forward_complete():
    """no deferred GroupNorm output (group_norm(defer=True)) is left unwritten: called at the end of FuseNet's forward"""
    if _GN_PENDING:
        _GN_PENDING.clear()
        raise RuntimeError('ops: a deferred GroupNorm output (group_norm(defer=True)) of this forward pass was never realised')


def check_backward_complete():
    """every token redeemed, every pre-reduced gradient picked up: called by FlatAdam.step() BEFORE the update (and by tests)"""
    if _GN_LAZY or _GN_TOKEN_FOR:
        _GN_LAZY.clear()
        _GN_TOKEN_FOR.clear()
        raise RuntimeError('ops: a deferred GroupNorm backward (token gradient) of this backward pass was not redeemed')
    if _GN_PRE:
        _GN_PRE.clear()
        raise RuntimeError('ops: a pre-reduced GroupNorm gradient of this backward pass was not consumed')


def _gn_lazy_materialize(ent):
    """the elementwise pass as a launch of its own (the consumer has no on-load form for this configuration)"""
    _, g, q, coef, in_act = ent
    n, c = g.shape[0], g.shape[-1]
    gx = torch.empty_like(g)
    lib.call('dis_gn_bwd_apply_coef', g, q, coef, gx, n, g.numel() // (n * c), c, in_act)
    return gx


def _gn_lazy_shape(cin, cout, k, stride, pad):
    return GN_LAZY and BF16X3 and cin == cout and cin in (16, 32) and k == 3 and stride == 1 and pad == 1


def _gn_lazy_k4s2(cin_pad, cin, cout, k, stride, pad):
    """FuseNet's 4 x 4 stride-2 down convolution (dis_conv2d_wgrad_k4s2_f16x2_gnb; DIS_F2_WGRAD_K4S2=0 keeps it on the fp32 kernel)"""
    return (GN_LAZY and BF16X3 and cin_pad == cin == 32 and cout == 32 and (k, stride, pad) == (4, 2, 1) and
            _os_env.environ.get('DIS_F2_WGRAD_K4S2', '1') != '0')


def begin_step(dev):
    if _GN_PRE:   # a gradient that was handed on pre-multiplied was never picked up by its GroupNorm: the step would be wrong
        _GN_PRE.clear()
        raise RuntimeError('ops: a pre-reduced GroupNorm gradient of the previous backward pass was not consumed')
    if _GN_PENDING:  # a deferred GroupNorm output was never written
        _GN_PENDING.clear()
        raise RuntimeError('ops: a deferred GroupNorm output (group_norm(defer=True)) of the previous forward pass was never realised')
    if _GN_LAZY or _GN_TOKEN_FOR:  # a token stood in for a gradient and nobody redeemed it (the conv output had a second consumer?)
        _GN_LAZY.clear()
        _GN_TOKEN_FOR.clear()
        raise RuntimeError('ops: a deferred GroupNorm backward (token gradient) of the previous backward pass was not redeemed')
    dev = torch.device(dev)
    if dev in _TOKEN_POOL:
        _TOKEN_POOL[dev][1] = 0
    a = _ARENA.get(dev)
    if a is None:
        a = _ARENA[dev] = [torch.zeros(_ARENA_DOUBLES, dtype=torch.float64, device=dev), 0, 0]
    elif a[2] > 0:
        a[0][:a[2]].zero_()   # only what was handed out since the last clear (the rest is still zero)
    a[1] = 0
    a[2] = 0
    _PackBatch.begin_step(dev)


class _PackBatch(object):
    """One weight-packing launch per training step for the bf16 DispNetS path (dis_convb_pack_record / dis_convb_pack_batch).
    Every dis_convb_run call packs its fp32 weights into its `wpack` workspace first (~100 launches of 5 - 7 us per step).  The packed
    form depends on the weights alone, and those change once per step, so: the call sites keep PERSISTENT workspaces (keyed by
    weight address and call geometry); the first begin_step() outside a graph capture starts a record, the next one stops it, uploads
    the descriptors and from then on every begin_step() (FlatAdam.zero_grad, i.e. in front of the forward pass) packs ALL recorded
    calls in one launch; the calls then pass DIS_CONVB_PREPACKED.  Calls that were not in the record (another network, inference
    without begin_step) keep packing for themselves.  DIS_PACK_BATCH=0 disables it.

    Lifetime rules (the descriptor table holds RAW device pointers of the weights and of the workspaces):
      * every recorded entry holds a weak reference to its weight tensor; when one dies (the network was freed) the table is dropped
        before the next launch and recording starts over (`invalidate`); a weight that was a temporary copy while recording (it
        died although its network lives) switches the batching off for the process - its pointer can never be replayed;
      * a recorded workspace is never freed while the table exists: a size / device change of a recorded entry drops the table and
        parks the old buffer in `retired` (a captured hipGraph may still replay the launch that writes it);
      * an entry counts as prepacked only while its weight is the recorded tensor at the recorded `_version` (torch-side writes:
        copy_, load_state_dict, broadcast) and no parameter update ran since the batch launch (`params_changed`, called by the
        Adam entry points and by every FlatAdam method that rewrites the parameters);
      * entries that are not part of the table are evicted when the cache passes 4 * CAP entries."""
    enabled = _os_env.environ.get('DIS_PACK_BATCH', '1') != '0'
    CAP = 1024
    cache = {}        # key -> [wpack tensor, in_table, weakref(weight), weight._version at the last batch launch]
    state = 'idle'    # idle -> recording -> ready   (off: more calls than the table holds / an unstable weight pointer)
    used = None       # keys seen while recording
    host = None
    table = None
    count = 0
    blocks = 0
    step = 0
    packed_step = -1  # the step whose begin_step ran the batch launch
    dead = False      # a recorded weight tensor was freed
    retired = []      # workspaces the dropped tables wrote into

    @classmethod
    def invalidate(cls, off=False):
        """drop the descriptor table: nothing launches through its pointers again (outside graphs captured earlier, whose
        buffers stay parked in `retired`)"""
        for k, ent in list(cls.cache.items()):
            if ent[1]:
                cls.retired.append(ent[0])
                del cls.cache[k]
        del cls.retired[:-4 * cls.CAP]
        cls.table, cls.count, cls.blocks, cls.used = None, 0, 0, None
        if cls.state == 'recording':   # stop the library's recorder
            import ctypes
            cnt = (ctypes.c_int * 2)(0, 0)
            lib.fn('dis_convb_pack_record')(None, 0, ctypes.cast(cnt, ctypes.c_void_p))
        cls.state = 'off' if off else 'idle'
        cls.packed_step = -1
        cls.dead = False

    @classmethod
    def _on_dead(cls, key):
        def cb(_ref):
            ent = cls.cache.get(key)
            if ent is None or ent[2] is not _ref:   # (the entry was dropped or re-bound to another tensor since)
                return
            if ent[1] or (cls.used is not None and key in cls.used):
                cls.dead = True     # handled at the next begin_step / workspace call (no library calls from a finalizer)
            else:
                cls.cache.pop(key, None)
        return cb

    @classmethod
    def begin_step(cls, dev):
        cls.step += 1
        if not cls.enabled:
            return
        capturing = torch.cuda.is_current_stream_capturing()
        if cls.dead and not capturing:
            # recorded while the weight was a temporary (died during the record): never batch; died later (its network went away): start over
            cls.invalidate(off=cls.state == 'recording')
        if cls.state == 'recording' and not capturing:
            import ctypes
            cnt = (ctypes.c_int * 2)(0, 0)
            rc = lib.fn('dis_convb_pack_record')(None, 0, ctypes.cast(cnt, ctypes.c_void_p))
            n, cls.blocks = cnt[0], cnt[1]
            if rc == 0 and 0 < n <= cls.CAP:
                nbytes = n * lib.fn('dis_convb_pack_desc_bytes')()
                cls.table = torch.frombuffer(bytearray(cls.host.raw[:nbytes]), dtype=torch.uint8).to(dev)
                cls.count = n
                for k in cls.used:
                    cls.cache[k][1] = True
                cls.state = 'ready'
            else:
                cls.state = 'idle' if n == 0 else 'off'   # (no bf16 convolution ran / more calls than the table holds)
            cls.used = None
        if cls.state == 'ready' and not cls.dead:   # (dead while capturing: the table points at freed weights - no launch, every call packs for itself)
            lib.call('dis_convb_pack_batch', cls.table, cls.count, cls.blocks)
            cls.packed_step = cls.step
            for ent in cls.cache.values():
                if ent[1]:
                    w = ent[2]()
                    ent[3] = w._version if w is not None else -1
        elif cls.state == 'idle' and not capturing:
            import ctypes
            if cls.host is None:
                cls.host = ctypes.create_string_buffer(cls.CAP * lib.fn('dis_convb_pack_desc_bytes')())
            if lib.fn('dis_convb_pack_record')(ctypes.cast(cls.host, ctypes.c_void_p), cls.CAP, None) == 0:
                cls.state, cls.used = 'recording', set()

    @classmethod
    def workspace(cls, key, words, dev, w=None):
        """-> (wpack tensor, prepacked): the call's persistent workspace and whether this step's batch launch has filled it from
        the weight tensor `w` as it is now"""
        if not cls.enabled:
            return torch.empty(words, dtype=torch.int16, device=dev), False
        ent = cls.cache.get(key)
        if ent is not None and (ent[0].numel() != words or ent[0].device != dev):
            if ent[1] or (cls.used is not None and key in cls.used):
                cls.invalidate()        # (the table writes into the old buffer: drop the table, park the buffer)
            cls.cache.pop(key, None)
            ent = None
        if ent is None:
            if len(cls.cache) >= 4 * cls.CAP:
                for k in [k for k, e in cls.cache.items() if not e[1] and not (cls.used is not None and k in cls.used)]:
                    del cls.cache[k]
            import weakref
            ent = cls.cache[key] = [torch.empty(words, dtype=torch.int16, device=dev), False,
                                    weakref.ref(w, cls._on_dead(key)) if w is not None else (lambda: None), -1]
        elif w is not None and ent[2]() is not w:
            # the same address and geometry, another tensor object (a re-created view / a new network at a recycled address)
            if ent[1] or (cls.used is not None and key in cls.used):
                cls.invalidate()
                return cls.workspace(key, words, dev, w)
            import weakref
            ent[2], ent[3] = weakref.ref(w, cls._on_dead(key)), -1
        if cls.state == 'recording' and cls.used is not None:
            cls.used.add(key)
        ok = (ent[1] and cls.state == 'ready' and not cls.dead and cls.packed_step == cls.step and w is not None and
              ent[2]() is w and w._version == ent[3])
        return ent[0], ok


def params_changed():
    """Tell the once-per-step weight packing that parameters were rewritten after this step's batch launch (optimizer step,
    broadcast, checkpoint load): the calls of the rest of this step pack for themselves."""
    _PackBatch.packed_step = -1


def _zeros_d(n, dev):
    a = _ARENA.get(dev)
    if a is not None and a[1] + n <= _ARENA_DOUBLES:
        v = a[0][a[1]:a[1] + n]
        a[1] += (n + 1) // 2 * 2
        a[2] = max(a[2], a[1])
        return v
    return torch.zeros(n, dtype=torch.float64, device=dev)


# --------------------------------------------------------------------------------------------------
# gradient sinks: weight-gradient kernels write straight into the flat gradient buffer of trainer.FlatAdam
# --------------------------------------------------------------------------------------------------
# data_ptr of a parameter -> [view of the flat gradient buffer, written-this-step flag].  The kernels OVERWRITE their
# output, so the first backward use of a parameter in a step may write the view directly and return None to autograd
# (saves one temporary + one `grad += g` launch per parameter, ~236 per DIS-MF step); any further use in the same step
# falls back to returning the gradient for autograd to accumulate.  FlatAdam.zero_grad() re-arms the flags.
_GRAD_SINK = {}
_SINK_NOTIFY = None   # FlatAdam._notify when the bucketed all-reduce is active: told about every finished flat-buffer write
_SINK_PENDING = []    # parameters handed to a kernel by _sink() in the backward that is running


class GradJoin(object):
    """Shared gradient buffer of a tensor that has exactly TWO consumers on the tape (a residual branch, the two heads of
    a fork).  The consumer whose backward runs first stores its input gradient here and returns None to autograd; the
    one that runs second accumulates into the stored buffer inside its own kernel (conv epilogue / atomic scatter /
    row gather) and returns the buffer.  This replaces autograd's separate `g1 + g2` pass over the activation.
    Create one per forward call; tensors may be views of different shape over the same contiguous memory."""

    def __init__(self):
        self.buf = None

    def first(self, g):
        """called by the first producer: keep g, return what autograd should see (nothing)"""
        self.buf = g
        return None

    def take(self, shape):
        b, self.buf = self.buf, None
        return b.view(shape)


class GradAccum(object):
    """Shared gradient buffer of a tensor with SEVERAL consumers whose backward kernels accumulate (+= / atomics) into
    their output, e.g. a frame's depth map in the 6 directional flow-consistency terms it takes part in.  Every consumer
    calls use() in its forward; in backward it accumulates into buffer(like) and returns done() to autograd: None for
    all but the consumer that runs last, which hands over the sum.  One zero fill and no autograd adds instead of one
    fill per consumer and a chain of `g1 + g2`.  Create one per tensor and forward pass."""

    def __init__(self):
        self.buf = None
        self.pending = 0

    def use(self):
        self.pending += 1

    def buffer(self, like):
        if self.buf is None:
            self.buf = torch.zeros_like(like)
        return self.buf

    def done(self):
        self.pending -= 1
        if self.pending == 0:
            b, self.buf = self.buf, None
            return b
        return None


def register_grad_sinks(params, notify=None):
    """notify(param): called once the kernel that writes a parameter's gradient straight into the flat buffer has been
    launched (autograd never sees that gradient, so a post-accumulate hook would not fire for it)."""
    global _SINK_NOTIFY
    _GRAD_SINK.clear()
    _SINK_NOTIFY = notify
    del _SINK_PENDING[:]
    for p in params:
        if p.grad is not None:
            _GRAD_SINK[p.data_ptr()] = [p.grad, False]


def reset_grad_sinks():
    for e in _GRAD_SINK.values():
        e[1] = False
    del _SINK_PENDING[:]


def _sinks_written():
    """end of a backward that used _sink(): its kernels are enqueued, the flat-buffer gradients it wrote are final"""
    if _SINK_PENDING:
        if _SINK_NOTIFY is not None:
            for p_ in _SINK_PENDING:
                _SINK_NOTIFY(p_)
        del _SINK_PENDING[:]


def _sink(param):
    """-> (tensor to write the gradient into, value to return to autograd)"""
    e = _GRAD_SINK.get(param.data_ptr())
    # the tensor must really be the registered parameter: its .grad IS the flat view (guards against a stale entry
    # whose address was reused by an unrelated tensor)
    if (e is not None and not e[1] and e[0].shape == param.shape and param.grad is not None
            and param.grad.data_ptr() == e[0].data_ptr()):
        e[1] = True
        _SINK_PENDING.append(param)
        return e[0], None
    g = torch.empty_like(param)
    return g, g


def _unsink(param, got):
    """hand a sink back (the launch it was taken for did not run): the next _sink(param) gets it again"""
    if param is None or got is None or got[1] is not None:
        return
    e = _GRAD_SINK.get(param.data_ptr())
    if e is not None and e[1]:
        e[1] = False
        for i_ in range(len(_SINK_PENDING) - 1, -1, -1):
            if _SINK_PENDING[i_] is param:
                del _SINK_PENDING[i_]
                break


def _sink_opt(param):
    """_sink for an optional parameter (a bias): (None, None) without one.  Parameter gradients leave through the sinks, not
    through autograd's AccumulateGrad nodes: one launch less per parameter, and a captured step (trainer.GraphedStep) then never
    runs an AccumulateGrad node that an earlier, still referenced eager step created on the default stream - joining the
    legacy default stream into a capture crashes the HIP runtime."""
    return _sink(param) if param is not None else (None, None)


def _sink_block(params):
    """Gradient sink for a kernel that writes the gradients of several parameters as ONE contiguous block (Conv3D's
    five arrays): -> flat float view covering all of them if they are registered, unused so far and adjacent in the
    flat gradient buffer in this order, else None (the caller then returns separate tensors to autograd)."""
    es = []
    for p_ in params:
        e = _GRAD_SINK.get(p_.data_ptr())
        if (e is None or e[1] or e[0].shape != p_.shape or p_.grad is None or p_.grad.data_ptr() != e[0].data_ptr()):
            return None
        es.append(e)
    for a, b in zip(es[:-1], es[1:]):
        if b[0].data_ptr() != a[0].data_ptr() + a[0].numel() * 4:
            return None
    for e in es:
        e[1] = True
    _SINK_PENDING.extend(params)
    first = es[0][0]
    return torch.as_strided(first, (sum(e[0].numel() for e in es),), (1,), first.storage_offset())


# --------------------------------------------------------------------------------------------------
# LCN
# --------------------------------------------------------------------------------------------------
def lcn(x, radius=5, eps=0.05):
    """x (N,1,H,W) -> (lcn, std); no gradient (inputs are data).  reference model/networks.py:679-689"""
    x = _c(x)
    _chk(x)
    n, c, h, w = x.shape
    assert c == 1
    out = torch.empty_like(x)
    std = torch.empty_like(x)
    lib.call('dis_lcn_fwd', x, out, std, n, h, w, int(radius), float(eps))
    return out, std


def draw_augment_params(n, rng, max_blur=0.5, max_noise=3.0, max_sp_noise=0.0005):
    """the per-image random choices of the reference's augment_image (data/data_manipulation.py:161-177) with the
    dataset's settings (data/dataset.py:67-70), drawn with a numpy RandomState in the reference's order:
    (n, 6) float32 {blur flag, sigma_im, sigma_amb, noise_im, noise_amb, sp_ratio or -1}"""
    import numpy as np
    out = np.zeros((n, 6), np.float32)
    for i in range(n):
        if rng.uniform(0, 1) < 0.5:
            out[i, 0] = 1.0
            out[i, 1] = rng.uniform(0.2, max_blur)
            out[i, 2] = rng.uniform(0.2, max_blur)
        out[i, 3] = rng.uniform(0.0, max_noise)
        out[i, 4] = rng.uniform(0.0, max_noise)
        out[i, 5] = rng.uniform(0.0, max_sp_noise) if rng.uniform(0, 1) < 0.5 else -1.0
    return out


def augment(im, amb, params, seed):
    """device-side training augmentation (dis_augment): im, amb (..., H, W) float32; params (n,6) float32 and seed (1,) int64
    DEVICE tensors (n = number of H x W planes).  Returns (im_aug, amb_aug); no gradient (inputs are data)."""
    im, amb, params = _c(im), _c(amb), _c(params)
    _chk(im, amb, params)
    h, w = im.shape[-2:]
    n = im.numel() // (h * w)
    if tuple(params.shape) != (n, 6) or seed.dtype != torch.int64 or not seed.is_cuda or amb.shape != im.shape:
        raise RuntimeError('augment: params must be (n,6) float32, seed a CUDA int64 tensor, amb shaped like im')
    ws = torch.empty(2 * n, dtype=torch.int32, device=im.device)
    out_im, out_amb = torch.empty_like(im), torch.empty_like(amb)
    lib.call('dis_augment', im, amb, params, seed, ws, out_im, out_amb, n, h, w)
    return out_im, out_amb


# --------------------------------------------------------------------------------------------------
# photometric
# --------------------------------------------------------------------------------------------------
class _Photometric(torch.autograd.Function):
    @staticmethod
    def forward(ctx, es, ta, block_size, type, eps):
        es, ta = _c(es), _c(ta)
        _chk(es, ta)
        n, c, h, w = es.shape
        out = torch.empty((n, 1, h, w), dtype=es.dtype, device=es.device)
        lib.call('dis_photometric_fwd', es, ta, out, n, c, h, w, int(block_size), int(type), float(eps))
        ctx.save_for_backward(es, ta)
        ctx.cfg = (int(block_size), int(type), float(eps))
        return out

    @staticmethod
    def backward(ctx, grad_out):
        es, ta = ctx.saved_tensors
        block, type, eps = ctx.cfg
        n, c, h, w = es.shape
        g = _c(grad_out)
        ges = torch.empty_like(es)
        lib.call('dis_photometric_bwd', es, ta, g, ges, n, c, h, w, block, type, eps)
        return ges, None, None, None, None


# DIS_PHOTO_SINGLE_VIA_MULTI=0: a single estimate stays on the general kernels (diagnostics)
PHOTO_SINGLE_VIA_MULTI = _os_env.environ.get('DIS_PHOTO_SINGLE_VIA_MULTI', '1') != '0'


def photometric(es, ta, block_size, type, eps):
    # the census forms at the reference's window (9 x 9, one channel) run the round-4 kernels (csrc/pixel_ops.hip
    # census_*_multi_kernel with one estimate: fused multiply-adds, sign by clamp); everything else the general kernels
    if PHOTO_SINGLE_VIA_MULTI and es.dim() == 4 and photometric_multi_ok(1, es.shape[1], block_size, type):
        return _PhotometricMulti.apply(es.unsqueeze(0), ta, block_size, type, eps)[0]
    return _Photometric.apply(es, ta, block_size, type, eps)


class _PhotometricMulti(torch.autograd.Function):
    """census window loss of S stacked estimates (S, n, 1, h, w) against one target (n, 1, h, w) in one launch
    (dis_photometric_fwd_multi): the target's soft signs are evaluated once for all S."""

    @staticmethod
    def forward(ctx, es, ta, block_size, type, eps):
        es, ta = _c(es), _c(ta)
        _chk(es, ta)
        s, n, c, h, w = es.shape
        assert c == 1 and tuple(ta.shape) == (n, 1, h, w)
        out = torch.empty_like(es)
        lib.call('dis_photometric_fwd_multi', es, ta, out, s, n, h, w, int(block_size), int(type), float(eps))
        ctx.save_for_backward(es, ta)
        ctx.cfg = (int(block_size), int(type), float(eps))
        return out

    @staticmethod
    def backward(ctx, grad_out):
        es, ta = ctx.saved_tensors
        block, type, eps = ctx.cfg
        s, n, _, h, w = es.shape
        ges = torch.empty_like(es)
        lib.call('dis_photometric_bwd_multi', es, ta, _c(grad_out), ges, s, n, h, w, block, type, eps)
        return ges, None, None, None, None


# DIS_PHOTO_LOSS_ONE_NODE=0: pattern warp, census loss and weighted mean of the S estimates as separate autograd nodes (A/B)
PHOTO_LOSS_ONE_NODE = _os_env.environ.get('DIS_PHOTO_LOSS_ONE_NODE', '1') != '0'


def photometric_multi_ok(n_estimates, channels, block_size, type):
    return 1 <= n_estimates <= 4 and channels == 1 and int(block_size) == 9 and int(type) in (2, 3)


def photometric_multi(es_list, ta, block_size, type, eps):
    """[photometric(es, ta, ...) for es in es_list] in one launch (census types, 9 x 9, one channel, <= 4 estimates)."""
    out = _PhotometricMulti.apply(torch.stack([_c(e) for e in es_list], 0), ta, block_size, type, eps)
    return list(out.unbind(0))


# --------------------------------------------------------------------------------------------------
# pattern projection
# --------------------------------------------------------------------------------------------------
class _PatternWarp(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pattern, disp):
        pattern, disp = _c(pattern), _c(disp)
        _chk(pattern, disp)
        n, _, h, w = disp.shape
        assert pattern.numel() == h * w
        proj = torch.empty_like(disp)
        lib.call('dis_pattern_warp_fwd', pattern, disp, proj, n, h, w)
        ctx.save_for_backward(pattern, disp)
        return proj

    @staticmethod
    def backward(ctx, g):
        pattern, disp = ctx.saved_tensors
        n, _, h, w = disp.shape
        gd = torch.empty_like(disp)
        lib.call('dis_pattern_warp_bwd', pattern, disp, _c(g), gd, n, h, w)
        return None, gd


def pattern_warp(pattern, disp):
    return _PatternWarp.apply(pattern, disp)


class _PatternPhotoLossMulti(torch.autograd.Function):
    """[weighted_mean(photometric(pattern_warp(pattern, d), ta), std) for d in disps] as ONE autograd node: the S projections are
    written into the slices of one (S, n, 1, h, w) buffer, the census loss of all S runs in one launch (dis_photometric_fwd_multi)
    and the S weighted means read its slices - and on the way back the S mean gradients are written into the slices of ONE buffer.
    As separate nodes (pattern_warp, photometric_multi over torch.stack, weighted_mean over unbind) autograd moved the S maps
    through a stack on the way in and a cat on the way back: 2 x 113 MB of copies per DIS-SF bs = 8 step.  Same launches, same
    arithmetic per launch as the separate nodes (reference model/single_frame_worker.py:110-118 -> model/networks.py:336-377)."""

    @staticmethod
    def forward(ctx, pattern, ta, std, block_size, type, eps, *disps):
        pattern, ta = _c(pattern), _c(ta)
        std = _c(std) if std is not None else None
        disps = [_c(d) for d in disps]
        _chk(pattern, ta, std, *disps)
        s = len(disps)
        n, c, h, w = disps[0].shape
        assert c == 1 and tuple(ta.shape) == (n, 1, h, w) and pattern.numel() == h * w
        assert all(tuple(d.shape) == (n, 1, h, w) for d in disps)
        proj = torch.empty((s, n, 1, h, w), dtype=torch.float32, device=ta.device)
        for k in range(s):
            lib.call('dis_pattern_warp_fwd', pattern, disps[k], proj[k], n, h, w)
        diff = torch.empty_like(proj)
        lib.call('dis_photometric_fwd_multi', proj, ta, diff, s, n, h, w, int(block_size), int(type), float(eps))
        accs = _zeros_d(2 * s, ta.device)
        outs = torch.empty(s, dtype=torch.float32, device=ta.device)
        for k in range(s):
            lib.call('dis_weighted_mean_fwd', diff[k], std, accs[2 * k:2 * k + 2], outs[k:k + 1], diff[k].numel())
        ctx.save_for_backward(pattern, ta, std, proj, accs, *disps)
        ctx.cfg = (int(block_size), int(type), float(eps), s)
        return tuple(outs[k] for k in range(s))

    @staticmethod
    def backward(ctx, *gs):
        pattern, ta, std, proj, accs = ctx.saved_tensors[:5]
        disps = ctx.saved_tensors[5:]
        block, type, eps, s = ctx.cfg
        n, _, h, w = disps[0].shape
        gdiff = torch.empty_like(proj)
        for k in range(s):
            g = gs[k] if gs[k] is not None else torch.zeros((), dtype=torch.float32, device=proj.device)
            lib.call('dis_weighted_mean_bwd', std, accs[2 * k:2 * k + 2], _c(g), gdiff[k], gdiff[k].numel())
        gproj = torch.empty_like(proj)
        lib.call('dis_photometric_bwd_multi', proj, ta, gdiff, gproj, s, n, h, w, block, type, eps)
        gds = []
        for k in range(s):
            gd = torch.empty_like(disps[k])
            lib.call('dis_pattern_warp_bwd', pattern, disps[k], gproj[k], gd, n, h, w)
            gds.append(gd)
        return (None, None, None, None, None, None) + tuple(gds)


def pattern_photo_loss_multi(pattern, disps, ta, std, block_size, type, eps):
    """per-estimate census loss values (device scalars) of the pattern warped by each disparity map against one image"""
    return list(_PatternPhotoLossMulti.apply(pattern, ta, std, block_size, type, eps, *disps))


# --------------------------------------------------------------------------------------------------
# scalar reductions
# --------------------------------------------------------------------------------------------------
class _WeightedMean(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w):
        x = _c(x)
        w = _c(w) if w is not None else None
        _chk(x, w)
        acc = _zeros_d(2, x.device)
        out = torch.empty((), dtype=torch.float32, device=x.device)
        lib.call('dis_weighted_mean_fwd', x, w, acc, out, x.numel())
        ctx.save_for_backward(w, acc) if w is not None else ctx.save_for_backward(acc)
        ctx.has_w = w is not None
        ctx.shape = x.shape
        return out

    @staticmethod
    def backward(ctx, g):
        if ctx.has_w:
            w, acc = ctx.saved_tensors
        else:
            (acc,) = ctx.saved_tensors
            w = None
        gx = torch.empty(ctx.shape, dtype=torch.float32, device=acc.device)
        lib.call('dis_weighted_mean_bwd', w, acc, _c(g), gx, gx.numel())
        return gx, None


def weighted_mean(x, w=None):
    """sum(w*x)/sum(w)  (w None => mean)"""
    return _WeightedMean.apply(x, w)


class LossTerms(list):
    """The weighted loss terms a worker's loss_forward returns (a list of device scalars, as the reference returns them) that also
    carries `total`, their sum as ONE autograd node, and `weighted`, the same values as one vector."""
    total = None
    weighted = None


_WVEC = {}


class _WeightedTotal(torch.autograd.Function):
    """total = sum_i w_i * val_i over a worker's loss terms in three launches (stack, multiply, sum) and one on the way back, instead
    of two elementwise launches per term each way plus a chain of scalar additions (reference model/multi_frame_worker.py:103-175,
    single_frame_worker.py:101-165: `val * 0.2 / ge_num`, `sum(errs)`): ~45 launches of 5 us per DIS-MF step."""

    @staticmethod
    def forward(ctx, wvec, *vals):
        weighted = torch.stack([v.reshape(()) for v in vals]) * wvec
        ctx.save_for_backward(wvec)
        ctx.set_materialize_grads(False)
        return weighted.sum(), weighted

    @staticmethod
    def backward(ctx, gtotal, gw):
        # (gtotal: through `total`, the step's path; gw: through the individual terms, for a caller that sums them itself)
        wvec, = ctx.saved_tensors
        g = wvec * gtotal if gtotal is not None else None
        if gw is not None:
            g = wvec * gw if g is None else g + wvec * gw
        return (None,) + tuple(g.unbind(0))


def weighted_terms(pairs):
    """pairs: [(unweighted loss value: device scalar, weight: python float)] -> LossTerms of the weighted values"""
    ws = tuple(float(w) for _, w in pairs)
    dev = pairs[0][0].device
    wvec = _WVEC.get((ws, dev))
    if wvec is None:   # (created by the first, eager step; a captured step finds it)
        wvec = _WVEC[(ws, dev)] = torch.tensor(ws, dtype=torch.float32, device=dev)
    total, weighted = _WeightedTotal.apply(wvec, *[v for v, _ in pairs])
    out = LossTerms(weighted.unbind(0))
    out.total, out.weighted = total, weighted
    return out


class _L1Mean(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        a, b = _c(a), _c(b)
        _chk(a, b)
        acc = _zeros_d(1, a.device)
        out = torch.empty((), dtype=torch.float32, device=a.device)
        lib.call('dis_l1_mean_fwd', a, b, acc, out, a.numel())
        ctx.save_for_backward(a, b)
        return out

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        ga = torch.empty_like(a)
        lib.call('dis_l1_mean_bwd', a, b, _c(g), ga, a.numel())
        return ga, None


def l1_mean(a, b):
    return _L1Mean.apply(a, b)


class _SgmL1(torch.autograd.Function):
    @staticmethod
    def forward(ctx, o, sgm, noise, thresh):
        o, sgm, noise = _c(o), _c(sgm), _c(noise)
        _chk(o, sgm, noise)
        acc = _zeros_d(2, o.device)
        out = torch.empty((), dtype=torch.float32, device=o.device)
        lib.call('dis_sgm_l1_fwd', o, sgm, noise, float(thresh), acc, out, o.numel())
        ctx.save_for_backward(o, sgm, noise, acc)
        ctx.thresh = float(thresh)
        return out

    @staticmethod
    def backward(ctx, g):
        o, sgm, noise, acc = ctx.saved_tensors
        go = torch.empty_like(o)
        lib.call('dis_sgm_l1_bwd', o, sgm, noise, ctx.thresh, acc, _c(g), go, o.numel())
        return go, None, None, None


def sgm_l1(o, sgm_disp, noise, thresh=30.0):
    """sum(|o - sgm + noise| * (sgm > thresh)) / sum(sgm > thresh)  (the `real`-data warm-up term)"""
    return _SgmL1.apply(o, sgm_disp, noise, thresh)


class _SmoothLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, disp, amb):
        disp, amb = _c(disp), _c(amb)
        _chk(disp, amb)
        n, _, h, w = disp.shape
        acc = _zeros_d(1, disp.device)
        out = torch.empty((), dtype=torch.float32, device=disp.device)
        lib.call('dis_smooth_loss_fwd', disp, amb, acc, out, n, h, w)
        ctx.save_for_backward(disp, amb)
        return out

    @staticmethod
    def backward(ctx, g):
        disp, amb = ctx.saved_tensors
        n, _, h, w = disp.shape
        gd = torch.empty_like(disp)
        ws = torch.empty((n, 2, h, w), dtype=torch.float32, device=disp.device)
        lib.call('dis_smooth_loss_bwd', disp, amb, _c(g), gd, ws, n, h, w)
        return gd, None


def smooth_loss(disp, amb):
    return _SmoothLoss.apply(disp, amb)


class _DispToDepth(torch.autograd.Function):
    @staticmethod
    def forward(ctx, disp, bf):
        disp = _c(disp)
        _chk(disp)
        depth = torch.empty_like(disp)
        lib.call('dis_disp_to_depth_fwd', disp, depth, float(bf), disp.numel())
        ctx.save_for_backward(disp)
        ctx.bf = float(bf)
        return depth

    @staticmethod
    def backward(ctx, g):
        (disp,) = ctx.saved_tensors
        gd = torch.empty_like(disp)
        lib.call('dis_disp_to_depth_bwd', disp, _c(g), gd, ctx.bf, disp.numel())
        return gd, None


def disp_to_depth(disp, bf):
    return _DispToDepth.apply(disp, bf)


class _GeoLossDir(torch.autograd.Function):
    """One direction of the flow-consistency loss (dis_geo_loss_fwd/bwd)."""

    @staticmethod
    def forward(ctx, depth0, depth1, flow0, flow1, amb0, amb1, pdepth1, R0, t0, R1, t1, K, Kinv, clamp, accs):
        ctx.accs = accs  # optional (GradAccum of depth0, GradAccum of depth1)
        if accs is not None:
            accs[0].use()
            accs[1].use()
        ts = [_c(t) for t in (depth0, depth1, flow0, flow1, amb0, amb1)]
        depth0, depth1, flow0, flow1, amb0, amb1 = ts
        pdepth1 = _c(pdepth1) if pdepth1 is not None else None
        R0, t0, R1, t1 = _c(R0), _c(t0), _c(R1), _c(t1)
        _chk(depth0, depth1, flow0, flow1, amb0, amb1, pdepth1, R0, t0, R1, t1)
        bs, _, h, w = depth0.shape
        mask = torch.empty_like(depth0)
        acc = torch.empty(lib.fn('dis_geo_loss_acc_doubles')(), dtype=torch.float64, device=depth0.device)  # sums + block slots
        out = torch.empty((), dtype=torch.float32, device=depth0.device)
        lib.call('dis_geo_loss_fwd', depth0, depth1, flow0, flow1, amb0, amb1, pdepth1, R0, t0, R1, t1, K, Kinv,
                 float(clamp), mask, acc, out, bs, h, w)
        ctx.save_for_backward(depth0, depth1, flow0, R0, t0, R1, t1, mask, acc)
        ctx.cfg = (K, Kinv, float(clamp))
        ctx.mark_non_differentiable(mask)
        ctx.set_materialize_grads(False)  # no zero tensor for the mask's (non-existent) gradient
        return out, mask

    @staticmethod
    def backward(ctx, g, _gmask):
        depth0, depth1, flow0, R0, t0, R1, t1, mask, acc = ctx.saved_tensors
        K, Kinv, clamp = ctx.cfg
        bs, _, h, w = depth0.shape
        accs = ctx.accs
        g0 = accs[0].buffer(depth0) if accs is not None else torch.zeros_like(depth0)
        g1 = accs[1].buffer(depth1) if accs is not None else torch.zeros_like(depth1)
        lib.call('dis_geo_loss_bwd', depth0, depth1, flow0, R0, t0, R1, t1, K, Kinv, clamp, mask, acc, _c(g), g0, g1,
                 bs, h, w)  # accumulates into g0 (+=) and g1 (atomics)
        if accs is not None:
            g0, g1 = accs[0].done(), accs[1].done()
        return (g0, g1) + (None,) * 13


def geo_loss_dir(depth0, depth1, flow0, flow1, amb0, amb1, pdepth1, R0, t0, R1, t1, K, Kinv, clamp=-1.0, accs=None):
    """K, Kinv: lib.host_floats(9).  accs: optional (GradAccum, GradAccum) shared gradient buffers of depth0 / depth1.
    Returns (loss, mask)."""
    return _GeoLossDir.apply(depth0, depth1, flow0, flow1, amb0, amb1, pdepth1, R0, t0, R1, t1, K, Kinv, clamp, accs)


import ctypes as _ct
import os as _os
GEO_MULTI = _os.environ.get('DIS_GEO_MULTI', '1') != '0'   # =0: one launch per directional term (dis_geo_loss_fwd / _bwd)
GEO_MULTI_MAX = 16


class _GeoTerm(_ct.Structure):   # include/dis_hip.h: DisGeoTerm
    _fields_ = [(n_, _ct.c_void_p) for n_ in ('depth0', 'depth1', 'flow0', 'flow1', 'amb0', 'amb1', 'pdepth1', 'R0', 't0', 'R1', 't1',
                                              'mask', 'gdepth0', 'gdepth1')]


class _GeoLossAll(torch.autograd.Function):
    """Every directional flow-consistency term of a step in one forward and one backward launch (dis_geo_loss_fwd_multi / _bwd_multi).
    depth / amb / pdepth (tl, bs, 1, h, w), R (tl, bs, 3, 3), t (tl, bs, 3); pairs: ((i0, i1), ...) one per term; flows: per term
    flow_{i0 i1}, flow_{i1 i0}.  Returns the (len(pairs),) vector of term values."""

    @staticmethod
    def _table(pairs, depth, R, t, mask, flows0, flows1=None, amb=None, pdepth=None, gdepth=None):
        tl, bs = depth.shape[0], depth.shape[1]
        fs = depth[0].numel() * 4
        tab = (_GeoTerm * len(pairs))()
        for k, (i0, i1) in enumerate(pairs):
            q = tab[k]
            q.depth0, q.depth1 = depth.data_ptr() + i0 * fs, depth.data_ptr() + i1 * fs
            q.flow0 = flows0[k].data_ptr()
            q.R0, q.R1 = R.data_ptr() + i0 * bs * 36, R.data_ptr() + i1 * bs * 36
            q.t0, q.t1 = t.data_ptr() + i0 * bs * 12, t.data_ptr() + i1 * bs * 12
            q.mask = mask.data_ptr() + k * fs
            if flows1 is not None:
                q.flow1 = flows1[k].data_ptr()
                q.amb0, q.amb1 = amb.data_ptr() + i0 * fs, amb.data_ptr() + i1 * fs
                q.pdepth1 = pdepth.data_ptr() + i1 * fs if pdepth is not None else None
            if gdepth is not None:
                q.gdepth0, q.gdepth1 = gdepth.data_ptr() + i0 * fs, gdepth.data_ptr() + i1 * fs
        return tab

    @staticmethod
    def forward(ctx, depth, amb, pdepth, R, t, K, Kinv, clamp, pairs, *flows):
        depth, amb, R, t = _c(depth), _c(amb), _c(R), _c(t)
        pdepth = _c(pdepth) if pdepth is not None else None
        flows = [_c(f) for f in flows]
        _chk(depth, amb, pdepth, R, t, *flows)
        tl, bs, _, h, w = depth.shape
        T = len(pairs)
        assert len(flows) == 2 * T and T <= GEO_MULTI_MAX and amb.shape == depth.shape and (pdepth is None or pdepth.shape == depth.shape)
        assert tuple(R.shape) == (tl, bs, 3, 3) and tuple(t.shape) == (tl, bs, 3) and all(tuple(f.shape) == (bs, 2, h, w) for f in flows)
        mask = torch.empty((T, bs, 1, h, w), dtype=torch.float32, device=depth.device)
        acc = torch.empty(lib.fn('dis_geo_loss_multi_acc_doubles')(T), dtype=torch.float64, device=depth.device)
        out = torch.empty(T, dtype=torch.float32, device=depth.device)
        tab = _GeoLossAll._table(pairs, depth, R, t, mask, flows[0::2], flows[1::2], amb, pdepth)
        lib.call('dis_geo_loss_fwd_multi', tab, T, K, Kinv, float(clamp), acc, out, bs, h, w)
        ctx.save_for_backward(depth, R, t, mask, acc, *flows[0::2])
        ctx.cfg = (K, Kinv, float(clamp), pairs)
        return out

    @staticmethod
    def backward(ctx, g):
        depth, R, t, mask, acc = ctx.saved_tensors[:5]
        flows0 = ctx.saved_tensors[5:]
        K, Kinv, clamp, pairs = ctx.cfg
        tl, bs, _, h, w = depth.shape
        gd = torch.zeros_like(depth)
        tab = _GeoLossAll._table(pairs, depth, R, t, mask, flows0, gdepth=gd)
        lib.call('dis_geo_loss_bwd_multi', tab, len(pairs), K, Kinv, clamp, acc, _c(g.float()), bs, h, w)
        return (gd,) + (None,) * (8 + 2 * len(pairs))


def geo_loss_all(depth, amb, pdepth, R, t, K, Kinv, clamp, pairs, flows):
    """all directional terms at once: pairs ((i0, i1), ...), flows [(flow_{i0 i1}, flow_{i1 i0}), ...] -> (len(pairs),) values"""
    flat = [f for pr in flows for f in pr]
    return _GeoLossAll.apply(depth, amb, pdepth, R, t, K, Kinv, clamp, tuple(pairs), *flat)


# --------------------------------------------------------------------------------------------------
# layout helpers (no gradient: they only touch network inputs)
# --------------------------------------------------------------------------------------------------
def pack4_nhwc(srcs, n, h, w):
    """srcs: list of up to 4 (tensor, sample_stride_in_floats) or None -> (n,h,w,4) nhwc."""
    dev = next(s[0] for s in srcs if s is not None).device
    out = torch.empty((n, h, w, 4), dtype=torch.float32, device=dev)
    args = []
    for i in range(4):
        s = srcs[i] if i < len(srcs) else None
        if s is None:
            args += [None, 0]
        else:
            args += [s[0], int(s[1])]
    lib.call('dis_pack4_nhwc_strided', *args, out, n, h, w)
    return out


def planar_to_nhwc(x):
    x = _c(x)
    _chk(x)
    n, c, h, w = x.shape
    y = torch.empty((n, h, w, c), dtype=torch.float32, device=x.device)
    lib.call('dis_planar_to_nhwc', x, y, n, c, h, w)
    return y


def nhwc_to_planar(x):
    x = _c(x)
    _chk(x)
    n, h, w, c = x.shape
    y = torch.empty((n, c, h, w), dtype=torch.float32, device=x.device)
    lib.call('dis_nhwc_to_planar', x, y, n, c, h, w)
    return y


class _ResizePlanar(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, size, align_corners, scale0, scale1, c_for_scale):
        x = _c(x)
        _chk(x)
        lead = x.shape[:-2]
        hin, win = x.shape[-2:]
        nc = 1
        for d in lead:
            nc *= d
        y = torch.empty((*lead, size[0], size[1]), dtype=torch.float32, device=x.device)
        lib.call('dis_resize_bilinear_planar_fwd', x, y, nc, hin, win, size[0], size[1], int(align_corners),
                 float(scale0), float(scale1), int(c_for_scale))
        ctx.cfg = (nc, hin, win, size[0], size[1], int(align_corners), x.shape, int(c_for_scale))
        return y

    @staticmethod
    def backward(ctx, g):
        nc, hin, win, ho, wo, ac, shape, cfs = ctx.cfg
        if cfs != 0:
            raise RuntimeError('resize with flow scaling has no backward (flows are data)')
        gx = torch.empty(shape, dtype=torch.float32, device=g.device)
        lib.call('dis_resize_bilinear_planar_bwd', _c(g), gx, nc, hin, win, ho, wo, ac)
        return gx, None, None, None, None, None


def resize_planar(x, size, align_corners=True, flow_scale=None):
    """bilinear resize over the last two dims of a planar tensor.  flow_scale=(sx,sy) multiplies channel 0/1
    of a (...,2,H,W) flow tensor (reference resize_flow_like)."""
    if flow_scale is None:
        return _ResizePlanar.apply(x, tuple(size), align_corners, 1.0, 1.0, 0)
    assert x.shape[-3] == 2
    return _ResizePlanar.apply(x, tuple(size), align_corners, flow_scale[0], flow_scale[1], 2)


class _ResizeNHWC(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, size, align_corners):
        x = _c(x)
        _chk(x)
        n, hin, win, c = x.shape
        y = torch.empty((n, size[0], size[1], c), dtype=torch.float32, device=x.device)
        lib.call('dis_resize_bilinear_nhwc_fwd', x, y, n, hin, win, size[0], size[1], c, int(align_corners))
        ctx.cfg = (n, hin, win, size[0], size[1], c, int(align_corners))
        return y

    @staticmethod
    def backward(ctx, g):
        n, hin, win, ho, wo, c, ac = ctx.cfg
        gx = torch.empty((n, hin, win, c), dtype=torch.float32, device=g.device)
        lib.call('dis_resize_bilinear_nhwc_bwd', _c(g), gx, n, hin, win, ho, wo, c, ac)
        return gx, None, None


def resize_nhwc(x, size, align_corners=True):
    return _ResizeNHWC.apply(x, tuple(size), align_corners)


# --------------------------------------------------------------------------------------------------
# convolution (MFMA implicit GEMM)
# --------------------------------------------------------------------------------------------------
def _pack_w(weight, cin_pad, mode):
    cout, cin, k, _ = weight.shape
    packed = torch.empty(k * k * cin_pad * cout, dtype=torch.float32, device=weight.device)
    lib.call('dis_conv2d_pack_weights', weight, packed, cout, cin, cin_pad, k, mode)
    return packed


import os as _os
BF16X3 = _os.environ.get('DIS_CONV_BF16X3', '1') != '0'


K4S2_F2 = _os.environ.get('DIS_K4S2_F2', '1') != '0'   # DIS_K4S2_F2=0: the 4 x 4 stride-2 forward on the exact-fp32 MFMA kernel


def _dgrad_k4s2(gpre, weight, gx, n, hin, win, cin, cout, accumulate):
    """input gradient of a 4 x 4 stride-2 pad-1 conv: the one-launch two-term kernel for 32 -> 32 (csrc/conv_k4s2.hip), else - other
    channel counts, the three-term mode, DIS_K4S2_F2=0 - the four parity launches on the exact-fp32 kernel"""
    if (BF16X3 and K4S2_F2 and cin == 32 and cout == 32 and weight.is_contiguous() and
            lib.call_try('dis_conv2d_dgrad_k4s2_f16x2', gpre, weight, gx, n, hin, win, 1 if accumulate else 0)):
        return
    ws = torch.empty(16 * cin * cout, dtype=torch.float32, device=gx.device)
    lib.call('dis_conv2d_dgrad_strided', gpre, weight, gx, ws, n, hin, win, cin, cout, 4, 2, 1, 1 if accumulate else 0)


def _conv_wgrad_any(x, gpre, gw, gb, ws, n, hin, win, cin_pad, cin, cout, k, stride, pad):
    """dis_conv2d_wgrad, or its bf16x3 form for 3x3 stride-1 layers with 16 / 32 channels on both sides."""
    name = 'dis_conv2d_wgrad'
    if BF16X3 and cin_pad in (16, 32) and cout in (16, 32) and k == 3 and stride == 1:
        name = 'dis_conv2d_wgrad_bf16x3'
    lib.call(name, x, gpre, gw, gb, ws, n, hin, win, cin_pad, cin, cout, k, stride, pad)


def _conv_fwd_any(x, weight, cin_pad, mode, bias, y, stats, n, hin, win, cin, cout, k, stride, pad, act):
    """dis_conv2d_fwd, or its bf16x3 form (fp32 accuracy on the bf16 matrix cores) for 3x3 stride-1 layers with 16 / 32
    channels on both sides.  `weight` is the module's OIHW tensor, `mode` the weight order (0 forward, 1 stride-1 input
    gradient); `cin` / `cout` are the channel counts of x and y in THIS call (swapped for the input gradient)."""
    if (BF16X3 and K4S2_F2 and (cin, cout, k, stride, pad) == (32, 32, 4, 2, 1) and weight.is_contiguous() and
            lib.call_try('dis_conv2d_fwd_k4s2_f16x2', x, weight, bias, y, stats, n, hin, win, act)):
        return   # (FuseNet's 4 x 4 stride-2 down convolution on the two-term kernel; under the three-term mode the call says unsupported)
    if BF16X3 and cin in (16, 32) and cout in (16, 32) and k == 3 and stride == 1:
        # `weight` may be a slice w[:, a:b] of a wider weight (conv2d_multi): the kernel reads it through its row stride
        assert weight.stride(1) == 9 and weight.stride(2) == 3 and weight.stride(3) == 1
        lib.call('dis_conv2d_fwd_bf16x3_oihw', x, weight, mode, weight.shape[0], weight.shape[1], weight.stride(0), bias, y,
                 stats, n, hin, win, cin, cout, k, stride, pad, act)
    else:
        lib.call('dis_conv2d_fwd', x, _pack_w(_c(weight), cin_pad, mode), bias, y, stats, n, hin, win, cin, cout, k, stride,
                 pad, act)


# Round 6: input gradient + weight gradient of a 3x3 stride-1 conv 32 -> 32 in ONE launch (dis_conv2d_bwd_fused_f16x2,
# csrc/conv_bwd_fused.hip): the gy halo a tile stages feeds both products, gpre (the GroupNorm-backward pass formed on load) is never
# written, x and gy are read once.  Same-box: 0.79 - 0.95 of the two launches per layer, DIS-MF step 513.9 -> 537.6 frames/s
# (profiles/r6_bwd_fused.md).  DIS_BWD_FUSED=0 keeps the two launches.
BWD_FUSED = _os.environ.get('DIS_BWD_FUSED', '1') != '0'
_FUSED_WS = {}


# DIS_WGRAD_ACT=0: the 4 -> 16 stems run dis_act_bwd + dis_conv2d_wgrad instead of dis_conv2d_wgrad_act (A/B only)
WGRAD_ACT = _os_env.environ.get('DIS_WGRAD_ACT', '1') != '0'
# DIS_GW_INPLACE=0: conv2d_multi's fused launches write their weight-gradient slice to a temporary that is copied into the full
# gradient afterwards (the form before the slab reduce took a row pitch; A/B only)
GW_INPLACE = _os_env.environ.get('DIS_GW_INPLACE', '1') != '0'


def _gw_slice(gw, off, cs_i):
    if GW_INPLACE:
        return gw[:, off:off + cs_i]
    return torch.empty((gw.shape[0], cs_i, gw.shape[2], gw.shape[3]), dtype=torch.float32, device=gw.device)


def _bwd_fused_ok(cin, cout, k, stride, pad):
    return BWD_FUSED and BF16X3 and cin == 32 and cout == 32 and k == 3 and stride == 1 and pad == 1


def _bwd_fused(g, q, coef, in_act, gpre_out, weight, gx, accumulate, ab_x, ab_act, ab, x, xgn, gw, gb, n, h, w):
    """one launch for both gradients of a 3x3 stride-1 pad-1 conv 32 -> 32; False: no instance (the caller runs its two launches).
    xgn = (stats, gamma, beta, eps) when the conv's input is GroupNorm(x) applied on load, else None"""
    c = x.shape[-1]
    wsz = _FUSED_WS.get(c)
    if wsz is None:
        wsz = _FUSED_WS[c] = lib.fn('dis_conv2d_bwd_fused_workspace')(c)
    if wsz < 0:
        return False
    ws = torch.empty(wsz, dtype=torch.float32, device=x.device)
    st, gam, bet, eps = xgn if xgn is not None else (None, None, None, 0.0)
    # gw may be the (c, c, 3, 3) slice of a wider OIHW gradient (conv2d_multi): the slab reduce writes it in place
    assert tuple(gw.stride()[1:]) == (9, 3, 1) and gw.stride(0) % 9 == 0
    return lib.call_try('dis_conv2d_bwd_fused_f16x2', g, q, coef, in_act, gpre_out, weight, weight.shape[0], weight.shape[1],
                        weight.stride(0), gx, 1 if accumulate else 0, ab_x, ab_act, ab, x, st, gam, bet, float(eps), gw, gb, ws, n, h, w, c,
                        0 if gw.is_contiguous() else gw.stride(0))


def _bx_shape(cin, cout, k, stride):
    return BF16X3 and cin in (16, 32) and cout in (16, 32) and k == 3 and stride == 1


class _Conv2d(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, stride, pad, act, want_stats, need_dgrad, gy_is_pre=False, join=None, gnres=None, pend=None):
        x, weight = _c(x), _c(weight)
        _chk(x, weight, bias)
        ctx.gnres = gnres
        n, hin, win, cin_pad = x.shape
        cout, cin, k, _ = weight.shape
        ho = (hin + 2 * pad - k) // stride + 1
        wo = (win + 2 * pad - k) // stride + 1
        y = torch.empty((n, ho, wo, cout), dtype=torch.float32, device=x.device)
        stats = _zeros_d(2 * n, x.device) if want_stats else None
        if pend is not None:
            # x = SELU(GroupNorm(x2) + res) has not been written yet: this launch forms it on load and stores it (conv2d() checked
            # the shape; an unsupported mode falls back to the pass of its own)
            x2, gst, gam, bet, res, eps = pend
            if lib.call_try('dis_conv2d_fwd_f16x2_gnres', x2, gst, gam, bet, float(eps), res, x, weight, cout, cin, weight.stride(0),
                            bias, y, stats, n, hin, win, cin_pad, cout, act):
                _GN_PENDING.pop(x.data_ptr(), None)
            else:
                lib.call('dis_gn_apply', x2, gst, gam, bet, res, x, n, hin * win, cin_pad, ACT_SELU, float(eps))
                _GN_PENDING.pop(x.data_ptr(), None)
                _conv_fwd_any(x, weight, cin_pad, 0, bias, y, stats, n, hin, win, cin_pad, cout, k, stride, pad, act)
        else:
            _conv_fwd_any(x, weight, cin_pad, 0, bias, y, stats, n, hin, win, cin_pad, cout, k, stride, pad, act)
        if gy_is_pre:
            act = ACT_NONE  # the consumer (group_norm(in_act=...)) hands back the pre-activation gradient
        ctx.save_for_backward(x, weight, y if act != ACT_NONE else None)
        ctx.cfg = (stride, pad, act, bias is not None, need_dgrad)
        ctx.bias_ref = bias  # only its address/shape are used (gradient sink lookup)
        ctx.join = join
        if want_stats:
            ctx.mark_non_differentiable(stats)
            ctx.set_materialize_grads(False)  # no zero tensor for the statistics' (non-existent) gradient
            return y, stats
        return y, None

    @staticmethod
    def backward(ctx, gy, _gstats):
        x, weight, y = ctx.saved_tensors
        stride, pad, act, has_bias, need_dgrad = ctx.cfg
        n, hin, win, cin_pad = x.shape
        cout, cin, k, _ = weight.shape
        # a token of a GroupNorm behind this conv (ops._GN_LAZY): its backward's elementwise pass is applied by this conv's
        # input-gradient launch on load, which also writes the values for the weight-gradient launch
        lz = _gn_lazy_pop(gy, ctx)
        if (lz is not None and act == ACT_NONE and _gn_lazy_k4s2(cin_pad, cin, cout, k, stride, pad) and
                lib.fn('dis_get_conv_split')() == 1 and x[0].numel() * 4 < 0x7fff0000):   # (the kernel's per-sample 31-bit offsets)
            # the 4 x 4 stride-2 down convolution: the WEIGHT-gradient launch applies the pass while it stages gy (no halo there)
            # and stores the values for the four parity launches of the input gradient
            _, lg, lq, lcoef, lin_act = lz
            gpre = torch.empty_like(lg)
            gw, gw_ret = _sink(weight)
            gb, gb_ret = _sink(ctx.bias_ref) if has_bias else (None, None)
            ws = torch.empty(lib.fn('dis_conv2d_wgrad_workspace')(cin_pad, cout, k, stride), dtype=torch.float32, device=x.device)
            if lib.call_try('dis_conv2d_wgrad_k4s2_f16x2_gnb', x, lg, lq, lcoef, lin_act, gpre, gw, gb, ws, n, hin, win):
                gx = None
                if need_dgrad and ctx.needs_input_grad[0]:
                    join = ctx.join
                    second = join is not None and join.buf is not None
                    gx = join.take(x.shape) if second else torch.empty_like(x)
                    _dgrad_k4s2(gpre, weight, gx, n, hin, win, cin, cout, second)
                    if join is not None and not second:
                        gx = join.first(gx)
                _sinks_written()
                return gx, gw_ret, gb_ret, None, None, None, None, None, None, None, None, None
            # (no instance for this configuration in this build - e.g. an input activation the kernel has no form for: the sinks go
            #  back, the pass runs as a launch of its own, the general path below does the rest)
            _unsink(weight, (gw, gw_ret))
            if has_bias:
                _unsink(ctx.bias_ref, (gb, gb_ret))
            gy, lz = _gn_lazy_materialize(lz), None
        if lz is not None and not (act == ACT_NONE and need_dgrad and ctx.needs_input_grad[0] and cin_pad == cin and
                                   _gn_lazy_shape(cin, cout, k, stride, pad) and lib.fn('dis_get_conv_split')() == 1):
            gy, lz = _gn_lazy_materialize(lz), None
        if lz is not None:
            _, lg, lq, lcoef, lin_act = lz
            join = ctx.join
            second = join is not None and join.buf is not None
            gx = join.take(x.shape) if second else torch.empty_like(x)
            gnres = ctx.gnres
            gpre = torch.empty_like(lg)
            ab = ab_x = ab_act = None
            if (gnres is not None and second and (GN_SUMS & 2) and tuple(gnres[0].shape) == tuple(x.shape)):
                slots = lib.fn('dis_conv2d_gnsums_slots')()   # (the ResNetBlock-chain / two-consumer epilogue, see below)
                ab = _zeros_d(n * slots * 2 * cin, x.device)
                ab_x, ab_act = gnres[0], (x if len(gnres) == 1 else None)
            if _bwd_fused_ok(cin, cout, k, stride, pad):
                # one launch: the operand formed on load feeds the input gradient AND the weight gradient (gpre is never written)
                gw, gw_ret = _sink(weight)
                gb, gb_ret = _sink(ctx.bias_ref) if has_bias else (None, None)
                if _bwd_fused(lg, lq, lcoef, lin_act, None, weight, gx, second, ab_x, ab_act, ab, x, None, gw, gb, n, hin, win):
                    if ab is not None:
                        _GN_PRE[gx.data_ptr()] = (ab, slots)
                    if join is not None and not second:
                        gx = join.first(gx)
                    _sinks_written()
                    return gx, gw_ret, gb_ret, None, None, None, None, None, None, None, None, None
                _unsink(weight, (gw, gw_ret))
                if has_bias:
                    _unsink(ctx.bias_ref, (gb, gb_ret))
            if lib.call_try('dis_conv2d_dgrad_f16x2_gnb', lg, lq, lcoef, lin_act, gpre, weight, cout, cin, weight.stride(0), gx,
                            1 if second else 0, ab_x, ab_act, ab, n, hin, win, cin):
                if ab is not None:
                    _GN_PRE[gx.data_ptr()] = (ab, slots)
                if join is not None and not second:
                    gx = join.first(gx)
                gw, gw_ret = _sink(weight)
                gb, gb_ret = _sink(ctx.bias_ref) if has_bias else (None, None)
                ws = torch.empty(lib.fn('dis_conv2d_wgrad_workspace')(cin_pad, cout, k, stride), dtype=torch.float32, device=x.device)
                _conv_wgrad_any(x, gpre, gw, gb, ws, n, hin, win, cin_pad, cin, cout, k, stride, pad)
                _sinks_written()
                return gx, gw_ret, gb_ret, None, None, None, None, None, None, None, None, None
            # (no instance for this combination in this build: the separate pass, then the general path below)
            if second:
                join.buf = gx   # (hand the joined buffer back: the general path takes it again)
            gy = _gn_lazy_materialize(lz)
        gy = _c(gy)
        # bf16x3 shapes: the activation gradient is applied while gy is staged (dgrad and wgrad kernels), no separate pass
        fuse_act = act != ACT_NONE and _bx_shape(cin_pad, cout, k, stride)
        # the 4 -> 16 stems (conv1, amb_conv): no input gradient is asked for, so gy act'(y) would be formed for the weight-gradient
        # launch alone - that launch forms it on load (dis_conv2d_wgrad_act), the pass (read gy, y; write gpre) does not exist
        wgrad_act = (WGRAD_ACT and act != ACT_NONE and not fuse_act and not (need_dgrad and ctx.needs_input_grad[0]) and
                     cin_pad == 4 and cout == 16 and (k, stride) in ((3, 1), (4, 2)))
        if act != ACT_NONE and not fuse_act and not wgrad_act:
            gpre = torch.empty_like(gy)
            lib.call('dis_act_bwd', gy, y, gpre, act, gy.numel())
        else:
            gpre = gy
        gx = None
        if (need_dgrad and ctx.needs_input_grad[0] and cin_pad == cin and _bwd_fused_ok(cin, cout, k, stride, pad) and
                (fuse_act or gpre is gy) and lib.fn('dis_get_conv_split')() == 1 and
                not (ctx.gnres is not None and ctx.join is not None and ctx.join.buf is not None and (GN_SUMS & 2) and
                     tuple(ctx.gnres[0].shape) == tuple(x.shape))):
            # no GroupNorm-backward operand, no channel-sum epilogue: gy (or gy act'(y)) feeds both gradients in one launch
            join = ctx.join
            second = join is not None and join.buf is not None
            gx = join.take(x.shape) if second else torch.empty_like(x)
            gw, gw_ret = _sink(weight)
            gb, gb_ret = _sink(ctx.bias_ref) if has_bias else (None, None)
            if _bwd_fused(gy, y if fuse_act else None, None, act if fuse_act else ACT_NONE, None, weight, gx, second, None, None, None, x,
                          None, gw, gb, n, hin, win):
                if join is not None and not second:
                    gx = join.first(gx)
                _sinks_written()
                return gx, gw_ret, gb_ret, None, None, None, None, None, None, None, None, None
            _unsink(weight, (gw, gw_ret))
            if has_bias:
                _unsink(ctx.bias_ref, (gb, gb_ret))
            if second:
                join.buf = gx
            gx = None
        if need_dgrad and ctx.needs_input_grad[0]:
            assert cin_pad == cin
            join = ctx.join
            second = join is not None and join.buf is not None
            gx = join.take(x.shape) if second else torch.empty_like(x)
            gnres = ctx.gnres
            if (fuse_act and gnres is not None and join is None and act == ACT_SELU and (GN_SUMS & 1) and (cout, cin) == (16, 32) and
                    tuple(gnres[0].shape) == tuple(x.shape) and lib.fn('dis_get_conv_split')() == 1):
                # (final_conv: x = SELU(GroupNorm(.) + res) of ref_res3 and this conv is its only consumer - as below)
                slots = lib.fn('dis_conv2d_gnsums_slots')()
                ab = _zeros_d(n * slots * 2 * cin, x.device)
                lib.call('dis_conv2d_dgrad_bf16x3_act_gnsums_res', gy, y, weight, cout, cin, weight.stride(0), gx, x, gnres[0], ab,
                         n, gy.shape[1], gy.shape[2], cout, cin, k - 1 - pad)
                _GN_PRE[gx.data_ptr()] = (ab, slots)
            elif fuse_act:
                lib.call('dis_conv2d_dgrad_bf16x3_act', gy, y, act, weight, cout, cin, weight.stride(0), gx, n, gy.shape[1],
                         gy.shape[2], cout, cin, k - 1 - pad, 1 if second else 0)
            elif (gnres is not None and second and (GN_SUMS & 2) and _bx_shape(cin_pad, cout, k, stride) and cin == cout and
                  tuple(gnres[0].shape) == tuple(x.shape) and lib.fn('dis_get_conv_split')() == 1 and gpre is gy):
                # x is out = SELU(GroupNorm(x2) + res) of the previous ResNetBlock, and gx - which arrives holding this block's
                # residual-branch gradient - becomes the complete gradient wrt out here.  The epilogue turns it into the gradient
                # wrt the pre-activation value (times SELU'(x)) and leaves the GroupNorm-backward sums: that GroupNorm's backward
                # then needs neither its reduce pass nor a residual-gradient write (_GroupNorm.backward looks the buffer up)
                slots = lib.fn('dis_conv2d_gnsums_slots')()
                ab = _zeros_d(n * slots * 2 * cin, x.device)
                # (gnres = (x2,): x = SELU(GroupNorm(x2) + res); gnres = (x2, None): x = GroupNorm(x2) with two consumers, no SELU)
                lib.call('dis_conv2d_dgrad_bf16x3_gnsums_res', gpre, weight, cout, cin, weight.stride(0), gx,
                         x if len(gnres) == 1 else None, gnres[0], ab, n, gpre.shape[1], gpre.shape[2], cout, cin, k - 1 - pad)
                _GN_PRE[gx.data_ptr()] = (ab, slots)
            elif stride == 1:
                _conv_fwd_any(gpre, weight, cin, 1, None, gx, None, n, gpre.shape[1], gpre.shape[2], cout, cin, k, 1,
                              k - 1 - pad, ACT_NONE | (CONV_ACCUM if second else 0))
            else:
                _dgrad_k4s2(gpre, weight, gx, n, hin, win, cin, cout, second)
            if join is not None and not second:
                gx = join.first(gx)
        gw, gw_ret = _sink(weight)
        gb, gb_ret = _sink(ctx.bias_ref) if has_bias else (None, None)
        wsz = lib.fn('dis_conv2d_wgrad_workspace')(cin_pad, cout, k, stride)
        if wsz < 0:
            raise lib.DisHipError(f'conv2d wgrad: unsupported shape cin={cin_pad} cout={cout} k={k} s={stride}')
        ws = torch.empty(wsz, dtype=torch.float32, device=x.device)
        if fuse_act:
            lib.call('dis_conv2d_wgrad_bf16x3_act', x, gy, y, act, gw, gb, ws, n, hin, win, cin_pad, cin, cout, k, stride,
                     pad)
        elif wgrad_act:
            if not lib.call_try('dis_conv2d_wgrad_act', x, gy, y, act, gw, gb, ws, n, hin, win, cin_pad, cin, cout, k, stride, pad):
                gpre = torch.empty_like(gy)   # (no instance in this build: the pass as a launch of its own)
                lib.call('dis_act_bwd', gy, y, gpre, act, gy.numel())
                _conv_wgrad_any(x, gpre, gw, gb, ws, n, hin, win, cin_pad, cin, cout, k, stride, pad)
        else:
            _conv_wgrad_any(x, gpre, gw, gb, ws, n, hin, win, cin_pad, cin, cout, k, stride, pad)
        _sinks_written()
        return gx, gw_ret, gb_ret, None, None, None, None, None, None, None, None, None


def conv2d(x, weight, bias, stride=1, pad=0, act=ACT_NONE, want_stats=False, need_dgrad=True, gy_is_pre=False,
           join=None, gnres=None):
    """x nhwc (n,h,w,cin_pad>=cin); weight OIHW.  Returns (y, stats|None).
    gy_is_pre: the only consumer of y is group_norm(..., in_act=act), whose backward already multiplies by act'(y);
    the incoming gradient is then taken as the pre-activation gradient.
    join: GradJoin shared with the other consumer of x (see GradJoin)."""
    pend = getattr(x, '_pending_gn', None)
    if pend is not None:
        cout_, cin_, k_, _ = weight.shape
        if not (k_ == 3 and stride == 1 and pad == 1 and act == ACT_SELU and x.shape[-1] == cin_ and BF16X3 and
                ((cin_ == cout_ and cin_ in (16, 32)) or ((cin_, cout_) == (32, 16) and not want_stats))):
            realize(x)
            pend = None
        else:
            x._pending_gn = None   # (written by the launch below)
    out = _Conv2d.apply(x, weight, bias, stride, pad, act, want_stats, need_dgrad, gy_is_pre, join, gnres, pend)
    if (act == ACT_NONE or gy_is_pre) and ((need_dgrad and _gn_lazy_shape(x.shape[-1], weight.shape[0], weight.shape[2], stride, pad)) or
                                           _gn_lazy_k4s2(x.shape[-1], weight.shape[1], weight.shape[0], weight.shape[2], stride, pad)):
        _mark_lazy_ok(out)   # (a GroupNorm that is this output's ONLY consumer may answer with a token, see _GN_LAZY)
    return out


GN_FUSE = _os.environ.get('DIS_GN_FUSE', '1') != '0'
# GroupNorm backward from the sums of the input-gradient epilogue.  DIS_GN_SUMS=0 disables it; as a bit mask (diagnostics) it selects
# the forms: 1 final_conv (activation + residual), 2 residual / two-consumer, 4 GroupNorm-on-load pairs, 8 conv_fuse's first slice
GN_SUMS = int(_os.environ.get('DIS_GN_SUMS', '15'))


def gn_fusable(cin, cout, k, stride):
    """conv2d_gn_in exists for the bf16x3 square shapes (3x3, stride 1, 16 -> 16 / 32 -> 32)"""
    return GN_FUSE and BF16X3 and cin == cout and cin in (16, 32) and k == 3 and stride == 1


class _Conv2dGnIn(torch.autograd.Function):
    """y = act(conv3x3(GroupNorm(x)) + bias) where the GroupNorm (1 group; its statistics `gn_stats` come from the epilogue of
    the conv that produced x) is applied by the conv kernels while they stage x: the normalised tensor is never written or
    read (reference: the conv -> SELU -> GroupNorm -> conv chains of ResNetBlock / Block2D3D, model/multi_frame_networks.py
    :338-345,:514-542).  One autograd node for the pair GroupNorm + conv: its backward runs the conv's input gradient, the
    GroupNorm backward (which hands the producer of x its PRE-activation gradient when `in_act` is that producer's
    activation) and the conv's weight gradient with the same on-load normalisation."""

    @staticmethod
    def forward(ctx, x, gn_stats, gamma, beta, weight, bias, pad, act, want_stats, gy_is_pre, eps, in_act, x_lazy=False):
        ctx.x_lazy = x_lazy   # the producer of x redeems a token for the GroupNorm's elementwise backward pass (_GN_LAZY)
        x, weight = _c(x), _c(weight)
        _chk(x, gamma, beta, weight, bias)
        n, h, w, cin = x.shape
        cout, _, k, _ = weight.shape
        assert gn_fusable(cin, cout, k, 1) and weight.shape[1] == cin
        ho, wo = h + 2 * pad - 2, w + 2 * pad - 2
        y = torch.empty((n, ho, wo, cout), dtype=torch.float32, device=x.device)
        stats = _zeros_d(2 * n, x.device) if want_stats else None
        lib.call('dis_conv2d_fwd_bf16x3_gn', x, gn_stats, gamma, beta, float(eps), weight, cout, cin, weight.stride(0), bias, y,
                 stats, n, h, w, cin, cout, k, 1, pad, act)
        if gy_is_pre:
            act = ACT_NONE
        ctx.save_for_backward(x, gn_stats, gamma, weight, y if act != ACT_NONE else None)
        ctx.cfg = (pad, act, bias is not None, float(eps), in_act)
        ctx.bias_ref, ctx.beta_ref = bias, beta
        if want_stats:
            ctx.mark_non_differentiable(stats)
            ctx.set_materialize_grads(False)
            return y, stats
        return y, None

    @staticmethod
    def backward(ctx, gy, _gstats):
        x, gn_stats, gamma, weight, y = ctx.saved_tensors
        pad, act, has_bias, eps, in_act = ctx.cfg
        n, h, w, cin = x.shape
        cout, _, k, _ = weight.shape
        sums = (GN_SUMS & 4) and lib.fn('dis_get_conv_split')() == 1
        # (this conv's own output gradient may be a token of the GroupNorm behind it: redeemed by the input-gradient launch below)
        lz = _gn_lazy_pop(gy, ctx)
        if lz is not None and not (sums and act == ACT_NONE and _gn_lazy_shape(cin, cout, k, 1, pad)):
            gy, lz = _gn_lazy_materialize(lz), None
        gnorm = torch.empty_like(x)
        slots = lib.fn('dis_conv2d_gnsums_slots')() if sums else 0
        ab = _zeros_d(n * slots * 2 * cin, x.device) if sums else None
        gpre = None
        fused_done = False
        if lz is not None:
            _, lg, lq, lcoef, lin_act = lz
            if _bwd_fused_ok(cin, cout, k, 1, pad):
                # one launch: input gradient (+ the channel sums for this node's own GroupNorm) and the weight gradient with the
                # GroupNorm of x applied on load; x is fetched once for both
                gw, gw_ret = _sink(weight)
                gb, gb_ret = _sink(ctx.bias_ref) if has_bias else (None, None)
                if _bwd_fused(lg, lq, lcoef, lin_act, None, weight, gnorm, False, x, None, ab, x, (gn_stats, gamma, ctx.beta_ref, eps), gw, gb,
                              n, h, w):
                    fused_done, gpre = True, lg   # (gpre: only its shape is used below)
                else:
                    _unsink(weight, (gw, gw_ret))
                    if has_bias:
                        _unsink(ctx.bias_ref, (gb, gb_ret))
            if not fused_done:
                gpre = torch.empty_like(lg)
                if not lib.call_try('dis_conv2d_dgrad_f16x2_gnb', lg, lq, lcoef, lin_act, gpre, weight, cout, cin, weight.stride(0),
                                    gnorm, 0, x, None, ab, n, h, w, cin):
                    gy, gpre = _gn_lazy_materialize(lz), None
        dgrad_done = gpre is not None
        if not dgrad_done:
            gy = _c(gy)
            if act != ACT_NONE:   # (not the case in the networks here: the consumers are followed by a GroupNorm themselves)
                gpre = torch.empty_like(gy)
                lib.call('dis_act_bwd', gy, y, gpre, act, gy.numel())
            else:
                gpre = gy
        # gradient wrt the normalised tensor, then through the GroupNorm to the producer's (pre-activation) output
        gg, gg_ret = _sink(gamma)
        gbt, gbt_ret = _sink(ctx.beta_ref)
        hw = h * w
        if sums:
            # the input-gradient launch leaves the per-(sample, channel) sums of g and g * x in its epilogue: the GroupNorm
            # backward is then ONE elementwise pass (no reduce pass over g and x) - applied by the producer's input-gradient
            # launch on load when the producer redeems tokens (x_lazy), else a launch of its own
            if not dgrad_done:
                lib.call('dis_conv2d_dgrad_bf16x3_gnsums', gpre, weight, cout, cin, weight.stride(0), gnorm, x, ab, n, gpre.shape[1],
                         gpre.shape[2], cout, cin, k - 1 - pad)
            if ctx.x_lazy and GN_LAZY:
                gx = _gn_lazy_defer(gnorm, x, gn_stats, gamma, ab, slots, gg, gbt, n, hw, cin, eps, in_act, uid=ctx.x_lazy)
            else:
                gx = torch.empty_like(x)
                coef = torch.empty(n * (cin + 2) + 4 * n * cin + 2, dtype=torch.float32, device=x.device)
                lib.call('dis_gn_bwd_from_sums', gnorm, x, gn_stats, gamma, ab, slots, gx, gg, gbt, coef, n, hw, cin, eps, in_act)
        else:
            gx = torch.empty_like(x)
            _conv_fwd_any(gpre, weight, cin, 1, None, gnorm, None, n, gpre.shape[1], gpre.shape[2], cout, cin, k, 1, k - 1 - pad,
                          ACT_NONE)
            wtot = lib.fn('dis_gn_bwd_workspace')(n, cin)
            ws = torch.empty(wtot, dtype=torch.float64, device=x.device)
            nred2 = wtot // (2 + 2 * cin) * 2
            lib.call('dis_gn_apply_bwd', gnorm, None, x, gn_stats, gamma, gx, None, gg, gbt, ws[:nred2], ws[nred2:], n, hw, cin,
                     ACT_NONE, eps, in_act)
        if not fused_done:
            gw, gw_ret = _sink(weight)
            gb, gb_ret = _sink(ctx.bias_ref) if has_bias else (None, None)
            wsz = lib.fn('dis_conv2d_wgrad_workspace')(cin, cout, k, 1)
            wws = torch.empty(wsz, dtype=torch.float32, device=x.device)
            lib.call('dis_conv2d_wgrad_bf16x3_gn', x, gn_stats, gamma, ctx.beta_ref, eps, gpre, gw, gb, wws, n, h, w, cin, cin, cout,
                     k, 1, pad)
        _sinks_written()
        return gx, None, gg_ret, gbt_ret, gw_ret, gb_ret, None, None, None, None, None, None, None


def conv2d_gn_in(x, gn_stats, gamma, beta, weight, bias, pad=1, act=ACT_NONE, want_stats=False, gy_is_pre=False, eps=1e-5,
                 in_act=ACT_NONE):
    """conv2d(group_norm(x, gamma, beta, stats=gn_stats, in_act=in_act), weight, bias, 1, pad, act, ...) with the
    normalisation applied on load.  Returns (y, stats|None).  Shapes: see gn_fusable()."""
    out = _Conv2dGnIn.apply(x, gn_stats, gamma, beta, weight, bias, pad, act, want_stats, gy_is_pre, eps, in_act,
                            int(getattr(x, '_gn_lazy_ok', 0)))
    if (act == ACT_NONE or gy_is_pre) and _gn_lazy_shape(x.shape[-1], weight.shape[0], weight.shape[2], 1, pad):
        _mark_lazy_ok(out)
    return out


class _Conv2dMulti(torch.autograd.Function):
    """Conv2d (stride 1) over the channel concatenation of several nhwc tensors WITHOUT materialising the
    concatenation: one launch per source with the matching slice of the weight, the later launches accumulate into
    y (DIS_CONV_ACCUM); bias enters with the first launch, the activation and the GroupNorm statistics with the last.
    The backward produces one input gradient per source directly (no split copies).
    gn_*: the FIRST source is the input of a GroupNorm that is applied on load (see _Conv2dGnIn): xs[0] is the pre-normalisation
    tensor, gn_meta = (eps, in_act); its GroupNorm backward runs inside this node (sums in the input-gradient epilogue)."""

    @staticmethod
    def forward(ctx, weight, bias, pad, act, want_stats, gy_is_pre, gn_stats, gn_gamma, gn_beta, gn_meta, *xs):
        ctx.x0_lazy = int(gn_meta[2]) if (gn_meta is not None and len(gn_meta) > 2 and gn_meta[2]) else 0   # (uid of xs[0]'s producer)
        gn_meta = gn_meta[:2] if gn_meta is not None else None
        # (tokens of a GroupNorm behind this conv are redeemed by the first source's 3 x 3 c -> c input-gradient launch)
        ctx.lazy_ok = _multi_lazy_ok(weight, xs, pad, act)
        # a source that is SELU(GroupNorm(x2) + residual) of a ResNetBlock and has no other consumer: its input-gradient launch
        # leaves that GroupNorm's backward sums (as _Conv2d.backward does for final_conv)
        ctx.gnres = [getattr(x, '_gn_res_src', None) if x.requires_grad else None for x in xs]
        xs = [_c(x) for x in xs]
        weight = _c(weight)
        _chk(weight, bias, *xs)
        cout, cin, k, _ = weight.shape
        n, h, w, _ = xs[0].shape
        cs = [x.shape[3] for x in xs]
        assert sum(cs) == cin and all(tuple(x.shape[:3]) == (n, h, w) for x in xs)
        ho, wo = h + 2 * pad - k + 1, w + 2 * pad - k + 1
        y = torch.empty((n, ho, wo, cout), dtype=torch.float32, device=weight.device)
        stats = _zeros_d(2 * n, weight.device) if want_stats else None
        off = 0
        for i, x in enumerate(xs):
            last = i == len(xs) - 1
            wi = weight[:, off:off + cs[i]]  # a view: no copy
            a = (act if last else ACT_NONE) | (CONV_ACCUM if i > 0 else 0)
            if i == 0 and gn_meta is not None:
                assert gn_fusable(cs[0], cout, k, 1) and len(xs) > 1
                lib.call('dis_conv2d_fwd_bf16x3_gn', x, gn_stats, gn_gamma, gn_beta, float(gn_meta[0]), wi, cout, cs[0],
                         wi.stride(0), bias, y, None, n, h, w, cs[0], cout, k, 1, pad, ACT_NONE)
            else:
                _conv_fwd_any(x, wi, cs[i], 0, bias if i == 0 else None, y, stats if last else None, n, h, w, cs[i], cout, k, 1,
                              pad, a)
            off += cs[i]
        if gy_is_pre:
            act = ACT_NONE
        ctx.save_for_backward(weight, y if act != ACT_NONE else None, gn_stats, gn_gamma, *xs)
        ctx.cfg = (pad, act, bias is not None, cs, gn_meta)
        ctx.bias_ref, ctx.beta_ref = bias, gn_beta
        if want_stats:
            ctx.mark_non_differentiable(stats)
            ctx.set_materialize_grads(False)  # no zero tensor for the statistics' (non-existent) gradient
            return y, stats
        return y, None

    @staticmethod
    def backward(ctx, gy, _gstats):
        weight, y, gn_stats, gn_gamma = ctx.saved_tensors[:4]
        xs = ctx.saved_tensors[4:]
        pad, act, has_bias, cs, gn_meta = ctx.cfg
        cout, cin, k, _ = weight.shape
        n, h, w, _ = xs[0].shape
        fuse_act = act != ACT_NONE and all(_bx_shape(c_, cout, k, 1) for c_ in cs)  # see _Conv2d.backward
        assert gn_meta is None or act == ACT_NONE
        # this conv's output gradient may be a token of the GroupNorm behind it (conv_fuse): the FIRST source's input-gradient launch
        # applies the elementwise pass on load and stores the result for every other launch of this node
        lz = _gn_lazy_pop(gy, ctx)
        if lz is not None and not (act == ACT_NONE and ctx.lazy_ok and ctx.needs_input_grad[10] and
                                   (gn_meta is None or (GN_SUMS & 8)) and lib.fn('dis_get_conv_split')() == 1):
            gy, lz = _gn_lazy_materialize(lz), None
        if lz is None:
            gy = _c(gy)
        if lz is not None:
            gpre = None   # (set by the first source's launch below)
        elif act != ACT_NONE and not fuse_act:
            gpre = torch.empty_like(gy)
            lib.call('dis_act_bwd', gy, y, gpre, act, gy.numel())
        else:
            gpre = gy
        gw, gw_ret = _sink(weight)
        gb, gb_ret = _sink(ctx.bias_ref) if has_bias else (None, None)
        gg_ret = gbt_ret = None
        gxs = []
        off = 0
        need_w = [None]   # the weight gradient of the current slice when a fused launch has produced it
        for i, x in enumerate(xs):
            wi = weight[:, off:off + cs[i]]  # a view: no copy
            gn0 = i == 0 and gn_meta is not None
            gx = None
            if ctx.needs_input_grad[10 + i]:
                gx = torch.empty_like(x)
                if gn0:
                    # gradient wrt the normalised tensor, then through the GroupNorm (as _Conv2dGnIn.backward)
                    eps, in_act = gn_meta
                    gnorm = torch.empty_like(x)
                    gg, gg_ret = _sink(gn_gamma)
                    gbt, gbt_ret = _sink(ctx.beta_ref)
                    if (GN_SUMS & 8) and lib.fn('dis_get_conv_split')() == 1:
                        slots = lib.fn('dis_conv2d_gnsums_slots')()
                        ab = _zeros_d(n * slots * 2 * cs[0], x.device)
                        if lz is not None:
                            _, lg, lq, lcoef, lin_act = lz
                            gpre = torch.empty_like(lg)
                            if _bwd_fused_ok(cs[0], cout, k, 1, pad) and need_w[0] is None:
                                # one launch: operand formed on load (stored for the other sources' launches), input gradient +
                                # channel sums, and this slice's weight / bias gradient with GroupNorm(x) on load
                                gw0 = _gw_slice(gw, off, cs[0])   # (the slab reduce writes the slice of the full gradient in place)
                                if _bwd_fused(lg, lq, lcoef, lin_act, gpre, wi, gnorm, False, x, None, ab, x,
                                              (gn_stats, gn_gamma, ctx.beta_ref, float(gn_meta[0])), gw0, gb if has_bias else None, n, h, w):
                                    need_w[0] = gw0
                            if need_w[0] is not None:
                                pass
                            elif not lib.call_try('dis_conv2d_dgrad_f16x2_gnb', lg, lq, lcoef, lin_act, gpre, wi, cout, cs[0], wi.stride(0),
                                                  gnorm, 0, x, None, ab, n, h, w, cs[0]):
                                gpre = _gn_lazy_materialize(lz)   # (no instance in this build: the pass as a launch of its own)
                                lib.call('dis_conv2d_dgrad_bf16x3_gnsums', gpre, wi, cout, cs[0], wi.stride(0), gnorm, x, ab, n,
                                         gpre.shape[1], gpre.shape[2], cout, cs[0], k - 1 - pad)
                            lz = None
                        else:
                            lib.call('dis_conv2d_dgrad_bf16x3_gnsums', gpre, wi, cout, cs[0], wi.stride(0), gnorm, x, ab, n,
                                     gpre.shape[1], gpre.shape[2], cout, cs[0], k - 1 - pad)
                        if ctx.x0_lazy and GN_LAZY:   # (the producer of xs[0] applies the elementwise pass on load: _GN_LAZY)
                            gx = _gn_lazy_defer(gnorm, x, gn_stats, gn_gamma, ab, slots, gg, gbt, n, h * w, cs[0], float(eps), in_act, uid=ctx.x0_lazy)
                        else:
                            coef = torch.empty(n * (cs[0] + 2) + 4 * n * cs[0] + 2, dtype=torch.float32, device=x.device)
                            lib.call('dis_gn_bwd_from_sums', gnorm, x, gn_stats, gn_gamma, ab, slots, gx, gg, gbt, coef, n, h * w,
                                     cs[0], float(eps), in_act)
                    else:
                        _conv_fwd_any(gpre, wi, cs[0], 1, None, gnorm, None, n, gpre.shape[1], gpre.shape[2], cout, cs[0], k, 1,
                                      k - 1 - pad, ACT_NONE)
                        wtot = lib.fn('dis_gn_bwd_workspace')(n, cs[0])
                        wsd = torch.empty(wtot, dtype=torch.float64, device=x.device)
                        nred2 = wtot // (2 + 2 * cs[0]) * 2
                        lib.call('dis_gn_apply_bwd', gnorm, None, x, gn_stats, gn_gamma, gx, None, gg, gbt, wsd[:nred2], wsd[nred2:],
                                 n, h * w, cs[0], ACT_NONE, float(eps), in_act)
                elif (fuse_act and act == ACT_SELU and ctx.gnres[i] is not None and (GN_SUMS & 1) and (cout, cs[i]) == (32, 16) and
                      k == 3 and pad == 1 and tuple(ctx.gnres[i][0].shape) == tuple(x.shape) and lib.fn('dis_get_conv_split')() == 1):
                    slots = lib.fn('dis_conv2d_gnsums_slots')()
                    ab = _zeros_d(n * slots * 2 * cs[i], x.device)
                    lib.call('dis_conv2d_dgrad_bf16x3_act_gnsums_res', gy, y, wi, cout, cs[i], wi.stride(0), gx, x, ctx.gnres[i][0], ab,
                             n, gy.shape[1], gy.shape[2], cout, cs[i], k - 1 - pad)
                    _GN_PRE[gx.data_ptr()] = (ab, slots)
                elif fuse_act:
                    if _bwd_fused_ok(cs[i], cout, k, 1, pad) and lib.fn('dis_get_conv_split')() == 1:
                        # (ref_conv's 32-channel slice: gy act'(y) feeds the input gradient and the weight gradient in one launch)
                        gwf = _gw_slice(gw, off, cs[i])
                        if _bwd_fused(gy, y, None, act, None, wi, gx, False, None, None, None, x, None, gwf,
                                      gb if (i == 0 and has_bias) else None, n, h, w):
                            need_w[0] = gwf
                    if need_w[0] is None:
                        lib.call('dis_conv2d_dgrad_bf16x3_act', gy, y, act, wi, cout, cs[i], wi.stride(0), gx, n, gy.shape[1],
                                 gy.shape[2], cout, cs[i], k - 1 - pad, 0)
                elif lz is not None:   # (i == 0: the token's pass on load, no epilogue)
                    _, lg, lq, lcoef, lin_act = lz
                    gpre = torch.empty_like(lg)
                    if not lib.call_try('dis_conv2d_dgrad_f16x2_gnb', lg, lq, lcoef, lin_act, gpre, wi, cout, cs[i], wi.stride(0), gx, 0,
                                        None, None, None, n, h, w, cs[i]):
                        gpre = _gn_lazy_materialize(lz)
                        _conv_fwd_any(gpre, wi, cs[i], 1, None, gx, None, n, gpre.shape[1], gpre.shape[2], cout, cs[i], k, 1,
                                      k - 1 - pad, ACT_NONE)
                    lz = None
                else:
                    if _bwd_fused_ok(cs[i], cout, k, 1, pad) and lib.fn('dis_get_conv_split')() == 1:
                        gwf = _gw_slice(gw, off, cs[i])
                        if _bwd_fused(gpre, None, None, ACT_NONE, None, wi, gx, False, None, None, None, x, None, gwf,
                                      gb if (i == 0 and has_bias) else None, n, h, w):
                            need_w[0] = gwf
                    if need_w[0] is None:
                        _conv_fwd_any(gpre, wi, cs[i], 1, None, gx, None, n, gpre.shape[1], gpre.shape[2], cout, cs[i], k, 1,
                                      k - 1 - pad, ACT_NONE)
            assert lz is None   # (redeemed by the first source's launch: ctx.lazy_ok guarantees that it has one)
            gxs.append(gx)
            if need_w[0] is not None:   # (the fused launch above has written this slice of the weight gradient in place)
                if not GW_INPLACE:
                    gw[:, off:off + cs[i]].copy_(need_w[0])
                need_w[0] = None
                off += cs[i]
                continue
            gwi = torch.empty((cout, cs[i], k, k), dtype=torch.float32, device=x.device)
            wsz = lib.fn('dis_conv2d_wgrad_workspace')(cs[i], cout, k, 1)
            if wsz < 0:
                raise lib.DisHipError(f'conv2d_multi wgrad: unsupported shape cin={cs[i]} cout={cout} k={k}')
            ws = torch.empty(wsz, dtype=torch.float32, device=x.device)
            gbi = gb if (i == 0 and has_bias) else None
            if gn0:
                lib.call('dis_conv2d_wgrad_bf16x3_gn', x, gn_stats, gn_gamma, ctx.beta_ref, float(gn_meta[0]), gpre, gwi, gbi, ws, n,
                         h, w, cs[0], cs[0], cout, k, 1, pad)
            elif fuse_act:
                lib.call('dis_conv2d_wgrad_bf16x3_act', x, gy, y, act, gwi, gbi, ws, n, h, w, cs[i], cs[i], cout, k, 1, pad)
            else:
                _conv_wgrad_any(x, gpre, gwi, gbi, ws, n, h, w, cs[i], cs[i], cout, k, 1, pad)
            gw[:, off:off + cs[i]].copy_(gwi)  # small strided memory move into the (flat) weight-gradient slice
            off += cs[i]
        _sinks_written()
        return (gw_ret, gb_ret, None, None, None, None, None, gg_ret, gbt_ret, None) + tuple(gxs)


def _multi_lazy_ok(weight, xs, pad, act):
    cout, _, k, _ = weight.shape
    return bool(act == ACT_NONE and xs[0].requires_grad and xs[0].shape[-1] == cout and _gn_lazy_shape(cout, cout, k, 1, pad))


def conv2d_multi(xs, weight, bias, pad=0, act=ACT_NONE, want_stats=False, gy_is_pre=False, gn0=None):
    """conv2d(cat(xs, channel dim), weight) without the cat.  Returns (y, stats|None).
    gn0 = (stats, gamma, beta, eps, in_act): xs[0] is the INPUT of a GroupNorm(1 group) that is applied on load (conv2d_gn_in)."""
    if gn0 is not None:
        out = _Conv2dMulti.apply(weight, bias, pad, act, want_stats, gy_is_pre, gn0[0], gn0[1], gn0[2],
                                 (gn0[3], gn0[4], int(getattr(xs[0], '_gn_lazy_ok', 0))), *xs)
    else:
        out = _Conv2dMulti.apply(weight, bias, pad, act, want_stats, gy_is_pre, None, None, None, None, *xs)
    if _multi_lazy_ok(weight, xs, pad, act):
        _mark_lazy_ok(out)   # (a GroupNorm that is this output's ONLY consumer may answer with a token, see _GN_LAZY)
    return out


class _Conv2dScaledIn(torch.autograd.Function):
    """y = conv2d(x * xscale[..., chunk]) with the multiplier applied inside the kernels: forward while the input is
    staged, input gradient in the epilogue (optionally accumulating into a GradJoin buffer), weight gradient while x
    is staged.  x (n,h,w,cin), xscale (n,h,w,cin/32)."""

    @staticmethod
    def forward(ctx, x, xscale, weight, bias, stride, pad, act, want_stats, join):
        x, xscale, weight = _c(x), _c(xscale), _c(weight)
        _chk(x, xscale, weight, bias)
        n, hin, win, cin = x.shape
        cout, _, k, _ = weight.shape
        assert cin % 32 == 0 and tuple(xscale.shape) == (n, hin, win, cin // 32) and act == ACT_NONE
        ho, wo = (hin + 2 * pad - k) // stride + 1, (win + 2 * pad - k) // stride + 1
        y = torch.empty((n, ho, wo, cout), dtype=torch.float32, device=x.device)
        stats = _zeros_d(2 * n, x.device) if want_stats else None
        lib.call('dis_conv2d_fwd_scaled', x, xscale, _pack_w(weight, cin, 0), bias, y, None, stats, n, hin, win, cin, cout,
                 k, stride, pad, act)
        ctx.save_for_backward(x, xscale, weight)
        ctx.cfg = (stride, pad, bias is not None)
        ctx.bias_ref = bias
        ctx.join = join
        if want_stats:
            ctx.mark_non_differentiable(stats)
            ctx.set_materialize_grads(False)  # no zero tensor for the statistics' (non-existent) gradient
            return y, stats
        return y, None

    @staticmethod
    def backward(ctx, gy, _gstats):
        x, xscale, weight = ctx.saved_tensors
        stride, pad, has_bias = ctx.cfg
        n, hin, win, cin = x.shape
        cout, _, k, _ = weight.shape
        # (a token of the GroupNorm behind this conv, ops._GN_LAZY: the 1 x 1 input-gradient launch applies the pass on load)
        lz = _gn_lazy_pop(gy, ctx)
        if lz is not None and not (ctx.needs_input_grad[0] and k == 1 and stride == 1 and pad == 0 and (cin, cout) == (128, 32)):
            gy, lz = _gn_lazy_materialize(lz), None
        gx = None
        if lz is not None:
            _, lg, lq, lcoef, lin_act = lz
            join = ctx.join
            second = join is not None and join.buf is not None
            gx = join.take(x.shape) if second else torch.empty_like(x)
            gy = torch.empty_like(lg)
            lib.call('dis_conv2d_dgrad1x1_scaled_gnb', lg, lq, lcoef, lin_act, gy, _pack_w(weight, cin, 1), gx, xscale, n, hin, win,
                     cout, cin, 1 if second else 0)
            if join is not None and not second:
                gx = join.first(gx)
        gy = _c(gy)
        if lz is None and ctx.needs_input_grad[0]:
            assert stride == 1
            join = ctx.join
            second = join is not None and join.buf is not None
            gx = join.take(x.shape) if second else torch.empty_like(x)
            lib.call('dis_conv2d_fwd_scaled', gy, None, _pack_w(weight, cin, 1), None, gx, xscale, None, n, gy.shape[1],
                     gy.shape[2], cout, cin, k, 1, k - 1 - pad, ACT_NONE | (CONV_ACCUM if second else 0))
            if join is not None and not second:
                gx = join.first(gx)
        gw, gw_ret = _sink(weight)
        gb, gb_ret = _sink(ctx.bias_ref) if has_bias else (None, None)
        wsz = lib.fn('dis_conv2d_wgrad_workspace')(cin, cout, k, stride)
        ws = torch.empty(wsz, dtype=torch.float32, device=x.device)
        lib.call('dis_conv2d_wgrad_scaled', x, xscale, gy, gw, gb, ws, n, hin, win, cin, cin, cout, k, stride, pad)
        _sinks_written()
        return gx, None, gw_ret, gb_ret, None, None, None, None, None


def conv2d_scaled_in(x, xscale, weight, bias, stride=1, pad=0, want_stats=False, join=None):
    out = _Conv2dScaledIn.apply(x, xscale, weight, bias, stride, pad, ACT_NONE, want_stats, join)
    if GN_LAZY and weight.shape[2] == 1 and stride == 1 and pad == 0 and (x.shape[-1], weight.shape[0]) == (128, 32):
        _mark_lazy_ok(out)   # (its backward redeems GroupNorm tokens: dis_conv2d_dgrad1x1_scaled_gnb)
    return out


def slot_weights(geom):
    """(tl,bs,h,w,tl,4) geometry -> (tl*bs,h,w,tl) multipliers mask/mean(mask) (reference multi_frame_networks.py:410)"""
    geom = _c(geom)
    tl, bs, h, w, s, _ = geom.shape
    out = torch.empty((tl * bs, h, w, s), dtype=torch.float32, device=geom.device)
    lib.call('dis_slot_weights', geom, out, tl * bs * h * w, s)
    return out


class _DispHead(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, alpha, offset):
        x, weight, bias = _c(x), _c(weight), _c(bias)
        _chk(x, weight, bias)
        n, h, w, cin = x.shape
        y = torch.empty((n, 1, h, w), dtype=torch.float32, device=x.device)
        lib.call('dis_disp_head_fwd', x, weight, bias, y, n, h, w, cin, float(alpha), float(offset))
        ctx.save_for_backward(x, weight, y)
        ctx.alpha = float(alpha)
        ctx.bias_ref = bias
        return y

    @staticmethod
    def backward(ctx, gy):
        x, weight, y = ctx.saved_tensors
        n, h, w, cin = x.shape
        gx = torch.empty_like(x)
        gw, gw_ret = _sink(weight)
        gb, gb_ret = _sink(ctx.bias_ref)
        ws = torch.empty(lib.fn('dis_disp_head_bwd_workspace')(n, h, w, cin), dtype=torch.float32, device=x.device)
        lib.call('dis_disp_head_bwd', x, weight, y, _c(gy), gx, gw, gb, ws, n, h, w, cin, ctx.alpha)
        _sinks_written()
        return gx, gw_ret, gb_ret, None, None


def disp_head(x, weight, bias, alpha, offset=3.0):
    return _DispHead.apply(x, weight, bias, alpha, offset)


# --------------------------------------------------------------------------------------------------
# general convolution family (DIS-SF / DispNetS): streaming implicit GEMM, csrc/conv_gen.hip
# --------------------------------------------------------------------------------------------------
CONVG_CONV, CONVG_CONV_DGRAD, CONVG_TCONV, CONVG_TCONV_DGRAD = 0, 1, 2, 3


def _ld(t):
    """pixel stride (floats) of an nhwc tensor that is dense or a channel range of a wider dense nhwc buffer
    (ConcatBuf.slot), None for any other layout"""
    if t.dim() == 4 and t.is_contiguous():
        return t.shape[3]
    if t.dim() != 4 or t.stride(3) != 1:
        return None
    ld = t.stride(2)
    n, h, w, c = t.shape
    if (ld < c or ld % 4 or t.stride(1) != w * ld or t.stride(0) != h * w * ld or t.data_ptr() % 16):
        return None
    return ld


def _nhwc(t):
    """t as the kernels can address it: itself if _ld(t) applies, else a dense copy"""
    return t if _ld(t) is not None else t.contiguous()


class ConcatBuf(object):
    """Channel concatenation without the copy (the reference concatenates decoder inputs with torch.cat,
    model/networks.py:262-288): ONE nhwc buffer per concatenation, zero-padded to a multiple of 4 channels; every
    producer writes its channel range in place (convg(..., out=buf.slot(off, c)), write_channels) and hands the range
    on as an ordinary tensor (a view); joined(parts) returns the whole buffer as the consumer's input and routes the
    gradient ranges back to the parts.  All writes go through the C ABI, never through ATen in-place ops: the views
    saved for backward keep their version."""

    def __init__(self, n, h, w, c, device, dtype=torch.float32):
        self.c = c
        q = 8 if dtype == torch.bfloat16 else 4   # 16-byte vectors: 4 floats / 8 bf16
        ld = (c + q - 1) // q * q
        self.q = q
        self.buf = torch.empty((n, h, w, ld), dtype=dtype, device=device)
        self.pad = ld - c  # zero lanes behind the last channel: written together with it (write_channels)

    def slot(self, off, c):
        assert off % self.q == 0 and off + c <= self.c
        return (self.buf, off, c)

    def joined(self, parts):
        """parts: [(tensor written into the buffer, channel offset)]"""
        return _CatFrom.apply(self, tuple(o for _, o in parts), *[p for p, _ in parts])


class _CatFrom(torch.autograd.Function):
    @staticmethod
    def forward(ctx, cbuf, offs, *parts):
        ctx.offs = offs
        ctx.cs = tuple(p.shape[3] for p in parts)
        return cbuf.buf.view(cbuf.buf.shape)

    @staticmethod
    def backward(ctx, g):
        return (None, None) + tuple(g[..., o:o + c] for o, c in zip(ctx.offs, ctx.cs))


class _WriteChannels(torch.autograd.Function):
    @staticmethod
    def forward(ctx, src, slot, zero_tail):
        buf, off, c = slot
        n, h, w, cs = src.shape
        assert cs == c and tuple(buf.shape[:3]) == (n, h, w)
        src = _nhwc(src)
        if not src.is_cuda or src.dtype != torch.float32:
            raise RuntimeError('depthinspace_amd ops need float32 CUDA(HIP) tensors: the HIP path is the only path')
        dst = buf[..., off:off + c]
        lib.call('dis_copy_channels', src, _ld(src), dst, buf.shape[3], n * h * w, c,
                 buf.shape[3] - off - c if zero_tail else 0)
        return dst

    @staticmethod
    def backward(ctx, g):
        return g, None, None


def write_channels(src, slot, zero_tail=False):
    """copy the nhwc tensor src into a ConcatBuf slot; returns the slot's view (differentiable).  zero_tail: also zero the
    buffer's padding lanes behind the slot (the slot must be the concatenation's last part)."""
    if slot[0].dtype == torch.bfloat16:
        return _WriteChannelsB.apply(src, slot, zero_tail)
    return _WriteChannels.apply(src, slot, zero_tail)


def _convg_run(mode, x, w, bias, y, n, hin, win, cin, cin_w, hout, wout, cout, cout_w, k, stride, pad, act):
    per = lib.fn('dis_convg_pack_workspace')(cin, cout, k)
    if per < 0:
        raise lib.DisHipError(f'convg: unsupported shape cin={cin} cout={cout} k={k}')
    phases = 4 if (mode in (CONVG_CONV_DGRAD, CONVG_TCONV) and stride == 2) else 1
    sk = lib.fn('dis_convg_splitk_workspace')(mode, n, hin, win, hout, wout, cin, cout, k, stride, pad)   # (small maps: split-K partials)
    wp = torch.empty(per * phases + max(sk, 0), dtype=torch.float32, device=x.device)
    lib.call('dis_convg_run', mode, x, _ld(x), 0, w, bias, y, _ld(y), 0, wp, n, hin, win, cin, cin_w, hout,
             wout, cout, cout_w, k, stride, pad, act)


def _convg_wgrad(X, hX, wX, cX, cX_w, G, hG, wG, cG, cG_w, gw, n, k, stride, pad):
    wsz = lib.fn('dis_convg_wgrad_workspace')(n, hG, wG, cX, cG, k)
    if wsz < 0:
        raise lib.DisHipError('convg wgrad: unsupported shape')
    ws = torch.empty(wsz, dtype=torch.float32, device=X.device)
    lib.call('dis_convg_wgrad', X, _ld(X), 0, hX, wX, cX, cX_w, G, _ld(G), 0, hG, wG, cG, cG_w, gw, ws, n, k,
             stride, pad)


def _colsum(G, c_real, out=None):
    """sum over all pixels of the first c_real channels of an nhwc tensor (into `out` if given)"""
    ld = _ld(G)
    npix = G.shape[0] * G.shape[1] * G.shape[2]
    if out is None:
        out = torch.empty(c_real, dtype=torch.float32, device=G.device)
    ws = torch.empty(lib.fn('dis_colsum_workspace')(c_real), dtype=torch.float32, device=G.device)
    lib.call('dis_colsum', G, ld, 0, npix, c_real, out, ws)
    return out


class _ConvG(torch.autograd.Function):
    """Conv2d / ConvTranspose2d(k,s2,p,op=1, cropped to out_hw) + bias + activation on nhwc tensors.
    x (n,h,w,cin_mem): cin_mem % 4 == 0 and >= the weight's input channels (extra channels must be zero-weight
    padding lanes; their gradient is returned as zeros)."""

    @staticmethod
    def forward(ctx, x, weight, bias, stride, pad, act, transposed, out_hw, need_dgrad, out=None):
        x, weight = _nhwc(x), _c(weight)  # (x may be a channel range of a ConcatBuf)
        _chk(weight, bias)
        if not x.is_cuda or x.dtype != torch.float32:
            raise RuntimeError('depthinspace_amd ops need float32 CUDA(HIP) tensors: the HIP path is the only path')
        n, hin, win, cin_mem = x.shape
        if transposed:
            cin_w, cout, k, _ = weight.shape
            hout, wout = out_hw
            if stride != 2 or hout > 2 * hin or wout > 2 * win:
                raise RuntimeError('transposed conv: stride 2 and out_hw <= 2x input expected')
        else:
            cout, cin_w, k, _ = weight.shape
            hout, wout = (hin + 2 * pad - k) // stride + 1, (win + 2 * pad - k) // stride + 1
        if cin_mem % 4 or cin_w > cin_mem or cout % 4:
            raise RuntimeError(f'convg: bad channel counts cin_mem={cin_mem} cin_w={cin_w} cout={cout}')
        if out is None:
            y = torch.empty((n, hout, wout, cout), dtype=torch.float32, device=x.device)
        else:  # write into a ConcatBuf slot: (buffer, channel offset, channels)
            buf, off, c = out
            if c != cout or tuple(buf.shape[:3]) != (n, hout, wout):
                raise RuntimeError(f'convg: out slot {tuple(buf.shape)}[{off}:{off + c}] does not fit ({n},{hout},{wout},{cout})')
            y = buf[..., off:off + c]
        _convg_run(CONVG_TCONV if transposed else CONVG_CONV, x, weight, bias, y, n, hin, win, cin_mem, cin_w, hout,
                   wout, cout, cout, k, stride, pad, act)
        ctx.save_for_backward(x, weight, y if act != ACT_NONE else None)
        ctx.bias_ref = bias  # (gradient sink lookup)
        ctx.cfg = (stride, pad, act, transposed, bias is not None, need_dgrad)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, weight, y = ctx.saved_tensors
        stride, pad, act, transposed, has_bias, need_dgrad = ctx.cfg
        n, hin, win, cin_mem = x.shape
        gy = _nhwc(gy)
        _, hout, wout, cout = gy.shape
        k = weight.shape[2]
        cin_w = weight.shape[0] if transposed else weight.shape[1]
        # (shapes whose weight gradient runs on the one-pass kernels of conv2d.hip get their bias gradient from that pass)
        onepass = (not transposed and x.is_contiguous()
                   and lib.fn('dis_conv2d_wgrad_workspace')(cin_mem, cout, k, stride) >= 0)
        gb, gb_ret = _sink_opt(ctx.bias_ref)
        gb_done = False
        if act != ACT_NONE or not gy.is_contiguous():
            # (gy / y may be channel ranges of wider buffers: the gradient of a concatenation, an output written into one)
            gpre = torch.empty(gy.shape, dtype=torch.float32, device=gy.device)
            if has_bias and not onepass and cout <= 1024 and 256 % (cout // 4) == 0 and gy.data_ptr() % 16 == 0 \
                    and (y is None or y.data_ptr() % 16 == 0):
                # activation gradient and bias gradient from one pass over gy
                gb_done = True
                ws = torch.empty(lib.fn('dis_act_bwd_ld_bias_workspace')(cout), dtype=torch.float32, device=gy.device)
                lib.call('dis_act_bwd_ld_bias', gy, _ld(gy), y, _ld(y) if y is not None else 0, gpre, act, n * hout * wout,
                         cout, gb, ws)
            elif gy.is_contiguous() and (y is None or y.is_contiguous()):
                lib.call('dis_act_bwd', gy, y, gpre, act, gy.numel())
            else:
                lib.call('dis_act_bwd_ld', gy, _ld(gy), y, _ld(y) if y is not None else 0, gpre, act, n * hout * wout, cout)
        else:
            gpre = gy
        gx = None
        if need_dgrad and ctx.needs_input_grad[0]:
            gx = torch.empty(x.shape, dtype=torch.float32, device=x.device)
            _convg_run(CONVG_TCONV_DGRAD if transposed else CONVG_CONV_DGRAD, gpre, weight, None, gx, n, hout, wout,
                       cout, cout, hin, win, cin_mem, cin_w, k, stride, pad, ACT_NONE)
        gw, gw_ret = _sink(weight)
        if transposed:
            _convg_wgrad(gpre, hout, wout, cout, cout, x, hin, win, cin_mem, cin_w, gw, n, k, stride, pad)
        else:
            # shapes whose whole (tap, cin) x cout accumulator fits a workgroup's registers go through the one-pass kernels
            # of conv2d.hip (x and gy are read once instead of once per tap; the bias gradient comes out of the same pass)
            if onepass:
                wsz = lib.fn('dis_conv2d_wgrad_workspace')(cin_mem, cout, k, stride)
                ws = torch.empty(wsz, dtype=torch.float32, device=x.device)
                _conv_wgrad_any(x, gpre, gw, gb, ws, n, hin, win, cin_mem, cin_w, cout, k, stride, pad)
                _sinks_written()
                return gx, gw_ret, gb_ret, None, None, None, None, None, None, None
            _convg_wgrad(x, hin, win, cin_mem, cin_w, gpre, hout, wout, cout, cout, gw, n, k, stride, pad)
        if has_bias and not gb_done:
            _colsum(gpre, cout, out=gb)
        _sinks_written()
        return gx, gw_ret, gb_ret, None, None, None, None, None, None, None


def convg(x, weight, bias, stride=1, pad=0, act=ACT_NONE, need_dgrad=True, out=None, dtype=torch.float32):
    """Conv2d on an nhwc tensor through the streaming MFMA kernel (any channel count that is a multiple of 4).
    out: optional ConcatBuf.slot the result is written into (the returned tensor is that channel range).
    dtype=torch.bfloat16: bf16 activation storage (the result is bf16; x bf16, or fp32 for the network input)."""
    if dtype == torch.bfloat16:
        return _ConvB.apply(x, weight, bias, stride, pad, act, False, None, need_dgrad, out)
    return _ConvG.apply(x, weight, bias, stride, pad, act, False, None, need_dgrad, out)


def convg_transposed(x, weight, bias, out_hw, pad=1, act=ACT_NONE, out=None, dtype=torch.float32):
    """ConvTranspose2d(k=3, stride=2, padding=pad, output_padding=1) cropped to out_hw (crop_like)."""
    if dtype == torch.bfloat16:
        return _ConvB.apply(x, weight, bias, 2, pad, act, True, tuple(out_hw), True, out)
    return _ConvG.apply(x, weight, bias, 2, pad, act, True, tuple(out_hw), True, out)


class _HeadG(torch.autograd.Function):
    """Conv2d(cin,1,3,pad 1) + alpha*sigmoid(. - offset) for any cin % 4 == 0: x nhwc -> y planar (n,1,h,w)."""

    @staticmethod
    def forward(ctx, x, weight, bias, alpha, offset):
        x, weight, bias = _c(x), _c(weight), _c(bias)
        _chk(x, weight, bias)
        n, h, w, cin = x.shape
        k = weight.shape[2]
        y = torch.empty((n, 1, h, w), dtype=torch.float32, device=x.device)
        ctx.direct = k == 3 and cin in (16, 32) and weight.shape[1] == cin  # the one-pass head kernels (16 / 32 channels)
        if ctx.direct:
            lib.call('dis_disp_head_fwd', x, weight, bias, y, n, h, w, cin, float(alpha), float(offset))
        else:
            _convg_run(CONVG_CONV, x, weight, bias, y.view(n, h, w, 1), n, h, w, cin, weight.shape[1], h, w, 1, 1, k, 1,
                       k // 2, ACT_NONE)
            lib.call('dis_sigmoid_affine_fwd', y, y, float(alpha), float(offset), y.numel())
        ctx.save_for_backward(x, weight, y)
        ctx.alpha = float(alpha)
        ctx.bias_ref = bias
        return y

    @staticmethod
    def backward(ctx, gy):
        x, weight, y = ctx.saved_tensors
        n, h, w, cin = x.shape
        k = weight.shape[2]
        cin_w = weight.shape[1]
        gw, gw_ret = _sink(weight)
        gb, gb_ret = _sink(ctx.bias_ref)
        if ctx.direct:
            gx = torch.empty_like(x)
            ws = torch.empty(lib.fn('dis_disp_head_bwd_workspace')(n, h, w, cin), dtype=torch.float32, device=x.device)
            lib.call('dis_disp_head_bwd', x, weight, y, _c(gy), gx, gw, gb, ws, n, h, w, cin, ctx.alpha)
            _sinks_written()
            return gx, gw_ret, gb_ret, None, None
        gpre4 = torch.empty((n, h, w, 4), dtype=torch.float32, device=x.device)
        lib.call('dis_sigmoid_affine_bwd', y, _c(gy), gpre4, ctx.alpha, n * h * w)
        gx = torch.empty_like(x)
        _convg_run(CONVG_CONV_DGRAD, gpre4, weight, None, gx, n, h, w, 4, 1, h, w, cin, cin_w, k, 1, k // 2, ACT_NONE)
        _convg_wgrad(x, h, w, cin, cin_w, gpre4, h, w, 4, 1, gw, n, k, 1, k // 2)
        _colsum(gpre4, 1, out=gb)
        _sinks_written()
        return gx, gw_ret, gb_ret, None, None


def disp_head_g(x, weight, bias, alpha, offset=3.0):
    if x.dtype == torch.bfloat16:
        return _HeadB.apply(x, weight, bias, alpha, offset)
    return _HeadG.apply(x, weight, bias, alpha, offset)


# --------------------------------------------------------------------------------------------------
# DispNetS with bf16 activation storage (BASELINE config 2; csrc/conv_bf16.hip): nhwc feature maps are torch.bfloat16,
# parameters / parameter gradients / disparities stay float32.  One bf16 product per MAC, fp32 accumulation.
# --------------------------------------------------------------------------------------------------
BF16 = torch.bfloat16


def _isbf(t):
    return 1 if t.dtype == BF16 else 0


def _chk_act(*ts):
    for t in ts:
        if t is None:
            continue
        if not t.is_cuda or t.dtype not in (torch.float32, BF16):
            raise RuntimeError('depthinspace_amd bf16 ops need float32 / bfloat16 CUDA(HIP) tensors: the HIP path is the only path')
        if _ld(t) is None or (t.dtype == BF16 and _ld(t) % 8):
            raise RuntimeError('expected a dense nhwc tensor or a channel range of one (bf16: channels in multiples of 8)')


CONVB_PREPACKED = 0x100   # include/dis_hip.h DIS_CONVB_PREPACKED


def _convb_run(mode, x, w, bias, y, n, hin, win, cin, cin_w, hout, wout, cout, cout_w, k, stride, pad, act):
    per = lib.fn('dis_convb_pack_workspace')(cin, cout, k)
    if per < 0:
        raise lib.DisHipError(f'convb: unsupported shape cin={cin} cout={cout} k={k}')
    sk = lib.fn('dis_convb_splitk_workspace')(mode, _isbf(x), n, hin, win, hout, wout, cin, cout, k, stride, pad)   # floats
    key = (w.data_ptr(), mode, _isbf(x), _isbf(y), n, hin, win, cin, cin_w, hout, wout, cout, cout_w, k, stride, pad)
    wp, prepacked = _PackBatch.workspace(key, per * 4 + 2 * max(sk, 0), x.device, w)
    ldy = y.stride(2) if y.dim() == 4 else 1
    lib.call('dis_convb_run', mode | (CONVB_PREPACKED if prepacked else 0), x, _isbf(x), _ld(x), 0, w, bias, y, _isbf(y), ldy, 0,
             wp, n, hin, win, cin, cin_w, hout, wout, cout, cout_w, k, stride, pad, act)


def _convb_wgrad(X, hX, wX, cX, cX_w, G, hG, wG, cG, cG_w, gw, n, k, stride, pad):
    wsz = lib.fn('dis_convb_wgrad_workspace')(n, hG, wG, cX, cG, k)
    if wsz < 0:
        raise lib.DisHipError('convb wgrad: unsupported shape')
    ws = torch.empty(wsz, dtype=torch.float32, device=X.device)
    lib.call('dis_convb_wgrad', X, _isbf(X), _ld(X), 0, hX, wX, cX, cX_w, G, _isbf(G), _ld(G), 0, hG, wG, cG, cG_w, gw, ws,
             n, k, stride, pad)


def _colsum_b(G, c_real, out=None):
    npix = G.shape[0] * G.shape[1] * G.shape[2]
    if out is None:
        out = torch.empty(c_real, dtype=torch.float32, device=G.device)
    ws = torch.empty(lib.fn('dis_colsum_bf16_workspace')(c_real), dtype=torch.float32, device=G.device)
    lib.call('dis_colsum_bf16', G, _ld(G), 0, npix, c_real, out, ws)
    return out


class _ConvB(torch.autograd.Function):
    """_ConvG with bf16 activation storage: x fp32 (the network input) or bf16, y bf16 (or a slot of a bf16 ConcatBuf)."""

    @staticmethod
    def forward(ctx, x, weight, bias, stride, pad, act, transposed, out_hw, need_dgrad, out=None):
        x, weight = _nhwc(x), _c(weight)
        _chk(weight, bias)
        _chk_act(x)
        n, hin, win, cin_mem = x.shape
        if transposed:
            cin_w, cout, k, _ = weight.shape
            hout, wout = out_hw
            if stride != 2 or hout > 2 * hin or wout > 2 * win:
                raise RuntimeError('transposed conv: stride 2 and out_hw <= 2x input expected')
        else:
            cout, cin_w, k, _ = weight.shape
            hout, wout = (hin + 2 * pad - k) // stride + 1, (win + 2 * pad - k) // stride + 1
        if cin_w > cin_mem or cout % 8:
            raise RuntimeError(f'convb: bad channel counts cin_mem={cin_mem} cin_w={cin_w} cout={cout}')
        if out is None:
            y = torch.empty((n, hout, wout, cout), dtype=BF16, device=x.device)
        else:
            buf, off, c = out
            if c != cout or tuple(buf.shape[:3]) != (n, hout, wout) or buf.dtype != BF16:
                raise RuntimeError(f'convb: out slot {tuple(buf.shape)}[{off}:{off + c}] does not fit ({n},{hout},{wout},{cout})')
            y = buf[..., off:off + c]
        _convb_run(CONVG_TCONV if transposed else CONVG_CONV, x, weight, bias, y, n, hin, win, cin_mem, cin_w, hout, wout,
                   cout, cout, k, stride, pad, act)
        ctx.save_for_backward(x, weight, y if act != ACT_NONE else None)
        ctx.bias_ref = bias  # (gradient sink lookup)
        ctx.cfg = (stride, pad, act, transposed, bias is not None, need_dgrad)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, weight, y = ctx.saved_tensors
        stride, pad, act, transposed, has_bias, need_dgrad = ctx.cfg
        n, hin, win, cin_mem = x.shape
        gy = _nhwc(gy)
        _, hout, wout, cout = gy.shape
        k = weight.shape[2]
        cin_w = weight.shape[0] if transposed else weight.shape[1]
        if x.dtype == torch.float32 and not transposed and not (need_dgrad and ctx.needs_input_grad[0]) and x.is_contiguous():
            # the network's first layer (fp32 input, no input gradient): fp32 pre-activation gradient and the one-pass fp32
            # weight-gradient kernel of conv2d.hip (x and gy read once for all taps, bias gradient from the same pass)
            wsz = lib.fn('dis_conv2d_wgrad_workspace')(cin_mem, cout, k, stride)
            if wsz >= 0:
                gpre = torch.empty(gy.shape, dtype=torch.float32, device=gy.device)
                lib.call('dis_act_bwd_bf16_f32', gy, _ld(gy), y, _ld(y) if y is not None else 0, gpre, act, n * hout * wout,
                         cout)
                gw, gw_ret = _sink(weight)
                ws = torch.empty(wsz, dtype=torch.float32, device=x.device)
                gb, gb_ret = _sink_opt(ctx.bias_ref)
                _conv_wgrad_any(x, gpre, gw, gb, ws, n, hin, win, cin_mem, cin_w, cout, k, stride, pad)
                _sinks_written()
                return None, gw_ret, gb_ret, None, None, None, None, None, None, None
        gb, gb_ret = _sink_opt(ctx.bias_ref)
        gb_done = False
        if act != ACT_NONE or not gy.is_contiguous():
            gpre = torch.empty(gy.shape, dtype=BF16, device=gy.device)
            if has_bias and cout <= 1024 and 256 % (cout // 4) == 0:   # bias gradient from the same pass over gy
                gb_done = True
                ws = torch.empty(lib.fn('dis_colsum_bf16_workspace')(cout), dtype=torch.float32, device=gy.device)
                lib.call('dis_act_bwd_bf16_bias', gy, _ld(gy), y, _ld(y) if y is not None else 0, gpre, act,
                         n * hout * wout, cout, gb, ws)
            else:
                lib.call('dis_act_bwd_bf16', gy, _ld(gy), y, _ld(y) if y is not None else 0, gpre, act, n * hout * wout,
                         cout)
        else:
            gpre = gy
        gx = None
        if need_dgrad and ctx.needs_input_grad[0]:
            gx = torch.empty(x.shape, dtype=x.dtype, device=x.device)
            _convb_run(CONVG_TCONV_DGRAD if transposed else CONVG_CONV_DGRAD, gpre, weight, None, gx, n, hout, wout, cout,
                       cout, hin, win, cin_mem, cin_w, k, stride, pad, ACT_NONE)
        gw, gw_ret = _sink(weight)
        if transposed:
            _convb_wgrad(gpre, hout, wout, cout, cout, x, hin, win, cin_mem, cin_w, gw, n, k, stride, pad)
        else:
            _convb_wgrad(x, hin, win, cin_mem, cin_w, gpre, hout, wout, cout, cout, gw, n, k, stride, pad)
        if has_bias and not gb_done:
            _colsum_b(gpre, cout, out=gb)
        _sinks_written()
        return gx, gw_ret, gb_ret, None, None, None, None, None, None, None


class _HeadB(torch.autograd.Function):
    """_HeadG on a bf16 feature map: Conv2d(cin,1,3,pad 1) + alpha*sigmoid(. - offset); the disparity is float32."""

    @staticmethod
    def forward(ctx, x, weight, bias, alpha, offset):
        x, weight, bias = _c(x), _c(weight), _c(bias)
        _chk(weight, bias)
        _chk_act(x)
        n, h, w, cin = x.shape
        k = weight.shape[2]
        y = torch.empty((n, 1, h, w), dtype=torch.float32, device=x.device)
        _convb_run(CONVG_CONV, x, weight, bias, y.view(n, h, w, 1), n, h, w, cin, weight.shape[1], h, w, 1, 1, k, 1, k // 2,
                   ACT_NONE)
        lib.call('dis_sigmoid_affine_fwd', y, y, float(alpha), float(offset), y.numel())
        ctx.save_for_backward(x, weight, y)
        ctx.alpha = float(alpha)
        ctx.bias_ref = bias
        return y

    @staticmethod
    def backward(ctx, gy):
        x, weight, y = ctx.saved_tensors
        n, h, w, cin = x.shape
        k = weight.shape[2]
        cin_w = weight.shape[1]
        gpre4 = torch.empty((n, h, w, 4), dtype=torch.float32, device=x.device)
        lib.call('dis_sigmoid_affine_bwd', y, _c(gy), gpre4, ctx.alpha, n * h * w)
        gx = torch.empty_like(x)
        _convb_run(CONVG_CONV_DGRAD, gpre4, weight, None, gx, n, h, w, 4, 1, h, w, cin, cin_w, k, 1, k // 2, ACT_NONE)
        gw, gw_ret = _sink(weight)
        gb, gb_ret = _sink(ctx.bias_ref)
        _convb_wgrad(x, h, w, cin, cin_w, gpre4, h, w, 4, 1, gw, n, k, 1, k // 2)
        _colsum(gpre4, 1, out=gb)
        _sinks_written()
        return gx, gw_ret, gb_ret, None, None


class _WriteChannelsB(torch.autograd.Function):
    @staticmethod
    def forward(ctx, src, slot, zero_tail):
        buf, off, c = slot
        n, h, w, cs = src.shape
        assert cs == c and tuple(buf.shape[:3]) == (n, h, w) and buf.dtype == BF16
        src = _nhwc(src)
        _chk_act(src) if src.dtype == BF16 else None
        dst = buf[..., off:off + c]
        lib.call('dis_copy_channels_bf16', src, _isbf(src), src.stride(2), dst, buf.shape[3], n * h * w, c,
                 buf.shape[3] - off - c if zero_tail else 0)
        ctx.src_dtype = src.dtype
        return dst

    @staticmethod
    def backward(ctx, g):
        return (g if ctx.src_dtype == BF16 else g.float()), None, None


# --------------------------------------------------------------------------------------------------
# GroupNorm(1 group) (+ residual + activation)
# --------------------------------------------------------------------------------------------------
class _GroupNorm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, stats, gamma, beta, residual, act, eps, in_act=ACT_NONE, join=None, x_lazy=False, defer=False):
        ctx.x_lazy = x_lazy   # the producer of x redeems a token for the elementwise backward pass (_GN_LAZY)
        x = _c(x)
        residual = _c(residual) if residual is not None else None
        _chk(x, gamma, beta, residual)
        n = x.shape[0]
        c = x.shape[-1]
        hw = x.numel() // (n * c)
        if stats is None:
            stats = _zeros_d(2 * n, x.device)
            lib.call('dis_gn_stats', x, stats, n, hw * c)
        y = torch.empty_like(x)
        if not defer:   # (defer: the 3x3 conv that consumes y first writes it, see _GN_PENDING)
            lib.call('dis_gn_apply', x, stats, gamma, beta, residual, y, n, hw, c, act, float(eps))
        ctx.save_for_backward(x, stats, gamma, y if act != ACT_NONE else None)
        ctx.cfg = (n, hw, c, act, float(eps), residual is not None, in_act)
        ctx.beta_ref = beta
        ctx.join = join
        return y

    @staticmethod
    def backward(ctx, gy):
        x, stats, gamma, y = ctx.saved_tensors
        n, hw, c, act, eps, has_res, in_act = ctx.cfg
        gy = _c(gy)
        gg, gg_ret = _sink(gamma)
        gb, gb_ret = _sink(ctx.beta_ref)
        pre = _GN_PRE.pop(gy.data_ptr(), None)
        if pre is not None and not has_res and act == ACT_NONE:
            # (a plain GroupNorm output with two consumers: the consumer whose backward ran second left the sums of the complete g)
            ab, slots = pre
            if ctx.x_lazy and GN_LAZY:
                gx = _gn_lazy_defer(gy, x, stats, gamma, ab, slots, gg, gb, n, hw, c, eps, in_act, uid=ctx.x_lazy)
            else:
                gx = torch.empty_like(x)
                coef = torch.empty(n * (c + 2) + 4 * n * c + 2, dtype=torch.float32, device=x.device)
                lib.call('dis_gn_bwd_from_sums', gy, x, stats, gamma, ab, slots, gx, gg, gb, coef, n, hw, c, eps, in_act)
            _sinks_written()
            return gx, None, gg_ret, gb_ret, None, None, None, None, None, None, None
        if pre is not None and has_res and act == ACT_SELU and in_act == ACT_NONE:
            # gy already IS the gradient wrt the pre-activation value (the producing input-gradient launch multiplied by
            # SELU'(y) and left the channel sums): it doubles as the residual gradient, and one elementwise pass gives gx
            ab, slots = pre
            if ctx.x_lazy and GN_LAZY:
                gx = _gn_lazy_defer(gy.view(x.shape), x, stats, gamma, ab, slots, gg, gb, n, hw, c, eps, ACT_NONE, uid=ctx.x_lazy)
            else:
                gx = torch.empty_like(x)
                coef = torch.empty(n * (c + 2) + 4 * n * c + 2, dtype=torch.float32, device=x.device)
                lib.call('dis_gn_bwd_from_sums', gy, x, stats, gamma, ab, slots, gx, gg, gb, coef, n, hw, c, eps, ACT_NONE)
            gres = gy.view(x.shape)
            if ctx.join is not None:
                if ctx.join.buf is None:
                    gres = ctx.join.first(gres)
                else:
                    gres = ctx.join.take(gres.shape).add_(gres)
            _sinks_written()
            return gx, None, gg_ret, gb_ret, gres, None, None, None, None, None, None
        if pre is not None:
            # the producer has ALREADY turned gy into the pre-activation gradient and formed its channel sums: the generic
            # two-pass form below would apply act' a second time.  No networks here reach this; a new caller must not do so silently.
            raise RuntimeError(f'group_norm backward: channel sums were registered for this gradient but no from-sums form matches '
                               f'(residual={has_res}, act={act}, in_act={in_act})')
        if (ctx.x_lazy and GN_LAZY and GN_RES_SUMS and c in (16, 32) and x.dim() == 4 and (act == ACT_NONE or has_res) and
                (in_act == ACT_NONE or act == ACT_NONE) and lib.fn('dis_get_conv_split')() == 1):
            # nobody left channel sums for this gradient (it comes from a join, a resize, a feature warp), but the producer of x
            # redeems tokens: ONE pass forms g = gy act'(y) (the residual gradient, stored), and the sums of g and g x; the
            # elementwise pass rides on the producer's input-gradient launch (instead of a reduce and an apply launch over gy, y, x)
            slots = lib.fn('dis_conv2d_gnsums_slots')()
            ab = _zeros_d(n * slots * 2 * c, x.device)
            gres = torch.empty_like(x) if (has_res or act != ACT_NONE) else None
            lib.call('dis_gn_bwd_res_sums', gy, y, x, gres, ab, slots, n, hw, c, act)
            g_ = gres if gres is not None else gy.view(x.shape)
            gx = _gn_lazy_defer(g_, x, stats, gamma, ab, slots, gg, gb, n, hw, c, eps, in_act, uid=ctx.x_lazy)
            if not has_res:
                gres = None
        else:
            gx = torch.empty_like(x)
            gres = torch.empty_like(x) if has_res else None
            wtot = lib.fn('dis_gn_bwd_workspace')(n, c)
            ws = torch.empty(wtot, dtype=torch.float64, device=x.device)
            nred2 = wtot // (2 + 2 * c) * 2  # (n * blocks-per-sample-max) pairs of per-block sums, then the parameter partials
            red, pacc = ws[:nred2], ws[nred2:]
            lib.call('dis_gn_apply_bwd', gy, y, x, stats, gamma, gx, gres, gg, gb, red, pacc, n, hw, c, act, eps, in_act)
        if gres is not None and ctx.join is not None:
            if ctx.join.buf is None:
                gres = ctx.join.first(gres)
            else:  # the other consumer ran first (not the case in the networks here): plain accumulation
                gres = ctx.join.take(gres.shape).add_(gres)
        _sinks_written()
        return gx, None, gg_ret, gb_ret, gres, None, None, None, None, None, None


def c_ok(x):
    return x.shape[-1] in (16, 32)


GN_DEFER = _os.environ.get('DIS_GN_DEFER', '1') != '0'
GN_RES_SUMS = _os.environ.get('DIS_GN_RES_SUMS', '1') != '0'   # =0: dis_gn_apply_bwd (reduce + apply launches) where no conv left the channel sums


def group_norm(x, gamma, beta, stats=None, residual=None, act=ACT_NONE, eps=1e-5, in_act=ACT_NONE, join=None, defer=False):
    """GroupNorm(1 group) over all but the first dim of an nhwc tensor; y = act(gn(x) (+ residual)).
    in_act: x is the output of that activation (conv2d(..., act, gy_is_pre=True)); the backward then returns the
    gradient wrt the producer's pre-activation output.
    defer (residual + SELU form with known statistics only): y is NOT written here; the caller guarantees that its first reader is
    conv2d(y, ...) - which forms the values on load and stores them - or calls ops.realize(y) (see _GN_PENDING)."""
    defer = bool(defer and GN_DEFER and residual is not None and act == ACT_SELU and in_act == ACT_NONE and stats is not None and
                 x.dim() == 4 and c_ok(x) and BF16X3)
    y = _GroupNorm.apply(x, stats, gamma, beta, residual, act, eps, in_act, join, int(getattr(x, '_gn_lazy_ok', 0)), defer)
    if defer:
        y._pending_gn = (_c(x), stats, gamma, beta, _c(residual), float(eps))
        _GN_PENDING[y.data_ptr()] = True
    if residual is not None and act == ACT_SELU and in_act == ACT_NONE:
        y._gn_res_src = (x,)   # (a ResNetBlock that takes y as its input hands this to its first conv: conv2d(gnres=...))
    elif residual is None and act == ACT_NONE and c_ok(x):
        y._gn_plain_src = (x, None)   # (two-consumer form: Block2D3D hands it to the consumer whose backward runs second)
    return y


# --------------------------------------------------------------------------------------------------
# multi-frame: gathered/warped features, geometry, slot weighting, Conv3D
# --------------------------------------------------------------------------------------------------
# DIS_GATHER_GNRES=0: the feature warp's backward leaves the plain gradient and the GroupNorm behind it runs dis_gn_bwd_res_sums (A/B)
GATHER_GNRES = _os_env.environ.get('DIS_GATHER_GNRES', '1') != '0'


class _GatherWarpedFeat(torch.autograd.Function):
    @staticmethod
    def forward(ctx, feat, flows, csr, join, gnres=None):
        feat, flows = _c(feat), _c(flows)
        _chk(feat, flows)
        tl, bs, h, w, c = feat.shape
        assert flows.shape == (tl * tl, bs, h, w, 2), flows.shape
        out = torch.empty((tl, bs, h, w, tl, c), dtype=torch.float32, device=feat.device)
        lib.call('dis_gather_warped_feat_fwd', feat, flows, out, tl, bs, h, w, c)
        # gnres = (x2,): feat IS y = SELU(GroupNorm(x2) + residual) and the other consumer of y shares `join`: the backward pass that
        # runs second completes the gradient wrt y - if that is this one, it also does the GroupNorm backward's first pass (below)
        ctx.gnres = bool(gnres is not None and join is not None and csr is not None and GATHER_GNRES and GN_RES_SUMS and GN_LAZY and
                         2 * c <= 64 and tuple(gnres[0].shape[-3:]) == (h, w, c) and gnres[0].numel() == feat.numel())
        if ctx.gnres:
            ctx.save_for_backward(flows, csr, feat, _c(gnres[0]))
        else:
            ctx.save_for_backward(flows, csr)
        ctx.shape = feat.shape
        ctx.join = join
        return out

    @staticmethod
    def backward(ctx, g):
        flows, csr = ctx.saved_tensors[:2]
        tl, bs, h, w, c = ctx.shape
        join = ctx.join
        second = join is not None and join.buf is not None
        if csr is not None:
            init = join.take(ctx.shape) if second else None
            gf = init if second else torch.empty(ctx.shape, dtype=torch.float32, device=g.device)
            done = False
            if ctx.gnres and second and lib.fn('dis_get_conv_split')() == 1:
                # this launch completes the gradient wrt y = SELU(GroupNorm(x2) + residual): it stores g SELU'(y) and leaves the channel
                # sums of the GroupNorm's backward (_GroupNorm.backward finds them: no pass over g, y, x2 of its own)
                y, x2 = ctx.saved_tensors[2:4]
                slots = lib.fn('dis_conv2d_gnsums_slots')()
                ab = _zeros_d(tl * bs * slots * 2 * c, g.device)
                if lib.call_try('dis_gather_warped_feat_bwd_csr_gnres', _c(g), csr, init, gf, y, x2, ab, slots, ACT_SELU, tl, bs, h, w, c):
                    _GN_PRE[gf.data_ptr()] = (ab, slots)
                    done = True
            if not done:
                lib.call('dis_gather_warped_feat_bwd_csr', _c(g), csr, init, gf, tl, bs, h, w, c)
        else:
            gf = torch.empty(ctx.shape, dtype=torch.float32, device=g.device)
            lib.call('dis_gather_warped_feat_bwd', _c(g), flows, gf, tl, bs, h, w, c)
            if second:
                gf.add_(join.take(ctx.shape))
        if join is not None and not second:
            gf = join.first(gf)
        return gf, None, None, None, None


def gather_csr(flows):
    """Scatter index of gather_warped_feat for `flows` (tl*tl,bs,h,w,2): build once per step and pass to every
    gather_warped_feat call that uses these flows; the backward then runs without atomics (deterministic)."""
    flows = _c(flows)
    _chk(flows)
    t2, bs, h, w, _ = flows.shape
    tl = int(round(t2 ** 0.5))
    n = lib.fn('dis_gather_csr_workspace')(tl, bs, h, w)
    if n < 0:
        raise lib.DisHipError('gather_csr: unsupported shape')
    csr = torch.empty(n, dtype=torch.int32, device=flows.device)
    lib.call('dis_gather_csr_build', flows, csr, tl, bs, h, w)
    return csr


def gather_warped_feat(feat, flows, csr=None, join=None, gnres=None):
    """gnres = (x2,): feat is SELU(GroupNorm(x2) + residual) (group_norm's `_gn_res_src`) whose other consumer shares `join`"""
    return _GatherWarpedFeat.apply(feat, flows, csr, join, gnres)


def mf_geometry(depth_core, R, t, flows_core, Kinv, u_step, v_step):
    depth_core, R, t, flows_core = _c(depth_core), _c(R), _c(t), _c(flows_core)
    _chk(depth_core, R, t, flows_core)
    tl, bs, h, w = depth_core.shape
    out = torch.empty((tl, bs, h, w, tl, 4), dtype=torch.float32, device=depth_core.device)
    lib.call('dis_mf_geometry', depth_core, R, t, flows_core, Kinv, int(u_step), int(v_step), out, tl, bs, h, w)
    return out


def mf_geometry_resize(geom, size):
    tl, bs, hin, win, _, _ = geom.shape
    out = torch.empty((tl, bs, size[0], size[1], tl, 4), dtype=torch.float32, device=geom.device)
    lib.call('dis_mf_geometry_resize', geom, out, tl, bs, hin, win, size[0], size[1])
    return out


class _MaskWeightSlots(torch.autograd.Function):
    @staticmethod
    def forward(ctx, wf, geom, join):
        wf, geom = _c(wf), _c(geom)
        tl, bs, h, w, s, c = wf.shape
        out = torch.empty_like(wf)
        lib.call('dis_mask_weight_slots', wf, geom, out, tl * bs * h * w, s, c, 0)
        ctx.save_for_backward(geom)
        ctx.join = join
        return out

    @staticmethod
    def backward(ctx, g):
        (geom,) = ctx.saved_tensors
        g = _c(g)
        tl, bs, h, w, s, c = g.shape
        join = ctx.join
        second = join is not None and join.buf is not None
        out = join.take(g.shape) if second else torch.empty_like(g)
        lib.call('dis_mask_weight_slots', g, geom, out, tl * bs * h * w, s, c, 1 if second else 0)
        if join is not None and not second:
            out = join.first(out)
        return out, None, None


def mask_weight_slots(wf, geom, join=None):
    return _MaskWeightSlots.apply(wf, geom, join)


# DIS_CONV3D_CSR=1: Conv3D feature gradient as a fixed-order gather (bitwise reproducible) instead of the float-atomic scatter.
# Off by default: measured on MI355X it costs 1.0 ms of the 42 ms DIS-MF step (3.04 + 0.37 ms build vs 2.41 ms; 369 vs 378
# frames/s) - the scatter is 9 whole 128-byte rows per output pixel, which L2 atomics handle at the rate of the plain traffic
# the staged form needs.
CONV3D_CSR = _os.environ.get('DIS_CONV3D_CSR', '0') == '1'
# DIS_CONV3D_BWD: which backward Conv3D runs.
#   det (default)  class-ordered plain read-modify-write of the feature-gradient rows (dis_conv3d_knn_bwd_det): bitwise
#                  reproducible, no index structure; the forward keeps its aggregate (128 B per output pixel) for it
#   agg            the same kernel in one launch with the float-atomic scatter (dis_conv3d_knn_bwd_agg)
#   atomic         round 1-2's kernel (recomputes the aggregate; float atomics), csr with DIS_CONV3D_CSR=1
CONV3D_BWD = _os.environ.get('DIS_CONV3D_BWD', 'det')
if CONV3D_CSR:
    CONV3D_BWD = 'atomic'
assert CONV3D_BWD in ('det', 'agg', 'atomic')


def conv3d_select(geom, stride, with_csr=False):
    """top-9 neighbour ids per output pixel (tl,bs,ho,wo,9) uint8; depends on the geometry only.
    with_csr: also build the by-source-row index of the sets (idx.c3csr) that makes the Conv3D feature gradient a fixed-order
    gather instead of a float-atomic scatter; shared by every layer that uses the sets."""
    geom = _c(geom)
    _chk(geom)
    tl, bs, h, wd, s, _ = geom.shape
    ho = (h + 2 - 3) // stride + 1
    wo = (wd + 2 - 3) // stride + 1
    idx = torch.empty((tl, bs, ho, wo, 9), dtype=torch.uint8, device=geom.device)
    lib.call('dis_conv3d_knn_select', geom, idx, tl, bs, h, wd, stride)
    if with_csr and CONV3D_CSR:
        conv3d_csr(idx, h, wd, stride)
    return idx


def conv3d_csr(idx, h, wd, stride):
    """by-source-row CSR of the neighbour sets `idx` (tl,bs,ho,wo,9), attached to the tensor as idx.c3csr"""
    tl, bs = idx.shape[0], idx.shape[1]
    n = lib.fn('dis_conv3d_csr_workspace')(tl, bs, h, wd, stride)
    if n < 0:
        raise lib.DisHipError('conv3d csr: unsupported shape')
    csr = torch.empty(n, dtype=torch.int32, device=idx.device)
    lib.call('dis_conv3d_csr_build', idx, csr, tl, bs, h, wd, stride)
    idx.c3csr = csr
    return csr


class _Conv3dKnn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, geom, wf, d1w, d1b, d2w, d2b, w, idx, stride, join=None):
        geom, wf = _c(geom), _c(wf)
        d1w, d1b, d2w, d2b, w = [_c(p) for p in (d1w, d1b, d2w, d2b, w)]
        _chk(geom, wf, d1w, d1b, d2w, d2b, w)
        tl, bs, h, wd, s, c = wf.shape
        assert c == 32 and s == tl
        ho = (h + 2 - 3) // stride + 1
        wo = (wd + 2 - 3) // stride + 1
        assert idx.dtype == torch.uint8 and tuple(idx.shape) == (tl, bs, ho, wo, 9) and idx.is_contiguous()
        y = torch.empty((tl, bs, ho, wo, c), dtype=torch.float32, device=wf.device)
        keep_agg = CONV3D_BWD != 'atomic' and any(ctx.needs_input_grad)
        agg = torch.empty_like(y) if keep_agg else None
        lib.call('dis_conv3d_knn_fwd_agg', geom, wf, d1w, d1b, d2w, d2b, w, idx, y, agg, tl, bs, h, wd, stride)
        ctx.save_for_backward(geom, wf, d1w, d1b, d2w, d2b, w, idx, y, agg)
        ctx.stride = stride
        ctx.join = join
        ctx.c3csr = getattr(idx, 'c3csr', None)
        return y

    @staticmethod
    def backward(ctx, gy):
        geom, wf, d1w, d1b, d2w, d2b, w, idx, y, agg = ctx.saved_tensors
        tl, bs, h, wd, s, c = wf.shape
        join = ctx.join
        second = join is not None and join.buf is not None
        sunk = _sink_block((w, d1w, d1b, d2w, d2b))  # the kernel's parameter-gradient block IS the flat buffer's order
        gp = sunk if sunk is not None else torch.empty(1632, dtype=torch.float32, device=wf.device)
        if agg is not None:
            # class-ordered read-modify-write (det) / one launch with float atomics (agg): both ADD to the rows
            gwf = join.take(wf.shape) if second else torch.zeros_like(wf)
            acc = torch.empty(lib.fn('dis_conv3d_knn_bwd_det_workspace')(tl, bs, h, wd, ctx.stride), dtype=torch.float32,
                              device=wf.device)
            lib.call('dis_conv3d_knn_bwd_det' if CONV3D_BWD == 'det' else 'dis_conv3d_knn_bwd_agg', geom, wf, d1w, d1b, d2w, d2b, w,
                     idx, y, agg, _c(gy), gwf, gp, acc, tl, bs, h, wd, ctx.stride)
            if join is not None and not second:
                gwf = join.first(gwf)
            _sinks_written()
            if sunk is not None:
                return (None, gwf, None, None, None, None, None, None, None, None)
            return (None, gwf, gp[1024:1072].view(16, 3), gp[1072:1088], gp[1088:1600].view(32, 16), gp[1600:1632],
                    gp[0:1024].view(32, 32), None, None, None)
        acc = torch.empty(lib.fn('dis_conv3d_knn_bwd_workspace')(), dtype=torch.float32, device=wf.device)
        if ctx.c3csr is not None:
            # deterministic form: per-entry gradient rows staged, then summed per source row in list order
            gwf = join.take(wf.shape) if second else torch.empty_like(wf)
            stage = torch.empty(lib.fn('dis_conv3d_knn_bwd_stage')(tl, bs, h, wd, ctx.stride), dtype=torch.float32,
                                device=wf.device)
            lib.call('dis_conv3d_knn_bwd_csr', geom, wf, d1w, d1b, d2w, d2b, w, idx, y, _c(gy), gwf, gp, acc, ctx.c3csr,
                     stage, 1 if second else 0, tl, bs, h, wd, ctx.stride)
        else:
            gwf = join.take(wf.shape) if second else torch.zeros_like(wf)  # the scatter accumulates (float atomics)
            lib.call('dis_conv3d_knn_bwd', geom, wf, d1w, d1b, d2w, d2b, w, idx, y, _c(gy), gwf, gp, acc, tl, bs, h, wd,
                     ctx.stride)
        if join is not None and not second:
            gwf = join.first(gwf)
        _sinks_written()
        if sunk is not None:
            return (None, gwf, None, None, None, None, None, None, None, None)
        return (None, gwf, gp[1024:1072].view(16, 3), gp[1072:1088], gp[1088:1600].view(32, 16), gp[1600:1632],
                gp[0:1024].view(32, 32), None, None, None)


def conv3d_knn(geom, wf, d1w, d1b, d2w, d2b, w, idx, stride, join=None):
    """y (tl,bs,ho,wo,32) = SELU(agg @ w) for the neighbour sets `idx` (from conv3d_select)."""
    return _Conv3dKnn.apply(geom, wf, d1w, d1b, d2w, d2b, w, idx, stride, join)


# --------------------------------------------------------------------------------------------------
# optimiser
# --------------------------------------------------------------------------------------------------
def adam_step_dev(param, grad, exp_avg, exp_avg_sq, state, lr=1e-4, beta1=0.9, beta2=0.999, eps=1e-8, grad_scale=1.0):
    """Adam with the step counter / bias corrections in the 4 x int32 device tensor `state` (advanced by the call):
    safe to capture in a hipGraph."""
    _chk(param, grad, exp_avg, exp_avg_sq)
    if not state.is_cuda or state.dtype != torch.int32 or state.numel() < 4:
        raise RuntimeError('adam_step_dev: state must be a 4-element int32 CUDA(HIP) tensor')
    lib.call('dis_adam_step_dev', param, grad, exp_avg, exp_avg_sq, param.numel(), float(lr), float(beta1), float(beta2),
             float(eps), state, float(grad_scale))
    params_changed()   # what this step's batch launch packed is stale (inference before the next step packs per call)


def adam_step(param, grad, exp_avg, exp_avg_sq, step, lr=1e-4, beta1=0.9, beta2=0.999, eps=1e-8, grad_scale=1.0):
    _chk(param, grad, exp_avg, exp_avg_sq)
    params_changed()
    lib.call('dis_adam_step', param, grad, exp_avg, exp_avg_sq, param.numel(), float(lr), float(beta1), float(beta2),
             float(eps), int(step), float(grad_scale))
