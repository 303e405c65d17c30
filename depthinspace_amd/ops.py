"""torch.autograd wrappers over the C ABI (include/dis_hip.h).

Each Function allocates outputs/workspaces with torch (the caller owns all memory), launches the HIP
kernels on the current stream through `lib.call`, and wires the hand-written backward kernels.
No ATen compute op is used for the arithmetic of the step; torch is memory + autograd tape only.
"""
import torch
from . import lib

ACT_NONE, ACT_SELU, ACT_RELU = 0, 1, 2
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


def _zeros_d(n, dev):
    return torch.zeros(n, dtype=torch.float64, device=dev)


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


def photometric(es, ta, block_size, type, eps):
    return _Photometric.apply(es, ta, block_size, type, eps)


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
    def forward(ctx, depth0, depth1, flow0, flow1, amb0, amb1, pdepth1, R0, t0, R1, t1, K, Kinv, clamp):
        ts = [_c(t) for t in (depth0, depth1, flow0, flow1, amb0, amb1)]
        depth0, depth1, flow0, flow1, amb0, amb1 = ts
        pdepth1 = _c(pdepth1) if pdepth1 is not None else None
        R0, t0, R1, t1 = _c(R0), _c(t0), _c(R1), _c(t1)
        _chk(depth0, depth1, flow0, flow1, amb0, amb1, pdepth1, R0, t0, R1, t1)
        bs, _, h, w = depth0.shape
        mask = torch.empty_like(depth0)
        acc = _zeros_d(2, depth0.device)
        out = torch.empty((), dtype=torch.float32, device=depth0.device)
        lib.call('dis_geo_loss_fwd', depth0, depth1, flow0, flow1, amb0, amb1, pdepth1, R0, t0, R1, t1, K, Kinv,
                 float(clamp), mask, acc, out, bs, h, w)
        ctx.save_for_backward(depth0, depth1, flow0, R0, t0, R1, t1, mask, acc)
        ctx.cfg = (K, Kinv, float(clamp))
        ctx.mark_non_differentiable(mask)
        return out, mask

    @staticmethod
    def backward(ctx, g, _gmask):
        depth0, depth1, flow0, R0, t0, R1, t1, mask, acc = ctx.saved_tensors
        K, Kinv, clamp = ctx.cfg
        bs, _, h, w = depth0.shape
        g0 = torch.zeros_like(depth0)
        g1 = torch.zeros_like(depth1)
        lib.call('dis_geo_loss_bwd', depth0, depth1, flow0, R0, t0, R1, t1, K, Kinv, clamp, mask, acc, _c(g), g0, g1,
                 bs, h, w)
        return (g0, g1) + (None,) * 12


def geo_loss_dir(depth0, depth1, flow0, flow1, amb0, amb1, pdepth1, R0, t0, R1, t1, K, Kinv, clamp=-1.0):
    """K, Kinv: lib.host_floats(9).  Returns (loss, mask)."""
    return _GeoLossDir.apply(depth0, depth1, flow0, flow1, amb0, amb1, pdepth1, R0, t0, R1, t1, K, Kinv, clamp)
