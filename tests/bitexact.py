"""TEST INFRASTRUCTURE: numpy restatement, rounding for rounding, of what torch's CPU kernels compute for the parts of the
step whose results are INDEX-CLASS outputs (north star: "integer/index ops bit-exact"): the thresholded masks of FuseNet
(reference model/multi_frame_networks.py:187-214), Conv3D's neighbour keys and top-9 (:490-498) and the fb / vc / rf
masks of the flow-consistency losses (model/networks.py:564-601, 619-655).

Why it exists: a mask is `a < b` on fp32 values, a neighbour set is an order on fp32 keys, so the HIP kernels can only
reproduce them exactly if every value that reaches a comparison carries the reference's roundings.  Those are not the
naive ones: ATen's CPU kernels are built with FMA contraction, and which operations fuse was established empirically
(each function below is compared with torch bit for bit in tests/test_bitexact_cpu.py):

  * torch.bmm / matmul, K = 3:        c = fma(a2, b2, fma(a1, b1, a0*b0))
  * grid_sample bilinear:              fma(se, w_se, fma(sw, w_sw, fma(ne, w_ne, nw*w_nw))), weights = products of the 1-D
                                       weights, coordinates through the reference's normalise / ATen's unnormalise round trip
  * upsample_bilinear2d(align_corners=True), out_h + out_w <= 128 (ATen's vectorized kernel):
                                       fma(d, w11, fma(c, w10, fma(a, w00, b*w01))), w = ly*lx
                                       otherwise (generic kernel): fma(top, ly0, bot*ly1), top = fma(a, lx0, b*lx1)
  * python_float / tensor:             tensor.reciprocal() * float  (two roundings; DispToDepth, model/networks.py:311-319)
  * topk(k=9 of 36, largest=False, sorted=False): std::nth_element on (key, id) pairs - restated in
                                       depthinspace_amd/csrc/nth_select.h, shared with the HIP kernel

The HIP kernels (layout_ops.hip: resize / mf_geometry; conv3d_knn.hip: select; pixel_ops.hip: geo_loss) spell out the same
chains with explicit __fmaf_rn under -ffp-contract=off, and tests/test_net_ops_gpu.py compares them with THIS file bit for
bit (numpy fp32 arithmetic is IEEE on every host, so the comparison does not depend on the GPU box's CPU).
Only tests/ may import this module.
"""
import ctypes
import os
import subprocess
import numpy as np

f32 = np.float32
_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_LIB = None


def host_lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_ROOT, 'oracle', '_build', 'libdis_host.so')
        if not os.path.exists(path):
            subprocess.check_call(['make', '-C', os.path.join(_ROOT, 'oracle')])
        _LIB = ctypes.CDLL(path)
        _LIB.heapsel_mismatches.restype = ctypes.c_long
    return _LIB


def fma(a, b, c):
    a, b, c = np.broadcast_arrays(np.asarray(a, f32), np.asarray(b, f32), np.asarray(c, f32))
    a, b, c = np.ascontiguousarray(a), np.ascontiguousarray(b), np.ascontiguousarray(c)
    o = np.empty(a.shape, f32)
    P = ctypes.c_void_p
    host_lib().vfma(P(a.ctypes.data), P(b.ctypes.data), P(c.ctypes.data), P(o.ctypes.data), ctypes.c_long(a.size))
    return o


def mul(a, b):
    return (np.asarray(a, f32) * np.asarray(b, f32)).astype(f32)


def add(a, b):
    return (np.asarray(a, f32) + np.asarray(b, f32)).astype(f32)


def sub(a, b):
    return (np.asarray(a, f32) - np.asarray(b, f32)).astype(f32)


def div(a, b):
    return (np.asarray(a, f32) / np.asarray(b, f32)).astype(f32)


# ------------------------------------------------------------------------------------------------ elementary pieces
def disp_to_depth(disp, focal, baseline):
    bf = f32(baseline * focal)
    return mul(div(f32(1), add(np.maximum(disp, f32(0)), f32(1e-12))), bf)


def lerp_idx(nin, nout):
    """ATen compute_source_index_and_lambda, align_corners=True"""
    if nin == nout:
        i = np.arange(nout)
        return i, i, np.ones(nout, f32), np.zeros(nout, f32)
    scale = f32(nin - 1) / f32(nout - 1)
    src = (scale * np.arange(nout, dtype=f32)).astype(f32)
    i0 = np.minimum(np.floor(src).astype(np.int64), nin - 1)
    i1 = np.minimum(i0 + 1, nin - 1)
    l1 = np.clip(src - i0.astype(f32), 0, 1).astype(f32)
    l0 = (f32(1) - l1).astype(f32)
    return i0, i1, l0, l1


def resize_ac(x, hout, wout):
    """F.interpolate(x, (hout, wout), 'bilinear', align_corners=True) over the last two dims"""
    y0, y1, ly0, ly1 = lerp_idx(x.shape[-2], hout)
    x0, x1, lx0, lx1 = lerp_idx(x.shape[-1], wout)
    a = x[..., y0[:, None], x0[None, :]]
    b = x[..., y0[:, None], x1[None, :]]
    c = x[..., y1[:, None], x0[None, :]]
    d = x[..., y1[:, None], x1[None, :]]
    if hout + wout <= 128:  # UpSampleKernel.cpp _use_vectorized_kernel_cond_2d
        w00, w01 = mul(ly0[:, None], lx0), mul(ly0[:, None], lx1)
        w10, w11 = mul(ly1[:, None], lx0), mul(ly1[:, None], lx1)
        return fma(d, w11, fma(c, w10, fma(a, w00, mul(b, w01))))
    top = fma(a, lx0, mul(b, lx1))
    bot = fma(c, lx0, mul(d, lx1))
    return fma(top, ly0[:, None], mul(bot, ly1[:, None]))


def resize_flow(flow, hout, wout):
    """reference resize_flow_like (multi_frame_networks.py:54-68) for one (..,2,h,w) array"""
    h, w = flow.shape[-2:]
    sc = np.array([f32(float(wout) / float(w)), f32(float(hout) / float(h))], f32).reshape(2, 1, 1)
    return mul(resize_ac(flow, hout, wout), sc)


def _roundtrip(p, size):
    g = mul(f32(2), sub(div(p, f32(size - 1)), f32(0.5)))  # the reference's normalisation (networks.py:363-364)
    return mul(add(g, f32(1)), f32(size - 1) / f32(2))     # ATen's unnormalise, align_corners=True


def sample_zeros(img, px, py):
    """grid_sample(img (bs,C,h,w), pixel positions px,py (bs,h',w'), bilinear, zeros, align_corners=True)"""
    bs, C, h, w = img.shape
    ix, iy = _roundtrip(px, w), _roundtrip(py, h)
    x0, y0 = np.floor(ix), np.floor(iy)
    wx = sub(ix, x0)
    ex = sub(f32(1), wx)
    wy = sub(iy, y0)
    ey = sub(f32(1), wy)
    nw, ne, sw, se = mul(ey, ex), mul(ey, wx), mul(wy, ex), mul(wy, wx)
    x0i = np.clip(x0, -4, w + 4).astype(np.int64)
    y0i = np.clip(y0, -4, h + 4).astype(np.int64)

    def tap(xi, yi):
        valid = (xi >= 0) & (xi < w) & (yi >= 0) & (yi < h)
        xc, yc = np.clip(xi, 0, w - 1), np.clip(yi, 0, h - 1)
        b = np.arange(bs)[:, None, None, None]
        c = np.arange(C)[None, :, None, None]
        return np.where(valid[:, None], img[b, c, yc[:, None], xc[:, None]], f32(0)).astype(f32)
    e = mul(tap(x0i, y0i), nw[:, None])
    e = fma(tap(x0i + 1, y0i), ne[:, None], e)
    e = fma(tap(x0i, y0i + 1), sw[:, None], e)
    return fma(tap(x0i + 1, y0i + 1), se[:, None], e)


def warp(x, flow):
    """reference warp (multi_frame_networks.py:83-99)"""
    h, w = x.shape[-2:]
    u, v = np.meshgrid(np.arange(w, dtype=f32), np.arange(h, dtype=f32))
    return sample_zeros(x, add(flow[:, 0], u), add(flow[:, 1], v))


def rowvec_mat(a, M):
    """a (...,3) row vectors @ M (...,3,3) as torch.bmm rounds it"""
    e = mul(a[..., 0:1], M[..., 0, :])
    e = fma(a[..., 1:2], M[..., 1, :], e)
    return fma(a[..., 2:3], M[..., 2, :], e)


def fb_mask(f0, f10, k):
    s = add(f0, f10)
    lhs = add(mul(s[:, 0], s[:, 0]), mul(s[:, 1], s[:, 1]))
    a0 = add(mul(f0[:, 0], f0[:, 0]), mul(f0[:, 1], f0[:, 1]))
    a1 = add(mul(f10[:, 0], f10[:, 0]), mul(f10[:, 1], f10[:, 1]))
    return (lhs < add(f32(0.5), mul(f32(k), add(a0, a1)))).astype(f32)[:, None]


# ------------------------------------------------------------------------------------------------ FuseNet geometry
def mf_geometry(depth_core, ray, R, t, flow_core):
    """oracle.dis_oracle.mf_geometry: depth_core (tl,bs,1,h,w), ray (hw,3), R (tl,bs,3,3), t (tl,bs,3),
    flow_core {flow_ij: (bs,2,h,w)} -> warped xyz (tl,slot,bs,3,h,w), mask (tl,slot,bs,1,h,w)"""
    tl, bs, _, h, w = depth_core.shape
    xyz = mul(depth_core.reshape(tl, bs, -1, 1), ray[None, None])
    xyz = sub(xyz, t.reshape(tl, bs, 1, 3))
    xyz = rowvec_mat(xyz, R[:, :, None])
    allx, allm = [], []
    for ti in range(tl):
        cam = add(rowvec_mat(xyz, np.swapaxes(R[ti], 1, 2)[None, :, None]), t[ti][None, :, None, :])
        img = lambda j: np.swapaxes(cam[j], 1, 2).reshape(bs, 3, h, w)
        xs, ms = [img(ti)], [np.ones((bs, 1, h, w), f32)]
        for j in [j for j in range(tl) if j != ti]:
            f0 = flow_core[f'flow_{ti}{j}']
            xs.append(warp(img(j), f0))
            ms.append(fb_mask(f0, warp(flow_core[f'flow_{j}{ti}'], f0), 0.01))
        allx.append(np.stack(xs, 0))
        allm.append(np.stack(ms, 0))
    return np.stack(allx, 0), np.stack(allm, 0)


def conv3d_keys(xyz, mask, stride, tl=4):
    """Conv3D's 36 candidate keys (multi_frame_networks.py:469-497) for one target: xyz (tl,bs,3,h,w) slots,
    mask (tl,bs,1,h,w) -> dist (bs,ho,wo,36) and valid (bs,ho,wo,36); a masked candidate's key is a fill above all others"""
    def cand(x):
        xp = np.pad(x, ((0, 0), (0, 0), (0, 0), (1, 1), (1, 1)))
        hp, wp = xp.shape[-2:]
        ho, wo = (hp - 3) // stride + 1, (wp - 3) // stride + 1
        cols = []
        for ky in range(3):
            for kx in range(3):
                win = xp[..., ky:ky + stride * (ho - 1) + 1:stride, kx:kx + stride * (wo - 1) + 1:stride]
                cols.append(np.transpose(win, (1, 3, 4, 0, 2)))
        c = np.stack(cols, 3)
        return c.reshape(c.shape[0], ho, wo, 9 * tl, c.shape[-1])
    X, M = cand(xyz), cand(mask)
    plane = div(X, add(X[..., 2:3], f32(1e-12)))
    ctr = (9 // 2) * tl
    pl = sub(plane, plane[..., ctr:ctr + 1, :])
    sq = mul(pl, pl)
    return add(add(sq[..., 0], sq[..., 1]), sq[..., 2]), M[..., 0]


def topk9(keys):
    """what torch.topk(keys (rows,36), 9, largest=False, sorted=False) returns (ids, in its order), through the host
    build of depthinspace_amd/csrc/nth_select.h"""
    keys = np.ascontiguousarray(keys, f32)
    rows, n = keys.shape
    out = np.empty((rows, 9), np.int32)
    host_lib().nthsel_rows(ctypes.c_void_p(keys.ctypes.data), ctypes.c_long(rows), ctypes.c_int(n), ctypes.c_int(9),
                           ctypes.c_void_p(out.ctypes.data))
    return out


def conv3d_select(wxyz, wmask, stride):
    """neighbour ids (tl,bs,ho,wo,9) for all targets: wxyz (tl,slot,bs,3,h,w), wmask (tl,slot,bs,1,h,w)"""
    out = []
    for ti in range(wxyz.shape[0]):
        dist, valid = conv3d_keys(wxyz[ti], wmask[ti], stride)
        key = np.where(valid > 0, dist, np.finfo(f32).max).astype(f32)
        out.append(topk9(key.reshape(-1, key.shape[-1])).reshape(*key.shape[:-1], 9))
    return np.stack(out, 0)


# ------------------------------------------------------------------------------------------------ loss masks
def unproject(depth, ray, R, t):
    bs = depth.shape[0]
    xyz = sub(mul(depth.reshape(bs, -1, 1), ray[None]), t.reshape(bs, 1, 3))
    return rowvec_mat(xyz, R[:, None])


def project(xyz, K, R, t):
    bs = xyz.shape[0]
    x = add(rowvec_mat(xyz, np.swapaxes(R, 1, 2)[:, None]), t.reshape(bs, 1, 3))
    uvw = rowvec_mat(x, np.swapaxes(K[None], 1, 2)[:, None])
    d = uvw[..., 2:3]
    return div(uvw[..., :2], add(np.maximum(d, f32(0)), f32(1e-12))), d


def flow_consistency_mask(K, ray, depth0, R0, t0, R1, t1, flow0, flow1, amb0, amb1, primary_depth1=None):
    """mask of oracle.dis_oracle.flow_consistency_dir (fb * vc [* rf]) and the reprojected depth d1"""
    bs, _, H, W = depth0.shape
    u, v = np.meshgrid(np.arange(W, dtype=f32), np.arange(H, dtype=f32))
    _, d1 = project(unproject(depth0, ray, R0, t0), K, R1, t1)
    px, py = add(flow0[:, 0], u), add(flow0[:, 1], v)
    m = fb_mask(flow0, sample_zeros(flow1, px, py), 0.02)
    m = m * (np.abs(sub(amb0, sample_zeros(amb1, px, py))) < f32(0.01)).astype(f32)
    if primary_depth1 is not None:
        uv0, _ = project(unproject(primary_depth1, ray, R1, t1), K, R0, t0)
        wuv = sample_zeros(np.transpose(uv0.reshape(bs, H, W, 2), (0, 3, 1, 2)), px, py)
        du, dv = sub(wuv[:, 0], u), sub(wuv[:, 1], v)
        m = m * (add(mul(du, du), mul(dv, dv)) < f32(1)).astype(f32)[:, None]
    return m, d1.reshape(bs, 1, H, W)
