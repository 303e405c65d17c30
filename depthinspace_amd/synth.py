"""Synthetic default-pattern scenes for tests, parity fixtures and the benchmark.

This is the recipe of SURVEY.md section 8(d): an analytic tilted plane seen by `tl` jittered cameras,
the default dot pattern projected through the rectified stereo geometry, exact rigid optical flow.
It produces a batch in the LOADER layout of the reference (`(bs, tl, ...)`, reference
data/dataset.py:90-125), i.e. what `Worker.copy_data` (reference model/worker.py:418-452) consumes.

Everything here is host-side numpy (float64 internally, float32 out).  It is input generation, not
part of the timed hot path.
"""
import os
import numpy as np

DEFAULT_H, DEFAULT_W = 512, 432
DEFAULT_K = np.array([[435.2, 0.0, 216.0], [0.0, 435.2, 256.0], [0.0, 0.0, 1.0]], dtype=np.float32)
DEFAULT_BASELINE = 0.025
_PLANE_N = np.array([0.15, -0.10, 1.0])
_PLANE_C = 3.0
_BLEND = 0.6

# BASELINE config 5 ("real-pattern dataset"): the camera the reference's settings.pkl holds for --pattern_type real,
# i.e. K_processed and the baseline of reference data/create_syn_data.py:286-296 + data/data_manipulation.py:91-105
# (focal length and principal point of the 1280x1080 sensor after the [128:-128, 108:-108] crop and the 2x down-scale).
REAL_K = np.array([[1112.1806640625 / 2, 0.0, (517.0896606445312 - 108) / 2],
                   [0.0, 1112.1806640625 / 2, (649.6329956054688 - 128) / 2], [0.0, 0.0, 1.0]], dtype=np.float32)
REAL_BASELINE = 0.0246
# second scene type: a non-planar surface  n.X = c + A sin(p X_x) cos(q X_y)
_BUMP_A, _BUMP_P, _BUMP_Q = 0.15, 4.0, 3.0

_here = os.path.dirname(os.path.abspath(__file__))


def load_default_pattern():
    """(512, 432) float32 default dot pattern mapped into the camera (see data/make_pattern_fixture.py)."""
    path = os.path.join(_here, 'data', 'default_pattern_512x432.npz')
    return np.load(path)['pattern'].astype(np.float32)


def load_real_pattern():
    """(512, 432) float32 `real` pattern after the reference's post_process (see data/make_pattern_fixture.py)."""
    path = os.path.join(_here, 'data', 'real_pattern_512x432.npz')
    return np.load(path)['pattern'].astype(np.float32)


def _rodrigues(w):
    th = np.linalg.norm(w)
    Kx = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])
    if th < 1e-12:
        return np.eye(3) + Kx
    return np.eye(3) + np.sin(th) / th * Kx + (1 - np.cos(th)) / th ** 2 * (Kx @ Kx)


def _bilinear_border(img, x, y):
    H, W = img.shape
    x = np.clip(x, 0, W - 1)
    y = np.clip(y, 0, H - 1)
    x0 = np.floor(x).astype(np.int64)
    y0 = np.floor(y).astype(np.int64)
    x1 = np.minimum(x0 + 1, W - 1)
    y1 = np.minimum(y0 + 1, H - 1)
    fx = x - x0
    fy = y - y0
    return (img[y0, x0] * (1 - fx) + img[y0, x1] * fx) * (1 - fy) + (img[y1, x0] * (1 - fx) + img[y1, x1] * fx) * fy


class Settings(object):
    """Equivalent of the reference `settings.pkl` (reference data/create_syn_data.py:332-341)."""

    def __init__(self, imsize, K, baseline, pattern):
        self.imsize = tuple(imsize)
        self.K = np.asarray(K, dtype=np.float32)
        self.baseline = float(baseline)
        self.pattern = pattern  # (H, W, 3) float32


def make_settings(height=DEFAULT_H, width=DEFAULT_W, crop_offset=None, pattern='default'):
    """Settings for the full 512x432 camera or a (height,width) crop of it.

    A crop keeps the focal length and shifts the principal point; crop_offset=(y0,x0) defaults to centred.
    pattern: 'default' (the synthetic default-pattern camera) or 'real' (BASELINE config 5: real pattern, K_processed,
    baseline 0.0246).
    """
    if pattern not in ('default', 'real'):
        raise ValueError(pattern)
    pat = load_default_pattern() if pattern == 'default' else load_real_pattern()
    if crop_offset is None:
        crop_offset = ((DEFAULT_H - height) // 2, (DEFAULT_W - width) // 2)
    y0, x0 = crop_offset
    assert 0 <= y0 and y0 + height <= DEFAULT_H and 0 <= x0 and x0 + width <= DEFAULT_W
    K = (DEFAULT_K if pattern == 'default' else REAL_K).copy()
    K[0, 2] -= x0
    K[1, 2] -= y0
    pat = pat[y0:y0 + height, x0:x0 + width]
    pat3 = np.ascontiguousarray(np.stack([pat, pat, pat], axis=2))
    return Settings((height, width), K, DEFAULT_BASELINE if pattern == 'default' else REAL_BASELINE, pat3)


def _surface_depth(ray, R, t, scene):
    """depth along every ray of camera (R, t) to the scene surface; X_w = (d*ray - t) R."""
    tR = t @ R
    den = (ray @ R) @ _PLANE_N
    d = (_PLANE_C + _PLANE_N @ tR) / den
    if scene == 'plane':
        return d
    for _ in range(60):  # fixed point of  n.X_w(d) = c + bump(X_w(d));  contraction factor <= A*max(p,q)/|n_z| = 0.6
        Xw = (d[:, None] * ray - t[None]) @ R
        bump = _BUMP_A * np.sin(_BUMP_P * Xw[:, 0]) * np.cos(_BUMP_Q * Xw[:, 1])
        d = (_PLANE_C + bump + _PLANE_N @ tR) / den
    return d


def make_batch(settings, bs, tl=4, seed=1234, with_flow=True, with_primary=True, with_pseudo_gt=False, scene='plane',
               motion=1.0):
    """Returns a dict of float32 numpy arrays in loader layout `(bs, tl, ...)`.

    keys: im0, ambient0, disp0 (bs,tl,1,H,W); R (bs,tl,3,3); t (bs,tl,3);
          flow_ij (bs,1,2,H,W) for ordered i!=j; primary_disp, pseudo_gt (bs,tl,1,H,W).
    scene: 'plane'  or 'bumps' (non-planar surface); motion scales the camera jitter
    (rotation +-0.02 rad, translation +-0.05 m at 1.0).  Flows are the exact rigid flow of the surface point
    (occlusions are not modelled).
    """
    if scene not in ('plane', 'bumps'):
        raise ValueError(scene)
    rng = np.random.RandomState(seed)
    H, W = settings.imsize
    K = settings.K.astype(np.float64)
    Ki = np.linalg.inv(K)
    f = float(settings.K[0, 0])
    pat = settings.pattern[..., 0].astype(np.float64)
    u, v = np.meshgrid(np.arange(W, dtype=np.float64), np.arange(H, dtype=np.float64))
    uv1 = np.stack([u, v, np.ones_like(u)], axis=-1).reshape(-1, 3)
    ray = uv1 @ Ki.T  # (HW,3)

    out = {k: np.zeros((bs, tl, 1, H, W), np.float32) for k in ('im0', 'ambient0', 'disp0')}
    out['R'] = np.zeros((bs, tl, 3, 3), np.float32)
    out['t'] = np.zeros((bs, tl, 3), np.float32)
    if with_primary:
        out['primary_disp'] = np.zeros((bs, tl, 1, H, W), np.float32)
    if with_pseudo_gt:
        out['pseudo_gt'] = np.zeros((bs, tl, 1, H, W), np.float32)

    for b in range(bs):
        Rs, ts, depths, xyzw = [], [], [], []
        for i in range(tl):
            R = _rodrigues(rng.uniform(-0.02, 0.02, 3) * motion)
            t = rng.uniform(-0.05, 0.05, 3) * motion
            # X_w = (d*ray - t) R on the surface (plane: n.X_w = c)
            d = _surface_depth(ray, R, t, scene)
            Xw = (d[:, None] * ray - t[None]) @ R
            disp = settings.baseline * f / d
            amb = 0.5 + 0.25 * np.sin(6 * Xw[:, 0]) * np.cos(5 * Xw[:, 1])
            proj = _bilinear_border(pat, u.reshape(-1) - disp, v.reshape(-1))
            im = _BLEND * proj + (1 - _BLEND) * amb + rng.normal(0, 1.0 / 255, size=proj.shape)
            im = np.clip(im, 0, 1)
            out['im0'][b, i, 0] = im.reshape(H, W)
            out['ambient0'][b, i, 0] = amb.reshape(H, W)
            out['disp0'][b, i, 0] = disp.reshape(H, W)
            if with_primary:
                out['primary_disp'][b, i, 0] = (disp + 0.3 * np.sin(20 * u.reshape(-1) / W)).reshape(H, W)
            if with_pseudo_gt:
                out['pseudo_gt'][b, i, 0] = (disp + 0.1 * np.sin(20 * v.reshape(-1) / H)).reshape(H, W)
            out['R'][b, i] = R
            out['t'][b, i] = t
            Rs.append(R); ts.append(t); depths.append(d); xyzw.append(Xw)
        if with_flow:
            for i in range(tl):
                for j in range(tl):
                    if i == j:
                        continue
                    key = f'flow_{i}{j}'
                    if key not in out:
                        out[key] = np.zeros((bs, 1, 2, H, W), np.float32)
                    Xc = xyzw[i] @ Rs[j].T + ts[j][None]
                    uvw = Xc @ K.T
                    uvj = uvw[:, :2] / uvw[:, 2:3]
                    out[key][b, 0, 0] = (uvj[:, 0] - u.reshape(-1)).reshape(H, W)
                    out[key][b, 0, 1] = (uvj[:, 1] - v.reshape(-1)).reshape(H, W)
    return out


def make_random_batch(settings, bs, tl=4, seed=0, with_pseudo_gt=False, flow_scale=1.5):
    """Random-valued (non-physical) batch with the same schema; used for gradient-heavy parity tests
    where every mask/branch should be exercised (random flows, partially failing masks)."""
    rng = np.random.RandomState(seed)
    base = make_batch(settings, bs, tl, seed=seed + 17, with_pseudo_gt=with_pseudo_gt)
    H, W = settings.imsize
    for key in list(base.keys()):
        if key.startswith('flow_'):
            base[key] = base[key] + rng.normal(0, flow_scale, size=base[key].shape).astype(np.float32) * \
                (rng.uniform(size=(bs, 1, 1, H, W)) < 0.3)
    base['primary_disp'] = base['primary_disp'] + rng.normal(0, 0.2, size=base['primary_disp'].shape).astype(np.float32)
    base['ambient0'] = np.clip(base['ambient0'] + rng.normal(0, 0.004, size=base['ambient0'].shape), 0, 1).astype(np.float32)
    return base
