"""CPU ORACLE for the DIS-SF / DIS-MF training step.  TEST INFRASTRUCTURE ONLY.

This file is a pure-PyTorch (CPU, fp32) restatement of the arithmetic of idiap/DepthInSpace's
training step.  It is the checker for the HIP path, never the thing shipped or measured:
only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import it.

Parity pin: the reference ships no tests or golden vectors.  This restatement is pinned against
the reference itself, imported on CPU in the build container by `oracle/make_golden.py`
(the recipe of SURVEY.md Appendix D); the resulting vectors are committed under `tests/golden/`.
The one third-party op on the path, connecting_the_dots' `photometric_loss_forward/backward`
(un-vendored, unpinned `git clone`), is restated from the reference's own in-repo description
`photometric_loss_pytorch` (reference model/ext_functions.py:156-183); parity for that single op
against the CTD binary is therefore UNPINNED (source absent), everything else is pinned.

Style: functional.  Network weights live in a flat dict {state_dict key: tensor} whose keys are the
reference's `state_dict()` keys, so that reference checkpoints map 1:1.

Reference line numbers are relative to /root/reference.
"""
import math
import numpy as np
import torch
import torch.nn.functional as F

SELU_ALPHA = 1.6732632423543772848170429916717
SELU_SCALE = 1.0507009873554804934193349852946


# ----------------------------------------------------------------------------------------------
# per-pixel operators
# ----------------------------------------------------------------------------------------------
def lcn(x, radius=5, eps=0.05):
    """Local contrast normalisation.  reference model/networks.py:663-689.
    x (N,1,H,W) -> (lcn, std)."""
    k = 2 * radius + 1
    ones = torch.ones(1, 1, k, k, dtype=x.dtype)
    xp = F.pad(x, (radius,) * 4, mode='reflect')
    box = F.conv2d(xp, ones)
    box2 = F.conv2d(xp * xp, ones)
    avg = box / k ** 2
    std = torch.sqrt(torch.clamp(box2 / k ** 2 - avg ** 2 + 1e-6, min=0)) + eps
    return (x - avg) / std, std


def soft_census(d, eps):
    return 0.5 * (1 + d / torch.sqrt(d * d + eps))


PHOTO_TYPES = {'mse': 0, 'sad': 1, 'census_mse': 2, 'census_sad': 3}


def photometric(es, ta, block=9, type='census_sad', eps=0.5):
    """Windowed photometric difference.  reference model/ext_functions.py:156-183 (the in-repo
    statement of CTD's ext op) ; called from model/networks.py:372 with (9,'census_sad',0.5).
    es, ta (N,C,H,W) -> (N,1,H,W).  Window offsets are accumulated one at a time (no 81x unfold)."""
    if isinstance(type, int):
        type = {v: k for k, v in PHOTO_TYPES.items()}[type]
    p = block // 2
    N, C, H, W = es.shape
    esp = F.pad(es, (p,) * 4, mode='replicate')
    tap = F.pad(ta, (p,) * 4, mode='replicate')
    acc = torch.zeros(N, 1, H, W, dtype=es.dtype)
    for dy in range(block):
        for dx in range(block):
            e = esp[:, :, dy:dy + H, dx:dx + W]
            t = tap[:, :, dy:dy + H, dx:dx + W]
            if type == 'mse':
                r = (e - t) ** 2
            elif type == 'sad':
                r = (e - t).abs()
            else:
                diff = soft_census(e - es, eps) - soft_census(t - ta, eps)
                r = diff * diff if type == 'census_mse' else diff.abs()
            acc = acc + r.sum(dim=1, keepdim=True)
    return acc / block ** 2


def pixel_grid(H, W):
    v, u = torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(W, dtype=torch.float32), indexing='ij')
    return u, v


def sample_at(x, px, py, padding):
    """Bilinear sample of x (N,C,H,W) at pixel coordinates px,py (N,H',W'); align_corners=True
    normalisation as the reference writes it: g = 2*(p/(size-1) - 0.5)  (networks.py:363-364)."""
    H, W = x.shape[-2:]
    gx = 2 * (px / (W - 1) - 0.5)
    gy = 2 * (py / (H - 1) - 0.5)
    grid = torch.stack((gx, gy), dim=-1)
    return F.grid_sample(x, grid, mode='bilinear', padding_mode=padding, align_corners=True)


def warp(x, flow):
    """reference model/multi_frame_networks.py:83-99: sample x at (u,v)+flow, zeros padding."""
    H, W = x.shape[-2:]
    u, v = pixel_grid(H, W)
    return sample_at(x, flow[:, 0] + u, flow[:, 1] + v, 'zeros')


def resize_ac(x, size):
    """bilinear, align_corners=True, over the last two dims (any leading dims).
    reference model/multi_frame_networks.py:42-51."""
    lead = x.shape[:-3]
    y = F.interpolate(x.reshape(-1, *x.shape[-3:]), size=size, mode='bilinear', align_corners=True)
    return y.reshape(*lead, *y.shape[1:])


def resize_flow(flow, size):
    """reference model/multi_frame_networks.py:54-68 (flow magnitudes rescaled with the grid)."""
    out = {}
    for k, f in flow.items():
        h, w = f.shape[-2:]
        r = F.interpolate(f, size=size, mode='bilinear', align_corners=True)
        scale = torch.tensor([float(size[1]) / float(w), float(size[0]) / float(h)], dtype=f.dtype).view(1, 2, 1, 1)
        out[k] = r * scale
    return out


def disp_to_depth(disp, focal, baseline):
    """reference model/networks.py:311-319"""
    return (baseline * focal) / (F.relu(disp) + 1e-12)


def pattern_loss(pattern, disp, im, std=None, block=9, type='census_sad', eps=0.5):
    """RectifiedPatternSimilarityLoss.  reference model/networks.py:336-377.
    pattern (1,1,H,W) (already LCN'd, single channel); disp, im, std (N,1,H,W)."""
    N, _, H, W = disp.shape
    u, v = pixel_grid(H, W)
    proj = sample_at(pattern.expand(N, -1, -1, -1), u - disp[:, 0], v.expand(N, -1, -1), 'border')
    diff = photometric(proj, im, block, type, eps)
    mask = torch.ones_like(im)
    if std is not None:
        mask = mask * std
    return (mask * diff).sum() / mask.sum(), proj


_SOBEL5 = np.array([[-5, -4, 0, 4, 5], [-8, -10, 0, 10, 8], [-10, -20, 0, 20, 10],
                    [-8, -10, 0, 10, 8], [-5, -4, 0, 4, 5]]) / 240.0


def sobel5(x):
    """reference model/networks.py:693-731 (ksize 5, norm False): replicate pad 2, returns cat(gx,gy)."""
    kx = torch.from_numpy(_SOBEL5).float().view(1, 1, 5, 5)
    ky = torch.from_numpy(_SOBEL5.T.copy()).float().view(1, 1, 5, 5)
    xp = F.pad(x, (2, 2, 2, 2), mode='replicate')
    return torch.cat((F.conv2d(xp, kx), F.conv2d(xp, ky)), dim=1)


def smooth_loss(disp, amb):
    """DisparitySmoothLoss.  reference model/networks.py:411-431."""
    g = sobel5(disp)
    ga = sobel5(amb)
    return (g * torch.exp(-(255 * ga).abs())).abs().mean()


# ----------------------------------------------------------------------------------------------
# projection geometry (row-vector convention, reference model/networks.py:433-493)
# ----------------------------------------------------------------------------------------------
def make_rays(K, H, W, us=None, vs=None):
    """ray = [u,v,1] . K^-T, built as int64 grid @ float32 K^-1 (promoted to float64) then cast to
    float32, exactly like reference networks.py:445-451 / multi_frame_networks.py:121-128."""
    Ki = np.linalg.inv(np.asarray(K, dtype=np.float32))
    if us is None:
        us = np.arange(W)
    if vs is None:
        vs = np.arange(H)
    u, v = np.meshgrid(us, vs)
    uv = np.stack((u, v, np.ones_like(u)), axis=2).reshape(-1, 3)
    ray = (uv @ Ki.T).astype(np.float32)
    return torch.from_numpy(ray)  # (HW,3)


def unproject(depth, ray, R, t):
    """X_w = (depth*ray - t) R.   depth (bs,1,H,W) -> (bs,HW,3)"""
    bs = depth.shape[0]
    xyz = depth.reshape(bs, -1, 1) * ray.unsqueeze(0)
    xyz = xyz - t.reshape(bs, 1, 3)
    return torch.bmm(xyz, R)


def project(xyz, K, R, t):
    """X_c = X_w R^T + t ; uvw = X_c K^T ; uv = uvw[:2]/(relu(w)+1e-12) ; d = w"""
    bs = xyz.shape[0]
    xyz = torch.bmm(xyz, R.transpose(1, 2)) + t.reshape(bs, 1, 3)
    uvw = torch.bmm(xyz, K.view(1, 3, 3).transpose(1, 2).expand(bs, -1, -1))
    d = uvw[:, :, 2:3]
    uv = uvw[:, :, :2] / (F.relu(d) + 1e-12)
    return uv, d


def flow_consistency_dir(K, ray, depth0, depth1, R0, t0, R1, t1, flow0, flow1, amb0, amb1,
                         primary_depth1=None, clamp=None):
    """One direction of the geometric loss.
    multi-frame (primary_depth1 given): reference model/networks.py:564-601
    single-frame (clamp given):         reference model/networks.py:619-655"""
    bs, _, H, W = depth0.shape
    u, v = pixel_grid(H, W)
    _, d1 = project(unproject(depth0, ray, R0, t0), K, R1, t1)
    d1 = d1.view(bs, 1, H, W)
    px = flow0[:, 0] + u
    py = flow0[:, 1] + v
    depth10 = sample_at(depth1, px, py, 'zeros')
    diff = (d1 - depth10).abs()
    if clamp is not None and clamp > 0:
        diff = torch.clamp(diff, 0, clamp)
    with torch.no_grad():
        flow10 = sample_at(flow1, px, py, 'zeros')
        fb = ((flow0 + flow10) ** 2).sum(dim=1) < 0.5 + 0.02 * ((flow0 ** 2).sum(dim=1) + (flow10 ** 2).sum(dim=1))
        fb = fb.float().unsqueeze(1)
        amb10 = sample_at(amb1, px, py, 'zeros')
        vc = ((amb0 - amb10).abs().mean(dim=1, keepdim=True) < 0.01).float()
        mask = fb * vc
        if primary_depth1 is not None:
            uv0, _ = project(unproject(primary_depth1, ray, R1, t1), K, R0, t0)
            uv0 = uv0.view(bs, H, W, 2).permute(0, 3, 1, 2)
            wuv0 = sample_at(uv0, px, py, 'zeros')
            rf = (((wuv0 - torch.stack([u, v], dim=0).unsqueeze(0)) ** 2).sum(dim=1, keepdim=True) < 1).float()
            mask = mask * rf
    return (diff * mask).sum() / (mask.sum() + 1e-8), mask


def mf_flow_consistency(K, ray, depth0, depth1, R0, t0, R1, t1, flow0, flow1, amb0, amb1, pd0, pd1):
    """Multi_Frame_Flow_Consistency_Loss.tforward, reference model/networks.py:603-607"""
    l0, _ = flow_consistency_dir(K, ray, depth0, depth1, R0, t0, R1, t1, flow0, flow1, amb0, amb1, primary_depth1=pd1)
    l1, _ = flow_consistency_dir(K, ray, depth1, depth0, R1, t1, R0, t0, flow1, flow0, amb1, amb0, primary_depth1=pd0)
    return l0 + l1


def sf_flow_consistency(K, ray, depth0, depth1, R0, t0, R1, t1, flow0, flow1, amb0, amb1, clamp=0.1):
    """Single_Frame_Flow_Consistency_Loss.tforward, reference model/networks.py:657-661"""
    l0, m0 = flow_consistency_dir(K, ray, depth0, depth1, R0, t0, R1, t1, flow0, flow1, amb0, amb1, clamp=clamp)
    l1, m1 = flow_consistency_dir(K, ray, depth1, depth0, R1, t1, R0, t0, flow1, flow0, amb1, amb0, clamp=clamp)
    return l0 + l1, m0, m1


# ----------------------------------------------------------------------------------------------
# DIS-SF network (DispNetS inside DispDecoder), reference model/networks.py:170-309
# ----------------------------------------------------------------------------------------------
_SF_ENC = [32, 64, 128, 256, 512, 512, 512]
_SF_UP = [512, 512, 256, 128, 64, 32, 16]


def sf_param_shapes(channels_in=2):
    """{state_dict key: shape} of DispDecoder (64 tensors)."""
    P = 'disp_decoder.'
    s = {}
    cin = channels_in
    for i, (c, k) in enumerate(zip(_SF_ENC, [7, 5, 3, 3, 3, 3, 3])):
        s[f'{P}conv{i + 1}.0.weight'] = (c, cin, k, k); s[f'{P}conv{i + 1}.0.bias'] = (c,)
        s[f'{P}conv{i + 1}.2.weight'] = (c, c, k, k); s[f'{P}conv{i + 1}.2.bias'] = (c,)
        cin = c
    ups_in = [_SF_ENC[6]] + _SF_UP[:6]
    for j, lvl in enumerate(range(7, 0, -1)):
        s[f'{P}upconv{lvl}.0.weight'] = (ups_in[j], _SF_UP[j], 3, 3); s[f'{P}upconv{lvl}.0.bias'] = (_SF_UP[j],)
    icin = {7: _SF_UP[0] + _SF_ENC[5], 6: _SF_UP[1] + _SF_ENC[4], 5: _SF_UP[2] + _SF_ENC[3], 4: _SF_UP[3] + _SF_ENC[2],
            3: 1 + _SF_UP[4] + _SF_ENC[1], 2: 1 + _SF_UP[5] + _SF_ENC[0], 1: 1 + _SF_UP[6]}
    for j, lvl in enumerate(range(7, 0, -1)):
        s[f'{P}iconv{lvl}.0.weight'] = (_SF_UP[j], icin[lvl], 3, 3); s[f'{P}iconv{lvl}.0.bias'] = (_SF_UP[j],)
    for lvl, c in zip([4, 3, 2, 1], [_SF_UP[3], _SF_UP[4], _SF_UP[5], _SF_UP[6]]):
        s[f'{P}predict_disp{lvl}.0.weight'] = (1, c, 3, 3); s[f'{P}predict_disp{lvl}.0.bias'] = (1,)
    return s


def sf_forward(p, x, max_disp=128):
    """DispDecoder.forward: x (N,2,H,W) -> 4 full-resolution disparities."""
    P = 'disp_decoder.'

    def c(name, t, stride=1):
        w = p[P + name + '.weight']
        return F.conv2d(t, w, p[P + name + '.bias'], stride=stride, padding=(w.shape[-1] - 1) // 2)

    def up(name, t, like):
        y = F.relu(F.conv_transpose2d(t, p[P + name + '.0.weight'], p[P + name + '.0.bias'], stride=2, padding=1,
                                      output_padding=1))
        return y[:, :, :like.shape[2], :like.shape[3]]

    def head(lvl, t, s):
        return (max_disp / 2 ** s) * torch.sigmoid(c(f'predict_disp{lvl}.0', t) - 3)

    def up2(d, like):
        y = F.interpolate(d, scale_factor=2, mode='bilinear', align_corners=False)
        return y[:, :, :like.shape[2], :like.shape[3]]

    enc = []
    t = x
    for i in range(1, 8):
        t = F.relu(c(f'conv{i}.0', t, stride=2))
        t = F.relu(c(f'conv{i}.2', t))
        enc.append(t)
    e1, e2, e3, e4, e5, e6, e7 = enc
    i7 = F.relu(c('iconv7.0', torch.cat((up('upconv7', e7, e6), e6), 1)))
    i6 = F.relu(c('iconv6.0', torch.cat((up('upconv6', i7, e5), e5), 1)))
    i5 = F.relu(c('iconv5.0', torch.cat((up('upconv5', i6, e4), e4), 1)))
    i4 = F.relu(c('iconv4.0', torch.cat((up('upconv4', i5, e3), e3), 1)))
    d4 = head(4, i4, 3)
    i3 = F.relu(c('iconv3.0', torch.cat((up('upconv3', i4, e2), e2, up2(d4, e2)), 1)))
    d3 = head(3, i3, 2)
    i2 = F.relu(c('iconv2.0', torch.cat((up('upconv2', i3, e1), e1, up2(d3, e1)), 1)))
    d2 = head(2, i2, 1)
    i1 = F.relu(c('iconv1.0', torch.cat((up('upconv1', i2, x), up2(d2, x)), 1)))
    d1 = head(1, i1, 0)
    size = d1.shape[-2:]
    rs = lambda d: F.interpolate(d, size=size, mode='bilinear', align_corners=False)
    return d1, rs(d2), rs(d3), rs(d4)


# ----------------------------------------------------------------------------------------------
# DIS-MF network (FuseNet), reference model/multi_frame_networks.py:101-542
# ----------------------------------------------------------------------------------------------
def mf_param_shapes(channels=32, tl=4, block_num=4):
    """{state_dict key: shape} of FuseNet (236 tensors)."""
    C = channels
    s = {}

    def conv(name, cin, cout, k):
        s[name + '.weight'] = (cout, cin, k, k); s[name + '.bias'] = (cout,)

    def gn(name, c):
        s[name + '.weight'] = (c,); s[name + '.bias'] = (c,)

    def res(name, c):
        conv(name + '.conv1', c, c, 3); gn(name + '.bn1', c); conv(name + '.conv2', c, c, 3); gn(name + '.bn2', c)

    conv('conv1.1', 4, C // 2, 4); conv('conv2.1', C // 2, C // 2, 3); conv('conv3.1', C // 2, C, 3); conv('conv4.1', C, C, 3)
    for r in ('res1', 'res2', 'res3'):
        res(r, C)
    for b in range(block_num):
        B = f'blocks.{b}.'
        conv(B + 'conv_mf.1', C * tl, C, 1); gn(B + 'conv_mf.2', C)
        for n, k in (('conv1_1', 3), ('conv1_2', 3), ('conv2_1', 4), ('conv2_2', 3)):
            conv(B + n + '.1', C, C, k); gn(B + n + '.3', C)
        conv(B + 'conv_fuse.1', 3 * C, C, 3); gn(B + 'conv_fuse.2', C)
        for n in ('conv3d_1', 'conv3d_2'):
            s[B + n + '.w'] = (C, C)
            s[B + n + '.dense1.0.weight'] = (C // 2, 3); s[B + n + '.dense1.0.bias'] = (C // 2,)
            s[B + n + '.dense2.0.weight'] = (C, C // 2); s[B + n + '.dense2.0.bias'] = (C,)
            gn(B + n + '.bn', C)
    for n in ('upconv1.0', 'upconv2.0'):   # constructed but never used (reference :143-144)
        s[n + '.weight'] = (C, C, 4, 4); s[n + '.bias'] = (C,)
    conv('amb_conv.1', 1, 16, 3); res('amb_res1', 16); res('amb_res2', 16)
    conv('ref_conv.1', 16 + C, 32, 3)
    for r in ('ref_res1', 'ref_res2', 'ref_res3'):
        res(r, 32)
    conv('final_conv.1', 32, 16, 3); conv('predict_disp.0', 16, 1, 3)
    return s


def _pconv(p, name, x, stride=1):
    """explicit ZeroPad2d((k-1)//2) followed by an unpadded conv (reference :159-164)."""
    w = p[name + '.weight']
    pad = (w.shape[-1] - 1) // 2
    return F.conv2d(F.pad(x, (pad,) * 4), w, p[name + '.bias'], stride=stride)


def _gn(p, name, x):
    return F.group_norm(x, 1, p[name + '.weight'], p[name + '.bias'], eps=1e-5)


def _resblock(p, name, x):
    """reference model/multi_frame_networks.py:514-542 (SELU before the first norm)."""
    o = _gn(p, name + '.bn1', F.selu(_pconv(p, name + '.conv1', x)))
    o = _gn(p, name + '.bn2', _pconv(p, name + '.conv2', o))
    return F.selu(o + x)


# test hook: when set to a list, every conv3d_knn call appends {'name','idx','key'} (neighbour ids and keys)
CONV3D_TAP = None
# test hook: when set to a dict {layer-geometry 'core'|'quarter': LongTensor (tl,bs,ho,wo,9)}, conv3d_knn uses
# these neighbour sets instead of its own top-k (used to propagate one selection through perturbed inputs)
CONV3D_FORCE = None


def conv3d_knn(p, name, xyz, feat, mask, stride, tl=4, neighbors=9, return_index=False, target=None):
    """Conv3D.tforward.  reference model/multi_frame_networks.py:469-512.
    xyz (tl,bs,3,h,w), feat (tl,bs,C,h,w), mask (tl,bs,1,h,w) -> (bs,C,h',w')."""
    def candidates(x):
        # zero pad 1, 3x3 window (stride s), candidate order (ky, kx, frame) with frame fastest
        xp = F.pad(x, (1, 1, 1, 1))
        hp, wp = xp.shape[-2:]
        ho = (hp - 3) // stride + 1
        wo = (wp - 3) // stride + 1
        cols = []
        for ky in range(3):
            for kx in range(3):
                win = xp[..., ky:ky + stride * (ho - 1) + 1:stride, kx:kx + stride * (wo - 1) + 1:stride]  # (tl,bs,c,ho,wo)
                cols.append(win.permute(1, 3, 4, 0, 2))  # (bs,ho,wo,tl,c)
        c = torch.stack(cols, dim=3)  # (bs,ho,wo,9,tl,c)
        return c.reshape(-1, 9 * tl, c.shape[-1]), (c.shape[0], ho, wo)

    X, bhw = candidates(xyz)
    Fe, _ = candidates(feat)
    M, _ = candidates(mask)
    plane = X / (X[..., 2:] + 1e-12)
    ctr = (9 // 2) * tl
    local = X - X[:, ctr:ctr + 1]
    plane_local = plane - plane[:, ctr:ctr + 1]
    dist = (plane_local ** 2).sum(dim=-1, keepdim=True)
    key = M * dist + (1 - M) * (dist.max() + 1)
    _, idx = torch.topk(key, neighbors, dim=1, largest=False, sorted=False)
    if CONV3D_FORCE is not None and target is not None:
        idx = CONV3D_FORCE['core' if stride == 2 else 'quarter'][target].reshape(-1, neighbors, 1).long()
    if CONV3D_TAP is not None:
        CONV3D_TAP.append({'name': name, 'target': target, 'idx': idx.view(*bhw, neighbors).clone(),
                           'key': key.view(*bhw, 9 * tl).clone(), 'plane_absmax': float(plane.abs().max())})
    nb_xyz = torch.gather(local, 1, idx.expand(-1, -1, 3))
    nb_feat = torch.gather(Fe, 1, idx.expand(-1, -1, Fe.shape[-1]))
    h1 = F.selu(F.linear(nb_xyz, p[name + '.dense1.0.weight'], p[name + '.dense1.0.bias']))
    h2 = F.selu(F.linear(h1, p[name + '.dense2.0.weight'], p[name + '.dense2.0.bias']))
    agg = (h2 * nb_feat).sum(dim=1)
    out = torch.matmul(agg, p[name + '.w']).view(*bhw, -1).permute(0, 3, 1, 2)
    out = _gn(p, name + '.bn', F.selu(out))
    if return_index:
        return out, idx.view(*bhw, neighbors), key.view(*bhw, 9 * tl)
    return out


def _other_frames(t, tl):
    return [j for j in range(tl) if j != t]


def mf_geometry(depth_core, ray_core, R, t, flow_core):
    """unproject + change_view_angle + gather_warped_xyz for every target frame.
    reference model/multi_frame_networks.py:172-214, 283-294.
    depth_core (tl,bs,1,h,w) -> warped_xyz (tl,4,bs,3,h,w), warped_mask (tl,4,bs,1,h,w)  [no grad]"""
    tl, bs, _, h, w = depth_core.shape
    xyz = depth_core.reshape(tl, bs, -1, 1) * ray_core.view(1, 1, -1, 3)
    xyz = torch.matmul(xyz - t.view(tl, bs, 1, 3), R)
    all_xyz, all_mask = [], []
    for ti in range(tl):
        cam = torch.matmul(xyz, R[ti].transpose(1, 2)) + t[ti].unsqueeze(1).unsqueeze(0)  # (tl,bs,hw,3)
        img = lambda j: cam[j].transpose(1, 2).reshape(bs, 3, h, w)
        xs = [img(ti)]
        ms = [torch.ones(bs, 1, h, w)]
        for j in _other_frames(ti, tl):
            f0 = flow_core[f'flow_{ti}{j}']
            xs.append(warp(img(j), f0))
            f10 = warp(flow_core[f'flow_{j}{ti}'], f0)
            fb = ((f0 + f10) ** 2).sum(dim=1) < 0.5 + 0.01 * ((f0 ** 2).sum(dim=1) + (f10 ** 2).sum(dim=1))
            ms.append(fb.float().unsqueeze(1))
        all_xyz.append(torch.stack(xs, 0))
        all_mask.append(torch.stack(ms, 0))
    return torch.stack(all_xyz, 0), torch.stack(all_mask, 0)


def _gather_warped_feat(feat, flow, ti, tl):
    """reference model/multi_frame_networks.py:347-360: slot 0 = own frame, then the others warped."""
    return torch.stack([feat[ti]] + [warp(feat[j], flow[f'flow_{ti}{j}']) for j in _other_frames(ti, tl)], 0)


def mf_block(p, B, feat, wxyz, wmask, flow_core, tl):
    """Block2D3D.  reference model/multi_frame_networks.py:362-430 (activation checkpoints dropped:
    they change memory, not arithmetic)."""
    bs = feat.shape[1]
    # 3-D branch, stride 2 (core -> quarter)
    wfeat = torch.stack([_gather_warped_feat(feat, flow_core, ti, tl) for ti in range(tl)], 0)  # (tl,4,bs,C,h,w)
    o3d1 = torch.stack([conv3d_knn(p, B + 'conv3d_1', wxyz[ti], wfeat[ti], wmask[ti], 2, tl, target=ti)
                        for ti in range(tl)], 0)
    # 3-D branch, stride 1 at quarter resolution
    q = o3d1.shape[-2:]
    flow_q = resize_flow(flow_core, q)
    wxyz_q = resize_ac(wxyz, q)
    wmask_q = (resize_ac(wmask, q) > 0.5).float()
    o3d2 = torch.stack([conv3d_knn(p, B + 'conv3d_2', wxyz_q[ti], _gather_warped_feat(o3d1, flow_q, ti, tl),
                                   wmask_q[ti], 1, tl, target=ti) for ti in range(tl)], 0)
    # 2-D branch
    x = (wfeat * wmask / wmask.mean(dim=1, keepdim=True)).transpose(1, 2)
    x = x.reshape(tl * bs, -1, *x.shape[4:])
    mf = _gn(p, B + 'conv_mf.2', _pconv(p, B + 'conv_mf.1', x))
    a = _gn(p, B + 'conv1_1.3', F.selu(_pconv(p, B + 'conv1_1.1', mf)))
    a = _gn(p, B + 'conv1_2.3', F.selu(_pconv(p, B + 'conv1_2.1', a)))
    b = _gn(p, B + 'conv2_1.3', F.selu(_pconv(p, B + 'conv2_1.1', mf, stride=2)))
    b = _gn(p, B + 'conv2_2.3', F.selu(_pconv(p, B + 'conv2_2.1', b)))
    b = F.interpolate(b, scale_factor=2, mode='bilinear', align_corners=True)
    c = F.interpolate(o3d2.reshape(tl * bs, *o3d2.shape[2:]), scale_factor=2, mode='bilinear', align_corners=True)
    fuse = _gn(p, B + 'conv_fuse.2', _pconv(p, B + 'conv_fuse.1', torch.cat((a, b, c), dim=1)))
    return F.selu(fuse.view(tl, bs, *fuse.shape[1:]) + feat)


def mf_core_rays(K, H, W):
    """rays at the EVEN full-res pixel coordinates (cv2 INTER_NEAREST half-size resize of the
    meshgrid, reference model/multi_frame_networks.py:121-128)."""
    h, w = H // 2, W // 2
    us = np.minimum(np.floor(np.arange(w) * (W / w)).astype(np.int64), W - 1)
    vs = np.minimum(np.floor(np.arange(h) * (H / h)).astype(np.int64), H - 1)
    return make_rays(K, h, w, us, vs)


def mf_forward(p, K, ir, amb, d, depth, R, t, flow, max_disp=128, block_num=4, return_aux=False):
    """FuseNet.tforward.  reference model/multi_frame_networks.py:269-305.
    ir (tl,bs,2,H,W), amb/d/depth (tl,bs,1,H,W), R (tl,bs,3,3), t (tl,bs,3), flow dict (bs,2,H,W)."""
    tl, bs, _, H, W = ir.shape
    h, w = H // 2, W // 2
    x = torch.cat((ir, amb), 2).reshape(tl * bs, 3, H, W)
    x = torch.cat([x, d.reshape(tl * bs, 1, H, W)], dim=1)
    x = F.selu(_pconv(p, 'conv1.1', x, stride=2))
    x = F.selu(_pconv(p, 'conv2.1', x))
    x = F.selu(_pconv(p, 'conv3.1', x))
    x = F.selu(_pconv(p, 'conv4.1', x))
    for r in ('res1', 'res2', 'res3'):
        x = _resblock(p, r, x)
    feat = x.view(tl, bs, *x.shape[1:])
    depth_core = resize_ac(depth, (h, w))
    flow_core = resize_flow(flow, (h, w))
    with torch.no_grad():
        wxyz, wmask = mf_geometry(depth_core, mf_core_rays(K, H, W), R, t, flow_core)
    for b in range(block_num):
        feat = mf_block(p, f'blocks.{b}.', feat, wxyz, wmask, flow_core, tl)
    x = feat.reshape(tl * bs, *feat.shape[2:])
    a = F.selu(_pconv(p, 'amb_conv.1', amb.reshape(tl * bs, 1, H, W)))
    a = _resblock(p, 'amb_res2', _resblock(p, 'amb_res1', a))
    x = F.interpolate(x, size=(H, W), mode='bilinear', align_corners=True)
    x = F.selu(_pconv(p, 'ref_conv.1', torch.cat([x, a], dim=1)))
    for r in ('ref_res1', 'ref_res2', 'ref_res3'):
        x = _resblock(p, r, x)
    x = F.selu(_pconv(p, 'final_conv.1', x))
    disp = max_disp * torch.sigmoid(F.conv2d(x, p['predict_disp.0.weight'], p['predict_disp.0.bias'], padding=1) - 3)
    out = disp.view(tl, bs, 1, H, W)
    if return_aux:
        return out, {'wxyz': wxyz, 'wmask': wmask, 'feat': feat}
    return out


# ----------------------------------------------------------------------------------------------
# the step: copy_data / net_forward / loss_forward / Adam
# ----------------------------------------------------------------------------------------------
class StepContext(object):
    """What the reference Worker builds in __init__/get_test_sets (model/worker.py:131-180,
    model/multi_frame_worker.py:50-85): LCN'd reference pattern, K, rays, focal*baseline."""

    def __init__(self, settings, lcn_radius=5, lcn_eps=0.05, max_disp=128, tl=4):
        self.H, self.W = settings.imsize
        self.K_np = np.asarray(settings.K, dtype=np.float32)
        self.K = torch.from_numpy(self.K_np.copy())
        self.baseline = float(settings.baseline)
        self.focal = float(self.K_np[0, 0])
        self.lcn_radius, self.lcn_eps, self.max_disp, self.tl = lcn_radius, lcn_eps, max_disp, tl
        pat = torch.from_numpy(settings.pattern.mean(axis=2)[None][None].astype(np.float32))
        pat, _ = lcn(pat, lcn_radius, lcn_eps)
        # reference: 3 identical channels then .mean(dim=1) (multi_frame_worker.py:69-70, networks.py:344)
        self.pattern = torch.cat([pat, pat, pat], dim=1).mean(dim=1, keepdim=True).contiguous()
        self.ray = make_rays(self.K_np, self.H, self.W)


def copy_data(ctx, batch):
    """reference model/worker.py:418-452, restricted to the keys the hot path reads.
    batch: dict of (bs,tl,...) tensors -> dict of (tl,bs,...) tensors with im0 2-channel and std0."""
    data = {}
    for k, v in batch.items():
        v = torch.as_tensor(v)
        data[k] = v.transpose(0, 1) if v.dim() > 2 else v
    im = data['im0']
    tl, bs = im.shape[:2]
    l, s = lcn(im.reshape(-1, *im.shape[2:]), ctx.lcn_radius, ctx.lcn_eps)
    data['std0'] = s.view(tl, bs, *im.shape[2:])
    data['im0'] = torch.cat((l.view(tl, bs, *im.shape[2:]), im), dim=2)
    return data


def read_optical_flow(data, tl):
    """reference model/worker.py:457-465"""
    return {f'flow_{i}{j}': data[f'flow_{i}{j}'][0] for i in range(tl) for j in range(tl) if i != j}


def mf_net_forward(ctx, p, data, flow):
    """reference model/multi_frame_worker.py:87-101"""
    depth = disp_to_depth(data['primary_disp'], ctx.focal, ctx.baseline)
    return mf_forward(p, ctx.K_np, data['im0'], data['ambient0'], data['primary_disp'], depth, data['R'], data['t'],
                      flow, max_disp=ctx.max_disp)


def sf_net_forward(ctx, p, data):
    """reference model/single_frame_worker.py:87-99"""
    im0 = data['im0']
    tl, bs = im0.shape[:2]
    outs = sf_forward(p, im0.reshape(tl * bs, *im0.shape[2:]), max_disp=ctx.max_disp)
    return [o.view(tl, bs, *o.shape[1:]) for o in outs]


def _common_losses(ctx, outs, data, smooth_w):
    vals = []
    im = data['im0'].reshape(-1, *data['im0'].shape[2:])[:, 0:1]
    std = data['std0'].reshape(-1, *data['std0'].shape[2:])
    for s, o in enumerate(outs):
        v, _ = pattern_loss(ctx.pattern, o.reshape(-1, *o.shape[2:]), im, std)
        vals.append(v / (2 ** s))
    amb = data['ambient0']
    vals.append(smooth_loss(outs[0].reshape(-1, *outs[0].shape[2:]), amb.reshape(-1, *amb.shape[2:])) * smooth_w)
    return vals


def sgm_warmup_term(o, sgm, noise):
    """reference model/multi_frame_worker.py:168-173, single_frame_worker.py:158-163 (`real` data, epoch < warmup_epochs):
    masked L1 to the SGM disparities where they exceed 30.  The reference draws `1.5 * torch.randn(o.size())` on the host
    inside the expression; here the draw (already scaled by 1.5) is an input so that the term is a function."""
    valid = (sgm > 30).float()
    return torch.sum(torch.abs(o - sgm + noise) * valid) / torch.sum(valid)


def mf_loss_forward(ctx, out, data, flow, train=True, epoch=0, data_type='synthetic', warmup_epochs=150):
    """reference model/multi_frame_worker.py:103-175."""
    outs = [out]
    vals = _common_losses(ctx, outs, data, 0.8)
    R, t, amb = data['R'], data['t'], data['ambient0']
    depth = disp_to_depth(out, ctx.focal, ctx.baseline)
    pdepth = disp_to_depth(data['primary_disp'], ctx.focal, ctx.baseline)
    tl = depth.shape[0]
    ge_num = tl * (tl - 1) / 2
    for i in range(tl):
        for j in range(i + 1, tl):
            v = mf_flow_consistency(ctx.K, ctx.ray, depth[i], depth[j], R[i], t[i], R[j], t[j],
                                    flow[f'flow_{i}{j}'], flow[f'flow_{j}{i}'], amb[i], amb[j], pdepth[i], pdepth[j])
            vals.append(v * 0.2 / ge_num)
    if train and epoch < 2:
        vals.append(torch.mean(torch.abs(out - data['primary_disp'])) * 0.1)
    if train and epoch < warmup_epochs and data_type == 'real':
        vals.append(sgm_warmup_term(out, data['sgm_disp'], data['_sgm_noise0']) * 0.1)
    return vals


def sf_loss_forward(ctx, outs, data, flow, train=True, use_pseudo_gt=False, epoch=0, data_type='synthetic', warmup_epochs=150):
    """reference model/single_frame_worker.py:101-165."""
    vals = _common_losses(ctx, outs, data, 0.4)
    R, t, amb = data['R'], data['t'], data['ambient0']
    depth = disp_to_depth(outs[0], ctx.focal, ctx.baseline)
    tl = depth.shape[0]
    ge_num = tl * (tl - 1) / 2
    for i in range(tl):
        for j in range(i + 1, tl):
            v, _, _ = sf_flow_consistency(ctx.K, ctx.ray, depth[i], depth[j], R[i], t[i], R[j], t[j],
                                          flow[f'flow_{i}{j}'], flow[f'flow_{j}{i}'], amb[i], amb[j], clamp=0.1)
            vals.append(v * 0.2 / ge_num)
    if use_pseudo_gt:
        for s, o in enumerate(outs):
            vals.append(torch.mean(torch.abs(o - data['pseudo_gt'])) * 0.1 / (2 ** s))
    if train and data_type == 'real' and epoch < warmup_epochs:
        for s, o in enumerate(outs):   # every scale, its own draw
            vals.append(sgm_warmup_term(o, data['sgm_disp'], data[f'_sgm_noise{s}']) * 0.1)
    return vals


def adam_step(params, grads, state, lr=1e-4, betas=(0.9, 0.999), eps=1e-8):
    """torch.optim.Adam defaults (reference train_val.py:55-56).  In place on `params`.
    state: {'step': int, 'm': {k: tensor}, 'v': {k: tensor}}"""
    state['step'] += 1
    k = state['step']
    b1, b2 = betas
    bc1 = 1 - b1 ** k
    bc2 = 1 - b2 ** k
    for name, g in grads.items():
        if g is None:
            continue
        m = state['m'].setdefault(name, torch.zeros_like(g))
        v = state['v'].setdefault(name, torch.zeros_like(g))
        m.mul_(b1).add_(g, alpha=1 - b1)
        v.mul_(b2).addcmul_(g, g, value=1 - b2)
        denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
        with torch.no_grad():
            params[name].addcdiv_(m, denom, value=-lr / bc1)


def train_step(ctx, arch, params, batch, adam_state=None, epoch=0, use_pseudo_gt=False, lr=1e-4, data_type='synthetic'):
    """One iteration of Worker.train_epoch (reference model/worker.py:499-539) on CPU.
    Returns dict(out, vals, grads).  If adam_state is given, also applies the Adam update."""
    for v in params.values():
        v.grad = None
    data = copy_data(ctx, batch)
    flow = read_optical_flow(data, ctx.tl)
    if arch == 'multi_frame':
        out = mf_net_forward(ctx, params, data, flow)
        vals = mf_loss_forward(ctx, out, data, flow, True, epoch, data_type=data_type)
    else:
        out = sf_net_forward(ctx, params, data)
        vals = sf_loss_forward(ctx, out, data, flow, True, use_pseudo_gt, epoch=epoch, data_type=data_type)
    sum(vals).backward()
    grads = {k: v.grad for k, v in params.items() if v.requires_grad}
    if adam_state is not None:
        adam_step(params, grads, adam_state, lr=lr)
    return {'out': out, 'vals': vals, 'grads': grads, 'data': data}


def init_params(shapes, seed=0):
    """Deterministic random parameters for tests and fixtures (NOT the reference's initialisation):
    conv/linear weights U(+-1/sqrt(fan_in)); 1-D '.weight' (GroupNorm scale) 1+0.1*N(0,1); biases
    0.1*U(-1,1).  CPU generator => identical on every box with the same torch build."""
    g = torch.Generator().manual_seed(seed)
    p = {}
    for k, shp in shapes.items():
        if len(shp) == 1 and k.endswith('.weight'):
            v = 1 + 0.1 * torch.randn(shp, generator=g)
        elif len(shp) == 1:
            v = 0.1 * (torch.rand(shp, generator=g) * 2 - 1)
        else:
            fan_in = int(np.prod(shp[1:]))
            v = (torch.rand(shp, generator=g) * 2 - 1) / math.sqrt(fan_in)
        p[k] = v.requires_grad_(True)
    return p
