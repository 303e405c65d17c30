"""Host-side mirror of the reference module API in model/networks.py, running on the HIP kernels.

Same class names, constructor/forward signatures, return structures and state_dict keys as
/root/reference/model/networks.py (line numbers cited per class); the internals are calls into
libdis_hip.so through `depthinspace_amd.ops`.  There is no ATen/CPU fallback for the arithmetic.

Differences that are deliberate and documented in DESIGN.md:
  * TimedModule does not synchronise the device (the reference brackets every module call with two
    torch.cuda.synchronize(), networks.py:66-71).
  * dead classes of the reference (PosOutput, MultiLinear, PosToDepth, SSIM,
    ProjectionDepthSimilarityLoss) are not provided ("dead code").
"""
import numpy as np
import torch

from .. import ops, lib
from . import ext_functions


class TimedModule(torch.nn.Module):
    """reference model/networks.py:58-71 (without the per-call device synchronisations)."""

    def __init__(self, mod_name):
        super().__init__()
        self.mod_name = mod_name

    def tforward(self, *args, **kwargs):
        raise Exception('not implemented')

    def forward(self, *args, **kwargs):
        return self.tforward(*args, **kwargs)


# ---------------------------------------------------------------------------------------------
# parameter containers that reproduce the reference's state_dict keys
# ---------------------------------------------------------------------------------------------
class ConvParams(torch.nn.Module):
    """weight/bias of a Conv2d (or ConvTranspose2d), initialised like torch's default."""

    def __init__(self, cin, cout, k, transposed=False):
        super().__init__()
        m = (torch.nn.ConvTranspose2d if transposed else torch.nn.Conv2d)(cin, cout, k)
        self.weight = torch.nn.Parameter(m.weight.detach().clone())
        self.bias = torch.nn.Parameter(m.bias.detach().clone())
        self.k = k


class NormParams(torch.nn.Module):
    """affine parameters of GroupNorm(num_groups=1)."""

    def __init__(self, c):
        super().__init__()
        self.weight = torch.nn.Parameter(torch.ones(c))
        self.bias = torch.nn.Parameter(torch.zeros(c))


class LinearParams(torch.nn.Module):
    def __init__(self, cin, cout):
        super().__init__()
        m = torch.nn.Linear(cin, cout)
        self.weight = torch.nn.Parameter(m.weight.detach().clone())
        self.bias = torch.nn.Parameter(m.bias.detach().clone())


class Slots(torch.nn.Module):
    """children registered under the integer positions the reference's nn.Sequential gives them."""

    def __init__(self, mods):
        super().__init__()
        for idx, m in mods.items():
            self.add_module(str(idx), m)

    def __getitem__(self, idx):
        return self._modules[str(idx)]


# ---------------------------------------------------------------------------------------------
class OutputLayerFactory(object):
    """reference model/networks.py:102-137; only type 'disp' is on the hot path."""

    def __init__(self, type='disp', params={}):
        if type != 'disp':
            raise Exception('unknown / unsupported output layer type')
        self.type = type
        self.params = params

    def __call__(self, channels_in, imsize=None):
        m = Slots({0: ConvParams(channels_in, 1, 3)})
        # SigmoidAffine constants (reference :140-149); gamma is 1 and beta 0 on the hot path
        m.alpha = float(self.params.get('alpha', 1.0))
        m.offset = float(self.params.get('offset', 0.0))
        return m


class DispNetS(TimedModule):
    """DIS-SF network.  reference model/networks.py:170-295: seven [conv s2 + conv s1, ReLU] encoder stages
    (k 7,5,3,3,3,3,3), seven ConvTranspose2d(k3,s2,p1,op1)+ReLU decoder stages with crop_like and skip concatenation
    followed by a 3x3 iconv, four disparity heads whose outputs are bilinearly x2-upsampled (align_corners=False)
    into the next level and finally resized to full resolution.  Same state_dict keys (`conv1.0.weight`,
    `upconv7.0.weight`, `iconv3.0.weight`, `predict_disp4.0.weight`, ...).  Feature maps are nhwc; every conv runs
    on the streaming MFMA kernels of csrc/conv_gen.hip."""

    def __init__(self, channels_in, imsizes, output_facs, coordconv=False, weight_init=False, channel_multiplier=1,
                 act_dtype=torch.float32):
        super().__init__(mod_name='DispNetS')
        # storage type of the nhwc feature maps: float32 (fp32 results, the parity path) or bfloat16 (BASELINE config 2:
        # one bf16 product per MAC, fp32 accumulate; parameters, disparities and losses stay float32)
        if act_dtype not in (torch.float32, torch.bfloat16):
            raise ValueError(act_dtype)
        self.act_dtype = act_dtype
        cp = [channel_multiplier * c for c in (32, 64, 128, 256, 512, 512, 512)]
        up = [channel_multiplier * c for c in (512, 512, 256, 128, 64, 32, 16)]
        self.channels_in = channels_in
        ks = [7, 5, 3, 3, 3, 3, 3]
        cin = channels_in
        for i in range(7):
            setattr(self, f'conv{i + 1}', Slots({0: ConvParams(cin, cp[i], ks[i]), 2: ConvParams(cp[i], cp[i], ks[i])}))
            cin = cp[i]
        ups_in = [cp[6]] + up[:6]
        for j, lvl in enumerate(range(7, 0, -1)):
            setattr(self, f'upconv{lvl}', Slots({0: ConvParams(ups_in[j], up[j], 3, transposed=True)}))
        icin = {7: up[0] + cp[5], 6: up[1] + cp[4], 5: up[2] + cp[3], 4: up[3] + cp[2], 3: 1 + up[4] + cp[1],
                2: 1 + up[5] + cp[0], 1: 1 + up[6]}
        for j, lvl in enumerate(range(7, 0, -1)):
            setattr(self, f'iconv{lvl}', Slots({0: ConvParams(icin[lvl], up[j], 3)}))
        facs = output_facs if isinstance(output_facs, list) else [output_facs] * 4
        self.predict_disp4 = facs[3](up[3], imsizes[3])
        self.predict_disp3 = facs[2](up[4], imsizes[2])
        self.predict_disp2 = facs[1](up[5], imsizes[1])
        self.predict_disp1 = facs[0](up[6], imsizes[0])

    def _conv(self, x, p, stride=1, need_dgrad=True, out=None):
        k = p.weight.shape[2]
        return ops.convg(x, p.weight, p.bias, stride, (k - 1) // 2, ops.ACT_RELU, need_dgrad, out, dtype=self.act_dtype)

    def _down(self, x, slots, need_dgrad=True, skip=None):
        """downsample_conv (:222-228).  skip = (channels in front of this stage's output in the decoder concatenation
        that will use it, total channels there): the stage then writes its output straight into that buffer."""
        a = self._conv(x, slots[0], 2, need_dgrad)
        if skip is None:
            return self._conv(a, slots[2], 1), None
        off, total = skip
        cb = ops.ConcatBuf(a.shape[0], a.shape[1], a.shape[2], total, a.device, self.act_dtype)
        slot = cb.slot(off, slots[2].weight.shape[0])
        if slots[2].weight.shape[2] == 7 and self.act_dtype == torch.float32:
            # the 7x7 layer runs as seven accumulating tap-row launches: they read-modify-write their output, which is
            # cheaper on a dense tensor; one copy into the slot afterwards
            return ops.write_channels(self._conv(a, slots[2], 1), slot), cb
        return self._conv(a, slots[2], 1, out=slot), cb

    def _up(self, x, slots, cb):
        """upconv (:236-240) + crop_like (:242-244), written into channels [0, cout) of the concatenation buffer cb"""
        cout = slots[0].weight.shape[1]
        return ops.convg_transposed(x, slots[0].weight, slots[0].bias, (cb.buf.shape[1], cb.buf.shape[2]), 1, ops.ACT_RELU,
                                    out=cb.slot(0, cout), dtype=self.act_dtype)

    @staticmethod
    def _head(x, m):
        return ops.disp_head_g(x, m[0].weight, m[0].bias, m.alpha, m.offset)

    @staticmethod
    def _up2_like(d, like):
        """x2 bilinear (align_corners=False) of a planar (n,1,h,w) disparity, cropped like `like` (nhwc), returned
        as an nhwc 1-channel tensor."""
        n, _, h, w = d.shape
        u = ops.resize_planar(d, (2 * h, 2 * w), False)
        u = u[:, :, :like.shape[1], :like.shape[2]]
        return u.reshape(n, like.shape[1], like.shape[2], 1)

    @staticmethod
    def _cat(parts):
        """channel concatenation, zero-padded to a multiple of 4 channels (memory op only)"""
        c = sum(p.shape[3] for p in parts)
        if c % 4:
            z = torch.zeros(parts[0].shape[:3] + (4 - c % 4,), dtype=parts[0].dtype, device=parts[0].device)
            parts = list(parts) + [z]
        return torch.cat(parts, dim=3)

    def tforward(self, x):
        """x planar (N,channels_in,H,W) -> 4 planar (N,1,H,W) disparities"""
        x = x.contiguous()
        N, C, H, W = x.shape
        assert C == self.channels_in and C <= 4
        HW = H * W
        flat = x.view(-1)
        x4 = ops.pack4_nhwc([(flat[c * HW:], C * HW) for c in range(C)], N, H, W)
        # The decoder's concatenations (:262-288: upconv output, encoder skip, up-sampled disparity) are never copied
        # together: every encoder stage writes its output straight into the buffer of the concatenation that will use
        # it (behind the channels of that level's upconv), the upconv and the disparity channel follow in place.
        up = {lvl: getattr(self, f'upconv{lvl}')[0].weight.shape[1] for lvl in range(1, 8)}
        enc = {i: getattr(self, f'conv{i}')[2].weight.shape[0] for i in range(1, 8)}
        e, cb = {0: x4}, {}
        for i in range(1, 7):  # stage i is the skip of decoder level i + 1 (levels 2, 3 also carry a disparity channel)
            lvl = i + 1
            total = up[lvl] + enc[i] + (1 if lvl in (2, 3) else 0)
            e[i], cb[lvl] = self._down(e[i - 1], getattr(self, f'conv{i}'), need_dgrad=(i > 1), skip=(up[lvl], total))
        e[7], _ = self._down(e[6], self.conv7)
        cb[1] = ops.ConcatBuf(N, H, W, up[1] + 1, x4.device, self.act_dtype)

        def level(lvl, below, disp=None):
            parts = [(self._up(below, getattr(self, f'upconv{lvl}'), cb[lvl]), 0)]
            if lvl > 1:
                parts.append((e[lvl - 1], up[lvl]))
            if disp is not None:
                off = up[lvl] + (enc[lvl - 1] if lvl > 1 else 0)
                # (the disparity channel is the last part: the buffer's zero padding lanes are written with it)
                parts.append((ops.write_channels(self._up2_like(disp, cb[lvl].buf), cb[lvl].slot(off, 1), True), off))
            return self._conv(cb[lvl].joined(parts), getattr(self, f'iconv{lvl}')[0])

        i7 = level(7, e[7])
        i6 = level(6, i7)
        i5 = level(5, i6)
        i4 = level(4, i5)
        d4 = self._head(i4, self.predict_disp4)
        i3 = level(3, i4, d4)
        d3 = self._head(i3, self.predict_disp3)
        i2 = level(2, i3, d3)
        d2 = self._head(i2, self.predict_disp2)
        i1 = level(1, i2, d2)
        d1 = self._head(i1, self.predict_disp1)
        rs = lambda d: ops.resize_planar(d, (H, W), False)
        return (d1, rs(d2), rs(d3), rs(d4))


class DispDecoder(TimedModule):
    """reference model/networks.py:297-309"""

    def __init__(self, *args, max_disp=128, **kwargs):
        super().__init__(mod_name='DispDecoder')
        output_facs_disp = [OutputLayerFactory(type='disp', params={'alpha': max_disp / (2 ** s), 'beta': 0, 'gamma': 1,
                                                                    'offset': 3}) for s in range(4)]
        self.disp_decoder = DispNetS(*args, output_facs=output_facs_disp, **kwargs)

    def tforward(self, x):
        return self.disp_decoder(x)


class DispToDepth(TimedModule):
    """reference model/networks.py:311-319"""

    def __init__(self, focal_length, baseline):
        super().__init__(mod_name='DispToDepth')
        self.baseline_focal_length = baseline * focal_length

    def tforward(self, disp):
        return ops.disp_to_depth(disp, self.baseline_focal_length)


class LCN(TimedModule):
    """reference model/networks.py:663-689.  No parameters are registered (the reference's frozen
    all-ones box filter is not a learnable tensor)."""

    def __init__(self, radius, epsilon):
        super().__init__(mod_name='LCN')
        self.radius = radius
        self.epsilon = epsilon

    def tforward(self, data):
        return ops.lcn(data, self.radius, self.epsilon)


class RectifiedPatternSimilarityLoss(TimedModule):
    """Photometric loss.  reference model/networks.py:336-377"""

    def __init__(self, im_height, im_width, pattern, loss_type='census_sad', loss_eps=0.5):
        super().__init__(mod_name='RectifiedPatternSimilarityLoss')
        self.im_height = im_height
        self.im_width = im_width
        # 3 identical channels -> 1 (reference :344); one-off set-up arithmetic, not part of the step
        self.pattern = pattern.mean(dim=1, keepdim=True).contiguous()
        self.loss_type = loss_type
        self.loss_eps = loss_eps

    def tforward(self, disp0, im, std=None, output_mean=True):
        if self.pattern.device != disp0.device:
            self.pattern = self.pattern.to(disp0.device)
        pattern_proj = ops.pattern_warp(self.pattern, disp0)
        diff = ext_functions.photometric_loss(pattern_proj, im.contiguous(), 9, self.loss_type, self.loss_eps)
        if output_mean:
            val = ops.weighted_mean(diff, std)
        else:
            val = diff
        return val, pattern_proj

    def forward_multi(self, disps, im, std=None):
        """[self(d, im, std)[0] for d in disps] with ONE photometric launch for all estimates (not in the reference: DIS-SF calls
        the loss once per output scale against the same image, model/single_frame_worker.py:110-118; the target's census terms are
        shared).  Falls back to the per-estimate calls for the non-census types."""
        type_id = {'mse': 0, 'sad': 1, 'census_mse': 2, 'census_sad': 3}[self.loss_type.lower()]
        if not ops.photometric_multi_ok(len(disps), im.shape[1], 9, type_id):
            return [self(d, im, std)[0] for d in disps]
        if self.pattern.device != disps[0].device:
            self.pattern = self.pattern.to(disps[0].device)
        if ops.PHOTO_LOSS_ONE_NODE:
            return ops.pattern_photo_loss_multi(self.pattern, disps, im.contiguous(), std, 9, type_id, self.loss_eps)
        projs = [ops.pattern_warp(self.pattern, d) for d in disps]
        diffs = ops.photometric_multi(projs, im.contiguous(), 9, type_id, self.loss_eps)
        return [ops.weighted_mean(d, std) for d in diffs]


class DisparitySmoothLoss(TimedModule):
    """reference model/networks.py:411-431 (Sobel-5 weights are constants of the kernel)."""

    def __init__(self):
        super().__init__(mod_name='DepthSmoothLoss')

    def tforward(self, disp, im):
        return ops.smooth_loss(disp, im)


class ProjectionBaseLoss(TimedModule):
    """reference model/networks.py:433-493: holds the intrinsics; the unproject/project algebra lives in the
    fused geometric-loss kernel (csrc/pixel_ops.hip)."""

    def __init__(self, K, Ki, im_height, im_width):
        super().__init__(mod_name='ProjectionBaseLoss')
        self.K = K.view(-1, 3, 3)
        self.im_height = im_height
        self.im_width = im_width
        self._K_host = lib.host_floats(np.asarray(K, dtype=np.float32).reshape(-1))
        self._Ki_host = lib.host_floats(np.asarray(Ki, dtype=np.float32).reshape(-1))


def _geo_terms_all(loss, depth, R, t, flow_out, amb, primary_depth, clamp):
    """The per-pair values l_{ij} + l_{ji}, i < j (the loops of reference multi_frame_worker.py:139-158 / single_frame_worker.py
    :126-149), from ONE forward / backward launch over all directional terms (ops.geo_loss_all); None: the caller runs the loops
    (more than 16 terms, DIS_GEO_MULTI=0).  depth / amb / primary_depth (tl, bs, 1, h, w)."""
    tl = depth.shape[0]
    if not ops.GEO_MULTI or tl * (tl - 1) > ops.GEO_MULTI_MAX or tl < 2:
        return None
    pairs, flows = [], []
    for i in range(tl):
        for j in range(i + 1, tl):
            pairs += [(i, j), (j, i)]
            flows += [(flow_out[f'flow_{i}{j}'], flow_out[f'flow_{j}{i}']), (flow_out[f'flow_{j}{i}'], flow_out[f'flow_{i}{j}'])]
    vals = ops.geo_loss_all(depth, amb, primary_depth, R, t, loss._K_host, loss._Ki_host, clamp, pairs, flows)
    return vals.view(-1, 2).sum(1).unbind(0)


class Multi_Frame_Flow_Consistency_Loss(ProjectionBaseLoss):
    """reference model/networks.py:554-607"""

    def __init__(self, *args, clamp=-1):
        super().__init__(*args)
        self.mod_name = 'Multi_Frame_Flow_Consistency_Loss'
        self.clamp = clamp  # stored but unused, as in the reference

    def forward_all(self, depth, R, t, flow_out, amb, primary_depth):
        return _geo_terms_all(self, depth, R, t, flow_out, amb, primary_depth, -1.0)

    def fwd(self, depth0, depth1, R0, t0, R1, t1, flow0, flow1, amb0, amb1, primary_depth1, accs=None):
        val, _ = ops.geo_loss_dir(depth0, depth1, flow0, flow1, amb0, amb1, primary_depth1, R0, t0, R1, t1,
                                  self._K_host, self._Ki_host, -1.0, accs)
        return val

    def tforward(self, depth0, depth1, R0, t0, R1, t1, flow0, flow1, amb0, amb1, primary_depth0, primary_depth1,
                 accs=None):
        """accs (not in the reference): optional (ops.GradAccum of depth0, of depth1), see ops.GradAccum."""
        l0 = self.fwd(depth0, depth1, R0, t0, R1, t1, flow0, flow1, amb0, amb1, primary_depth1, accs)
        l1 = self.fwd(depth1, depth0, R1, t1, R0, t0, flow1, flow0, amb1, amb0, primary_depth0,
                      (accs[1], accs[0]) if accs is not None else None)
        return l0 + l1


class Single_Frame_Flow_Consistency_Loss(ProjectionBaseLoss):
    """reference model/networks.py:609-661.  The reference's host copy of a mask per call (:640) is dropped;
    `orig_mask` is returned as None (single_frame_worker.py:148 discards it)."""

    def __init__(self, *args, clamp=-1):
        super().__init__(*args)
        self.mod_name = 'Single_Frame_Flow_Consistency_Loss'
        self.clamp = clamp

    def forward_all(self, depth, R, t, flow_out, amb):
        return _geo_terms_all(self, depth, R, t, flow_out, amb, None, float(self.clamp))

    def fwd(self, depth0, depth1, R0, t0, R1, t1, flow0, flow1, amb0, amb1, accs=None):
        val, mask = ops.geo_loss_dir(depth0, depth1, flow0, flow1, amb0, amb1, None, R0, t0, R1, t1,
                                     self._K_host, self._Ki_host, float(self.clamp), accs)
        return val, mask, None

    def tforward(self, depth0, depth1, R0, t0, R1, t1, flow0, flow1, amb0, amb1, accs=None):
        """accs (not in the reference): optional (ops.GradAccum of depth0, of depth1), see ops.GradAccum."""
        l0, mask0, orig_mask = self.fwd(depth0, depth1, R0, t0, R1, t1, flow0, flow1, amb0, amb1, accs)
        l1, mask1, _ = self.fwd(depth1, depth0, R1, t1, R0, t0, flow1, flow0, amb1, amb0,
                                (accs[1], accs[0]) if accs is not None else None)
        return l0 + l1, mask0, mask1, orig_mask
