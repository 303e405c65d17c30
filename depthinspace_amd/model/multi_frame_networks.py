"""Host-side mirror of /root/reference/model/multi_frame_networks.py (FuseNet, Block2D3D, Conv3D,
ResNetBlock and the helpers) on the HIP kernels.

Same class names, constructor/forward signatures and state_dict keys (236 tensors, e.g.
`conv1.1.weight`, `blocks.0.conv3d_1.dense1.0.weight`, `ref_res3.bn2.bias`).  Internally every feature
map is nhwc; the track (tl) and batch (bs) dims are folded exactly like `merge_tl_bs` (frame slow).

Deliberate differences: no activation checkpointing (288 GB HBM; arithmetic unchanged),
no per-module device syncs, the parameter-free geometry/flow pyramids are computed once per forward and
shared by the four blocks (the reference recomputes identical values in every block).
"""
import numpy as np
import torch

from .. import ops, lib
from .networks import TimedModule, OutputLayerFactory, ConvParams, NormParams, LinearParams, Slots

SELU, NONE = ops.ACT_SELU, ops.ACT_NONE


def merge_tl_bs(x):
    """reference :36-37"""
    return x.contiguous().view(-1, *x.shape[2:])


def split_tl_bs(x, tl, bs):
    """reference :39-40"""
    return x.contiguous().view(tl, bs, *x.shape[1:])


class FlowDict(dict):
    """dict of flow_ij -> (bs,2,H,W) that may also carry the same flows pre-stacked as one
    (tl*tl, bs, 2, H, W) tensor (`.stacked`, entry i*tl+j), so the step does not have to re-pack them."""
    stacked = None


def stack_flows(flow, tl):
    st = getattr(flow, 'stacked', None)
    if st is not None:
        return st
    any_f = next(iter(flow.values()))
    out = torch.zeros((tl * tl,) + tuple(any_f.shape), dtype=any_f.dtype, device=any_f.device)
    for i in range(tl):
        for j in range(tl):
            if i != j:
                out[i * tl + j].copy_(flow[f'flow_{i}{j}'])
    return out


def resize_like(x, target):
    """reference :42-51 for planar tensors (..., C, H, W): bilinear, align_corners=True"""
    return ops.resize_planar(x, (target.shape[-2], target.shape[-1]), True)


def resize_flow_like(flow, target):
    """reference :54-68"""
    height, width = target.shape[-2], target.shape[-1]
    out = {}
    for key, val in flow.items():
        fh, fw = val.shape[-2], val.shape[-1]
        out[key] = ops.resize_planar(val, (height, width), True,
                                     flow_scale=(float(width) / float(fw), float(height) / float(fh)))
    return out


def warp(x, flow):
    """reference :83-99 for a planar x (bs,C,h,w): sample x at (u,v)+flow, zeros padding."""
    bs, c, h, w = x.shape
    cp = (c + 3) // 4 * 4
    xn = ops.planar_to_nhwc(x)
    if cp != c:
        xn = torch.cat([xn, torch.zeros(bs, h, w, cp - c, device=x.device)], dim=3)
    flows = torch.zeros(4, bs, h, w, 2, device=x.device)
    flows[1] = ops.planar_to_nhwc(flow)
    feat = torch.zeros(2, bs, h, w, cp, device=x.device)
    feat[1] = xn
    out = ops.gather_warped_feat(feat, flows)  # target 0, slot 1 = frame 1 warped by flow_01
    return ops.nhwc_to_planar(out[0, :, :, :, 1, :c].contiguous())


class ResNetBlock(torch.nn.Module):
    """reference :514-542: out = GN1(SELU(conv1(pad x))); out = GN2(conv2(pad out)); SELU(out + x)"""

    def __init__(self, planes):
        super().__init__()
        self.conv1 = ConvParams(planes, planes, 3)
        self.bn1 = NormParams(planes)
        self.conv2 = ConvParams(planes, planes, 3)
        self.bn2 = NormParams(planes)

    def forward(self, x, defer_out=False):
        """defer_out: the caller hands the result STRAIGHT to a 3x3 conv (the next ResNetBlock, final_conv): that conv writes the
        block's output while it loads it (ops.group_norm(defer=True)); x may itself be such an unwritten output."""
        j = ops.GradJoin() if x.requires_grad else None  # x feeds conv1 and the residual: one shared gradient buffer
        # (x may be the previous ResNetBlock's SELU(GroupNorm(.) + residual): conv1's input-gradient launch then also serves
        # that GroupNorm's backward, ops._Conv2d.backward)
        o, st = ops.conv2d(x, self.conv1.weight, self.conv1.bias, 1, 1, SELU, want_stats=True, gy_is_pre=True, join=j,
                           gnres=getattr(x, '_gn_res_src', None) if j is not None else None)
        if ops.gn_fusable(o.shape[-1], self.conv2.weight.shape[0], 3, 1):
            # bn1 is applied by conv2 while it stages its input: GN1(...) is never written
            o, st = ops.conv2d_gn_in(o, st, self.bn1.weight, self.bn1.bias, self.conv2.weight, self.conv2.bias, 1, NONE,
                                     want_stats=True, in_act=SELU)
        else:
            o = ops.group_norm(o, self.bn1.weight, self.bn1.bias, stats=st, in_act=SELU)
            o, st = ops.conv2d(o, self.conv2.weight, self.conv2.bias, 1, 1, NONE, want_stats=True)
        return ops.group_norm(o, self.bn2.weight, self.bn2.bias, stats=st, residual=x, act=SELU, join=j, defer=defer_out)


class Conv3D(TimedModule):
    """reference :432-512.  forward(geom, wf) works on ALL target frames at once:
    geom (tl,bs,h,w,tl,4) xyz+mask per slot, wf (tl,bs,h,w,tl,C) gathered features -> (tl,bs,h',w',C)."""

    def __init__(self, channels_in, channels_out, neighbors=9, tl=4, ksize=3, stride=1, radius_sq=0.04):
        super().__init__(mod_name='Conv3D')
        assert channels_in == channels_out == 32 and neighbors == 9 and tl == 4 and ksize == 3
        self.stride = stride
        self.dense1 = Slots({0: LinearParams(3, channels_out // 2)})
        self.dense2 = Slots({0: LinearParams(channels_out // 2, channels_out)})
        self.w = torch.nn.Parameter(torch.zeros([channels_out, channels_out]))
        torch.nn.init.xavier_uniform_(self.w, gain=0.1)
        self.bn = NormParams(channels_out)

    def tforward(self, geom, wf, idx=None, join=None):
        if idx is None:
            idx = ops.conv3d_select(geom, self.stride)
        y = ops.conv3d_knn(geom, wf, self.dense1[0].weight, self.dense1[0].bias, self.dense2[0].weight,
                           self.dense2[0].bias, self.w, idx, self.stride, join)
        tl, bs, ho, wo, c = y.shape
        o = ops.group_norm(y.view(tl * bs, ho, wo, c), self.bn.weight, self.bn.bias)
        return o.view(tl, bs, ho, wo, c)


class Block2D3D(TimedModule):
    """reference :307-430"""

    def __init__(self, channels, tl):
        super().__init__(mod_name='Block2D3D')
        self.channels = channels
        self.tl = tl
        C = channels
        self.conv_mf = Slots({1: ConvParams(C * tl, C, 1), 2: NormParams(C)})
        self.conv1_1 = Slots({1: ConvParams(C, C, 3), 3: NormParams(C)})
        self.conv1_2 = Slots({1: ConvParams(C, C, 3), 3: NormParams(C)})
        self.conv2_1 = Slots({1: ConvParams(C, C, 4), 3: NormParams(C)})
        self.conv2_2 = Slots({1: ConvParams(C, C, 3), 3: NormParams(C)})
        self.conv_fuse = Slots({1: ConvParams(C * 3, C, 3), 2: NormParams(C)})
        self.conv3d_1 = Conv3D(channels_in=C, channels_out=C, tl=tl, stride=2)
        self.conv3d_2 = Conv3D(channels_in=C, channels_out=C, tl=tl, stride=1)

    @staticmethod
    def _conv_gn(x, slots, gn_idx, stride, pad, act, join=None):
        o, st = ops.conv2d(x, slots[1].weight, slots[1].bias, stride, pad, act, want_stats=True, gy_is_pre=True,
                           join=join)
        return ops.group_norm(o, slots[gn_idx].weight, slots[gn_idx].bias, stats=st, in_act=act)

    @staticmethod
    def _conv_gn_pair(x, s1, s2, stride1, join=None, gnres=None, apply_last=True):
        """GN(SELU(conv_s2(GN(SELU(conv_s1(x)))))) (reference :338-345: two Conv-SELU-GroupNorm stages): the first GroupNorm is
        applied by the second conv while it stages its input (ops.conv2d_gn_in), the second one is written"""
        if not ops.gn_fusable(s1[1].weight.shape[0], s2[1].weight.shape[0], s2[1].weight.shape[2], 1):
            return Block2D3D._conv_gn(Block2D3D._conv_gn(x, s1, 3, stride1, 1, SELU, join), s2, 3, 1, 1, SELU)
        o, st = ops.conv2d(x, s1[1].weight, s1[1].bias, stride1, 1, SELU, want_stats=True, gy_is_pre=True, join=join,
                           gnres=gnres)
        o, st = ops.conv2d_gn_in(o, st, s1[3].weight, s1[3].bias, s2[1].weight, s2[1].bias, 1, SELU, want_stats=True,
                                 gy_is_pre=True, in_act=SELU)
        if not apply_last:   # the caller applies the second GroupNorm on load as well (conv_fuse): (pre-normalisation tensor, stats)
            return o, st
        return ops.group_norm(o, s2[3].weight, s2[3].bias, stats=st, in_act=SELU)

    def tforward(self, feat, geom, geom_q, flows, flows_q, idx=None, idx_q=None, csr=None, csr_q=None, wgt=None):
        """feat (tl,bs,h,w,C) nhwc.  geom/geom_q: core / quarter geometry; flows/flows_q: (tl*tl,bs,.,.,2);
        idx/idx_q: neighbour sets of the two Conv3D layers (shared by all blocks: they depend on geometry only)."""
        tl, bs, h, w, C = feat.shape
        N = tl * bs
        # tensors with two consumers on the tape share one gradient buffer each (ops.GradJoin): feat (gather + residual),
        # wf (Conv3D + the 2-D branch) and mf (the two 2-D branches)
        grad = torch.is_grad_enabled() and feat.requires_grad
        j_feat, j_wf, j_mf = (ops.GradJoin(), ops.GradJoin(), ops.GradJoin()) if grad else (None, None, None)
        # 3-D branch (fwd_3d_1 / fwd_3d_2, reference :376-404)
        # (feat is the previous block's / res3's SELU(GroupNorm(.) + residual): the feature warp's backward, which completes the
        #  gradient wrt feat, then also does the first pass of that GroupNorm's backward - ops._GatherWarpedFeat)
        wf = ops.gather_warped_feat(feat, flows, csr, j_feat, gnres=getattr(feat, '_gn_res_src', None) if grad else None)
        o3d1 = self.conv3d_1(geom, wf, idx, j_wf)
        wfq = ops.gather_warped_feat(o3d1, flows_q, csr_q)
        o3d2 = self.conv3d_2(geom_q, wfq, idx_q)
        hq, wq = o3d2.shape[2:4]
        # 2-D branch (fwd_2d, reference :406-430)
        # conv_mf over the mask-weighted slots: the weights mask/mean(mask) (:410) are applied inside the 1x1 conv kernels
        if wgt is None:
            wgt = ops.slot_weights(geom)
        o, st = ops.conv2d_scaled_in(wf.view(N, h, w, tl * C), wgt, self.conv_mf[1].weight, self.conv_mf[1].bias, 1, 0,
                                     want_stats=True, join=j_wf)
        mf = ops.group_norm(o, self.conv_mf[2].weight, self.conv_mf[2].bias, stats=st)
        # (mf = GroupNorm(conv_mf(.)) has two consumers; the a-branch's backward runs second: conv1_1's accumulating input-gradient
        # launch completes the gradient wrt mf and leaves the sums for that GroupNorm's backward)
        fuse_a = ops.gn_fusable(C, C, 3, 1)   # conv_fuse applies conv1_2's GroupNorm on load (its first input slice)
        a = self._conv_gn_pair(mf, self.conv1_1, self.conv1_2, 1, j_mf, gnres=getattr(mf, '_gn_plain_src', None) if grad else None,
                               apply_last=not fuse_a)
        b = self._conv_gn_pair(mf, self.conv2_1, self.conv2_2, 2, j_mf)
        b = ops.resize_nhwc(b, (2 * b.shape[1], 2 * b.shape[2]), True)
        c = ops.resize_nhwc(o3d2.view(N, hq, wq, C), (2 * hq, 2 * wq), True)
        # conv over cat(a, b, c) as three accumulating 32->32 launches: the 96-channel tensor never exists
        if fuse_a:
            a, st_a = a
            f, st = ops.conv2d_multi((a, b, c), self.conv_fuse[1].weight, self.conv_fuse[1].bias, 1, NONE, want_stats=True,
                                     gn0=(st_a, self.conv1_2[3].weight, self.conv1_2[3].bias, 1e-5, SELU))
        else:
            f, st = ops.conv2d_multi((a, b, c), self.conv_fuse[1].weight, self.conv_fuse[1].bias, 1, NONE, want_stats=True)
        out = ops.group_norm(f, self.conv_fuse[2].weight, self.conv_fuse[2].bias, stats=st,
                             residual=feat.view(N, h, w, C), act=SELU, join=j_feat)
        o5 = out.view(tl, bs, h, w, C)
        o5._gn_res_src = getattr(out, '_gn_res_src', None)   # (the next block's feature warp reads it, see above)
        return o5


class FuseNet(TimedModule):
    """reference :101-305.  forward(ir, amb, d, depth, R, t, flow) -> (tl,bs,1,H,W)"""

    def __init__(self, imsize, K, baseline, track_length=4, block_num=4, channels=32, max_disp=128,
                 movement_mask_en=1):
        super().__init__(mod_name='FuseNet')
        self.movement_mask_en = movement_mask_en
        self.im_height = imsize[0]
        self.im_width = imsize[1]
        self.core_height = self.im_height // 2
        self.core_width = self.im_width // 2
        if self.im_height % 2 or self.im_width % 2:
            raise ValueError('FuseNet (HIP path) needs even image sizes')
        self.K = np.asarray(K, dtype=np.float32)
        self.Ki = np.linalg.inv(self.K)
        self._Ki_host = lib.host_floats(self.Ki.reshape(-1))
        self.baseline = baseline
        self.track_length = track_length
        self.block_num = block_num
        self.channels = channels
        self.max_disp = max_disp
        self.knn_index_override = None  # diagnostics only (externally supplied neighbour sets); no test or entry point sets it
        self.last_knn_index = None
        C = channels
        self.conv1 = Slots({1: ConvParams(4, C // 2, 4)})
        self.conv2 = Slots({1: ConvParams(C // 2, C // 2, 3)})
        self.conv3 = Slots({1: ConvParams(C // 2, C, 3)})
        self.conv4 = Slots({1: ConvParams(C, C, 3)})
        self.res1 = ResNetBlock(C)
        self.res2 = ResNetBlock(C)
        self.res3 = ResNetBlock(C)
        self.blocks = torch.nn.ModuleList([Block2D3D(channels=C, tl=track_length) for _ in range(block_num)])
        # constructed but never used by the reference either (:143-144, :239-244): kept for state_dict parity
        self.upconv1 = Slots({0: ConvParams(C, C, 4, transposed=True)})
        self.upconv2 = Slots({0: ConvParams(C, C, 4, transposed=True)})
        self.amb_conv = Slots({1: ConvParams(1, 16, 3)})
        self.amb_res1 = ResNetBlock(16)
        self.amb_res2 = ResNetBlock(16)
        self.ref_conv = Slots({1: ConvParams(16 + C, 32, 3)})
        self.ref_res1 = ResNetBlock(32)
        self.ref_res2 = ResNetBlock(32)
        self.ref_res3 = ResNetBlock(32)
        self.final_conv = Slots({1: ConvParams(32, 16, 3)})
        self.predict_disp = OutputLayerFactory(type='disp', params={'alpha': self.max_disp, 'beta': 0, 'gamma': 1,
                                                                    'offset': 3})(16)

    def pre_process(self, x4):
        """reference :216-227; x4 = cat(ir, amb, d) packed nhwc (N,H,W,4)"""
        c = lambda x, s, stride, pad, dg=True: ops.conv2d(x, s[1].weight, s[1].bias, stride, pad, SELU,
                                                          need_dgrad=dg)[0]
        x = c(x4, self.conv1, 2, 1, False)
        x = c(x, self.conv2, 1, 1)
        x = c(x, self.conv3, 1, 1)
        x = c(x, self.conv4, 1, 1)
        return self.res3(self.res2(self.res1(x, defer_out=True), defer_out=True))

    def post_process(self, feat, amb4):
        """reference :229-267; feat (N,h,w,C) nhwc, amb4 (N,H,W,4) = ambient + zero channels"""
        H, W = self.im_height, self.im_width
        a = ops.conv2d(amb4, self.amb_conv[1].weight, self.amb_conv[1].bias, 1, 1, SELU, need_dgrad=False)[0]
        a = self.amb_res2(self.amb_res1(a, defer_out=True))
        up = ops.resize_nhwc(feat, (H, W), True)
        x = ops.conv2d_multi((up, a), self.ref_conv[1].weight, self.ref_conv[1].bias, 1, SELU)[0]
        x = self.ref_res3(self.ref_res2(self.ref_res1(x, defer_out=True), defer_out=True), defer_out=True)
        # (x = ref_res3's SELU(GroupNorm(.) + residual) and final_conv is its only consumer: gnres, see ResNetBlock.forward)
        x = ops.conv2d(x, self.final_conv[1].weight, self.final_conv[1].bias, 1, 1, SELU,
                       gnres=getattr(x, '_gn_res_src', None) if x.requires_grad else None)[0]
        return ops.disp_head(x, self.predict_disp[0].weight, self.predict_disp[0].bias, float(self.max_disp), 3.0)

    def tforward(self, ir, amb, d, depth, R, t, flow):
        tl, bs = ir.shape[0], ir.shape[1]
        H, W, h, w = self.im_height, self.im_width, self.core_height, self.core_width
        assert tuple(ir.shape[2:]) == (2, H, W) and tl == self.track_length
        N, HW = tl * bs, H * W
        ir, amb, d, depth = ir.contiguous(), amb.contiguous(), d.contiguous(), depth.contiguous()
        R, t = R.contiguous(), t.contiguous()
        x4 = ops.pack4_nhwc([(ir, 2 * HW), (ir.view(-1)[HW:], 2 * HW), (amb, HW), (d, HW)], N, H, W)
        f4 = self.pre_process(x4)
        feat = f4.view(tl, bs, h, w, self.channels)
        feat._gn_res_src = getattr(f4, '_gn_res_src', None)   # (res3's output: Block2D3D's feature warp reads it)

        # parameter-free pyramids, no gradient (reference :280-294, :392-394)
        with torch.no_grad():
            depth_core = ops.resize_planar(depth.view(tl, bs, H, W), (h, w), True)
            ff = stack_flows(flow, tl)  # (tl*tl,bs,2,H,W)
            fc_p = ops.resize_planar(ff, (h, w), True, flow_scale=(float(w) / float(W), float(h) / float(H)))
            flows = ops.planar_to_nhwc(fc_p.view(tl * tl * bs, 2, h, w)).view(tl * tl, bs, h, w, 2)
            geom = ops.mf_geometry(depth_core, R, t, flows, self._Ki_host, W // w, H // h)
            hq, wq = (h + 2 - 3) // 2 + 1, (w + 2 - 3) // 2 + 1
            fq_p = ops.resize_planar(fc_p, (hq, wq), True, flow_scale=(float(wq) / float(w), float(hq) / float(h)))
            flows_q = ops.planar_to_nhwc(fq_p.view(tl * tl * bs, 2, hq, wq)).view(tl * tl, bs, hq, wq, 2)
            geom_q = ops.mf_geometry_resize(geom, (hq, wq))
            # neighbour selection of Conv3D: a function of the geometry only => once per forward, not once per
            # layer (the reference repeats the identical top-k in all 8 Conv3D calls).  `knn_index_override`
            # (tests only) substitutes externally supplied neighbour sets, see DESIGN.md "top-k conditioning".
            ov = self.knn_index_override
            train = torch.is_grad_enabled() or feat.requires_grad
            # (training: the sets' by-source-row index rides along as idx.c3csr - the Conv3D feature gradient then is a
            # fixed-order gather instead of a float-atomic scatter; built once, used by the 4 blocks' backward passes)
            idx = ov[0] if ov is not None else ops.conv3d_select(geom, 2, with_csr=train)
            idx_q = ov[1] if ov is not None else ops.conv3d_select(geom_q, 1, with_csr=train)
            if ov is not None and train and ops.CONV3D_CSR:
                ops.conv3d_csr(idx, geom.shape[2], geom.shape[3], 2)
                ops.conv3d_csr(idx_q, geom_q.shape[2], geom_q.shape[3], 1)
            # scatter index of the feature warps (data only): built once, shared by the 4 blocks' backward passes
            csr = ops.gather_csr(flows) if train else None
            csr_q = ops.gather_csr(flows_q) if csr is not None else None
            wgt = ops.slot_weights(geom)
        self.last_knn_index = (idx, idx_q)

        for block in self.blocks:
            feat = block(feat, geom, geom_q, flows, flows_q, idx, idx_q, csr, csr_q, wgt)

        amb4 = ops.pack4_nhwc([(amb, HW)], N, H, W)
        disp = self.post_process(feat.view(N, h, w, self.channels), amb4)
        # every deferred block output (ops.group_norm(defer=True)) has been written by its first consumer by now; checked HERE, in the
        # forward that created it, because evaluation / inference loops never reach the optimizer's begin_step() check
        ops.check_forward_complete()
        return disp.view(tl, bs, 1, H, W)
