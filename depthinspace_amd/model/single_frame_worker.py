"""Host-side mirror of /root/reference/model/single_frame_worker.py: the DIS-SF / DIS-FTSF stage worker
(loss construction :50-85, net_forward :87-99, loss_forward / weighting :101-165).
Visualisation callbacks (write_img, matplotlib) are out of scope."""
import itertools

import numpy as np
import torch

from . import networks
from . import multi_frame_worker
from .. import ops


class Worker(multi_frame_worker.Worker):
    """Shares the dataset / loss-object construction with the multi-frame worker (the reference duplicates that
    code, single_frame_worker.py:50-85 == multi_frame_worker.py:50-85 up to the geometric-loss class)."""

    load_primary_data = False  # reference single_frame_worker.py:46-52
    n_sgm_draws = 4            # one noise draw per output scale (:158-163)

    def _train_pseudo_gt(self):
        return self.use_pseudo_gt

    def _ge_loss_cls(self):
        return networks.Single_Frame_Flow_Consistency_Loss

    def net_forward(self, net, flow=None):
        """reference :87-99"""
        im0 = self.data['im0']
        tl, bs = im0.shape[0], im0.shape[1]
        out = net(im0.view(-1, *im0.shape[2:]))
        if not (isinstance(out, tuple) or isinstance(out, list)):
            return out.view(tl, bs, *out.shape[1:])
        return [o.view(tl, bs, *o.shape[1:]) for o in out]

    def loss_forward(self, out, train, flow_out=None):
        """reference :101-165"""
        if not (isinstance(out, tuple) or isinstance(out, list)):
            out = [out]
        vals = []
        # photometric: every scale against the full-resolution image (weights 1/2^s)
        im = self.data['im0']
        im = im.view(-1, *im.shape[2:])
        std = self.data['std0']
        std = std.view(-1, *std.shape[2:])
        im_lcn = im[:, 0:1, ...].contiguous()
        # (one launch for all scales: the census terms of the image are shared, networks.RectifiedPatternSimilarityLoss.forward_multi)
        ph = self.ph_losses[0].forward_multi([o.view(-1, *o.shape[2:]) for o in out], im_lcn, std)
        for s, val in zip(itertools.count(), ph):
            vals.append((val, 1.0 / (2 ** s)))
        # smoothness on scale 0
        amb0 = self.data['ambient0']
        amb0 = amb0.contiguous().view(-1, *amb0.shape[2:])
        o = out[0].view(-1, *out[0].shape[2:])
        vals.append((self.disparity_loss(o, amb0), 0.4))
        # geometric
        R, t, amb = self.data['R'], self.data['t'], self.data['ambient0']
        ge_num = self.track_length * (self.track_length - 1) / 2
        depth = self.d2ds[0](out[0])
        ge_loss = self.ge_losses[0]
        # one unbind (backward: one stack) instead of a select per use, and one shared gradient buffer per frame for the
        # 6 directional terms it takes part in (their backward kernels accumulate): as in multi_frame_worker
        allv = ge_loss.forward_all(depth, R, t, flow_out, amb)   # (one launch each way for the 12 directional terms)
        if allv is not None:
            vals += [(v, 0.2 / ge_num) for v in allv]
        ds = depth.unbind(0) if allv is None else ()
        accs = [ops.GradAccum() for _ in ds] if depth.requires_grad else None
        for tidx0 in range(len(ds)):
            for tidx1 in range(tidx0 + 1, len(ds)):
                val, _, _, _ = ge_loss(ds[tidx0], ds[tidx1], R[tidx0], t[tidx0], R[tidx1], t[tidx1],
                                       flow_out[f'flow_{tidx0}{tidx1}'], flow_out[f'flow_{tidx1}{tidx0}'],
                                       amb[tidx0], amb[tidx1],
                                       accs=(accs[tidx0], accs[tidx1]) if accs is not None else None)
                vals.append((val, 0.2 / ge_num))
        # pseudo ground truth (DIS-FTSF)
        if self.use_pseudo_gt:
            for s, o in zip(itertools.count(), out):
                vals.append((ops.l1_mean(o, self.data['pseudo_gt']), 0.1 / (2 ** s)))
        if train and self.data_type == 'real' and self.current_epoch < self.warmup_epochs:
            for s, o in zip(itertools.count(), out):  # every scale, a fresh noise draw each (reference :158-163)
                vals.append((self.sgm_warmup_term(o, s), 0.1))
        # (value, weight) pairs -> the weighted terms the reference returns, with their sum as one autograd node (ops.LossTerms)
        return ops.weighted_terms(vals)
