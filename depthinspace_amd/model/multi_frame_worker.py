"""Host-side mirror of /root/reference/model/multi_frame_worker.py: the DIS-MF stage worker
(loss construction :50-85, net_forward :87-101, loss_forward / weighting :103-175).
Visualisation callbacks (write_img, matplotlib) are out of scope."""
import itertools

import numpy as np
import torch

from . import networks
from . import worker
from .. import ops


class Worker(worker.Worker):
    def __init__(self, args, **kwargs):
        super().__init__(args, **kwargs)
        self.disparity_loss = networks.DisparitySmoothLoss()

    def _make_dataset(self, n, seed, pseudo):
        from ..data.dataset import SyntheticTrackDataset
        return SyntheticTrackDataset(self.settings, n, self.track_length, seed=seed, load_primary_data=True,
                                     load_pseudo_gt=pseudo)

    def get_train_set(self):
        return self._make_dataset(64, 1234, False)

    def get_test_sets(self):
        test_sets = worker.TestSets()
        test_set = self._make_dataset(8, 99, self.use_pseudo_gt)
        test_sets.append('simple', test_set, test_frequency=1)
        self.build_losses(test_set)
        return test_sets

    def build_losses(self, test_set=None, device='cuda'):
        """the loss objects the reference builds inside get_test_sets (:55-83)"""
        self.patterns, self.ph_losses, self.ge_losses, self.d2ds = [], [], [], []
        imsize = self.imsizes[0]
        pat = self.ref_pattern.mean(axis=2)
        pat = torch.from_numpy(pat[None][None].astype(np.float32)).to(device)
        pat, _ = self.lcn_in(pat)
        self.patterns.append(pat)
        pat3 = torch.cat([pat for _ in range(3)], dim=1)
        self.ph_losses.append(networks.RectifiedPatternSimilarityLoss(imsize[0], imsize[1], pattern=pat3))
        K = torch.from_numpy(self.K)
        Ki = torch.from_numpy(np.linalg.inv(self.K))
        self.ge_losses.append(self._ge_loss_cls()(K, Ki, imsize[0], imsize[1], clamp=0.1))
        self.d2ds.append(networks.DispToDepth(float(self.K[0, 0]), float(self.baseline)))

    def _ge_loss_cls(self):
        return networks.Multi_Frame_Flow_Consistency_Loss

    def net_forward(self, net, flow):
        im0 = self.data['im0']
        ambient0 = self.data['ambient0']
        disp0 = self.data['primary_disp']
        R = self.data['R']
        t = self.data['t']
        depth = self.d2ds[0](disp0)
        return net(im0, ambient0, disp0, depth, R, t, flow)

    def loss_forward(self, out, train, flow_out=None):
        if not (isinstance(out, tuple) or isinstance(out, list)):
            out = [out]
        vals = []
        # photometric
        im = self.data['im0']
        im = im.view(-1, *im.shape[2:])
        std = self.data['std0']
        std = std.view(-1, *std.shape[2:])
        im_lcn = im[:, 0:1, ...].contiguous()
        for s, o in zip(itertools.count(), out):
            o = o.view(-1, *o.shape[2:])
            val, _ = self.ph_losses[0](o, im_lcn, std)
            vals.append(val / (2 ** s))
        # smoothness
        amb0 = self.data['ambient0']
        amb0 = amb0.contiguous().view(-1, *amb0.shape[2:])
        o = out[0].view(-1, *out[0].shape[2:])
        vals.append(self.disparity_loss(o, amb0) * 0.8)
        # geometric
        R, t, amb = self.data['R'], self.data['t'], self.data['ambient0']
        primary_disp = self.data['primary_disp']
        ge_num = self.track_length * (self.track_length - 1) / 2
        depth = self.d2ds[0](out[0])
        with torch.no_grad():
            primary_depth = self.d2ds[0](primary_disp)
        ge_loss = self.ge_losses[0]
        # one unbind (backward: one stack) instead of a select per use, and one shared gradient buffer per frame for the
        # 6 directional terms it takes part in (their backward kernels accumulate)
        ds = depth.unbind(0)
        accs = [ops.GradAccum() for _ in ds] if depth.requires_grad else None
        for tidx0 in range(depth.shape[0]):
            for tidx1 in range(tidx0 + 1, depth.shape[0]):
                val = ge_loss(ds[tidx0], ds[tidx1], R[tidx0], t[tidx0], R[tidx1], t[tidx1],
                              flow_out[f'flow_{tidx0}{tidx1}'], flow_out[f'flow_{tidx1}{tidx0}'], amb[tidx0],
                              amb[tidx1], primary_depth[tidx0], primary_depth[tidx1],
                              accs=(accs[tidx0], accs[tidx1]) if accs is not None else None)
                vals.append(val * 0.2 / ge_num)
        # warm-up terms
        if train:
            if self.current_epoch < 2:
                vals.append(ops.l1_mean(out[0], self.data['primary_disp']) * 0.1)
            if self.current_epoch < self.warmup_epochs and self.data_type == 'real':
                raise NotImplementedError('real-data SGM warm-up term (reference :168-173) is not on the synthetic path')
        return vals
