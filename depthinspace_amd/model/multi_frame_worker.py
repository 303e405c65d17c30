"""Host-side mirror of /root/reference/model/multi_frame_worker.py: the DIS-MF stage worker
(loss construction :50-85, net_forward :87-101, loss_forward / weighting :103-175).
Visualisation callbacks (write_img, matplotlib) are out of scope."""
import itertools
import logging

import numpy as np
import torch

from . import networks
from . import worker
from .. import ops


class Worker(worker.Worker):
    def __init__(self, args, **kwargs):
        super().__init__(args, **kwargs)
        self.disparity_loss = networks.DisparitySmoothLoss()

    # which pre-saved disparities the stage reads (reference :46-52: DIS-MF trains on the DIS-SF output)
    load_primary_data = True
    n_sgm_draws = 1  # noise draws of the real-data warm-up term per step (:168-173: scale 0 only)

    def _train_pseudo_gt(self):
        return False  # reference :47: load_pseudo_gt=False for the multi-frame train set

    def _make_dataset(self, paths, train, pseudo, n_synth, seed):
        """the on-disk tracks of DATA_DIR (reference data/dataset.py TrackSynDataset; .npz mirror of the HDF5 schema) or,
        only when the worker was constructed from a `settings` object without a data root (tests, bench, demo runs), the
        in-memory synthetic scenes of synth.py"""
        if self.data_root is not None:
            from ..data.dataset import TrackNpzDataset
            return TrackNpzDataset(self.settings_path, paths, track_length=self.track_length, train=train, data_aug=train,
                                   load_flow_data=True, load_primary_data=self.load_primary_data, load_pseudo_gt=pseudo,
                                   data_type=self.data_type)
        from ..data.dataset import SyntheticTrackDataset
        return SyntheticTrackDataset(self.settings, n_synth, self.track_length, seed=seed,
                                     load_primary_data=self.load_primary_data, load_pseudo_gt=pseudo)

    def get_train_set(self):
        return self._make_dataset(getattr(self, 'train_paths', None), True, self._train_pseudo_gt(), 64, 1234)

    def get_test_sets(self):
        test_sets = worker.TestSets()
        test_set = self._make_dataset(getattr(self, 'test_paths', None), False, self.use_pseudo_gt, 8, 99)
        test_sets.append('simple', test_set, test_frequency=1)
        self.build_losses(test_set, device=self.train_device)
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
            vals.append((val, 1.0 / (2 ** s)))
        # smoothness
        amb0 = self.data['ambient0']
        amb0 = amb0.contiguous().view(-1, *amb0.shape[2:])
        o = out[0].view(-1, *out[0].shape[2:])
        vals.append((self.disparity_loss(o, amb0), 0.8))
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
        allv = ge_loss.forward_all(depth, R, t, flow_out, amb, primary_depth)   # (one launch each way for the 12 directional terms)
        if allv is not None:
            vals += [(v, 0.2 / ge_num) for v in allv]
        ds = depth.unbind(0) if allv is None else ()
        accs = [ops.GradAccum() for _ in ds] if depth.requires_grad else None
        for tidx0 in range(len(ds)):
            for tidx1 in range(tidx0 + 1, len(ds)):
                val = ge_loss(ds[tidx0], ds[tidx1], R[tidx0], t[tidx0], R[tidx1], t[tidx1],
                              flow_out[f'flow_{tidx0}{tidx1}'], flow_out[f'flow_{tidx1}{tidx0}'], amb[tidx0],
                              amb[tidx1], primary_depth[tidx0], primary_depth[tidx1],
                              accs=(accs[tidx0], accs[tidx1]) if accs is not None else None)
                vals.append((val, 0.2 / ge_num))
        # warm-up terms
        if train:
            if self.current_epoch < 2:
                vals.append((ops.l1_mean(out[0], self.data['primary_disp']), 0.1))
            if self.current_epoch < self.warmup_epochs and self.data_type == 'real':
                vals.append((self.sgm_warmup_term(out[0]), 0.1))
        # (value, weight) pairs -> the weighted terms the reference returns, with their sum as one autograd node (ops.LossTerms)
        return ops.weighted_terms(vals)

    def sgm_warmup_term(self, o, k=0):
        """reference :168-173 / single_frame_worker.py:158-163: masked L1 to the SGM disparities (valid where > 30) with
        1.5 * N(0,1) noise drawn on the HOST generator and copied over, exactly like the reference's
        `torch.randn(o.size()).cuda()`; a captured step supplies draw k as data['_sgm_noise<k>'] instead."""
        sgm = self.data['sgm_disp']
        noise = self.data.get(f'_sgm_noise{k}')
        if noise is None:
            noise = (1.5 * torch.randn(o.size())).to(o.device)
        return ops.sgm_l1(o, sgm, noise, 30.0)

    # ------------------------------------------------------------------ evaluation (reference :176-186, :236-263)
    def callback_test_start(self, epoch, set_idx):
        from ..co import metric
        self.metric = metric.MultipleMetric(metric.DistanceMetric(vec_length=1),
                                            metric.OutlierFractionMetric(vec_length=1, thresholds=[0.1, 0.5, 1, 2, 5]))

    def callback_test_add(self, epoch, set_idx, batch_idx, n_batches, output, masks):
        if not (isinstance(output, tuple) or isinstance(output, list)):
            output = [output]
        gt = self.data['disp0']
        es = output[0].detach() * (gt > 0)      # reference numpy_in_out :176-186, kept on the device
        self.metric.add(es.reshape(-1, 1), gt.reshape(-1, 1))

    def callback_test_stop(self, epoch, set_idx, loss):
        vals = self.metric.get()   # merged over the ranks
        logging.info(', '.join(f'{k}={v:.5f}' for k, v in vals.items()))
        for k, v in vals.items():
            self.metric_add_test(epoch, set_idx, k, v)
