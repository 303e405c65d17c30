"""The rows either side of the hot path on the HIP path, end to end at 64x64:
on-disk .npz schema -> DIS-SF training through Worker.do (retrain, then resume from state.dict) -> presave of the
DIS-SF disparities -> DIS-MF training on them -> presave of the DIS-MF disparities -> DIS-FTSF with pseudo-GT."""
import argparse
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _args(arch, epochs, bs=2, pgt=False):
    return argparse.Namespace(use_pseudo_gt=pgt, lcn_radius=5, track_length=4, data_type='synthetic', architecture=arch,
                              epochs=epochs, warmup_epochs=150, train_batch_size=bs, max_disp=128)


def test_sf_mf_ftsf_pipeline(tmp_path):
    from depthinspace_amd import synth
    from depthinspace_amd.data import dataset as D
    from depthinspace_amd.data.presave_disp import presave_disp
    from depthinspace_amd.model import networks, multi_frame_networks, single_frame_worker, multi_frame_worker
    from depthinspace_amd.trainer import FlatAdam
    H = W = 64
    settings = synth.make_settings(H, W)
    root = str(tmp_path / 'data')
    paths = D.write_synthetic_dataset(root, settings, 6, seed=50)
    assert sorted(os.listdir(paths[0])) == ['flow.npz', 'frames.npz']

    def with_sets(worker_cls, pgt, primary):
        class W_(worker_cls):
            def _mk(self, train):
                return D.TrackNpzDataset(root, paths[:4] if train else paths[4:], 4, train=train, load_flow_data=True,
                                         load_primary_data=primary, load_pseudo_gt=pgt)

            def get_train_set(self):
                return self._mk(True)

            def get_test_sets(self):
                from depthinspace_amd.model.worker import TestSets
                ts = TestSets()
                ts.append('simple', self._mk(False), test_frequency=1)
                self.build_losses()
                return ts
        return W_

    # ---- DIS-SF: retrain 1 epoch, then resume to epoch 2 (state.dict / net_%04d.params / metrics.json layout)
    out = str(tmp_path / 'out')
    st = D.load_settings(root)
    w = with_sets(single_frame_worker.Worker, False, False)(_args('single_frame', 1), settings=st, output_dir=out, num_workers=0)
    net = networks.DispDecoder(channels_in=2, max_disp=128, imsizes=w.imsizes).cuda()
    w.do(net, FlatAdam(net.parameters(), lr=1e-4), cmd='retrain')
    exp = os.path.join(out, 'single_frame')
    assert os.path.exists(os.path.join(exp, 'state.dict')) and os.path.exists(os.path.join(exp, 'net_0000.params'))
    sd = torch.load(os.path.join(exp, 'net_0000.params'))
    assert len(sd) == 64 and 'disp_decoder.conv1.0.weight' in sd  # reference state_dict keys
    w2 = with_sets(single_frame_worker.Worker, False, False)(_args('single_frame', 2), settings=st, output_dir=out, num_workers=0)
    net2 = networks.DispDecoder(channels_in=2, max_disp=128, imsizes=w2.imsizes).cuda()
    w2.do(net2, FlatAdam(net2.parameters(), lr=1e-4), cmd='resume')
    assert os.path.exists(os.path.join(exp, 'net_0001.params'))
    import json
    m = json.load(open(os.path.join(exp, 'metrics.json')))
    assert set(m.keys()) == {'0', '1'} and len(m['1']['train']['loss']) == 11
    # ---- presave DIS-SF -> primary_disp for DIS-MF
    assert presave_disp('single_frame', net2, root) == 6
    d = np.load(os.path.join(paths[0], 'single_frame_disp.npz'))['disp']
    assert d.shape == (4, 1, H, W) and np.isfinite(d).all() and d.min() >= 0 and d.max() <= 128
    # ---- DIS-MF on the presaved disparities
    wm = with_sets(multi_frame_worker.Worker, False, True)(_args('multi_frame', 1), settings=st, output_dir=out, num_workers=0)
    netm = multi_frame_networks.FuseNet(imsize=wm.imsizes[0], K=wm.K, baseline=wm.baseline, track_length=4, max_disp=128).cuda()
    wm.do(netm, FlatAdam(netm.parameters(), lr=1e-4), cmd='retrain')
    assert len(torch.load(os.path.join(out, 'multi_frame', 'net_0000.params'))) == 236
    assert presave_disp('multi_frame', netm, root) == 6
    # ---- DIS-FTSF: single-frame net with the multi-frame disparities as pseudo ground truth
    wf = with_sets(single_frame_worker.Worker, True, False)(_args('single_frame', 3, pgt=True), settings=st, output_dir=out, num_workers=0)
    wf.do(net2, FlatAdam(net2.parameters(), lr=1e-4), cmd='resume')
    m = json.load(open(os.path.join(exp, 'metrics.json')))
    assert len(m['2']['train']['loss']) == 15 and all(np.isfinite(m['2']['train']['loss']))
