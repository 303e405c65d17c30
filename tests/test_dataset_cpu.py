"""On-disk .npz batch schema: write/load round trip, loader layout, flow re-keying under the
training-time frame permutation (reference data/dataset.py:82-85,114-117).  No GPU."""
import os

import numpy as np
import torch


def test_npz_schema_roundtrip_and_flow_rekeying(tmp_path):
    from depthinspace_amd import synth
    from depthinspace_amd.data import dataset as D
    st = synth.make_settings(32, 24)
    root = str(tmp_path / 'd')
    paths = D.write_synthetic_dataset(root, st, 3, seed=9)
    s2 = D.load_settings(root)
    assert s2.imsize == (32, 24) and np.array_equal(s2.K, st.K) and abs(s2.baseline - st.baseline) < 1e-12
    f = np.load(os.path.join(paths[1], 'frames.npz'))
    assert f['im'].shape == (4, 1, 32, 24) and f['R'].shape == (4, 3, 3) and f['t'].shape == (4, 3)
    fl = np.load(os.path.join(paths[1], 'flow.npz'))
    assert len(fl.files) == 12 and fl['flow_03'].shape == (1, 2, 32, 24)
    # deterministic test-time sample == the generator's batch
    ds = D.TrackNpzDataset(root, paths, 4, train=False, load_flow_data=True)
    s = ds[1]
    b = synth.make_batch(st, 1, 4, seed=10, with_primary=False)
    assert torch.equal(s['im0'], torch.from_numpy(b['im0'][0])) and torch.equal(s['flow_21'], torch.from_numpy(b['flow_21'][0]))
    batch = D.collate([ds[0], ds[2]])
    assert batch['im0'].shape == (2, 4, 1, 32, 24) and batch['flow_01'].shape == (2, 1, 2, 32, 24)
    # training: frames are permuted and flow_{i0 i1} must be the stored flow between the permuted frames
    np.random.seed(3)
    dt = D.TrackNpzDataset(root, paths, 4, train=True, load_flow_data=True)
    np.random.seed(3)
    perm = np.random.permutation(4)
    np.random.seed(3)
    t = dt[1]
    for i0 in range(4):
        assert torch.equal(t['im0'][i0], torch.from_numpy(f['im'][perm[i0]]))
        for i1 in range(4):
            if i0 != i1:
                assert torch.equal(t[f'flow_{i0}{i1}'], torch.from_numpy(fl[f'flow_{perm[i0]}{perm[i1]}']))
