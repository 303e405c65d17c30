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


class _FakeH5(dict):
    """stand-in for h5py.File(path, 'r'): a mapping name -> array, usable as a context manager (h5py is not in the image;
    the converter takes the opener as an argument)"""

    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


def test_reference_hdf5_layout_converts_to_the_npz_schema(tmp_path):
    """scripts/convert_hdf5.py / data/convert.py: a dataset root in the REFERENCE's layout (settings.pkl, %08d/frames.hdf5,
    flow.hdf5, single_frame_disp.hdf5: data/create_syn_data.py:245-255,327-337, data/presave_disp.py:116-117) becomes a root
    TrackNpzDataset loads, value for value."""
    import pickle
    from depthinspace_amd import synth
    from depthinspace_amd.data import convert, dataset as D
    st = synth.make_settings(32, 24)
    src, dst = tmp_path / 'ref', tmp_path / 'npz'
    src.mkdir()
    with open(src / 'settings.pkl', 'wb') as f:
        pickle.dump({'imsize': st.imsize, 'pattern': st.pattern, 'baseline': st.baseline, 'K': st.K}, f)
    store = {}
    for i in range(2):
        b = synth.make_batch(st, 1, 4, seed=40 + i, with_primary=True)
        d = src / f'{i:08d}'
        d.mkdir()
        im = b['im0'][0]
        store[str(d / 'frames.hdf5')] = _FakeH5(im=im, ambient=b['ambient0'][0], grad=np.zeros_like(im), disp=b['disp0'][0],
                                                R=b['R'][0], t=b['t'][0], sgm_disp=b['disp0'][0] + 1)
        store[str(d / 'flow.hdf5')] = _FakeH5({k: v[0] for k, v in b.items() if k.startswith('flow_')})
        store[str(d / 'single_frame_disp.hdf5')] = _FakeH5(disp=b['primary_disp'][0])
        for k in store:
            open(k, 'wb').close()   # the converter looks for the files
    n = convert.convert_dataset(str(src), str(dst), open_h5=lambda p: store[p], log=lambda s: None)
    assert n == 2
    s2 = D.load_settings(str(dst))
    assert s2.imsize == st.imsize and np.array_equal(s2.K, st.K) and np.array_equal(s2.pattern, st.pattern)
    paths = [str(dst / f'{i:08d}') for i in range(2)]
    ds = D.TrackNpzDataset(str(dst), paths, 4, train=False, load_flow_data=True, load_primary_data=True, data_type='real')
    s = ds[1]
    b = synth.make_batch(st, 1, 4, seed=41, with_primary=True)
    for k in ('im0', 'ambient0', 'disp0', 'R', 't', 'primary_disp', 'flow_30'):
        assert torch.equal(s[k], torch.from_numpy(b[k][0])), k
    assert torch.equal(s['sgm_disp'], torch.from_numpy(b['disp0'][0] + 1))
    # a track without frames.hdf5 is an error, not a silent skip
    import pytest
    (src / 'broken').mkdir()
    with pytest.raises(FileNotFoundError):
        convert.convert_track(str(src / 'broken'), str(dst / 'broken'), open_h5=lambda p: store[p])
