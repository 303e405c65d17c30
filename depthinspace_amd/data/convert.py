"""Reference dataset (HDF5 tracks + settings.pkl) -> this package's on-disk schema (.npz files, same names / keys / shapes).

The reference writes one directory per track (`data/create_syn_data.py:245-255` frames.hdf5 with im / ambient / grad / disp /
R / t [/ sgm_disp]; the LiteFlowNet presave writes flow.hdf5 with flow_ij; `data/presave_disp.py:116-117` writes
single_frame_disp.hdf5 / multi_frame_disp.hdf5 with disp) plus `settings.pkl` (`create_syn_data.py:327-337`: imsize, pattern,
baseline, K) and reads them back in `data/dataset.py:90-125`.  `TrackNpzDataset` (dataset.py here) reads the same datasets from
.npz files; this module copies them over, dataset by dataset, without touching the values.

h5py is not part of this image: `open_h5` is any callable path -> context manager of a mapping name -> array-like (h5py.File
in real use, a stand-in in the CPU test)."""
import os
import pickle

import numpy as np

TRACK_FILES = ('frames', 'flow', 'single_frame_disp', 'multi_frame_disp')
FRAME_KEYS = ('im', 'ambient', 'grad', 'disp', 'R', 't')   # + 'sgm_disp' on `real` data


def _h5py_open(path):
    try:
        import h5py
    except ImportError as e:   # pragma: no cover
        raise SystemExit('convert_hdf5 needs h5py to read the reference files (pip install h5py on the machine that holds '
                         'the dataset); the training path itself reads the converted .npz files only') from e
    return h5py.File(path, 'r')


def convert_settings(src_root, dst_root):
    """settings.pkl (a dict with imsize, pattern, baseline, K) -> settings.npz"""
    with open(os.path.join(src_root, 'settings.pkl'), 'rb') as f:
        s = pickle.load(f)
    os.makedirs(dst_root, exist_ok=True)
    np.savez(os.path.join(dst_root, 'settings.npz'), imsize=np.asarray(s['imsize']), K=np.asarray(s['K'], np.float32),
             baseline=np.float64(s['baseline']), pattern=np.asarray(s['pattern']))


def convert_track(src_dir, dst_dir, open_h5=_h5py_open):
    """every <name>.hdf5 of one track directory -> <name>.npz with identical dataset names, dtypes and shapes.
    Returns the list of files written.  frames.hdf5 must hold FRAME_KEYS."""
    os.makedirs(dst_dir, exist_ok=True)
    written = []
    for name in TRACK_FILES:
        src = os.path.join(src_dir, name + '.hdf5')
        if not os.path.exists(src):
            if name == 'frames':
                raise FileNotFoundError(src)
            continue
        with open_h5(src) as f:
            arrays = {k: np.asarray(f[k][...] if hasattr(f[k], 'shape') and not isinstance(f[k], np.ndarray) else f[k])
                      for k in f.keys()}
        if name == 'frames':
            missing = [k for k in FRAME_KEYS if k not in arrays]
            if missing:
                raise KeyError(f'{src}: datasets {missing} missing')
        np.savez(os.path.join(dst_dir, name + '.npz'), **arrays)
        written.append(name + '.npz')
    return written


def convert_dataset(src_root, dst_root, open_h5=_h5py_open, log=print):
    """whole dataset root: settings + every track directory (the reference names them %08d)"""
    convert_settings(src_root, dst_root)
    n = 0
    for d in sorted(os.listdir(src_root)):
        src = os.path.join(src_root, d)
        if os.path.isdir(src) and os.path.exists(os.path.join(src, 'frames.hdf5')):
            w = convert_track(src, os.path.join(dst_root, d), open_h5)
            n += 1
            log(f'{d}: {", ".join(w)}')
    return n
