"""In-memory synthetic track dataset with the batch schema of the reference's TrackSynDataset
(/root/reference/data/dataset.py:90-125): per sample a dict of (tl, ...) arrays; a DataLoader (or
`collate`) turns them into the (bs, tl, ...) loader layout that Worker.copy_data consumes.
The reference's HDF5 reader / augmentation are host I/O and out of scope."""
import numpy as np
import torch

from .. import synth


class SyntheticTrackDataset(torch.utils.data.Dataset):
    def __init__(self, settings, n_samples, track_length=4, seed=1234, load_primary_data=True, load_pseudo_gt=False):
        self.settings = settings
        self.n = n_samples
        self.tl = track_length
        self.seed = seed
        self.primary = load_primary_data
        self.pseudo = load_pseudo_gt
        self.imsizes = [settings.imsize]
        for _ in range(3):
            self.imsizes.append((int(self.imsizes[-1][0] / 2), int(self.imsizes[-1][1] / 2)))
        self.patterns = [settings.pattern]
        self.baseline = settings.baseline
        self.focal_lengths = [settings.K[0, 0]]
        self.current_epoch = 0

    def getK(self, sidx=0):
        return self.settings.K

    def __len__(self):
        return self.n

    def __getitem__(self, idx):
        b = synth.make_batch(self.settings, 1, self.tl, seed=self.seed + idx, with_primary=self.primary,
                             with_pseudo_gt=self.pseudo)
        return {k: torch.from_numpy(v[0]) for k, v in b.items()}


def collate(samples):
    return {k: torch.stack([s[k] for s in samples], 0) for k in samples[0]}


# ----------------------------------------------------------------------------------------------------------------------
# On-disk schema.  The reference stores one directory per track with HDF5 files
# (/root/reference/data/dataset.py:90-125, data/create_syn_data.py:245-255, data/presave_disp.py:116-117); h5py is not
# available here, so the same datasets live in .npz files with the same names, keys and shapes:
#   <root>/settings.npz                 imsize (2,), K (3,3), baseline (), pattern (H,W,3)      [reference: settings.pkl]
#   <root>/<sample>/frames.npz          im, ambient, grad, disp (4,1,H,W); R (4,3,3); t (4,3); optional sgm_disp
#   <root>/<sample>/flow.npz            flow_ij (1,2,H,W) for the 12 ordered pairs i != j
#   <root>/<sample>/single_frame_disp.npz / multi_frame_disp.npz      disp (4,1,H,W)
# ----------------------------------------------------------------------------------------------------------------------
import os


def save_settings(root, settings):
    os.makedirs(root, exist_ok=True)
    np.savez(os.path.join(root, 'settings.npz'), imsize=np.asarray(settings.imsize), K=settings.K,
             baseline=np.float64(settings.baseline), pattern=settings.pattern)


def load_settings(root):
    s = np.load(os.path.join(root, 'settings.npz'))
    return synth.Settings(tuple(int(v) for v in s['imsize']), s['K'], float(s['baseline']), s['pattern'])


def write_synthetic_dataset(root, settings, n_samples, track_length=4, seed=1234):
    """Writes `n_samples` synthetic tracks in the on-disk schema (the reference's create_syn_data.py needs ShapeNet and
    CTD's renderer; this is the analytic-scene generator of synth.py).  Returns the sample directories."""
    save_settings(root, settings)
    paths = []
    for i in range(n_samples):
        b = synth.make_batch(settings, 1, track_length, seed=seed + i, with_primary=False, with_pseudo_gt=False)
        d = os.path.join(root, f'{i:08d}')
        os.makedirs(d, exist_ok=True)
        im = b['im0'][0]
        np.savez(os.path.join(d, 'frames.npz'), im=im, ambient=b['ambient0'][0], grad=np.zeros_like(im), disp=b['disp0'][0],
                 R=b['R'][0], t=b['t'][0])
        np.savez(os.path.join(d, 'flow.npz'), **{k: v[0] for k, v in b.items() if k.startswith('flow_')})
        paths.append(d)
    return paths


class TrackNpzDataset(torch.utils.data.Dataset):
    """Mirror of the reference TrackSynDataset (data/dataset.py:36-125) on the .npz schema, restricted to the keys the
    hot path consumes (full-resolution scale only; the reference also loads three down-scaled copies that nothing on
    the training path reads).  Training draws a random frame order per sample and re-keys the flows (:82-85,:114-117);
    testing is deterministic.  Augmentation (:128-186) is out of scope (host-side)."""

    def __init__(self, settings_path, sample_paths, track_length=4, train=True, data_aug=False, load_flow_data=False,
                 load_primary_data=False, load_pseudo_gt=False, data_type='synthetic'):
        assert track_length <= 4
        self.settings = load_settings(os.path.dirname(settings_path) if settings_path.endswith('.npz') else settings_path)
        self.sample_paths = list(sample_paths)
        self.track_length = track_length
        self.train = train
        self.load_flow_data = load_flow_data
        self.load_primary_data = load_primary_data
        self.load_pseudo_gt = load_pseudo_gt
        self.data_type = data_type
        s = self.settings
        self.imsizes = [(s.imsize[0] // (2 ** k), s.imsize[1] // (2 ** k)) for k in range(4)]
        self.patterns = [s.pattern]
        self.baseline = s.baseline
        self.K = s.K
        self.focal_lengths = [s.K[0, 0] / (2 ** k) for k in range(4)]
        self.current_epoch = 0

    def getK(self, sidx=0):
        return self.K

    def __len__(self):
        return len(self.sample_paths)

    def __getitem__(self, idx):
        p = self.sample_paths[idx]
        track = np.random.permutation(4)[:self.track_length] if self.train else np.arange(self.track_length)
        ret = {}
        with np.load(os.path.join(p, 'frames.npz')) as f:
            ret['im0'] = np.stack([f['im'][t] for t in track], 0)
            ret['ambient0'] = np.stack([f['ambient'][t] for t in track], 0)
            ret['disp0'] = np.stack([f['disp'][t] for t in track], 0)
            ret['R'] = np.stack([f['R'][t] for t in track], 0)
            ret['t'] = np.stack([f['t'][t] for t in track], 0)
            if self.data_type == 'real':
                ret['sgm_disp'] = np.stack([f['sgm_disp'][t] for t in track], 0)
        if self.load_flow_data:
            with np.load(os.path.join(p, 'flow.npz')) as f:
                for i0, t0 in enumerate(track):
                    for i1, t1 in enumerate(track):
                        if t0 != t1:
                            ret[f'flow_{i0}{i1}'] = f[f'flow_{t0}{t1}']
        if self.load_primary_data:
            with np.load(os.path.join(p, 'single_frame_disp.npz')) as f:
                ret['primary_disp'] = np.stack([f['disp'][t] for t in track], 0)
        if self.load_pseudo_gt:
            with np.load(os.path.join(p, 'multi_frame_disp.npz')) as f:
                ret['pseudo_gt'] = np.stack([f['disp'][t] for t in track], 0)
        return {k: torch.from_numpy(np.ascontiguousarray(v, dtype=np.float32)) for k, v in ret.items()}
