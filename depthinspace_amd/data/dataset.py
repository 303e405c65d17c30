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
