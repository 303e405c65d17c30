"""Inference / presave path: run a trained DIS-SF or DIS-MF network over every track of a dataset
root and store its disparities next to the frames, exactly what /root/reference/data/presave_disp.py:84-117 does with
HDF5 (`single_frame_disp` feeds DIS-MF as `primary_disp`, `multi_frame_disp` feeds DIS-FTSF as `pseudo_gt`).
Same kernels as training, under no_grad, one track (4 frames) per call."""
import os

import numpy as np
import torch

from ..model import networks, multi_frame_networks
from .dataset import load_settings


def presave_disp(architecture, net, data_root, device='cuda'):
    """architecture: 'single_frame' | 'multi_frame'.  Writes <sample>/<architecture>_disp.npz (disp (4,1,H,W)) for every
    sample directory of data_root.  Returns the number of tracks processed."""
    settings = load_settings(data_root)
    d2d = networks.DispToDepth(float(settings.K[0, 0]), float(settings.baseline))
    lcn_in = networks.LCN(5, 0.05)
    net = net.to(device).eval()
    samples = sorted(os.path.join(data_root, o) for o in os.listdir(data_root) if os.path.isdir(os.path.join(data_root, o)))
    with torch.no_grad():
        for sp in samples:
            with np.load(os.path.join(sp, 'frames.npz')) as f:
                im = torch.from_numpy(f['im']).to(device)
                amb = torch.from_numpy(f['ambient']).to(device)
                R = torch.from_numpy(f['R']).to(device)
                t = torch.from_numpy(f['t']).to(device)
            im_lcn, _ = lcn_in(im)
            im2 = torch.cat([im_lcn, im], dim=1)  # (4,2,H,W)
            if architecture == 'single_frame':
                disp = net(im2)[0]
            else:
                with np.load(os.path.join(sp, 'flow.npz')) as f:
                    flow = {k: torch.from_numpy(f[k]).to(device) for k in f.files}
                with np.load(os.path.join(sp, 'single_frame_disp.npz')) as f:
                    primary = torch.from_numpy(f['disp']).to(device)
                disp = net(im2.unsqueeze(1), amb.unsqueeze(1), primary.unsqueeze(1), d2d(primary.unsqueeze(1).contiguous()),
                           R.unsqueeze(1), t.unsqueeze(1), flow)[:, 0]
            np.savez(os.path.join(sp, f'{architecture}_disp.npz'), disp=disp.detach().cpu().numpy())
    return len(samples)
