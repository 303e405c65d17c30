"""Entry point with the reference's CLI (/root/reference/train_val.py:26-58; flags of co/args.py) on the HIP path.

    python train_val.py --architecture multi_frame|single_frame --train_batch_size 4 [--cmd retrain|resume|retest|test_init]

Without a DATA_DIR/settings.pkl (config.json) it trains on the in-memory synthetic default-pattern scenes of
`depthinspace_amd.synth` (the reference's HDF5 dataset layer is out of scope)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

import torch
from depthinspace_amd.co.args import parse_args
from depthinspace_amd.model import multi_frame_worker, multi_frame_networks, single_frame_worker, networks
from depthinspace_amd.trainer import FlatAdam
from depthinspace_amd import synth


def main():
    args = parse_args()
    if args.use_pseudo_gt and args.architecture != 'single_frame':
        print('Using pseudo-gt is only possible in single-frame architecture')
        raise NotImplementedError
    cfg = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'config.json')
    settings = None if os.path.exists(cfg) else synth.make_settings()
    out_dir = None if os.path.exists(cfg) else os.environ.get('DIS_OUTPUT_DIR', './output')
    if args.architecture == 'single_frame':
        worker = single_frame_worker.Worker(args, settings=settings, output_dir=out_dir)
        net = networks.DispDecoder(channels_in=2, max_disp=args.max_disp, imsizes=worker.imsizes).to(worker.train_device)
    else:
        worker = multi_frame_worker.Worker(args, settings=settings, output_dir=out_dir)
        net = multi_frame_networks.FuseNet(imsize=worker.imsizes[0], K=worker.K, baseline=worker.baseline,
                                           track_length=worker.track_length,
                                           max_disp=args.max_disp).to(worker.train_device)
    optimizer = FlatAdam(net.parameters(), lr=1e-4)
    worker.do(net, optimizer, cmd=args.cmd, epoch=args.epoch)


if __name__ == '__main__':
    main()
