"""Entry point with the reference's CLI (/root/reference/train_val.py:26-58; flags of co/args.py) on the HIP path.

    python train_val.py --architecture multi_frame|single_frame --train_batch_size 4 [--cmd retrain|resume|retest|test_init]

With a config.json (DATA_DIR, OUTPUT_DIR) it trains on the tracks under DATA_DIR (the reference's on-disk schema, .npz
mirror: depthinspace_amd/data/dataset.py), which is how the three stages chain: DIS-SF -> presave_disp -> DIS-MF ->
presave_disp -> DIS-FTSF.  Without one it trains on the in-memory synthetic default-pattern scenes of
`depthinspace_amd.synth` (demo / smoke runs).

Data parallel (new; the reference is single-GPU): one process per GPU,

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 train_val.py --architecture ...

every rank trains on its own shard of the tracks, gradients are all-reduced over RCCL (bucketed, overlapped with the
backward pass), rank 0 writes the checkpoints.  DIS_TRAIN_GRAPH=1 captures the step in one hipGraph (single GPU);
DIS_ACT_DTYPE=bf16 selects bf16 activation storage for the single-frame network."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

import torch
from depthinspace_amd.co.args import parse_args
from depthinspace_amd.model import multi_frame_worker, multi_frame_networks, single_frame_worker, networks
from depthinspace_amd.trainer import FlatAdam, init_distributed
from depthinspace_amd import synth


def main():
    args = parse_args()
    rank, world, local_rank = init_distributed()  # before the first HIP call of the process
    if args.use_pseudo_gt and args.architecture != 'single_frame':
        print('Using pseudo-gt is only possible in single-frame architecture')
        raise NotImplementedError
    cfg = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'config.json')
    settings = None if os.path.exists(cfg) else synth.make_settings()
    out_dir = None if os.path.exists(cfg) else os.environ.get('DIS_OUTPUT_DIR', './output')
    if args.architecture == 'single_frame':
        worker = single_frame_worker.Worker(args, settings=settings, output_dir=out_dir)
        # DIS_ACT_DTYPE=bf16: DispNetS with bf16 activation storage (BASELINE config 2; parameters / disparities / losses fp32,
        # held to its own tolerance - tests/test_sf_bf16_gpu.py).  Default: the fp32 parity path.
        act = {'fp32': torch.float32, 'f32': torch.float32, 'bf16': torch.bfloat16}[os.environ.get('DIS_ACT_DTYPE', 'fp32')]
        net = networks.DispDecoder(channels_in=2, max_disp=args.max_disp, imsizes=worker.imsizes,
                                   act_dtype=act).to(worker.train_device)
    else:
        worker = multi_frame_worker.Worker(args, settings=settings, output_dir=out_dir)
        net = multi_frame_networks.FuseNet(imsize=worker.imsizes[0], K=worker.K, baseline=worker.baseline,
                                           track_length=worker.track_length,
                                           max_disp=args.max_disp).to(worker.train_device)
    optimizer = FlatAdam(net.parameters(), lr=1e-4)   # world size / process group from torch.distributed
    worker.do(net, optimizer, cmd=args.cmd, epoch=args.epoch)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
