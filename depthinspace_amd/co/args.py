"""Command-line flags of the reference (/root/reference/co/args.py:30-73): same names and defaults."""
import argparse


def str2bool(v):
    """reference co/utils.py:32-40"""
    if isinstance(v, bool):
        return v
    if v.lower() in ('yes', 'true', 't', 'y', '1'):
        return True
    if v.lower() in ('no', 'false', 'f', 'n', '0'):
        return False
    raise argparse.ArgumentTypeError('Boolean value expected.')


def get_parser():
    parser = argparse.ArgumentParser()
    parser.add_argument('--data_type', default='synthetic', choices=['synthetic', 'real'], type=str)
    parser.add_argument('--cmd', help='Start training or test', default='resume',
                        choices=['retrain', 'resume', 'retest', 'test_init'], type=str)
    parser.add_argument('--epoch', help='If larger than -1, retest on the specified epoch', default=-1, type=int)
    parser.add_argument('--epochs', help='Training epochs', default=100, type=int)
    parser.add_argument('--warmup_epochs', default=150, type=int,
                        help='Number of epochs where SGM Disparities are used as supervisor on the real dataset')
    parser.add_argument('--lcn_radius', help='Radius of the window for LCN pre-processing', default=5, type=int)
    parser.add_argument('--max_disp', help='Maximum disparity', default=128, type=int)
    parser.add_argument('--track_length', help='Track length for geometric loss', default=4, type=int)
    parser.add_argument('--train_batch_size', help='Train Batch Size', default=8, type=int)
    parser.add_argument('--architecture', help='The architecture which will be used', default='single_frame',
                        choices=['single_frame', 'multi_frame'], type=str)
    parser.add_argument('--use_pseudo_gt', help='Only applicable in single-frame model', default=False, type=str2bool)
    return parser


def parse_args(argv=None):
    args, _ = get_parser().parse_known_args(argv)
    return args
