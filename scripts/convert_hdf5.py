#!/usr/bin/env python
"""Convert a dataset written by the reference (data/create_syn_data.py + the presave scripts: HDF5 tracks, settings.pkl) into
the .npz schema `depthinspace_amd.data.dataset.TrackNpzDataset` trains from.  Needs h5py on the machine that runs it.

    python scripts/convert_hdf5.py /path/to/reference/DATA_DIR /path/to/npz_root
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from depthinspace_amd.data import convert

if __name__ == '__main__':
    if len(sys.argv) != 3:
        raise SystemExit(__doc__)
    n = convert.convert_dataset(sys.argv[1], sys.argv[2])
    print(f'{n} tracks converted')
