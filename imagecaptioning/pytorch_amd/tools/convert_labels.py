#!/usr/bin/env python3
"""One-time conversion of the reference's label file (scripts/prepro_labels.py:158-163: an HDF5 with ``labels``,
``label_start_ix``, ``label_end_ix``, ``label_length``) into the ``.npz`` the h5py-free loader reads
(captioning/data/feature_loader.py).  Needs h5py -- run it wherever the reference's preprocessing ran.

    python -m imagecaptioning.pytorch_amd.tools.convert_labels data/cocotalk_label.h5 data/cocotalk_label.npz
"""
import sys

import numpy as np


def convert(h5_path, npz_path):
    import h5py
    with h5py.File(h5_path, 'r') as f:
        arrays = {k: f[k][:] for k in ('labels', 'label_start_ix', 'label_end_ix', 'label_length') if k in f}
    np.savez(npz_path, **arrays)
    return npz_path


if __name__ == '__main__':
    print(convert(sys.argv[1], sys.argv[2]))
