#!/usr/bin/env python3
"""One-time conversion of the reference's label file (scripts/prepro_labels.py:158-163: an HDF5 with ``labels``,
``label_start_ix``, ``label_end_ix``, ``label_length``) into an ``.npz``.  The loader reads the reference's file directly (captioning/data/h5lite.py, no h5py); this converter is
for files h5lite refuses (chunked / compressed datasets, libver='latest') -- those need h5py.

    python -m imagecaptioning.pytorch_amd.tools.convert_labels data/cocotalk_label.h5 data/cocotalk_label.npz
"""
import sys

import numpy as np


def convert(h5_path, npz_path):
    names = ('labels', 'label_start_ix', 'label_end_ix', 'label_length')
    try:
        from ..captioning.data import h5lite
        f = h5lite.H5File(h5_path)
        arrays = {k: f[k] for k in names if k in f}
    except Exception:
        import h5py
        with h5py.File(h5_path, 'r') as f:
            arrays = {k: f[k][:] for k in names if k in f}
    np.savez(npz_path, **arrays)
    return npz_path


if __name__ == '__main__':
    print(convert(sys.argv[1], sys.argv[2]))
