#!/usr/bin/env python3
"""One-time converter for the CIDEr-D document-frequency table: the pickle written by the reference's
scripts/prepro_ngrams.py:79-80 ({'document_frequency': {tuple -> count}, 'ref_len': int}, keyed by token-id strings when
built with --bpe 0 and idxs) becomes the flat open-addressing image that capmi_ciderd_score probes.

    python -m imagecaptioning.pytorch_amd.tools.convert_df data/coco-train-idxs.p        # -> data/coco-train-idxs.capmi.npz

rewards.init_scorer('coco-train-idxs') picks the image up automatically when it sits next to the pickle.
"""
import os
import pickle
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(HERE))))


def convert(src, dst=None):
    from imagecaptioning.pytorch_amd.ciderd import save_df_image
    with open(src, 'rb') as f:
        pkl = pickle.load(f, encoding='latin1')
    dst = dst or os.path.splitext(src)[0] + '.capmi.npz'
    save_df_image(pkl['document_frequency'], pkl['ref_len'], dst)
    return dst


if __name__ == '__main__':
    if len(sys.argv) not in (2, 3):
        raise SystemExit(__doc__)
    print(convert(*sys.argv[1:]))
