#!/usr/bin/env python3
"""Feature-loader throughput on files of the BASELINE shape (36 x 2048 fp32 regions per image, np.savez_compressed
like scripts/prepro_feats.py writes them): the streaming FeatureLoader (decompress + pad every epoch, what the reference does)
against the HBM-resident store from its second epoch on.  The SCST step consumes 10 images per 4.7 ms = 2 100 images/s per GPU.
    python scripts/loader_bench.py [n_images] [device]"""
import json
import os
import sys
import tempfile
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'imagecaptioning', 'pytorch_amd'))
from captioning.utils import opts                                   # noqa: E402
from captioning.data.feature_loader import FeatureLoader           # noqa: E402
from captioning.data.resident import ResidentFeatures              # noqa: E402

def main():
    n_img = int(sys.argv[1]) if len(sys.argv) > 1 else 400
    device = sys.argv[2] if len(sys.argv) > 2 else ('cuda' if torch.cuda.is_available() else 'cpu')
    tmp = tempfile.mkdtemp()
    att = os.path.join(tmp, 'att')
    os.mkdir(att)
    rng = np.random.default_rng(0)
    labels, start, end, images = [], [], [], []
    for i in range(n_img):
        feat = np.clip(rng.standard_normal((36, 2048)), 0, None).astype(np.float32)      # post-ReLU CNN features: half zeros
        np.savez_compressed(os.path.join(att, '%d.npz' % i), feat=feat)
        start.append(len(labels) + 1)
        for _ in range(5):
            row = np.zeros(16, dtype=np.uint32)
            row[:10] = rng.integers(1, 9488, size=10)
            labels.append(row)
        end.append(len(labels))
        images.append({'id': i, 'split': 'train'})
    json.dump({'images': images, 'ix_to_word': {str(i): 'w%d' % i for i in range(1, 9488)}}, open(os.path.join(tmp, 'd.json'), 'w'))
    np.savez(os.path.join(tmp, 'l.npz'), labels=np.stack(labels), label_start_ix=np.array(start, dtype=np.uint32),
             label_end_ix=np.array(end, dtype=np.uint32))
    argv = ['--input_json', os.path.join(tmp, 'd.json'), '--input_label_h5', os.path.join(tmp, 'l.npz'), '--input_att_dir', att,
            '--batch_size', '10', '--seq_per_img', '5']
    size = sum(os.path.getsize(os.path.join(att, f)) for f in os.listdir(att)) / n_img


    def rate(loader, batches, sync):
        t0 = time.perf_counter()
        for _ in range(batches):
            b = loader.get_batch('train')
        if sync:
            torch.cuda.synchronize()
        return batches * 10 / (time.perf_counter() - t0), b


    per_epoch = n_img // 10
    for workers, procs in ((1, False), (4, False), (8, False), (16, False), (4, True), (8, True)):
        ld = FeatureLoader(opts.parse_opt(argv), workers=workers, processes=procs)
        rate(ld, 2, False)                                   # worker start-up
        r, _ = rate(ld, per_epoch, False)
        print('streaming FeatureLoader, %2d decode %s: %7.0f images/s (%.0f KB compressed per image)'
              % (workers, 'processes' if procs else 'threads  ', r, size / 1e3), flush=True)
        ld.pool.shutdown()
    res = ResidentFeatures(FeatureLoader(opts.parse_opt(argv), workers=4), device)
    r1, _ = rate(res, per_epoch, device != 'cpu')
    r2, b = rate(res, 3 * per_epoch, device != 'cpu')
    print('resident store on %s: first epoch %7.0f images/s, later epochs %7.0f images/s (%.1f MB resident, att_feats %s on %s)'
          % (device, r1, r2, res.resident_bytes / 1e6, tuple(b['att_feats'].shape), b['att_feats'].device), flush=True)


if __name__ == '__main__':       # worker processes are spawned: they re-import this file
    main()
