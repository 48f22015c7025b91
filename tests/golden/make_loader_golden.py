#!/usr/bin/env python3
"""Golden batches of the REFERENCE's own data loader (captioning/data/dataloader.py) on a tiny dataset that is committed next to
them -- what pins captioning/data/feature_loader.py to the reference instead of to a hand-restated contract.

Run only in the build container (``/root/reference`` does not exist on the GPU box):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_loader_golden.py

The reference module imports ``h5py`` and ``lmdbdict``, neither of which is installed for this interpreter.  They are only
touched for (a) reading the four label arrays and (b) lmdb feature stores (unused: the dataset is a directory of .npy / .npz
files), so both are stubbed: ``h5py.File`` serves the label arrays this script also writes as ``labels.npz`` for our loader (the
real-h5 reading of those arrays is pinned separately: tests/golden/make_h5.py -> tests/test_feature_loader.py).  Everything else --
the Dataset, its collate function, the sampler, torch's DataLoader with 4 worker processes -- is the reference's own code, only
*called*.

What is deterministic in the reference and therefore stored verbatim: the val / test splits (no shuffling; every image there has
exactly seq_per_img captions, so ``get_captions`` has no choice to make), including the partial last batch and the restart of
the split.  The train split is shuffled by the global numpy RNG and its captions are drawn by ``random`` inside worker
processes: stored are the fields that do not depend on those draws (``bounds`` of every batch, batch sizes) and, per batch, the
image indices, so that the test can check every row against the image it claims to be."""
import argparse
import json
import os
import sys
import types

import numpy as np

REF = os.environ.get('CAPMI_REFERENCE', '/root/reference')
HERE = os.path.dirname(os.path.abspath(__file__))
DS = os.path.join(HERE, 'loader_ds')


def build_dataset():
    """11 images: 5 train, 1 restval, 3 val, 2 test; 3..7 regions of 6 features; vocabulary of 20 words, seq_length 5"""
    rng = np.random.default_rng(20240924)
    os.makedirs(os.path.join(DS, 'fc'), exist_ok=True)
    os.makedirs(os.path.join(DS, 'att'), exist_ok=True)
    splits = ['train', 'val', 'train', 'restval', 'test', 'train', 'val', 'train', 'test', 'val', 'train']
    ncaps = {'train': [5, 1, 3, 2, 4], 'restval': [2], 'val': [2, 2, 2], 'test': [2, 2]}       # seq_per_img = 2
    images, labels, start, end = [], [], [], []
    for i, sp in enumerate(splits):
        img_id = 1000 + 7 * i
        images.append({'id': img_id, 'split': sp, 'file_path': 'img/%d.jpg' % img_id})
        k = int(rng.integers(3, 8))
        if sp == 'val' and i == 6:
            k = 7                                                    # val batch 0 = images 1 and 6: different region counts
        att = rng.standard_normal((k, 6)).astype(np.float32)
        np.savez_compressed(os.path.join(DS, 'att', '%d.npz' % img_id), feat=att)
        if i != 5:                                                   # image 5 has no fc file: mean of the regions (:295-298)
            np.save(os.path.join(DS, 'fc', '%d.npy' % img_id), rng.standard_normal(6).astype(np.float32))
        n = ncaps[sp].pop(0)
        start.append(len(labels) + 1)
        for _ in range(n):
            ln = int(rng.integers(1, 6))
            row = np.zeros(5, dtype=np.uint32)
            row[:ln] = rng.integers(1, 21, size=ln)
            labels.append(row)
        end.append(len(labels))
    # images 9 (val) and 4, 8 (test) all have 3..7 regions; make the test pair equal-sized so that att_masks is None there (:240-241)
    for i in (4, 8):
        np.savez_compressed(os.path.join(DS, 'att', '%d.npz' % (1000 + 7 * i)), feat=rng.standard_normal((4, 6)).astype(np.float32))
    info = {'images': images, 'ix_to_word': {str(i): 'w%d' % i for i in range(1, 21)}}
    json.dump(info, open(os.path.join(DS, 'dataset.json'), 'w'))
    lab = dict(labels=np.stack(labels).astype(np.uint32), label_start_ix=np.array(start, dtype=np.uint32),
               label_end_ix=np.array(end, dtype=np.uint32),
               label_length=np.array([(r != 0).sum() for r in labels], dtype=np.uint32))
    np.savez(os.path.join(DS, 'labels.npz'), **lab)
    return lab


def install_stubs(lab):
    class _DS:
        def __init__(self, a):
            self.a, self.shape = a, a.shape

        def __getitem__(self, k):
            return self.a[k]

    class _File(dict):
        def __init__(self, path, mode='r', driver=None):
            super().__init__({k: _DS(v) for k, v in lab.items()})

    h5 = types.ModuleType('h5py')
    h5.File = _File
    sys.modules['h5py'] = h5
    ld = types.ModuleType('lmdbdict')
    ld.lmdbdict = lambda *a, **k: (_ for _ in ()).throw(RuntimeError('lmdb stores are not part of this fixture'))
    lm = types.ModuleType('lmdbdict.methods')
    lm.DUMPS_FUNC, lm.LOADS_FUNC = {'ascii': None}, {'identity': None}
    sys.modules['lmdbdict'], sys.modules['lmdbdict.methods'] = ld, lm


def flat(prefix, data, out):
    """one reference batch dict -> npz entries"""
    out[prefix + 'fc_feats'] = data['fc_feats'].numpy()
    out[prefix + 'att_feats'] = data['att_feats'].numpy()
    out[prefix + 'has_att_masks'] = np.array(data['att_masks'] is not None)
    if data['att_masks'] is not None:
        out[prefix + 'att_masks'] = data['att_masks'].numpy()
    out[prefix + 'labels'] = data['labels'].numpy()
    out[prefix + 'masks'] = data['masks'].numpy()
    out[prefix + 'n_gts'] = np.array(len(data['gts']))
    for i, g in enumerate(data['gts']):
        out[prefix + 'gts%d' % i] = np.asarray(g)
    b = data['bounds']
    out[prefix + 'bounds'] = np.array([b['it_pos_now'], b['it_max'], int(b['wrapped'])])
    out[prefix + 'ix'] = np.array([d['ix'] for d in data['infos']])
    out[prefix + 'id'] = np.array([d['id'] for d in data['infos']])
    out[prefix + 'file_path'] = np.array([d['file_path'] for d in data['infos']])


def main():
    lab = build_dataset()
    install_stubs(lab)
    sys.path.insert(0, REF)
    sys.dont_write_bytecode = True
    from captioning.data.dataloader import DataLoader        # noqa: E402  (the reference)

    out = {}
    for tag, B in (('b2.', 2), ('b4.', 4)):
        opt = argparse.Namespace(batch_size=B, seq_per_img=2, use_fc=True, use_att=True, use_box=0, norm_att_feat=0,
                                 norm_box_feat=0, input_json=os.path.join(DS, 'dataset.json'), input_label_h5='stub.h5',
                                 input_fc_dir=os.path.join(DS, 'fc'), input_att_dir=os.path.join(DS, 'att'), input_box_dir='',
                                 train_only=0, data_in_memory=False)
        np.random.seed(7)
        loader = DataLoader(opt)
        out[tag + 'vocab_size'] = np.array(loader.vocab_size)
        out[tag + 'seq_length'] = np.array(loader.seq_length)
        for split, calls in (('val', 5), ('test', 3), ('train', 8)):
            for c in range(calls):
                flat('%s%s%d.' % (tag, split, c), loader.get_batch(split), out)
        # reset_iterator (eval_utils.py calls it before every evaluation): the split starts over
        loader.get_batch('val')
        loader.reset_iterator('val')
        flat(tag + 'val_after_reset.', loader.get_batch('val'), out)
        del loader
    np.savez_compressed(os.path.join(HERE, 'loader_ref.npz'), **out)
    print('written loader_ref.npz with', len(out), 'arrays and', DS)


if __name__ == '__main__':
    main()
