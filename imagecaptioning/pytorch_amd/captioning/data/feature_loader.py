"""Real feature / label loader with the reference's batch-dict contract (captioning/data/dataloader.py:182-299), host side of
SURVEY 8f-1/2: what feeds ``DevicePrefetcher`` when training on precomputed bottom-up features instead of synthetic data.

Inputs (same options as the reference, ``opts.py:23-37``):
  input_json      the dataset json of scripts/prepro_labels.py: ``images`` [{id, split, file_path}], ``ix_to_word``
  input_label_h5  the label file.  h5py is not installable here, so the native format is an ``.npz`` with the SAME four arrays the
                  reference's h5 holds (prepro_labels.py:158-163): ``labels`` uint32 [M, L], ``label_start_ix`` / ``label_end_ix``
                  (1-based, inclusive), ``label_length``; an ``.h5`` path is read through h5py when that module exists
                  (``tools/convert_labels.py`` turns one into the other once)
  input_fc_dir    ``<id>.npy`` pooled features [F] (optional: mean of the regions when missing, dataloader.py:295-298)
  input_att_dir   ``<id>.npz`` with key ``feat`` (or ``z``, dataloader.py:40) [K_i, F], K_i variable (10..100 for adaptive
                  bottom-up features) -- or ``<id>.npy``

``get_batch(split)`` returns CPU tensors: ``fc_feats [B,F]``, ``att_feats [B,Kmax,F]`` zero padded, ``att_masks [B,Kmax]`` or
None when every image of the batch has Kmax regions (:240-241), ``labels [B,n,L+2]`` int64 with BOS/EOS columns 0, ``masks``
(nonzeros + 2 ones, :245-249), ``gts`` (all reference rows of each image, uint32), ``bounds``, ``infos``.  Decompression of the
next images runs on a small thread pool while the device computes (the reference uses DataLoader worker processes,
:350-372)."""
import json
import multiprocessing
import os
import random
from concurrent.futures import ProcessPoolExecutor, ThreadPoolExecutor

import numpy as np
import torch


def load_labels(path):
    """-> dict(labels uint32 [M,L], label_start_ix, label_end_ix) from the reference's ``<name>_label.h5``
    (scripts/prepro_labels.py:158-163: plain contiguous datasets, read by captioning/data/h5lite.py without h5py) or from the
    ``.npz`` of tools/convert_labels.py.  A file h5lite refuses (chunked / compressed / written with libver='latest') goes to
    h5py when it is importable."""
    keys = ('labels', 'label_start_ix', 'label_end_ix')
    if str(path).endswith('.npz'):
        z = np.load(path)
        return {k: z[k] for k in keys}
    from . import h5lite
    try:
        return h5lite.read_datasets(path, keys)
    except h5lite.H5LiteError as lite_err:
        try:
            import h5py
        except ImportError as e:
            raise RuntimeError('%s: %s -- and h5py is not available here: convert the file once with tools/convert_labels.py '
                               '(on a machine that has h5py) and pass the .npz' % (path, lite_err)) from e
        with h5py.File(path, 'r') as f:
            return {k: f[k][:] for k in keys}


def decode_image(att_dir, fc_dir, img_id, use_fc, norm_att_feat):
    """(fc [F] or [0], att [K_i, F]) of one image from its files (dataloader.py:186-229, 295-298)"""
    p = os.path.join(att_dir, str(img_id) + '.npz')
    if os.path.exists(p):
        z = np.load(p)
        a = z['feat'] if 'feat' in z else z['z']
    else:
        a = np.load(os.path.join(att_dir, str(img_id) + '.npy'))
    a = np.asarray(a, dtype=np.float32).reshape(-1, a.shape[-1])
    if norm_att_feat:
        a = a / np.linalg.norm(a, 2, 1, keepdims=True)
    if not use_fc:
        return np.zeros((0,), dtype=np.float32), a
    p = os.path.join(fc_dir, str(img_id) + '.npy') if fc_dir else None
    fc = np.load(p).astype(np.float32) if p and os.path.exists(p) else a.mean(0)          # dataloader.py:295-298
    return fc, a


class EmptySplit(ValueError):
    """this rank's partition of a split holds no image (a val split smaller than the number of data-parallel ranks)"""


class FeatureLoader:
    def __init__(self, opt, workers=4, processes=None, lookahead=3, rank=0, world=1):
        """rank / world: data-parallel partition of every pass (below).  workers: size of the decode pool; processes: worker PROCESSES instead of threads (default: CAPMI_LOADER_PROCS=1).
        np.load of a compressed .npz holds the GIL for most of its time: threads top out at ~400 images/s of 36 x 2048
        features whatever their number, N processes scale to ~N x 400 (scripts/loader_bench.py)."""
        self.opt = opt
        self.batch_size = opt.batch_size
        self.seq_per_img = opt.seq_per_img
        self.info = json.load(open(opt.input_json))
        self.ix_to_word = self.info['ix_to_word']
        self.vocab_size = len(self.ix_to_word)
        lab = load_labels(opt.input_label_h5)
        self.label = lab['labels']
        self.label_start_ix, self.label_end_ix = lab['label_start_ix'], lab['label_end_ix']
        self.seq_length = int(self.label.shape[1])
        self.fc_dir, self.att_dir = getattr(opt, 'input_fc_dir', None), opt.input_att_dir
        self.use_fc = bool(getattr(opt, 'use_fc', True))
        self.norm_att_feat = bool(getattr(opt, 'norm_att_feat', 0))
        self.split_ix = {'train': [], 'val': [], 'test': []}
        for ix, img in enumerate(self.info['images']):        # dataloader.py:143-156
            sp = img.get('split')
            if sp is None:
                for k in self.split_ix:
                    self.split_ix[k].append(ix)
            elif sp in self.split_ix:
                self.split_ix[sp].append(ix)
            elif getattr(opt, 'train_only', 0) == 0:            # restval
                self.split_ix['train'].append(ix)
        # Data-parallel ranks PARTITION one shuffled pass (tools/train_pl.py:60-73 + the DistributedSampler Lightning injects:
        # `seed` is shared, rank r takes elements r, r + world, ... of the permutation, the tail padded from its head so that all
        # ranks hold equally many and wrap -- i.e. reshuffle -- on the same batch).  The order RNG is therefore seeded alike on
        # every rank; which captions of an image are drawn is per rank.
        self.rank, self.world = int(rank), max(1, int(world))
        assert 0 <= self.rank < self.world
        self.rng = random.Random(getattr(opt, 'seed', 1234))                  # epoch order (shared by the ranks)
        self.cap_rng = random.Random(getattr(opt, 'seed', 1234) + 7919 + 104729 * self.rank)   # which captions of an image (own
        #                                                       stream: the order RNG is drawn `lookahead` batches ahead)
        self.lookahead = max(1, int(lookahead))
        self.full_order = {k: list(v) for k, v in self.split_ix.items()}       # the whole pass, identical on every rank
        self.rng.shuffle(self.full_order['train'])             # MySampler shuffles the train split (dataloader.py:394-397)
        #                                                        (its permutation comes from numpy's global RNG: the ORDER of a
        #                                                        pass is not reproducible across the two loaders, its contents are)
        self.order = {k: self._mine(k) for k in self.full_order}               # this rank's part of the pass
        # what a checkpoint needs to resume INSIDE an epoch (the reference saves its sampler's index_list + iter_counter,
        # dataloader.py:376-405): a copy of the (whole) order taken once per shuffle, handed out with every batch next to the RNG
        # states
        self._snap = {k: list(v) for k, v in self.full_order.items()}
        self.pos = {'train': 0, 'val': 0, 'test': 0}
        if processes is None:
            processes = os.environ.get('CAPMI_LOADER_PROCS', '0') == '1'
        self.processes = bool(processes)
        if self.processes:        # spawn: the parent may already hold a HIP context, which a forked child must not inherit
            self.pool = ProcessPoolExecutor(max_workers=max(1, workers), mp_context=multiprocessing.get_context('spawn'))
        else:
            self.pool = ThreadPoolExecutor(max_workers=max(1, workers))
        self._pending = {}

    def _mine(self, split):
        """this rank's elements of the split's current pass: every world-th one; train passes are padded from their head to
        a multiple of `world` (DistributedSampler's rule), val / test are not (a short last batch is part of their contract)"""
        full = self.full_order[split]
        if self.world == 1:
            return list(full)
        if split == 'train' and full:
            pad = (-len(full)) % self.world
            full = full + full[:pad]
        return full[self.rank::self.world]

    def _reshuffle(self, split):
        self.rng.shuffle(self.full_order[split])
        self.order[split] = self._mine(split)
        self._snap[split] = list(self.full_order[split])

    def load_state(self, split, order=None, pos=None, rng=None, cap_rng=None):
        """resume inside an epoch (tools/train.py): the pass's order as a checkpoint stored it (`loader_order`), the position in
        this rank's part, both RNG states -- everything a later checkpoint of the same epoch hands out again"""
        if order is not None and len(order) == len(self.full_order.get(split, ())):
            self.full_order[split] = list(order)
            self.order[split] = self._mine(split)
            self._snap[split] = list(order)
        if pos is not None:
            self.pos[split] = int(pos)
        if rng is not None:
            self.rng.setstate(rng)
        if cap_rng is not None:
            self.cap_rng.setstate(cap_rng)

    # ---- reference accessors
    def get_vocab(self):
        return self.ix_to_word

    def get_vocab_size(self):
        return self.vocab_size

    def get_seq_length(self):
        return self.seq_length

    def document_frequency(self):
        """{n-gram tuple -> #train images containing it}, #images -- what scripts/prepro_ngrams.py:17-80 stores in
        data/<cached_tokens>.p (used when that pickle is not at hand)."""
        from imagecaptioning.pytorch_amd import synthetic
        refs = [self.label[self.label_start_ix[ix] - 1: self.label_end_ix[ix]] for ix in self.split_ix['train']]
        return synthetic.document_frequency(refs)

    # ---- one image
    def _image(self, ix):
        return decode_image(self.att_dir, self.fc_dir, self.info['images'][ix]['id'], self.use_fc, self.norm_att_feat)

    def submit_image(self, ix):
        """future of (fc, att) of image `ix` on the decode pool (threads, or worker processes: a pure function of paths)"""
        if self.processes:
            return self.pool.submit(decode_image, self.att_dir, self.fc_dir, self.info['images'][ix]['id'], self.use_fc,
                                    self.norm_att_feat)
        return self.pool.submit(self._image, ix)

    def _captions(self, ix, rng):
        """dataloader.py:165-184: seq_per_img consecutive captions from a random start, or sampling with replacement"""
        ix1, ix2 = int(self.label_start_ix[ix]) - 1, int(self.label_end_ix[ix]) - 1
        ncap = ix2 - ix1 + 1
        assert ncap > 0, 'an image does not have any label'
        n = self.seq_per_img
        if ncap < n:
            return np.stack([self.label[rng.randint(ix1, ix2)] for _ in range(n)]).astype(np.int64)
        s = rng.randint(ix1, ix2 - n + 1)
        return self.label[s:s + n].astype(np.int64)

    # ---- batches
    def _next_indices(self, split, B):
        """The next batch's image indices with MySampler's semantics (dataloader.py:376-391, pinned by
        tests/golden/loader_ref.npz = batches of the reference loader itself).  train (shuffle, wrap): when the pass is used up
        the order is reshuffled and the FIRST element of the new pass carries ``wrapped``, so a batch may straddle two passes.
        val / test (no shuffle, no wrap): the sampler stops at the end of the split -- torch's DataLoader then yields the
        partial last batch (drop_last=False) and the next get_batch starts the split over (:349-354); never ``wrapped``."""
        order, out, wrapped = self.order[split], [], False
        if not order:
            raise EmptySplit('split %r has no images' % split)
        for _ in range(B):
            if self.pos[split] >= len(order):
                if split == 'train':
                    self._reshuffle(split)
                    order = self.order[split]
                    wrapped = True
                elif out:
                    break
                self.pos[split] = 0
            out.append(order[self.pos[split]])
            self.pos[split] += 1
        return out, wrapped

    def reset_iterator(self, split):
        """DataLoader.reset_iterator (dataloader.py:356-358): the split starts over (batches already scheduled ahead are dropped);
        the train split is reshuffled, as MySampler._reset_iter does."""
        for key in [k for k in self._pending if k[0] == split]:
            for item in self._pending.pop(key):
                for f in (item[3].values() if isinstance(item[3], dict) else item[3]):
                    f.cancel()
        if split == 'train':
            self._reshuffle(split)
        self.pos[split] = 0

    def _schedule(self, split, B):
        idx, wrapped = self._next_indices(split, B)
        return idx, wrapped, self.pos[split], [self.submit_image(ix) for ix in idx], self._snap[split], self.rng.getstate()

    def label_part(self, idx):
        """labels [B,n,L+2] int64, masks, gts, infos of the images `idx` (dataloader.py:165-184, 243-259)"""
        B, n, L = len(idx), self.seq_per_img, self.seq_length
        labels = np.zeros((B, n, L + 2), dtype=np.int64)
        masks = np.zeros((B, n, L + 2), dtype=np.float32)
        gts, infos = [], []
        for b, ix in enumerate(idx):
            seq = self._captions(ix, self.cap_rng)
            labels[b, :, 1:L + 1] = seq
            for j in range(n):
                masks[b, j, :int((seq[j] != 0).sum()) + 2] = 1
            gts.append(self.label[self.label_start_ix[ix] - 1: self.label_end_ix[ix]])
            im = self.info['images'][ix]
            infos.append({'ix': ix, 'id': im['id'], 'file_path': im.get('file_path', '')})
        return labels, masks, gts, infos

    def get_batch(self, split, batch_size=None):
        B = batch_size or self.batch_size
        key = (split, B)
        q = self._pending.setdefault(key, [])
        while len(q) < self.lookahead + 1:                     # decode the NEXT batches' features in the background
            q.append(self._schedule(split, B))
        idx, wrapped, pos_now, futs, snap, rng_state = q.pop(0)
        B = len(idx)                                           # (the last batch of a val / test pass may be short)
        feats = [f.result() for f in futs]
        F = feats[0][1].shape[1]
        kmax = max(a.shape[0] for _, a in feats)
        fc = np.stack([f for f, _ in feats]).astype(np.float32)
        att = np.zeros((B, kmax, F), dtype=np.float32)
        att_masks = np.zeros((B, kmax), dtype=np.float32)
        for b, (_, a) in enumerate(feats):
            att[b, :a.shape[0]] = a
            att_masks[b, :a.shape[0]] = 1
        labels, masks, gts, infos = self.label_part(idx)
        state = {'loader_order': {split: snap}, 'loader_rng': rng_state, 'loader_cap_rng': self.cap_rng.getstate()}
        am = None if att_masks.sum() == att_masks.size else torch.from_numpy(att_masks)      # :240-241
        if am is not None:
            am._capmi_kmax = kmax         # clip_att's K, known here on the host: the step never syncs for it (ops.clip_len)
        return {'fc_feats': torch.from_numpy(fc), 'att_feats': torch.from_numpy(att), 'att_masks': am,
                'labels': torch.from_numpy(labels), 'masks': torch.from_numpy(masks), 'gts': gts,
                'bounds': {'it_pos_now': pos_now, 'it_max': len(self.order[split]), 'wrapped': wrapped,
                           'loader_state': state}, 'infos': infos}
