"""The real feature / label loader (captioning/data/feature_loader.py) against the reference's batch contract
(captioning/data/dataloader.py:182-299): built on a tiny on-disk dataset with a VARIABLE number of regions per image."""
import json
import os
import sys

import numpy as np
import pytest
import torch

from conftest import ROOT, GOLDEN

PKG = os.path.join(ROOT, 'imagecaptioning', 'pytorch_amd')


def make_dataset(tmp, n_img=9, F=24, L=7, vocab=30, fixed_regions=None, seed=0):
    rng = np.random.default_rng(seed)
    att_dir, fc_dir = tmp / 'att', tmp / 'fc'
    att_dir.mkdir(), fc_dir.mkdir()
    images, labels, start, end = [], [], [], []
    for i in range(n_img):
        img_id = 1000 + i
        K = fixed_regions or int(rng.integers(3, 9))
        feat = np.clip(rng.standard_normal((K, F)), 0, None).astype(np.float32)
        np.savez_compressed(att_dir / ('%d.npz' % img_id), feat=feat)
        if i % 2 == 0:                                           # odd images have no fc file: mean of the regions (dataloader.py:295-298)
            np.save(fc_dir / ('%d.npy' % img_id), feat.mean(0) + 1.0)
        ncap = int(rng.integers(2, 7))                           # some images have fewer captions than seq_per_img
        start.append(len(labels) + 1)
        for _ in range(ncap):
            ln = int(rng.integers(2, L + 1))
            row = np.zeros(L, dtype=np.uint32)
            row[:ln] = rng.integers(1, vocab + 1, size=ln)
            labels.append(row)
        end.append(len(labels))
        images.append({'id': img_id, 'split': 'train' if i < n_img - 3 else ('val' if i < n_img - 1 else 'restval'),
                       'file_path': 'img/%d.jpg' % img_id})
    info = {'images': images, 'ix_to_word': {str(i): 'w%d' % i for i in range(1, vocab + 1)}}
    (tmp / 'data.json').write_text(json.dumps(info))
    np.savez(tmp / 'labels.npz', labels=np.stack(labels), label_start_ix=np.array(start, dtype=np.uint32),
             label_end_ix=np.array(end, dtype=np.uint32), label_length=np.array([(r > 0).sum() for r in labels], dtype=np.uint32))
    return ['--input_json', str(tmp / 'data.json'), '--input_label_h5', str(tmp / 'labels.npz'), '--input_att_dir', str(att_dir),
            '--input_fc_dir', str(fc_dir), '--fc_feat_size', str(F), '--att_feat_size', str(F)]


def _opts(argv):
    sys.path.insert(0, PKG)
    from captioning.utils import opts
    return opts.parse_opt(argv)


def test_batch_contract_variable_regions(tmp_path):
    sys.path.insert(0, PKG)
    from captioning.data.feature_loader import FeatureLoader
    args = make_dataset(tmp_path)
    ld = FeatureLoader(_opts(args + ['--batch_size', '4', '--seq_per_img', '3']))
    assert ld.vocab_size == 30 and ld.seq_length == 7
    assert len(ld.split_ix['train']) == 7 and len(ld.split_ix['val']) == 2      # restval joins train (train_only 0)
    seen, wraps = [], 0
    for _ in range(5):
        d = ld.get_batch('train')
        B, n, L = 4, 3, 7
        K = d['att_feats'].shape[1]
        assert d['fc_feats'].shape == (B, 24) and d['att_feats'].shape == (B, K, 24) and d['att_feats'].dtype == torch.float32
        assert d['labels'].shape == (B, n, L + 2) and d['labels'].dtype == torch.int64 and d['masks'].shape == (B, n, L + 2)
        assert bool((d['labels'][..., 0] == 0).all()) and bool((d['labels'][..., -1] == 0).all())
        for b in range(B):
            ix = d['infos'][b]['ix']
            feat = np.load(tmp_path / 'att' / ('%d.npz' % d['infos'][b]['id']))['feat']
            k = feat.shape[0]
            np.testing.assert_array_equal(d['att_feats'][b, :k].numpy(), feat)
            assert float(d['att_feats'][b, k:].abs().sum()) == 0
            if d['att_masks'] is not None:
                assert d['att_masks'][b].tolist() == [1.0] * k + [0.0] * (K - k)
                # clip_att's K rides on the mask from the host side, so the step never syncs for it (ops.clip_len)
                assert d['att_masks']._capmi_kmax == int(d['att_masks'].sum(1).max())
            else:
                assert k == K
            want_fc = feat.mean(0) + (1.0 if (d['infos'][b]['id'] - 1000) % 2 == 0 else 0.0)
            np.testing.assert_allclose(d['fc_feats'][b].numpy(), want_fc, rtol=1e-6)
            refs = ld.label[ld.label_start_ix[ix] - 1: ld.label_end_ix[ix]]
            np.testing.assert_array_equal(d['gts'][b], refs)
            for j in range(n):
                row = d['labels'][b, j, 1:L + 1].numpy()
                assert any((row == r).all() for r in refs)                          # every label row is one of the image's captions
                nz = int((row != 0).sum())
                assert d['masks'][b, j].tolist() == [1.0] * (nz + 2) + [0.0] * (L - nz)   # dataloader.py:245-249
        seen += [i['ix'] for i in d['infos']]
        wraps += int(d['bounds']['wrapped'])
        assert d['bounds']['it_max'] == 7
    assert wraps == 2 and set(seen) == set(ld.split_ix['train'])                  # 20 draws over 7 images: two epoch wraps
    v = ld.get_batch('val', batch_size=2)
    # (a val / test pass never sets `wrapped`: MySampler(wrap=False), pinned by test_val_and_test_batches_are_the_reference_loaders)
    assert [i['ix'] for i in v['infos']] == ld.split_ix['val'] and not v['bounds']['wrapped']


def test_fixed_region_count_gives_no_att_masks_and_df_table(tmp_path):
    sys.path.insert(0, PKG)
    from captioning.data.feature_loader import FeatureLoader
    args = make_dataset(tmp_path, fixed_regions=5)
    ld = FeatureLoader(_opts(args + ['--batch_size', '3', '--seq_per_img', '2']))
    d = ld.get_batch('train')
    assert d['att_masks'] is None and d['att_feats'].shape[1] == 5               # dataloader.py:240-241
    df, ref_len = ld.document_frequency()
    assert ref_len == len(ld.split_ix['train']) and all(1 <= c <= ref_len for c in df.values())
    assert (0,) in df                                                           # the terminating 0 counts as a token (rewards.py:33-39)


def test_h5_labels_need_h5py_or_the_converter(tmp_path):
    sys.path.insert(0, PKG)
    from captioning.data import feature_loader as FL
    (tmp_path / 'x.h5').write_bytes(b'not really hdf5')
    try:
        import h5py  # noqa: F401
    except ImportError:
        with pytest.raises(RuntimeError, match='convert_labels'):
            FL.load_labels(str(tmp_path / 'x.h5'))


@pytest.mark.parametrize('name', ['labels_small', 'labels_many'])
def test_h5lite_reads_label_files_written_by_h5py(name):
    """the label file of scripts/prepro_labels.py:158-163 (h5py.File(path, 'w') + create_dataset(dtype='uint32',
    data=...)) is read WITHOUT h5py.  The fixtures were written by genuine h5py 3.3 / HDF5 1.10.6 (tests/golden/make_h5.py,
    build container only); expected arrays travel beside them."""
    sys.path.insert(0, PKG)
    from captioning.data import h5lite, feature_loader as FL
    want = np.load(os.path.join(GOLDEN, name + '_expected.npz'))
    f = h5lite.H5File(os.path.join(GOLDEN, name + '.h5'))
    assert f.keys() == sorted(want.files)
    for k in want.files:
        got = f[k]
        assert got.dtype == np.uint32 and got.shape == want[k].shape and np.array_equal(got, want[k]), k
    got = FL.load_labels(os.path.join(GOLDEN, name + '.h5'))
    assert set(got) == {'labels', 'label_start_ix', 'label_end_ix'} and np.array_equal(got['labels'], want['labels'])


def test_h5lite_refuses_what_it_does_not_read():
    sys.path.insert(0, PKG)
    from captioning.data import h5lite
    with pytest.raises(h5lite.H5LiteError, match='compression|chunked'):
        h5lite.read_datasets(os.path.join(GOLDEN, 'labels_chunked.h5'))


def test_convert_labels_runs_without_h5py(tmp_path):
    sys.path.insert(0, PKG)
    from imagecaptioning.pytorch_amd.tools import convert_labels
    out = convert_labels.convert(os.path.join(GOLDEN, 'labels_small.h5'), str(tmp_path / 'l.npz'))
    z, want = np.load(out), np.load(os.path.join(GOLDEN, 'labels_small_expected.npz'))
    assert sorted(z.files) == sorted(want.files) and all(np.array_equal(z[k], want[k]) for k in want.files)


@pytest.mark.gpu
def test_train_on_real_feature_files_xe_then_scst(tmp_path):
    """tools/train.py end to end on on-disk features with 3..8 regions per image (att_masks path), through the prefetcher"""
    sys.path.insert(0, PKG)
    from imagecaptioning.pytorch_amd.tools import train as T
    from captioning.utils import rewards
    data = tmp_path / 'data'
    data.mkdir()
    args = make_dataset(data, n_img=12)
    small = args + ['--caption_model', 'updown', '--rnn_size', '32', '--input_encoding_size', '32', '--att_hid_size', '16',
                    '--batch_size', '4', '--seq_per_img', '3', '--losses_log_every', '5', '--checkpoint_path', str(tmp_path / 'ck'),
                    '--learning_rate', '0.01']
    l0 = T.train(_opts(small + ['--max_iters', '1']))
    l1 = T.train(_opts(small + ['--max_iters', '30', '--save_checkpoint_every', '30']))
    assert l1 < l0
    rewards.reset_scorer()
    T.train(_opts(small + ['--max_iters', '33', '--self_critical_after', '0', '--train_sample_n', '3', '--start_from', str(tmp_path / 'ck'),
                           '--cached_tokens', 'no-such-table']))
    rewards.reset_scorer()


@pytest.mark.parametrize('fixed,budget_rows', [(None, None), (6, None), (None, 14)])
def test_resident_feature_store_returns_the_streaming_loaders_batches(tmp_path, fixed, budget_rows):
    """captioning/data/resident.py: after an image's first draw its features live in the device store and later batches are a
    row gather -- three epochs of batches (variable and fixed region counts, partial batches over the wrap, and a budget that
    only holds some of the images) equal the streaming FeatureLoader's, tensor for tensor.  Runs on the CPU device here; the GPU
    suite runs the same store under DevicePrefetcher."""
    args = make_dataset(tmp_path, fixed_regions=fixed)
    from captioning.data.feature_loader import FeatureLoader
    from captioning.data.resident import ResidentFeatures
    a = FeatureLoader(_opts(args + ['--batch_size', '4', '--seq_per_img', '3']))
    b = ResidentFeatures(FeatureLoader(_opts(args + ['--batch_size', '4', '--seq_per_img', '3'])), 'cpu',
                         budget_bytes=None if budget_rows is None else budget_rows * 24 * 4, first_rows=8)
    n_train = len(a.split_ix['train'])
    for it in range(3 * ((n_train + 3) // 4) + 1):
        x, y = a.get_batch('train'), b.get_batch('train')
        for k in ('fc_feats', 'att_feats', 'labels', 'masks'):
            assert torch.equal(x[k], y[k].cpu()), (it, k)
        assert (x['att_masks'] is None) == (y['att_masks'] is None)
        if x['att_masks'] is not None:
            assert torch.equal(x['att_masks'], y['att_masks'])
            assert x['att_masks']._capmi_kmax == y['att_masks']._capmi_kmax == int(y['att_masks'].sum(1).max())
        assert all(np.array_equal(g, h) for g, h in zip(x['gts'], y['gts']))
        assert x['bounds'] == y['bounds'] and x['infos'] == y['infos']
    if budget_rows is None:
        assert b.misses == n_train and b.streamed == 0 and b.hits > 2 * n_train - 4      # every image decoded exactly once
    else:
        assert b.streamed > 0 and b.used <= budget_rows                                  # the rest is streamed every time
    assert b.get_vocab() == a.get_vocab() and b.pos is b.loader.pos                      # the wrapped loader's interface


@pytest.mark.gpu
def test_resident_store_under_the_device_prefetcher_gpu(tmp_path):
    """The HBM-resident store behind DevicePrefetcher (its gathers and inserts run on the prefetcher's copy stream) delivers the
    same device batches as the streaming loader behind DevicePrefetcher, over three epochs with variable region counts."""
    args = make_dataset(tmp_path, n_img=23, F=40)
    from captioning.data.feature_loader import FeatureLoader
    from captioning.data.resident import ResidentFeatures
    from captioning.data.prefetch import DevicePrefetcher
    dev = torch.device('cuda:0')
    a = DevicePrefetcher(FeatureLoader(_opts(args + ['--batch_size', '5', '--seq_per_img', '3'])), dev)
    res = ResidentFeatures(FeatureLoader(_opts(args + ['--batch_size', '5', '--seq_per_img', '3'])), dev, first_rows=16)
    b = DevicePrefetcher(res, dev)
    for it in range(14):
        x, y = a.get_batch('train'), b.get_batch('train')
        for k in ('fc_feats', 'att_feats', 'labels', 'masks', 'att_masks'):
            assert (x[k] is None) == (y[k] is None), (it, k)
            if x[k] is not None:
                assert y[k].is_cuda and torch.equal(x[k], y[k]), (it, k)
        assert x['bounds'] == y['bounds']
    assert res.streamed == 0 and res.misses == len(res.split_ix['train']) and res.hits > 0


# ---- pinned to the REFERENCE loader: tests/golden/loader_ref.npz holds batches that /root/reference/captioning/data/dataloader.py
# itself produced on tests/golden/loader_ds (tests/golden/make_loader_golden.py)
def _ref_loader(B):
    import argparse
    from imagecaptioning.pytorch_amd.captioning.data.feature_loader import FeatureLoader
    ds = os.path.join(GOLDEN, 'loader_ds')
    opt = argparse.Namespace(batch_size=B, seq_per_img=2, use_fc=True, norm_att_feat=0, train_only=0, seed=3,
                             input_json=os.path.join(ds, 'dataset.json'), input_label_h5=os.path.join(ds, 'labels.npz'),
                             input_fc_dir=os.path.join(ds, 'fc'), input_att_dir=os.path.join(ds, 'att'))
    return FeatureLoader(opt, workers=2, processes=False)


def _same_batch(z, pre, d, exact_rows=True):
    assert [d['bounds'][k] for k in ('it_pos_now', 'it_max')] == z[pre + 'bounds'][:2].tolist(), pre
    assert bool(d['bounds']['wrapped']) == bool(z[pre + 'bounds'][2]), pre
    assert d['fc_feats'].shape[0] == len(d['infos']) == len(d['gts']) == z[pre + 'ix'].shape[0], pre
    assert (d['att_masks'] is not None) == bool(z[pre + 'has_att_masks']) or not exact_rows, pre
    if not exact_rows:
        return
    assert [i['ix'] for i in d['infos']] == z[pre + 'ix'].tolist()
    assert [i['id'] for i in d['infos']] == z[pre + 'id'].tolist()
    assert [i['file_path'] for i in d['infos']] == z[pre + 'file_path'].tolist()
    for k in ('fc_feats', 'att_feats', 'labels', 'masks'):
        want = z[pre + k]
        assert d[k].numpy().dtype == want.dtype and d[k].shape == want.shape, (pre, k)
        assert np.array_equal(d[k].numpy(), want), (pre, k)
    if d['att_masks'] is not None:
        assert np.array_equal(d['att_masks'].numpy(), z[pre + 'att_masks'])
    for i, g in enumerate(d['gts']):
        assert np.array_equal(np.asarray(g), z[pre + 'gts%d' % i]) and np.asarray(g).dtype == z[pre + 'gts%d' % i].dtype


@pytest.mark.parametrize('B', [2, 4])
def test_val_and_test_batches_are_the_reference_loaders(B):
    """deterministic splits: every field of every batch, the partial last batch of a pass, the restart of the split, reset_iterator"""
    z = np.load(os.path.join(GOLDEN, 'loader_ref.npz'))
    ld = _ref_loader(B)
    tag = 'b%d.' % B
    assert ld.vocab_size == int(z[tag + 'vocab_size']) and ld.seq_length == int(z[tag + 'seq_length'])
    for split, calls in (('val', 5), ('test', 3)):
        for c in range(calls):
            _same_batch(z, '%s%s%d.' % (tag, split, c), ld.get_batch(split))
    ld.get_batch('val')
    ld.reset_iterator('val')
    _same_batch(z, tag + 'val_after_reset.', ld.get_batch('val'))


@pytest.mark.parametrize('B', [2, 4])
def test_train_batches_follow_the_reference_samplers_bookkeeping(B):
    """shuffled split: the order of a pass comes from another RNG, but `bounds` of every batch (position in the pass, the
    `wrapped` flag on the batch that holds the first image of a new pass), the batch sizes and the contents of a pass must be the
    reference's; every row must be the image it claims to be, with captions drawn the reference's way (dataloader.py:165-184)"""
    z = np.load(os.path.join(GOLDEN, 'loader_ref.npz'))
    ld = _ref_loader(B)
    tag = 'b%d.' % B
    lab = np.load(os.path.join(GOLDEN, 'loader_ds', 'labels.npz'))
    seen, passes = [], []
    for c in range(8):
        d = ld.get_batch('train')
        pre = '%strain%d.' % (tag, c)
        _same_batch(z, pre, d, exact_rows=False)
        for b, info in enumerate(d['infos']):
            ix = info['ix']
            if d['bounds']['wrapped'] and len(seen) == 6:
                passes.append(seen)
                seen = []
            seen.append(ix)
            s, e = int(lab['label_start_ix'][ix]) - 1, int(lab['label_end_ix'][ix])
            own = lab['labels'][s:e].astype(np.int64)
            rows = d['labels'][b, :, 1:-1].numpy()
            assert (d['labels'][b, :, 0] == 0).all() and (d['labels'][b, :, -1] == 0).all()
            if e - s >= 2:                                     # a window of seq_per_img consecutive captions
                assert any(np.array_equal(rows, own[j:j + 2]) for j in range(e - s - 1)), (c, b)
            else:                                              # fewer captions than seq_per_img: drawn with replacement
                assert all(any(np.array_equal(r, o) for o in own) for r in rows)
            assert np.array_equal(np.asarray(d['gts'][b]), lab['labels'][s:e])
    assert passes and all(sorted(p) == [0, 2, 3, 5, 7, 10] for p in passes)       # 5 train + 1 restval image, each once per pass
    # the reference's own passes have the same contents
    ref_seq = np.concatenate([z['%strain%d.ix' % (tag, c)] for c in range(8)])
    assert sorted(ref_seq[:6].tolist()) == [0, 2, 3, 5, 7, 10]


@pytest.mark.gpu
@pytest.mark.parametrize('resident', [False, True])
def test_reference_val_batches_through_the_device_wrappers_gpu(resident):
    """the same golden val / test batches behind ResidentFeatures and DevicePrefetcher: short last batch, restart of the split,
    reset_iterator with batches already in flight (the wrappers drop what they scheduled ahead)"""
    from imagecaptioning.pytorch_amd.captioning.data.resident import ResidentFeatures
    from imagecaptioning.pytorch_amd.captioning.data.prefetch import DevicePrefetcher
    z = np.load(os.path.join(GOLDEN, 'loader_ref.npz'))
    dev = torch.device('cuda:0')
    base = _ref_loader(2)
    ld = DevicePrefetcher(ResidentFeatures(base, dev, first_rows=8) if resident else base, dev)

    def host(d):
        out = dict(d)
        for k in ('fc_feats', 'att_feats', 'labels', 'masks', 'att_masks'):
            if out[k] is not None:
                assert out[k].is_cuda or k in ('labels', 'masks')
                out[k] = out[k].cpu()
        out['gts'] = list(getattr(out['gts'], 'gts', out['gts']))
        return out
    for split, calls in (('val', 5), ('test', 3)):
        for c in range(calls):
            _same_batch(z, 'b2.%s%d.' % (split, c), host(ld.get_batch(split)))
    ld.get_batch('val')
    ld.reset_iterator('val')
    _same_batch(z, 'b2.val_after_reset.', host(ld.get_batch('val')))


@pytest.mark.parametrize('world', [1, 2])
def test_resume_twice_inside_an_epoch_keeps_the_epochs_sequence(world):
    """a checkpoint written AFTER a resume and before the next epoch wrap must carry the same permutation as
    the first one -- tools/train.py restores the loader through FeatureLoader.load_state (order, position, both RNG states, and
    the snapshot later checkpoints hand out).  save -> resume -> save -> resume: the index sequence equals the uninterrupted run's,
    also across the following reshuffle, for one rank and for a rank of a 2-way partition."""
    def mk():
        ld = _ref_loader(2)
        return type(ld)(ld.opt, workers=1, processes=False, lookahead=2, rank=world - 1, world=world)

    def take(ld, k):
        out, states = [], []
        for _ in range(k):
            d = ld.get_batch('train')
            out.append([i['ix'] for i in d['infos']])
            states.append((d['bounds']['it_pos_now'], d['bounds']['loader_state']))
        return out, states

    want, _ = take(mk(), 12)                       # 12 batches of 2: several passes of the small train split

    def resume(state):
        pos, st = state
        ld = mk()
        ld.load_state('train', order=st['loader_order']['train'], pos=pos, rng=st['loader_rng'], cap_rng=st['loader_cap_rng'])
        return ld
    a, sa = take(mk(), 3)
    ld = resume(sa[-1])                            # first resume (the "checkpoint" = the last CONSUMED batch's state)
    b, sb = take(ld, 1)                            # one more batch inside the same epoch ...
    ld = resume(sb[-1])                            # ... checkpoint again, resume again
    c, _ = take(ld, 8)
    assert a + b + c == want
