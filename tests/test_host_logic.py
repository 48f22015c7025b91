"""Host-side training-loop logic mirrored from the reference's tools/train.py / captioning/utils/misc.py (no GPU)."""
import argparse
import os
import sys

import pytest
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'imagecaptioning', 'pytorch_amd'))


def _misc():
    from imagecaptioning.pytorch_amd.captioning.utils import misc
    return misc


def test_epoch_decay_matches_train_py_formula():
    misc = _misc()
    opt = argparse.Namespace(learning_rate=5e-4, learning_rate_decay_start=0, learning_rate_decay_every=3,
                             learning_rate_decay_rate=0.8)
    s = misc.LRSchedule(opt)
    for epoch in range(0, 12):
        want = 5e-4 * 0.8 ** ((epoch - 0) // 3) if epoch > 0 else 5e-4        # tools/train.py:134-141
        assert s.epoch_start(epoch) == pytest.approx(want, rel=1e-12)
        assert s.rate(epoch * 100) == pytest.approx(want, rel=1e-12)


def test_noam_rate_matches_noamopt():
    """misc.py:159-185: NoamOpt.step() increments _step, then lr = factor * d^-0.5 * min(step^-0.5, step * warmup^-1.5)."""
    misc = _misc()
    opt = argparse.Namespace(learning_rate=1.0, noamopt=1, noamopt_factor=1.0, noamopt_warmup=20000, d_model=512)
    s = misc.LRSchedule(opt)
    for it in (0, 1, 99, 19999, 20000, 123456):
        step = it + 1
        want = 1.0 * (512 ** -0.5) * min(step ** -0.5, step * 20000 ** -1.5)
        assert s.rate(it) == pytest.approx(want, rel=1e-12)
    assert s.rate(19999) == pytest.approx(max(s.rate(i) for i in (0, 5000, 19999, 40000)), rel=1e-12)   # peak at warm-up end


def test_linear_warmup():
    misc = _misc()
    opt = argparse.Namespace(learning_rate=2e-4, use_warmup=1, noamopt_warmup=10)
    s = misc.LRSchedule(opt)
    s.epoch_start(0)
    rates = [s.rate(i) for i in range(12)]                # tools/train.py:171-173, evaluated every iteration
    assert rates[:3] == pytest.approx([2e-5, 4e-5, 6e-5])
    assert rates[9] == pytest.approx(2e-4) and rates[11] == pytest.approx(2e-4)


def test_reduce_on_plateau_matches_torch_scheduler():
    misc = _misc()
    opt = argparse.Namespace(learning_rate=1e-3, reduce_on_plateau=1, reduce_on_plateau_factor=0.5, reduce_on_plateau_patience=2)
    s = misc.LRSchedule(opt)
    p = torch.nn.Parameter(torch.zeros(1))
    ref_opt = torch.optim.Adam([p], lr=1e-3)
    ref = torch.optim.lr_scheduler.ReduceLROnPlateau(ref_opt, mode='min', factor=0.5, patience=2, threshold=1e-4,
                                                     threshold_mode='rel', cooldown=0, min_lr=0, eps=1e-8)
    vals = [3.0, 2.5, 2.6, 2.55, 2.7, 2.49, 2.5, 2.5, 2.5, 2.5, 2.4999, 2.6, 2.6, 2.6]
    for v in vals:
        ref.step(v)
        assert s.plateau_step(v) == pytest.approx(ref_opt.param_groups[0]['lr'], rel=1e-12), v


def test_scheduled_sampling_probability():
    misc = _misc()
    opt = argparse.Namespace(scheduled_sampling_start=0, scheduled_sampling_increase_every=5,
                             scheduled_sampling_increase_prob=0.05, scheduled_sampling_max_prob=0.25)
    assert misc.scheduled_sampling_prob(opt, 0) == 0.0
    assert misc.scheduled_sampling_prob(opt, 11) == pytest.approx(0.10)
    assert misc.scheduled_sampling_prob(opt, 500) == pytest.approx(0.25)
    assert misc.scheduled_sampling_prob(argparse.Namespace(), 500) == 0.0


def test_df_image_roundtrip(tmp_path):
    """pickle (scripts/prepro_ngrams.py format) -> flat hash image -> same table as hashing the dict directly."""
    import pickle
    import numpy as np
    pytest.importorskip('ctypes')
    try:
        from imagecaptioning.pytorch_amd import ciderd
    except OSError:
        pytest.skip('libcapmi.so not built')
    from imagecaptioning.pytorch_amd.tools import convert_df
    rng = np.random.default_rng(0)
    df = {}
    for _ in range(500):
        n = int(rng.integers(1, 5))
        df[tuple(str(int(t)) for t in rng.integers(1, 200, size=n))] = float(rng.integers(1, 50))
    src = tmp_path / 'toy-idxs.p'
    with open(src, 'wb') as f:
        pickle.dump({'document_frequency': df, 'ref_len': 123}, f)
    dst = convert_df.convert(str(src))
    assert dst.endswith('toy-idxs.capmi.npz')
    keys, vals, ref_len = ciderd.load_df_image(dst)
    k2, v2 = ciderd.build_table(df)
    assert ref_len == 123.0 and np.array_equal(keys, k2) and np.array_equal(vals, v2)
    # every n-gram is found by linear probing from its home slot
    cap = keys.shape[0]
    for g, c in list(df.items())[:50]:
        key = np.uint64(ciderd.pack_ngram([int(t) for t in g]))
        s = int(ciderd._mix64(np.array([key], dtype=np.uint64))[0] & np.uint64(cap - 1))
        while keys[s] != key:
            assert keys[s] != 0
            s = (s + 1) & (cap - 1)
        assert vals[s] == c


def test_decode_option_flags_and_eval_kwargs():
    """Host-side plumbing of the decode options: which requests leave the one-call rollouts (decode.wants_options), which
    constraint bits apply at which step (AttModel.py:293, 298, 307: t > 0, t > 0, t >= 3), how finished beams are padded for
    decode_sequence, and that tools/eval.py hands every sampler flag of the command line to the model (eval_utils.py:169-171)."""
    from imagecaptioning.pytorch_amd import decode, _lib
    from imagecaptioning.pytorch_amd.tools import eval as E
    assert not decode.wants_options({'sample_method': 'greedy', 'beam_size': 5})
    for k in ('decoding_constraint', 'block_trigrams', 'remove_bad_endings'):
        assert decode.wants_options({k: 1})
    assert decode.wants_options({'group_size': 2}) and not decode.wants_options({'group_size': 1})
    o = {'decoding_constraint': 1, 'remove_bad_endings': 1, 'block_trigrams': 1}
    assert decode._flags(o, 0) == 0
    assert decode._flags(o, 1) == decode._flags(o, 2) == _lib.DECODE_NO_REPEAT | _lib.DECODE_NO_BAD_ENDING
    assert decode._flags(o, 3) == _lib.DECODE_NO_REPEAT | _lib.DECODE_NO_BAD_ENDING | _lib.DECODE_BLOCK_TRIGRAMS
    assert decode._flags({'block_trigrams': 1}, 2) == 0
    st = E._pad_stack([torch.tensor([4, 5, 0]), torch.tensor([7, 0]), torch.tensor([1, 2, 3, 9])])
    assert st.tolist() == [[4, 5, 0, 0], [7, 0, 0, 0], [1, 2, 3, 9]]
    opt = argparse.Namespace(sample_method='top5', beam_size=1, temperature=0.7, suppress_UNK=1, length_penalty='wu_0.5', group_size=2,
                             diversity_lambda=0.3, decoding_constraint=1, block_trigrams=1, remove_bad_endings=0, max_length=16,
                             unrelated=123)
    kw = E.eval_kwargs_of(opt)
    assert kw == {'sample_method': 'top5', 'beam_size': 1, 'temperature': 0.7, 'suppress_UNK': 1, 'length_penalty': 'wu_0.5',
                  'group_size': 2, 'diversity_lambda': 0.3, 'decoding_constraint': 1, 'block_trigrams': 1, 'remove_bad_endings': 0,
                  'max_length': 16}


def test_parse_sample_method_and_bad_endings():
    """CaptionModel.sample_next_word's method strings (CaptionModel.py:370-407) and the bad-ending vocabulary (AttModel.py:27,96)."""
    from imagecaptioning.pytorch_amd.captioning.models.utils import parse_sample_method
    from imagecaptioning.pytorch_amd.captioning.models.CaptionModel import CaptionModel
    assert parse_sample_method('greedy', 0.5) == ('greedy', 0.5, 0, 0.0)
    assert parse_sample_method('sample', 1.3) == ('sample', 1.3, 0, 0.0)
    assert parse_sample_method('gumbel', 1.3) == ('sample', 1.0, 0, 0.0)          # the temperature cancels in the arg-max
    assert parse_sample_method('top5', 2.0) == ('sample', 2.0, 5, 0.0)
    assert parse_sample_method('top0.9', 1.0) == ('sample', 1.0, 0, 0.9)
    with pytest.raises(NotImplementedError):
        parse_sample_method('beam_search_xyz', 1.0)
    m = CaptionModel()
    m.vocab = {'1': 'a', '2': 'dog', '3': 'with', '4': 'the', '5': 'UNK'}
    assert sorted(m.bad_endings_ix) == [1, 3, 4]
    m.bad_endings_ix = [2]
    assert m.bad_endings_ix == [2]


def test_misc_helpers_match_the_reference_module(monkeypatch):
    """decode_sequence (incl. REMOVE_BAD_ENDINGS and BPE joins), the beam-search length penalties and the Noam rate against
    outputs of the reference's captioning/utils/misc.py (tests/golden/misc_helpers.npz, ``make_golden.py misc``)."""
    import numpy as np
    from imagecaptioning.pytorch_amd import beam
    misc = _misc()
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'misc_helpers.npz'))
    ix_to_word = {str(i + 1): str(w) for i, w in enumerate(z['words'])}
    seq = torch.from_numpy(z['seq'])
    for env in ('0', '1'):
        monkeypatch.setenv('REMOVE_BAD_ENDINGS', env)
        assert misc.decode_sequence(ix_to_word, seq) == [str(s) for s in z['decoded_' + env]]
    monkeypatch.delenv('REMOVE_BAD_ENDINGS')
    assert misc.decode_sequence(ix_to_word, seq) == [str(s) for s in z['decoded_0']]
    for cfg in ('', 'wu_0.7', 'wu_0.0', 'avg_0.0'):
        f = beam._penalty(cfg)
        got = [f(float(l), float(y)) for l, y in zip(z['pen_lengths'], z['pen_logps'])]
        assert np.allclose(got, z['penalty_' + cfg], rtol=1e-12)
    for d_model, factor, warmup in ((512, 1.0, 2000), (1024, 2.0, 10000)):
        sched = misc.LRSchedule(argparse.Namespace(noamopt=1, noamopt_factor=factor, noamopt_warmup=warmup, learning_rate=0.0),
                                model_size=d_model)
        got = [sched.noam_rate(int(st)) for st in z['noam_steps_%d' % warmup]]
        assert np.allclose(got, z['noam_%d_%g_%d' % (d_model, factor, warmup)], rtol=1e-12)


def test_option_defaults_are_the_references():
    """Every flag the mirror's opts.py shares with the reference has the reference's default (tests/golden/opts_defaults.json,
    ``make_golden.py opts_defaults``), except the deliberate ones: the default model family, paths, logging cadence."""
    import json
    from imagecaptioning.pytorch_amd.captioning.utils import opts
    ref = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'opts_defaults.json')))
    deliberate = {'caption_model', 'batch_size', 'checkpoint_path', 'id', 'save_checkpoint_every', 'losses_log_every', 'start_from',
                  'input_json', 'input_label_h5', 'input_fc_dir', 'input_att_dir'}
    shared = [k for k in opts.DEFAULTS if k in ref]
    assert len(shared) > 60
    wrong = {k: (opts.DEFAULTS[k], ref[k]) for k in shared if k not in deliberate and opts.DEFAULTS[k] != ref[k]}
    assert not wrong, wrong


def test_flat_optimizer_state_survives_a_layout_change():
    """r4: FlatParams lays a model's _flat_groups() back to back, so the moment buffers of a Transformer / AoA checkpoint written
    before round 4 (model.parameters() order, no 'offsets' table) are permuted relative to today's buffers: load_state_dict moves
    every parameter's moments by name; a state_dict of today carries its own table and round-trips."""
    import torch
    from imagecaptioning.pytorch_amd.flat import FlatParams

    class Net(torch.nn.Module):
        def __init__(self, grouped):
            super().__init__()
            self.a, self.b, self.c, self.d = (torch.nn.Linear(4, 4) for _ in range(4))
            self.grouped = grouped

        def _flat_groups(self):
            return [['a.weight', 'c.weight', 'd.weight'], ['a.bias', 'c.bias', 'd.bias']] if self.grouped else []

    torch.manual_seed(0)
    old, new = FlatParams(Net(False)), FlatParams(Net(True))
    assert old.offsets != new.offsets and sorted(old.offsets) == sorted(new.offsets) or old.total == new.total
    for n, p, o in zip(old.names, old.params, old.offsets):              # moments that identify their parameter
        old.exp_avg[o:o + p.numel()] = float(old.names.index(n) + 1)
        old.exp_avg_sq[o:o + p.numel()] = float(old.names.index(n) + 1) * 10
    old.step_count = 7
    sd = old.state_dict()
    sd.pop('offsets')                                                    # a checkpoint of rounds 1-3
    new.load_state_dict(sd)
    for n, p, o in zip(new.names, new.params, new.offsets):
        assert float(new.exp_avg[o]) == new.names.index(n) + 1 and float(new.exp_avg_sq[o + p.numel() - 1]) == (new.names.index(n) + 1) * 10
    assert new.step_count == 7
    back = FlatParams(Net(True))
    back.load_state_dict(new.state_dict())                              # same layout: plain copy
    assert torch.equal(back.exp_avg, new.exp_avg)
    older = FlatParams(Net(False))
    older.load_state_dict(new.state_dict())                             # and the other way, through the table
    assert torch.equal(older.exp_avg, old.exp_avg)


def test_cached_parameter_walk_follows_the_module():
    """r4: the models ask for their parameters through CaptionModel._param_slots (a cached walk: nn.Module.named_parameters() cost
    the Transformer 0.5 ms per call, twice per step).  It must equal named_parameters() -- after .to(), load_state_dict, a swapped
    Parameter object, and a replaced top-level submodule."""
    import torch.nn as nn
    from imagecaptioning.pytorch_amd import synthetic
    from imagecaptioning.pytorch_amd.captioning import models
    for fam, kw in (('transformer', dict(input_encoding_size=32, rnn_size=64, d_model=32, d_ff=64, N_enc=2, N_dec=2, num_att_heads=4)),
                    ('updown', dict(input_encoding_size=24, rnn_size=24, att_hid_size=16))):
        model = models.setup(synthetic.updown_opt(caption_model=fam, **kw))

        def same():
            want = list(model.named_parameters())
            got = model._named_param_list()
            assert [n for n, _ in got] == [n for n, _ in want] == model._param_name_list()
            assert all(a is b for (_, a), (_, b) in zip(got, want)) and all(a is b for a, (_, b) in zip(model._param_list(), want))
        same()
        model.double()
        same()
        model.load_state_dict({k: v.clone() for k, v in model.state_dict().items()})
        same()
        name, p = next(iter(model.named_parameters()))
        owner = model.get_submodule(name.rpartition('.')[0])
        setattr(owner, name.rpartition('.')[2], nn.Parameter(torch.zeros_like(p)))       # a swapped object deep in the tree: found
        same()
        if fam == 'updown':
            model.logit = nn.Linear(24, 7)
        else:
            model.att_embed = nn.Sequential(nn.Linear(2048, 32), nn.ReLU(), nn.Dropout(0.1), nn.Linear(32, 32))   # one more parameter pair
        same()


def test_collect_grads_adopts_copies_and_zeroes():
    """FlatParams.collect_grads makes the flat gradient buffer authoritative: a .grad that already IS the flat view (what autograd
    adopts from the native backwards) is left alone (r4: no per-parameter re-assignment), a foreign .grad is copied in, a missing one
    becomes zero -- and afterwards every p.grad is the flat view's memory"""
    import torch
    from imagecaptioning.pytorch_amd.flat import FlatParams
    torch.manual_seed(1)
    net = torch.nn.Sequential(torch.nn.Linear(3, 4), torch.nn.Linear(4, 2))
    flat = FlatParams(net)
    names = flat.names
    flat.grad.fill_(7.0)                                   # stale content of an earlier step
    flat.zero_grad()
    p0, p1, p2, p3 = flat.params
    flat.grad_views[names[0]].fill_(1.5)
    p0.grad = flat.grad_views[names[0]]                    # adopted view (native backward wrote in place)
    p1.grad = torch.full_like(p1, 2.5)                     # a gradient torch autograd produced elsewhere
    p2.grad = None                                         # a parameter the loss did not reach
    p3.grad = flat.grad_views[names[3]].clone() * 0 + 4.0  # foreign again
    flat.collect_grads()
    for p, n, want in zip(flat.params, names, (1.5, 2.5, 0.0, 4.0)):
        assert p.grad.data_ptr() == flat.grad_views[n].data_ptr()
        assert bool((p.grad == want).all()) and bool((flat.grad_views[n] == want).all())


@pytest.mark.parametrize('family,extra', [('transformer', dict(N_enc=1, N_dec=1, d_model=16, d_ff=32, num_att_heads=2, dropout=0.1)),
                                          ('aoa', dict(num_heads=2, num_layers=2)), ('newfc', {})])
def test_raw_logit_rollouts_are_refused_where_logprobs_would_come_back(family, extra):
    """LossWrapper asks for output_logsoftmax=0 with the margin structure losses (loss_wrapper.py:34-35).
    r5: the sampled / greedy rollouts of every family store raw logits (tests/test_model_api_gpu.py); beam search and the
    host-stepped decode-time options still return log-probabilities and must refuse instead of silently handing those to a margin
    loss -- before touching the device (no GPU needed to see the error)."""
    import torch
    from imagecaptioning.pytorch_amd import synthetic
    from imagecaptioning.pytorch_amd.captioning import models
    opt = synthetic.updown_opt(caption_model=family, input_encoding_size=16, rnn_size=16, att_hid_size=8, seq_length=5, max_length=5,
                               vocab_size=20, fc_feat_size=12, att_feat_size=12, vocab={str(i): 'w%d' % i for i in range(1, 21)}, **extra)
    model = models.setup(opt)
    fc, att = torch.zeros(2, 12), torch.zeros(2, 3, 12)
    with pytest.raises(NotImplementedError, match='output_logsoftmax=0'):
        model(fc, att, None, opt={'beam_size': 2, 'output_logsoftmax': 0}, mode='sample')
    with pytest.raises(NotImplementedError, match='output_logsoftmax=0'):
        model(fc, att, None, opt={'sample_method': 'sample', 'sample_n': 2, 'block_trigrams': 1, 'output_logsoftmax': 0}, mode='sample')


def test_synthetic_loader_ranks_hold_equal_shares_and_wrap_together():
    """with synthetic_images % world != 0 the ranks used to hold different counts and raise `wrapped` (=> epoch, lr
    decay, ss_prob, the XE -> SCST switch) on different iterations.  Each rank now holds ceil(n / world) images, the tail padded
    from the head of the pass as FeatureLoader._mine does."""
    import argparse
    from imagecaptioning.pytorch_amd.captioning.data.synthetic_loader import SyntheticLoader
    opt = argparse.Namespace(batch_size=2, seq_per_img=2, seq_length=5, vocab_size=20, seed=1, synthetic_images=7, fc_feat_size=4,
                             att_feat_size=4, synthetic_regions=3)
    loaders = [SyntheticLoader(opt, rank=r, world=3) for r in range(3)]
    assert [len(l.mine) for l in loaders] == [3, 3, 3]
    assert sorted(set(sum((l.mine for l in loaders), []))) == list(range(7))          # every image is somebody's
    wraps = [[l.get_batch('train')['bounds']['wrapped'] for _ in range(6)] for l in loaders]
    assert wraps[0] == wraps[1] == wraps[2] and any(wraps[0])
    assert all(l.get_batch('val')['bounds']['it_max'] == 3 for l in loaders)


def test_validation_loss_returns_sum_and_count_and_survives_an_empty_partition():
    """per-rank validation = (sum of per-image losses, images) so the all-reduce weighs every image once; a rank's
    share is val_images / world, capped by what its partition holds; an empty partition contributes (0, 0)."""
    import argparse
    import torch
    from imagecaptioning.pytorch_amd.tools import train as T

    class Model:
        def eval(self): pass
        def train(self): pass
        def __call__(self, fc, att, seq, am): return fc.sum(1)

    class LW:
        model = Model()
        @staticmethod
        def crit(logp, labels, masks): return logp.mean()

    class Loader:
        def __init__(self, n): self.n, self.pos = n, 0
        def reset_iterator(self, split): self.pos = 0
        def get_batch(self, split):
            if self.n == 0:
                from captioning.data.feature_loader import EmptySplit
                raise EmptySplit('split has no images')       # (any other exception must propagate: ADVICE r5)
            B = min(2, self.n - self.pos) or 2
            self.pos = (self.pos + B) % self.n if self.pos + B < self.n else 0
            z = torch.zeros(B, 1, 3)
            return {'fc_feats': torch.ones(B, 4), 'att_feats': torch.zeros(B, 2, 4), 'labels': z.long(), 'masks': z, 'att_masks': None,
                    'bounds': {'it_max': self.n}}
    opt = argparse.Namespace(val_images=40, batch_size=2)
    tot, n = T.validation_loss(LW, Loader(5), opt, 'cpu', world=4)        # share 10, partition 5: every image once
    assert n == 5 and abs(tot - 4.0 * 5) < 1e-6
    tot, n = T.validation_loss(LW, Loader(0), opt, 'cpu', world=4)
    assert (tot, n) == (0.0, 0)
    tot, n = T.validation_loss(LW, Loader(100), opt, 'cpu', world=1)
    assert n == 40

    class Broken(Loader):                                     # a genuine data error is NOT an empty partition
        def get_batch(self, split):
            raise ValueError('all input arrays must have the same shape')
    with pytest.raises(ValueError, match='same shape'):
        T.validation_loss(LW, Broken(5), opt, 'cpu', world=4)


@pytest.mark.parametrize('family,extra', [('transformer', dict(N_enc=1, N_dec=1, d_model=16, d_ff=32, num_att_heads=2, dropout=0.1)),
                                          ('aoa', dict(num_heads=2, num_layers=2)), ('newfc', {}), ('updown', {})])
def test_every_sample_option_reaches_the_device_check(family, extra):
    """The option handling of `_sample` (beam search incl. sample_method='beam_search', the sampling methods, raw logits, decode
    constraints, diverse sampling) is plain host logic in front of the kernels: on CPU tensors every combination must arrive at
    the loud "HIP device only" error -- not at a NameError / a parse error of its own making (r5: moving one parse call in
    NewFCModel._sample broke sample_method='beam_search', and only the GPU suite noticed)."""
    import torch
    from imagecaptioning.pytorch_amd import synthetic
    from imagecaptioning.pytorch_amd.captioning import models
    from imagecaptioning.pytorch_amd._lib import CapmiError
    opt = synthetic.updown_opt(caption_model=family, input_encoding_size=16, rnn_size=16, att_hid_size=8, seq_length=5, max_length=5,
                               vocab_size=20, fc_feat_size=12, att_feat_size=12, vocab={str(i): 'w%d' % i for i in range(1, 21)}, **extra)
    model = models.setup(opt)
    fc, att = torch.zeros(2, 12), torch.zeros(2, 3, 12)
    combos = [dict(), dict(beam_size=2), dict(sample_method='beam_search', beam_size=2), dict(sample_method='sample', sample_n=2),
              dict(sample_method='top3', sample_n=2), dict(sample_method='top0.8'), dict(sample_method='gumbel', sample_n=2),
              dict(sample_method='sample', sample_n=2, output_logsoftmax=0), dict(sample_method='greedy', output_logsoftmax=0),
              dict(block_trigrams=1), dict(decoding_constraint=1, remove_bad_endings=1), dict(group_size=2, sample_method='sample'),
              dict(beam_size=2, group_size=2, diversity_lambda=0.5)]
    for o in combos:
        for train in (False, True):
            model.train(train)
            with pytest.raises(CapmiError, match='HIP device only'):
                model(fc, att, None, opt=o, mode='sample')
