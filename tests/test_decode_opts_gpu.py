"""Decode-time options on the device vs fixtures of the REAL reference (tests/golden/*_tiny_opts.npz, made by
`make_golden.py opts`): decoding_constraint, remove_bad_endings, block_trigrams in AttModel._sample; _diverse_sample;
diverse / constrained / tempered beam search -- for all four model families -- plus the reference's step API
(get_logprobs_state) and the raw kernels against oracle/decode_opts.py."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN
from test_model_api_gpu import tiny_opt, DEV
from test_oracle_decode_opts import SAMPLE_CASES, DIVERSE_CASES, BEAM_CASES

pytestmark = pytest.mark.gpu
FAMILIES = ['updown', 'newfc', 'transformer', 'aoa']


def family(name):
    from imagecaptioning.pytorch_amd.captioning import models
    z = np.load(os.path.join(GOLDEN, name + '_tiny_opts.npz'))
    if name == 'transformer':
        opt = tiny_opt(caption_model='transformer', N_enc=2, N_dec=2, d_model=16, d_ff=32, num_att_heads=2, dropout=0.0)
    elif name == 'aoa':
        opt = tiny_opt(caption_model='aoa', refine=1, refine_aoa=1, use_ff=0, decoder_type='AoA', use_multi_head=2, num_heads=2,
                       multi_head_scale=1, mean_feats=1, ctx_drop=1, dropout_aoa=0.3, num_layers=2)
    else:
        opt = tiny_opt(caption_model=name)
    model = models.setup(opt)
    model.load_state_dict({k[2:]: torch.from_numpy(z[k]) for k in z.files if k.startswith('P.')})
    model = model.to(DEV).eval()
    model.bad_endings_ix = [int(v) for v in z['bad_endings_ix']]
    u = np.load(os.path.join(GOLDEN, 'updown_tiny.npz'))
    fc, att, am = (torch.from_numpy(u[k]).to(DEV) for k in ('fc', 'att', 'att_masks'))
    return z, model, fc, att, am


@pytest.mark.parametrize('name', FAMILIES)
def test_plain_greedy_of_the_option_fixture(name):
    z, model, fc, att, am = family(name)
    with torch.no_grad():
        seq, _ = model(fc, att, am, opt={'sample_method': 'greedy'}, mode='sample')
    assert np.array_equal(seq.cpu().numpy(), z['plain_seq'])


@pytest.mark.parametrize('name', FAMILIES)
@pytest.mark.parametrize('tag', sorted(SAMPLE_CASES))
def test_constrained_greedy_matches_reference(name, tag):
    z, model, fc, att, am = family(name)
    o = {'sample_method': 'greedy', 'beam_size': 1, 'sample_n': 1}
    o.update(SAMPLE_CASES[tag])
    with torch.no_grad():
        seq, slp = model(fc, att, am, opt=o, mode='sample')
    assert np.array_equal(seq.cpu().numpy(), z[tag + '_seq'])
    got, want = slp.cpu().numpy(), z[tag + '_logp']
    assert np.array_equal(np.isnan(got), np.isnan(want))           # -inf * 0 of finished rows, as in the reference
    assert np.array_equal(np.isneginf(got), np.isneginf(want))
    np.testing.assert_allclose(got, want, rtol=5e-5, atol=2e-5, equal_nan=True)


@pytest.mark.parametrize('name', FAMILIES)
@pytest.mark.parametrize('tag', sorted(DIVERSE_CASES))
def test_diverse_sample_matches_reference(name, tag):
    z, model, fc, att, am = family(name)
    o = {'sample_method': 'greedy', 'beam_size': 1, 'sample_n': 1}
    o.update(DIVERSE_CASES[tag])
    with torch.no_grad():
        seq, slp = model(fc, att, am, opt=o, mode='sample')
    assert np.array_equal(seq.cpu().numpy(), z[tag + '_seq'])
    np.testing.assert_allclose(slp.cpu().numpy(), z[tag + '_logp'], rtol=5e-5, atol=2e-5)


@pytest.mark.parametrize('name', FAMILIES)
@pytest.mark.parametrize('tag', sorted(BEAM_CASES))
def test_beam_search_options_match_reference(name, tag):
    z, model, fc, att, am = family(name)
    o = {'sample_method': 'beam_search', 'sample_n': 1}
    o.update(BEAM_CASES[tag])
    with torch.no_grad():
        seq, slp = model(fc, att, am, opt=o, mode='sample')
    assert np.array_equal(seq.cpu().numpy(), z[tag + '_seq'])
    got, want = slp.cpu().numpy(), z[tag + '_logp']
    assert np.array_equal(np.isneginf(got), np.isneginf(want))
    np.testing.assert_allclose(got, want, rtol=5e-5, atol=3e-5)
    for k, beams in enumerate(model.done_beams):
        assert len(beams) == int(z['%s_n%d' % (tag, k)])
        for j, bm in enumerate(beams):
            assert np.array_equal(bm['seq'].cpu().numpy(), z['%s_%d_%d_seq' % (tag, k, j)]), (k, j)
            np.testing.assert_allclose(bm['p'], z['%s_%d_%d_p' % (tag, k, j)], rtol=3e-5, atol=3e-5)
            if np.isfinite(z['%s_%d_%d_unaug' % (tag, k, j)]):
                np.testing.assert_allclose(bm['unaug_p'], z['%s_%d_%d_unaug' % (tag, k, j)], rtol=1e-4)


def test_get_logprobs_state_is_the_reference_step_api():
    """AttModel.get_logprobs_state / init_hidden / _prepare_feature (AttModel.py:99-124, 166-176) driven from outside, the way
    ensembles and custom searches do: a hand-rolled greedy loop over it reproduces the one-call rollout."""
    from imagecaptioning.pytorch_amd.captioning.models import utils as mu
    z, model, fc, att, am = family('updown')
    with torch.no_grad():
        want_seq, want_logp = model(fc, att, am, opt={'sample_method': 'greedy'}, mode='sample')
        n = 2
        p_fc, p_att, pp_att, p_masks = model._prepare_feature(fc, att, am)
        p_fc, p_att, pp_att, p_masks = mu.repeat_tensors(n, [p_fc, p_att, pp_att, p_masks])
        N = fc.shape[0] * n
        state = model.init_hidden(N)
        it = torch.zeros(N, dtype=torch.long, device=DEV)
        unfinished = torch.ones(N, dtype=torch.bool, device=DEV)
        for t in range(model.seq_length):
            logp, state = model.get_logprobs_state(it, p_fc, p_att, pp_att, p_masks, state)
            assert logp.shape == (N, model.vocab_size + 1) and state[0].shape == (2, N, model.rnn_size)
            it = logp.argmax(1) * unfinished
            np.testing.assert_allclose((logp * unfinished[:, None])[::n].cpu().numpy(), want_logp[:, t].cpu().numpy(), rtol=2e-5,
                                       atol=2e-6)
            assert torch.equal(it[::n], want_seq[:, t]) and torch.equal(it[1::n], want_seq[:, t])
            unfinished = unfinished & (it != 0)
        raw, _ = model.get_logprobs_state(it, p_fc, p_att, pp_att, p_masks, state, output_logsoftmax=0)
        logp, _ = model.get_logprobs_state(it, p_fc, p_att, pp_att, p_masks, state)
        np.testing.assert_allclose(torch.log_softmax(raw, 1).cpu().numpy(), logp.cpu().numpy(), rtol=1e-5, atol=1e-6)


def test_constraint_and_diversity_kernels_against_the_oracle():
    """capmi_decode_constrain / capmi_beam_diversity / capmi_column_penalty / capmi_select_logp on random rows at the
    BASELINE vocabulary size, bit-for-bit against the torch restatement (scattered adds of exactly representable terms)."""
    from imagecaptioning.pytorch_amd import _lib
    from imagecaptioning.pytorch_amd._lib import lib, ptr, check, stream_ptr
    from oracle import decode_opts as D
    g = torch.Generator().manual_seed(5)
    N, V1, L, t, B = 12, 9488, 20, 9, 7
    logp = torch.log_softmax(torch.randn(N, V1, generator=g) * 3, 1)
    seq = torch.randint(1, 6, (N, L), generator=g)              # few distinct words -> many repeated trigrams
    seq[3, 4:] = 0
    seq[:, t:] = 0
    bad = [2, 4]
    want = D._constrain(logp.clone(), seq, t, bad, True, True)
    want = D._trigram_penalty(want, seq, t, B)
    x, s = logp.clone().to(DEV), seq.to(DEV)
    badt = torch.tensor(bad, device=DEV)
    flags = _lib.DECODE_NO_REPEAT | _lib.DECODE_NO_BAD_ENDING | _lib.DECODE_BLOCK_TRIGRAMS
    check(lib.capmi_decode_constrain(ptr(x), N, V1, s.data_ptr() + 8 * (t - 1), L, flags, ptr(badt), 2, ptr(s), L, t, B,
                                     stream_ptr()), 'constrain')
    assert torch.equal(x.cpu(), want)
    assert (x.cpu()[B:] == D._constrain(logp.clone(), seq, t, bad, True, True)[B:]).all()      # rows >= B: no trigram term

    # add_diversity: B images, cur rows each, counts with multiplicity
    Bq, cur, n_prev, W = 3, 2, 4, 6
    lp = torch.log_softmax(torch.randn(Bq * cur, V1, generator=g), 1)
    prev = torch.randint(0, 5, (Bq, W), generator=g)
    change = torch.zeros(Bq, V1)
    for j in range(n_prev):
        change.scatter_add_(1, prev[:, j:j + 1], torch.ones(Bq, 1))
    want = lp - change.repeat_interleave(cur, 0) * 0.7
    out, lp_d, prev_d = torch.empty_like(lp).to(DEV), lp.to(DEV), prev.to(DEV)
    check(lib.capmi_beam_diversity(ptr(lp_d), ptr(out), Bq, cur, V1, ptr(prev_d), W, n_prev, 0.7, stream_ptr()), 'div')
    assert torch.equal(out.cpu(), want)

    # _diverse_sample's column penalty: every row, once per distinct token
    toks = torch.randint(0, 4, (N, L), generator=g)
    want = logp.clone()
    want[:, toks[:, 2]] = want[:, toks[:, 2]] - 0.5
    x, toks_d = logp.clone().to(DEV), toks.to(DEV)
    check(lib.capmi_column_penalty(ptr(x), N, V1, toks_d.data_ptr() + 8 * 2, N, L, 0.5, stream_ptr()), 'colpen')
    assert torch.equal(x.cpu(), want)

    # selection from constrained rows: arg-max, stored rows = rows * unfinished (NaN where -inf * 0)
    rows = want.clone()
    rows[:, 7] = float('-inf')
    unf = torch.tensor([1, 0] * (N // 2), dtype=torch.uint8)
    seq_out = torch.zeros(N, L, dtype=torch.long, device=DEV)
    dense = torch.zeros(N, L, V1, device=DEV)
    it = torch.zeros(N, dtype=torch.long, device=DEV)
    unf_d, rows_d = unf.clone().to(DEV), rows.to(DEV)
    check(lib.capmi_select_logp(ptr(rows_d), N, V1, 3, L, 0, 1.0, None, 0, ptr(seq_out), L, ptr(it), ptr(unf_d), ptr(dense),
                                None, 0, None, stream_ptr()), 'select')
    want_it = rows.argmax(1) * unf.long()
    assert torch.equal(it.cpu(), want_it) and torch.equal(seq_out[:, 3].cpu(), want_it)
    want_dense = rows * unf[:, None].float()
    got = dense[:, 3].cpu()
    assert torch.equal(torch.isnan(got), torch.isnan(want_dense))
    assert torch.equal(torch.nan_to_num(got, nan=0.0, neginf=-1e30), torch.nan_to_num(want_dense, nan=0.0, neginf=-1e30))
    assert torch.equal(unf_d.cpu(), (unf.bool() & (want_it != 0)).to(torch.uint8))


def test_sampling_with_constraints_never_repeats_a_token():
    """Categorical sampling under decoding_constraint: the previous token has probability zero (property over many draws)."""
    z, model, fc, att, am = family('updown')
    torch.manual_seed(3)
    with torch.no_grad():
        seq, slp = model(fc, att, am, opt={'sample_method': 'sample', 'sample_n': 16, 'decoding_constraint': 1, 'temperature': 1.5},
                         mode='sample')
    s = seq.cpu()
    rep = (s[:, 1:] == s[:, :-1]) & (s[:, 1:] != 0)
    assert not rep.any()
    assert len(torch.unique(s, dim=0)) > 8                       # genuinely sampled


def _full_size_model_and_oracle(B):
    """UpDown at the BASELINE sizes (R=E=1000, A=512, V1=9488, K=36, L=20) with random weights, on both sides."""
    import argparse
    from imagecaptioning.pytorch_amd.captioning import models
    from oracle import att_lstm as O
    from test_updown_gpu import full_size_params
    P = full_size_params(seed=5)
    V = 9487
    vocab = {str(i): 'w%d' % i for i in range(1, V + 1)}
    opt = argparse.Namespace(caption_model='updown', vocab_size=V, input_encoding_size=1000, rnn_size=1000, num_layers=1,
                             drop_prob_lm=0.5, seq_length=20, max_length=20, fc_feat_size=2048, att_feat_size=2048, att_hid_size=512,
                             use_bn=0, logit_layers=1, vocab=vocab)
    model = models.setup(opt)
    model.load_state_dict(P)
    model = model.to(DEV).eval()
    g = torch.Generator().manual_seed(8)
    fc = (torch.randn(B, 2048, generator=g) * 0.5).clamp_min(0)
    att = (torch.randn(B, 36, 2048, generator=g) * 0.5).clamp_min(0)
    feats = O.prepare_feature(P, fc, att, None)

    def stepper(n):
        f = O.repeat_rows(n, *feats) if n > 1 else feats
        return lambda it, state: O.updown_step(P, it, f[0], f[1], f[2], f[3], state)
    return model, P, fc, att, stepper, O


def test_full_size_constrained_greedy_and_diverse_sample_vs_oracle():
    """decoding_constraint + block_trigrams + remove_bad_endings in AttModel._sample and a 3-group _diverse_sample at the
    BASELINE sizes: tokens exact, log-probs to 1e-4 against oracle/decode_opts.py (itself pinned to the reference)."""
    from oracle import decode_opts as D
    B, L, V1 = 4, 20, 9488
    model, P, fc, att, stepper, O = _full_size_model_and_oracle(B)
    with torch.no_grad():
        plain, _ = model(fc.to(DEV), att.to(DEV), None, opt={'sample_method': 'greedy'}, mode='sample')
    bad = sorted({int(w) for row in plain.cpu().tolist() for w in row[:3] if w > 0})       # words the decode really uses
    model.bad_endings_ix = bad
    kw = dict(decoding_constraint=1, remove_bad_endings=1, block_trigrams=1)
    with torch.no_grad():
        want_seq, want_lp = D.constrained_sample(stepper(1), O.zero_state(P, B), B, B, V1, L, bad_endings=bad, **kw)
        seq, lp = model(fc.to(DEV), att.to(DEV), None, opt=dict(sample_method='greedy', **kw), mode='sample')
    assert torch.equal(seq.cpu(), want_seq)
    got, want = lp.cpu(), want_lp
    assert torch.equal(torch.isnan(got), torch.isnan(want)) and torch.equal(torch.isneginf(got), torch.isneginf(want))
    fin = torch.isfinite(want)
    assert float((got[fin] - want[fin]).abs().max()) < 1e-4
    with torch.no_grad():
        want_seq, want_lp = D.diverse_sample(stepper(1), lambda: O.zero_state(P, B), B, V1, L, 3, diversity_lambda=0.7,
                                             decoding_constraint=1)
        seq, lp = model(fc.to(DEV), att.to(DEV), None, opt=dict(sample_method='greedy', group_size=3, diversity_lambda=0.7,
                                                              decoding_constraint=1), mode='sample')
    assert torch.equal(seq.cpu(), want_seq)
    assert float((lp.cpu() - want_lp).abs().max()) < 1e-4
    assert len({tuple(r) for r in seq.cpu().tolist()}) > B           # the groups really differ


def test_full_size_diverse_beam_search_vs_oracle():
    """Diverse beam search (beam 6 = 3 groups x 2, lambda 0.5, decoding_constraint) at the BASELINE sizes vs the oracle:
    every finished beam's tokens, score and un-augmented score."""
    from oracle import decode_opts as D
    B, L, V1 = 2, 20, 9488
    model, P, fc, att, stepper, O = _full_size_model_and_oracle(B)
    kw = dict(beam_size=6, group_size=3, diversity_lambda=0.5, decoding_constraint=1)
    with torch.no_grad():
        logp0, state = stepper(1)(torch.zeros(B, dtype=torch.long), O.zero_state(P, B))
        want_seq, want_lp, want_done = D.beam_search(stepper(2), state, logp0, V1, L, unk_col=None, **kw)
        seq, lp = model(fc.to(DEV), att.to(DEV), None, opt=dict(sample_method='beam_search', sample_n=1, suppress_UNK=0, **kw),
                        mode='sample')
    assert torch.equal(seq.cpu(), want_seq)
    fin = torch.isfinite(want_lp)
    assert torch.equal(torch.isfinite(lp.cpu()), fin) and float((lp.cpu()[fin] - want_lp[fin]).abs().max()) < 1e-4
    for k in range(B):
        assert len(model.done_beams[k]) == len(want_done[k]) == 6
        for a, b in zip(model.done_beams[k], want_done[k]):
            assert torch.equal(a['seq'].cpu(), b['seq'])
            assert abs(a['p'] - b['p']) < 1e-3
