"""Pins oracle/decode_opts.py (decoding constraints, diverse sampling, diverse / constrained beam search) to the REAL
reference through tests/golden/updown_tiny_opts.npz (tests/golden/make_golden.py opts).  CPU only."""
import os

import numpy as np
import pytest
import torch

from oracle import att_lstm as O
from oracle import decode_opts as D
from conftest import GOLDEN

L = 8
SAMPLE_CASES = {
    'dc': dict(decoding_constraint=1),
    'rbe': dict(remove_bad_endings=1),
    'tri': dict(block_trigrams=1),
    'all': dict(decoding_constraint=1, remove_bad_endings=1, block_trigrams=1),
    'tri_n2': dict(block_trigrams=1, decoding_constraint=1, sample_n=2),
}
DIVERSE_CASES = {
    'div3': dict(group_size=3, diversity_lambda=0.5),
    'div2c': dict(group_size=2, diversity_lambda=0.8, decoding_constraint=1, remove_bad_endings=1, temperature=1.5),
}
BEAM_CASES = {
    'bT': dict(beam_size=3, temperature=2.0),
    'bdc': dict(beam_size=3, decoding_constraint=1, remove_bad_endings=1),
    'bg2': dict(beam_size=4, group_size=2, diversity_lambda=0.5),
    'bg3': dict(beam_size=3, group_size=3, diversity_lambda=1.0, temperature=1.3),
    'bg2c': dict(beam_size=4, group_size=2, diversity_lambda=0.5, decoding_constraint=1, remove_bad_endings=1, sample_n=2,
                 length_penalty='wu_0.5'),
}


def setup(rows_per_image):
    z = np.load(os.path.join(GOLDEN, 'updown_tiny_opts.npz'))
    u = np.load(os.path.join(GOLDEN, 'updown_tiny.npz'))
    P = {k[2:]: torch.from_numpy(z[k]) for k in z.files if k.startswith('P.')}
    fc, att, am = (torch.from_numpy(u[k]) for k in ('fc', 'att', 'att_masks'))
    feats = O.prepare_feature(P, fc, att, am)
    B = fc.shape[0]

    def stepper(n):
        f = O.repeat_rows(n, *feats) if n > 1 else feats
        return lambda it, state: O.updown_step(P, it, f[0], f[1], f[2], f[3], state)
    return z, P, B, stepper, [int(v) for v in z['bad_endings_ix']]


@pytest.mark.parametrize('tag', sorted(SAMPLE_CASES))
def test_constrained_greedy_matches_reference(tag):
    kw = dict(SAMPLE_CASES[tag])
    n = kw.pop('sample_n', 1)
    z, P, B, stepper, bad = setup(n)
    V1 = P['logit.weight'].shape[0]
    with torch.no_grad():
        seq, slp = D.constrained_sample(stepper(n), O.zero_state(P, B * n), B * n, B, V1, L, bad_endings=bad, **kw)
    assert np.array_equal(seq.numpy(), z[tag + '_seq'])
    np.testing.assert_allclose(slp.numpy(), z[tag + '_logp'], rtol=1e-5, atol=2e-6, equal_nan=True)
    assert np.array_equal(np.isnan(slp.numpy()), np.isnan(z[tag + '_logp']))


@pytest.mark.parametrize('tag', sorted(DIVERSE_CASES))
def test_diverse_sample_matches_reference(tag):
    kw = dict(DIVERSE_CASES[tag])
    z, P, B, stepper, bad = setup(1)
    V1 = P['logit.weight'].shape[0]
    G = kw.pop('group_size')
    with torch.no_grad():
        seq, slp = D.diverse_sample(stepper(1), lambda: O.zero_state(P, B), B, V1, L, G, bad_endings=bad, **kw)
    assert np.array_equal(seq.numpy(), z[tag + '_seq'])
    np.testing.assert_allclose(slp.numpy(), z[tag + '_logp'], rtol=1e-5, atol=2e-6)


@pytest.mark.parametrize('tag', sorted(BEAM_CASES))
def test_beam_search_options_match_reference(tag):
    kw = dict(BEAM_CASES[tag])
    z, P, B, stepper, bad = setup(1)
    V1 = P['logit.weight'].shape[0]
    bd = kw['beam_size'] // kw.get('group_size', 1)
    with torch.no_grad():
        logp0, state = stepper(1)(torch.zeros(B, dtype=torch.long), O.zero_state(P, B))
        seq, slp, done = D.beam_search(stepper(bd), state, logp0, V1, L, bad_endings=bad, **kw)
    assert np.array_equal(seq.numpy(), z[tag + '_seq'])
    np.testing.assert_allclose(slp.numpy(), z[tag + '_logp'], rtol=1e-5, atol=2e-6)
    for k, beams in enumerate(done):
        assert len(beams) == int(z['%s_n%d' % (tag, k)])
        for j, bm in enumerate(beams):
            assert np.array_equal(bm['seq'].numpy(), z['%s_%d_%d_seq' % (tag, k, j)]), (k, j)
            np.testing.assert_allclose(bm['p'], z['%s_%d_%d_p' % (tag, k, j)], rtol=1e-5, atol=1e-5)
            np.testing.assert_allclose(bm['unaug_p'], z['%s_%d_%d_unaug' % (tag, k, j)], rtol=1e-4)
