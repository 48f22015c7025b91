"""The reward plumbing against the REAL reference call site.

``tests/golden/rewards_callsite.npz`` was produced by the reference's own ``captioning/utils/rewards.py``
(``get_self_critical_reward`` :41-81, ``get_scores`` :83-114, ``array_to_str`` :33-39) with only the external scorer
object stubbed (``tests/golden/make_golden.py rewards``).  It pins the strings handed to the scorer, the res/gts layout,
``cider_reward_weight``, the advantage and its repeat along L.  The CIDEr-D arithmetic stays unpinned (upstream absent)."""
import argparse
import os

import numpy as np
import pytest

from conftest import GOLDEN
from oracle import ciderd as C


def _load():
    z = np.load(os.path.join(GOLDEN, 'rewards_callsite.npz'))
    B = int(z['B'])
    gts = [z['gts_%d' % i] for i in range(B)]
    corpus = C.synthetic_corpus(int(z['corpus_images']), int(z['vocab']), 5, int(z['L']), seed=int(z['corpus_seed']))
    df, ref_len = C.build_document_frequency([[C.tokens_of(r) for r in g] for g in corpus])
    return z, gts, df, ref_len


def test_oracle_strings_are_the_references_array_to_str():
    z, gts, _, _ = _load()
    B, n = int(z['B']), int(z['n'])
    N = z['gen'].shape[0]
    rows = list(z['gen']) + list(z['greedy'])
    want = [str(s) for s in z['res_strings']]
    got = [' '.join(str(t) for t in C.tokens_of(r)) for r in rows]
    assert got == want
    assert want[0] == '0' and '0' not in want[1].split()           # EOS at step 0 / no EOS at all
    for i, s in enumerate(z['gts_strings']):
        img = i // n if i < N else i - N
        assert str(s) == '|'.join(' '.join(str(t) for t in C.tokens_of(r)) for r in gts[img])


def test_oracle_reward_matches_reference_call_site():
    z, gts, df, ref_len = _load()
    oracle = C.CiderD(df, ref_len)
    rew, scores = C.self_critical_reward(oracle, z['greedy'], gts, z['gen'])
    np.testing.assert_allclose(rew, z['reward_w1'], rtol=0, atol=1e-12)
    np.testing.assert_allclose(0.5 * rew, z['reward_w0.5'], rtol=0, atol=1e-12)
    np.testing.assert_allclose(C.sample_scores(oracle, gts, z['gen']), z['scores_w1'], rtol=0, atol=1e-12)
    assert (z['reward_w1'] == z['reward_w1'][:, :1]).all() and z['reward_w1'].shape == z['gen'].shape


@pytest.mark.gpu
def test_hip_reward_matches_reference_call_site():
    """the product's reference-compatible entry points (np.ndarray [N,L] float64 / [N]) against the reference's outputs"""
    import torch
    from imagecaptioning.pytorch_amd.captioning.utils import rewards as R
    z, gts, df, ref_len = _load()
    R.reset_scorer()
    R.init_scorer((df, ref_len), device=torch.device('cuda:0'))
    try:
        gen = torch.from_numpy(z['gen']).cuda()
        greedy = torch.from_numpy(z['greedy']).cuda()
        for w in (1.0, 0.5):
            opt = argparse.Namespace(cider_reward_weight=w, bleu_reward_weight=0)
            rew = R.get_self_critical_reward(greedy, gts, gen, opt)
            assert rew.dtype == np.float64 and rew.shape == z['gen'].shape
            np.testing.assert_allclose(rew, z['reward_w%g' % w], rtol=0, atol=1e-9)
            np.testing.assert_allclose(R.get_scores(gts, gen, opt), z['scores_w%g' % w], rtol=0, atol=1e-9)
            # the sync-free variant the LossWrapper uses: advantage [N] float32 on the device
            adv, _ = R.self_critical_reward_device(greedy, R.pack_gts(gts), gen, opt)
            np.testing.assert_allclose(adv.cpu().numpy(), z['reward_w%g' % w][:, 0], rtol=1e-5, atol=1e-5)
    finally:
        R.reset_scorer()


@pytest.mark.gpu
def test_two_batches_never_share_packed_references():
    """round 1 keyed a cache of packed references on id() of the arrays: a later batch whose arrays were allocated at
    recycled addresses got the previous batch's references.  Now the packed image is carried by the batch (GtsBatch) or
    rebuilt per call; batches built at the same addresses must score against their own references."""
    import torch
    from imagecaptioning.pytorch_amd.captioning.utils import rewards as R
    z, gts, df, ref_len = _load()
    R.reset_scorer()
    R.init_scorer((df, ref_len), device=torch.device('cuda:0'))
    try:
        opt = argparse.Namespace(cider_reward_weight=1.0, bleu_reward_weight=0)
        gen = torch.from_numpy(z['gen']).cuda()
        first = R.get_scores(gts, gen, opt)
        # overwrite the SAME array objects in place (same id(), same addresses) with other references
        keep = [g.copy() for g in gts]
        for g in gts:
            g[:] = np.roll(g, 3, axis=1)
            g[g == 0] = 7
        second = R.get_scores(gts, gen, opt)
        oracle = C.CiderD(df, ref_len)
        np.testing.assert_allclose(second, C.sample_scores(oracle, gts, z['gen']), rtol=0, atol=1e-9)
        assert np.abs(first - second).max() > 1e-3
        # an explicit handle keeps ITS references even if the source arrays change afterwards
        handle = R.pack_gts(keep)
        assert isinstance(handle, R.GtsBatch) and handle.packed is not None
        np.testing.assert_allclose(R.get_scores(handle, gen, opt), first, rtol=0, atol=1e-9)
        # and a subset selection drops the handle instead of mis-indexing it
        sub = R.select_gts(handle, torch.tensor([1, 0]))
        assert not isinstance(sub, R.GtsBatch) and len(sub) == 2
        assert R.select_gts(handle, torch.arange(len(keep))) is handle
    finally:
        R.reset_scorer()
