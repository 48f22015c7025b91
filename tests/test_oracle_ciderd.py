"""Known-answer tests for the CIDEr-D restatement (oracle/ciderd.py).

The upstream scorer is absent from /root/reference (empty submodule) => PARITY UNPINNED; these are
the hand-derived cases of SURVEY.md Appendix A.4 plus the reference's own call-site semantics
(rewards.py:33-81)."""
import math

import numpy as np

from oracle import ciderd as C


def scorer(df=None, ref_len=100.0):
    return C.CiderD(df or {}, ref_len)


def test_tokens_keep_first_zero():
    assert C.tokens_of([5, 3, 0, 9, 0]) == [5, 3, 0]
    assert C.tokens_of([5, 3, 7]) == [5, 3, 7]
    assert C.tokens_of([0, 4]) == [0]


def test_identical_hyp_and_ref_scores_ten():
    s = scorer()
    h = [4, 8, 15, 16, 23, 0]
    assert abs(s.score_one(h, [h]) - 10.0) < 1e-12
    # two identical refs: sum over refs / n_refs -> unchanged
    assert abs(s.score_one(h, [h, h]) - 10.0) < 1e-12


def test_short_hyp_loses_orders_without_ngrams():
    # 2 tokens: only 1- and 2-grams exist -> 10 * 2/4
    s = scorer()
    assert abs(s.score_one([7, 0], [[7, 0]]) - 5.0) < 1e-12


def test_disjoint_is_zero():
    s = scorer()
    assert s.score_one([1, 2, 3, 4, 0], [[5, 6, 7, 8, 9]]) == 0.0


def test_length_penalty_uses_bigram_counts():
    """hyp = first 5 tokens of a 6-token ref => 'length' (number of bigrams) differs by 1:
    each order's cosine is computed by hand and the Gaussian factor is exp(-1/72)."""
    s = scorer()
    ref = [1, 2, 3, 4, 5, 6]
    hyp = [1, 2, 3, 4, 5]
    idf = math.log(100.0)
    expect = 0.0
    for k in range(1, 5):
        nh, nr = len(hyp) - k + 1, len(ref) - k + 1
        # all n-grams unique, tf=1, weight idf; hyp's are a subset of ref's
        dot = nh * idf * idf
        expect += dot / (math.sqrt(nh) * idf * math.sqrt(nr) * idf)
    expect = expect / 4 * math.exp(-1.0 / 72.0) * 10.0
    assert abs(s.score_one(hyp, [ref]) - expect) < 1e-12


def test_idf_from_table_and_full_df_zeroes_term():
    # DF == ref_len => idf 0 => that n-gram contributes nothing; unseen => idf = log(ref_len)
    df = {(1,): 100.0, (2,): 10.0}
    s = scorer(df, 100.0)
    vec, norm, length = s._vec(C.precook([1, 2]))
    assert vec[0][(1,)] == 0.0
    assert abs(vec[0][(2,)] - (math.log(100.0) - math.log(10.0))) < 1e-15
    assert abs(vec[1][(1, 2)] - math.log(100.0)) < 1e-15
    assert length == 1
    # a hypothesis whose only unigram has idf 0 has a zero norm; guarded, no NaN
    v = s.score_one([1], [[1]])
    assert v == 0.0


def test_clipping_min_of_hyp_and_ref_counts():
    # hyp repeats a word 3x, ref has it once: val uses min(vec_h, vec_r)*vec_r
    s = scorer()
    idf = math.log(100.0)
    hyp, ref = [9, 9, 9], [9]
    # unigram order: vec_h = 3 idf, vec_r = idf -> min*vec_r = idf^2 ; norms 3 idf, idf
    uni = (idf * idf) / (3 * idf * idf)
    # hyp has bigram (9,9) x2 and trigram x1, ref none -> 0 ; delta = len_h(2 bigrams) - len_r(0) = 2
    expect = uni / 4 * math.exp(-4.0 / 72.0) * 10
    assert abs(s.score_one(hyp, [ref]) - expect) < 1e-12


def test_self_critical_reward_layout():
    """rewards.py:41-81: N sampled then B greedy; refs of image i//n; advantage broadcast along L."""
    rng = np.random.default_rng(0)
    B, n, L = 3, 2, 6
    gts = C.synthetic_corpus(B, 12, refs_per_image=2, width=L, seed=1)
    df, ref_len = C.build_document_frequency([[C.tokens_of(r) for r in g] for g in C.synthetic_corpus(50, 12, 2, L, 2)])
    s = C.CiderD(df, ref_len)
    sampled = rng.integers(0, 13, size=(B * n, L))
    greedy = rng.integers(0, 13, size=(B, L))
    sampled[1] = gts[0][0]                      # an exact copy of a reference
    rewards, scores = C.self_critical_reward(s, greedy, gts, sampled)
    assert rewards.shape == (B * n, L) and scores.shape == (B * n + B,)
    assert (rewards == rewards[:, :1]).all()
    for i in range(B * n):
        refs = [C.tokens_of(r) for r in gts[i // n]]
        assert abs(scores[i] - s.score_one(C.tokens_of(sampled[i]), refs)) < 1e-12
        assert abs(rewards[i, 0] - (scores[i] - scores[B * n + i // n])) < 1e-12
    assert scores[1] > scores[0]
    np.testing.assert_allclose(C.sample_scores(s, gts, sampled), scores[:B * n])


def test_document_frequency_counts_images_not_occurrences():
    df, ref_len = C.build_document_frequency([[[1, 2, 1], [1, 2]], [[1, 3]]])
    assert ref_len == 2
    assert df[(1,)] == 2 and df[(2,)] == 1 and df[(1, 2)] == 1 and df[(2, 1)] == 1 and df[(3,)] == 1


def test_c_restatement_agrees_with_python_restatement():
    """oracle/ciderd_c/ciderd.c (sorted arrays + binary search) vs oracle/ciderd.py (dictionaries): two independent
    restatements of the same published algorithm must agree to float64 rounding -- on random hypotheses, exact copies,
    EOS-at-0 rows, rows without EOS, ragged reference counts and widths, and n-grams absent from the DF table."""
    import os
    import subprocess
    import numpy as np
    from oracle import ciderd as C
    cdir = os.path.join(os.path.dirname(os.path.abspath(C.__file__)), 'ciderd_c')
    subprocess.check_call(['make', '-s', '-C', cdir])
    rng = np.random.default_rng(11)
    vocab, L, B, n = 60, 14, 6, 3
    corpus = C.synthetic_corpus(200, vocab, 5, L, seed=2)
    df, ref_len = C.build_document_frequency([[C.tokens_of(r) for r in g] for g in corpus])
    gts = []
    for i in range(B):
        k, w = 1 + i % 5, L - (i % 3)
        g = np.zeros((k, w), dtype=np.uint32)
        for j in range(k):
            ln = int(rng.integers(1, w + 1))
            g[j, :ln] = rng.integers(1, vocab + 1, size=ln)
        gts.append(g)
    gts[1] = corpus[0][:, :L]                                     # an image whose refs are in the DF corpus
    N = B * n
    hyp = np.zeros((N, L), dtype=np.int64)
    for i in range(3, N):
        ln = int(rng.integers(1, L + 1))
        hyp[i, :ln] = rng.integers(1, vocab + 1, size=ln)
    hyp[1] = rng.integers(1, vocab + 1, size=L)                   # no EOS
    hyp[2, :gts[0].shape[1]] = gts[0][0]                          # copy of image 0's only reference
    hyp[4, :L] = corpus[0][1][:L].astype(np.int64)              # a reference of image 1 (rows 3..5 score against it)
    hyp[7, :4] = [vocab + 5, vocab + 6, 1, 2]                     # unseen tokens
    hyp_img = np.arange(N) // n
    py = C.CiderD(df, ref_len)
    ref_tok = [[C.tokens_of(r) for r in g] for g in gts]
    _, want = py.compute_score([C.tokens_of(r) for r in hyp], [ref_tok[i] for i in hyp_img])
    got = C.CiderDRefC(df, ref_len).score(hyp, hyp_img, gts)
    np.testing.assert_allclose(got, want, rtol=1e-12, atol=1e-13)
    # an exact copy of the only reference scores 10 on every n-gram order it is long enough to have
    assert got[2] >= 2.5 - 1e-9 and abs(got[2] / 2.5 - round(got[2] / 2.5)) < 1e-9
    assert got[4] > 2.0 and got[0] >= 0.0 and got.max() <= 10.0 + 1e-9
