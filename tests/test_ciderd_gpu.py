"""CIDEr-D kernel vs the float64 oracle (oracle/ciderd.py; parity unpinned upstream, see there)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def test_table_roundtrip_cpu_side():
    from imagecaptioning.pytorch_amd import ciderd as D
    df = {(1,): 3.0, (1, 2): 2.0, (5, 0): 1.0, (9, 9, 9, 9): 7.0, (0,): 4.0}
    keys, vals = D.build_table(df)
    for g, c in df.items():
        k = np.uint64(D.pack_ngram(g))
        s = int(D._mix64(np.array([k], dtype=np.uint64))[0] & np.uint64(len(keys) - 1))
        while keys[s] != k:
            assert keys[s] != 0
            s = (s + 1) & (len(keys) - 1)
        assert vals[s] == c


@pytest.mark.parametrize('vocab,L,B,n', [(12, 6, 3, 2), (9487, 20, 10, 5), (50, 16, 7, 1)])
def test_ciderd_kernel_matches_oracle(vocab, L, B, n):
    from oracle import ciderd as C
    from imagecaptioning.pytorch_amd import ciderd as D
    rng = np.random.default_rng(vocab)
    corpus = C.synthetic_corpus(300, vocab, 5, L, seed=3)
    df, ref_len = C.build_document_frequency([[C.tokens_of(r) for r in g] for g in corpus])
    oracle = C.CiderD(df, ref_len)
    gts = corpus[:B]
    N = B * n
    # hypotheses: mixtures of reference fragments and noise so scores are non-trivial
    sampled = np.zeros((N, L), dtype=np.int64)
    for i in range(N):
        ref = gts[i // n][rng.integers(0, 5)]
        row = ref.astype(np.int64).copy()
        flip = rng.random(L) < 0.3
        row[flip] = rng.integers(0, vocab + 1, size=int(flip.sum()))
        sampled[i] = row
    sampled[0] = gts[0][0]                  # exact copy
    sampled[N - 1, :] = rng.integers(1, vocab + 1, size=L)   # no EOS at all
    greedy = np.stack([gts[i][1].astype(np.int64) for i in range(B)])
    greedy[0, 3:] = 0
    rewards_ref, scores_ref = C.self_critical_reward(oracle, greedy, gts, sampled)
    dev = D.DeviceCiderD(df, ref_len, DEV)
    refs, n_refs = dev.pack_refs(gts)
    reward, scores = dev.self_critical_reward(torch.from_numpy(greedy).to(DEV), torch.from_numpy(sampled).to(DEV), refs,
                                              n_refs, n)
    np.testing.assert_allclose(scores.cpu().numpy(), scores_ref, rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(reward.cpu().numpy(), rewards_ref[:, 0], rtol=1e-5, atol=1e-6)
    assert abs(scores_ref[0] - 10.0) < 1e-9 or n_refs[0] > 1
    # the independent plain-C restatement (oracle/ciderd_c) as a second checker of the kernel
    import os
    import subprocess
    subprocess.check_call(['make', '-s', '-C', os.path.join(os.path.dirname(os.path.abspath(C.__file__)), 'ciderd_c')])
    hyp_all = np.concatenate([sampled, greedy], 0)
    img_all = np.concatenate([np.arange(N) // n, np.arange(B)])
    scores_c = C.CiderDRefC(df, ref_len).score(hyp_all, img_all, gts)
    np.testing.assert_allclose(scores.cpu().numpy(), scores_c, rtol=1e-10, atol=1e-12)
    # references cooked once per batch (capmi_ciderd_cook_refs) must give bit-identical scores
    packed = dev.pack_refs(gts)
    assert packed.cooked is not None and packed.cooked.shape == (B * refs.shape[1], dev.COOKED_BYTES)
    hyp_t = torch.from_numpy(hyp_all).to(DEV)
    img_t = torch.from_numpy(img_all.astype(np.int32)).to(DEV)
    plain = dev.score(hyp_t, img_t, refs, n_refs)
    cooked = dev.score(hyp_t, img_t, refs, n_refs, packed.cooked)
    assert torch.equal(plain, cooked)
    assert torch.equal(plain, scores)


def test_ciderd_edge_cases_empty_ragged_and_full_length():
    """Edge cases of the reward path (rewards.py:33-81): a hypothesis that is EOS at step 0 (array_to_str keeps the first
    0, so it is the one-token caption "0"), hypotheses without any EOS (full length), images with a ragged number of
    references (1..5) of different widths, a hypothesis identical to its only reference, and n-grams absent from the
    document-frequency table."""
    from oracle import ciderd as C
    from imagecaptioning.pytorch_amd import ciderd as D
    vocab, L, B, n = 40, 12, 5, 3
    rng = np.random.default_rng(5)
    corpus = C.synthetic_corpus(100, vocab, 5, L, seed=9)
    df, ref_len = C.build_document_frequency([[C.tokens_of(r) for r in g] for g in corpus])
    oracle = C.CiderD(df, ref_len)
    gts = []
    for i in range(B):
        k = 1 + i % 5                                            # 1..5 references
        w = L - (i % 3)                                          # different widths
        g = np.zeros((k, w), dtype=np.uint32)
        for j in range(k):
            ln = int(rng.integers(1, w + 1))
            g[j, :ln] = rng.integers(1, vocab + 1, size=ln)
        gts.append(g)
    N = B * n
    sampled = np.zeros((N, L), dtype=np.int64)                   # rows left at zero: EOS at step 0
    sampled[1] = rng.integers(1, vocab + 1, size=L)              # no EOS at all
    sampled[2, :gts[0].shape[1]] = gts[0][0]                     # identical to image 0's only reference
    sampled[4, :5] = [vocab + 7, vocab + 8, vocab + 9, 1, 2]     # tokens never seen by the DF table
    for i in range(6, N):
        ln = int(rng.integers(1, L + 1))
        sampled[i, :ln] = rng.integers(1, vocab + 1, size=ln)
    greedy = np.zeros((B, L), dtype=np.int64)
    greedy[1, :4] = [3, 1, 4, 1]
    rewards_ref, scores_ref = C.self_critical_reward(oracle, greedy, gts, sampled)
    dev = D.DeviceCiderD(df, ref_len, DEV)
    refs, n_refs = dev.pack_refs(gts)
    assert n_refs.tolist() == [1, 2, 3, 4, 5]
    reward, scores = dev.self_critical_reward(torch.from_numpy(greedy).to(DEV), torch.from_numpy(sampled).to(DEV), refs,
                                              n_refs, n)
    np.testing.assert_allclose(scores.cpu().numpy(), scores_ref, rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(reward.cpu().numpy(), rewards_ref[:, 0], rtol=1e-5, atol=1e-6)
    assert np.isfinite(scores_ref).all()
    packed = dev.pack_refs(gts)                       # ragged reference counts / widths through the pre-cooked path
    reward2, scores2 = dev.self_critical_reward(torch.from_numpy(greedy).to(DEV), torch.from_numpy(sampled).to(DEV), refs,
                                                n_refs, n, cooked=packed.cooked)
    assert torch.equal(scores, scores2) and torch.equal(reward, reward2)
