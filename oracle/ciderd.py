"""float64 CPU restatement of CIDEr-D and of the reference's reward plumbing.

TEST INFRASTRUCTURE (see ``oracle/__init__.py``).

**PARITY UNPINNED.**  The arithmetic lives in the third-party git submodule ``cider`` ->
``https://github.com/ruotianluo/cider.git`` (``/root/reference/.gitmodules:1-3``), whose directory
is EMPTY in the reference checkout and whose pinned commit is unknowable (no ``.git``).  What
follows restates the published algorithm of ``pyciderevalcap/ciderD/ciderD_scorer.py`` /
``ciderD.py`` (n = 4, sigma = 6.0, pickle document frequencies) as summarised in SURVEY.md
Appendix A, and is anchored on the reference's own call sites:

* ``captioning/utils/rewards.py:33-39``  ``array_to_str``: space-joined decimal ids, cut AFTER the
  first 0 (the 0 is kept as a token);
* ``rewards.py:41-81``  ``get_self_critical_reward``: N sampled + B greedy hypotheses, refs of image
  ``i // n`` resp. ``i - N``, advantage = sample - greedy, repeated along L;
* ``rewards.py:83-114`` ``get_scores`` (new-self-critical: sampled hypotheses only);
* ``scripts/prepro_ngrams.py:17-80`` the pickle ``{'document_frequency': {ngram tuple -> #images},
  'ref_len': #images}``.

The reference's tests hold no vectors for it; the only pins are the hand-derived known-answer tests
of SURVEY.md A.4 (``tests/test_oracle_ciderd.py``).
"""
from __future__ import annotations

import math
from collections import defaultdict
from typing import Dict, List, Sequence, Tuple

import numpy as np

NGRAM_MAX = 4
SIGMA = 6.0


def tokens_of(row: Sequence[int]) -> List[int]:
    """rewards.py:33-39: ids up to and INCLUDING the first 0."""
    out = []
    for v in row:
        v = int(v)
        out.append(v)
        if v == 0:
            break
    return out


def precook(words: Sequence[int]) -> Dict[Tuple[int, ...], int]:
    """upstream ``precook``: count every k-gram, k = 1..4."""
    counts: Dict[Tuple[int, ...], int] = defaultdict(int)
    for k in range(1, NGRAM_MAX + 1):
        for i in range(len(words) - k + 1):
            counts[tuple(words[i:i + k])] += 1
    return counts


class CiderD:
    """CiderScorer(df_mode=<pickle>) semantics: DF table and ref_len are fixed, no corpus pass."""

    def __init__(self, document_frequency: Dict[Tuple[int, ...], float], ref_len: float):
        self.df = document_frequency
        self.log_ref_len = math.log(float(ref_len))

    # upstream counts2vec
    def _vec(self, counts):
        vec = [dict() for _ in range(NGRAM_MAX)]
        norm = [0.0] * NGRAM_MAX
        length = 0
        for ngram, tf in counts.items():
            df = math.log(max(1.0, float(self.df.get(ngram, 0.0))))
            k = len(ngram) - 1
            v = float(tf) * (self.log_ref_len - df)
            vec[k][ngram] = v
            norm[k] += v * v
            if k == 1:                       # upstream quirk: "length" counts bigrams only
                length += tf
        return vec, [math.sqrt(x) for x in norm], length

    # upstream sim
    def _sim(self, vh, vr, nh, nr, lh, lr):
        delta = float(lh - lr)
        val = np.zeros(NGRAM_MAX)
        for k in range(NGRAM_MAX):
            acc = 0.0
            for ngram, h in vh[k].items():
                r = vr[k].get(ngram, 0.0)
                acc += min(h, r) * r
            if nh[k] != 0 and nr[k] != 0:
                acc /= nh[k] * nr[k]
            val[k] = acc * math.e ** (-(delta ** 2) / (2 * SIGMA ** 2))
        return val

    def score_one(self, hyp: Sequence[int], refs: Sequence[Sequence[int]]) -> float:
        vh, nh, lh = self._vec(precook(hyp))
        score = np.zeros(NGRAM_MAX)
        for ref in refs:
            vr, nr, lr = self._vec(precook(ref))
            score += self._sim(vh, vr, nh, nr, lh, lr)
        return float(np.mean(score) / len(refs) * 10.0)

    def compute_score(self, hyps: Sequence[Sequence[int]], refs_per_hyp: Sequence[Sequence[Sequence[int]]]):
        scores = np.array([self.score_one(h, r) for h, r in zip(hyps, refs_per_hyp)])
        return float(np.mean(scores)), scores


def build_document_frequency(ref_sets: Sequence[Sequence[Sequence[int]]]):
    """scripts/prepro_ngrams.py:17-22 + upstream ``compute_doc_freq``: an n-gram counts once per
    image (set over all refs of the image).  Returns (df dict, ref_len = #images)."""
    df: Dict[Tuple[int, ...], float] = defaultdict(float)
    for refs in ref_sets:
        seen = set()
        for ref in refs:
            seen.update(precook(ref).keys())
        for g in seen:
            df[g] += 1.0
    return df, len(ref_sets)


def self_critical_reward(scorer: CiderD, greedy: np.ndarray, gts: Sequence[np.ndarray],
                         sampled: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    """rewards.py:41-81 with cider_reward_weight 1, bleu_reward_weight 0 (opts.py:171,185 defaults).
    Returns (rewards [N,L] float64, raw scores [N+B])."""
    B = len(gts)
    N, L = sampled.shape
    n = N // B
    assert greedy.shape[0] == B
    hyps = [tokens_of(sampled[i]) for i in range(N)] + [tokens_of(greedy[i]) for i in range(B)]
    ref_tok = [[tokens_of(r) for r in gts[i]] for i in range(B)]
    refs = [ref_tok[i // n] for i in range(N)] + [ref_tok[i] for i in range(B)]
    _, scores = scorer.compute_score(hyps, refs)
    adv = scores[:N].reshape(B, n) - scores[N:][:, None]
    return np.repeat(adv.reshape(N)[:, None], L, 1), scores


def sample_scores(scorer: CiderD, gts: Sequence[np.ndarray], sampled: np.ndarray) -> np.ndarray:
    """rewards.py:83-114 ``get_scores`` (cider weight 1): CIDEr-D of each sampled row."""
    B = len(gts)
    N = sampled.shape[0]
    n = N // B
    ref_tok = [[tokens_of(r) for r in gts[i]] for i in range(B)]
    hyps = [tokens_of(sampled[i]) for i in range(N)]
    _, scores = scorer.compute_score(hyps, [ref_tok[i // n] for i in range(N)])
    return scores


def synthetic_corpus(num_images: int, vocab: int, refs_per_image: int = 5, width: int = 20,
                     seed: int = 1234, zipf_a: float = 1.2):
    """Synthetic reference sets with a Zipfian unigram distribution: rows are
    uint32 [refs_per_image, width], lengths ~U{8..width}, zero padded."""
    rng = np.random.default_rng(seed)
    ranks = np.arange(1, vocab + 1, dtype=np.float64)
    p = ranks ** (-zipf_a)
    p /= p.sum()
    out = []
    for _ in range(num_images):
        arr = np.zeros((refs_per_image, width), dtype=np.uint32)
        for j in range(refs_per_image):
            ln = int(rng.integers(min(8, width), width + 1))
            arr[j, :ln] = rng.choice(vocab, size=ln, p=p) + 1
        out.append(arr)
    return out


# --------------------------------------------------------------------------------------------------
# second, independent restatement in plain C (oracle/ciderd_c/ciderd.c): sorted key arrays + binary search
# --------------------------------------------------------------------------------------------------
class CiderDRefC:
    """ctypes wrapper of libciderd_ref.so (make -C oracle/ciderd_c).  Same contract as :class:`CiderD`; used to
    cross-check the two restatements (tests/test_oracle_ciderd.py) and as a fast CPU scorer."""

    def __init__(self, document_frequency, ref_len, lib_path=None):
        import ctypes as C
        import os
        path = lib_path or os.path.join(os.path.dirname(os.path.abspath(__file__)), 'ciderd_c', 'libciderd_ref.so')
        self._lib = C.CDLL(path)
        self._lib.ciderd_ref_score.restype = C.c_int
        keys = np.fromiter((self._pack(g) for g in document_frequency.keys()), dtype=np.uint64, count=len(document_frequency))
        vals = np.fromiter((float(v) for v in document_frequency.values()), dtype=np.float64, count=len(document_frequency))
        order = np.argsort(keys, kind='stable')
        self.keys, self.vals = np.ascontiguousarray(keys[order]), np.ascontiguousarray(vals[order])
        self.log_ref_len = math.log(float(ref_len))

    @staticmethod
    def _pack(tokens):
        k = 0
        for q, t in enumerate(tokens):
            k |= (int(t) + 1) << (16 * q)
        return k

    def score(self, hyp: np.ndarray, hyp_img: np.ndarray, gts: Sequence[np.ndarray]) -> np.ndarray:
        """hyp [H,L] int64 rows, hyp_img [H] image of each row, gts list of [n_ref, w] arrays -> scores [H]."""
        import ctypes as C
        B = len(gts)
        max_refs = max(len(g) for g in gts)
        w = max(np.asarray(g).shape[1] for g in gts)
        refs = np.zeros((B, max_refs, w), dtype=np.int32)
        n_refs = np.zeros(B, dtype=np.int32)
        for i, g in enumerate(gts):
            g = np.asarray(g).astype(np.int32)
            refs[i, :g.shape[0], :g.shape[1]] = g
            if g.shape[1] < w:
                refs[i, :g.shape[0], g.shape[1]][(g != 0).all(1)] = -1      # full rows of a narrower array keep no EOS
            n_refs[i] = g.shape[0]
        hyp = np.ascontiguousarray(hyp, dtype=np.int64)
        hyp_img = np.ascontiguousarray(hyp_img, dtype=np.int32)
        out = np.zeros(hyp.shape[0], dtype=np.float64)
        p = lambda a: a.ctypes.data_as(C.c_void_p)                            # noqa: E731
        rc = self._lib.ciderd_ref_score(p(hyp), C.c_int(hyp.shape[0]), C.c_int(hyp.shape[1]), p(hyp_img), p(refs), p(n_refs),
                                        C.c_int(max_refs), C.c_int(w), p(self.keys), p(self.vals), C.c_int64(len(self.keys)),
                                        C.c_double(self.log_ref_len), p(out))
        if rc:
            raise RuntimeError('ciderd_ref_score failed: %d' % rc)
        return out
