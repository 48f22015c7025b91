"""Device-resident CIDEr-D scorer (host side): builds the open-addressing document-frequency table
consumed by ``capmi_ciderd_score`` and wraps the launch.

Replaces ``CiderD(df=cached_tokens)`` + ``CiderD_scorer.compute_score`` as used by
captioning/utils/rewards.py:25-31, 64, 101 (external pyciderevalcap; see oracle/ciderd.py for the
provenance note: PARITY UNPINNED).  The pickle format is the one written by
scripts/prepro_ngrams.py:79-80: ``{'document_frequency': {tuple[str] -> count}, 'ref_len': int}``.
"""
import math

import numpy as np
import torch

from . import _lib
from ._lib import lib, ptr, check, stream_ptr

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _mix64(x):
    """splitmix64 finaliser on a uint64 array (twin of mix64 in csrc/ciderd.hip)."""
    x = x.copy()
    with np.errstate(over='ignore'):
        x ^= x >> np.uint64(30)
        x *= np.uint64(0xbf58476d1ce4e5b9)
        x ^= x >> np.uint64(27)
        x *= np.uint64(0x94d049bb133111eb)
        x ^= x >> np.uint64(31)
    return x


def pack_ngram(tokens):
    """<= 4 token ids (each < 65535) -> uint64 key: 16-bit fields of (id + 1), first token lowest."""
    k = 0
    for q, t in enumerate(tokens):
        t = int(t)
        assert 0 <= t < 65535
        k |= (t + 1) << (16 * q)
    return k


def build_table(document_frequency, load_factor=0.5):
    """dict {tuple[int|str] -> count} -> (keys uint64[cap], vals float64[cap]); empty slot = key 0.
    Vectorised linear-probing insertion (the COCO table has a few million n-grams)."""
    n = len(document_frequency)
    cap = 1 << max(4, int(math.ceil(math.log2(max(1, n) / load_factor))))
    keys = np.zeros(cap, dtype=np.uint64)
    vals = np.zeros(cap, dtype=np.float64)
    if n == 0:
        return keys, vals
    k = np.fromiter((pack_ngram([int(t) for t in g]) for g in document_frequency.keys()), dtype=np.uint64, count=n)
    v = np.fromiter((float(c) for c in document_frequency.values()), dtype=np.float64, count=n)
    slot = (_mix64(k) & np.uint64(cap - 1)).astype(np.int64)
    pending = np.arange(n)
    while pending.size:
        s = slot[pending]
        free = keys[s] == 0
        # among the candidates for the same free slot keep the first
        cand = pending[free]
        cs = slot[cand]
        _, first = np.unique(cs, return_index=True)
        win = cand[first]
        keys[slot[win]] = k[win]
        vals[slot[win]] = v[win]
        placed = np.zeros(n, dtype=bool)
        placed[win] = True
        pending = pending[~placed[pending]]
        slot[pending] = (slot[pending] + 1) & (cap - 1)
    return keys, vals


def save_df_image(document_frequency, ref_len, path):
    """One-time conversion of the scripts/prepro_ngrams.py pickle content into the flat hash image the
    kernel probes: an .npz with `keys` uint64[cap], `vals` float64[cap], `ref_len`.  Loading it is two array reads
    instead of unpickling and re-hashing a few million tuple keys at every start-up."""
    keys, vals = build_table(document_frequency)
    np.savez(path, keys=keys, vals=vals, ref_len=np.float64(ref_len), n=np.int64(len(document_frequency)))
    return path if str(path).endswith('.npz') else str(path) + '.npz'


def load_df_image(path):
    z = np.load(path)
    keys, vals = z['keys'], z['vals']
    if keys.dtype != np.uint64 or vals.dtype != np.float64 or keys.shape != vals.shape or keys.shape[0] & (keys.shape[0] - 1):
        raise ValueError('%s is not a capmi document-frequency image' % path)
    return keys, vals, float(z['ref_len'])


class PackedRefs(tuple):
    """(refs int32 [B,max_refs,w], n_refs int32 [B]) -- unpacks like the pair it used to be -- plus `.cooked`"""

    def __new__(cls, refs, n_refs, cooked=None):
        self = super().__new__(cls, (refs, n_refs))
        self.cooked = cooked
        return self


class DeviceCiderD:
    def __init__(self, document_frequency, ref_len, device, _table=None):
        keys, vals = _table if _table is not None else build_table(document_frequency)
        self.cap = int(keys.shape[0])
        self.keys = torch.from_numpy(keys.view(np.int64)).to(device)     # bit pattern preserved
        self.vals = torch.from_numpy(vals).to(device)
        self.log_ref_len = math.log(float(ref_len))
        self.device = device

    @classmethod
    def from_image(cls, path, device):
        keys, vals, ref_len = load_df_image(path)
        return cls(None, ref_len, device, _table=(keys, vals))

    @classmethod
    def from_pickle(cls, path, device):
        import pickle
        with open(path, 'rb') as f:
            pkl = pickle.load(f, encoding='latin1')
        return cls(pkl['document_frequency'], pkl['ref_len'], device)

    def pack_refs(self, gts):
        """list (per image) of [n_ref_i, w] integer arrays -> (refs int32 [B,max_refs,w], n_refs int32 [B])."""
        B = len(gts)
        max_refs = max(len(g) for g in gts)
        w = max(np.asarray(g).shape[1] for g in gts)
        refs = np.zeros((B, max_refs, w), dtype=np.int32)
        n_refs = np.zeros(B, dtype=np.int32)
        for i, g in enumerate(gts):
            g = np.asarray(g).astype(np.int32)
            refs[i, :g.shape[0], :g.shape[1]] = g
            # rows of a NARROWER array that are completely filled carry no terminating 0 (array_to_str, rewards.py:33-39):
            # mark the first padded column with -1 so the kernel does not read the zero padding as an EOS token
            if g.shape[1] < w:
                full = (g != 0).all(1)
                refs[i, :g.shape[0], g.shape[1]][full] = -1
            n_refs[i] = g.shape[0]
        refs_d, n_refs_d = torch.from_numpy(refs).to(self.device), torch.from_numpy(n_refs).to(self.device)
        return PackedRefs(refs_d, n_refs_d, self.cook(refs_d, n_refs_d))

    COOKED_BYTES = 4392            # capmi.h CAPMI_CIDERD_COOKED_BYTES

    def cook(self, refs, n_refs):
        """references cooked once per batch (capmi_ciderd_cook_refs): uint8 [B*max_refs, COOKED_BYTES] on the device"""
        B, max_refs, w = refs.shape
        # (slots >= n_refs[image] are never read by the scoring kernel, valid slots are written whole: no zero fill)
        cooked = torch.empty(B * max_refs, self.COOKED_BYTES, dtype=torch.uint8, device=refs.device)
        check(lib.capmi_ciderd_cook_refs(ptr(refs), ptr(n_refs), B, max_refs, w, ptr(self.keys), ptr(self.vals), self.cap,
                                         self.log_ref_len, ptr(cooked), stream_ptr()), 'capmi_ciderd_cook_refs')
        return cooked

    def score(self, hyp, hyp_img, refs, n_refs, cooked=None):
        """hyp int64 [H,L] (device), hyp_img int32 [H] -> float64 [H] CIDEr-D, no host sync."""
        assert hyp.dtype == torch.long and hyp.is_contiguous() and hyp.is_cuda
        H, L = hyp.shape
        scores = torch.empty(H, dtype=torch.float64, device=hyp.device)
        if cooked is not None:
            check(lib.capmi_ciderd_score_cooked(ptr(hyp), H, L, ptr(hyp_img), ptr(cooked), ptr(n_refs), refs.shape[1],
                                                ptr(self.keys), ptr(self.vals), self.cap, self.log_ref_len, ptr(scores),
                                                stream_ptr()), 'capmi_ciderd_score_cooked')
            return scores
        check(lib.capmi_ciderd_score(ptr(hyp), H, L, ptr(hyp_img), ptr(refs), ptr(n_refs), refs.shape[1], refs.shape[2],
                                     ptr(self.keys), ptr(self.vals), self.cap, self.log_ref_len, ptr(scores),
                                     stream_ptr()), 'capmi_ciderd_score')
        return scores

    def self_critical_reward(self, greedy, sampled, refs, n_refs, n, hyp_all=None, cooked=None):
        """rewards.py:41-81 on device: scores of N sampled + B greedy rows, advantage [N] float32.
        hyp_all: the [N+B, L] tensor holding `sampled` then `greedy` already side by side (the fused SCST rollout writes them
        that way): scored in place, no concatenation."""
        N = sampled.shape[0]
        B = greedy.shape[0]
        hyp = hyp_all if hyp_all is not None else torch.cat([sampled, greedy], 0).contiguous()
        key = (N, B, n, str(hyp.device))
        img = self._img_cache.get(key) if hasattr(self, '_img_cache') else None
        if img is None:
            if not hasattr(self, '_img_cache'):
                self._img_cache = {}
            img = torch.cat([torch.arange(N, device=hyp.device) // n, torch.arange(B, device=hyp.device)]).to(torch.int32)
            self._img_cache[key] = img
        scores = self.score(hyp, img, refs, n_refs, cooked)
        buf = torch.empty(N + 1, dtype=torch.float32, device=hyp.device)      # [advantage of the N rows | their mean]
        reward = buf[:N]
        check(lib.capmi_scst_advantage_mean(ptr(scores), N, n, ptr(buf), buf.data_ptr() + 4 * N, stream_ptr()),
              'capmi_scst_advantage_mean')
        reward._capmi_mean = buf[N]             # 0-dim view: LossWrapper's out['reward'] without an ATen mean launch
        return reward, scores
