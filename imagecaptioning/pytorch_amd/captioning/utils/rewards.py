"""Reward plumbing of SCST (reference captioning/utils/rewards.py), CIDEr-D on the device.

``init_scorer(cached_tokens)`` loads ``data/<cached_tokens>.p`` (scripts/prepro_ngrams.py:79-80) into a
device hash table once.  ``get_self_critical_reward`` keeps the reference signature and return type
(np.ndarray [N,L] float64, rewards.py:41-81) for drop-in callers; ``self_critical_reward_device``
is the sync-free variant our LossWrapper uses (advantage stays in HBM).
The CIDEr-D arithmetic restates the external pyciderevalcap package: PARITY UNPINNED (oracle/ciderd.py).
"""
import os

import numpy as np
import torch

from imagecaptioning.pytorch_amd.ciderd import DeviceCiderD

CiderD_scorer = None


class GtsBatch(list):
    """The reference's ``data['gts']`` (list, per image, of [n_ref, L] integer arrays; dataloader.py:213-214, 256) carrying
    the device image of itself: ``packed = (refs int32 [B,max_refs,w], n_refs int32 [B])``.  The loader / prefetcher builds it
    once per batch (``pack_gts``), so the reward path neither re-packs nor guesses identity from object addresses; a plain
    list is packed on the spot every time (no cache)."""
    packed = None


def pack_gts(data_gts):
    """list of reference arrays -> GtsBatch with its device image (needs init_scorer first)."""
    if isinstance(data_gts, GtsBatch) and data_gts.packed is not None:
        return data_gts
    out = GtsBatch(data_gts)
    out.packed = CiderD_scorer.pack_refs(list(data_gts))
    return out


def init_scorer(cached_tokens, device=None):
    """rewards.py:25-31.  ``cached_tokens``: pickle stem under data/, a path to the pickle or to a converted
    flat image (.npz, tools/convert_df.py), or (df dict, ref_len)."""
    global CiderD_scorer
    if CiderD_scorer is not None:
        return CiderD_scorer
    device = device or torch.device('cuda', torch.cuda.current_device())
    if isinstance(cached_tokens, tuple):
        CiderD_scorer = DeviceCiderD(cached_tokens[0], cached_tokens[1], device)
    else:
        path = cached_tokens if os.path.exists(str(cached_tokens)) else os.path.join('data', cached_tokens + '.p')
        image = path if str(path).endswith('.npz') else os.path.splitext(str(path))[0] + '.capmi.npz'
        if os.path.exists(image):
            CiderD_scorer = DeviceCiderD.from_image(image, device)       # pre-hashed table: two array reads
        else:
            CiderD_scorer = DeviceCiderD.from_pickle(path, device)
    return CiderD_scorer


def reset_scorer():
    global CiderD_scorer
    CiderD_scorer = None


def select_gts(gts, gt_indices):
    """loss_wrapper.py:69 ``gts = [gts[_] for _ in gt_indices.tolist()]``; the packed device image survives when the
    selection is the whole batch in order (always, outside nn.DataParallel scatter)."""
    idx = gt_indices.tolist() if hasattr(gt_indices, 'tolist') else list(gt_indices)
    if isinstance(gts, GtsBatch) and gts.packed is not None and idx == list(range(len(gts))):
        return gts
    return [gts[i] for i in idx]


def _pack(data_gts):
    packed = getattr(data_gts, 'packed', None)
    return packed if packed is not None else CiderD_scorer.pack_refs(list(data_gts))


def self_critical_reward_device(greedy_res, data_gts, gen_result, opt):
    """advantage [N] float32 on device + raw scores [N+B] float64; no host synchronisation."""
    if getattr(opt, 'bleu_reward_weight', 0) > 0:
        raise NotImplementedError('BLEU reward (default weight 0, opts.py:185) is out of scope')
    B = len(data_gts)
    n = gen_result.shape[0] // B
    packed = _pack(data_gts)
    refs, n_refs = packed
    hyp_all = getattr(gen_result, '_capmi_all', None)        # fused SCST rollout: sampled + greedy rows already side by side
    if hyp_all is not None and not (hyp_all.dtype == torch.long and hyp_all.is_contiguous() and
                                    hyp_all.shape[0] == gen_result.shape[0] + greedy_res.shape[0]):
        hyp_all = None
    reward, scores = CiderD_scorer.self_critical_reward(greedy_res.long().contiguous(), gen_result.long().contiguous(),
                                                         refs, n_refs, n, hyp_all=hyp_all,
                                                         cooked=getattr(packed, 'cooked', None))
    w = getattr(opt, 'cider_reward_weight', 1)
    if w != 1:
        reward = reward * w
    return reward, scores


def get_self_critical_reward(greedy_res, data_gts, gen_result, opt):
    """Reference-compatible: np.ndarray [N, L] float64 (rewards.py:41-81)."""
    reward, scores = self_critical_reward_device(greedy_res, data_gts, gen_result, opt)
    N = gen_result.shape[0]
    B = len(data_gts)
    s = scores.cpu().numpy() * getattr(opt, 'cider_reward_weight', 1)
    adv = s[:N].reshape(B, N // B) - s[N:][:, None]
    return np.repeat(adv.reshape(N)[:, None], gen_result.shape[1], 1)


def get_scores(data_gts, gen_result, opt, as_tensor=False):
    """rewards.py:83-114: CIDEr-D of each sampled row (np.ndarray [N], or a device tensor)."""
    B = len(data_gts)
    N = gen_result.shape[0]
    n = N // B
    packed = _pack(data_gts)
    refs, n_refs = packed
    img = (torch.arange(N, device=gen_result.device) // n).to(torch.int32)
    scores = CiderD_scorer.score(gen_result.long().contiguous(), img, refs, n_refs, getattr(packed, 'cooked', None)) * \
        getattr(opt, 'cider_reward_weight', 1)
    return scores if as_tensor else scores.cpu().numpy()
