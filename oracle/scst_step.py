"""CPU port of one full SCST training iteration of the reference (UpDown), used as

* the checker for the HIP path's end-to-end step (tests), and
* the ``cpu_baseline`` ("port") leg of ``bench.py``: the reference's algorithm on the host cores.

TEST INFRASTRUCTURE (see oracle/__init__.py).  It follows the reference's control flow op for op:
``LossWrapper.forward`` SC branch (captioning/modules/loss_wrapper.py:56-73): greedy rollout in eval
mode under no_grad -> sampled rollout in train mode (dropout p, graph kept, DENSE [N,L,V1] log-prob
buffer filled by slice assignment like AttModel.py:347) -> CIDEr-D reward on the host
(rewards.py:41-81) -> RewardCriterion (losses.py:18-37) -> backward -> clip_grad_value_ (train.py:194)
-> Adam (misc.py:125-126).  Sampling uses torch's Categorical as the reference does unless noise is
injected.
"""
from __future__ import annotations

import time
from typing import Dict, Optional

import numpy as np
import torch

from . import att_lstm as O
from . import ciderd as C


class ScstOracle:
    def __init__(self, P: Dict[str, torch.Tensor], scorer: C.CiderD, *, drop_prob=0.5, lr=5e-4, clip=0.1,
                 sample_n=5, max_len=20):
        self.P = {k: v.clone().requires_grad_(True) for k, v in P.items()}
        self.opt = torch.optim.Adam(list(self.P.values()), lr=lr, betas=(0.9, 0.999), eps=1e-8)
        self.scorer, self.p, self.clip, self.n, self.L = scorer, drop_prob, clip, sample_n, max_len

    def step(self, fc, att, gts, att_masks=None, *, drops: Optional[O.Drops] = None, gumbel=None, gen=None):
        P, n, L = self.P, self.n, self.L
        B = fc.shape[0]
        N = B * n
        with torch.no_grad():                                           # loss_wrapper.py:57-62
            greedy, _ = O.rollout(P, fc, att, att_masks, method='greedy', max_len=L)
        if drops is None and self.p > 0:                                 # loss_wrapper.py:63: train mode
            gen = gen or torch.Generator().manual_seed(int(torch.randint(0, 2 ** 31, (1,))))
            K = att.shape[1]
            R, E = P['logit.weight'].shape[1], P['embed.0.weight'].shape[1]
            drops = O.make_drops(self.p, B, K, N, L, E, R, gen)
        if gumbel is None:
            u = torch.rand(L, N, P['logit.weight'].shape[0]).clamp_min(1e-20)
            gumbel = -torch.log(-torch.log(u))
        seq, logp = O.rollout(P, fc, att, att_masks, method='sample', sample_n=n, max_len=L, drops=drops or O.Drops(),
                              gumbel=gumbel)                             # loss_wrapper.py:64-68
        rewards, scores = C.self_critical_reward(self.scorer, greedy.numpy(), gts, seq.numpy())   # :70
        reward = torch.from_numpy(rewards).to(logp)                      # :71
        loss = O.reward_criterion(logp, seq, reward)                     # :72
        self.opt.zero_grad()
        loss.backward()
        if self.clip > 0:
            torch.nn.utils.clip_grad_value_(list(P.values()), self.clip)
        self.opt.step()
        return dict(loss=float(loss), reward=float(reward[:, 0].mean()), seq=seq, greedy=greedy, scores=scores)


def time_iterations(oracle: ScstOracle, fc, att, gts, iters=3, warmup=1):
    for _ in range(warmup):
        oracle.step(fc, att, gts)
    t = []
    for _ in range(iters):
        t0 = time.perf_counter()
        oracle.step(fc, att, gts)
        t.append(time.perf_counter() - t0)
    return float(np.median(t))
