"""CPU restatement of the reference's decode-time options, over a generic one-step decoder callback.

TEST INFRASTRUCTURE (see ``oracle/__init__.py``): checker only, never imported by the product package.

    constrained_sample   AttModel._sample          AttModel.py:258-352   decoding_constraint / remove_bad_endings /
                                                                         block_trigrams, greedy or Gumbel-max sampling
    diverse_sample       AttModel._diverse_sample  AttModel.py:354-447
    beam_search          AttModel._sample_beam + CaptionModel.beam_search
                                                   AttModel.py:218-256, CaptionModel.py:35-209 (diverse groups, both
                                                   constraints, suppress_UNK, temperature, length_penalty)

Parity status: PINNED for everything but one combination -- ``tests/golden/updown_tiny_opts.npz`` holds the outputs of
the real reference (``make_golden.py opts``) for each function and option; ``tests/test_oracle_golden.py`` replays them.
UNPINNED: block_trigrams inside diverse_sample (the reference calls ``.cuda()`` there, AttModel.py:424, so it cannot run
in the CPU-only container); that branch is the same arithmetic as the pinned block_trigrams of constrained_sample.

``step(it [N] int64, state) -> (logp [N,V1] log-softmax, state)`` with ``state`` a tuple of ``[layers, N, R]`` tensors is
``get_logprobs_state`` with the (already repeated) features bound by the caller.  All citations are into /root/reference.
"""
from __future__ import annotations

from typing import Callable, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
NEG_INF = float('-inf')


def _constrain(logp: Tensor, seq: Tensor, t: int, bad: Sequence[int], no_repeat: bool, no_bad_ending: bool):
    """AttModel.py:293-303 (also CaptionModel.py:152-155): additive -inf masks built from the previous token."""
    if t == 0:
        return logp
    prev = seq[:, t - 1]
    if no_repeat:
        mask = torch.zeros_like(logp)
        mask[torch.arange(logp.shape[0]), prev] = NEG_INF
        logp = logp + mask
    if no_bad_ending:
        mask = torch.zeros_like(logp)
        hit = torch.tensor([int(p) in set(bad) for p in prev.tolist()])
        mask[hit, 0] = NEG_INF
        logp = logp + mask
    return logp


def _trigram_penalty(logp: Tensor, seq: Tensor, t: int, rows: int):
    """AttModel.py:305-330.  For row i < rows: every earlier position e (2 <= e < t) whose two predecessors equal the last
    two tokens adds one to mask[i, seq[i, e]]; logprobs += (mask * -0.693) * 2.0.  (`rows` is the IMAGE count: the loops at
    :310 and :322 run over range(batch_size) even when sample_n > 1.)"""
    if t < 3:
        return logp
    mask = torch.zeros_like(logp)
    for i in range(rows):
        a, b = int(seq[i, t - 2]), int(seq[i, t - 1])
        for e in range(2, t):
            if int(seq[i, e - 2]) == a and int(seq[i, e - 1]) == b:
                mask[i, int(seq[i, e])] += 1
    return logp + (mask * -0.693 * 2.0)


def _choose(logp: Tensor, method: str, temperature: float, gumbel: Optional[Tensor]):
    """CaptionModel.sample_next_word (CaptionModel.py:370-407) with injected Gumbel noise for 'sample'.
    Returns (token, log-prob of the token in the tensor the sampler gathered from)."""
    if method == 'greedy':
        val, it = torch.max(logp, 1)
        return it, val
    scaled = logp / temperature
    it = torch.max(scaled + gumbel, 1)[1]
    return it, scaled.gather(1, it.unsqueeze(1)).squeeze(1)


def constrained_sample(step: Callable, state, N: int, B: int, V1: int, max_len: int, *, method='greedy', temperature=1.0,
                       gumbel: Optional[Tensor] = None, bad_endings: Sequence[int] = (), decoding_constraint=0,
                       remove_bad_endings=0, block_trigrams=0):
    seq = torch.zeros(N, max_len, dtype=torch.long)
    seq_logp = torch.zeros(N, max_len, V1)
    it = torch.zeros(N, dtype=torch.long)
    unfinished = None
    for t in range(max_len):
        logp, state = step(it, state)
        logp = _constrain(logp, seq, t, bad_endings, bool(decoding_constraint), bool(remove_bad_endings))
        if block_trigrams:
            logp = _trigram_penalty(logp, seq, t, B)
        it, _ = _choose(logp, method, temperature, None if gumbel is None else gumbel[t])
        if t == 0:
            unfinished = it != 0
        else:
            it = it * unfinished.long()
            logp = logp * unfinished.unsqueeze(1).to(logp)          # -inf * 0 = NaN, as in the reference (:345)
            unfinished = unfinished & (it != 0)
        seq[:, t] = it
        seq_logp[:, t] = logp
        if int(unfinished.sum()) == 0:
            break
    return seq, seq_logp


def diverse_sample(step: Callable, make_state: Callable, B: int, V1: int, max_len: int, group_size: int, *, method='greedy',
                   temperature=1.0, diversity_lambda=0.5, gumbel: Optional[Tensor] = None, bad_endings: Sequence[int] = (),
                   decoding_constraint=0, remove_bad_endings=0, block_trigrams=0):
    """gumbel [L, G, B, V1].  Returns (seq [B*G, L], chosen log-probs [B*G, L])."""
    G = group_size
    seqs = [torch.zeros(B, max_len, dtype=torch.long) for _ in range(G)]
    slps = [torch.zeros(B, max_len) for _ in range(G)]
    states = [make_state() for _ in range(G)]
    for tt in range(max_len + G):
        for g in range(G):
            t = tt - g
            if t < 0 or t > max_len - 1:
                continue
            seq = seqs[g]
            it = torch.zeros(B, dtype=torch.long) if t == 0 else seq[:, t - 1]
            logp, states[g] = step(it, states[g])
            logp = F.log_softmax(logp / temperature, dim=-1)                     # :389
            for pg in range(g):                                                  # :392-397: columns, for EVERY row
                cols = seqs[pg][:, t]
                logp[:, cols] = logp[:, cols] - diversity_lambda
            logp = _constrain(logp, seq, t, bad_endings, bool(decoding_constraint), bool(remove_bad_endings))
            if block_trigrams:
                logp = _trigram_penalty(logp, seq, t, B)
            it, val = _choose(logp, method, 1.0, None if gumbel is None else gumbel[t, g])      # temperature 1 (:434)
            if t > 0:
                unfinished = seq[:, t - 1] != 0                                  # :443
                it = it * unfinished.long()
            seq[:, t] = it
            slps[g][:, t] = val                                                  # not masked (:447)
    return torch.stack(seqs, 1).reshape(B * G, -1), torch.stack(slps, 1).reshape(B * G, -1)


def _length_penalty(cfg: str):
    """captioning/utils/misc.py:133-157."""
    if cfg == '':
        return lambda length, p: p
    kind, alpha = cfg.split('_')
    alpha = float(alpha)
    if kind == 'wu':
        return lambda length, p: p / (((5 + length) ** alpha) / ((5 + 1) ** alpha))
    return lambda length, p: p / length


def beam_search(step: Callable, state, init_logp: Tensor, V1: int, max_len: int, *, beam_size=10, group_size=1,
                diversity_lambda=0.5, temperature=1.0, decoding_constraint=0, remove_bad_endings=0, bad_endings=(),
                unk_col: Optional[int] = None, length_penalty='', sample_n=1):
    """init_logp [B,V1] / state: after feeding BOS to B rows (AttModel.py:235-239).  step() is called on B*bdash rows per
    group (features repeated by the caller: row r belongs to image r // bdash).
    Returns (seq [B*sample_n, L], seqLogprobs [.., L, V1], done_beams)."""
    B = init_logp.shape[0]
    G = group_size
    bd = beam_size // G
    penalty = _length_penalty(length_penalty)
    seq_tab = [torch.zeros(B, bd, 0, dtype=torch.long) for _ in range(G)]
    slp_tab = [torch.zeros(B, bd, 0, V1) for _ in range(G)]
    sum_tab = [torch.zeros(B, bd) for _ in range(G)]
    state_tab = [tuple(s.clone() for s in state) for _ in range(G)]
    logp_tab = [init_logp.clone() for _ in range(G)]
    done: List[List[List[dict]]] = [[[] for _ in range(G)] for _ in range(B)]
    for t in range(max_len + G - 1):
        for g in range(G):
            lt = t - g
            if lt < 0 or lt > max_len - 1:
                continue
            logp = logp_tab[g]
            if lt > 0:
                prev = seq_tab[g][:, :, lt - 1].reshape(-1)
                if decoding_constraint:
                    logp[torch.arange(logp.shape[0]), prev] = NEG_INF               # CaptionModel.py:152-153 (scatter_)
                if remove_bad_endings:
                    hit = torch.tensor([int(p) in set(bad_endings) for p in prev.tolist()])
                    logp[hit, 0] = NEG_INF                                          # :154-155
            if unk_col is not None:
                logp[:, unk_col] = logp[:, unk_col] - 1000                          # :157-162
            unaug = logp.clone()
            if g > 0:                                                               # add_diversity, :38-57
                change = torch.zeros(B, V1)
                for pg in range(g):
                    for j in range(bd):
                        change.scatter_add_(1, seq_tab[pg][:, j, lt].unsqueeze(-1), torch.ones(B, 1))
                pen = change if lt == 0 else change.repeat_interleave(bd, 0)
                logp = logp - pen * diversity_lambda
            # beam_step, :61-112
            rows = logp.reshape(B, -1, V1)
            sums = sum_tab[g][:, :1] if lt == 0 else sum_tab[g]
            cand = (sums.unsqueeze(-1) + rows).reshape(B, -1)
            ys, ix = torch.sort(cand, -1, True)
            ys, ix = ys[:, :bd], ix[:, :bd]
            beam_ix, tok = ix // V1, ix % V1
            state_ix = (beam_ix + torch.arange(B).unsqueeze(-1) * rows.shape[1]).reshape(-1)
            if lt > 0:
                seq_tab[g] = seq_tab[g].gather(1, beam_ix.unsqueeze(-1).expand_as(seq_tab[g]))
                slp_tab[g] = slp_tab[g].gather(1, beam_ix.unsqueeze(-1).unsqueeze(-1).expand_as(slp_tab[g]))
            seq_tab[g] = torch.cat([seq_tab[g], tok.unsqueeze(-1)], -1)
            sum_tab[g] = sums.gather(1, beam_ix) + rows.reshape(B, -1).gather(1, ix)
            picked = unaug.reshape(B, -1, V1).gather(1, beam_ix.unsqueeze(-1).expand(-1, -1, V1))
            slp_tab[g] = torch.cat([slp_tab[g], picked.reshape(B, -1, 1, V1)], 2)
            state_tab[g] = tuple(s[:, state_ix] for s in state_tab[g])
            # finished beams, :176-194
            for b in range(B):
                is_end = seq_tab[g][b, :, lt] == 0
                if lt == max_len - 1:
                    is_end = torch.ones_like(is_end)
                for j in range(bd):
                    if is_end[j]:
                        done[b][g].append({'seq': seq_tab[g][b, j].clone(), 'logps': slp_tab[g][b, j].clone(),
                                           'unaug_p': slp_tab[g][b, j].sum().item(),
                                           'p': penalty(lt + 1, sum_tab[g][b, j].item())})
                sum_tab[g][b, is_end] -= 1000
            it = seq_tab[g][:, :, lt].reshape(-1)
            logp_next, state_tab[g] = step(it, state_tab[g])
            logp_tab[g] = F.log_softmax(logp_next / temperature, dim=-1)            # :203-204
    done_beams = [sum([sorted(done[b][g], key=lambda x: -x['p'])[:bd] for g in range(G)], []) for b in range(B)]
    seq = torch.zeros(B * sample_n, max_len, dtype=torch.long)
    seq_logp = torch.zeros(B * sample_n, max_len, V1)
    for k in range(B):
        if sample_n == beam_size:                                                   # AttModel.py:245
            for n in range(sample_n):
                ln = done_beams[k][n]['seq'].shape[0]
                seq[k * sample_n + n, :ln] = done_beams[k][n]['seq']
                seq_logp[k * sample_n + n, :ln] = done_beams[k][n]['logps']
        else:
            ln = done_beams[k][0]['seq'].shape[0]
            seq[k, :ln] = done_beams[k][0]['seq']
            seq_logp[k, :ln] = done_beams[k][0]['logps']
    return seq, seq_logp, done_beams
