"""Criteria of the hot path (reference captioning/modules/losses.py).  They consume the dense
log-prob tensor the model API returns; the arithmetic is a gather + masked mean over [N,L] values
(K14/K15 of SURVEY.md 2.3) -- device tensor ops on the same HIP stream, no host sync.  The gather goes
through ``sparse_logp.select_logp``: when the tensor comes from a capmi rollout its gradient travels back
as [N,L] values + token ids (``capmi_logsoftmax_bwd_sparse``), never as a dense [N,L,V1] tensor."""
import math

import torch
import torch.nn as nn

from ..utils.rewards import get_scores
from imagecaptioning.pytorch_amd.sparse_logp import select_logp, sum_logp, fused_reward_criterion


def _shifted_mask(seq, like):
    """position 0 on, then (seq>0) shifted right: the EOS step still counts (losses.py:28-29)."""
    m = (seq > 0).to(like)
    return torch.cat([m.new_ones(m.size(0), 1), m[:, :-1]], 1)


class RewardCriterion(nn.Module):
    """losses.py:18-37."""

    def forward(self, input, seq, reward, reduction='mean'):
        # straight from the rollout's saved selected log-probs when `input` / `seq` are its untouched outputs: one launch
        fused = fused_reward_criterion(input, seq, reward, per_row=(reduction == 'none'))
        if fused is not None:
            return fused
        sel = select_logp(input, seq)
        mask = _shifted_mask(seq, sel)
        out = -sel * reward.to(sel) * mask
        if reduction == 'none':
            return out.sum(1) / mask.sum(1)
        return out.sum() / mask.sum()


class LanguageModelCriterion(nn.Module):
    """losses.py:204-224."""

    def forward(self, input, target, mask, reduction='mean'):
        if target.ndim == 3:
            target = target.reshape(-1, target.shape[2])
            mask = mask.reshape(-1, mask.shape[2])
        T = input.size(1)
        target = target[:, :T]
        mask = mask[:, :T].to(input)
        out = -select_logp(input, target) * mask
        if reduction == 'none':
            return out.sum(1) / mask.sum(1)
        return out.sum() / mask.sum()


class LabelSmoothing(nn.Module):
    """losses.py:227-265: KL(true_dist || p), off-target mass smoothing/(V1-1)."""

    def __init__(self, size=0, padding_idx=0, smoothing=0.0):
        super().__init__()
        self.confidence = 1.0 - smoothing
        self.smoothing = smoothing

    def forward(self, input, target, mask, reduction='mean'):
        if target.ndim == 3:
            target = target.reshape(-1, target.shape[2])
            mask = mask.reshape(-1, mask.shape[2])
        N, T, V1 = input.shape
        tgt2 = target[:, :T]
        target = tgt2.reshape(-1)
        mask = mask[:, :T].reshape(-1).to(input)
        off = self.smoothing / (V1 - 1)
        # sum_v q log q is a constant of (smoothing, V1); -sum_v q logp = -off*sum(lp) - (conf-off)*lp[target]
        ent = (V1 - 1) * (off * math.log(off) if off > 0 else 0.0) + \
              (self.confidence * math.log(self.confidence) if self.confidence > 0 else 0.0)
        cross = off * sum_logp(input).reshape(-1) + (self.confidence - off) * select_logp(input, tgt2.contiguous()).reshape(-1)
        out = (ent - cross) * mask
        if reduction == 'none':
            return out.view(N, T).sum(1) / mask.view(N, T).sum(1)
        return out.sum() / mask.sum()


class StructureLosses(nn.Module):
    """losses.py:40-202.  The ``structure_loss_type``s whose input is log-probabilities -- 'new_self_critical' (168-187, the one
    the *_nsc BASELINE configs use), 'seqnll' (81-88), 'risk' (89-103), 'softmax_margin' (147-155), 'best_of_n' (189-199) -- and, r4,
    the margin types that take RAW LOGITS -- 'max_margin' (105-114), 'multi_margin' (128-137), 'real_softmax_margin' (157-166):
    sampled with output_logsoftmax=0 (loss_wrapper.py:31-37), served by rollouts that store the logits (capmi.h CAPMI_SELECT_RAW).
    The optional ``entropy_reward_weight`` (66-69) is supported.  All of them only read the entries of the sampled tokens, so the
    gradient stays sparse (``select_logp``).  The self-CIDEr reward is not implemented."""

    LOGPROB_TYPES = ('new_self_critical', 'seqnll', 'risk', 'softmax_margin', 'best_of_n')
    LOGIT_TYPES = ('max_margin', 'multi_margin', 'real_softmax_margin')

    def __init__(self, opt):
        super().__init__()
        self.opt = opt
        self.loss_type = opt.structure_loss_type

    def forward(self, input, seq, data_gts, reduction='mean'):
        if self.loss_type not in self.LOGPROB_TYPES + self.LOGIT_TYPES:
            raise NotImplementedError('structure_loss_type %r: implemented are %s'
                                      % (self.loss_type, ', '.join(self.LOGPROB_TYPES + self.LOGIT_TYPES)))
        if getattr(self.opt, 'self_cider_reward_weight', 0) > 0:
            raise NotImplementedError('self_cider_reward_weight (get_self_cider_scores, rewards.py:116-137) is out of scope')
        out = {}
        N = input.size(0)
        n = N // len(data_gts)
        assert n == self.opt.train_sample_n, n
        ew = getattr(self.opt, 'entropy_reward_weight', 0)
        ent = None
        if ew > 0:                      # mean per-token entropy of each sampled sequence, no gradient (:66-69); needs the dense rows
            with torch.no_grad():
                lp = torch.log_softmax(input.detach(), 2)
                ent = -(lp.exp() * lp).sum(2)
        sel = select_logp(input, seq)
        mask = _shifted_mask(seq, sel)
        scores = get_scores(data_gts, seq, self.opt, as_tensor=True)
        scores = (scores if torch.is_tensor(scores) else torch.as_tensor(scores)).to(sel).view(-1, n)
        out['reward'] = scores
        if ent is not None:
            scores = scores + ew * ((ent * mask).sum(1) / mask.sum(1)).view(-1, n)
        lt = self.loss_type
        if lt in ('new_self_critical', 'best_of_n'):
            if lt == 'new_self_critical':           # leave-one-out baseline: the mean score of the image's other samples
                w = scores - (scores.sum(1, keepdim=True) - scores) / (scores.shape[1] - 1)
            else:                                   # supervise only the best-scoring sample(s) of each image
                w = (scores == scores.max(1, keepdim=True)[0]).to(sel)
            output = -sel * mask * w.reshape(-1, 1)
            output = output.sum(1) / mask.sum(1) if reduction == 'none' else output.sum() / mask.sum()
        elif lt in ('max_margin', 'multi_margin'):
            # hinge between every sample and the cheapest one of its image, on the mean (masked) logit of the sampled tokens
            assert reduction == 'mean'
            costs = -scores
            avg = ((sel * mask).sum(1) / mask.sum(1)).view(-1, n)
            c_star, i_star = costs.min(1, keepdim=True)
            hinge = torch.relu(costs - c_star - avg.gather(1, i_star) + avg)
            output = (hinge.max(1)[0] / 2).mean() if lt == 'max_margin' else hinge.mean()
        else:
            costs = -scores
            if lt in ('risk', 'softmax_margin'):     # rescale the costs of each image to [0, 1]
                costs = costs - costs.min(1, keepdim=True)[0]
                costs = costs / costs.max(1, keepdim=True)[0]
            tot = (sel * mask).sum(1)
            if lt == 'risk':                         # expected cost under softmax(exp(sequence log-prob)) over the n samples
                assert reduction == 'mean'
                output = (torch.softmax(tot.view(-1, n).exp(), 1) * costs).sum(1).mean()
            else:                                    # cross-entropy towards the cheapest sample of each image
                avg = (tot / mask.sum(1)).view(-1, n)
                if lt in ('softmax_margin', 'real_softmax_margin'):
                    avg = avg + costs
                output = nn.functional.cross_entropy(avg, costs.min(1)[1], reduction=reduction)
        out['loss'] = output
        return out
