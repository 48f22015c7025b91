"""Sparse gradients for the dense log-prob tensor the model API returns.

The reference's criteria read ``seqLogprobs [N,L,V1]`` only through ``input.gather(2, target)`` (losses.py:24, :81, :213) and
-- LabelSmoothing -- through the row sum (:258-262); autograd then materialises a dense [N,L,V1] gradient (zero fill + scatter:
45 MB twice at bs10 x 5, 255 MB twice at bs64) that the rollout backward reads once.  Here the dense tensor still exists as the
API output, but it carries a *sink*: ``select_logp`` / ``sum_logp`` gather through autograd Functions whose backward hands the
[N,L] gradient, the token ids and the row-sum gradient to the sink, and leave the dense tensor's gradient undefined (the
rollout Functions run with ``set_materialize_grads(False)``, so no zero tensor is created for it).  The rollout's backward (``capmi_logsoftmax_bwd_sparse``) builds d(logits) from the sink directly.  Anything else that
differentiates through the dense tensor still works: its real dense gradient is added on top.
"""
import ctypes as C
import os

import torch

from . import _lib
from ._lib import ptr


class LogpSink:
    """collects the sparse loss gradient of ONE rollout between the criterion's backward and the rollout's backward"""

    def __init__(self):
        self.tok = self.g_sel = self.g_sum = self.scale = None
        self.sel_taken = self.sum_taken = False
        self.sel = self.seq = None        # set by rollouts that save the selected log-probs / tokens (fused criteria)

    def take(self):
        out = (self.tok, self.g_sel, self.g_sum, self.scale)
        self.tok = self.g_sel = self.g_sum = self.scale = None
        return None if out[1] is None and out[2] is None else out


_DENSE = os.environ.get('CAPMI_DENSE_LOSS_GRAD', '0') == '1'      # A/B switch: the reference's dense autograd route


def attach(logp, sink):
    if not _DENSE:
        logp._capmi_sink = sink
    return logp


class _SelectLogp(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logp, idx, sink):
        ctx.sink = sink
        ctx.save_for_backward(idx)
        return logp.gather(2, idx.unsqueeze(2)).squeeze(2)

    @staticmethod
    def backward(ctx, g):
        (idx,) = ctx.saved_tensors
        sink = ctx.sink
        sink.tok = idx.contiguous()
        sink.g_sel = g.contiguous().float() if sink.g_sel is None else sink.g_sel + g
        return None, None, None          # undefined gradient for the dense tensor: the sink carries it


class _SumLogp(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logp, sink):
        ctx.sink = sink
        return logp.sum(2)

    @staticmethod
    def backward(ctx, g):
        sink = ctx.sink
        sink.g_sum = g.contiguous().float() if sink.g_sum is None else sink.g_sum + g
        return None, None


def attach_masked(out, parent, row_mask):
    """`out` = parent * row_mask[..., None] (rows zeroed after a caption finished, AttModel.py:343-347): gathers on `out` are
    served from `parent` (which carries the sink) and masked afterwards, so the sparse path survives the masking."""
    out._capmi_masked = (parent, row_mask)
    return out


def attach_rows(out, parent):
    """`out` = parent[:out.shape[0]] (the sampled rows of the fused SCST rollout, whose greedy-baseline rows ride in the same
    tensor): gathers on `out` are served from `parent` with the remaining rows padded (token 0, zero gradient)."""
    out._capmi_rows = parent
    return out


def select_logp(logp, idx):
    """logp[r, t, idx[r, t]] for idx [N, T] int64 (T == logp.shape[1]); sparse gradient when the tensor carries a sink."""
    parent = getattr(logp, '_capmi_rows', None)
    if parent is not None and idx.shape == logp.shape[:2]:
        pad = idx.new_zeros(parent.shape[0] - idx.shape[0], idx.shape[1])
        return select_logp(parent, torch.cat([idx, pad], 0))[:idx.shape[0]]
    masked = getattr(logp, '_capmi_masked', None)
    if masked is not None and idx.shape == logp.shape[:2]:
        parent, row_mask = masked
        return select_logp(parent, idx) * row_mask.to(parent.dtype)
    sink = getattr(logp, '_capmi_sink', None)
    if sink is None or not logp.requires_grad or sink.sel_taken or idx.shape != logp.shape[:2]:
        return logp.gather(2, idx.unsqueeze(2)).squeeze(2)
    sink.sel_taken = True              # one token per (row, step): a second, different gather goes the dense way
    return _SelectLogp.apply(logp, idx, sink)


def sum_logp(logp):
    """sum_v logp[r, t, v]; sparse gradient when the tensor carries a sink."""
    sink = getattr(logp, '_capmi_sink', None)
    if sink is None or not logp.requires_grad or sink.sum_taken:
        return logp.sum(2)
    sink.sum_taken = True
    return _SumLogp.apply(logp, sink)


class _FusedReward(torch.autograd.Function):
    """RewardCriterion on the selected log-probs the rollout saved (capmi_reward_criterion): forward = ONE launch, backward
    = nothing but handing (coefficients, tokens, upstream scalar) to the sink."""

    @staticmethod
    def forward(ctx, logp, sink, reward, n_used, per_row):
        from . import ops
        loss, gcoef = ops.reward_criterion(sink.sel, sink.seq, reward, n_used, per_row)
        ctx.sink, ctx.gcoef, ctx.per_row, ctx.n_used = sink, gcoef, per_row, n_used
        return loss if per_row else loss.squeeze(0)

    @staticmethod
    def backward(ctx, g):
        sink = ctx.sink
        sink.tok = sink.seq
        if ctx.per_row:                               # one upstream value per row
            gc = ctx.gcoef.clone()
            gc[:ctx.n_used] *= g.reshape(-1, 1).to(gc)
            sink.g_sel = gc
        else:                                         # scalar loss: the upstream gradient stays a device scalar
            sink.g_sel = ctx.gcoef
            sink.scale = g.reshape(1).float().contiguous()
        return None, None, None, None, None


def fused_reward_criterion(logp, seq, reward, per_row=False):
    """RewardCriterion(logp, seq, reward) through the rollout's saved selected log-probs, or None when `logp` / `seq` are not
    the untouched outputs of one capmi rollout (then the caller takes the generic route)."""
    if _DENSE or not logp.requires_grad:
        return None
    parent = getattr(logp, '_capmi_rows', None)
    full = parent if parent is not None else logp
    sink = getattr(full, '_capmi_sink', None)
    if sink is None or sink.sel is None or sink.sel_taken or sink.seq is None:
        return None
    n_used = logp.shape[0]
    if seq.data_ptr() != sink.seq.data_ptr() or seq.shape != (n_used, sink.seq.shape[1]) or not seq.is_contiguous():
        return None                                   # not the rollout's own tokens
    if reward.shape[0] != n_used or reward.ndim not in (1, 2) or not reward.is_cuda:
        return None
    if per_row and n_used > 2048:
        return None                                   # capmi_reward_criterion keeps the per-row losses in one workgroup's LDS
    sink.sel_taken = True
    return _FusedReward.apply(full, sink, reward.float(), n_used, per_row)


def split_grad(g_logp, sink, like=None):
    """What a rollout backward receives -> (dense gradient or None, SparseLogpGrad struct or None, keep-alive tuple).
    `like`: the saved dense log-probs, for the (rare) case that no gradient at all arrived."""
    dense = None if g_logp is None else g_logp.contiguous()
    sp = sink.take() if sink is not None else None
    if sp is None:
        if dense is None and like is not None:
            dense = torch.zeros_like(like)
        return dense, None, ()
    tok, g_sel, g_sum, scale = sp
    s = _lib.SparseLogpGrad()
    s.g_sel, s.g_sum, s.scale = ptr(g_sel), ptr(g_sum), ptr(scale)
    if g_sel is not None:
        assert tok.dtype == torch.long and tok.is_contiguous() and tok.shape == g_sel.shape
        s.tok, s.tok_ld = ptr(tok), tok.shape[1]
    return dense, s, (tok, g_sel, g_sum, scale, s)


def byref_or_none(s):
    return C.pointer(s) if s is not None else None
