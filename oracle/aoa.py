"""Plain PyTorch fp32 CPU restatement of the reference's AoANet captioner (BASELINE configs[4], configs/aoa.yml:
refine 1, refine_aoa 1, use_ff 0, decoder_type AoA, use_multi_head 2, mean_feats 1, ctx_drop 1).

TEST INFRASTRUCTURE (see oracle/__init__.py).  PINNED by tests/golden/aoa_tiny.npz (outputs of the imported
reference in eval mode: teacher-forced log-probs, XE loss, every parameter gradient, greedy decode).

Restates captioning/models/AoAModel.py over the reference's state_dict keys.
Dropout is injected through ``drop(name, tensor)`` (identity when None).  Hard-coded reference rates:
attention-probability dropout 0.1 (AoAModel.py:18,53), refiner sublayer dropout 0.1 (:119), dropout_aoa (opt).
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F

from .transformer import layer_norm, _d


def dot_attention(q, k, v, mask, h, drop, tag):
    """attention() of TransformerModel.py:152-162 on [N,Tq,D] / [N,Tk,D] tensors with h heads."""
    N, Tq, D = q.shape
    dk = D // h
    qh, kh, vh = (x.view(N, -1, h, dk).transpose(1, 2) for x in (q, k, v))
    s = qh @ kh.transpose(-2, -1) / math.sqrt(dk)
    if mask is not None:
        s = s.masked_fill(mask.view(N, 1, 1, -1) == 0, float('-inf'))
    p = _d(drop, tag + '.attn', F.softmax(s, -1))
    return (p @ vh).transpose(1, 2).contiguous().view(N, Tq, D)


def refiner(P, x, att_masks, h, drop=None, n_layers=6):
    """AoA_Refiner_Core (AoAModel.py:115-126): 6 x [x + Drop(AoA(MHA(LN x)))], then LN.  MultiHeadedDotAttention with
    project_k_v=1, do_aoa=1 (:56-98): q,k,v = linears[0..2](y); AoA: GLU(Linear(cat[att, y]))."""
    for i in range(n_layers):
        pre = 'refiner.layers.%d' % i
        y = layer_norm(P, pre + '.sublayer.0.norm', x)
        q = y @ P[pre + '.self_attn.linears.0.weight'].t() + P[pre + '.self_attn.linears.0.bias']
        k = y @ P[pre + '.self_attn.linears.1.weight'].t() + P[pre + '.self_attn.linears.1.bias']
        v = y @ P[pre + '.self_attn.linears.2.weight'].t() + P[pre + '.self_attn.linears.2.bias']
        a = dot_attention(q, k, v, att_masks, h, drop, 'ref%d' % i)
        z = _d(drop, 'ref%d.aoa' % i, torch.cat([a, y], -1))
        g = F.glu(z @ P[pre + '.self_attn.aoa_layer.0.weight'].t() + P[pre + '.self_attn.aoa_layer.0.bias'], -1)
        x = x + _d(drop, 'ref%d.res' % i, g)
    return layer_norm(P, 'refiner.norm', x)


def prepare(P, att_feats, att_masks, h, drop=None):
    """AoAModel._prepare_feature (AoAModel.py:203-226)."""
    if att_masks is not None:
        ml = int(att_masks.long().sum(1).max())
        att_feats, att_masks = att_feats[:, :ml], att_masks[:, :ml]
    x = _d(drop, 'att_embed', F.relu(att_feats @ P['att_embed.0.weight'].t() + P['att_embed.0.bias']))
    if att_masks is not None:
        x = x * att_masks.unsqueeze(-1)
    x = refiner(P, x, att_masks, h, drop)
    if att_masks is None:
        mean = x.mean(1)
    else:
        mean = (x * att_masks.unsqueeze(-1)).sum(1) / att_masks.unsqueeze(-1).sum(1)
    p_att = x @ P['ctx2att.weight'].t() + P['ctx2att.bias']          # [B,K,2R]: value | key  (AoAModel.py:168)
    return mean, x, p_att, att_masks


def step(P, it, mean, p_att, att_masks, state, h, drop=None, t=0, want_logsoftmax=True):
    """get_logprobs_state (AttModel.py:166-176) with AoA_Decoder_Core.forward (AoAModel.py:163-186)."""
    hs, cs = state
    R = mean.shape[1]
    xt = _d(drop, 'xt%d' % t, F.relu(P['embed.0.weight'][it]))
    x1 = torch.cat([xt, mean + _d(drop, 'ctx%d' % t, hs[1])], 1)
    gates = x1 @ P['core.att_lstm.weight_ih'].t() + P['core.att_lstm.bias_ih'] + hs[0] @ P['core.att_lstm.weight_hh'].t() + \
        P['core.att_lstm.bias_hh']
    i, f, g, o = gates.chunk(4, 1)
    c_att = torch.sigmoid(f) * cs[0] + torch.sigmoid(i) * torch.tanh(g)
    h_att = torch.sigmoid(o) * torch.tanh(c_att)
    qn = layer_norm(P, 'core.attention.norm', h_att)
    q = qn @ P['core.attention.linears.0.weight'].t() + P['core.attention.linears.0.bias']
    att = dot_attention(q.unsqueeze(1), p_att[:, :, R:], p_att[:, :, :R], att_masks, h, drop, 'dec%d' % t).squeeze(1)
    out = F.glu(torch.cat([att, h_att], 1) @ P['core.att2ctx.0.weight'].t() + P['core.att2ctx.0.bias'], -1)
    state = (torch.stack([h_att, out]), torch.stack([c_att, cs[1]]))
    logits = _d(drop, 'out%d' % t, out) @ P['logit.weight'].t() + P['logit.bias']
    return (F.log_softmax(logits, 1) if want_logsoftmax else logits), state       # output_logsoftmax = 0: AttModel.py:171-175


def forward_teacher(P, att_feats, seq, att_masks, h, drop=None, want_logsoftmax=True):
    """AttModel._forward (AttModel.py:126-164) for AoAModel."""
    if seq.ndim == 3:
        seq = seq.reshape(-1, seq.shape[2])
    B = att_feats.shape[0]
    N, T = seq.shape
    n = N // B
    mean, att, p_att, masks = prepare(P, att_feats, att_masks, h, drop)
    if n > 1:
        mean, p_att = mean.repeat_interleave(n, 0), p_att.repeat_interleave(n, 0)
        masks = None if masks is None else masks.repeat_interleave(n, 0)
    R = mean.shape[1]
    state = (mean.new_zeros(2, N, R), mean.new_zeros(2, N, R))
    out = att_feats.new_zeros(N, T, P['logit.weight'].shape[0])
    for t in range(T):
        if t >= 1 and int(seq[:, t].sum()) == 0:
            break
        logp, state = step(P, seq[:, t], mean, p_att, masks, state, h, drop, t, want_logsoftmax)
        out[:, t] = logp
    return out


def greedy(P, att_feats, att_masks, h, max_len):
    B = att_feats.shape[0]
    mean, att, p_att, masks = prepare(P, att_feats, att_masks, h)
    R = mean.shape[1]
    state = (mean.new_zeros(2, B, R), mean.new_zeros(2, B, R))
    V1 = P['logit.weight'].shape[0]
    seq = torch.zeros(B, max_len, dtype=torch.long)
    slp = att_feats.new_zeros(B, max_len, V1)
    it = torch.zeros(B, dtype=torch.long)
    unf = None
    for t in range(max_len):
        logp, state = step(P, it, mean, p_att, masks, state, h, None, t)
        it = torch.max(logp, 1)[1]
        if t == 0:
            unf = it != 0
        else:
            it = it * unf.long()
            logp = logp * unf.unsqueeze(1).to(logp)
            unf = unf & (it != 0)
        seq[:, t] = it
        slp[:, t] = logp
        if int(unf.sum()) == 0:
            break
    return seq, slp


def sample(P, att_feats, att_masks, h, max_len, n=1, gen=None, drop=None, forced=None):
    """AttModel._sample (AttModel.py:258-352) for AoAModel with sample_method='sample', sample_n=n: n rows per image drawn from
    torch.distributions-style multinomial of exp(logprobs) (CaptionModel.sample_next_word, :388-395), WITH the autograd graph
    (the new_self_critical step differentiates the rollout it sampled, loss_wrapper.py:25-48).  Returns seq [B*n, L] and the
    dense log-probs [B*n, L, V1] filled like AttModel.py:347.  forced [B*n, L]: use these tokens instead of drawing (teacher-forcing a
    sequence the reference sampled: the parity protocol for stochastic decoding, as att_lstm.rollout)."""
    B = att_feats.shape[0]
    mean, att, p_att, masks = prepare(P, att_feats, att_masks, h, drop)
    if n > 1:
        mean, p_att = mean.repeat_interleave(n, 0), p_att.repeat_interleave(n, 0)
        masks = None if masks is None else masks.repeat_interleave(n, 0)
    N, R = B * n, mean.shape[1]
    V1 = P['logit.weight'].shape[0]
    state = (mean.new_zeros(2, N, R), mean.new_zeros(2, N, R))
    seq = torch.zeros(N, max_len, dtype=torch.long)
    slp = mean.new_zeros(N, max_len, V1)
    it = torch.zeros(N, dtype=torch.long)
    unf = None
    for t in range(max_len):
        logp, state = step(P, it, mean, p_att, masks, state, h, drop, t)
        with torch.no_grad():
            it = forced[:, t].clone() if forced is not None else torch.multinomial(logp.detach().exp(), 1, generator=gen).squeeze(1)
        if t == 0:
            unf = it != 0
        else:
            it = it * unf.long()
            logp = logp * unf.unsqueeze(1).to(logp)
            unf = unf & (it != 0)
        seq[:, t] = it
        slp[:, t] = logp
        if int(unf.sum()) == 0:
            break
    return seq, slp
