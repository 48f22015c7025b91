"""Plain PyTorch fp32 CPU restatement of the reference's Transformer captioner (BASELINE configs[3]).

TEST INFRASTRUCTURE (see oracle/__init__.py).  PINNED by tests/golden/transformer_tiny.npz (outputs of the
imported reference: teacher-forced log-probs, XE loss, every parameter gradient, greedy decode).

Restates captioning/models/TransformerModel.py functionally over the reference's state_dict keys
(SURVEY.md Appendix C):  att_embed.0.{weight,bias};  model.encoder.layers.{i}.self_attn.linears.{0-3},
.feed_forward.w_{1,2}, .sublayer.{0,1}.norm.{a_2,b_2};  model.encoder.norm;  model.decoder.layers.{i}.
{self_attn,src_attn}.linears.{0-3}, .feed_forward, .sublayer.{0,1,2}.norm;  model.decoder.norm;
model.tgt_embed.0.lut.weight;  model.tgt_embed.1.pe (buffer);  model.generator.proj.{weight,bias}.
Dropout is injected as pre-scaled masks through ``drop(name, tensor)`` (identity when None).
"""
from __future__ import annotations

import math
from typing import Callable, Dict, Optional

import torch
import torch.nn.functional as F

Params = Dict[str, torch.Tensor]
Drop = Optional[Callable[[str, torch.Tensor], torch.Tensor]]


def _d(drop: Drop, name: str, x):
    return x if drop is None else drop(name, x)


def layer_norm(P: Params, pre: str, x):
    """TransformerModel.LayerNorm.forward, TransformerModel.py:84-87: a*(x-mean)/(std_unbiased+1e-6)+b."""
    mean = x.mean(-1, keepdim=True)
    std = x.std(-1, keepdim=True)
    return P[pre + '.a_2'] * (x - mean) / (std + 1e-6) + P[pre + '.b_2']


def mha(P: Params, pre: str, q_in, k_in, v_in, mask, h: int, drop: Drop, tag: str):
    """MultiHeadedAttention.forward (:176-195) + attention (:152-162).  mask broadcastable to [N,1,Tq,Tk]
    (0 = masked out with -inf)."""
    N, Tq, D = q_in.shape
    dk = D // h

    def proj(i, x):
        y = x @ P['%s.linears.%d.weight' % (pre, i)].t() + P['%s.linears.%d.bias' % (pre, i)]
        return y.view(N, -1, h, dk).transpose(1, 2)

    q, k, v = proj(0, q_in), proj(1, k_in), proj(2, v_in)
    scores = q @ k.transpose(-2, -1) / math.sqrt(dk)
    if mask is not None:
        scores = scores.masked_fill(mask.unsqueeze(1) == 0, float('-inf'))
    p = _d(drop, tag + '.attn', F.softmax(scores, dim=-1))
    x = (p @ v).transpose(1, 2).contiguous().view(N, Tq, D)
    return x @ P[pre + '.linears.3.weight'].t() + P[pre + '.linears.3.bias']


# test probe: when a dict, ffn() records per FFN the hidden units whose ReLU input came within RELU_TIE_MARGIN of zero for some
# token.  ReLU's derivative jumps there: two fp32 implementations whose pre-activations differ by rounding can take different
# sides, and the unit's row of dW1 / db1 then differs by that token's whole term (tests/test_full_size_parity_gpu.py)
RELU_TIES: Optional[Dict[str, torch.Tensor]] = None
RELU_TIE_MARGIN = 1e-4


def ffn(P: Params, pre: str, x, drop: Drop, tag: str):
    """PositionwiseFeedForward (:205-206)."""
    a = x @ P[pre + '.w_1.weight'].t() + P[pre + '.w_1.bias']
    if RELU_TIES is not None:
        RELU_TIES[pre] = (a.detach().abs() < RELU_TIE_MARGIN).reshape(-1, a.shape[-1]).any(0)
    hdn = _d(drop, tag + '.ff', F.relu(a))
    return hdn @ P[pre + '.w_2.weight'].t() + P[pre + '.w_2.bias']


def encode(P: Params, att_feats, att_masks, h: int, n_layers: int, drop: Drop = None, rows_per_image: int = 1):
    """att_embed (TransformerModel.py:280-285) + Encoder (:61-74) with pre-norm residual sublayers (:89-102).

    rows_per_image > 1 is TransformerModel._forward's order (:316-321, 343-345): att_embed (and its Dropout) runs on the B images,
    the embedded regions are THEN repeated seq_per_img times and the ENCODER runs on all B * seq_per_img rows -- in train mode
    every caption row draws its own encoder dropout masks (pinned by tests/golden/train_mode.npz: the reference's attention
    mask of encoder layer 0 is [B * n, h, K, K]).  With dropout off the rows of one image are identical copies.  _sample goes
    through _prepare_feature (:306-311) instead: it encodes the B images and repeats the memory (callers pass 1)."""
    x = _d(drop, 'att_embed', F.relu(att_feats @ P['att_embed.0.weight'].t() + P['att_embed.0.bias']))
    if att_masks is not None:
        x = x * att_masks.unsqueeze(-1).to(x)              # pack_wrapper zero-pads (AttModel.py:44-49)
    if rows_per_image > 1:
        x = x.repeat_interleave(rows_per_image, 0)
        att_masks = None if att_masks is None else att_masks.repeat_interleave(rows_per_image, 0)
    smask = None if att_masks is None else att_masks.unsqueeze(-2)
    for i in range(n_layers):
        pre = 'model.encoder.layers.%d' % i
        y = layer_norm(P, pre + '.sublayer.0.norm', x)
        x = x + _d(drop, 'enc%d.res0' % i, mha(P, pre + '.self_attn', y, y, y, smask, h, drop, 'enc%d' % i))
        y = layer_norm(P, pre + '.sublayer.1.norm', x)
        x = x + _d(drop, 'enc%d.res1' % i, ffn(P, pre + '.feed_forward', y, drop, 'enc%d' % i))
    return layer_norm(P, 'model.encoder.norm', x), smask


def decode(P: Params, memory, smask, seq, tmask, h: int, n_layers: int, drop: Drop = None):
    """tgt_embed (Embeddings*sqrt(d) :215 + PositionalEncoding :231-233) + Decoder (:104-130)."""
    D = memory.shape[-1]
    x = P['model.tgt_embed.0.lut.weight'][seq] * math.sqrt(D)
    x = _d(drop, 'tgt_embed', x + P['model.tgt_embed.1.pe'][:, :seq.shape[1]])
    for i in range(n_layers):
        pre = 'model.decoder.layers.%d' % i
        y = layer_norm(P, pre + '.sublayer.0.norm', x)
        x = x + _d(drop, 'dec%d.res0' % i, mha(P, pre + '.self_attn', y, y, y, tmask, h, drop, 'dec%d.self' % i))
        y = layer_norm(P, pre + '.sublayer.1.norm', x)
        x = x + _d(drop, 'dec%d.res1' % i, mha(P, pre + '.src_attn', y, memory, memory, smask, h, drop, 'dec%d.src' % i))
        y = layer_norm(P, pre + '.sublayer.2.norm', x)
        x = x + _d(drop, 'dec%d.res2' % i, ffn(P, pre + '.feed_forward', y, drop, 'dec%d' % i))
    return layer_norm(P, 'model.decoder.norm', x)


def target_mask(seq):
    """(seq != eos) & (seq != pad) with position 0 forced on, AND causal (TransformerModel.py:324-328)."""
    m = (seq != 0)
    m[:, 0] = True
    T = seq.shape[1]
    causal = torch.tril(torch.ones(T, T, dtype=torch.bool))
    return m.unsqueeze(-2) & causal.unsqueeze(0)


def forward_teacher(P: Params, att_feats, seq, att_masks, h: int, n_enc: int, n_dec: int, drop: Drop = None, want_logsoftmax=True,
                    encode_per_caption: bool = True):
    """TransformerModel._forward (:340-348): log-probs [N,T,V1] (no early break, no zero columns).

    encode_per_caption=True is _forward's own order (regions repeated BEFORE the encoder, see encode()).  False restates the
    teacher-forced re-run of a rollout that AttModel._sample produced: _sample encodes the B images once (_prepare_feature,
    :306-311) and repeats the memory, so its dropout realisation has per-IMAGE encoder masks.  Same numbers when dropout is off."""
    if seq.ndim == 3:
        seq = seq.reshape(-1, seq.shape[2])
    B = att_feats.shape[0]
    n = seq.shape[0] // B
    memory, smask = encode(P, att_feats, att_masks, h, n_enc, drop, rows_per_image=n if encode_per_caption else 1)
    if n > 1 and not encode_per_caption:
        memory = memory.repeat_interleave(n, 0)
        smask = None if smask is None else smask.repeat_interleave(n, 0)
    out = decode(P, memory, smask, seq, target_mask(seq.clone()), h, n_dec, drop)
    logits = out @ P['model.generator.proj.weight'].t() + P['model.generator.proj.bias']
    return F.log_softmax(logits, dim=-1) if want_logsoftmax else logits      # output_logsoftmax = 0: AttModel.py:171-175


def greedy(P: Params, att_feats, att_masks, h: int, n_enc: int, n_dec: int, max_len: int):
    """AttModel._sample (greedy) with TransformerModel.core (:351-362): stateless re-decode of the prefix
    each step under a causal mask; finished rows emit 0 / zero rows (AttModel.py:340-347)."""
    B = att_feats.shape[0]
    V1 = P['model.generator.proj.weight'].shape[0]
    memory, smask = encode(P, att_feats, att_masks, h, n_enc)
    ys = torch.zeros(B, 1, dtype=torch.long)
    seq = torch.zeros(B, max_len, dtype=torch.long)
    seq_logp = att_feats.new_zeros(B, max_len, V1)
    unfinished = None
    for t in range(max_len):
        T = ys.shape[1]
        causal = torch.tril(torch.ones(T, T, dtype=torch.bool)).unsqueeze(0)
        out = decode(P, memory, smask, ys, causal, h, n_dec)
        logp = F.log_softmax(out[:, -1] @ P['model.generator.proj.weight'].t() + P['model.generator.proj.bias'], dim=-1)
        it = torch.max(logp, 1)[1]
        if t == 0:
            unfinished = it != 0
        else:
            it = it * unfinished.long()
            logp = logp * unfinished.unsqueeze(1).to(logp)
            unfinished = unfinished & (it != 0)
        seq[:, t] = it
        seq_logp[:, t] = logp
        if int(unfinished.sum()) == 0:
            break
        ys = torch.cat([ys, it.unsqueeze(1)], 1)
    return seq, seq_logp
