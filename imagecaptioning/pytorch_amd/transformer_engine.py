"""Host-side driver of the Transformer captioner (BASELINE configs[3]) on libcapmi.

Restates TransformerModel.py of the reference as a sequence of C-ABI launches: every contraction is
``capmi_gemm_f32`` (fused bias / ReLU / dropout / residual epilogues), LayerNorm / short-sequence MHA /
embedding+PE / log-softmax are the kernels of csrc/transformer.hip.  The backward is hand-written (no
autograd graph): each block object keeps what its backward needs.

MI355X-first differences from the reference's dataflow (same numbers):
 * cross-attention K/V are projected once per IMAGE ([B*K, D]) and shared by the n caption rows of the image
   inside the attention kernel (the reference repeats the memory n times first: TransformerModel.py:330-334,
   5x redundant K/V GEMMs);
 * sampling uses a real KV cache written in place by the K/V GEMMs (ldc = Lmax*D) instead of re-decoding the
   whole prefix every step (TransformerModel.core :351-362 -- sum_t t = 210 token-steps per row instead of 20).
"""
import os

import torch

from . import _lib, ops
from ._lib import lib, ptr, check, stream_ptr

_f32 = torch.float32
EPS = 1e-6


# --------------------------------------------------------------------------- thin op wrappers
def layernorm_fwd(x, a, b):
    M, D = x.shape
    y = torch.empty_like(x)
    mean = torch.empty(M, dtype=_f32, device=x.device)
    inv = torch.empty(M, dtype=_f32, device=x.device)
    check(lib.capmi_layernorm_fwd(ptr(x), ptr(a), ptr(b), ptr(y), ptr(mean), ptr(inv), M, D, EPS, stream_ptr()), 'layernorm_fwd')
    return y, mean, inv


def layernorm_bwd(dy, x, a, mean, inv, dx_accum, d_a=None, d_b=None):
    """dx_accum += d(LN)/dx ; d_a, d_b (given or new) receive the gain / bias gradients.  `dy` must not be modified by the
    caller afterwards (inside deferred_grads its column sum is taken at the end of the backward)."""
    M, D = x.shape
    d = Lin.deferred
    if d is not None and d_a is not None and D <= 2048 and os.environ.get('CAPMI_LN_PARTS', '1') != '0':
        # r6: per-wave partial sums of both parameter gradients leave the backward launch itself ([~M/16, D] rows each); the batched
        # column sum at the end of the backward reads those instead of g_scaled and dy ([M, D] each, g_scaled written here first)
        W = int(lib.capmi_layernorm_bwd_parts_rows(M))
        parts = torch.empty(2, W, D, dtype=_f32, device=x.device)
        check(lib.capmi_layernorm_bwd_parts(ptr(dy), ptr(x), ptr(a), ptr(mean), ptr(inv), ptr(dx_accum), 1, parts[0].data_ptr(),
                                            parts[1].data_ptr(), M, D, EPS, stream_ptr()), 'layernorm_bwd_parts')
        d.colsum(parts[0], d_a)
        d.colsum(parts[1], d_b)
        return d_a, d_b
    g = torch.empty_like(x)
    check(lib.capmi_layernorm_bwd(ptr(dy), ptr(x), ptr(a), ptr(mean), ptr(inv), ptr(dx_accum), 1, ptr(g), M, D, EPS,
                                  stream_ptr()), 'layernorm_bwd')
    if d is not None and d_a is not None:
        d.colsum(g, d_a)
        d.colsum(dy, d_b)
        return d_a, d_b
    return ops.colsum(g, out=d_a), ops.colsum(dy, out=d_b)


def mha_fwd(q, k, v, ldkv, Nq, q_per_kv, Tq, Tk, h, mask=None, mask_tq=1, mask_per_q=0, causal=0, q_pos0=0, drop=None,
            want_p=True, kstride=0, qstride=0, D=None):
    """q / k / v: tensors or (tensor, element offset) column blocks of a fused projection (then pass D and the row pitches)"""
    dev = (q[0] if isinstance(q, tuple) else q).device
    D = D or q.shape[-1]
    if (D // h) % 4:
        raise NotImplementedError('multi-head attention with head size %d: the kernels move the head dimension in 16-byte pieces '
                                  '(head size %% 4 == 0; every BASELINE config uses 64 or 128)' % (D // h))
    o = torch.empty(Nq, Tq, D, dtype=_f32, device=dev)
    p = torch.empty(Nq, h, Tq, Tk, dtype=_f32, device=dev) if want_p else None
    check(lib.capmi_mha_fwd_s(_a(q), qstride, _a(k), _a(v), ldkv, kstride, ptr(mask), mask_tq, mask_per_q, causal, q_pos0, ptr(drop),
                              ptr(o), ptr(p), Nq, q_per_kv, Tq, Tk, h, D // h, stream_ptr()), 'mha_fwd')
    return o, p


def _a(x):
    """tensor or (tensor, element offset) -> device address"""
    if isinstance(x, tuple):
        return x[0].data_ptr() + 4 * x[1]
    return None if x is None else x.data_ptr()


def mha_bwd(d_o, q, k, v, ldkv, p, drop, Nq, q_per_kv, Tq, Tk, h, kstride=0, dk_out=None, dv_out=None, dkv_ld=0, dkv_stride=0,
            accumulate=False, qstride=0, dq=None, dq_stride=0):
    D = d_o.shape[-1]
    Nkv = Nq // q_per_kv
    if dq is None:
        dq = torch.empty(Nq, Tq, D, dtype=_f32, device=d_o.device)
    if dk_out is None:
        dk_out = torch.empty(Nkv, Tk, D, dtype=_f32, device=d_o.device)
        dv_out = torch.empty(Nkv, Tk, D, dtype=_f32, device=d_o.device)
    check(lib.capmi_mha_bwd_s(ptr(d_o), _a(q), qstride, _a(k), _a(v), ldkv, kstride, ptr(p), ptr(drop), _a(dq), dq_stride, _a(dk_out),
                              _a(dv_out), dkv_ld, dkv_stride, int(accumulate), Nq, q_per_kv, Tq, Tk, h, D // h, stream_ptr()), 'mha_bwd')
    return dq, dk_out, dv_out


class Lin:
    """y = [residual +] mask * act(x W^T + b); backward writes dW, db into `grads` and returns dx."""

    deferred = None      # an ops.DeferredGrads while a whole-model backward runs (see deferred_grads)

    def __init__(self, P, grads, wname, bname):
        self.P, self.grads, self.wn, self.bn = P, grads, wname, bname

    def fwd(self, x, relu=False, mask=None, residual=None):
        W = self.P[self.wn]
        M, K = x.shape
        N = W.shape[0]
        self.x, self.relu, self.mask = x, relu, mask
        if residual is not None:
            y = torch.empty(M, N, dtype=_f32, device=x.device)       # residual added in the GEMM epilogue (capmi_gemm_desc.addend)
            ops.gemm([(x, K, W, K, K, 1)], M, N, y, bias=self.P[self.bn], relu=relu, mul_mask=mask, addend=residual)
            self.y_act = None
        else:
            y = torch.empty(M, N, dtype=_f32, device=x.device)
            ops.gemm([(x, K, W, K, K, 1)], M, N, y, bias=self.P[self.bn], relu=relu, mul_mask=mask)
            self.y_act = y if relu else None
        return y

    def bwd_params(self, dy, fresh=False):
        """dW, db; returns the gradient at the GEMM output (after the activation / mask Jacobian).  fresh: the caller will
        not modify `dy` afterwards (a running gradient accumulator must not be read at the end of the backward).
        (r4: the ReLU / mask Jacobian as an epilogue of the dX GEMM above was measured: its two extra reads per output stall the
        MFMA waves of the persistent fat kernel -- 12 FFN dX GEMMs 56 -> 83 us each for 0.3 ms of elementwise passes saved)"""
        if self.relu or self.mask is not None:
            dy = ops.relu_mask_bwd(dy.contiguous(), self.y_act if self.relu else None, self.mask)
            fresh = True
        d = Lin.deferred
        if d is not None:
            d.dw(dy, self.x, self.grads[self.wn], final=fresh, colsum_out=self.grads[self.bn])
        else:
            ops.matmul_tn(dy, self.x, out=self.grads[self.wn])
            ops.colsum(dy, out=self.grads[self.bn])
        return dy

    def bwd(self, dy, need_dx=True, fresh=False):
        dy = self.bwd_params(dy, fresh)
        return ops.matmul_nn(dy, self.P[self.wn]) if need_dx else None


class deferred_grads:
    """with deferred_grads(device): every Lin.bwd inside leaves its dW split-K slabs / records its bias column sum; leaving
    the block finishes them all in two launches (ops.DeferredGrads)."""

    def __init__(self, device):
        self.dev = device

    def __enter__(self):
        self.prev = Lin.deferred
        Lin.deferred = ops.DeferredGrads(self.dev) if os.environ.get('CAPMI_DEFER_GRADS', '1') != '0' else None
        return self

    def __exit__(self, et, ev, tb):
        d, Lin.deferred = Lin.deferred, self.prev
        if d is not None and et is None:
            d.flush()
        elif d is not None:
            d.abandon()          # error path: the side stream may still read the kept operands
        return False


def bwd_sum(lins, dys):
    """sum_i dy_i W_i for projections of the SAME input (Q/K/V of self-attention, K/V of cross-attention) as ONE GEMM whose
    K dimension walks the (dy_i, W_i) pairs as segments: no per-projection dX launches and no elementwise adds."""
    dys = [l.bwd_params(dy, fresh=True) for l, dy in zip(lins, dys)]       # mha_bwd outputs: written once
    M = dys[0].shape[0]
    Ws = [l.P[l.wn] for l in lins]
    N = Ws[0].shape[1]
    out = torch.empty(M, N, dtype=_f32, device=dys[0].device)
    ops.gemm([(dy, dy.shape[1], W, N, dy.shape[1], 1) for dy, W in zip(dys, Ws)], M, N, out, a_layout=0, b_layout=1)
    return out


class Norm:
    def __init__(self, P, grads, pre):
        self.P, self.grads, self.pre = P, grads, pre

    def fwd(self, x):
        self.x = x
        y, self.mean, self.inv = layernorm_fwd(x, self.P[self.pre + '.a_2'], self.P[self.pre + '.b_2'])
        return y

    def bwd(self, dy, dx_accum):
        layernorm_bwd(dy, self.x, self.P[self.pre + '.a_2'], self.mean, self.inv, dx_accum, self.grads[self.pre + '.a_2'],
                      self.grads[self.pre + '.b_2'])


def _back_to_back(ts):
    """views of ONE storage (the flat buffer) that follow each other without a gap"""
    if any(t is None for t in ts):
        return False
    base = ts[0].untyped_storage().data_ptr()
    return all(t.untyped_storage().data_ptr() == base for t in ts) and \
        all(a.data_ptr() + 4 * a.numel() == b.data_ptr() for a, b in zip(ts, ts[1:]))


def fused_lin(P, grads, wnames, bnames):
    """ONE Lin over several Linear layers of the same input when their weights, biases and gradient views sit back to back in the
    flat buffers (FlatParams lays out a model's `_flat_groups()` that way): y = x [W0; W1; ...]^T, one dW, one column sum, one dX.
    None when they do not (a model that was not flattened: the per-layer path runs)."""
    Ws, bs = [P.get(n) for n in wnames], [P.get(n) for n in bnames]
    gW, gb = [grads.get(n) for n in wnames], [grads.get(n) for n in bnames]
    if not (_back_to_back(Ws) and _back_to_back(bs)):
        return None
    if grads and not (_back_to_back(gW) and _back_to_back(gb)):
        return None
    rows, K = sum(w.shape[0] for w in Ws), Ws[0].shape[1]
    view = lambda t, shape: t.as_strided(shape, (shape[1], 1) if len(shape) == 2 else (1,))     # noqa: E731
    Pf = {'w': view(Ws[0], (rows, K)), 'b': view(bs[0], (rows,))}
    gf = {'w': view(gW[0], (rows, K)), 'b': view(gb[0], (rows,))} if grads else {}
    return Lin(Pf, gf, 'w', 'b')


class Attn:
    """MultiHeadedAttention (TransformerModel.py:164-195) with projections; kv_src None => self-attention.
    r4: self-attention projects q, k, v with ONE GEMM (N = 3D) when the three weights are one flat-buffer group; cross-attention
    may be handed K / V that one GEMM projected for all layers (`kv_fused`)."""

    def __init__(self, P, grads, pre, h, fuse_qkv=True):
        self.h = h
        self.lq, self.lk, self.lv, self.lo = (Lin(P, grads, '%s.linears.%d.weight' % (pre, i), '%s.linears.%d.bias' % (pre, i))
                                              for i in range(4))
        self.lqkv = fused_lin(P, grads, ['%s.linears.%d.weight' % (pre, i) for i in range(3)],
                              ['%s.linears.%d.bias' % (pre, i) for i in range(3)]) if fuse_qkv else None

    def fwd(self, x, Nq, Tq, kv=None, Nkv=None, Tk=None, q_per_kv=1, mask=None, mask_tq=1, mask_per_q=0, drop_p=None,
            residual=None, res_mask=None, kv_fused=None):
        """kv_fused = (buffer [Nkv*Tk, W], column of K, column of V, gradient buffer): K / V already projected"""
        D = x.shape[1]
        self.self_attn = kv is None and kv_fused is None
        if self.self_attn:
            kv, Nkv, Tk = x, Nq, Tq
        self.dims = (Nq, Tq, Nkv, Tk, q_per_kv)
        self.drop_p = drop_p
        self.kv_fused = kv_fused
        self.qs = self.ks = 0
        if self.self_attn and self.lqkv is not None:
            y = self.lqkv.fwd(x)                                   # [rows, 3D]: q | k | v
            self.q, self.k, self.v, self.qs, self.ks = (y, 0), (y, D), (y, 2 * D), 3 * D, 3 * D
        else:
            self.q = self.lq.fwd(x)
            if kv_fused is not None:
                buf, ck, cv, _ = kv_fused
                self.k, self.v, self.ks = (buf, ck), (buf, cv), buf.shape[1]
            else:
                self.k = self.lk.fwd(kv)
                self.v = self.lv.fwd(kv)
        o, self.p = mha_fwd(self.q, self.k, self.v, Tk * (self.ks or D), Nq, q_per_kv, Tq, Tk, self.h, mask, mask_tq, mask_per_q, 0, 0,
                            drop_p, kstride=self.ks, qstride=self.qs, D=D)
        return self.lo.fwd(o.view(Nq * Tq, D), mask=res_mask, residual=residual)

    def bwd(self, dy):
        """returns (dx_query_side, dkv) -- for self-attention both are summed into one tensor."""
        Nq, Tq, Nkv, Tk, q_per_kv = self.dims
        d_o = self.lo.bwd(dy)
        D = d_o.shape[1]
        ldkv = Tk * (self.ks or D)
        if self.self_attn and self.lqkv is not None:
            # dq | dk | dv land in the column blocks of ONE [rows, 3D] buffer: one dW GEMM, one column sum, one dX GEMM (K = 3D)
            dqkv = torch.empty(Nq * Tq, 3 * D, dtype=_f32, device=d_o.device)
            mha_bwd(d_o.view(Nq, Tq, D), self.q, self.k, self.v, ldkv, self.p, self.drop_p, Nq, q_per_kv, Tq, Tk, self.h,
                    kstride=self.ks, qstride=self.qs, dq=(dqkv, 0), dq_stride=3 * D, dk_out=(dqkv, D), dv_out=(dqkv, 2 * D),
                    dkv_ld=Tk * 3 * D, dkv_stride=3 * D)
            return self.lqkv.bwd(dqkv, fresh=True), None
        if self.kv_fused is not None:
            buf, ck, cv, dbuf = self.kv_fused                     # dK / dV into this layer's columns of the shared gradient buffer
            dq, _, _ = mha_bwd(d_o.view(Nq, Tq, D), self.q, self.k, self.v, ldkv, self.p, self.drop_p, Nq, q_per_kv, Tq, Tk, self.h,
                               kstride=self.ks, dk_out=(dbuf, ck), dv_out=(dbuf, cv), dkv_ld=Tk * dbuf.shape[1],
                               dkv_stride=dbuf.shape[1])
            return self.lq.bwd(dq.view(Nq * Tq, D), fresh=True), None
        dq, dk, dv = mha_bwd(d_o.view(Nq, Tq, D), self.q, self.k, self.v, ldkv, self.p, self.drop_p, Nq, q_per_kv, Tq, Tk,
                             self.h)
        if self.self_attn:
            return bwd_sum((self.lq, self.lk, self.lv), (dq.view(Nq * Tq, D), dk.view(Nkv * Tk, D), dv.view(Nkv * Tk, D))), None
        dx = self.lq.bwd(dq.view(Nq * Tq, D), fresh=True)
        return dx, bwd_sum((self.lk, self.lv), (dk.view(Nkv * Tk, D), dv.view(Nkv * Tk, D)))


class FFN:
    def __init__(self, P, grads, pre):
        self.l1 = Lin(P, grads, pre + '.w_1.weight', pre + '.w_1.bias')
        self.l2 = Lin(P, grads, pre + '.w_2.weight', pre + '.w_2.bias')

    def fwd(self, x, drop_ff, residual, res_mask):
        return self.l2.fwd(self.l1.fwd(x, relu=True, mask=drop_ff), mask=res_mask, residual=residual)

    def bwd(self, dy):
        return self.l1.bwd(self.l2.bwd(dy), fresh=True)


class Dropper:
    def __init__(self, p, seed, device, training):
        self.p, self.seed, self.dev, self.on, self.k = p, seed, device, (training and p > 0), 0

    def __call__(self, *shape):
        if not self.on:
            return None
        self.k += 1
        return ops.dropout_mask(shape, self.p, self.seed, self.k << 36, self.dev)

    def many(self, shapes):
        """several masks (<= CAPMI_MAX_MASKS) in ONE launch, e.g. the per-step masks of all T steps of a rollout as [T, ...]"""
        if not self.on:
            return [None] * len(shapes)
        specs = []
        for sh in shapes:
            self.k += 1
            specs.append((tuple(sh), self.k << 36, None, self.dev))
        out = []
        for i in range(0, len(specs), 4):                  # CAPMI_MAX_MASKS per launch
            out += ops.dropout_masks(specs[i:i + 4], self.p, self.seed)
        return out


class TransformerGraph:
    """One teacher-forced forward (and its backward) of TransformerModel._forward (TransformerModel.py:340-348)."""

    def __init__(self, P, grads, h, n_enc, n_dec, drop_att_embed, dropout, training, seed):
        self.P, self.grads, self.h, self.n_enc, self.n_dec = P, grads, h, n_enc, n_dec
        self.dev = P['att_embed.0.weight'].device
        self.drop_embed = Dropper(drop_att_embed, seed, self.dev, training)
        self.drop = Dropper(dropout, seed ^ 0x5bd1e995, self.dev, training)

    # ---------------- encoder
    def encode(self, att_feats, att_masks, rows_per_image=1):
        """rows_per_image = n > 1: TransformerModel._forward's train-mode dataflow (TransformerModel.py:316-321, 343-345) -- att_embed
        and ITS dropout run on the B images, the embedded regions are then repeated n times and the encoder runs on all B * n rows,
        each caption row under its own dropout masks (pinned by tests/golden/train_mode.npz).  From here on the graph's `B` is
        B * n and decode() must be called with n = 1.  With dropout off the n copies are identical: callers pass 1 (one encoder
        pass per image, same numbers)."""
        P, g = self.P, self.grads
        B, K, F = att_feats.shape
        D = P['att_embed.0.weight'].shape[0]
        self.B, self.K, self.D = B, K, D
        self.enc_rep = int(rows_per_image)
        m = self.drop_embed(B * K, D)
        if att_masks is not None:
            mm = att_masks.reshape(B * K, 1).expand(B * K, D)
            m = (mm if m is None else m * mm).contiguous()
            self.smask = att_masks.to(torch.uint8).contiguous()          # [B,K] -> broadcast over queries
        else:
            self.smask = None
        self.embed = Lin(P, g, 'att_embed.0.weight', 'att_embed.0.bias')
        x = self.embed.fwd(att_feats.reshape(B * K, F), relu=True, mask=m)
        if self.enc_rep > 1:
            n = self.enc_rep
            x = x.view(B, 1, K * D).expand(B, n, K * D).reshape(B * n * K, D)
            if self.smask is not None:
                self.smask = self.smask.view(B, 1, K).expand(B, n, K).reshape(B * n, K)
            self.B = B = B * n
        self.enc = []
        shapes = []
        for i in range(self.n_enc):                      # the encoder's masks, 4 per launch, in the order they are consumed
            Fh = self.P['model.encoder.layers.%d.feed_forward.w_1.weight' % i].shape[0]
            shapes += [(B, self.h, K, K), (B * K, D), (B * K, Fh), (B * K, D)]
        masks = iter(self.drop.many(shapes))
        for i in range(self.n_enc):
            pre = 'model.encoder.layers.%d' % i
            n0, at = Norm(P, g, pre + '.sublayer.0.norm'), Attn(P, g, pre + '.self_attn', self.h)
            n1, ff = Norm(P, g, pre + '.sublayer.1.norm'), FFN(P, g, pre + '.feed_forward')
            x = at.fwd(n0.fwd(x), B, K, mask=self.smask, mask_tq=1, mask_per_q=1, drop_p=next(masks), residual=x,
                       res_mask=next(masks))
            x = ff.fwd(n1.fwd(x), next(masks), residual=x, res_mask=next(masks))
            self.enc.append((n0, at, n1, ff))
        self.enc_norm = Norm(P, g, 'model.encoder.norm')
        self.memory = self.enc_norm.fwd(x)                             # [B*K, D]
        return self.memory

    def decoder_masks(self, N, T):
        """The decoder's dropout masks of one teacher-forced pass over [N,T] tokens, in the order decode() consumes them:
        target embedding [N,T,D], then per layer (self-attention probabilities [N,h,T,T], residual [N*T,D], source-attention
        probabilities [N,h,T,K], residual, feed-forward hidden [N*T,F], residual).  One definition for decode() and for the
        KV-cached sampler (Decoder), whose step t applies position t's rows of the same masks: with the same seed the
        differentiated pass is the sampled pass (loss_wrapper.py:63-68)."""
        D, K, h = self.D, self.K, self.h
        shapes = [(N, T, D)]
        for i in range(self.n_dec):
            F = self.P['model.decoder.layers.%d.feed_forward.w_1.weight' % i].shape[0]
            shapes += [(N, h, T, T), (N * T, D), (N, h, T, K), (N * T, D), (N * T, F), (N * T, D)]
        return self.drop.many(shapes)                      # 4 masks per launch (same Philox offsets as one call per mask)

    # ---------------- decoder (teacher forced)
    def decode(self, seq, n, raw=False):
        P, g = self.P, self.grads
        N, T = seq.shape
        B, K, D = self.B, self.K, self.D
        self.N, self.T, self.n, self.seq = N, T, n, seq
        masks = iter(self.decoder_masks(N, T))
        pad = (seq != 0)
        pad[:, 0] = True                                               # TransformerModel.py:324-326
        causal = torch.tril(torch.ones(T, T, dtype=torch.bool, device=seq.device))
        tmask = (pad.unsqueeze(-2) & causal.unsqueeze(0)).to(torch.uint8).contiguous()      # [N,T,T]
        self.drop_tgt = next(masks)
        x = torch.empty(N * T, D, dtype=_f32, device=seq.device)
        check(lib.capmi_embed_pe_fwd(ptr(seq), T, ptr(P['model.tgt_embed.0.lut.weight']), ptr(P['model.tgt_embed.1.pe']),
                                     ptr(self.drop_tgt), ptr(x), N, T, D, 0, stream_ptr()), 'embed_pe_fwd')
        self.dec = []
        # r4: the cross-attention K / V of ALL decoder layers are one GEMM over the memory (N = n_dec * 2D) when their weights are
        # one flat-buffer group; each layer reads -- and its backward fills -- its own column blocks
        names = [('model.decoder.layers.%d.src_attn.linears.%d' % (i, j)) for i in range(self.n_dec) for j in (1, 2)]
        self.kv_all = fused_lin(P, g, [nm + '.weight' for nm in names], [nm + '.bias' for nm in names]) if self.n_dec else None
        if self.kv_all is not None:
            self.kv_buf = self.kv_all.fwd(self.memory)                          # [B*K, n_dec * 2D]
            self.dkv_buf = torch.empty_like(self.kv_buf) if g else None
        for i in range(self.n_dec):
            pre = 'model.decoder.layers.%d' % i
            n0, sa = Norm(P, g, pre + '.sublayer.0.norm'), Attn(P, g, pre + '.self_attn', self.h)
            n1, ca = Norm(P, g, pre + '.sublayer.1.norm'), Attn(P, g, pre + '.src_attn', self.h, fuse_qkv=False)
            n2, ff = Norm(P, g, pre + '.sublayer.2.norm'), FFN(P, g, pre + '.feed_forward')
            x = sa.fwd(n0.fwd(x), N, T, mask=tmask, mask_tq=T, mask_per_q=1, drop_p=next(masks), residual=x,
                       res_mask=next(masks))
            kvf = None if self.kv_all is None else (self.kv_buf, 2 * i * D, (2 * i + 1) * D, self.dkv_buf)
            x = ca.fwd(n1.fwd(x), N, T, kv=None if kvf else self.memory, Nkv=B, Tk=K, q_per_kv=n, mask=self.smask, mask_tq=1,
                       mask_per_q=0, drop_p=next(masks), residual=x, res_mask=next(masks), kv_fused=kvf)
            x = ff.fwd(n2.fwd(x), next(masks), residual=x, res_mask=next(masks))
            self.dec.append((n0, sa, n1, ca, n2, ff))
        self.dec_norm = Norm(P, g, 'model.decoder.norm')
        out = self.dec_norm.fwd(x)
        self.gen = Lin(P, g, 'model.generator.proj.weight', 'model.generator.proj.bias')
        logits = self.gen.fwd(out)
        V1 = logits.shape[1]
        self.raw = bool(raw)
        if self.raw:        # AttModel._sample(output_logsoftmax=0) (AttModel.py:171-175, 265): the rows ARE the logits; backward: raw
            self.logp = logits.view(N, T, V1)
            return self.logp
        self.logp = torch.empty(N, T, V1, dtype=_f32, device=seq.device)
        check(lib.capmi_log_softmax_rows(ptr(logits), ptr(self.logp), N * T, V1, stream_ptr()), 'log_softmax_rows')
        return self.logp

    # ---------------- backward of both
    def backward(self, g_logp, sparse=None):
        P, g = self.P, self.grads
        N, T, B, K, D = self.N, self.T, self.B, self.K, self.D
        V1 = self.logp.shape[-1]
        dev = self.logp.device
        dlogits = torch.empty(N * T, V1, dtype=_f32, device=dev)
        g_logp = None if g_logp is None else g_logp.contiguous()
        if sparse is not None:
            sparse.tok_ld = 1                    # [N,T] tokens / gradients seen as N*T rows of one step
        ops.logsoftmax_bwd(g_logp, sparse, self.logp, None, dlogits, N * T, 1, 1, V1, raw=self.raw)
        with deferred_grads(dev):
            self._backward_layers(dlogits, dev)

    def _backward_layers(self, dlogits, dev):
        P, g = self.P, self.grads
        N, T, B, K, D = self.N, self.T, self.B, self.K, self.D
        d_out = self.gen.bwd(dlogits, fresh=True)
        dx = torch.zeros(N * T, D, dtype=_f32, device=dev)
        self.dec_norm.bwd(d_out, dx)
        d_mem = torch.zeros(B * K, D, dtype=_f32, device=dev)
        for (n0, sa, n1, ca, n2, ff) in reversed(self.dec):
            # x3 = x2 + m*ff(n2(x2)) ; dx currently = d x3
            n2.bwd(ff.bwd(dx), dx)
            dq, dkv = ca.bwd(dx)
            if dkv is not None:
                d_mem += dkv
            n1.bwd(dq, dx)
            dself, _ = sa.bwd(dx)
            n0.bwd(dself, dx)
        if self.kv_all is not None:                     # every layer's dK | dV -> d(memory), dW and db of all 2 n_dec projections at once
            d_mem = self.kv_all.bwd(self.dkv_buf, fresh=True)
        g['model.tgt_embed.0.lut.weight'].zero_()
        check(lib.capmi_embed_pe_bwd(ptr(self.seq), T, ptr(dx), ptr(self.drop_tgt), ptr(g['model.tgt_embed.0.lut.weight']), N, T,
                                     D, stream_ptr()), 'embed_pe_bwd')
        # encoder
        dxe = torch.zeros(B * K, D, dtype=_f32, device=dev)
        self.enc_norm.bwd(d_mem, dxe)
        for (n0, at, n1, ff) in reversed(self.enc):
            n1.bwd(ff.bwd(dxe), dxe)
            dself, _ = at.bwd(dxe)
            n0.bwd(dself, dxe)
        if self.enc_rep > 1:                             # the n encoder rows of an image share one att_embed output
            dxe = dxe.view(B // self.enc_rep, self.enc_rep, K * D).sum(1).view(-1, D)
        self.embed.bwd(dxe, need_dx=False, fresh=self.enc_rep > 1)


# --------------------------------------------------------------------------- KV-cached sampling
class Decoder:
    """Incremental Transformer decoder (eval numerics) with a real KV cache written in place by the K/V GEMMs
    (ldc = L*D), instead of TransformerModel.core's re-decode of the whole prefix (TransformerModel.py:351-362).
    Holds the state of `rows` = B * rows_per_image hypotheses; used by greedy/sampling rollouts and by beam search
    (the self-attention caches of all layers are ONE stacked [2*n_dec, N, L*D] array so that a beam reorder is one
    launch)."""

    def __init__(self, P, att_feats, att_masks, h, n_enc, n_dec, L, rows_per_image_max, drop=None):
        """drop: None (eval numerics) or (drop_att_embed, dropout, seed): train-mode rollouts -- the encoder runs with the masks
        a TransformerGraph of that seed draws and step t applies position t's rows of its decoder masks, i.e. the rollout
        samples from exactly the dropout realisation that a teacher-forced pass with the same seed differentiates."""
        self.P, self.h, self.n_dec, self.L = P, h, n_dec, L
        dev = att_feats.device
        if drop is None:
            self.g = g = TransformerGraph(P, {}, h, n_enc, n_dec, 0.0, 0.0, False, 0)
        else:
            self.g = g = TransformerGraph(P, {}, h, n_enc, n_dec, drop[0], drop[1], True, drop[2])
        memory = g.encode(att_feats, att_masks)            # Lin objects keep refs only
        self.B, self.K, self.D = g.B, g.K, g.D
        self.N = N = self.B * rows_per_image_max
        D = self.D
        self.masks = None
        if drop is not None and g.drop.on:
            # time-major copies: step t reads contiguous [N, ...] slices (the attention rows are cut to their t+1 keys per step)
            self.masks = []
            for m in g.decoder_masks(N, L):
                if m.dim() == 4:                            # [N,h,T,Tk] -> [T,N,h,Tk]
                    self.masks.append(m.permute(2, 0, 1, 3).contiguous())
                else:                                       # [N,T,D] or [N*T,D] -> [T,N,D]
                    self.masks.append(m.reshape(N, L, -1).transpose(0, 1).contiguous())
        self.V1 = P['model.generator.proj.weight'].shape[0]
        z = lambda *s: torch.empty(*s, dtype=_f32, device=dev)       # noqa: E731
        self.z = z
        self.mem_k, self.mem_v = [], []
        for i in range(n_dec):
            pre = 'model.decoder.layers.%d.src_attn' % i
            self.mem_k.append(ops.linear(memory, P[pre + '.linears.1.weight'], P[pre + '.linears.1.bias']))
            self.mem_v.append(ops.linear(memory, P[pre + '.linears.2.weight'], P[pre + '.linears.2.bias']))
        self.kv = z(2 * n_dec, N, L * D)                   # [2i] = K cache, [2i+1] = V cache of layer i
        self.kv_alt = None                                  # ping-pong partner, allocated on the first beam reorder
        self.logits = z(N, self.V1)
        self.x = z(N, D)

    def _lin(self, xx, wname, bname, out=None, ldc=None, relu=False, residual=None, mask=None):
        P = self.P
        W = P[wname]
        M, Kd = xx.shape
        Nn = W.shape[0]
        if residual is not None:
            ops.gemm([(xx, Kd, W, Kd, Kd, 1)], M, Nn, residual, bias=P[bname], mul_mask=mask, accumulate=True)
            return residual
        if out is None:
            out = self.z(M, Nn)
        ops.gemm([(xx, Kd, W, Kd, Kd, 1)], M, Nn, out, ldc=ldc, bias=P[bname], relu=relu, mul_mask=mask)
        return out

    def step(self, t, it, rows_per_image):
        """token ids `it` [rows] at position t -> logits [rows, V1] (a view of an internal buffer)."""
        P, h, L, D, K = self.P, self.h, self.L, self.D, self.K
        rows = self.B * rows_per_image
        lin = self._lin
        st = stream_ptr()
        x = self.x[:rows]
        mk = self.masks
        if mk is not None and rows != self.N:
            raise ValueError('train-mode (dropout) decoding serves all %d rows of its masks, got %d' % (self.N, rows))
        m = (lambda j: mk[j][t]) if mk is not None else (lambda j: None)                     # noqa: E731
        ma = (lambda j: mk[j][t][..., :t + 1].contiguous().unsqueeze(2)) if mk is not None else (lambda j: None)   # noqa: E731
        check(lib.capmi_embed_pe_fwd(ptr(it), 1, ptr(P['model.tgt_embed.0.lut.weight']), ptr(P['model.tgt_embed.1.pe']), ptr(m(0)),
                                     ptr(x), rows, 1, D, t, st), 'embed_pe_fwd')
        xs = x.clone()
        for i in range(self.n_dec):
            pre = 'model.decoder.layers.%d' % i
            j0 = 1 + 6 * i                                  # TransformerGraph.decoder_masks order
            kc, vc = self.kv[2 * i], self.kv[2 * i + 1]
            y, _, _ = layernorm_fwd(xs, P[pre + '.sublayer.0.norm.a_2'], P[pre + '.sublayer.0.norm.b_2'])
            q = lin(y, pre + '.self_attn.linears.0.weight', pre + '.self_attn.linears.0.bias')
            lin(y, pre + '.self_attn.linears.1.weight', pre + '.self_attn.linears.1.bias', out=(kc, t * D), ldc=L * D)
            lin(y, pre + '.self_attn.linears.2.weight', pre + '.self_attn.linears.2.bias', out=(vc, t * D), ldc=L * D)
            o, _ = mha_fwd(q, kc, vc, L * D, rows, 1, 1, t + 1, h, want_p=False, drop=ma(j0))
            xs = lin(o.view(rows, D), pre + '.self_attn.linears.3.weight', pre + '.self_attn.linears.3.bias', residual=xs,
                     mask=m(j0 + 1))
            y, _, _ = layernorm_fwd(xs, P[pre + '.sublayer.1.norm.a_2'], P[pre + '.sublayer.1.norm.b_2'])
            q = lin(y, pre + '.src_attn.linears.0.weight', pre + '.src_attn.linears.0.bias')
            o, _ = mha_fwd(q, self.mem_k[i], self.mem_v[i], K * D, rows, rows_per_image, 1, K, h, mask=self.g.smask, mask_tq=1,
                           mask_per_q=0, want_p=False, drop=None if mk is None else mk[j0 + 2][t].unsqueeze(2))
            xs = lin(o.view(rows, D), pre + '.src_attn.linears.3.weight', pre + '.src_attn.linears.3.bias', residual=xs,
                     mask=m(j0 + 3))
            y, _, _ = layernorm_fwd(xs, P[pre + '.sublayer.2.norm.a_2'], P[pre + '.sublayer.2.norm.b_2'])
            hdn = lin(y, pre + '.feed_forward.w_1.weight', pre + '.feed_forward.w_1.bias', relu=True, mask=m(j0 + 4))
            xs = lin(hdn, pre + '.feed_forward.w_2.weight', pre + '.feed_forward.w_2.bias', residual=xs, mask=m(j0 + 5))
        y, _, _ = layernorm_fwd(xs, P['model.decoder.norm.a_2'], P['model.decoder.norm.b_2'])
        logits = self.logits[:rows]
        lin(y, 'model.generator.proj.weight', 'model.generator.proj.bias', out=logits)
        return logits

    def reorder(self, parent, cur):
        """beam search: cache row b*cur + parent[b,j] -> row b*bd + j, all layers in one launch."""
        from . import beam
        bd = self.N // self.B
        if self.kv_alt is None:
            self.kv_alt = torch.empty_like(self.kv)
        beam.reorder_rows(self.kv, self.kv_alt, parent, self.B, cur, bd)
        self.kv, self.kv_alt = self.kv_alt, self.kv


def sample(P, att_feats, att_masks, h, n_enc, n_dec, L, sample_n=1, mode='greedy', temperature=1.0, seed=0, forced=None,
           gumbel=None, top_k=0, top_p=0.0, drop=None, raw=False):
    """AttModel._sample with TransformerModel.core semantics, KV cache instead of prefix re-decode.  drop: see Decoder
    (None = eval numerics).  raw: the stored rows are the logits (output_logsoftmax=0, CAPMI_SELECT_RAW); same tokens.
    Returns (seq [N,L], seq_logp [N,L,V1])."""
    dev = att_feats.device
    dec = Decoder(P, att_feats, att_masks, h, n_enc, n_dec, L, sample_n, drop=drop)
    N, V1, n = dec.N, dec.V1, sample_n
    seq = torch.zeros(N, L, dtype=torch.long, device=dev)
    seq_logp = torch.zeros(N, L, V1, dtype=_f32, device=dev)
    sel = torch.zeros(N, L, dtype=_f32, device=dev)
    live = torch.zeros(N, L, dtype=torch.uint8, device=dev)
    it = torch.zeros(N, dtype=torch.long, device=dev)
    unf = torch.ones(N, dtype=torch.uint8, device=dev)
    mode_i = {'greedy': 0, 'sample': 1, 'forced': 2}[mode] | (_lib.SELECT_RAW if raw else 0)
    st = stream_ptr()
    for t in range(L):
        logits = dec.step(t, it, n)
        ops.logsoftmax_select(logits, t, L, mode_i, temperature, None if gumbel is None else gumbel[t], seed, forced, 0, seq, it, unf,
                              seq_logp, sel, live, top_k, top_p)
    return seq, seq_logp


def sample_beam(model, P, att_feats, att_masks, h, n_enc, n_dec, L, opt):
    """AttModel._sample_beam (AttModel.py:218-256) for the Transformer: beam_size hypotheses per image share the image's
    encoder memory (no repeat_tensors copy), the KV caches follow the beams by parent pointer."""
    from . import beam
    B, V1 = att_feats.shape[0], P['model.generator.proj.weight'].shape[0]
    return beam.beam_search_steps(model, lambda rows: Decoder(P, att_feats, att_masks, h, n_enc, n_dec, L, rows), B, V1, L, opt,
                                  att_feats.device)
