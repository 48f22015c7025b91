"""Host-side driver of the AoANet captioner (BASELINE configs[4], configs/aoa.yml) on libcapmi.

Restates AoAModel.py of the reference (refine 1, refine_aoa 1, use_ff 0, decoder_type AoA, use_multi_head 2,
mean_feats 1, ctx_drop 1) as C-ABI launches with a hand-written backward:
  prefill   att_embed -> 6 x [x + Drop(GLU(Linear([MHA(LN x) | LN x])))] -> LN -> masked mean, ctx2att (V|K halves)
  step      LSTMCell([xt | mean + Drop(ctx_prev)], h_att) -> LN -> q -> 8-head dot attention over the image's K/V halves
            -> GLU(Linear([att | h_att])) -> Drop -> logit           (AoA_Decoder_Core.forward, AoAModel.py:163-186)
MI355X-first: K/V stay per IMAGE (the attention kernel serves the n caption rows of an image from one LDS copy,
key stride 2R inside p_att rows, no repeat_tensors / narrow copies); the mean-feature term of the LSTM gates is
constant over time and enters as a per-image row bias; every core weight gradient is one time-batched GEMM.
"""
import os

import torch

from . import _lib, ops
from ._lib import lib, ptr, check, stream_ptr
from .transformer_engine import Lin, Norm, Dropper, mha_fwd, mha_bwd, layernorm_fwd, layernorm_bwd, EPS, fused_lin, deferred_grads

_f32 = torch.float32


def glu_fwd(pre, mask=None, residual=None):
    M, R2 = pre.shape
    out = torch.empty(M, R2 // 2, dtype=_f32, device=pre.device)
    check(lib.capmi_glu_fwd(ptr(pre), ptr(mask), ptr(residual), ptr(out), M, R2 // 2, stream_ptr()), 'glu_fwd')
    return out


def glu_bwd(d_out, mask, pre):
    d_pre = torch.empty_like(pre)
    check(lib.capmi_glu_bwd(ptr(d_out), ptr(mask), ptr(pre), ptr(d_pre), pre.shape[0], pre.shape[1] // 2, stream_ptr()), 'glu_bwd')
    return d_pre


def mul_mask(x, mask):
    return x if mask is None else ops.relu_mask_bwd(x.contiguous(), None, mask)


def dcat_halves(d_pre, W, M, R, mask_lo=None, mask_hi=None, ws=None):
    """(d_lo, d_hi) = the two [M,R] halves of d_pre [M,2R] @ W [2R,2R] (backward of a Linear over cat([att, query], -1),
    AoAModel.py:92,174), each through its dropout mask.  The GEMM leaves its K-slice slabs, capmi_split_halves finishes them into
    the two contiguous operands the consumers want: 2 launches instead of GEMM + reduction + 2 slice copies (+ 2 mask kernels)."""
    dev = d_pre.device
    ws = ws or ops.default_workspace(dev)
    lo, hi = torch.empty(M, R, dtype=_f32, device=dev), torch.empty(M, R, dtype=_f32, device=dev)
    if ws.capacity - ops.Workspace.COUNTER_FLOATS >= M * 2 * R:
        sp = ops.gemm([(d_pre, 2 * R, W, 2 * R, 2 * R, 1)], M, 2 * R, ws.buf, a_layout=0, b_layout=1, ws=ws, defer_reduce=True)
        src, stride = ws.slabs, M * 2 * R
    else:                                    # (a product larger than the workspace: finished by the GEMM itself)
        src, sp, stride = ops.matmul_nn(d_pre, W), 1, 0
    check(lib.capmi_split_halves(src.data_ptr(), sp, stride, ptr(mask_lo), ptr(mask_hi), ptr(lo), ptr(hi), M, R, stream_ptr()),
          'split_halves')
    return lo, hi


_dh_ws = {}


def _dh_workspace(dev, tag='dh'):
    """split-K slabs that outlive the next GEMM launch (the default workspace is reused by the GEMMs in between): 'dh' -- d_gates W_hh
    between two steps of the BPTT; r5, slab consumers: 'ctx' -- d_gates W_ih[:, E:] until the previous step's GLU backward,
    'dqn' -- dq Wq until the LayerNorm backward (which also reads the query half of d_cat from the default workspace)"""
    key = (str(dev), tag)
    if key not in _dh_ws:
        _dh_ws[key] = ops.Workspace(dev, floats=2 * 1024 * 1024)
    return _dh_ws[key]


# r5: the decode step's three remaining stand-alone split-K reductions (query projection, dq Wq, the context-input gradient) and the
# split of d_cat into its halves are done by the kernels that consume them (capmi_mha_fwd_qslabs, capmi_layernorm_bwd_slabs,
# capmi_mha_bwd_slabs, capmi_glu_bwd_add): 4 launches fewer per step and direction, same bits.  CAPMI_AOA_SLABS=0: the r4 launches.
SLAB_CONSUMERS = os.environ.get('CAPMI_AOA_SLABS', '1') != '0'


def _dw(dy, x, out, ldc=None, off=0, accumulate=False, bias=None):
    """out (+)= dy^T x, a weight gradient nothing reads before the optimizer: recorded for the grouped launch at the end of the
    backward (ops.DeferredGrads, r6) or issued at once.  bias: receives the column sums of dy (the bias gradient that goes with it)."""
    d = Lin.deferred
    if d is not None:
        d.dw(dy, x, out, final=True, ldc=ldc, out_off=off, accumulate=accumulate, colsum_out=bias)
        return
    if bias is not None:
        ops.colsum(dy, out=bias)
    K, M = dy.shape
    N = x.shape[1]
    ops.gemm([(dy, M, x, N, K, 1)], M, N, (out, off), ldc=N if ldc is None else ldc, a_layout=1, b_layout=1, accumulate=accumulate)


def _colsum(x, out):
    """out = column sums of x (which nobody writes any more): one batched launch at the end of the backward, or at once"""
    d = Lin.deferred
    if d is not None:
        d.colsum(x, out)
    else:
        ops.colsum(x, out=out)


class AoAGraph:
    def __init__(self, P, grads, h, drop_prob_lm, dropout_aoa, training, seed):
        self.P, self.g, self.h = P, grads, h
        dev = P['logit.weight'].device
        self.dev = dev
        self.d_lm = Dropper(drop_prob_lm, seed, dev, training)               # embed / att_embed / ctx_drop / out_drop
        self.d_att = Dropper(0.1, seed ^ 0x1234567, dev, training)            # attention probabilities (AoAModel.py:18,53)
        self.d_res = Dropper(0.1, seed ^ 0x7654321, dev, training)            # refiner SublayerConnection (AoAModel.py:119)
        self.d_aoa = Dropper(dropout_aoa, seed ^ 0x2468ace, dev, training)    # AoA input (AoAModel.py:44-48)

    # ------------------------------------------------------------------ prefill
    def prepare(self, att_feats, att_masks):
        P, g, h = self.P, self.g, self.h
        B, K, F = att_feats.shape
        R = P['att_embed.0.weight'].shape[0]
        self.B, self.K, self.R = B, K, R
        m = self.d_lm(B * K, R)
        self.att_masks = att_masks
        if att_masks is not None:
            mm = att_masks.reshape(B * K, 1).expand(B * K, R)
            m = (mm if m is None else m * mm).contiguous()
            self.smask = att_masks.to(torch.uint8).contiguous()
        else:
            self.smask = None
        self.embed = Lin(P, g, 'att_embed.0.weight', 'att_embed.0.bias')
        x = self.embed.fwd(att_feats.reshape(B * K, F), relu=True, mask=m)
        self.ref = []
        # the refiner's dropout masks of all 6 layers, 4 per launch and per dropper, drawn in the order the layers consume them
        # (same Philox offsets as one launch per mask: 24 -> 7 launches)
        att_masks_it = iter(self.d_att.many([(B, h, K, K)] * 6))
        aoa_masks_it = iter(self.d_aoa.many([(B * K, R)] * 12))
        res_masks_it = iter(self.d_res.many([(B * K, R)] * 6))
        for i in range(6):
            pre = 'refiner.layers.%d' % i
            n0 = Norm(P, g, pre + '.sublayer.0.norm')
            y = n0.fwd(x)
            lq, lk, lv = (Lin(P, g, '%s.self_attn.linears.%d.weight' % (pre, j), '%s.self_attn.linears.%d.bias' % (pre, j))
                          for j in range(3))
            # r4: q | k | v as ONE GEMM (N = 3R) when the model is flattened (AoAModel._flat_groups keeps the three weights back to
            # back); the attention kernels read / write the column blocks in place
            lqkv = fused_lin(P, g, ['%s.self_attn.linears.%d.weight' % (pre, j) for j in range(3)],
                             ['%s.self_attn.linears.%d.bias' % (pre, j) for j in range(3)])
            dp = next(att_masks_it)
            if lqkv is not None:
                qkv = lqkv.fwd(y)
                q, k, v = (qkv, 0), (qkv, R), (qkv, 2 * R)
                o, p = mha_fwd(q, k, v, K * 3 * R, B, 1, K, K, h, self.smask, 1, 1, 0, 0, dp, kstride=3 * R, qstride=3 * R, D=R)
            else:
                q, k, v = lq.fwd(y), lk.fwd(y), lv.fwd(y)
                o, p = mha_fwd(q, k, v, K * R, B, 1, K, K, h, self.smask, 1, 1, 0, 0, dp)
            o2 = o.view(B * K, R)
            m_o, m_y = next(aoa_masks_it), next(aoa_masks_it)
            od, yd = mul_mask(o2, m_o), mul_mask(y, m_y)
            W = P[pre + '.self_attn.aoa_layer.0.weight']                      # [2R, 2R], input [att | query]
            pre_act = torch.empty(B * K, 2 * R, dtype=_f32, device=self.dev)
            ops.gemm([(od, R, W, 2 * R, R, 1), (yd, R, (W, R), 2 * R, R, 1)], B * K, 2 * R, pre_act,
                     bias=P[pre + '.self_attn.aoa_layer.0.bias'])
            m_res = next(res_masks_it)
            x_new = glu_fwd(pre_act, m_res, x)
            self.ref.append(dict(pre=pre, n0=n0, lq=lq, lk=lk, lv=lv, lqkv=lqkv, q=q, k=k, v=v, p=p, dp=dp, od=od, yd=yd, m_o=m_o, m_y=m_y,
                                 pre_act=pre_act, m_res=m_res))
            x = x_new
        self.ref_norm = Norm(P, g, 'refiner.norm')
        self.att = self.ref_norm.fwd(x)                                        # [B*K, R]
        self.mean = torch.empty(B, R, dtype=_f32, device=self.dev)
        check(lib.capmi_meanpool_fwd(ptr(self.att), ptr(att_masks), ptr(self.mean), B, K, R, stream_ptr()), 'meanpool_fwd')
        self.ctx2att = Lin(P, g, 'ctx2att.weight', 'ctx2att.bias')
        self.p_att = self.ctx2att.fwd(self.att)                               # [B*K, 2R]  value | key
        return self.mean, self.att, self.p_att

    # ------------------------------------------------------------------ rollout
    def rollout(self, n, T, L, mode='forced', forced=None, teacher=False, temperature=1.0, seed=0, gumbel=None, keep=True,
                top_k=0, top_p=0.0, raw=False):
        """T decoder steps on N = B*n rows.  teacher: inputs forced[:, t] (AttModel._forward); else AttModel._sample with
        mode greedy / sample / forced (tokens chosen at t are fed at t+1).  raw (free-running only): the returned rows are the LOGITS
        (AttModel._sample(output_logsoftmax=0), AttModel.py:171-175, 265: what the margin structure losses read), the choice of the
        tokens is the same; the backward then takes the loss gradient as d(logits)."""
        P, h, B, K, R = self.P, self.h, self.B, self.K, self.R
        N = B * n
        V1, E = P['embed.0.weight'].shape
        dev = self.dev
        self.n, self.N, self.T, self.L, self.keep = n, N, T, L, keep
        z = lambda *s: torch.empty(*s, dtype=_f32, device=dev)       # noqa: E731
        W_ih, W_hh = P['core.att_lstm.weight_ih'], P['core.att_lstm.weight_hh']
        ld_ih = E + R
        ws = ops.default_workspace(dev)
        # mean-feature term of the gates, once per rollout: [B,4R] = mean W_ih[:, E:]^T
        self.mean_gates = z(B, 4 * R)
        ops.gemm([(self.mean, R, (W_ih, E), ld_ih, R, 1)], B, 4 * R, self.mean_gates)
        self.h_att, self.c_att, self.out = torch.zeros(T + 1, N, R, device=dev), torch.zeros(T + 1, N, R, device=dev), \
            torch.zeros(T + 1, N, R, device=dev)
        self.xt, self.ctx_in, self.gates = z(T, N, E), z(T, N, R), z(T, N, 4 * R)
        self.qn, self.q_ln_mean, self.q_ln_inv = z(T, N, R), z(T, N), z(T, N)
        self.q, self.att_o, self.pre2, self.out_drop = z(T, N, R), z(T, N, R), z(T, N, 2 * R), z(T, N, R)
        self.p_dec = z(T, N, h, 1, K)
        self.it_all = torch.empty(T, N, dtype=torch.long, device=dev)
        # the dropout masks of ALL T steps in two launches ([T, ...] arrays; step t uses slice t)
        self.m_xt_all, self.m_ctx_all, self.m_out_all = self.d_lm.many([(T, N, E), (T, N, R), (T, N, R)])
        self.m_patt_all = self.d_att(T, N, h, 1, K)
        unb = lambda a: [None] * T if a is None else list(a.unbind(0))       # noqa: E731
        self.m_xt, self.m_ctx, self.m_out, self.m_patt = unb(self.m_xt_all), unb(self.m_ctx_all), unb(self.m_out_all), unb(self.m_patt_all)
        self.seq = torch.zeros(N, L, dtype=torch.long, device=dev)
        # (the select kernel writes every (row, step < T) slot, zeros for finished rows included: only a rollout shorter than L needs
        #  the 38 MB fill -- 66 us of the NSC step)
        self.seq_logp = (torch.empty if T == L else torch.zeros)(N, L, V1, dtype=_f32, device=dev)
        self.sel = torch.zeros(N, L, dtype=_f32, device=dev)
        self.live = torch.zeros(N, L, dtype=torch.uint8, device=dev)
        it = torch.zeros(N, dtype=torch.long, device=dev)
        unf = torch.ones(N, dtype=torch.uint8, device=dev)
        mode_i = 2 if teacher else {'greedy': 0, 'sample': 1, 'forced': 2}[mode]
        self.raw = bool(raw) and not teacher
        if self.raw:
            mode_i |= _lib.SELECT_RAW
        st = stream_ptr()
        a_n, b_n = P['core.attention.norm.a_2'], P['core.attention.norm.b_2']
        Wq, bq = P['core.attention.linears.0.weight'], P['core.attention.linears.0.bias']
        Wc, bc = P['core.att2ctx.0.weight'], P['core.att2ctx.0.bias']
        # r3: the step's GEMM operands as producer-written bf16x3 planes (N <= 64 rows: the loader / consumer GEMM stages them by
        # LDS-DMA) and ONE launch behind the att2ctx GEMM -- capmi_glu_fwd_fused finishes its slabs, applies the GLU and writes
        # out, out_drop (+ planes: the logit GEMM's operand) and the NEXT step's dropped context input (+ planes): 14 -> 11 launches
        # per step
        use_pl = (N <= 64 and R % 4 == 0 and E % 4 == 0 and os.environ.get('CAPMI_AOA_PLANES', '1') != '0')
        if use_pl:
            nbR, nbE = int(lib.capmi_planes_bytes(R)), int(lib.capmi_planes_bytes(E))
            # (keyed by the exact K: columns >= K of a buffer are never written and must stay zero -- a buffer of another width with
            #  the same number of chunks would leave stale values there)
            pl_xt = ops.planes_scratch(dev, ('aoa_xt', E), nbE)
            pl_ctx, pl_h, pl_od = (ops.planes_scratch(dev, ('aoa_' + k, R), nbR) for k in ('ctx', 'h', 'od'))
            pl_zero = ops.zero_planes(dev, max(nbR, nbE) // 12288)
            self.ctx_in[0].zero_()                               # out_0 = 0 (AoAModel.py:127-129)
        # r5: a free-running rollout gets the embedding of step t+1 from the select launch of step t (capmi_next_embed, as the UpDown
        # driver does): one launch fewer per step; a teacher-forced one knows its tokens and keeps the per-step launch
        fold_embed = SLAB_CONSUMERS and use_pl and not teacher
        for t in range(T):
            m_xt, m_ctx, m_out, m_p = self.m_xt[t], self.m_ctx[t], self.m_out[t], self.m_patt[t]
            tok_src = (forced.data_ptr() + 8 * t, forced.shape[1]) if teacher else (ptr(it), 1)
            if fold_embed and t > 0:
                pass                                              # written by the select launch of step t-1
            elif use_pl:
                check(lib.capmi_embed_fwd_pl(tok_src[0], tok_src[1], ptr(self.it_all[t]), ptr(P['embed.0.weight']), ptr(m_xt),
                                             ptr(self.xt[t]), N, E, 1, ptr(pl_xt), st), 'embed_fwd_pl')
            else:
                check(lib.capmi_embed_fwd(tok_src[0], tok_src[1], ptr(self.it_all[t]), ptr(P['embed.0.weight']), ptr(m_xt),
                                          ptr(self.xt[t]), N, E, 1, st), 'embed_fwd')
                ctx_prev = self.out[t]
                if m_ctx is None:
                    self.ctx_in[t].copy_(ctx_prev)
                else:
                    check(lib.capmi_relu_mask_bwd(ptr(ctx_prev), None, ptr(m_ctx), ptr(self.ctx_in[t]), N * R, st), 'ctx_drop')
            splits = ops.gemm([(self.xt[t], E, W_ih, ld_ih, E, 1), (self.ctx_in[t], R, (W_ih, E), ld_ih, R, 1),
                               (self.h_att[t], R, W_hh, R, R, 1)], N, 4 * R, ws.buf, ws=ws, defer_reduce=True,
                              a_planes=[pl_xt, pl_zero if t == 0 else pl_ctx, pl_zero if t == 0 else pl_h] if use_pl else None)
            if use_pl:
                check(lib.capmi_lstm_cell_fwd_pl(ws.slabs.data_ptr(), splits, ptr(P['core.att_lstm.bias_ih']),
                                                 ptr(P['core.att_lstm.bias_hh']), ptr(self.mean_gates), n, None, ptr(self.c_att[t]),
                                                 ptr(self.h_att[t + 1]), ptr(self.c_att[t + 1]), ptr(self.gates[t]), None, None, N, R,
                                                 ptr(pl_h), None, st), 'lstm_cell_fwd_pl')
            else:
                check(lib.capmi_lstm_cell_fwd(ws.slabs.data_ptr(), splits, ptr(P['core.att_lstm.bias_ih']),
                                              ptr(P['core.att_lstm.bias_hh']), ptr(self.mean_gates), n, None, ptr(self.c_att[t]),
                                              ptr(self.h_att[t + 1]), ptr(self.c_att[t + 1]), ptr(self.gates[t]), None, None, N, R, st),
                      'lstm_cell_fwd')
            check(lib.capmi_layernorm_fwd(ptr(self.h_att[t + 1]), ptr(a_n), ptr(b_n), ptr(self.qn[t]), ptr(self.q_ln_mean[t]),
                                          ptr(self.q_ln_inv[t]), N, R, EPS, st), 'layernorm_fwd')
            # keys = second half of p_att rows, values = first half (AoAModel.py:168); per image, stride 2R
            if SLAB_CONSUMERS:
                spq = ops.gemm([(self.qn[t], R, Wq, R, R, 1)], N, R, ws.buf, ws=ws, defer_reduce=True)
                check(lib.capmi_mha_fwd_qslabs(ws.slabs.data_ptr(), R, spq, N * R, ptr(bq), ptr(self.q[t]), self.p_att.data_ptr() + 4 * R,
                                               ptr(self.p_att), K * 2 * R, 2 * R, ptr(self.smask), 1, 0, 0, 0, ptr(m_p), ptr(self.att_o[t]),
                                               ptr(self.p_dec[t]), N, n, 1, K, h, R // h, st), 'mha_fwd_qslabs')
            else:
                ops.gemm([(self.qn[t], R, Wq, R, R, 1)], N, R, self.q[t], bias=bq)
                check(lib.capmi_mha_fwd(ptr(self.q[t]), self.p_att.data_ptr() + 4 * R, ptr(self.p_att), K * 2 * R, 2 * R, ptr(self.smask),
                                        1, 0, 0, 0, ptr(m_p), ptr(self.att_o[t]), ptr(self.p_dec[t]), N, n, 1, K, h, R // h, st), 'mha_fwd')
            c_segs = [(self.att_o[t], R, Wc, 2 * R, R, 1), (self.h_att[t + 1], R, (Wc, R), 2 * R, R, 1)]
            if use_pl:
                sp2 = ops.gemm(c_segs, N, 2 * R, ws.buf, ws=ws, defer_reduce=True)
                nxt = t + 1 < T
                check(lib.capmi_glu_fwd_fused(ws.slabs.data_ptr(), sp2, N * 2 * R, ptr(bc), ptr(self.pre2[t]), ptr(self.out[t + 1]),
                                              ptr(m_out), ptr(self.out_drop[t]), ptr(pl_od),
                                              ptr(self.m_ctx[t + 1]) if nxt else None, ptr(self.ctx_in[t + 1]) if nxt else None,
                                              ptr(pl_ctx) if nxt else None, N, R, st), 'glu_fwd_fused')
            else:
                ops.gemm(c_segs, N, 2 * R, self.pre2[t], bias=bc)
                check(lib.capmi_glu_fwd(ptr(self.pre2[t]), None, None, ptr(self.out[t + 1]), N, R, st), 'glu_fwd')
                if m_out is None:
                    self.out_drop[t].copy_(self.out[t + 1])
                else:
                    check(lib.capmi_relu_mask_bwd(ptr(self.out[t + 1]), None, ptr(m_out), ptr(self.out_drop[t]), N * R, st), 'out_drop')
            # the logit GEMM leaves its K-slice slabs; log-softmax + select finishes them with the bias (no reduce launch)
            sp = ops.gemm([(self.out_drop[t], R, P['logit.weight'], R, R, 1)], N, V1, ws.buf, ws=ws, defer_reduce=True,
                          a_planes=[pl_od] if use_pl else None)
            ne = None
            if fold_embed and t + 1 < T:
                ne = dict(E=P['embed.0.weight'], mask=self.m_xt[t + 1], x=self.xt[t + 1], it_save=self.it_all[t + 1], relu=1, x_planes=pl_xt)
            ops.logsoftmax_select(ws.slabs, t, L, mode_i, temperature, None if gumbel is None else gumbel[t], seed, forced,
                                  1 if teacher else 0, self.seq, it, unf, self.seq_logp, self.sel, self.live, top_k, top_p,
                                  splits=sp, stride=N * V1, bias=P['logit.bias'], shape=(N, V1), next_embed=ne)
        return self.seq, self.seq_logp

    # ------------------------------------------------------------------ backward
    def backward(self, g_logp, sparse=None):
        # weight-gradient reductions and bias column sums of the Lin / Norm objects (refiner, att_embed, ctx2att) are finished by
        # two batched launches at the end (ops.DeferredGrads), as in the Transformer's backward
        self._bias_hh_copy = False
        with deferred_grads(self.dev):
            self._backward(g_logp, sparse)
        if self._bias_hh_copy:               # (the bias_ih column sums ride in the grouped launch that leaving the block issued)
            self.g['core.att_lstm.bias_hh'].copy_(self.g['core.att_lstm.bias_ih'])

    def _backward(self, g_logp, sparse=None):
        P, g, h, B, K, R, n, N, T, L = self.P, self.g, self.h, self.B, self.K, self.R, self.n, self.N, self.T, self.L
        V1, E = P['embed.0.weight'].shape
        dev = self.dev
        st = stream_ptr()
        z = lambda *s: torch.empty(*s, dtype=_f32, device=dev)       # noqa: E731
        g_logp = None if g_logp is None else g_logp.contiguous()
        dlogits = z(T, N, V1)
        ops.logsoftmax_bwd(g_logp, sparse, self.seq_logp, self.live, dlogits, N, L, T, V1, raw=self.raw)
        TN = T * N
        d_outdrop = ops.matmul_nn(dlogits.view(TN, V1), P['logit.weight'])            # [TN,R]
        _dw(dlogits.view(TN, V1), self.out_drop.view(TN, R), g['logit.weight'], bias=g['logit.bias'])
        d_outdrop = d_outdrop.view(T, N, R)
        W_ih, W_hh = P['core.att_lstm.weight_ih'], P['core.att_lstm.weight_hh']
        ld_ih = E + R
        Wq, Wc = P['core.attention.linears.0.weight'], P['core.att2ctx.0.weight']
        a_n = P['core.attention.norm.a_2']
        d_pre2_all, dq_all, dg_all, ln_g, ln_dy = z(T, N, 2 * R), z(T, N, R), z(T, N, 4 * R), z(T, N, R), z(T, N, R)
        d_p_att = torch.zeros(B * K, 2 * R, dtype=_f32, device=dev)
        dc_next = None
        d_out_all = mul_mask(d_outdrop, self.m_out_all)              # the out_drop Jacobian of all steps in one launch
        # r4 launch diet of the time loop (15 -> 11 launches per step):
        #  * the gradient reaching step t's context input, Drop(d_gates W_ih[:, E:]), is ACCUMULATED into d_out of step t-1 by the
        #    epilogue of that GEMM's split-K reduction (mask + accumulate) -- no mul_mask launch, no elementwise add;
        #  * dh_next = d_gates W_hh stays as K-slice slabs (its own workspace) and the next LSTM-cell backward sums them -- no reduce;
        #  * N <= 64: the cell backward also writes d_gates as bf16x3 planes, so both dX GEMMs run on the loader / consumer kernel;
        #  * step 0 needs neither product (out_0 = 0 and there is no earlier state).
        use_pl = N <= 64 and R % 4 == 0 and os.environ.get('CAPMI_AOA_PLANES', '1') != '0'
        pl_dg = ops.planes_scratch(dev, ('aoa_dg', 4 * R), int(lib.capmi_planes_bytes(4 * R))) if use_pl else None
        ws2 = _dh_workspace(dev)
        dh_slabs, dh_splits = None, 0
        # (capmi_layernorm_bwd_slabs moves rows in 16-byte pieces of at most 2 048 columns; other sizes take the r4 route)
        fuse = SLAB_CONSUMERS and R % 4 == 0 and R <= 2048
        ws, ws3, ws4 = ops.default_workspace(dev), _dh_workspace(dev, 'ctx'), _dh_workspace(dev, 'dqn')
        ctx_splits = 0
        for t in range(T - 1, -1, -1):
            # out_{t+1}: from the logit (through out_drop) and -- accumulated by step t+1 -- from its ctx input
            dq = dq_all[t]                                               # [N,R] = [N,1,R], written in place
            if fuse:
                # the ctx-input gradient of step t+1 is still the slabs of its GEMM: this launch finishes them (mask, + d_out)
                check(lib.capmi_glu_bwd_add(ptr(d_out_all[t]), None, ws3.slabs.data_ptr() if ctx_splits else None, ctx_splits, N * R,
                                            ptr(self.m_ctx[t + 1]) if ctx_splits else None, ptr(self.pre2[t]), ptr(d_pre2_all[t]), N, R, st),
                      'glu_bwd_add')
                # [d_att | d_h_att] = d_pre2 Wc stays K-slice slabs of pitch 2R: the attention backward sums the first half while
                # staging d_o, the LayerNorm backward the second as the gradient it adds its own term to
                sp = ops.gemm([(d_pre2_all[t], 2 * R, Wc, 2 * R, 2 * R, 1)], N, 2 * R, ws.buf, a_layout=0, b_layout=1, ws=ws, defer_reduce=True)
                check(lib.capmi_mha_bwd_slabs(ws.slabs.data_ptr(), sp, N * 2 * R, 2 * R, ptr(self.q[t]), 0, self.p_att.data_ptr() + 4 * R,
                                              ptr(self.p_att), K * 2 * R, 2 * R, ptr(self.p_dec[t]), ptr(self.m_patt[t]), ptr(dq), 0,
                                              d_p_att.data_ptr() + 4 * R, ptr(d_p_att), K * 2 * R, 2 * R, 1, N, n, 1, K, h, R // h, st),
                      'mha_bwd_slabs')
                spq = ops.gemm([(dq, R, Wq, R, R, 1)], N, R, ws4.buf, a_layout=0, b_layout=1, ws=ws4, defer_reduce=True)
                dh = z(N, R)
                check(lib.capmi_layernorm_bwd_slabs(ws4.slabs.data_ptr(), spq, N * R, ptr(ln_dy[t]), ptr(self.h_att[t + 1]), ptr(a_n),
                                                    ptr(self.q_ln_mean[t]), ptr(self.q_ln_inv[t]), ws.slabs.data_ptr() + 4 * R, sp,
                                                    N * 2 * R, 2 * R, ptr(dh), ptr(ln_g[t]), N, R, EPS, st), 'layernorm_bwd_slabs')
            else:
                check(lib.capmi_glu_bwd(ptr(d_out_all[t]), None, ptr(self.pre2[t]), ptr(d_pre2_all[t]), N, R, st), 'glu_bwd')
                d_att, dh = dcat_halves(d_pre2_all[t], Wc, N, R)                          # [d_att | d_h_att] of [N,2R] = d_pre2 Wc
                # attention: dq, dK/dV accumulated into the two halves of d_p_att across rows of an image and across time
                check(lib.capmi_mha_bwd(ptr(d_att), ptr(self.q[t]), self.p_att.data_ptr() + 4 * R, ptr(self.p_att), K * 2 * R, 2 * R,
                                        ptr(self.p_dec[t]), ptr(self.m_patt[t]), ptr(dq), d_p_att.data_ptr() + 4 * R, ptr(d_p_att),
                                        K * 2 * R, 2 * R, 1, N, n, 1, K, h, R // h, st), 'mha_bwd')
                d_qn = ops.matmul_nn(dq_all[t], Wq, out=ln_dy[t])
                check(lib.capmi_layernorm_bwd(ptr(d_qn), ptr(self.h_att[t + 1]), ptr(a_n), ptr(self.q_ln_mean[t]), ptr(self.q_ln_inv[t]),
                                              ptr(dh), 1, ptr(ln_g[t]), N, R, EPS, st), 'layernorm_bwd')
            # LSTM cell: dh = (att2ctx + query path) + the slabs of d_gates(t+1) W_hh
            dc_prev = z(N, R)
            dh_b = None if dh_slabs is None else dh_slabs.data_ptr()
            if use_pl:
                check(lib.capmi_lstm_cell_bwd_partial_pl(ptr(dh), R, None, dh_b, R, max(dh_splits, 1), N * R, None, R, 1, 0,
                                                         ptr(dc_next), ptr(self.gates[t]), ptr(self.c_att[t]), ptr(self.c_att[t + 1]),
                                                         ptr(dg_all[t]), ptr(dc_prev), N, R, ptr(pl_dg), st), 'lstm_cell_bwd_partial_pl')
            else:
                check(lib.capmi_lstm_cell_bwd_partial(ptr(dh), R, None, dh_b, R, max(dh_splits, 1), N * R, None, R, 1, 0,
                                                      ptr(dc_next), ptr(self.gates[t]), ptr(self.c_att[t]), ptr(self.c_att[t + 1]),
                                                      ptr(dg_all[t]), ptr(dc_prev), N, R, st), 'lstm_cell_bwd_partial')
            dc_next = dc_prev
            if t > 0:
                pl = [pl_dg] if use_pl else None
                if fuse:
                    ctx_splits = ops.gemm([(dg_all[t], 4 * R, (W_ih, E), ld_ih, 4 * R, 1)], N, R, ws3.buf, a_layout=0, b_layout=1, ws=ws3,
                                          defer_reduce=True, a_planes=pl)
                else:
                    ops.gemm([(dg_all[t], 4 * R, (W_ih, E), ld_ih, 4 * R, 1)], N, R, d_out_all[t - 1], a_layout=0, b_layout=1,
                             mul_mask=self.m_ctx[t], accumulate=True, a_planes=pl)
                dh_splits = ops.gemm([(dg_all[t], 4 * R, W_hh, R, 4 * R, 1)], N, R, ws2.buf, a_layout=0, b_layout=1, ws=ws2,
                                     defer_reduce=True, a_planes=pl)
                dh_slabs = ws2.slabs
        # ---- time-batched core gradients
        dg2 = dg_all.view(TN, 4 * R)
        _dw(dg2, self.xt.view(TN, E), g['core.att_lstm.weight_ih'], ldc=ld_ih, bias=g['core.att_lstm.bias_ih'])
        # columns E: of W_ih multiply (mean + ctx_in): ctx_in part time-batched, mean part through the per-image sum
        sum_dg = z(B, 4 * R)
        check(lib.capmi_group_rowsum(ptr(dg_all), T, N * 4 * R, B, n, 4 * R, ptr(sum_dg), st), 'group_rowsum')
        gW = g['core.att_lstm.weight_ih']
        # (the small per-image product is written first, the time-batched one is added to it -- possibly at the end of the backward)
        ops.gemm([(sum_dg, 4 * R, self.mean, R, B, 1)], 4 * R, R, (gW, E), ldc=ld_ih, a_layout=1, b_layout=1)
        _dw(dg2, self.ctx_in.view(TN, R), gW, ldc=ld_ih, off=E, accumulate=True)
        _dw(dg2, self.h_att[:T].reshape(TN, R), g['core.att_lstm.weight_hh'])
        self._bias_hh_copy = True            # bias_hh's gradient = bias_ih's: copied behind the grouped launch (backward())
        # (r5: the W_ih column blocks are read in place -- [K = 4R][N] operands of pitch E + R -- instead of through
        #  .contiguous() copies of 16 MB each: 2 copy launches, ~135 us per step)
        d_mean = z(B, R)
        ops.gemm([(sum_dg, 4 * R, (W_ih, E), ld_ih, 4 * R, 1)], B, R, d_mean, a_layout=0, b_layout=1)          # [B,R]
        # embedding
        d_xt = z(TN, E)
        ops.gemm([(dg2, 4 * R, W_ih, ld_ih, 4 * R, 1)], TN, E, d_xt, a_layout=0, b_layout=1)
        g['embed.0.weight'].zero_()
        masks_xt = self.m_xt_all
        check(lib.capmi_embed_bwd(ptr(self.it_all), ptr(d_xt), ptr(self.xt), ptr(masks_xt), ptr(g['embed.0.weight']), TN, E, 1, st),
              'embed_bwd')
        # attention query path
        _dw(dq_all.view(TN, R), self.qn.view(TN, R), g['core.attention.linears.0.weight'], bias=g['core.attention.linears.0.bias'])
        _colsum(ln_g.view(TN, R), g['core.attention.norm.a_2'])
        _colsum(ln_dy.view(TN, R), g['core.attention.norm.b_2'])
        # att2ctx
        gWc = g['core.att2ctx.0.weight']
        dp2 = d_pre2_all.view(TN, 2 * R)
        _dw(dp2, self.att_o.view(TN, R), gWc, ldc=2 * R, bias=g['core.att2ctx.0.bias'])
        _dw(dp2, self.h_att[1:].reshape(TN, R), gWc, ldc=2 * R, off=R)
        # ---- prefill backward
        d_att = self.ctx2att.bwd(d_p_att)                                              # [B*K,R]
        check(lib.capmi_meanpool_bwd(ptr(d_mean), ptr(self.att_masks), ptr(d_att), 1, B, K, R, st), 'meanpool_bwd')
        dx = torch.zeros(B * K, R, dtype=_f32, device=dev)
        self.ref_norm.bwd(d_att, dx)
        for lay in reversed(self.ref):
            pre = lay['pre']
            d_pre = glu_bwd(dx, lay['m_res'], lay['pre_act'])                          # residual path stays in dx
            W = P[pre + '.self_attn.aoa_layer.0.weight']
            gW2 = g[pre + '.self_attn.aoa_layer.0.weight']
            BK = B * K
            _dw(d_pre, lay['od'], gW2, ldc=2 * R, bias=g[pre + '.self_attn.aoa_layer.0.bias'])
            _dw(d_pre, lay['yd'], gW2, ldc=2 * R, off=R)
            d_o, d_y = dcat_halves(d_pre, W, BK, R, lay['m_o'], lay['m_y'])            # [d_od | d_yd] of [BK,2R] = d_pre W
            if lay['lqkv'] is not None:
                dqkv = torch.empty(BK, 3 * R, dtype=_f32, device=dev)
                mha_bwd(d_o.view(B, K, R), lay['q'], lay['k'], lay['v'], K * 3 * R, lay['p'], lay['dp'], B, 1, K, K, h, kstride=3 * R,
                        qstride=3 * R, dq=(dqkv, 0), dq_stride=3 * R, dk_out=(dqkv, R), dv_out=(dqkv, 2 * R), dkv_ld=K * 3 * R,
                        dkv_stride=3 * R)
                d_y = d_y + lay['lqkv'].bwd(dqkv, fresh=True)                   # one dW, one column sum, one dX (K = 3R)
            else:
                dq, dk, dv = mha_bwd(d_o.view(B, K, R), lay['q'], lay['k'], lay['v'], K * R, lay['p'], lay['dp'], B, 1, K, K, h)
                d_y = d_y + lay['lq'].bwd(dq.view(BK, R))
                d_y += lay['lk'].bwd(dk.view(BK, R))
                d_y += lay['lv'].bwd(dv.view(BK, R))
            lay['n0'].bwd(d_y, dx)
        self.embed.bwd(dx, need_dx=False)


# --------------------------------------------------------------------------- beam search (eval)
class BeamDecoder:
    """One AoA decoder step at a time for beam search (AoA_Decoder_Core, AoAModel.py:128-186, eval numerics): the
    state of every hypothesis is (h_att, c_att, previous output) = three [N, R] arrays stacked so that a beam reorder is
    one launch; the beam_size hypotheses of an image share its refined features (no repeat_tensors copy)."""

    def __init__(self, graph, rows_per_image_max):
        self.g = g = graph
        P, B, R = g.P, g.B, g.R
        self.B, self.R = B, R
        self.N = N = B * rows_per_image_max
        dev = g.dev
        self.V1, self.E = P['embed.0.weight'].shape
        z = lambda *s: torch.empty(*s, dtype=_f32, device=dev)       # noqa: E731
        self.z = z
        self.state = torch.zeros(3, N, R, dtype=_f32, device=dev)   # h_att, c_att, out (ctx of the previous step)
        self.state_alt = torch.empty_like(self.state)
        W_ih = P['core.att_lstm.weight_ih']
        self.mean_gates = z(B, 4 * R)                                # mean-feature term of the gates, once per image
        ops.gemm([(g.mean, R, (W_ih, self.E), self.E + R, R, 1)], B, 4 * R, self.mean_gates)
        self.logits = z(N, self.V1)

    def step(self, t, it, rows_per_image):
        g = self.g
        P, h, B, K, R, E, V1 = g.P, g.h, g.B, g.K, self.R, self.E, self.V1
        rows = B * rows_per_image
        z, st = self.z, stream_ptr()
        ws = ops.default_workspace(g.dev)
        W_ih, W_hh = P['core.att_lstm.weight_ih'], P['core.att_lstm.weight_hh']
        h_prev, c_prev, ctx_prev = self.state[0, :rows], self.state[1, :rows], self.state[2, :rows]
        new = self.state_alt
        xt = z(rows, E)
        check(lib.capmi_embed_fwd(ptr(it), 1, None, ptr(P['embed.0.weight']), None, ptr(xt), rows, E, 1, st), 'embed_fwd')
        splits = ops.gemm([(xt, E, W_ih, E + R, E, 1), (ctx_prev, R, (W_ih, E), E + R, R, 1), (h_prev, R, W_hh, R, R, 1)], rows,
                          4 * R, ws.buf, ws=ws, defer_reduce=True)
        check(lib.capmi_lstm_cell_fwd(ws.slabs.data_ptr(), splits, ptr(P['core.att_lstm.bias_ih']), ptr(P['core.att_lstm.bias_hh']),
                                      ptr(self.mean_gates), rows_per_image, None, ptr(c_prev), ptr(new[0]), ptr(new[1]), None, None,
                                      None, rows, R, st), 'lstm_cell_fwd')
        h_new = new[0, :rows]
        qn, mu, inv = z(rows, R), z(rows), z(rows)
        check(lib.capmi_layernorm_fwd(ptr(h_new), ptr(P['core.attention.norm.a_2']), ptr(P['core.attention.norm.b_2']), ptr(qn),
                                      ptr(mu), ptr(inv), rows, R, EPS, st), 'layernorm_fwd')
        q = z(rows, R)
        ops.gemm([(qn, R, P['core.attention.linears.0.weight'], R, R, 1)], rows, R, q, bias=P['core.attention.linears.0.bias'])
        att_o = z(rows, R)
        # keys = second half of p_att rows, values = first half (AoAModel.py:168); per image, stride 2R
        check(lib.capmi_mha_fwd(ptr(q), g.p_att.data_ptr() + 4 * R, ptr(g.p_att), K * 2 * R, 2 * R, ptr(g.smask), 1, 0, 0, 0, None,
                                ptr(att_o), None, rows, rows_per_image, 1, K, h, R // h, st), 'mha_fwd')
        Wc = P['core.att2ctx.0.weight']
        pre2 = z(rows, 2 * R)
        ops.gemm([(att_o, R, Wc, 2 * R, R, 1), (h_new, R, (Wc, R), 2 * R, R, 1)], rows, 2 * R, pre2, bias=P['core.att2ctx.0.bias'])
        check(lib.capmi_glu_fwd(ptr(pre2), None, None, ptr(new[2]), rows, R, st), 'glu_fwd')
        logits = self.logits[:rows]
        ops.gemm([(new[2, :rows], R, P['logit.weight'], R, R, 1)], rows, V1, logits, bias=P['logit.bias'])
        self.state, self.state_alt = new, self.state
        self._keep = (xt, qn, mu, inv, q, att_o, pre2)           # scratch stays alive until the stream has used it
        return logits

    def reorder(self, parent, cur):
        from . import beam
        beam.reorder_rows(self.state, self.state_alt, parent, self.B, cur, self.N // self.B)
        self.state, self.state_alt = self.state_alt, self.state


def sample_beam(model, P, att_feats, att_masks, h, L, opt):
    from . import beam
    g = AoAGraph(P, {}, h, 0.0, 0.0, False, 0)
    g.prepare(att_feats, att_masks)               # refined features are shared by every decoder (diverse groups)
    return beam.beam_search_steps(model, lambda rows: BeamDecoder(g, rows), g.B, P['embed.0.weight'].shape[0], L, opt,
                                  att_feats.device)
