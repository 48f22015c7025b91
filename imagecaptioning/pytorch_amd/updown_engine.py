"""Host-side driver of the UpDown decode path on libcapmi (MI355X).

Python here only owns device memory (torch tensors), fills the C structs of include/capmi.h and makes
ONE native call per rollout (forward) / per BPTT (backward); all arithmetic is in the HIP kernels.

Mirrors, for the UpDown model (AttModel.py:875-879 + UpDownCore 615-640):
  prepare()      AttModel._prepare_feature        AttModel.py:114-124
  Rollout.run()  AttModel._sample / _forward      AttModel.py:258-352 / 126-164
  Rollout.backward()  the autograd graph torch would have recorded for them
"""
import ctypes as C
import os
import math

import torch

from . import _lib, ops
from .ops import clip_len
from ._lib import lib, ptr, check, stream_ptr

_f32 = torch.float32

PARAM_KEYS = ('embed.0.weight', 'fc_embed.0.weight', 'fc_embed.0.bias', 'att_embed.0.weight', 'att_embed.0.bias',
              'ctx2att.weight', 'ctx2att.bias', 'core.att_lstm.weight_ih', 'core.att_lstm.weight_hh',
              'core.att_lstm.bias_ih', 'core.att_lstm.bias_hh', 'core.lang_lstm.weight_ih', 'core.lang_lstm.weight_hh',
              'core.lang_lstm.bias_ih', 'core.lang_lstm.bias_hh', 'core.attention.h2att.weight',
              'core.attention.h2att.bias', 'core.attention.alpha_net.weight', 'core.attention.alpha_net.bias',
              'logit.weight', 'logit.bias')

_W_FIELDS = (('embed', 'embed.0.weight'), ('att_w_ih', 'core.att_lstm.weight_ih'), ('att_w_hh', 'core.att_lstm.weight_hh'),
             ('att_b_ih', 'core.att_lstm.bias_ih'), ('att_b_hh', 'core.att_lstm.bias_hh'),
             ('lang_w_ih', 'core.lang_lstm.weight_ih'), ('lang_w_hh', 'core.lang_lstm.weight_hh'),
             ('lang_b_ih', 'core.lang_lstm.bias_ih'), ('lang_b_hh', 'core.lang_lstm.bias_hh'),
             ('h2att_w', 'core.attention.h2att.weight'), ('h2att_b', 'core.attention.h2att.bias'),
             ('alpha_w', 'core.attention.alpha_net.weight'), ('alpha_b', 'core.attention.alpha_net.bias'),
             ('logit_w', 'logit.weight'), ('logit_b', 'logit.bias'))


def weights_struct(P):
    w = _lib.UpDownWeights()
    for f, k in _W_FIELDS:
        t = P[k]
        if not (t.is_cuda and t.is_contiguous() and t.dtype == _f32):
            raise _lib.CapmiError('parameter %s must be a contiguous fp32 device tensor' % k)
        setattr(w, f, t.data_ptr())
    return w


class Prepared:
    """fc' [B,R], att' [B,K,R], p_att [B,K,A] (+ what the backward of the prefill needs)."""
    __slots__ = ('fc', 'att', 'p_att', 'att_masks', 'fc_in', 'att_in', 'drop_fc', 'drop_att', 'K')


def prepare(P, fc_feats, att_feats, att_masks=None, drop_fc=None, drop_att=None, ws=None, out=None):
    """AttModel._prepare_feature (AttModel.py:114-124): three MFMA GEMMs with fused bias/ReLU/dropout
    epilogues.  Padded regions (att_masks == 0) are zeroed like pad_packed_sequence does (44-49).
    out: optional (fc [B,R], att [B,K,R], p_att [B,K,A]) contiguous targets (slices of a caller's larger buffers)."""
    B = fc_feats.shape[0]
    if att_masks is not None:
        max_len = clip_len(att_masks)          # clip_att, AttModel.py:106-112
        att_feats = att_feats[:, :max_len].contiguous()
        att_masks = att_masks[:, :max_len].contiguous().float()
        if drop_att is not None:
            drop_att = drop_att[:, :max_len].contiguous()
    K = att_feats.shape[1]
    pr = Prepared()
    pr.K = K
    pr.fc_in, pr.att_in, pr.drop_fc = fc_feats.contiguous(), att_feats.contiguous(), drop_fc
    o_fc, o_att, o_patt = out if out is not None else (None, None, None)
    pr.fc = ops.linear(pr.fc_in, P['fc_embed.0.weight'], P['fc_embed.0.bias'], relu=True, mul_mask=drop_fc, ws=ws, out=o_fc)
    R = pr.fc.shape[1]
    att_mask_full = drop_att
    if att_masks is not None:
        m = att_masks.unsqueeze(-1).expand(B, K, R)
        att_mask_full = (m if drop_att is None else m * drop_att).contiguous()
    pr.drop_att = att_mask_full
    att2d = ops.linear(pr.att_in.view(B * K, -1), P['att_embed.0.weight'], P['att_embed.0.bias'], relu=True,
                       mul_mask=None if att_mask_full is None else att_mask_full.view(B * K, R), ws=ws,
                       out=None if o_att is None else o_att.view(B * K, R))
    pr.att = att2d.view(B, K, R)
    pr.p_att = ops.linear(att2d, P['ctx2att.weight'], P['ctx2att.bias'], ws=ws,
                          out=None if o_patt is None else o_patt.view(B * K, -1)).view(B, K, -1)
    pr.att_masks = att_masks
    return pr


def prepare_backward(P, pr, d_fc, d_att, d_p_att, grads, ws=None):
    """Backward of prepare(): fills grads[...] for fc_embed / att_embed / ctx2att (overwrite)."""
    B, K, R = pr.att.shape
    A = pr.p_att.shape[2]
    dp = d_p_att.view(B * K, A)
    att2d = pr.att.view(B * K, R)
    # r6: the three weight gradients (K = B * regions rows / B rows) with their bias gradients as ONE grouped launch at the end
    # (ops.gemm_group_tn: 144 tiles of one round instead of three sub-wave GEMMs + reductions + three column sums);
    # CAPMI_PREP_GROUP=0: one launch each, as in r5
    group = [] if os.environ.get('CAPMI_PREP_GROUP', '1') != '0' else None

    def dw(dy, x, wname, bname):
        if group is not None and grads[bname].data_ptr() % 16 == 0:
            group.append((dy, x, grads[wname], False, None, 0, grads[bname]))
        else:
            ops.matmul_tn(dy, x, out=grads[wname], ws=ws)
            ops.colsum(dy, out=grads[bname])
    # ctx2att: p_att = att W^T + b
    dw(dp, att2d, 'ctx2att.weight', 'ctx2att.bias')
    d_att_total = d_att.view(B * K, R)
    ops.gemm([(dp, A, P['ctx2att.weight'], R, A, 1)], B * K, R, d_att_total, a_layout=0, b_layout=1, accumulate=True, ws=ws)
    # att_embed: att = drop(relu(x W^T + b))
    # relu gate: pre-activation > 0  <=>  relu output > 0; with dropout the saved output may be zero for kept
    # units only if relu clipped, and for dropped units the mask already zeroes the gradient.
    d_pre = _relu_drop_bwd(d_att_total, att2d, None if pr.drop_att is None else pr.drop_att.view(B * K, R))
    dw(d_pre, pr.att_in.view(B * K, -1), 'att_embed.0.weight', 'att_embed.0.bias')
    d_pre_fc = _relu_drop_bwd(d_fc, pr.fc, pr.drop_fc)
    dw(d_pre_fc, pr.fc_in, 'fc_embed.0.weight', 'fc_embed.0.bias')
    if group:
        ops.gemm_group_tn(group, ws=ws, cache_key=('updown_prepare', str(dp.device)))


def _relu_drop_bwd(dy, y_saved, mask):
    """dx = dy * mask * [y_saved > 0]; y_saved = relu(pre)*mask.  A unit with y_saved == 0 was either
    clipped by the ReLU (gradient 0) or dropped (mask 0 => gradient 0)."""
    return ops.relu_mask_bwd(dy.contiguous(), y_saved.contiguous(), mask)


_alive = {}


def _alive_buffer(dev, L):
    """[L] int32 of pinned host memory the select kernels can write (capmi.h capmi_updown_rollout.alive_host), one per stream:
    the words are only ever SET by kernels and cleared by the host before a rollout is enqueued, so a late store of the
    previous rollout can at worst postpone an early exit."""
    key = (str(dev), stream_ptr(), L)
    t = _alive.get(key)
    if t is None:
        t = _alive[key] = torch.zeros(max(L, 32), dtype=torch.int32).pin_memory()
    return t


class Rollout:
    """Device buffers + one native call for a T-step rollout of N = B*n caption rows."""

    def __init__(self, P, pr, n, T, L=None, mode='greedy', temperature=1.0, drop_xt=None, drop_out=None,
                 gumbel=None, seed=0, forced=None, teacher=False, row_mode=None, ws=None, keep_for_backward=True,
                 row_img=None, B_grad=None, top_k=0, top_p=0.0, ss_mode=None, early_exit=None, early_exit_from=4, raw_logits=False):
        """ss_mode (uint8 [T,N], teacher only): scheduled sampling, 1 = the input of (step, row) is drawn from the previous
        step's distribution, 2 = teacher-forced (capmi.h capmi_updown_rollout.ss_mode).
        row_img (int32 [N]) + B_grad: ragged grouping for the fused SCST rollout -- the first B_grad
        feature images own rows b*n..b*n+n-1 (sampled, with gradient), the remaining rows (greedy baseline)
        point at further feature images through row_img."""
        dev = pr.fc.device
        B_feat, K, R = pr.att.shape
        A = pr.p_att.shape[2]
        V1, E = P['embed.0.weight'].shape
        B = B_feat if B_grad is None else B_grad
        N = B * n if row_img is None else row_img.shape[0]
        self.row_img = row_img
        L = T if L is None else L
        self.P, self.pr, self.dims = P, pr, (B, n, N, K, A, R, E, V1, T, L)
        self.ws = ws or ops.default_workspace(dev)
        z = lambda *s: torch.empty(*s, dtype=_f32, device=dev)       # noqa: E731
        self.h_att, self.c_att, self.h_lang, self.c_lang = (z(T + 1, N, R) for _ in range(4))
        self.xt = z(T, N, E)
        self.it_all = torch.empty(T, N, dtype=torch.long, device=dev)
        self.gates_att, self.gates_lang = z(T, N, 4 * R), z(T, N, 4 * R)
        self.att_h, self.alpha, self.ctx, self.h_drop = z(T, N, A), z(T, N, K), z(T, N, R), z(T, N, R)
        # the select kernel writes every (row, step < T) slot of seq / seq_logp / sel_logp / live, zeros included, and the driver
        # clears the tail behind an early exit: fills are only needed when fewer steps than the pitch are run
        zl = torch.empty if T == L else torch.zeros
        self.seq = zl(N, L, dtype=torch.long, device=dev)
        # the select kernel writes every (row, step < T) slice of the dense log-probs, zeros included: a 45 MB memset per
        # rollout is only needed when fewer steps than the pitch are run (XE with an early all-pad column)
        self.seq_logp = (torch.empty if T == L else torch.zeros)(N, L, V1, dtype=_f32, device=dev)
        self.sel_logp = zl(N, L, dtype=_f32, device=dev)
        self.live = zl(N, L, dtype=torch.uint8, device=dev)
        self.fc_gates = z(B_feat, 4 * R)
        self.logits = z(N, V1)
        self.it = torch.empty(N, dtype=torch.long, device=dev)
        self.unfinished = torch.empty(N, dtype=torch.uint8, device=dev)
        self.drop_xt, self.drop_out, self.gumbel, self.forced, self.row_mode = drop_xt, drop_out, gumbel, forced, row_mode

        r = _lib.UpDownRollout()
        r.B, r.n, r.N, r.K, r.A, r.R, r.E, r.V1, r.T, r.L = B, n, N, K, A, R, E, V1, T, L
        r.B_feat, r.row_img = B_feat, ptr(row_img)
        r.fc, r.att, r.p_att, r.att_mask = ptr(pr.fc), ptr(pr.att), ptr(pr.p_att), ptr(pr.att_masks)
        r.drop_xt, r.drop_out = ptr(drop_xt), ptr(drop_out)
        r.mode = {'greedy': 0, 'sample': 1, 'forced': 2}[mode]
        r.row_mode = ptr(row_mode)
        r.temperature = float(temperature)
        r.top_k, r.top_p = int(top_k), float(top_p)
        r.gumbel = ptr(gumbel)
        r.seed = int(seed) & 0xFFFFFFFFFFFFFFFF
        if forced is not None:
            assert forced.dtype == torch.long and forced.is_contiguous()
            r.forced, r.forced_ld = ptr(forced), forced.shape[1]
        r.teacher = int(teacher)
        if ss_mode is not None:
            assert teacher and ss_mode.dtype == torch.uint8 and ss_mode.shape == (T, N) and ss_mode.is_contiguous()
        self.ss_mode = ss_mode
        r.ss_mode = ptr(ss_mode)
        for k in ('h_att', 'c_att', 'h_lang', 'c_lang', 'xt', 'it_all', 'gates_att', 'gates_lang', 'att_h', 'alpha',
                  'ctx', 'h_drop', 'seq', 'seq_logp', 'sel_logp', 'live', 'fc_gates', 'logits', 'it', 'unfinished'):
            setattr(r, k, getattr(self, k).data_ptr())
        r.partial, r.partial_capacity = self.ws.buf.data_ptr(), self.ws.capacity
        if N <= 64 and os.environ.get('CAPMI_PLANES', '1') != '0':
            # decode GEMMs stage their activations as producer-written bf16x3 planes (capmi.h capmi_updown_rollout.planes);
            # the scratch is zero-filled once and shared by the rollouts of this stream with the same R / E
            nb = int(lib.capmi_updown_planes_bytes(R, E))
            self.planes = ops.planes_scratch(dev, ('updown_fwd', R, E), nb)
            r.planes, r.planes_bytes = self.planes.data_ptr(), nb
            if not teacher and os.environ.get('CAPMI_FUSED_SELECT', '1') != '0':
                # r4: slab workspace of the part of the next step's attention-LSTM gate GEMM that is computed inside the select
                # launch (capmi.h capmi_updown_rollout.pre_partial): [ticket words | up to 8 K-slice slabs of N x 4R]
                per = ops.Workspace.COUNTER_FLOATS + 8 * 64 * 4 * R
                self.pre = ops.planes_scratch(dev, ('updown_pre', R), per * 4).view(torch.float32)
                r.pre_partial, r.pre_capacity = self.pre.data_ptr(), per
        # early exit of free-running rollouts (AttModel.py:349-350): behind steps early_exit_from + k * early_exit - 1 (7, 11, 15 by
        # default) the driver looks, two steps later, at a pinned word the select kernels set and stops enqueuing once every row
        # has emitted its EOS (CAPMI_EARLY_EXIT=0: never)
        if early_exit is None:
            early_exit = int(os.environ.get('CAPMI_EARLY_EXIT', '4'))
        self.steps_run = T
        if early_exit > 0 and not teacher and mode != 'forced' and T >= 12:
            self.alive = _alive_buffer(dev, L)
            r.early_exit, r.early_exit_from, r.alive_host = int(early_exit), int(early_exit_from), self.alive.data_ptr()
        r.raw_logits = int(bool(raw_logits) and not teacher)     # AttModel._sample(output_logsoftmax=0): logits, not log-probs
        self.r = r
        self.T_cfg = int(r.T)
        self.w = weights_struct(P)

    def run(self):
        self.r.T = self.T_cfg                # (a previous run on this object may have ended early: every run starts from the configured T)
        check(lib.capmi_updown_rollout_fwd(C.byref(self.w), C.byref(self.r), stream_ptr()), 'capmi_updown_rollout_fwd')
        self.steps_run = int(self.r.steps_run)
        self.r.T = self.steps_run            # the backward runs over the steps that were enqueued
        return self.seq, self.seq_logp

    # backward phases in launch order with the parameter gradients each one completes (capmi.h CAPMI_BWD_*)
    BWD_PHASES = ((1, ('logit.weight', 'logit.bias')),
                  (2, ()),
                  (4, ('core.lang_lstm.weight_ih', 'core.lang_lstm.weight_hh', 'core.lang_lstm.bias_ih',
                       'core.lang_lstm.bias_hh')),
                  (8, ('core.att_lstm.weight_ih', 'core.att_lstm.weight_hh', 'core.att_lstm.bias_ih',
                       'core.att_lstm.bias_hh', 'embed.0.weight')),
                  (16, ('core.attention.h2att.weight', 'core.attention.h2att.bias', 'core.attention.alpha_net.weight',
                        'core.attention.alpha_net.bias')))

    def backward(self, g_seq_logp, grads, on_ready=None, sparse=None):
        """g_seq_logp [N,L,V1] (None when `sparse`, a _lib.SparseLogpGrad, carries the loss gradient).  grads: dict name -> preallocated fp32 tensor (overwritten) for every
        PARAM_KEYS entry.  Also returns (d_fc, d_att, d_p_att) consumed by prepare_backward.
        on_ready(names): called after the launches that complete the gradients `names` have been enqueued, so a
        data-parallel trainer can start reducing that bucket while the later phases still run."""
        B, n, N, K, A, R, E, V1, T, L = self.dims
        dev = self.seq.device
        z = lambda *s: torch.empty(*s, dtype=_f32, device=dev)       # noqa: E731
        s = _lib.UpDownBwdScratch()
        keep = dict(dlogits=z(T, N, V1), d_hdrop=z(T, N, R), dg_att=z(T, N, 4 * R), dg_lang=z(T, N, 4 * R),
                    d_x2=z(T, N, 3 * R), d_e_all=z(T, N, K), d_att_h_all=z(T, N, A), dh_att_attn=z(N, R),
                    d_x1=z(4), dc_att=z(2, N, R), dc_lang=z(2, N, R), d_xt_all=z(T, N, E),
                    sum_dg_att=z(B, 4 * R), w_lang_cat=z(4 * R, 3 * R), w_att_cat=z(4 * R, 2 * R))
        for k, t in keep.items():
            setattr(s, k, t.data_ptr())
        s.partial, s.partial_capacity = self.ws.buf.data_ptr(), self.ws.capacity
        if self.row_img is not None and B * n < N and os.environ.get('CAPMI_BWD_ALL_ROWS') != '1':
            # fused SCST rollout: rows [B*n, N) are the greedy baseline (eval mode, no gradient) -- the backward runs on the
            # sampled rows only and packs the saved activations once for its time-batched GEMMs
            nb = B * n
            keep['pack'] = z(nb * (T * (4 * R + 2 * E + 2 + A + K) + R) + 64)
            s.n_grad_rows, s.pack, s.pack_capacity = nb, keep['pack'].data_ptr(), keep['pack'].numel()
        if (s.n_grad_rows or N) <= 64 and os.environ.get('CAPMI_PLANES', '1') != '0':
            nb = int(lib.capmi_updown_bwd_planes_bytes(R))
            keep['planes'] = ops.planes_scratch(dev, ('updown_bwd', R), nb)
            s.planes, s.planes_bytes = keep['planes'].data_ptr(), nb
        g = _lib.UpDownGrads()
        for f, k in _W_FIELDS:
            setattr(g, f, grads[k].data_ptr())
        d_fc, d_att, d_p_att = z(B, R), z(B, K, R), z(B, K, A)
        g.d_fc, g.d_att, g.d_p_att = d_fc.data_ptr(), d_att.data_ptr(), d_p_att.data_ptr()
        g_seq_logp = None if g_seq_logp is None else g_seq_logp.contiguous()
        if sparse is not None:                    # the loss gradient in sparse form (sparse_logp.split_grad)
            s.sparse = C.pointer(sparse)
        if on_ready is None:
            check(lib.capmi_updown_rollout_bwd(C.byref(self.w), C.byref(self.r), ptr(g_seq_logp), C.byref(s), C.byref(g),
                                               stream_ptr()), 'capmi_updown_rollout_bwd')
        else:
            for mask, names in self.BWD_PHASES:
                check(lib.capmi_updown_rollout_bwd_phases(C.byref(self.w), C.byref(self.r), ptr(g_seq_logp), C.byref(s),
                                                          C.byref(g), mask, stream_ptr()), 'capmi_updown_rollout_bwd_phases')
                if names:
                    on_ready(names)
        self._bwd_keep = keep     # keep scratch alive until the stream has consumed it
        return d_fc, d_att, d_p_att
