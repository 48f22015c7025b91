"""Host-side driver of the NewFC decoder (configs/fc.yml) on libcapmi: buffers + one native call per rollout."""
import ctypes as C

import torch

from . import _lib, ops
from ._lib import lib, ptr, check, stream_ptr

_f32 = torch.float32
_W = (('embed', 'embed.weight'), ('i2h_w', '_core.i2h.weight'), ('i2h_b', '_core.i2h.bias'), ('h2h_w', '_core.h2h.weight'),
      ('h2h_b', '_core.h2h.bias'), ('logit_w', 'logit.weight'), ('logit_b', 'logit.bias'))


class Rollout:
    def __init__(self, P, fc_feats, n, T, L=None, mode='greedy', temperature=1.0, drop_out=None, gumbel=None, seed=0,
                 forced=None, teacher=False, ws=None, raw=False):
        # raw (free-running rollouts, r5): the stored rows are the LOGITS (AttModel._sample(output_logsoftmax=0), AttModel.py:171-175,
        # 265; CAPMI_SELECT_RAW in the select's mode), the backward takes the loss gradient as d(logits)
        dev = fc_feats.device
        B = fc_feats.shape[0]
        V1, E = P['embed.weight'].shape
        R = P['_core.h2h.weight'].shape[1]
        N = B * n
        L = T if L is None else L
        self.P, self.dims, self.fc_in = P, (B, n, N, R, E, V1, T, L), fc_feats
        self.ws = ws or ops.default_workspace(dev)
        # fc_embed is a plain Linear (AttModel.py:907)
        self.fc_emb = ops.linear(fc_feats, P['fc_embed.weight'], P['fc_embed.bias'], ws=self.ws)
        z = lambda *s: torch.empty(*s, dtype=_f32, device=dev)          # noqa: E731
        self.h, self.c = z(T + 2, N, R), z(T + 2, N, R)
        self.x, self.saved, self.h_drop = z(T, N, E), z(T + 1, N, 5 * R), z(T, N, R)
        self.it_all = torch.empty(T, N, dtype=torch.long, device=dev)
        self.seq = torch.zeros(N, L, dtype=torch.long, device=dev)
        self.seq_logp = torch.zeros(N, L, V1, dtype=_f32, device=dev)
        self.sel_logp = torch.zeros(N, L, dtype=_f32, device=dev)
        self.live = torch.zeros(N, L, dtype=torch.uint8, device=dev)
        self.logits = z(N, V1)
        self.it = torch.empty(N, dtype=torch.long, device=dev)
        self.unfinished = torch.empty(N, dtype=torch.uint8, device=dev)
        self.drop_out, self.gumbel, self.forced = drop_out, gumbel, forced
        r = _lib.NewFCRollout()
        r.B, r.n, r.N, r.R, r.E, r.V1, r.T, r.L = B, n, N, R, E, V1, T, L
        r.fc_emb, r.drop_out = ptr(self.fc_emb), ptr(drop_out)
        r.mode = {'greedy': 0, 'sample': 1, 'forced': 2}[mode] | (_lib.SELECT_RAW if (raw and not teacher) else 0)
        r.temperature, r.gumbel, r.seed = float(temperature), ptr(gumbel), int(seed) & 0xFFFFFFFFFFFFFFFF
        if forced is not None:
            r.forced, r.forced_ld = ptr(forced), forced.shape[1]
        r.teacher = int(teacher)
        for k in ('h', 'c', 'x', 'it_all', 'saved', 'h_drop', 'seq', 'seq_logp', 'sel_logp', 'live', 'logits', 'it',
                  'unfinished'):
            setattr(r, k, getattr(self, k).data_ptr())
        r.partial, r.partial_capacity = self.ws.buf.data_ptr(), self.ws.capacity
        self.r = r
        w = _lib.NewFCWeights()
        for f, k in _W:
            setattr(w, f, P[k].data_ptr())
        self.w = w

    def run(self):
        check(lib.capmi_newfc_rollout_fwd(C.byref(self.w), C.byref(self.r), stream_ptr()), 'capmi_newfc_rollout_fwd')
        return self.seq, self.seq_logp

    def backward(self, g_seq_logp, grads, sparse=None):
        B, n, N, R, E, V1, T, L = self.dims
        dev = self.seq.device
        z = lambda *s: torch.empty(*s, dtype=_f32, device=dev)          # noqa: E731
        keep = dict(dlogits=z(T, N, V1), d_hdrop=z(T, N, R), d_sums=z(T + 1, N, 5 * R), dh_prev=z(2, N, R), dc=z(2, N, R),
                    d_x_all=z(max(T * N * E, B * 5 * R)), d_ximg=z(N, E))
        s = _lib.NewFCBwdScratch()
        for k, t in keep.items():
            setattr(s, k, t.data_ptr())
        s.partial, s.partial_capacity = self.ws.buf.data_ptr(), self.ws.capacity
        g = _lib.NewFCGrads()
        for f, k in _W:
            setattr(g, f, grads[k].data_ptr())
        d_fc_emb = z(B, E)
        g.d_fc_emb = d_fc_emb.data_ptr()
        g_seq_logp = None if g_seq_logp is None else g_seq_logp.contiguous()
        if sparse is not None:
            s.sparse = C.pointer(sparse)
        check(lib.capmi_newfc_rollout_bwd(C.byref(self.w), C.byref(self.r), ptr(g_seq_logp), C.byref(s), C.byref(g),
                                          stream_ptr()), 'capmi_newfc_rollout_bwd')
        # fc_embed (plain Linear) backward
        ops.matmul_tn(d_fc_emb, self.fc_in, out=grads['fc_embed.weight'], ws=self.ws)
        ops.colsum(d_fc_emb, out=grads['fc_embed.bias'])
        self._keep = (keep, d_fc_emb, g_seq_logp)
