"""Single-step decoders ("steppers") for the LSTM families, and the reference's step API on top of them.

A stepper holds the recurrent state of ``B * rows_per_image_max`` hypotheses (image-major rows) and exposes

    step(t, it, rows_per_image) -> logits [B*rows_per_image, V1]      # AttModel.get_logprobs_state up to the logits
    reorder(parent [B,bd] int32, cur)                                 # beam search: row b*cur+parent[b,j] -> b*bd+j
    snapshot() / restore(s)                                           # (scheduled host loops that fork the state)

which is the protocol the Transformer ``Decoder`` and the AoA ``BeamDecoder`` already implement; the host-stepped
samplers of ``decode.py`` / ``beam.py`` (constrained decoding, diverse sampling, diverse beam search) drive any of them.
The fast paths (one native call per rollout / per beam search) do not go through here.

  UpDownStepper   capmi_updown_decode_step  (UpDownCore.forward, AttModel.py:615-640, eval numerics)
  NewFCStepper    maxout LSTMCore.forward   (FCModel.py:13-42 via AttModel.py:904-945)
"""
import ctypes as C

import torch

from . import _lib, ops, updown_engine as engine
from ._lib import lib, ptr, check, stream_ptr

_f32 = torch.float32


class UpDownStepper:
    def __init__(self, P, pr, rows_per_image_max):
        self.P, self.pr = P, pr
        dev = pr.att.device
        B, K, R = pr.att.shape
        A = pr.p_att.shape[2]
        V1, E = P['embed.0.weight'].shape
        self.B, self.R, self.V1, self.cap = B, R, V1, int(rows_per_image_max)
        self.N = N = B * self.cap
        z = lambda *s: torch.empty(*s, dtype=_f32, device=dev)       # noqa: E731
        self.ws = ops.default_workspace(dev)
        self.state = torch.zeros(2, 4, N, R, dtype=_f32, device=dev)   # ping-pong of (h_att, c_att, h_lang, c_lang)
        self.cur = 0
        self.bufs = dict(xt=z(N, E), gates=z(N, 4 * R), att_h=z(N, A), alpha=z(N, K), ctx=z(N, R), fc_gates=z(B, 4 * R),
                         logits=z(N, V1), it=torch.zeros(N, dtype=torch.long, device=dev))
        b = _lib.UpDownBeam()
        b.B, b.bd, b.K, b.A, b.R, b.E, b.V1, b.L = B, self.cap, K, A, R, E, V1, 0
        b.fc, b.att, b.p_att, b.att_mask = ptr(pr.fc), ptr(pr.att), ptr(pr.p_att), ptr(pr.att_masks)
        b.temperature, b.unk_col = 1.0, -1
        for k, t in self.bufs.items():
            setattr(b, k, t.data_ptr())
        b.partial, b.partial_capacity = self.ws.buf.data_ptr(), self.ws.capacity
        self.b, self.w = b, engine.weights_struct(P)
        self._first = True

    def step(self, t, it, rows_per_image):
        rows = self.B * rows_per_image
        assert it.shape[0] == rows and rows_per_image <= self.cap
        self.bufs['it'][:rows].copy_(it)
        src, dst = self.state[self.cur], self.state[1 - self.cur]
        check(lib.capmi_updown_decode_step(C.byref(self.w), C.byref(self.b), rows, rows_per_image, ptr(src), ptr(dst),
                                           1 if self._first else 0, stream_ptr()), 'capmi_updown_decode_step')
        self._first = False
        self.cur = 1 - self.cur
        return self.bufs['logits'][:rows]

    def reorder(self, parent, cur):
        from . import beam
        src, dst = self.state[self.cur], self.state[1 - self.cur]
        beam.reorder_rows(src, dst, parent, self.B, cur, parent.shape[1])
        self.cur = 1 - self.cur

    # reference layout of the recurrent state: (h [2,N,R], c [2,N,R]) with index 0 = att_lstm, 1 = lang_lstm
    def load_state(self, state, rows):
        h, c = state
        s = self.state[self.cur]
        s[0, :rows], s[1, :rows], s[2, :rows], s[3, :rows] = h[0], c[0], h[1], c[1]

    def export_state(self, rows):
        s = self.state[self.cur]
        return (torch.stack([s[0, :rows], s[2, :rows]]), torch.stack([s[1, :rows], s[3, :rows]]))


class NewFCStepper:
    """AttModel.py:925-936: the first call feeds the image (state all zero), then words.  The image step is taken in the
    constructor so that step(0, BOS) is the first WORD step like for every other family."""

    def __init__(self, P, fc_feats, rows_per_image_max):
        dev = fc_feats.device
        self.P = P
        self.B = B = fc_feats.shape[0]
        self.V1, self.E = P['embed.weight'].shape
        self.R = R = P['_core.h2h.weight'].shape[1]
        self.cap = int(rows_per_image_max)
        self.N = N = B * self.cap
        self.ws = ops.default_workspace(dev)
        self.state = torch.zeros(2, 2, N, R, dtype=_f32, device=dev)   # ping-pong of (h, c)
        self.cur = 0
        self.saved = torch.empty(N, 5 * R, dtype=_f32, device=dev)
        self.logits = torch.empty(N, self.V1, dtype=_f32, device=dev)
        fc_emb = ops.linear(fc_feats.float().contiguous(), P['fc_embed.weight'], P['fc_embed.bias'], ws=self.ws)
        self._cell(fc_emb, B)                                           # one row per image, cur = 1 afterwards

    def _cell(self, x, rows):
        P, R, E = self.P, self.R, self.E
        src, dst = self.state[self.cur], self.state[1 - self.cur]
        splits = ops.gemm([(x, E, P['_core.i2h.weight'], E, E, 1), (src[0, :rows], R, P['_core.h2h.weight'], R, R, 1)], rows,
                          5 * R, self.ws.buf, ws=self.ws, defer_reduce=True)
        check(lib.capmi_maxout_cell_fwd(self.ws.slabs.data_ptr(), splits, ptr(P['_core.i2h.bias']), ptr(P['_core.h2h.bias']),
                                        ptr(src[1]), ptr(dst[0]), ptr(dst[1]), ptr(self.saved), None, None, rows, R,
                                        stream_ptr()), 'capmi_maxout_cell_fwd')
        self.cur = 1 - self.cur
        self._keep = x

    def step(self, t, it, rows_per_image):
        rows = self.B * rows_per_image
        if t == 0 and rows_per_image > 1:
            # the image step left one row per image: fan it out (AttModel._sample repeats the features BEFORE the image
            # step, same values)
            idx = torch.arange(rows, device=it.device) // rows_per_image
            s = self.state[self.cur]
            s[:, :rows] = s[:, :self.B][:, idx]
        x = ops.embed_fwd(it, self.P['embed.weight'], relu=False)      # plain Embedding (AttModel.py:908)
        self._cell(x, rows)
        h = self.state[self.cur][0, :rows]
        logits = self.logits[:rows]
        ops.gemm([(h, self.R, self.P['logit.weight'], self.R, self.R, 1)], rows, self.V1, logits, bias=self.P['logit.bias'],
                 ws=self.ws)
        return logits

    def reorder(self, parent, cur):
        from . import beam
        src, dst = self.state[self.cur], self.state[1 - self.cur]
        beam.reorder_rows(src, dst, parent, self.B, cur, parent.shape[1])
        self.cur = 1 - self.cur


def updown_step(model, it, fc_feats, att_feats, p_att_feats, att_masks, state, output_logsoftmax=1):
    """AttModel.get_logprobs_state (AttModel.py:166-176) for callers that drive the decoder themselves (ensembles, custom
    searches): features are per ROW (already repeated by the caller, as in the reference), state = (h [2,N,R], c [2,N,R]).
    Eval numerics; returns (logprobs [N,V1], new state)."""
    model._device_check(fc_feats)
    P = {k: v.detach() for k, v in model.named_parameters()}
    pr = engine.Prepared()
    pr.fc, pr.att, pr.p_att = fc_feats.float().contiguous(), att_feats.float().contiguous(), p_att_feats.float().contiguous()
    pr.att_masks = None if att_masks is None else att_masks.float().contiguous()
    N = pr.fc.shape[0]
    st = UpDownStepper(P, pr, 1)
    st.load_state(state, N)
    logits = st.step(0, it.long().contiguous(), 1)
    new_state = st.export_state(N)
    if not output_logsoftmax:
        return logits.clone(), new_state
    logp = torch.empty_like(logits)
    check(lib.capmi_log_softmax_rows(ptr(logits), ptr(logp), N, st.V1, stream_ptr()), 'capmi_log_softmax_rows')
    return logp, new_state
