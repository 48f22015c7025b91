"""Beam search for the UpDown model on libcapmi: ONE native call runs all L steps on the device
(capmi_updown_beam_search); the host then reads the small [L,B,b] parent/token/score tables once and
assembles ``done_beams`` exactly as CaptionModel.beam_search does (CaptionModel.py:183-209), group_size 1.
"""
import ctypes as C

import torch

from . import _lib, ops, updown_engine as engine
from ._lib import lib, ptr, check, stream_ptr

_f32 = torch.float32


def _penalty(cfg):
    """captioning/utils/misc.py:133-157 penalty_builder."""
    if cfg == '':
        return lambda length, logp: logp
    kind, alpha = cfg.split('_')
    alpha = float(alpha)
    if kind == 'wu':
        return lambda length, logp: logp / (((5 + length) ** alpha) / ((5 + 1) ** alpha))
    if kind == 'avg':
        return lambda length, logp: logp / length
    raise ValueError(cfg)


def sample_beam(model, fc_feats, att_feats, att_masks, opt):
    """AttModel._sample_beam (AttModel.py:218-256): returns (seq [B*sample_n, L], seqLogprobs [.., L, V1]) and
    sets model.done_beams[k] = [{'seq','logps','unaug_p','p'}, ...] sorted by descending p."""
    beam_size = opt.get('beam_size', 10)
    group_size = opt.get('group_size', 1)
    sample_n = opt.get('sample_n', 10)
    if group_size != 1:
        raise NotImplementedError('diverse beam search (group_size > 1) is outside the hot-path scope')
    for k in ('decoding_constraint', 'remove_bad_endings'):
        if opt.get(k, 0):
            raise NotImplementedError('%s is not part of the accelerated beam search' % k)
    assert sample_n == 1 or sample_n == beam_size // group_size, 'when beam search, sample_n == 1 or beam search'
    model._device_check(fc_feats)
    P = {k: v.detach() for k, v in model.named_parameters()}
    pr = engine.prepare(P, fc_feats.float().contiguous(), att_feats.float().contiguous(),
                        None if att_masks is None else att_masks.float())
    dev = fc_feats.device
    B, K, R = pr.att.shape
    A = pr.p_att.shape[2]
    V1, E = P['embed.0.weight'].shape
    L, bd = model.seq_length, beam_size
    assert bd <= V1
    N = B * bd
    ws = ops.default_workspace(dev)
    z = lambda *s: torch.empty(*s, dtype=_f32, device=dev)       # noqa: E731
    bufs = dict(state=z(2, 4, N, R), xt=z(N, E), gates=z(N, 4 * R), att_h=z(N, A), alpha=z(N, K), ctx=z(N, R),
                fc_gates=z(B, 4 * R), logits=z(N, V1), it=torch.empty(N, dtype=torch.long, device=dev), sums=z(2, B, bd),
                logp_rows=z(L, N, V1), parent=torch.empty(L, B, bd, dtype=torch.int32, device=dev),
                token=torch.empty(L, B, bd, dtype=torch.long, device=dev), score=z(L, B, bd),
                ended=torch.empty(L, B, bd, dtype=torch.uint8, device=dev))
    b = _lib.UpDownBeam()
    b.B, b.bd, b.K, b.A, b.R, b.E, b.V1, b.L = B, bd, K, A, R, E, V1, L
    b.fc, b.att, b.p_att, b.att_mask = ptr(pr.fc), ptr(pr.att), ptr(pr.p_att), ptr(pr.att_masks)
    b.temperature = float(opt.get('temperature', 1))
    unk = -1
    if opt.get('suppress_UNK', 0) and hasattr(model, 'vocab') and model.vocab.get(str(V1 - 1)) == 'UNK':
        unk = V1 - 1                                              # CaptionModel.py:159-160
    elif getattr(model, 'unk_idx', None) is not None:
        unk = int(model.unk_idx)                                  # :161-162
    b.unk_col = unk
    for k, t in bufs.items():
        setattr(b, k, t.data_ptr())
    b.partial, b.partial_capacity = ws.buf.data_ptr(), ws.capacity
    w = engine.weights_struct(P)
    check(lib.capmi_updown_beam_search(C.byref(w), C.byref(b), stream_ptr()), 'capmi_updown_beam_search')

    return assemble_done_beams(model, bufs['parent'], bufs['token'], bufs['score'], bufs['ended'], bufs['logp_rows'], B, bd, L, V1,
                               sample_n, beam_size, opt)


def assemble_done_beams(model, parent, token, score, ended, logp_rows, B, bd, L, V1, sample_n, beam_size, opt):
    """One device->host transfer of the small [L,B,bd] tables, then the bookkeeping of CaptionModel.py:183-209."""
    dev = logp_rows.device
    N = B * bd
    parent, token, score, ended = (t.cpu().numpy() for t in (parent, token, score, ended))
    penalty = _penalty(opt.get('length_penalty', ''))
    seq = torch.zeros(B * sample_n, L, dtype=torch.long, device=dev)
    seq_logp = torch.zeros(B * sample_n, L, V1, dtype=_f32, device=dev)
    done_beams = []
    gather_idx, gather_meta = [], []
    for k in range(B):
        fin = []
        for t in range(L):
            for j in range(bd):
                if ended[t, k, j]:
                    fin.append((penalty(t + 1, float(score[t, k, j])), t, j))
        fin = sorted(fin, key=lambda x: -x[0])[:bd]            # stable, like the reference's sorted()
        beams = []
        for p, t, j in fin:
            toks, rows = [], []
            jj = j
            for s in range(t, -1, -1):
                toks.append(int(token[s, k, jj]))
                par = int(parent[s, k, jj])
                rows.append(s * N + (k if s == 0 else k * bd + par))   # step-0 rows are one per image
                jj = par
            toks.reverse()
            rows.reverse()
            beams.append({'seq': torch.tensor(toks, dtype=torch.long, device=dev), 'p': p, '_rows': rows})
            gather_idx.extend(rows)
            gather_meta.append((k, len(beams) - 1, len(rows)))
        done_beams.append(beams)
    if gather_idx:
        flat = logp_rows.view(L * N, V1)[torch.tensor(gather_idx, device=dev)]
        off = 0
        sums = []
        for k, bi, ln in gather_meta:
            lp = flat[off:off + ln]
            off += ln
            done_beams[k][bi]['logps'] = lp
            sums.append(lp.sum())
        sums = torch.stack(sums).cpu().tolist()
        for (k, bi, _), s in zip(gather_meta, sums):
            done_beams[k][bi]['unaug_p'] = s
            del done_beams[k][bi]['_rows']
    model.done_beams = done_beams
    for k in range(B):
        if sample_n == beam_size:
            for _n in range(sample_n):
                ln = done_beams[k][_n]['seq'].shape[0]
                seq[k * sample_n + _n, :ln] = done_beams[k][_n]['seq']
                seq_logp[k * sample_n + _n, :ln] = done_beams[k][_n]['logps']
        else:
            ln = done_beams[k][0]['seq'].shape[0]
            seq[k, :ln] = done_beams[k][0]['seq']
            seq_logp[k, :ln] = done_beams[k][0]['logps']
    return seq, seq_logp


def unk_column(model, opt, V1):
    """CaptionModel.py:159-162: the column pushed down by 1000 when suppress_UNK is on."""
    if opt.get('suppress_UNK', 0) and hasattr(model, 'vocab') and model.vocab.get(str(V1 - 1)) == 'UNK':
        return V1 - 1
    if getattr(model, 'unk_idx', None) is not None:
        return int(model.unk_idx)
    return -1


def beam_search_steps(model, step, reorder, B, V1, L, opt, dev):
    """Beam search (CaptionModel.beam_search, group_size 1) for decoders whose step is orchestrated from the host
    (Transformer, AoA): the selection / reordering / normalisation kernels and the finished-beam assembly are the ones of
    the UpDown path, only the decoder step is a callback.

    step(t, it, rows_per_image) -> logits [B*rows_per_image, V1]: consumes tokens `it` (BOS zeros at t = 0, one row per
        image; afterwards beam_size rows per image) and advances the decoder state held by the caller.
    reorder(parent [B,bd] int32, cur): state row b*cur + parent[b,j] becomes row b*bd + j.
    """
    beam_size = opt.get('beam_size', 10)
    sample_n = opt.get('sample_n', 10)
    if opt.get('group_size', 1) != 1:
        raise NotImplementedError('diverse beam search (group_size > 1) is outside the hot-path scope')
    for k in ('decoding_constraint', 'remove_bad_endings'):
        if opt.get(k, 0):
            raise NotImplementedError('%s is not part of the accelerated beam search' % k)
    assert sample_n == 1 or sample_n == beam_size, 'when beam search, sample_n == 1 or beam search'
    bd = beam_size
    N = B * bd
    temperature = float(opt.get('temperature', 1))
    unk = unk_column(model, opt, V1)
    logp_rows = torch.zeros(L, N, V1, dtype=_f32, device=dev)
    parent = torch.zeros(L, B, bd, dtype=torch.int32, device=dev)
    token = torch.zeros(L, B, bd, dtype=torch.long, device=dev)
    score = torch.zeros(L, B, bd, dtype=_f32, device=dev)
    ended = torch.zeros(L, B, bd, dtype=torch.uint8, device=dev)
    sums = torch.zeros(2, B, bd, dtype=_f32, device=dev)
    st = stream_ptr()
    it = torch.zeros(B, dtype=torch.long, device=dev)                       # BOS
    logits = step(0, it, 1)
    check(lib.capmi_beam_logsoftmax(ptr(logits), ptr(logp_rows[0]), B, V1, temperature, unk, st), 'beam_logsoftmax')
    cur = 1
    for t in range(L):
        check(lib.capmi_beam_select(ptr(logp_rows[t]), ptr(sums[t & 1]), B, cur, bd, V1, 1 if t == L - 1 else 0, ptr(parent[t]),
                                    ptr(token[t]), ptr(score[t]), ptr(sums[(t + 1) & 1]), ptr(ended[t]), st), 'beam_select')
        if t == L - 1:
            break
        reorder(parent[t], cur)
        logits = step(t + 1, token[t].reshape(N), bd)
        check(lib.capmi_beam_logsoftmax(ptr(logits), ptr(logp_rows[t + 1]), N, V1, temperature, unk, st), 'beam_logsoftmax')
        cur = bd
    return assemble_done_beams(model, parent, token, score, ended, logp_rows, B, bd, L, V1, sample_n, beam_size, opt)


def reorder_rows(src, dst, parent, B, cur, bd):
    """dst[a, b*bd + j, :] = src[a, b*cur + parent[b,j], :] for stacked state arrays [arrays, rows, R] (capmi_beam_reorder)."""
    arrays, _, R = src.shape
    check(lib.capmi_beam_reorder(ptr(src), ptr(dst), ptr(parent), arrays, B, cur, bd, R, stream_ptr()), 'beam_reorder')
