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
    assert sample_n == 1 or sample_n == beam_size // group_size, 'when beam search, sample_n == 1 or beam search'
    model._device_check(fc_feats)
    P = {k: v.detach() for k, v in model.named_parameters()}
    pr = engine.prepare(P, fc_feats.float().contiguous(), att_feats.float().contiguous(),
                        None if att_masks is None else att_masks.float())
    dev = fc_feats.device
    if group_size != 1 or opt.get('decoding_constraint', 0) or opt.get('remove_bad_endings', 0):
        # diverse groups / decoding constraints: same kernels, host-stepped (the one-call search below has no hooks)
        from .step import UpDownStepper
        return beam_search_steps(model, lambda rows: UpDownStepper(P, pr, rows), pr.att.shape[0], P['embed.0.weight'].shape[0],
                                 model.seq_length, opt, dev)
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


def assemble_done_beams(model, parent, token, score, ended, logp_rows, B, bd, L, V1, sample_n, beam_size, opt, groups=1):
    """One device->host transfer of the small [L,B,groups*bd] tables, then the bookkeeping of CaptionModel.py:183-209:
    per image and group the finished beams sorted by p, best bd kept, groups concatenated (:207-208)."""
    dev = logp_rows.device
    W = groups * bd
    N = B * W
    parent, token, score, ended = (t.cpu().numpy() for t in (parent, token, score, ended))
    penalty = _penalty(opt.get('length_penalty', ''))
    seq = torch.zeros(B * sample_n, L, dtype=torch.long, device=dev)
    seq_logp = torch.zeros(B * sample_n, L, V1, dtype=_f32, device=dev)
    done_beams = []
    gather_idx, gather_meta = [], []
    for k in range(B):
        beams = []
        for g in range(groups):
            fin = []
            for t in range(L):
                for j in range(g * bd, (g + 1) * bd):
                    if ended[t, k, j]:
                        fin.append((penalty(t + 1, float(score[t, k, j])), t, j))
            fin = sorted(fin, key=lambda x: -x[0])[:bd]            # stable, like the reference's sorted()
            for p, t, j in fin:
                toks, rows = [], []
                jj = j
                for s in range(t, -1, -1):
                    toks.append(int(token[s, k, jj]))
                    par = int(parent[s, k, jj])                   # index among the image's W rows of step s-1
                    rows.append(s * N + (k if s == 0 else k * W + par))   # step-0 rows are one per image
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


def _decode_flags(model, opt, dev):
    flags = (_lib.DECODE_NO_REPEAT if opt.get('decoding_constraint', 0) else 0) | \
            (_lib.DECODE_NO_BAD_ENDING if opt.get('remove_bad_endings', 0) else 0)
    bad = torch.tensor(sorted(getattr(model, 'bad_endings_ix', [])), dtype=torch.long, device=dev)
    return flags, bad


def beam_search_steps(model, make_decoder, B, V1, L, opt, dev):
    """CaptionModel.beam_search (CaptionModel.py:35-209) for any decoder given as a single-step object (step.py
    protocol): the selection / reordering / normalisation / constraint / diversity kernels are native, only the decoder
    step is a callback.  Covers decoding_constraint, remove_bad_endings, suppress_UNK, temperature, length_penalty and
    -- group_size > 1 -- diverse beam search, which is delegated to diverse_beam_search_steps.

    make_decoder(rows_per_image) -> decoder with
      step(t, it, rows_per_image) -> logits [B*rows_per_image, V1]: consumes tokens `it` (BOS zeros at t = 0, one row per
          image; afterwards rows_per_image rows per image, image-major) and advances the state;
      reorder(parent [B,bd] int32, cur): state row b*cur + parent[b,j] becomes row b*bd + j.
    """
    beam_size = opt.get('beam_size', 10)
    sample_n = opt.get('sample_n', 10)
    if int(opt.get('group_size', 1)) != 1:
        return diverse_beam_search_steps(model, make_decoder, B, V1, L, opt, dev)
    assert sample_n == 1 or sample_n == beam_size, 'when beam search, sample_n == 1 or beam search'
    dec = make_decoder(beam_size)
    step, reorder = dec.step, dec.reorder
    bd = beam_size
    N = B * bd
    temperature = float(opt.get('temperature', 1))
    unk = unk_column(model, opt, V1)
    flags, bad = _decode_flags(model, opt, dev)
    logp_rows = torch.zeros(L, N, V1, dtype=_f32, device=dev)
    parent = torch.zeros(L, B, bd, dtype=torch.int32, device=dev)
    token = torch.zeros(L, B, bd, dtype=torch.long, device=dev)
    score = torch.zeros(L, B, bd, dtype=_f32, device=dev)
    ended = torch.zeros(L, B, bd, dtype=torch.uint8, device=dev)
    sums = torch.zeros(2, B, bd, dtype=_f32, device=dev)
    st = stream_ptr()
    it = torch.zeros(B, dtype=torch.long, device=dev)                       # BOS
    logits = step(0, it, 1)
    # the first distribution is the model's own log_softmax; the temperature enters at CaptionModel.py:203-204 only
    check(lib.capmi_beam_logsoftmax(ptr(logits), ptr(logp_rows[0]), B, V1, 1.0, unk, st), 'beam_logsoftmax')
    cur = 1
    for t in range(L):
        check(lib.capmi_beam_select(ptr(logp_rows[t]), ptr(sums[t & 1]), B, cur, bd, V1, 1 if t == L - 1 else 0, ptr(parent[t]),
                                    ptr(token[t]), ptr(score[t]), ptr(sums[(t + 1) & 1]), ptr(ended[t]), st), 'beam_select')
        if t == L - 1:
            break
        reorder(parent[t], cur)
        logits = step(t + 1, token[t].reshape(N), bd)
        check(lib.capmi_beam_logsoftmax(ptr(logits), ptr(logp_rows[t + 1]), N, V1, temperature, unk, st), 'beam_logsoftmax')
        if flags:                                                           # CaptionModel.py:152-155
            check(lib.capmi_decode_constrain(ptr(logp_rows[t + 1]), N, V1, ptr(token[t]), 1, flags, ptr(bad), bad.numel(), None, 0,
                                             t + 1, 0, st), 'decode_constrain')
        cur = bd
    return assemble_done_beams(model, parent, token, score, ended, logp_rows, B, bd, L, V1, sample_n, beam_size, opt)


def diverse_beam_search_steps(model, make_decoder, B, V1, L, opt, dev):
    """Diverse beam search: group_size groups of bdash = beam_size // group_size beams; group g is penalised by
    diversity_lambda for every token the beams of groups < g hold at the same position (CaptionModel.add_diversity, :38-57).

    The groups run staggered in time exactly as in the reference (:145-204: at time t group g takes its local step t - g),
    and that matters: `beam_seq_table[pg][:, :, local_time]` is read AFTER group pg has already advanced g - pg further
    steps, so it holds position local_time of pg's *surviving* beams, not what pg chose at that step.  Here that column is
    recovered from pg's parent pointers by a (g - pg)-deep gather on the device.  Every group owns a decoder of bdash rows
    per image."""
    beam_size = opt.get('beam_size', 10)
    sample_n = opt.get('sample_n', 10)
    G = int(opt.get('group_size', 1))
    lam = float(opt.get('diversity_lambda', 0.5))
    assert beam_size % G == 0, 'beam_size must be a multiple of group_size (split_tensors, models/utils.py:19)'
    bd = beam_size // G
    assert sample_n == 1 or sample_n == bd, 'when beam search, sample_n == 1 or beam search'
    W, n = G * bd, B * bd
    temperature = float(opt.get('temperature', 1))
    unk = unk_column(model, opt, V1)
    flags, bad = _decode_flags(model, opt, dev)
    decs = [make_decoder(bd) for _ in range(G)]
    logp_rows = torch.zeros(L, B * W, V1, dtype=_f32, device=dev)          # rows image-major, then group, then beam
    lparent = torch.zeros(G, L, B, bd, dtype=torch.int32, device=dev)      # per-group tables
    ltoken = torch.zeros(G, L, B, bd, dtype=torch.long, device=dev)
    lscore = torch.zeros(G, L, B, bd, dtype=_f32, device=dev)
    lended = torch.zeros(G, L, B, bd, dtype=torch.uint8, device=dev)
    sums = torch.zeros(G, 2, B, bd, dtype=_f32, device=dev)
    cur_logp = torch.empty(G, n, V1, dtype=_f32, device=dev)               # the rows group g selects from next
    aug = torch.empty(n, V1, dtype=_f32, device=dev)
    ident = torch.arange(bd, device=dev).expand(B, bd)
    st = stream_ptr()
    it0 = torch.zeros(B, dtype=torch.long, device=dev)                      # BOS
    for g in range(G):                                                      # logprobs_table = [init_logprobs.clone() ...] (:136)
        logits = decs[g].step(0, it0, 1)
        check(lib.capmi_beam_logsoftmax(ptr(logits), ptr(cur_logp[g]), B, V1, 1.0, unk, st), 'beam_logsoftmax')
    logp_rows[0, :B] = cur_logp[0, :B]
    for t in range(L + G - 1):
        for g in range(G):
            lt = t - g
            if lt < 0 or lt > L - 1:
                continue
            cur = 1 if lt == 0 else bd
            src = cur_logp[g]
            sel = src
            if g > 0:
                prev = []
                for pg in range(g):
                    idx = ident
                    for s in range(min(t - pg, L - 1), lt, -1):            # follow pg's survivors back to position lt
                        idx = lparent[pg, s].long().gather(1, idx)
                    prev.append(ltoken[pg, lt].gather(1, idx))
                prev = torch.cat(prev, 1).contiguous()                     # [B, g*bd]
                check(lib.capmi_beam_diversity(ptr(src), ptr(aug), B, cur, V1, ptr(prev), g * bd, g * bd, lam, st), 'beam_diversity')
                sel = aug
            last = 1 if lt == L - 1 else 0
            check(lib.capmi_beam_select(ptr(sel), ptr(sums[g, lt & 1]), B, cur, bd, V1, last, ptr(lparent[g, lt]), ptr(ltoken[g, lt]),
                                        ptr(lscore[g, lt]), ptr(sums[g, (lt + 1) & 1]), ptr(lended[g, lt]), st), 'beam_select')
            if last:
                continue
            decs[g].reorder(lparent[g, lt], cur)
            logits = decs[g].step(lt + 1, ltoken[g, lt].reshape(n), bd)
            check(lib.capmi_beam_logsoftmax(ptr(logits), ptr(cur_logp[g]), n, V1, temperature, unk, st), 'beam_logsoftmax')
            if flags:
                check(lib.capmi_decode_constrain(ptr(cur_logp[g]), n, V1, ptr(ltoken[g, lt]), 1, flags, ptr(bad), bad.numel(), None,
                                                 0, lt + 1, 0, st), 'decode_constrain')
            logp_rows[lt + 1].view(B, G, bd, V1)[:, g] = cur_logp[g].view(B, bd, V1)
    # image-major global tables for the shared assembly: beam j of group g is column g*bd + j
    offs = (torch.arange(G, device=dev, dtype=torch.int32) * bd).view(G, 1, 1, 1)
    parent = (lparent + offs).permute(1, 2, 0, 3).reshape(L, B, W).contiguous()
    token = ltoken.permute(1, 2, 0, 3).reshape(L, B, W).contiguous()
    score = lscore.permute(1, 2, 0, 3).reshape(L, B, W).contiguous()
    ended = lended.permute(1, 2, 0, 3).reshape(L, B, W).contiguous()
    return assemble_done_beams(model, parent, token, score, ended, logp_rows, B, bd, L, V1, sample_n, beam_size, opt, groups=G)


def reorder_rows(src, dst, parent, B, cur, bd):
    """dst[a, b*bd + j, :] = src[a, b*cur + parent[b,j], :] for stacked state arrays [arrays, rows, R] (capmi_beam_reorder)."""
    arrays, _, R = src.shape
    check(lib.capmi_beam_reorder(ptr(src), ptr(dst), ptr(parent), arrays, B, cur, bd, R, stream_ptr()), 'beam_reorder')
