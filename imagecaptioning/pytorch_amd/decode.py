"""Host-stepped samplers for the decode-time options the one-call rollouts have no hooks for (evaluation only, no gradient):

    sample_steps          AttModel._sample           AttModel.py:258-352  decoding_constraint, remove_bad_endings,
                                                                          block_trigrams (+ every sample_method)
    diverse_sample_steps  AttModel._diverse_sample   AttModel.py:354-447  group_size > 1 without beam search

Both drive a stepper (step.py protocol; any model family) and keep every per-step edit on the device: normalisation,
constraints (capmi_decode_constrain), group penalties (capmi_column_penalty) and the token choice
(capmi_select_logp) are kernels of libcapmi, the finished flags never leave the device, there is no `.item()` in the loop.
Results follow the reference value for value, including its quirks (noted inline).
"""
import ctypes as C

import torch

from . import _lib
from ._lib import lib, ptr, check, stream_ptr

_f32 = torch.float32
_MODES = {'greedy': 0, 'sample': 1}


def wants_options(opt):
    """True when `opt` asks for something only the host-stepped samplers implement."""
    return bool(opt.get('decoding_constraint', 0) or opt.get('block_trigrams', 0) or opt.get('remove_bad_endings', 0)
                or opt.get('group_size', 1) > 1)


def _filter(top_k, top_p):
    return C.byref(_lib.SampleFilter(int(top_k), float(top_p))) if (top_k or top_p) else None


def _flags(opt, t):
    f = 0
    if t > 0 and opt.get('decoding_constraint', 0):
        f |= _lib.DECODE_NO_REPEAT
    if t > 0 and opt.get('remove_bad_endings', 0):
        f |= _lib.DECODE_NO_BAD_ENDING
    if t >= 3 and opt.get('block_trigrams', 0):
        f |= _lib.DECODE_BLOCK_TRIGRAMS
    return f


def _prev(seq, t):
    """address of seq[0, t-1] (int64 rows of length L): the previous token of every row with stride L."""
    return seq.data_ptr() + 8 * (t - 1) if t > 0 else None


def sample_steps(model, stepper, B, L, opt, dev, seed=0):
    """Returns (seq [B*sample_n, L] int64, seqLogprobs [B*sample_n, L, V1])."""
    if not opt.get('output_logsoftmax', 1):
        raise NotImplementedError('output_logsoftmax=0 is only used by margin structure losses')
    from .captioning.models.utils import parse_sample_method
    n = int(opt.get('sample_n', 1))
    mode, temperature, top_k, top_p = parse_sample_method(opt.get('sample_method', 'greedy'), opt.get('temperature', 1.0))
    V1 = stepper.V1
    N = B * n
    seq = torch.zeros(N, L, dtype=torch.long, device=dev)
    seq_logp = torch.zeros(N, L, V1, dtype=_f32, device=dev)
    it = torch.zeros(N, dtype=torch.long, device=dev)                       # BOS
    unf = torch.ones(N, dtype=torch.uint8, device=dev)
    logp = torch.empty(N, V1, dtype=_f32, device=dev)
    bad = torch.tensor(sorted(getattr(model, 'bad_endings_ix', [])), dtype=torch.long, device=dev)
    gumbel = opt.get('_gumbel')                                              # test hook: injected noise [L,N,V1]
    flt = _filter(top_k, top_p)
    any_flags = False
    st = stream_ptr()
    for t in range(L):
        logits = stepper.step(t, it, n)
        check(lib.capmi_log_softmax_rows(ptr(logits), ptr(logp), N, V1, st), 'capmi_log_softmax_rows')
        flags = _flags(opt, t)
        if flags:
            any_flags = True
            # block_trigrams: the reference loops over range(batch_size) = the IMAGE count, so with sample_n > 1 only the
            # first B rows are ever blocked (AttModel.py:310, 322); kept
            check(lib.capmi_decode_constrain(ptr(logp), N, V1, _prev(seq, t), L, flags, ptr(bad), bad.numel(), ptr(seq), L, t, B,
                                             st), 'capmi_decode_constrain')
        check(lib.capmi_select_logp(ptr(logp), N, V1, t, L, _MODES[mode], float(temperature),
                                    None if gumbel is None else ptr(gumbel[t]), int(seed) & 0xFFFFFFFFFFFFFFFF, ptr(seq), L,
                                    ptr(it), ptr(unf), ptr(seq_logp), None, 0, flt, st), 'capmi_select_logp')
    if any_flags:
        # the reference leaves the loop once every row has finished (AttModel.py:350-351), so later columns stay zero; here
        # the loop always runs L steps and finished rows hold logprobs * 0 (NaN where a constraint put -inf): blank the
        # columns the reference never wrote.  Device-side, no sync.
        alive = (seq != 0).cumprod(1).any(0)                                 # some row unfinished after step t
        written = torch.cat([alive.new_ones(1), alive[:-1]])
        seq_logp = torch.where(written.view(1, L, 1), seq_logp, torch.zeros((), dtype=_f32, device=dev))
    return seq, seq_logp


def diverse_sample_steps(model, stepper, B, L, opt, dev, seed=0):
    """AttModel._diverse_sample: group g is penalised in every column any row of groups < g picked at the same step.
    All groups advance together (one decoder step on B*group_size rows); the reference's time stagger only exists so that
    the earlier groups' choices at step t are known, which they are here because the groups choose in order.
    Returns (seq [B*group_size, L], seqLogprobs [B*group_size, L]) -- 2-D log-probs, as the reference does."""
    from .captioning.models.utils import parse_sample_method
    G = int(opt.get('group_size', 1))
    lam = float(opt.get('diversity_lambda', 0.5))
    temperature = float(opt.get('temperature', 1.0))
    mode, _, top_k, top_p = parse_sample_method(opt.get('sample_method', 'greedy'), 1.0)   # sample_next_word(.., 1): :434
    V1 = stepper.V1
    N = B * G
    seq_tab = torch.zeros(G, B, L, dtype=torch.long, device=dev)
    slp_tab = torch.zeros(G, B, L, dtype=_f32, device=dev)
    it_tab = torch.zeros(G, B, dtype=torch.long, device=dev)
    unf = torch.ones(G, B, dtype=torch.uint8, device=dev)
    logp_all = torch.empty(N, V1, dtype=_f32, device=dev)
    lp = torch.empty(G, B, V1, dtype=_f32, device=dev)
    bad = torch.tensor(sorted(getattr(model, 'bad_endings_ix', [])), dtype=torch.long, device=dev)
    gumbel = opt.get('_gumbel')                                              # test hook [L,G,B,V1]
    flt = _filter(top_k, top_p)
    st = stream_ptr()
    for t in range(L):
        it = it_tab.t().reshape(N)                                           # rows image-major: b*G + g
        logits = stepper.step(t, it.contiguous(), G)
        # F.log_softmax(get_logprobs_state(..) / temperature) (:388-389)
        check(lib.capmi_beam_logsoftmax(ptr(logits), ptr(logp_all), N, V1, temperature, -1, st), 'capmi_beam_logsoftmax')
        lp.copy_(logp_all.view(B, G, V1).transpose(0, 1))
        for g in range(G):
            x = lp[g]
            for pg in range(g):                                              # :392-397
                check(lib.capmi_column_penalty(ptr(x), B, V1, seq_tab[pg].data_ptr() + 8 * t, B, L, lam, st),
                      'capmi_column_penalty')
            flags = _flags(opt, t)
            if flags:
                check(lib.capmi_decode_constrain(ptr(x), B, V1, _prev(seq_tab[g], t), L, flags, ptr(bad), bad.numel(),
                                                 ptr(seq_tab[g]), L, t, B, st), 'capmi_decode_constrain')
            check(lib.capmi_select_logp(ptr(x), B, V1, t, L, _MODES[mode], 1.0, None if gumbel is None else ptr(gumbel[t, g]),
                                        (int(seed) + 0x9E3779B97F4A7C15 * g) & 0xFFFFFFFFFFFFFFFF, ptr(seq_tab[g]), L,
                                        ptr(it_tab[g]), ptr(unf[g]), None, ptr(slp_tab[g]), 1, flt, st), 'capmi_select_logp')
    return seq_tab.transpose(0, 1).reshape(N, L), slp_tab.transpose(0, 1).reshape(N, L)
