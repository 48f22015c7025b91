"""ctypes binding of libcapmi.so (the C ABI declared in include/capmi.h).

The product path has NO CPU fallback: if the shared object is missing or does not export a symbol
this module raises at import time (build it with ``python -m imagecaptioning.pytorch_amd.build``).
``import torch`` comes first on purpose -- torch ships its own ``libamdhip64.so.7``; loading it first
makes libcapmi resolve the same HIP runtime, so torch streams and device pointers are valid inside
our launches.
"""
import ctypes as C
import os

import torch  # noqa: F401  (must precede the CDLL: shared HIP runtime)

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('CAPMI_LIB') or os.path.join(HERE, 'libcapmi.so')   # CAPMI_LIB: experiment builds only
MAX_SEG = 4
EINVAL = -1

c_f = C.c_void_p      # device pointers travel as integers (tensor.data_ptr())


class GemmSeg(C.Structure):
    _fields_ = [('A', c_f), ('B', c_f), ('lda', C.c_int), ('ldb', C.c_int), ('K', C.c_int), ('a_row_div', C.c_int)]


class GemmDesc(C.Structure):
    _fields_ = [('seg', GemmSeg * MAX_SEG), ('nseg', C.c_int), ('a_layout', C.c_int), ('b_layout', C.c_int),
                ('M', C.c_int), ('N', C.c_int), ('C', c_f), ('ldc', C.c_int), ('bias', c_f), ('bias2', c_f),
                ('row_bias', c_f), ('row_bias_div', C.c_int), ('mul_mask', c_f), ('relu', C.c_int),
                ('accumulate', C.c_int), ('partial', c_f), ('partial_capacity', C.c_int64), ('splits', C.c_int),
                ('defer_reduce', C.c_int), ('splits_used', C.c_int), ('a_planes', c_f * MAX_SEG), ('addend', c_f),
                ('allow_wide_deferred', C.c_int)]


class NextEmbed(C.Structure):
    """capmi_next_embed (include/capmi.h): the next step's token embedding written by the select launch"""
    _fields_ = [('E', c_f), ('mask', c_f), ('x', c_f), ('it_save', c_f), ('Edim', C.c_int), ('relu', C.c_int), ('x_planes', c_f),
                ('alive', c_f)]


class UpDownWeights(C.Structure):
    _fields_ = [(k, c_f) for k in (
        'embed', 'att_w_ih', 'att_w_hh', 'att_b_ih', 'att_b_hh', 'lang_w_ih', 'lang_w_hh', 'lang_b_ih', 'lang_b_hh',
        'h2att_w', 'h2att_b', 'alpha_w', 'alpha_b', 'logit_w', 'logit_b')]


class UpDownRollout(C.Structure):
    _fields_ = ([(k, C.c_int) for k in ('B', 'n', 'N', 'K', 'A', 'R', 'E', 'V1', 'B_feat')] + [('row_img', c_f)] +
                [(k, C.c_int) for k in ('T', 'L')] +
                [(k, c_f) for k in ('fc', 'att', 'p_att', 'att_mask', 'drop_xt', 'drop_out')] +
                [('mode', C.c_int), ('row_mode', c_f), ('temperature', C.c_float), ('gumbel', c_f),
                 ('seed', C.c_uint64), ('forced', c_f), ('forced_ld', C.c_int), ('teacher', C.c_int)] +
                [(k, c_f) for k in ('h_att', 'c_att', 'h_lang', 'c_lang', 'xt', 'it_all', 'gates_att', 'gates_lang',
                                    'att_h', 'alpha', 'ctx', 'h_drop', 'seq', 'seq_logp', 'sel_logp', 'live',
                                    'fc_gates', 'logits', 'it', 'unfinished', 'partial')] +
                [('partial_capacity', C.c_int64), ('top_k', C.c_int), ('top_p', C.c_float), ('ss_mode', c_f),
                 ('planes', c_f), ('planes_bytes', C.c_int64), ('early_exit', C.c_int), ('early_exit_from', C.c_int),
                 ('alive_host', c_f), ('steps_run', C.c_int), ('pre_partial', c_f), ('pre_capacity', C.c_int64), ('raw_logits', C.c_int)])


class SampleFilter(C.Structure):
    _fields_ = [('top_k', C.c_int), ('top_p', C.c_float)]


class UpDownGrads(C.Structure):
    _fields_ = [(k, c_f) for k in (
        'embed', 'att_w_ih', 'att_w_hh', 'att_b_ih', 'att_b_hh', 'lang_w_ih', 'lang_w_hh', 'lang_b_ih', 'lang_b_hh',
        'h2att_w', 'h2att_b', 'alpha_w', 'alpha_b', 'logit_w', 'logit_b', 'd_fc', 'd_att', 'd_p_att')]


class SparseLogpGrad(C.Structure):
    _fields_ = [('g_sel', c_f), ('g_sum', c_f), ('tok', c_f), ('tok_ld', C.c_int), ('raw', C.c_int), ('scale', c_f)]


class MaskDesc(C.Structure):
    _fields_ = [('mask', c_f), ('count', C.c_int64), ('offset', C.c_uint64), ('row_len', C.c_int), ('rows', C.c_int),
                ('keep_from', C.c_int)]


class ReduceItem(C.Structure):
    _fields_ = [('partial', c_f), ('C', c_f), ('bias', c_f), ('splits', C.c_int32), ('M', C.c_int32), ('N', C.c_int32),
                ('ldc', C.c_int32), ('accumulate', C.c_int32), ('reserved', C.c_int32)]


class GroupGemm(C.Structure):
    _fields_ = [('A', c_f), ('B', c_f), ('C', c_f), ('lda', C.c_int32), ('ldb', C.c_int32), ('ldc', C.c_int32), ('K', C.c_int32),
                ('M', C.c_int32), ('N', C.c_int32), ('accumulate', C.c_int32), ('splits_used', C.c_int32), ('colsum', c_f)]


class ColsumItem(C.Structure):
    _fields_ = [('in', c_f), ('out', c_f), ('out2', c_f), ('rows', C.c_int32), ('cols', C.c_int32), ('ld', C.c_int32),
                ('accumulate', C.c_int32)]


class UpDownBwdScratch(C.Structure):
    _fields_ = ([(k, c_f) for k in ('dlogits', 'd_hdrop', 'dg_att', 'dg_lang', 'd_x2', 'd_e_all', 'd_att_h_all',
                                    'dh_att_attn', 'd_x1', 'dc_att', 'dc_lang', 'd_xt_all', 'sum_dg_att', 'w_lang_cat',
                                    'w_att_cat', 'partial')] +
                [('partial_capacity', C.c_int64), ('sparse', C.POINTER(SparseLogpGrad)), ('n_grad_rows', C.c_int), ('pack', c_f),
                 ('pack_capacity', C.c_int64), ('planes', c_f), ('planes_bytes', C.c_int64)])


class UpDownBeam(C.Structure):
    _fields_ = ([(k, C.c_int) for k in ('B', 'bd', 'K', 'A', 'R', 'E', 'V1', 'L')] +
                [(k, c_f) for k in ('fc', 'att', 'p_att', 'att_mask')] + [('temperature', C.c_float), ('unk_col', C.c_int)] +
                [(k, c_f) for k in ('state', 'xt', 'gates', 'att_h', 'alpha', 'ctx', 'fc_gates', 'logits', 'it', 'sums',
                                    'logp_rows', 'parent', 'token', 'score', 'ended', 'partial')] +
                [('partial_capacity', C.c_int64)])


class NewFCWeights(C.Structure):
    _fields_ = [(k, c_f) for k in ('embed', 'i2h_w', 'i2h_b', 'h2h_w', 'h2h_b', 'logit_w', 'logit_b')]


class NewFCRollout(C.Structure):
    _fields_ = ([(k, C.c_int) for k in ('B', 'n', 'N', 'R', 'E', 'V1', 'T', 'L')] + [('fc_emb', c_f), ('drop_out', c_f)] +
                [('mode', C.c_int), ('temperature', C.c_float), ('gumbel', c_f), ('seed', C.c_uint64), ('forced', c_f),
                 ('forced_ld', C.c_int), ('teacher', C.c_int)] +
                [(k, c_f) for k in ('h', 'c', 'x', 'it_all', 'saved', 'h_drop', 'seq', 'seq_logp', 'sel_logp', 'live',
                                    'logits', 'it', 'unfinished', 'partial')] + [('partial_capacity', C.c_int64)])


class NewFCGrads(C.Structure):
    _fields_ = [(k, c_f) for k in ('embed', 'i2h_w', 'i2h_b', 'h2h_w', 'h2h_b', 'logit_w', 'logit_b', 'd_fc_emb')]


class StepState(C.Structure):
    """capmi_step_state (include/capmi.h): the per-iteration record a captured training step reads from DEVICE memory"""
    _fields_ = [('epoch', C.c_uint64), ('adam_step', C.c_int32), ('lr', C.c_float), ('bc1', C.c_float), ('bc2_sqrt', C.c_float),
                ('reserved', C.c_float * 2)]


class NewFCBwdScratch(C.Structure):
    _fields_ = ([(k, c_f) for k in ('dlogits', 'd_hdrop', 'd_sums', 'dh_prev', 'dc', 'd_x_all', 'd_ximg', 'partial')] +
                [('partial_capacity', C.c_int64), ('sparse', C.POINTER(SparseLogpGrad))])


_I, _F, _P, _U64, _I64 = C.c_int, C.c_float, C.c_void_p, C.c_uint64, C.c_int64
DECODE_NO_REPEAT, DECODE_NO_BAD_ENDING, DECODE_BLOCK_TRIGRAMS = 1, 2, 4      # capmi.h CAPMI_DECODE_*
SELECT_RAW = 256       # capmi.h CAPMI_SELECT_RAW: OR into the select `mode` -- the stored rows are the logits, not the log-probabilities

# name -> argtypes (restype is always int unless noted).  Must list EVERY symbol of include/capmi.h:
# tests/test_abi.py cross-checks this table against the header and the built library.
SIGNATURES = {
    'capmi_version': [],
    'capmi_arch': [],
    'capmi_gemm_f32': [C.POINTER(GemmDesc), _P],
    'capmi_planes_bytes': [_I],
    'capmi_planes_from_f32': [_P, _I, _I, _I, _P, _P],
    'capmi_updown_planes_bytes': [_I, _I],
    'capmi_updown_bwd_planes_bytes': [_I],
    'capmi_attention_fwd': [_P] * 8 + [_I] * 5 + [_P, _I, _P],
    'capmi_attention_fwd_partial': [_P, _I, _I64, _P, _P] + [_P] * 7 + [_I] * 5 + [_P, _I, _P],
    'capmi_attention_fwd_partial_pl': [_P, _I, _I64, _P, _P] + [_P] * 7 + [_I] * 5 + [_P, _I, _P, _P],
    'capmi_attention_bwd': [_P, _I] + [_P] * 8 + [_I] * 5 + [_P, _I, _P],
    'capmi_attention_bwd_partial': [_P, _I, _I64, _I, _P] + [_P] * 7 + [_I] * 5 + [_P, _I, _P],
    'capmi_attention_bwd_batched': [_P, _I] + [_P] * 9 + [_I] * 7 + [_P],
    'capmi_attention_bwd_batched_ws': [_P, _I] + [_P] * 9 + [_I] * 7 + [_P, _P],
    'capmi_lstm_cell_fwd': [_P, _I, _P, _P, _P, _I, _P] + [_P] * 6 + [_I, _I, _P],
    'capmi_lstm_cell_fwd_pl': [_P, _I, _P, _P, _P, _I, _P] + [_P] * 6 + [_I, _I, _P, _P, _P],
    'capmi_lstm_cell_fwd_pl2': [_P, _I, _P, _I, _P, _P, _P, _I, _P] + [_P] * 6 + [_I, _I, _P, _P, _P],
    'capmi_lstm_cell_bwd': [_P, _I, _P, _P, _I, _P, _I] + [_P] * 6 + [_I, _I, _P],
    'capmi_lstm_cell_bwd_partial_pl': [_P, _I, _P, _P, _I, _I, _I64, _P, _I, _I, _I64] + [_P] * 6 + [_I, _I, _P, _P],
    'capmi_embed_fwd_pl': [_P, _I, _P, _P, _P, _P, _I, _I, _I, _P, _P],
    'capmi_lstm_cell_bwd_partial': [_P, _I, _P, _P, _I, _I, _I64, _P, _I, _I, _I64] + [_P] * 6 + [_I, _I, _P],
    'capmi_embed_fwd': [_P, _I, _P, _P, _P, _P, _I, _I, _I, _P],
    'capmi_embed_bwd': [_P] * 5 + [_I, _I, _I, _P],
    'capmi_logsoftmax_select': [_P, _I, _I, _I, _I, _I, _P, _F, _P, _U64, _P, _I, _I, _P, _I, _P, _P, _P, _P, _P, _P],
    'capmi_logsoftmax_select_partial': [_P, _I, _I64, _P, _I, _I, _I, _I, _I, _P, _F, _P, _U64, _P, _I, _I, _P, _I, _P, _P, _P,
                                        _P, _P, _P, _P, _P],
    'capmi_logsoftmax_select_partial_gemm': [_P, _I, _I64, _P, _I, _I, _I, _I, _I, _P, _F, _P, _U64, _P, _I, _I, _P, _I, _P, _P, _P,
                                        _P, _P, _P, _P, _P, _P],
    'capmi_logsoftmax_bwd': [_P, _P, _P, _P, _I, _I, _I, _I, _P],
    'capmi_logsoftmax_bwd_sparse': [C.POINTER(SparseLogpGrad), _P, _P, _P, _P, _I, _I, _I, _I, _P],
    'capmi_splitk_reduce': [_P, _I, _P, _I, _I, _I, _P, _P, _P, _I, _P, _I, _I, _P],
    'capmi_dropout_mask': [_P, _I64, _F, _U64, _U64, _P],
    'capmi_dropout_masks': [C.POINTER(MaskDesc), _I, _F, _U64, _P],
    'capmi_rollout_init': [_P, _P, _P, _P, _I64, _P, _P, _I, _P],
    'capmi_reward_criterion': [_P, _I, _P, _I, _P, _I, _I, _I, _I, _I, _I, _P, _P, _P],
    'capmi_colsum': [_P, _I, _I, _I, _P, _I, _P],
    'capmi_colsum_batch': [_P, _I, _P],
    'capmi_colsum_batch_args': [C.POINTER(ColsumItem), _I, _P],
    'capmi_splitk_reduce_batch': [_P, _I, _P],
    'capmi_group_rowsum': [_P, _I, _I64, _I, _I, _I, _P, _P],
    'capmi_relu_mask_bwd': [_P, _P, _P, _P, _I64, _P],
    'capmi_relu_scale_bwd': [_P, _P, _F, _P, _I64, _P],
    'capmi_adam_step': [_P, _P, _P, _P, _I64, _F, _F, _F, _F, _F, _F, _F, _I, _P],
    'capmi_upload_async': [_P, _P, _I64, _P],
    'capmi_step_advance': [_P, _F, _F, _P],
    'capmi_step_set_lr': [_P, _F, _P],
    'capmi_rng_bind_epoch': [_P, _P],
    'capmi_adam_step_dyn': [_P, _P, _P, _P, _I64, _P, _F, _F, _F, _F, _F, _F, _P],
    'capmi_ciderd_score': [_P, _I, _I, _P, _P, _P, _I, _I, _P, _P, C.c_uint32, C.c_double, _P, _P],
    'capmi_ciderd_cook_refs': [_P, _P, _I, _I, _I, _P, _P, C.c_uint32, C.c_double, _P, _P],
    'capmi_ciderd_score_cooked': [_P, _I, _I, _P, _P, _P, _I, _P, _P, C.c_uint32, C.c_double, _P, _P],
    'capmi_scst_advantage': [_P, _I, _I, _P, _P],
    'capmi_scst_advantage_mean': [_P, _I, _I, _P, _P, _P],
    'capmi_prof_enable': [_I],
    'capmi_prof_reset': [],
    'capmi_prof_read': [_I, C.POINTER(C.c_double), C.POINTER(C.c_int64), C.POINTER(C.c_double), C.POINTER(C.c_double)],
    'capmi_beam_select': [_P, _P, _I, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P],
    'capmi_beam_reorder': [_P, _P, _P, _I, _I, _I, _I, _I, _P],
    'capmi_beam_logsoftmax': [_P, _P, _I, _I, _F, _I, _P],
    'capmi_updown_beam_search': [C.POINTER(UpDownWeights), C.POINTER(UpDownBeam), _P],
    'capmi_updown_decode_step': [C.POINTER(UpDownWeights), C.POINTER(UpDownBeam), _I, _I, _P, _P, _I, _P],
    'capmi_decode_constrain': [_P, _I, _I, _P, _I, _I, _P, _I, _P, _I, _I, _I, _P],
    'capmi_beam_diversity': [_P, _P, _I, _I, _I, _P, _I, _I, _F, _P],
    'capmi_column_penalty': [_P, _I, _I, _P, _I, _I, _F, _P],
    'capmi_select_logp': [_P, _I, _I, _I, _I, _I, _F, _P, _U64, _P, _I, _P, _P, _P, _P, _I, _P, _P],
    'capmi_layernorm_fwd': [_P] * 6 + [_I, _I, _F, _P],
    'capmi_layernorm_bwd': [_P] * 6 + [_I, _P, _I, _I, _F, _P],
    'capmi_layernorm_bwd_parts_rows': [_I],
    'capmi_layernorm_bwd_parts': [_P, _P, _P, _P, _P, _P, _I, _P, _P, _I, _I, _F, _P],
    'capmi_mha_fwd': [_P, _P, _P, _I, _I, _P, _I, _I, _I, _I, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P],
    'capmi_mha_bwd': [_P, _P, _P, _P, _I, _I, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P],
    'capmi_mha_fwd_s': [_P, _I, _P, _P, _I, _I, _P, _I, _I, _I, _I, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P],
    'capmi_mha_bwd_s': [_P, _P, _I, _P, _P, _I, _I, _P, _P, _P, _I, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P],
    'capmi_glu_fwd': [_P, _P, _P, _P, _I, _I, _P],
    'capmi_glu_fwd_fused': [_P, _I, _I64, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _P],
    'capmi_glu_bwd': [_P, _P, _P, _P, _I, _I, _P],
    'capmi_glu_bwd_add': [_P, _P, _P, _I, _I64, _P, _P, _P, _I, _I, _P],
    'capmi_layernorm_bwd_slabs': [_P, _I, _I64, _P, _P, _P, _P, _P, _P, _I, _I64, _I, _P, _P, _I, _I, _F, _P],
    'capmi_mha_fwd_qslabs': [_P, _I, _I, _I64, _P, _P, _P, _P, _I, _I, _P, _I, _I, _I, _I, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P],
    'capmi_mha_bwd_slabs': [_P, _I, _I64, _I, _P, _I, _P, _P, _I, _I, _P, _P, _P, _I, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P],
    'capmi_gemm_set_policy': [_I],
    'capmi_gemm_group_tn': [_P, _I, _P, _I64, _P],
    'capmi_split_halves': [_P, _I, _I64, _P, _P, _P, _P, _I, _I, _P],
    'capmi_meanpool_fwd': [_P, _P, _P, _I, _I, _I, _P],
    'capmi_meanpool_bwd': [_P, _P, _P, _I, _I, _I, _I, _P],
    'capmi_embed_pe_fwd': [_P, _I, _P, _P, _P, _P, _I, _I, _I, _I, _P],
    'capmi_embed_pe_bwd': [_P, _I, _P, _P, _P, _I, _I, _I, _P],
    'capmi_log_softmax_rows': [_P, _P, _I, _I, _P],
    'capmi_caption_stats': [_P, _P, _I, _I, _I, _P, _P, _P, _P],
    'capmi_maxout_cell_fwd': [_P, _I] + [_P] * 8 + [_I, _I, _P],
    'capmi_maxout_cell_bwd': [_P] * 9 + [_I, _I, _P],
    'capmi_newfc_rollout_fwd': [C.POINTER(NewFCWeights), C.POINTER(NewFCRollout), _P],
    'capmi_newfc_rollout_bwd': [C.POINTER(NewFCWeights), C.POINTER(NewFCRollout), _P, C.POINTER(NewFCBwdScratch),
                                C.POINTER(NewFCGrads), _P],
    'capmi_updown_rollout_fwd': [C.POINTER(UpDownWeights), C.POINTER(UpDownRollout), _P],
    'capmi_updown_rollout_bwd': [C.POINTER(UpDownWeights), C.POINTER(UpDownRollout), _P, C.POINTER(UpDownBwdScratch),
                                 C.POINTER(UpDownGrads), _P],
    'capmi_updown_rollout_bwd_phases': [C.POINTER(UpDownWeights), C.POINTER(UpDownRollout), _P, C.POINTER(UpDownBwdScratch),
                                 C.POINTER(UpDownGrads), _I, _P],
}


class CapmiError(RuntimeError):
    pass


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            'libcapmi.so not found at %s -- the HIP backend is mandatory (no CPU fallback). '
            'Build it: python -m imagecaptioning.pytorch_amd.build' % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, argtypes in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:  # pragma: no cover
            raise ImportError('libcapmi.so does not export %s (stale build?)' % name) from e
        fn.argtypes = argtypes
        fn.restype = (C.c_char_p if name == 'capmi_arch' else
                      C.c_int64 if name in ('capmi_planes_bytes', 'capmi_updown_planes_bytes',
                                                   'capmi_updown_bwd_planes_bytes')
                      else C.c_int)
    return lib


lib = _load()


def check(rc, what):
    if rc != 0:
        raise CapmiError('%s failed: %s' % (what, 'invalid argument' if rc == EINVAL else 'hipError %d' % rc))


def ptr(t):
    """device pointer of a tensor (None -> NULL)."""
    return None if t is None else t.data_ptr()


_raw_stream = getattr(torch._C, '_cuda_getCurrentRawStream', None)
_cur_device = getattr(torch._C, '_cuda_getDevice', None)


def stream_ptr():
    """the current HIP stream of the current device as the integer the C ABI takes.  (r4: torch.cuda.current_stream() builds a
    Stream object through three Python layers -- 4 us per launch, 3 ms of an AoA step's 744 launches; the two C entry points
    below are what it ends up calling)"""
    if _raw_stream is not None and _cur_device is not None:
        return _raw_stream(_cur_device())
    return torch.cuda.current_stream().cuda_stream
