/*
 * capmi.h -- C ABI of libcapmi.so, the MI355X (gfx950 / CDNA4) caption-decoding backend.
 *
 * The reference (ruotianluo/ImageCaptioning.pytorch) has NO native/FFI boundary: its hot path is
 * Python calling torch.nn ops.  This header is therefore the boundary a
 * maintainer would bind with ctypes from the reference's own Python classes (INTEGRATION.md shows
 * the stubs).  Every entry point below names the reference code it replaces (file:line under
 * /root/reference).
 *
 * Conventions
 *   - plain C: raw device pointers, explicit sizes/strides, no torch types;
 *   - all tensors fp32 row-major unless stated; token ids int64 (torch.long) as in the reference;
 *   - `stream` is a hipStream_t (torch.cuda.current_stream().cuda_stream); calls only enqueue work:
 *     they never allocate, never synchronise, never throw;
 *   - return value: 0 on success, a hipError_t (>0) from the launch, or CAPMI_EINVAL (-1) on bad
 *     arguments.  Nothing here falls back to the CPU.
 */
#ifndef CAPMI_H
#define CAPMI_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CAPMI_EINVAL (-1)
#define CAPMI_MAX_SEG 4
/* GEMM workspace layout: the first CAPMI_WS_COUNTER_FLOATS 4-byte words are per-tile arrival tickets for
 * the in-launch split-K reduction (must be ZERO when a workspace is first used; kernels re-arm them),
 * K-slice slabs follow.  A fused consumer of deferred slices reads them at partial + this offset. */
#define CAPMI_WS_COUNTER_FLOATS 16384

/* library / device introspection -------------------------------------------------------------- */
int capmi_version(void);                 /* ABI version, bumped on any signature change */
const char *capmi_arch(void);            /* "gfx950" */

/* ---------------------------------------------------------------------------------------------
 * fp32 MFMA GEMM (v_mfma_f32_32x32x2_f32; exact-f32 products, f32 accumulate).
 * Replaces the addmm/mm calls behind nn.Linear / nn.LSTMCell on the hot path:
 *   AttModel.py:118-122 (fc_embed, att_embed, ctx2att), :628/:635 (LSTMCell gate GEMMs),
 *   :733 (h2att), :172 (logit), and their autograd backward (dX = dY W, dW = dY^T X).
 *
 *   C[M,N] = epi( sum_s opA(A_s)[M,K_s] * opB(B_s)[K_s,N] )
 *
 * a_layout: 0 = A_s stored [M][K_s] (row stride lda)      1 = A_s stored [K_s][M] (row stride lda)
 * b_layout: 0 = B_s stored [N][K_s] (nn.Linear weight)    1 = B_s stored [K_s][N]
 * a_row_div: operand row r reads stored row r / a_row_div (image-sharing of per-image activations,
 *            replaces models/utils.py:3-14 repeat_tensors; a_layout 0 only, >= 1).
 * epilogue (applied when the K reduction is complete):
 *   v = acc + bias[n] + bias2[n] + row_bias[(m / row_bias_div) * N + n]
 *   if relu: v = max(v, 0);  if mul_mask: v *= mul_mask[m * N + n];  if accumulate: v += (addend ? addend : C)[m*ldc+n]
 * split-K: splits > 1 writes raw K-slice sums to the workspace `partial` (layout: see
 *   CAPMI_WS_COUNTER_FLOATS; slabs are [splits][M][N]).  Unless defer_reduce, the slices are combined
 *   and the epilogue applied INSIDE the launch by the last-arriving workgroup of each tile (skinny
 *   path) or by a follow-up reduce kernel (fat path).  splits == 0 lets the library choose.
 * ------------------------------------------------------------------------------------------- */
typedef struct capmi_gemm_seg {
    const float *A;
    const float *B;
    int lda;
    int ldb;
    int K;
    int a_row_div;
} capmi_gemm_seg;

typedef struct capmi_gemm_desc {
    capmi_gemm_seg seg[CAPMI_MAX_SEG];
    int nseg;
    int a_layout;
    int b_layout;
    int M;
    int N;
    float *C;
    int ldc;
    const float *bias;
    const float *bias2;
    const float *row_bias;
    int row_bias_div;
    const float *mul_mask;
    int relu;
    int accumulate;
    float *partial;
    int64_t partial_capacity;   /* in floats */
    int splits;                 /* 0 = auto */
    int defer_reduce;           /* leave partials for a fused consumer (e.g. capmi_lstm_cell_fwd) */
    int splits_used;            /* out: number of K slices actually written */
    /* optional (M <= 64 decode GEMMs): segment s's activations ALSO delivered pre-split as "A planes" (see
     * capmi_planes_from_f32); with planes for every segment the loader / consumer kernel stages them by LDS-DMA instead of
     * splitting them inside every workgroup.  Same result bit for bit. */
    const void *a_planes[CAPMI_MAX_SEG];
    /* optional, with accumulate: v += addend[m*ldc+n] instead of C's previous content -- a residual stream that has to stay
     * intact for the backward (x + sublayer(norm(x)), TransformerModel.py:99-102) is added in the epilogue without a copy of it
     * into C first.  Same row pitch as C; may not overlap C. */
    const float *addend;
    /* r6: with defer_reduce, the caller states PER CALL that nothing runs beside this GEMM on another stream, so the planner may give
     * it 256 x 128 tiles (a wide workgroup owns its CU; see capmi_gemm_set_policy, whose process-wide flag this replaces for callers
     * that know -- ops.DeferredGrads without its side stream).  0: the process-wide policy decides. */
    int allow_wide_deferred;
} capmi_gemm_desc;

int capmi_gemm_f32(capmi_gemm_desc *d, void *stream);
/* r5 planner policy.  The fat GEMMs (bf16x3) run on 128 x 128 or 256 x 128 tiles, whichever the planner costs lower; a 256 x 128
 * workgroup (16 waves, 144 KB of LDS) owns its CU, so GEMMs whose reduction is DEFERRED -- the weight gradients a trainer may run on
 * a side stream beside its backward chain (train.py:193 `loss.backward()` has no such notion: autograd runs them in line) -- are
 * kept on 128 x 128 tiles unless the caller states that nothing runs beside them: allow_wide_deferred != 0.  Process-wide, returns
 * the previous setting; default 0. */
int capmi_gemm_set_policy(int allow_wide_deferred);

/* r6: n INDEPENDENT weight-gradient GEMMs in one launch,
 *     C_i [M_i, N_i] (row pitch ldc) (+)= A_i^T B_i,   A_i [K_i, M_i] (pitch lda), B_i [K_i, N_i] (pitch ldb)
 * -- the dW = dY^T X products that autograd's backward (train.py:193 `loss.backward()`; nn.Linear / nn.LSTMCell weight gradients
 * behind AttModel.py:615-640, TransformerModel.py:84-160, AoAModel.py:100-186) produces one by one and that nothing reads before the
 * optimizer.  All their output tiles form ONE persistent grid on the 256 x 128 bf16x3 kernel: the full rounds of the grid write whole-K
 * tiles straight into C, the last partial round is cut into K slices ([256 x 128] pieces in `slabs`, <= 256 of them per launch)
 * that a small second launch sums -- instead of n sub-wave grids with a K split, a slab round trip and a prologue / tail each.
 * The item table travels in the kernel arguments (no device table, no upload: the call is capturable into a hipGraph); more than
 * ~40 items go out as several launches, longest K first.  Items the fat kernel cannot take (M or N not a multiple of 4 -- any K is fine --, unaligned
 * operands / pitches) are issued through capmi_gemm_f32 behind the group.  Same numbers, bit for bit, as capmi_gemm_f32 on each
 * item with the K split reported in splits_used.
 * slabs / slab_floats: scratch for the K-slice pieces (>= 8.4 M floats serves any group; less only limits the tail's K split). */
typedef struct capmi_group_gemm {
    const float *A, *B;
    float *C;
    int32_t lda, ldb, ldc, K, M, N;
    int32_t accumulate;          /* C += instead of C = */
    int32_t splits_used;         /* out: K slices of this item's tail tiles (1 when every tile was written whole-K); -1: went through capmi_gemm_f32 */
    float *colsum;               /* optional: colsum[m] = sum_k A[k, m] -- the BIAS gradient that goes with the weight gradient (nn.Linear:
                                  * db = column sums of dY).  Taken by the staging waves from the A panel they stage anyway instead of a
                                  * second pass over dY (capmi_colsum_batch: 1.0 ms of a Transformer XE step); deterministic (fixed order). */
} capmi_group_gemm;
int capmi_gemm_group_tn(capmi_group_gemm *items, int n, float *slabs, int64_t slab_floats, void *stream);

/* "A planes" of an activation matrix X[M <= 64, K] (row pitch ld): K in chunks of 32, chunk = [3 planes][64 rows][32 bf16],
 * x = h + m + l split exactly into three truncated bf16 values, 16-byte pieces of a row XOR-swizzled by (row >> 2) & 3 -- the
 * LDS image the decode GEMM reads its MFMA fragments from.  The buffer (capmi_planes_bytes(K) bytes, 16-byte aligned) must
 * be zero-filled ONCE when it is allocated: rows >= M and columns >= K are never written.  On the hot path the producers of
 * the activations (capmi_lstm_cell_fwd_pl, capmi_attention_fwd_partial_pl, the select kernel's next-token embedding,
 * capmi_lstm_cell_bwd_partial_pl) write the planes themselves; this call converts any other operand. */
int64_t capmi_planes_bytes(int K);
int capmi_planes_from_f32(const float *X, int ld, int M, int K, void *planes, void *stream);

/* ---------------------------------------------------------------------------------------------
 * Fused additive region attention, forward (AttModel.py:728-748 Attention.forward; the
 * 7-9 ATen launches of SURVEY.md K5 in one kernel):
 *   e[r,k]   = sum_a w[a] * tanh(p_att[r/n, k, a] + att_h[r, a]) + b
 *   alpha    = softmax_k(e);  if mask: alpha = alpha*mask[r/n,k] / sum_k(alpha*mask)
 *   ctx[r,:] = sum_k alpha[r,k] * att[r/n, k, :]
 * One workgroup per image: its p_att / att tiles are read from HBM once and reused by the n rows
 * of that image (no repeat_tensors copy).  att_h [N,A], p_att [B,K,A], att [B,K,R], mask [B,K] or
 * NULL, w [A], b scalar pointer; outputs ctx [N,R], alpha [N,K].  N = B*n, n <= 8.
 * ------------------------------------------------------------------------------------------- */
/* row_img (optional, int32 [N]): explicit row -> image map for ragged groupings (the fused
 * greedy+sample SCST rollout: B*n sampled rows on train-mode features followed by B greedy rows on
 * eval-mode features).  With row_img the call takes N rows (n is ignored) and one row per workgroup. */
int capmi_attention_fwd(const float *att_h, const float *p_att, const float *att, const float *mask,
                        const float *w, const float *b, float *ctx, float *alpha,
                        int B, int n, int K, int A, int R, const int32_t *row_img, int N, void *stream);

/* Same, with att_h delivered as the h2att GEMM's K-slice slabs (capmi_gemm_f32, defer_reduce = 1):
 * att_h[r,:] = sum_{s<h_splits} h_partial[s*h_stride + r*A + :] + h_bias.  The finished rows are also written to
 * att_h_out [N,A] when given (the backward pass needs them). */
int capmi_attention_fwd_partial(const float *h_partial, int h_splits, int64_t h_stride, const float *h_bias,
                                float *att_h_out, const float *p_att, const float *att, const float *mask,
                                const float *w, const float *b, float *ctx, float *alpha,
                                int B, int n, int K, int A, int R, const int32_t *row_img, int N, void *stream);
/* Same; ctx is also written as "A planes" (capmi_planes_from_f32 layout, N <= 64) for the language-LSTM gate GEMM. */
int capmi_attention_fwd_partial_pl(const float *h_partial, int h_splits, int64_t h_stride, const float *h_bias,
                                   float *att_h_out, const float *p_att, const float *att, const float *mask,
                                   const float *w, const float *b, float *ctx, float *alpha,
                                   int B, int n, int K, int A, int R, const int32_t *row_img, int N, void *ctx_planes,
                                   void *stream);

/* backward of the above for one time step.  Inputs d_ctx [N,R] plus the saved att_h/alpha.
 * Outputs d_att_h [N,A] (feeds h2att backward) and d_e [N,K] (softmax-input gradient, kept for the
 * time-batched parameter/feature gradients below). */
int capmi_attention_bwd(const float *d_ctx, int ld_dctx, const float *att_h, const float *alpha,
                        const float *p_att, const float *att, const float *mask, const float *w,
                        float *d_att_h, float *d_e, int B, int n, int K, int A, int R,
                        const int32_t *row_img, int N, void *stream);
/* Same, with d_ctx delivered as the first R of x_cols columns of a dX GEMM left as K-slice slabs (capmi_gemm_f32,
 * defer_reduce = 1): x[r,:] = sum_{s<x_splits} x_slabs[s*x_stride + r*x_cols + :].  Every workgroup finishes the
 * reduction of its rows over all x_cols columns and publishes them to x_out [N, x_cols] (so the LSTM-cell backward and
 * the time-batched pass read finished rows) -- the dX GEMM needs no reduce launch.  x_cols % 4 == 0, 16-byte aligned. */
int capmi_attention_bwd_partial(const float *x_slabs, int x_splits, int64_t x_stride, int x_cols, float *x_out,
                                const float *att_h, const float *alpha, const float *p_att, const float *att,
                                const float *w, float *d_att_h, float *d_e, int B, int n, int K, int A, int R,
                                const int32_t *row_img, int N, void *stream);

/* time-batched feature/parameter gradients of the attention over a whole rollout of T steps:
 *   d_att[b,k,:]   += sum_{t, r in image b} alpha[t,r,k] * d_ctx[t,r,:]
 *   d_p_att[b,k,a] += sum_{t, r in b} d_e[t,r,k] * w[a] * (1 - tanh^2(p_att[b,k,a] + att_h[t,r,a]))
 *   d_w[a]         += sum_{t,r,k} d_e[t,r,k] * tanh(p_att[b,k,a] + att_h[t,r,a]);  d_b += sum d_e
 * All *_all arrays are [T,N_stride,...] (d_ctx_all rows have pitch ld_dctx); image b owns rows
 * b*n .. b*n+n-1 of every time slab (rows >= B*n, e.g. greedy rows, are ignored).  d_att/d_p_att/d_w/d_b
 * are overwritten (not accumulated). */
int capmi_attention_bwd_batched(const float *d_ctx_all, int ld_dctx, const float *att_h_all, const float *alpha_all,
                                const float *d_e_all, const float *p_att, const float *w,
                                float *d_att, float *d_p_att, float *d_w, float *d_b,
                                int T, int B, int n, int N_stride, int K, int A, int R, void *stream);
/* Same; with dw_partial ([B*K, A] floats of scratch) the alpha_net weight gradient is left as one partial row per (image,
 * region) -- d_w = column sum of dw_partial, to be taken by the caller (capmi_colsum / capmi_colsum_batch_args; d_w itself is
 * not touched and may be NULL) -- instead of B*K*A atomicAdds onto A addresses. */
int capmi_attention_bwd_batched_ws(const float *d_ctx_all, int ld_dctx, const float *att_h_all, const float *alpha_all,
                                   const float *d_e_all, const float *p_att, const float *w,
                                   float *d_att, float *d_p_att, float *d_w, float *d_b,
                                   int T, int B, int n, int N_stride, int K, int A, int R, float *dw_partial, void *stream);

/* ---------------------------------------------------------------------------------------------
 * LSTM cell pointwise stage (torch.nn.LSTMCell gate math, gate order i,f,g,o; AttModel.py:628,635):
 *   gates = sum_s partial[s] + b_ih + b_hh + row_bias[(r / row_bias_div)]   ([N,4R])
 *   c' = sig(f)*c + sig(i)*tanh(g);  h' = sig(o)*tanh(c')
 * Consumes the split-K partials of capmi_gemm_f32 directly (defer_reduce).  Writes h', c', the
 * activated gates [N,4R] (saved for backward) and, if out_mask, h_drop = h' * out_mask
 * (F.dropout at AttModel.py:637).
 * ------------------------------------------------------------------------------------------- */
/* row_bias_idx (optional int32 [N]) replaces r / row_bias_div as the row_bias row index. */
int capmi_lstm_cell_fwd(const float *partial, int splits, const float *b_ih, const float *b_hh,
                        const float *row_bias, int row_bias_div, const int32_t *row_bias_idx, const float *c_prev,
                        float *h, float *c, float *gates_act, const float *out_mask, float *h_drop,
                        int N, int R, void *stream);
/* Same; h and / or h_drop are also written as "A planes" (capmi_planes_from_f32 layout, N <= 64, either may be NULL) for the
 * GEMMs that consume them next (h2att + both LSTM gate GEMMs / the vocabulary projection). */
int capmi_lstm_cell_fwd_pl(const float *partial, int splits, const float *b_ih, const float *b_hh,
                           const float *row_bias, int row_bias_div, const int32_t *row_bias_idx, const float *c_prev,
                           float *h, float *c, float *gates_act, const float *out_mask, float *h_drop,
                           int N, int R, void *h_planes, void *h_drop_planes, void *stream);
/* Same; the gate pre-activations are the sum of TWO slab sets (same slab stride N*4R): `partial` and `partial2` (K segments of
 * the gate GEMM computed by separate launches meet here).  splits2 = 0: one set. */
int capmi_lstm_cell_fwd_pl2(const float *partial, int splits, const float *partial2, int splits2, const float *b_ih,
                            const float *b_hh, const float *row_bias, int row_bias_div, const int32_t *row_bias_idx,
                            const float *c_prev, float *h, float *c, float *gates_act, const float *out_mask, float *h_drop,
                            int N, int R, void *h_planes, void *h_drop_planes, void *stream);

/* backward: given dh (total gradient reaching h'), dc_next (gradient reaching c' from step t+1),
 * the saved activated gates, c_prev and c': d_gates [N,4R] (pre-activation) and dc_prev [N,R].
 * dh may be the sum of up to three addends (dh_a + dh_b + dh_c, each optional) and dh_a may be
 * multiplied by a dropout mask first (dropout backward of AttModel.py:637). */
int capmi_lstm_cell_bwd(const float *dh_a, int ld_a, const float *dh_a_mask, const float *dh_b, int ld_b,
                        const float *dh_c, int ld_c, const float *dc_next, const float *gates_act,
                        const float *c_prev, const float *c_new, float *d_gates, float *dc_prev, int N, int R,
                        void *stream);
/* Same, with dh_b and/or dh_c delivered as split-K slabs of the dX GEMMs (capmi_gemm_f32, defer_reduce = 1):
 * dh_b[r,j] = sum_{s<b_splits} dh_b[s*b_stride + r*ld_b + j] (likewise dh_c); splits = 1 is a plain matrix. */
int capmi_lstm_cell_bwd_partial(const float *dh_a, int ld_a, const float *dh_a_mask,
                                const float *dh_b, int ld_b, int b_splits, int64_t b_stride,
                                const float *dh_c, int ld_c, int c_splits, int64_t c_stride,
                                const float *dc_next, const float *gates_act, const float *c_prev,
                                const float *c_new, float *d_gates, float *dc_prev, int N, int R, void *stream);
/* Same; d_gates [N,4R] is also written as "A planes" (N <= 64) for the dX GEMM of the BPTT step. */
int capmi_lstm_cell_bwd_partial_pl(const float *dh_a, int ld_a, const float *dh_a_mask,
                                   const float *dh_b, int ld_b, int b_splits, int64_t b_stride,
                                   const float *dh_c, int ld_c, int c_splits, int64_t c_stride,
                                   const float *dc_next, const float *gates_act, const float *c_prev,
                                   const float *c_new, float *d_gates, float *dc_prev, int N, int R,
                                   void *d_gates_planes, void *stream);

/* token embedding: x[r,:] = relu(E[it[r],:]) * mask[r,:]  (AttModel.py:74-76,168). relu/mask optional. */
/* it[r*it_stride] is row r's token; it_save [N] (optional) records the tokens consumed. */
int capmi_embed_fwd(const int64_t *it, int it_stride, int64_t *it_save, const float *E, const float *mask,
                    float *x, int N, int Edim, int relu, void *stream);
/* Same; x is also written as "A planes" (N <= 64) for the attention-LSTM gate GEMM. */
int capmi_embed_fwd_pl(const int64_t *it, int it_stride, int64_t *it_save, const float *E, const float *mask,
                       float *x, int N, int Edim, int relu, void *x_planes, void *stream);
/* scatter-add backward into dE [V1,Edim] (caller zeroes dE): dE[it[r]] += dx[r]*mask[r]*(x[r]>0) */
int capmi_embed_bwd(const int64_t *it, const float *dx, const float *x_saved, const float *mask,
                    float *dE, int rows, int Edim, int relu, void *stream);

/* ---------------------------------------------------------------------------------------------
 * log-softmax + next-token choice + rollout bookkeeping for one step
 * (F.log_softmax AttModel.py:172; CaptionModel.sample_next_word CaptionModel.py:370-407;
 *  unfinished masking / seq and seqLogprobs stores AttModel.py:340-347).
 *   logits [N,V1] -> logp = logits - logsumexp
 *   mode 0 greedy: first maximal index (torch.max tie-break)
 *   mode 1 sample: argmax_v(logp[v]/temperature + gumbel(v)), gumbel from `gumbel` [N,V1] if given
 *                  else Philox(seed, step, row, v)
 *   mode 2 forced: token = forced[r*forced_ld + step]
 *   step > 0: token *= unfinished[r]; logp row *= unfinished[r]   (finished rows emit pad=0 / zeros)
 *   unfinished[r] = (step==0 ? 1 : unfinished[r]) && token != 0
 *   (mode 2 with no_finish_mask = 1: teacher forcing a_5, nothing is masked)
 * writes seq[r*seq_ld+step], it_next[r], seq_logp[(r*L+step)*V1 + v] (dense, API parity),
 * sel_logp[r*L+step], live[r*L+step] (1 = the row was unfinished when this step was produced).
 * row_mode [N] (optional) overrides `mode` per row (fused greedy+sample).
 * ------------------------------------------------------------------------------------------- */
int capmi_logsoftmax_select(const float *logits, int N, int V1, int step, int L,
                            int mode, const uint8_t *row_mode, float temperature,
                            const float *gumbel, uint64_t seed,
                            const int64_t *forced, int forced_ld, int no_finish_mask,
                            int64_t *seq, int seq_ld, int64_t *it_next, uint8_t *unfinished,
                            float *seq_logp, float *sel_logp, uint8_t *live, void *stream);

/* Optional tail of capmi_logsoftmax_select_partial: the workgroup that chose row r's token also writes the NEXT step's
 * input embedding x[r,:] = relu?(E[token,:]) * mask[r,:] and it_save[r] = token (capmi_embed_fwd of step t+1 folded in). */
typedef struct {
    const float *E;      /* [V1, Edim] embedding table */
    const float *mask;   /* [N, Edim] dropout mask of the next step or NULL */
    float *x;            /* [N, Edim] out; NULL disables the tail */
    int64_t *it_save;    /* [N] or NULL */
    int Edim;
    int relu;
    void *x_planes;      /* optional (N <= 64): the same rows also as "A planes" (capmi_planes_from_f32) for the next gate GEMM */
    int *alive;          /* optional, independent of x: every row that is still unfinished after this step stores 1 here (device-
                          * visible memory, e.g. pinned host memory: the rollout driver's early exit, AttModel.py:349-350) */
} capmi_next_embed;

/* Optional top-k / nucleus filter of the sampling modes (CaptionModel.sample_next_word, CaptionModel.py:388-404,
 * sample_method 'top<k>' / 'top<p>'): only the top_k most probable tokens, or the smallest set of most probable tokens
 * whose probability (softmax of logp / temperature) reaches top_p, can be drawn.  At most one of the two is non-zero. */
typedef struct {
    int top_k;      /* 0 = off */
    float top_p;    /* 0 = off, else in (0,1) */
} capmi_sample_filter;

/* Same, fed straight from the vocabulary GEMM's K-slice slabs (capmi_gemm_f32 with defer_reduce = 1):
 * logits[r,:] = sum_{s<splits} partial[s*slab_stride + r*V1 + :] + bias (bias may be NULL).  With V1 % 4 == 0,
 * V1 <= 12288 and 16-byte aligned buffers the row lives in registers (no split-K reduce launch, no logits
 * round trip); any other size / alignment runs a streaming kernel that re-assembles the row per pass. */
/* mode flag of the select entry points (OR it into `mode`): store the row of LOGITS (slabs + bias) in seq_logp / sel_logp instead of
 * the log-probabilities -- AttModel.get_logprobs_state(output_logsoftmax=0), AttModel.py:171-175, used by the margin structure losses
 * (loss_wrapper.py:31-37).  The choice of the token is unaffected (arg-max / Gumbel-max are shift invariant). */
#define CAPMI_SELECT_RAW 256
int capmi_logsoftmax_select_partial(const float *partial, int splits, int64_t slab_stride, const float *bias,
                                    int N, int V1, int step, int L,
                                    int mode, const uint8_t *row_mode, float temperature,
                                    const float *gumbel, uint64_t seed,
                                    const int64_t *forced, int forced_ld, int no_finish_mask,
                                    int64_t *seq, int seq_ld, int64_t *it_next, uint8_t *unfinished,
                                    float *seq_logp, float *sel_logp, uint8_t *live,
                                    const capmi_next_embed *next, const capmi_sample_filter *filter, void *stream);
/* r4: the same select with a GEMM riding in the idle workgroups of ITS launch.  `ahead` describes a loader / consumer decode GEMM
 * (M <= 64, every segment with A planes, defer_reduce = 1: K-slice slabs in ahead->partial, ahead->splits_used on return) that
 * depends on nothing this select writes -- in the rollout: the h_lang / h_att segments of the NEXT step's attention-LSTM gates
 * (AttModel.py:626-627), 2/3 of that GEMM's weights.  N select workgroups + the GEMM's workgroups must fit 256; when they do
 * not, or the GEMM / select take another kernel, both are launched one after the other with identical results. */
int capmi_logsoftmax_select_partial_gemm(const float *partial, int splits, int64_t slab_stride, const float *bias, int N,
                                         int V1, int step, int L, int mode, const uint8_t *row_mode, float temperature,
                                         const float *gumbel, uint64_t seed, const int64_t *forced, int forced_ld,
                                         int no_finish_mask, int64_t *seq, int seq_ld, int64_t *it_next, uint8_t *unfinished,
                                         float *seq_logp, float *sel_logp, uint8_t *live, const capmi_next_embed *next,
                                         const capmi_sample_filter *filter, capmi_gemm_desc *ahead, void *stream);

/* gradient of the dense log-probs w.r.t. the logits for ALL steps at once:
 *   dlogits[r,t,:] = g[r,t,:] - exp(logp[r,t,:]) * sum_v g[r,t,v]      (rows where logp was masked
 *   to zero receive zero: `live` [N,L] uint8, 1 = the row was live when step t was produced) */
/* g, seq_logp are [N,L,V1]; only steps t < T are produced; dlogits is written TIME-MAJOR
 * [T,N,V1] so that it lines up with the saved [T,N,...] activations for the batched GEMMs. */
int capmi_logsoftmax_bwd(const float *g, const float *seq_logp, const uint8_t *live, float *dlogits,
                         int N, int L, int T, int V1, void *stream);

/* Sparse form of the same gradient: every criterion of the hot path reads the dense
 * log-probs only through ONE entry per (row, step) -- `input.gather(2, target)` at losses.py:24 (RewardCriterion), :81
 * (StructureLosses), :213 (LanguageModelCriterion) -- and, for LabelSmoothing (:258-262), through the row sum.  With
 *   g_sel [N,L] = dL / d logp[r,t,tok[r,t]]   (NULL: none)      tok [N,tok_ld] int64, tok_ld >= T
 *   g_sum [N,L] = dL / d sum_v logp[r,t,v]    (NULL: none)
 * the gradient w.r.t. the logits is  g_sel (onehot(tok) - p) + g_sum (1 - V1 p),  p = exp(logp); an additional dense
 * gradient `g` [N,L,V1] (NULL: none) is added in its dense form.  The [N,L,V1] gradient tensor of the reference's autograd
 * (zero fill + scatter) is never materialised. */
typedef struct capmi_sparse_logp_grad {
    const float *g_sel;
    const float *g_sum;
    const int64_t *tok;
    int tok_ld;
    int raw;              /* r4: 1 = the rollout returned the raw logits (CAPMI_SELECT_RAW, output_logsoftmax = 0): d(logits) is the loss
                           * gradient itself -- g_sel at the token, g_sum everywhere, plus the dense g -- without the softmax Jacobian */
    const float *scale;   /* NULL, or a DEVICE scalar multiplying g_sel / g_sum (the upstream gradient of a scalar loss) */
} capmi_sparse_logp_grad;
int capmi_logsoftmax_bwd_sparse(const capmi_sparse_logp_grad *sp, const float *g, const float *seq_logp,
                                const uint8_t *live, float *dlogits, int N, int L, int T, int V1, void *stream);

/* RewardCriterion.forward (losses.py:18-37) on the selected log-probs a rollout wrote (capmi_updown_rollout.sel_logp):
 *   mask[r,t] = 1 for t == 0, else (seq[r,t-1] > 0)            (the EOS step still counts, losses.py:28-29)
 *   loss = -sum(sel * reward * mask) / sum(mask)                 (per_row: one value per row, reduction='none')
 *   gcoef[r,t] = d loss / d sel[r,t]  (rows N_used .. N_all-1, e.g. the greedy rows of a fused SCST rollout: 0)
 * reward[r,t] is read at reward[r * reward_row_stride + t * reward_col_stride] ([N] advantage: strides 1, 0). */
int capmi_reward_criterion(const float *sel, int sel_ld, const int64_t *seq, int seq_ld, const float *reward,
                           int reward_row_stride, int reward_col_stride, int N_used, int N_all, int L, int per_row,
                           float *loss, float *gcoef, void *stream);

/* ---------------------------------------------------------------------------------------------
 * Elementwise helpers
 * ------------------------------------------------------------------------------------------- */
/* out = sum_s partial[s] (+bias +bias2) then relu / mask (split-K reduce with the GEMM epilogue) */
int capmi_splitk_reduce(const float *partial, int splits, float *C, int ldc, int M, int N,
                        const float *bias, const float *bias2, const float *row_bias, int row_bias_div,
                        const float *mul_mask, int relu, int accumulate, void *stream);
/* The same for MANY independent GEMMs in one launch: the weight-gradient GEMMs of a layer-by-layer backward (dW = dY^T X of
 * every nn.Linear, TransformerModel.py / AoAModel.py) leave their K-slice slabs (capmi_gemm_desc.defer_reduce = 1, one
 * workspace region each) and nothing reads dW before the optimizer, so ONE launch at the end of the backward finishes them
 * all.  `items` is DEVICE memory (the kernel reads the table); slabs of item i: partial + s * M * N, s < splits. */
typedef struct capmi_reduce_item {
    const float *partial;
    float *C;
    const float *bias;
    int32_t splits, M, N, ldc, accumulate, reserved;
} capmi_reduce_item;
int capmi_splitk_reduce_batch(const capmi_reduce_item *items, int n_items, void *stream);
/* Bernoulli keep-masks scaled by 1/(1-p): mask[i] in {0, 1/(1-p)}; Philox4x32-10(seed, offset+i) */
int capmi_dropout_mask(float *mask, int64_t count, float p, uint64_t seed, uint64_t offset, void *stream);
/* up to CAPMI_MAX_MASKS masks in one launch (the fc / att / xt / output dropouts of one rollout, AttModel.py:83-90, 122,
 * 640).  Viewing mask i as slabs of `rows` rows of `row_len` floats, rows >= keep_from are written as 1.0 (eval-mode rows of
 * the fused SCST rollout); rows == 0 or keep_from < 0: none. */
#define CAPMI_MAX_MASKS 4
typedef struct capmi_mask_desc {
    float *mask;
    int64_t count;
    uint64_t offset;
    int row_len, rows, keep_from;
} capmi_mask_desc;
int capmi_dropout_masks(const capmi_mask_desc *descs, int n, float p, uint64_t seed, void *stream);
/* zero `count` floats of up to four state buffers (h1 / c1 may be NULL), it[0..N) = 0 (BOS), unfinished[0..N) = 1:
 * the initial state of AttModel._sample / _forward (init_hidden AttModel.py:99-102, :281-285) in one launch */
int capmi_rollout_init(float *h0, float *c0, float *h1, float *c1, int64_t count, int64_t *it, uint8_t *unfinished, int N,
                       void *stream);
/* out[c] = sum_r in[r*ld + c]  (bias gradients) ; accumulate optional */
int capmi_colsum(const float *in, int rows, int cols, int ld, float *out, int accumulate, void *stream);
/* all bias gradients of a backward in one launch (no atomics, no zero-fill): `items` is DEVICE memory */
typedef struct capmi_colsum_item {
    const float *in;
    float *out;
    float *out2;                 /* optional second copy of the result (nn.LSTMCell's bias_hh gradient = bias_ih's) */
    int32_t rows, cols, ld, accumulate;
} capmi_colsum_item;
int capmi_colsum_batch(const capmi_colsum_item *items, int n_items, void *stream);
/* the same with the items in HOST memory (<= CAPMI_COLSUM_ARGS_MAX of them, passed to the kernel by value) */
#define CAPMI_COLSUM_ARGS_MAX 8
int capmi_colsum_batch_args(const capmi_colsum_item *host_items, int n_items, void *stream);
/* out[g, c] = sum_{j<group} in[(g*group + j)*cols + c], summed over T slabs of stride slab */
int capmi_group_rowsum(const float *in, int T, int64_t slab, int groups, int group, int cols,
                       float *out, void *stream);
/* y = x * (m ? m : 1) * (gate_on_positive && ref<=0 ? 0 : 1): relu/dropout backward */
int capmi_relu_mask_bwd(const float *dy, const float *y_ref, const float *mask, float *dx, int64_t count,
                        void *stream);
/* the same Jacobian when y_ref is the layer's output AFTER its dropout mask (y = relu(pre) * mask, the fused epilogue of the
 * forward GEMM; nn.Sequential(Linear, ReLU, Dropout): AttModel.py:83-90, TransformerModel.py:215 PositionwiseFeedForward): y > 0
 * exactly where mask and ReLU both passed, and there mask == 1 / (1 - p) == scale -- the mask tensor is not read.
 * count % 4 == 0, 16-byte aligned operands. */
int capmi_relu_scale_bwd(const float *dy, const float *y_ref, float scale, float *dx, int64_t count, void *stream);

/* ---------------------------------------------------------------------------------------------
 * Fused value-clip + Adam on one flat fp32 buffer (torch.nn.utils.clip_grad_value_ train.py:194-195
 * + torch.optim.Adam misc.py:125-126; the flat buffer is also what the single RCCL all-reduce moves).
 *   g = clamp(g * grad_scale, -clip, clip) (clip <= 0: none);  m,v,p updated with bias correction.
 * ------------------------------------------------------------------------------------------- */
int capmi_adam_step(float *p, const float *g, float *m, float *v, int64_t count, float lr, float beta1,
                    float beta2, float eps, float weight_decay, float clip, float grad_scale, int step,
                    void *stream);

/* ---------------------------------------------------------------------------------------------
 * Per-step state in DEVICE memory, so that a whole training iteration (tools/train.py:185-196: forward, loss, backward,
 * clip + Adam) can be captured ONCE into a hipGraph and replayed: what changes from one iteration to the next -- the
 * dropout / sampling random stream and Adam's step count (bias corrections) and learning rate -- is read from this record
 * by the kernels instead of arriving as kernel arguments, which a graph would freeze.
 *   capmi_step_advance   one thread: epoch += 1; adam_step += 1; bc1 = 1 - beta1^adam_step; bc2_sqrt = sqrt(1 - beta2^adam_step)
 *                        (double arithmetic, like capmi_adam_step does on the host).  First launch of every iteration.
 *   capmi_step_set_lr    lr = value (launched outside the graph, when the schedule changes it: misc.py set_lr)
 *   capmi_rng_bind_epoch every kernel that takes a Philox `seed` (dropout masks, token sampling) uses
 *                        seed + 0x9E3779B97F4A7C15 * (*epoch) while a non-NULL epoch pointer is bound (process-wide; one
 *                        process per GPU).  NULL (the default) = the seed argument alone, as before.  Returns the
 *                        previous binding through *prev when prev != NULL.
 *   capmi_adam_step_dyn  capmi_adam_step with lr, bc1, bc2_sqrt read from the record.
 * ------------------------------------------------------------------------------------------- */
typedef struct capmi_step_state {
    uint64_t epoch;
    int32_t adam_step;
    float lr, bc1, bc2_sqrt;
    float reserved[2];
} capmi_step_state;
/* hipMemcpyAsync host -> device on `stream` from PINNED host memory the caller keeps alive and unchanged (inside a graph capture
 * this becomes a memcpy node that re-reads `src` at every replay: the item tables of capmi_splitk_reduce_batch /
 * capmi_colsum_batch of a captured backward). */
int capmi_upload_async(void *dst, const void *src_pinned, int64_t bytes, void *stream);
int capmi_step_advance(capmi_step_state *state, float beta1, float beta2, void *stream);
int capmi_step_set_lr(capmi_step_state *state, float lr, void *stream);
int capmi_rng_bind_epoch(const uint64_t *epoch, const uint64_t **prev);
int capmi_adam_step_dyn(float *p, const float *g, float *m, float *v, int64_t count, const capmi_step_state *state,
                        float beta1, float beta2, float eps, float weight_decay, float clip, float grad_scale,
                        void *stream);

/* ---------------------------------------------------------------------------------------------
 * CIDEr-D reward (external pyciderevalcap.ciderD -- see oracle/ciderd.py for the provenance note;
 * call sites rewards.py:41-81, 83-114).  float64 arithmetic on device.
 *   table: open-addressing hash of the document-frequency pickle; keys = n-gram of <= 4 token ids
 *          packed as 4 x 16-bit (id+1), vals = document count (double); capacity power of two;
 *          empty slot key = 0.
 *   hyp   [H, L] int64 token rows (cut after the first 0, which is kept: rewards.py:33-39)
 *   refs  [B, max_refs, ref_w] int32, n_refs [B]; hypothesis h scores against image hyp_img[h];
 *          a negative entry ends a reference WITHOUT an EOS token (a completely filled row of a narrower
 *          source array padded to ref_w: the zero padding must not read as its terminating 0)
 *   scores[H] double = 10 * mean_k( sum_refs sim_k ) / n_refs
 * ------------------------------------------------------------------------------------------- */
int capmi_ciderd_score(const int64_t *hyp, int H, int L, const int32_t *hyp_img,
                       const int32_t *refs, const int32_t *n_refs, int max_refs, int ref_w,
                       const uint64_t *table_keys, const double *table_vals, uint32_t table_cap,
                       double log_ref_len, double *scores, void *stream);
/* References cooked once per batch: n-gram keys, tf-idf vectors, norms and length of every reference of
 * every image, CAPMI_CIDERD_COOKED_BYTES per (image, reference slot) in `cooked` [B * max_refs]; capmi_ciderd_score_cooked
 * then scores hypotheses against them without re-cooking a reference per hypothesis (same arithmetic, same results). */
#define CAPMI_CIDERD_COOKED_BYTES 4392
int capmi_ciderd_cook_refs(const int32_t *refs, const int32_t *n_refs, int B, int max_refs, int ref_w,
                           const uint64_t *table_keys, const double *table_vals, uint32_t table_cap,
                           double log_ref_len, void *cooked, void *stream);
int capmi_ciderd_score_cooked(const int64_t *hyp, int H, int L, const int32_t *hyp_img, const void *cooked,
                              const int32_t *n_refs, int max_refs, const uint64_t *table_keys,
                              const double *table_vals, uint32_t table_cap, double log_ref_len, double *scores,
                              void *stream);
/* advantage + broadcast (rewards.py:76-79): reward[r] = scores[r] - scores[N + r/n]  (float32 [N]) */
int capmi_scst_advantage(const double *scores, int N, int n, float *reward, void *stream);
/* same; mean_out[0] = mean of reward[0..N) (what LossWrapper reports as out['reward'], loss_wrapper.py:72) in the same launch */
int capmi_scst_advantage_mean(const double *scores, int N, int n, float *reward, float *mean_out, void *stream);

/* ---------------------------------------------------------------------------------------------
 * Launch instrumentation (counterpart of the reference's `time/batch` prints, train.py:198-208).
 * capmi_prof_enable(class_mask): launches of the selected kernel classes carry a HIP event pair in the
 * dispatch itself (hipExtLaunchKernelGGL start/stop events on the kernel's own stream: no extra queue
 * packets, negligible perturbation) and their algorithmic bytes/flops are accumulated; read returns the
 * totals of one class.  Classes with in-dispatch events: 0 decode GEMM (M <= 64, x W^T), 1 BPTT GEMM
 * (M <= 64, dG W), 2 fat GEMM, 3 attention fwd, 4 attention bwd.  capmi_prof_read synchronises on the
 * recorded events.
 * ------------------------------------------------------------------------------------------- */
int capmi_prof_enable(int class_mask);
int capmi_prof_reset(void);
int capmi_prof_read(int cls, double *total_ms, int64_t *launches, double *bytes, double *flops);

/* ---------------------------------------------------------------------------------------------
 * Whole-rollout drivers for the UpDown decoder (AttModel._sample AttModel.py:258-352 and
 * AttModel._forward :126-164 with UpDownCore :615-640).  One host call enqueues every kernel of all
 * T steps on `stream` with no host synchronisation: the reference's per-step
 * `.sum()==0` syncs are replaced by on-device finished flags.
 * ------------------------------------------------------------------------------------------- */
typedef struct capmi_updown_weights {
    const float *embed;                                /* [V1,E]   embed.0.weight */
    const float *att_w_ih, *att_w_hh, *att_b_ih, *att_b_hh;   /* core.att_lstm.*  [4R,2R+E],[4R,R] */
    const float *lang_w_ih, *lang_w_hh, *lang_b_ih, *lang_b_hh; /* core.lang_lstm.* [4R,2R],[4R,R] */
    const float *h2att_w, *h2att_b;                    /* [A,R],[A] */
    const float *alpha_w, *alpha_b;                    /* [A],[1]   */
    const float *logit_w, *logit_b;                    /* [V1,R],[V1] */
} capmi_updown_weights;

typedef struct capmi_updown_rollout {
    /* sizes: B images carry gradients and own rows b*n..b*n+n-1; B_feat >= B feature images exist in
     * fc/att/p_att (B_feat > B only with row_img); N rows in total (N == B*n unless row_img is given) */
    int B, n, N, K, A, R, E, V1;
    int B_feat;
    const int32_t *row_img; /* [N] row -> feature image, or NULL (= row / n) */
    int T;                  /* steps to run */
    int L;                  /* step pitch of seq / seq_logp / sel_logp / live (>= T) */
    /* per-image prepared features (outputs of the prefill GEMMs) */
    const float *fc;        /* [B,R]   */
    const float *att;       /* [B,K,R] */
    const float *p_att;     /* [B,K,A] */
    const float *att_mask;  /* [B,K] or NULL */
    /* randomness / control */
    const float *drop_xt;   /* [T,N,E] pre-scaled keep masks or NULL (eval) */
    const float *drop_out;  /* [T,N,R] or NULL */
    int mode;               /* 0 greedy, 1 sample, 2 forced */
    const uint8_t *row_mode;/* optional per-row override */
    float temperature;
    const float *gumbel;    /* [T,N,V1] injected noise or NULL (=> Philox(seed)) */
    uint64_t seed;
    const int64_t *forced;  /* [N, forced_ld] */
    int forced_ld;
    int teacher;            /* 1: _forward semantics: inputs are forced[:,t], nothing masked */
    /* state + saved activations, all [T+1,N,R] with slot 0 = initial zeros */
    float *h_att, *c_att, *h_lang, *c_lang;
    float *xt;              /* [T,N,E]  */
    int64_t *it_all;        /* [T,N]    token fed at each step */
    float *gates_att;       /* [T,N,4R] activated */
    float *gates_lang;      /* [T,N,4R] */
    float *att_h;           /* [T,N,A]  */
    float *alpha;           /* [T,N,K]  */
    float *ctx;             /* [T,N,R]  */
    float *h_drop;          /* [T,N,R]  h_lang after dropout (logit input) */
    /* outputs */
    int64_t *seq;           /* [N,L]   */
    float *seq_logp;        /* [N,L,V1] dense */
    float *sel_logp;        /* [N,L]   */
    uint8_t *live;          /* [N,L]   1 if row was unfinished when step t was computed */
    /* scratch */
    float *fc_gates;        /* [B,4R]  fc-term of the attention LSTM, computed once */
    float *logits;          /* [N,V1]  */
    int64_t *it;            /* [N]     */
    uint8_t *unfinished;    /* [N]     */
    float *partial;         /* split-K workspace */
    int64_t partial_capacity;
    /* sampling filter (mode 1 rows only): see capmi_sample_filter */
    int top_k;
    float top_p;
    /* scheduled sampling (teacher == 1 only; AttModel.py:145-154): [T,N] or NULL.  ss_mode[t*N + r] says where the INPUT
     * token of row r at step t comes from: 2 = forced[r, t] (teacher forcing), 1 = a draw from the model's own distribution
     * of step t-1 (torch.multinomial(exp(outputs[:, t-1])), here Gumbel-max at temperature 1).  Row t = 0 is ignored (BOS). */
    const uint8_t *ss_mode;
    /* Round 3, optional (N <= 64): scratch for the "A planes" of the step's activations (h_att, h_lang, ctx, xt, h_drop + a zero
     * image; capmi_updown_planes_bytes(R, E) bytes, 16-byte aligned, ZERO-FILLED ONCE by the caller when it is allocated and
     * reusable by later rollouts of the same R / E).  With it the producers of the activations write the bf16x3 planes the
     * gate / logit GEMMs stage by LDS-DMA (capmi_gemm_desc.a_planes); NULL keeps the in-GEMM split.  Same results. */
    void *planes;
    int64_t planes_bytes;
    /* Round 3: early exit of free-running rollouts (AttModel.py:349-350 `if unfinished.sum() == 0: break`).  early_exit = k > 0:
     * after every k-th step (from step early_exit_from on) the driver looks, two steps later and through an event, at a word of
     * `alive_host` ([L] int32 of PINNED HOST memory the device can write: the select kernel of step t stores 1 in alive_host[t]
     * for every row that goes on) and stops enqueuing when no row is left.  The steps that were queued behind the decisive one
     * run on finished rows only and write what the reference leaves behind the break (pad tokens, zero log-probs); the
     * log-probs of the steps never run are zero-filled.  steps_run (out) = steps enqueued; the backward then takes
     * r->T = steps_run.  0 / NULL: all T steps are enqueued without any host wait. */
    int early_exit;
    int early_exit_from;
    int32_t *alive_host;
    int steps_run;
    /* r4, optional (needs `planes`, free-running rollouts): K-slice slab workspace (pre_capacity floats, laid out like `partial`:
     * CAPMI_WS_COUNTER_FLOATS zeroed words + slabs) of the AHEAD part of the attention-LSTM gate GEMM.  The K segments fed by
     * h_lang(t) and h_att(t) -- 2/3 of the next step's gate GEMM, independent of the token being chosen -- are computed inside the
     * select launch of step t (capmi_logsoftmax_select_partial_gemm: the select keeps 60 of 256 CUs busy); the gate GEMM of
     * step t+1 keeps the token-embedding segment and the LSTM cell sums both slab sets.  NULL: one gate GEMM per step. */
    float *pre_partial;
    int64_t pre_capacity;
    /* r4: 1 = free-running rollouts return the raw logits in seq_logp / sel_logp (AttModel._sample with output_logsoftmax = 0,
     * AttModel.py:265, 292: the margin structure losses); the backward then takes the loss gradient as d(logits).  Teacher-forced
     * passes always return log-probabilities (AttModel._forward). */
    int raw_logits;
} capmi_updown_rollout;

int64_t capmi_updown_planes_bytes(int R, int E);

int capmi_updown_rollout_fwd(const capmi_updown_weights *w, capmi_updown_rollout *r, void *stream);

typedef struct capmi_updown_grads {
    float *embed;
    float *att_w_ih, *att_w_hh, *att_b_ih, *att_b_hh;
    float *lang_w_ih, *lang_w_hh, *lang_b_ih, *lang_b_hh;
    float *h2att_w, *h2att_b;
    float *alpha_w, *alpha_b;
    float *logit_w, *logit_b;
    float *d_fc;       /* [B,R]   gradient w.r.t. prepared fc   */
    float *d_att;      /* [B,K,R] gradient w.r.t. prepared att (attention path only) */
    float *d_p_att;    /* [B,K,A] */
} capmi_updown_grads;

typedef struct capmi_updown_bwd_scratch {
    float *dlogits;     /* [T,N,V1]  time-major */
    float *d_hdrop;     /* [T,N,R]     */
    float *dg_att;      /* [T,N,4R]    */
    float *dg_lang;     /* [T,N,4R]    */
    float *d_x2;        /* [T,N,3R]  per step (d_ctx | dh_att | dh_lang_prev) = dg_lang [W_ih | W_hh] */
    float *d_e_all;     /* [T,N,K]     */
    float *d_att_h_all; /* [T,N,A]     */
    float *dh_att_attn; /* [N,R]       */
    float *d_x1;        /* [T,N,2R]  per step (dh_lang_prev | dh_att_prev) = dg_att [W_ih[:, :R] | W_hh] */
    float *dc_att, *dc_lang;   /* [2][N,R] ping-pong */
    float *d_xt_all;    /* [T,N,E]     */
    float *sum_dg_att;  /* [B,4R]      */
    float *w_lang_cat;  /* [4R,3R]  = [lang W_ih | lang W_hh], packed once per BPTT: one dX GEMM per step */
    float *w_att_cat;   /* [4R,2R]  = [att W_ih(:, 0:R) | att W_hh] */
    float *partial;
    int64_t partial_capacity;
    const capmi_sparse_logp_grad *sparse;   /* NULL, or the loss gradient in sparse form (then g_seq_logp may be NULL) */
    /* Rows [0, n_grad_rows) of the rollout carry gradient (0: all N).  The fused SCST rollout (row_img) keeps its greedy-baseline
     * rows behind the sampled ones; with n_grad_rows = B * n the backward runs on the sampled rows only and every [T,N,..]
     * array above is [T,n_grad_rows,..].  It then needs `pack`: >= n_grad_rows * (T * (4R + 2E + 2 + A + K) + R) + 64 floats
     * for gap-free copies of the saved activations. */
    int n_grad_rows;
    float *pack;
    int64_t pack_capacity;
    /* Round 3, optional (gradient rows <= 64): scratch for the A planes of d_gates of the two LSTM cells + a zero image
     * (capmi_updown_bwd_planes_bytes(R) bytes, zero-filled once by the caller); the dX GEMMs of the BPTT then stage their
     * activations by LDS-DMA.  NULL keeps the in-GEMM split. */
    void *planes;
    int64_t planes_bytes;
} capmi_updown_bwd_scratch;

int64_t capmi_updown_bwd_planes_bytes(int R);

/* g_seq_logp [N,T,V1]: gradient w.r.t. the dense log-probs returned by the forward (NULL when s->sparse carries it). */
int capmi_updown_rollout_bwd(const capmi_updown_weights *w, const capmi_updown_rollout *r,
                             const float *g_seq_logp, capmi_updown_bwd_scratch *s,
                             capmi_updown_grads *g, void *stream);

/* The same backward in separately callable phases, in this order, so that a data-parallel caller can start the
 * all-reduce of a gradient bucket while later phases still compute (phases is a bit mask; a full backward is
 * CAPMI_BWD_ALL or the five phases one after the other on the same stream):
 *   LOGIT      -> g->logit_w, logit_b (and d_hdrop for the recurrent part)
 *   RECURRENT  -> the BPTT loop (no parameter gradient yet)
 *   LANG_LSTM  -> g->lang_w_ih, lang_w_hh, lang_b_ih, lang_b_hh
 *   ATT_LSTM   -> g->att_w_ih, att_w_hh, att_b_ih, att_b_hh, embed, d_fc
 *   ATTENTION  -> g->h2att_w, h2att_b, alpha_w, alpha_b, d_att, d_p_att */
#define CAPMI_BWD_LOGIT 1
#define CAPMI_BWD_RECURRENT 2
#define CAPMI_BWD_LANG_LSTM 4
#define CAPMI_BWD_ATT_LSTM 8
#define CAPMI_BWD_ATTENTION 16
#define CAPMI_BWD_ALL 31
int capmi_updown_rollout_bwd_phases(const capmi_updown_weights *w, const capmi_updown_rollout *r,
                                    const float *g_seq_logp, capmi_updown_bwd_scratch *s,
                                    capmi_updown_grads *g, int phases, void *stream);

/* ---------------------------------------------------------------------------------------------
 * Batched beam search for the UpDown decoder, entirely on the device (AttModel._sample_beam
 * AttModel.py:218-256 + CaptionModel.beam_search CaptionModel.py:35-209, group_size 1).
 * Replaces, per step: the full torch.sort over [B, b*V1] (:80) by a segmented top-b selection, the O(t)
 * re-gather of beam histories (:90-103) by parent pointers, the state gather (:105-109) by one reorder
 * kernel, and the 2*B*b `.item()` host syncs of the finished-beam loop (:183-198) by device flags.
 * ------------------------------------------------------------------------------------------- */
/* one selection step.  logp [B*cur, V1] (rows of image b: b*cur .. b*cur+cur-1), sums [B,cur] ->
 * top `bd` candidates per image sorted by descending score (ties: lowest flat index):
 * parent [B,bd] (beam the candidate extends), token [B,bd], score [B,bd] (= joint log-prob, recorded for
 * the finished-beam table), next_sums [B,bd] (= score, minus 1000 where the beam ended: token == 0 or
 * force_end), ended [B,bd]. */
int capmi_beam_select(const float *logp, const float *sums, int B, int cur, int bd, int V1, int force_end,
                      int32_t *parent, int64_t *token, float *score, float *next_sums, uint8_t *ended,
                      void *stream);
/* dst[a][b*bd + j][:] = src[a][b*cur + parent[b,j]][:] for `arrays` stacked [rows,R] arrays */
int capmi_beam_reorder(const float *src, float *dst, const int32_t *parent, int arrays, int B, int cur, int bd,
                       int R, void *stream);
/* out = log_softmax(log_softmax(logits) / temperature)  (CaptionModel.py:203-204 applies it to the already
 * normalised output); optionally out[:, unk_col] -= 1000 (suppress_UNK, :159-160; unk_col < 0: off). */
int capmi_beam_logsoftmax(const float *logits, float *out, int N, int V1, float temperature, int unk_col,
                          void *stream);

typedef struct capmi_updown_beam {
    int B, bd, K, A, R, E, V1, L;
    const float *fc, *att, *p_att, *att_mask;   /* prepared features of the B images */
    float temperature;
    int unk_col;            /* column to suppress by -1000, or -1 */
    /* work: state ping-pong [2][4][B*bd,R] (h_att,c_att,h_lang,c_lang), per-step scratch */
    float *state;
    float *xt, *gates, *att_h, *alpha, *ctx, *fc_gates, *logits;   /* [B*bd,*] scratch, fc_gates [B,4R] */
    int64_t *it;            /* [B*bd] */
    float *sums;            /* [2][B,bd] ping-pong */
    /* outputs, per step t */
    float *logp_rows;       /* [L][B*bd][V1]  normalised rows the selection of step t was made from
                               (step 0 uses only rows b*bd + 0) */
    int32_t *parent;        /* [L][B,bd] */
    int64_t *token;         /* [L][B,bd] */
    float *score;           /* [L][B,bd] */
    uint8_t *ended;         /* [L][B,bd] */
    float *partial; int64_t partial_capacity;
} capmi_updown_beam;

int capmi_updown_beam_search(const capmi_updown_weights *w, capmi_updown_beam *b, void *stream);

/* ONE UpDown decoder step on already prepared features: AttModel.get_logprobs_state (AttModel.py:166-176) up to the raw
 * logits (b->logits [rows,V1]); the caller normalises.  Tokens are read from b->it [rows]; `rows` = b->B *
 * rows_per_image hypotheses, image-major.  state_in / state_out are distinct [4][b->B*b->bd, R] arrays
 * (h_att, c_att, h_lang, c_lang).  first != 0 also (re)computes b->fc_gates.  Uses of b: B, bd (row capacity per image),
 * K, A, R, E, V1, fc, att, p_att, att_mask, xt, gates, att_h, alpha, ctx, fc_gates, logits, it, partial. */
int capmi_updown_decode_step(const capmi_updown_weights *w, capmi_updown_beam *b, int rows, int rows_per_image,
                             const float *state_in, float *state_out, int first, void *stream);

/* ---------------------------------------------------------------------------------------------
 * Decode-time options (eval): the reference edits the [rows,V1] log-probabilities of a step on the host side
 * (AttModel.py:293-330, 391-432; CaptionModel.py:38-57, 152-157).  Here they are in-place device edits.
 * ------------------------------------------------------------------------------------------- */
#define CAPMI_DECODE_NO_REPEAT 1       /* decoding_constraint: the previous token gets -inf              */
#define CAPMI_DECODE_NO_BAD_ENDING 2   /* remove_bad_endings: column 0 gets -inf after a bad-ending word  */
#define CAPMI_DECODE_BLOCK_TRIGRAMS 4  /* block_trigrams: -0.693*2 per earlier occurrence of the trigram  */
/* logp [N,V1] in place.  prev[r*prev_stride] = token row r emitted at step t-1 (flags 1, 2; call only for t > 0).
 * bad_endings [n_bad] token ids.  seq [N,seq_ld] tokens emitted so far (columns 0..t-1; flag 4, rows < trigram_rows
 * only -- the reference loops over the image batch, AttModel.py:310, 322). */
int capmi_decode_constrain(float *logp, int N, int V1, const int64_t *prev, int prev_stride, int flags,
                           const int64_t *bad_endings, int n_bad, const int64_t *seq, int seq_ld, int t, int trigram_rows,
                           void *stream);
/* diverse beam search, CaptionModel.add_diversity (CaptionModel.py:38-57): out [B*cur,V1] = logp - change*lambda where
 * change[b,v] counts v among prev_tokens[b*prev_stride + 0..n_prev-1] (choices of the earlier groups at this local
 * time).  logp != out. */
int capmi_beam_diversity(const float *logp, float *out, int B, int cur, int V1, const int64_t *prev_tokens, int prev_stride,
                         int n_prev, float diversity_lambda, void *stream);
/* AttModel._diverse_sample (AttModel.py:395-397): logp[:, tokens] -= lambda for EVERY row, once per distinct token;
 * tokens[i*token_stride], i < n_tokens.  One call per earlier group. */
int capmi_column_penalty(float *logp, int N, int V1, const int64_t *tokens, int n_tokens, int token_stride,
                         float diversity_lambda, void *stream);
/* Token choice from rows that ALREADY hold (constrained) log-probabilities: nothing is renormalised, the rows are
 * stored as they are times the unfinished flag (AttModel.py:333-347; -inf * 0 = NaN like there).  mode 0 arg-max,
 * 1 Categorical(logits = logp / temperature) with the optional top-k / nucleus filter.  sel_logp [N,L] (optional):
 * the picked token's log-prob times the unfinished flag, or -- sel_unmasked != 0, AttModel._diverse_sample
 * (AttModel.py:436-447) -- the log-prob of the token the sampler picked even for rows that had finished.  Other
 * arguments as in capmi_logsoftmax_select_partial. */
int capmi_select_logp(const float *logp, int N, int V1, int step, int L, int mode, float temperature, const float *gumbel,
                      uint64_t seed, int64_t *seq, int seq_ld, int64_t *it_next, uint8_t *unfinished, float *seq_logp,
                      float *sel_logp, int sel_unmasked, const capmi_sample_filter *filter, void *stream);

/* ---------------------------------------------------------------------------------------------
 * Transformer captioner building blocks (BASELINE configs[3]; TransformerModel.py).  The contractions
 * (QKV / output projections, FFN, generator) run on capmi_gemm_f32 with fused bias / ReLU / dropout /
 * residual epilogues; these are the non-GEMM pieces.
 * ------------------------------------------------------------------------------------------- */
/* custom LayerNorm of the reference (TransformerModel.py:76-87): y = a*(x-mean)/(std_unbiased+eps)+b.
 * x,y [M,D]; saves mean[M] and inv[M] = 1/(std+eps) for the backward. */
int capmi_layernorm_fwd(const float *x, const float *a, const float *b, float *y, float *mean, float *inv,
                        int M, int D, float eps, void *stream);
/* dx [M,D] (accumulate: dx += ... for the residual branch), g_scaled [M,D] = dy*(x-mean)*inv (column-sum it
 * for d_a; column-sum dy for d_b). */
int capmi_layernorm_bwd(const float *dy, const float *x, const float *a, const float *mean, const float *inv,
                        float *dx, int accumulate, float *g_scaled, int M, int D, float eps, void *stream);
/* r6: the same backward with the PARAMETER gradients started in the launch: every workgroup leaves the sums of dy (x - mean) inv and of
 * dy over its own rows (fixed order) as one row of part_a / part_b, [capmi_layernorm_bwd_parts_rows(M), D] each; d_a / d_b are their
 * column sums (capmi_colsum_batch over ~M/8 rows instead of M, and no [M, D] g_scaled is written).  Deterministic.  D <= 2048. */
int capmi_layernorm_bwd_parts_rows(int M);
int capmi_layernorm_bwd_parts(const float *dy, const float *x, const float *a, const float *mean, const float *inv, float *dx,
                              int accumulate, float *part_a, float *part_b, int M, int D, float eps, void *stream);
/* r5, AoA decode step (AoAModel.py:163-186): the same backward when BOTH inputs are still K-slice slabs of the GEMMs that made
 * them -- dy = sum_s dy_slabs[s] ([M,D] slabs dy_stride floats apart; the finished dy is also written to dy_out when not NULL: its
 * column sums are d_b) and the running gradient the LayerNorm term is added to, dx = sum_s acc_slabs[s] + ..., slabs of row pitch
 * acc_ld (a column block of a wider product: the query half of d_cat, AoAModel.py:174) acc_stride floats apart.  Slabs are added
 * in slab order from 0.f, the bits of capmi_splitk_reduce / capmi_split_halves followed by capmi_layernorm_bwd(accumulate=1). */
int capmi_layernorm_bwd_slabs(const float *dy_slabs, int dy_splits, int64_t dy_stride, float *dy_out, const float *x, const float *a,
                              const float *mean, const float *inv, const float *acc_slabs, int acc_splits, int64_t acc_stride,
                              int acc_ld, float *dx, float *g_scaled, int M, int D, float eps, void *stream);
/* multi-head scaled-dot-product attention for short sequences (TransformerModel.py:152-195), one workgroup
 * per (key/value batch row, head).  q [Nq,Tq,D], k,v rows of pitch ldkv floats ([Nkv,Tk,D] or a KV cache
 * [Nkv,Lmax,D]), D = h*dk, heads interleaved along D exactly like `.view(N,-1,h,dk)`.  Query row r attends
 * key/value batch row r / q_per_kv (cross-attention over per-image memory: no repeat_tensors copy).
 * kstride: floats between consecutive keys of one kv row (D for packed [Tk,D]; 2R when K and V are the two halves of
 * AoA's p_att rows, AoAModel.py:168).  mask: uint8 [Nq or Nkv-broadcast, mask_tq (1 or Tq), Tk] (0 = -inf), or NULL; causal != 0 additionally
 * masks key j > query position (q_pos0 + i).  drop: optional pre-scaled keep mask [Nq,h,Tq,Tk] applied to the
 * probabilities.  Outputs o [Nq,Tq,D] and p [Nq,h,Tq,Tk] (softmax probabilities BEFORE dropout, saved).
 * dk % 4 == 0 and 16-byte aligned q / k / v / o rows (ldkv, kstride multiples of 4): the head dimension moves in 16-byte pieces. */
int capmi_mha_fwd(const float *q, const float *k, const float *v, int ldkv, int kstride, const uint8_t *mask, int mask_tq,
                  int mask_per_q, int causal, int q_pos0, const float *drop, float *o, float *p, int Nq, int q_per_kv,
                  int Tq, int Tk, int h, int dk, void *stream);
/* Same with a query row pitch: qstride floats between consecutive query rows (0 = D).  Lets q, k, v be the three column blocks
 * of ONE fused projection y = x [Wq; Wk; Wv]^T of pitch 3D (r4: the three Linear calls of TransformerModel.py:179-181 as one
 * GEMM with N = 3D; k, v then use kstride = 3D). */
int capmi_mha_fwd_s(const float *q, int qstride, const float *k, const float *v, int ldkv, int kstride, const uint8_t *mask,
                    int mask_tq, int mask_per_q, int causal, int q_pos0, const float *drop, float *o, float *p, int Nq,
                    int q_per_kv, int Tq, int Tk, int h, int dk, void *stream);
/* r5: q still as the K-slice slabs of its projection GEMM (q_splits >= 1 slabs of row pitch qstride, q_slab_stride floats apart):
 * every workgroup finishes its head's columns -- slabs in order from 0.f, + q_bias (NULL = none): the bits of capmi_splitk_reduce --
 * and writes them to q_out [Nq,Tq,D] (NULL = not wanted) for the backward.  q_splits = 0: plain capmi_mha_fwd_s. */
int capmi_mha_fwd_qslabs(const float *q, int qstride, int q_splits, int64_t q_slab_stride, const float *q_bias, float *q_out,
                         const float *k, const float *v, int ldkv, int kstride, const uint8_t *mask, int mask_tq, int mask_per_q,
                         int causal, int q_pos0, const float *drop, float *o, float *p, int Nq, int q_per_kv, int Tq, int Tk, int h,
                         int dk, void *stream);
/* backward: d_o [Nq,Tq,D] -> dq [Nq,Tq,D], dk/dv summed over the q_per_kv query rows of a kv row, written at
 * out + kv_row*dkv_ld + key*dkv_stride (+= when accumulate: BPTT over time steps) */
int capmi_mha_bwd(const float *d_o, const float *q, const float *k, const float *v, int ldkv, int kstride, const float *p,
                  const float *drop, float *dq, float *dk_out, float *dv_out, int dkv_ld, int dkv_stride, int accumulate,
                  int Nq, int q_per_kv, int Tq, int Tk, int h, int dk, void *stream);
/* Same with query / dq row pitches (0 = D): dq may be a column block of the fused [rows, 3D] gradient that one dX GEMM
 * (K = 3D) and one dW GEMM consume. */
int capmi_mha_bwd_s(const float *d_o, const float *q, int qstride, const float *k, const float *v, int ldkv, int kstride,
                    const float *p, const float *drop, float *dq, int dq_stride, float *dk_out, float *dv_out, int dkv_ld,
                    int dkv_stride, int accumulate, int Nq, int q_per_kv, int Tq, int Tk, int h, int dk, void *stream);
/* r5: d_o with a row pitch do_ld >= D and possibly still `do_splits` K-slice slabs do_stride floats apart (the attention half of
 * d_cat in an AoA decode step), summed in slab order from 0.f like capmi_split_halves.  do_splits = 1, do_ld = D: capmi_mha_bwd_s. */
int capmi_mha_bwd_slabs(const float *d_o, int do_splits, int64_t do_stride, int do_ld, const float *q, int qstride, const float *k,
                        const float *v, int ldkv, int kstride, const float *p, const float *drop, float *dq, int dq_stride,
                        float *dk_out, float *dv_out, int dkv_ld, int dkv_stride, int accumulate, int Nq, int q_per_kv, int Tq,
                        int Tk, int h, int dk, void *stream);
/* x[r,t,:] = E[tok[r,t]]*sqrt(D) + pe[pos0+t,:], then * drop (TransformerModel.py:215, 231-233) */
int capmi_embed_pe_fwd(const int64_t *tok, int tok_ld, const float *E, const float *pe, const float *drop, float *x,
                       int N, int T, int D, int pos0, void *stream);
/* dE[tok] += dx*drop*sqrt(D)  (caller zeroes dE) */
int capmi_embed_pe_bwd(const int64_t *tok, int tok_ld, const float *dx, const float *drop, float *dE, int N, int T, int D,
                       void *stream);
/* AoA / GLU blocks (AoAModel.py:44, 143): out = [residual +] mask * (pre[:, :R] * sigmoid(pre[:, R:2R])), pre [M,2R] */
int capmi_glu_fwd(const float *pre, const float *mask, const float *residual, float *out, int M, int R, void *stream);
/* The same for one AoA decode step, finishing the att2ctx GEMM's K-slice slabs itself (slabs [splits][M,2R] + bias -> pre, kept
 * for the backward) and writing every consumer's operand: out, out_a = out * mask_a (input of the logit GEMM), out_b = out * mask_b
 * (context input of the NEXT step) -- masks NULL = plain copies, out_a / out_b NULL = not wanted -- the last two also as "A planes"
 * (M <= 64; capmi_planes_from_f32 layout, buffers zero-filled once by the caller).  R % 4 == 0, 16-byte aligned operands. */
int capmi_glu_fwd_fused(const float *slabs, int splits, int64_t stride, const float *bias, float *pre, float *out,
                        const float *mask_a, float *out_a, void *planes_a, const float *mask_b, float *out_b, void *planes_b,
                        int M, int R, void *stream);
/* d_pre [M,2R] from d_out [M,R] (mask applied first) */
int capmi_glu_bwd(const float *d_out, const float *mask, const float *pre, float *d_pre, int M, int R, void *stream);
/* r5: the same with a second gradient source that is still K-slice slabs: g = mask * d_out + add_mask * sum_s add_slabs[s]
 * ([M,R] slabs add_stride floats apart, added in slab order from 0.f, then the mask, then the sum: the bits of a split-K reduction
 * with mul_mask + accumulate into d_out).  In an AoA BPTT step it is the gradient reaching out_t through step t+1's context input
 * (AoAModel.py:165-166), whose GEMM then needs no reduce launch. */
int capmi_glu_bwd_add(const float *d_out, const float *mask, const float *add_slabs, int add_splits, int64_t add_stride,
                      const float *add_mask, const float *pre, float *d_pre, int M, int R, void *stream);
/* out_lo [M,R] = mask_lo * sum_s slabs[s][:, 0:R], out_hi [M,R] = mask_hi * sum_s slabs[s][:, R:2R]: the two halves of a [M,2R]
 * product that is still `splits` K-slice slabs of pitch `stride` floats (or a finished matrix, splits = 1), each through its
 * dropout mask (NULL = none).  The backward of torch.cat([att, query], -1) in the AoA blocks (AoAModel.py:92,174): d_cat = d_pre W
 * feeds two consumers that want contiguous [M,R] operands -- one launch instead of a split-K reduction, two slice copies and
 * two mask multiplies. */
int capmi_split_halves(const float *slabs, int splits, int64_t stride, const float *mask_lo, const float *mask_hi, float *out_lo,
                       float *out_hi, int M, int R, void *stream);
/* masked mean over regions (AoAModel.py:214-219): mean[b,:] = sum_k m[b,k] x[b,k,:] / sum_k m[b,k] (m NULL = ones) */
int capmi_meanpool_fwd(const float *x, const float *mask, float *mean, int B, int K, int D, void *stream);
/* dx[b,k,:] (+)= m[b,k]/cnt * dmean[b,:] */
int capmi_meanpool_bwd(const float *dmean, const float *mask, float *dx, int accumulate, int B, int K, int D, void *stream);
/* r6: caption statistics of the evaluation loop (captioning/utils/eval_utils.py:173-174) from a decode's dense log-probs
 * seq_logp [N, L, V1] and tokens seq [N, L]:
 *   entropy[r]    = -sum_t sum_v softmax(seq_logp[r,t,:])_v * seq_logp[r,t,v] / (count(seq[r,:] > 0) + 1)
 *   perplexity[r] = -sum_t seq_logp[r,t,seq[r,t]]                            / (count(seq[r,:] > 0) + 1)
 * One pass over seq_logp, no dense temporaries (the ATen formula built three).  A -inf log-prob contributes 0 (0 * -inf := 0).
 * scratch: 2 * N * L floats.  V1 <= 32768. */
int capmi_caption_stats(const float *seq_logp, const int64_t *seq, int N, int L, int V1, float *scratch, float *entropy, float *perplexity,
                        void *stream);
/* out = log_softmax(logits) row-wise (Generator, TransformerModel.py:50-57) */
int capmi_log_softmax_rows(const float *logits, float *out, int rows, int V1, void *stream);

/* ---------------------------------------------------------------------------------------------
 * NewFC decoder (BASELINE configs[0], configs/fc.yml): NewFCModel (AttModel.py:904-945) over the maxout
 * LSTMCore (FCModel.py:13-42).  The image embedding is fed as a first LSTM step when the state is all
 * zero (AttModel.py:925-927), then one word per step; log-softmax / choice / bookkeeping are shared
 * with the UpDown path (capmi_logsoftmax_select).
 * ------------------------------------------------------------------------------------------- */
/* maxout cell: sums = sum_s partial[s] + b_i2h + b_h2h ([N,5R]: in, forget, out, cand_a, cand_b);
 * c' = sig(f)*c + sig(in)*max(cand_a,cand_b); h' = sig(out)*tanh(c').  saved [N,5R] = (sig(in), sig(f),
 * sig(out), cand_a, cand_b). */
int capmi_maxout_cell_fwd(const float *partial, int splits, const float *b_i2h, const float *b_h2h,
                          const float *c_prev, float *h, float *c, float *saved, const float *out_mask,
                          float *h_drop, int N, int R, void *stream);
/* dh = dh_a (* dh_a_mask) + dh_b; d_sums [N,5R]; dc_prev [N,R] */
int capmi_maxout_cell_bwd(const float *dh_a, const float *dh_a_mask, const float *dh_b, const float *dc_next,
                          const float *saved, const float *c_prev, const float *c_new, float *d_sums,
                          float *dc_prev, int N, int R, void *stream);

typedef struct capmi_newfc_weights {
    const float *embed;            /* [V1,E]  embed.weight (plain Embedding, AttModel.py:908) */
    const float *i2h_w, *i2h_b;    /* [5R,E],[5R]  _core.i2h */
    const float *h2h_w, *h2h_b;    /* [5R,R],[5R]  _core.h2h */
    const float *logit_w, *logit_b;
} capmi_newfc_weights;

typedef struct capmi_newfc_rollout {
    int B, n, N, R, E, V1, T, L;
    const float *fc_emb;    /* [B,E]  fc_embed(fc_feats) (no ReLU / dropout: AttModel.py:907) */
    const float *drop_out;  /* [T,N,R] keep masks for the LSTMCore output dropout, or NULL */
    int mode; float temperature; const float *gumbel; uint64_t seed;
    const int64_t *forced; int forced_ld; int teacher;
    float *h, *c;           /* [T+2,N,R]  slot 0 zeros, slot 1 after the image step, slot t+2 after word t */
    float *x;               /* [T,N,E]   word embeddings */
    int64_t *it_all;        /* [T,N] */
    float *saved;           /* [T+1,N,5R] cell activations, slot 0 = image step */
    float *h_drop;          /* [T,N,R] */
    int64_t *seq; float *seq_logp; float *sel_logp; uint8_t *live;   /* [N,L], [N,L,V1], [N,L], [N,L] */
    float *logits; int64_t *it; uint8_t *unfinished;
    float *partial; int64_t partial_capacity;
} capmi_newfc_rollout;

typedef struct capmi_newfc_grads {
    float *embed, *i2h_w, *i2h_b, *h2h_w, *h2h_b, *logit_w, *logit_b;
    float *d_fc_emb;   /* [B,E] */
} capmi_newfc_grads;

typedef struct capmi_newfc_bwd_scratch {
    float *dlogits;   /* [T,N,V1] */
    float *d_hdrop;   /* [T,N,R]  */
    float *d_sums;    /* [T+1,N,5R] slot 0 = image step */
    float *dh_prev;   /* [2][N,R] ping-pong */
    float *dc;        /* [2][N,R] */
    float *d_x_all;   /* [T,N,E]  */
    float *d_ximg;    /* [N,E]    */
    float *partial; int64_t partial_capacity;
    const capmi_sparse_logp_grad *sparse;   /* as in capmi_updown_bwd_scratch */
} capmi_newfc_bwd_scratch;

int capmi_newfc_rollout_fwd(const capmi_newfc_weights *w, capmi_newfc_rollout *r, void *stream);
int capmi_newfc_rollout_bwd(const capmi_newfc_weights *w, const capmi_newfc_rollout *r, const float *g_seq_logp,
                            capmi_newfc_bwd_scratch *s, capmi_newfc_grads *g, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* CAPMI_H */
