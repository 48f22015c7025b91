// Host-side drivers that enqueue a whole UpDown rollout (all T decoder steps, forward or BPTT) on one
// HIP stream with no host synchronisation.  This is the native replacement of the Python time loops
// AttModel._sample (AttModel.py:288-350) / AttModel._forward (:144-162) + UpDownCore.forward
// (:624-640) and of the autograd graph torch would record for them.
//
// MI355X-first structure:
//  * per step the recurrent GEMMs read [h_lang | xt | h_att] and [ctx | h_att | h_lang] in place as
//    K segments (no torch.cat), image-shared tensors are indexed row/n (no repeat_tensors);
//  * the fc-feature term of the attention LSTM is constant over time: it is one [B,4R] GEMM per
//    rollout, added in the LSTM-cell epilogue;
//  * everything that is not on the recurrent critical path of BPTT is TIME-BATCHED into fat MFMA
//    GEMMs over all T*N rows after the loop: every weight gradient, the embedding gradient, the
//    logit layer (forward-saved activations are stored time-major [T,N,...] for exactly this);
//    only the 4 skinny "dX" GEMMs + the pointwise cells + the attention Jacobian stay in the loop.
#include <chrono>
#include "capmi_common.h"
#include <cstdlib>
#include "../../../include/capmi.h"

namespace {

#define RC(x)                 \
    do {                      \
        int rc__ = (x);       \
        if (rc__) return rc__;\
    } while (0)

struct SegSpec {
    const float *A;
    int lda;
    const float *B;
    int ldb;
    int K;
    int a_row_div;
    const void *Apl;        // the same activations as A planes (capmi.h capmi_planes_from_f32), or null
};

// chunk images of a K-wide planes buffer
inline int64_t pl_chunks(int K) { return (K + 31) / 32; }
constexpr int64_t PL_CHUNK = 12288;

// C[M,N] = sum_s A_s op B_s ; thin wrapper filling capmi_gemm_desc
int gemm(void *stream, int a_layout, int b_layout, int M, int N, float *C, int ldc, const SegSpec *segs, int nseg,
         float *partial, int64_t cap, int defer, int *splits_used, const float *bias = nullptr,
         const float *bias2 = nullptr, int accumulate = 0, const void *zero_planes = nullptr, int splits_hint = 0) {
    capmi_gemm_desc d{};
    d.nseg = nseg;
    for (int i = 0; i < nseg; ++i) {
        d.seg[i].A = segs[i].A; d.seg[i].lda = segs[i].lda;
        d.seg[i].B = segs[i].B; d.seg[i].ldb = segs[i].ldb;
        d.seg[i].K = segs[i].K; d.seg[i].a_row_div = segs[i].a_row_div > 0 ? segs[i].a_row_div : 1;
        d.a_planes[i] = zero_planes ? segs[i].Apl : nullptr;
    }
    d.a_layout = a_layout; d.b_layout = b_layout;
    d.M = M; d.N = N; d.C = C; d.ldc = ldc;
    d.bias = bias; d.bias2 = bias2;
    d.accumulate = accumulate;
    d.partial = partial; d.partial_capacity = cap;
    d.splits = splits_hint; d.defer_reduce = defer;
    const int rc = capmi_gemm_f32(&d, stream);
    if (splits_used) *splits_used = d.splits_used;
    return rc;
}

// teacher forcing: [T,N,R] (time-major, as the steps wrote it) -> [N,T,R] so that ONE fat GEMM over all T*N rows lands in
// the [N,L,V1] layout of seqLogprobs
__global__ void tn_to_nt_kernel(const float *__restrict__ src, float *__restrict__ dst, int T, int N, int R) {
    const size_t total = (size_t)T * N * R;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % R);
        const size_t row = i / R;                     // = n * T + t
        const int n = (int)(row / T), t = (int)(row % T);
        dst[i] = src[((size_t)t * N + n) * R + c];
    }
}

// [lang W_ih | lang W_hh] -> w_lang_cat [4R,3R] and [att W_ih(:, 0:R) | att W_hh] -> w_att_cat [4R,2R] in one launch (were four
// hipMemcpy2DAsync: 42 us + 4 kernel boundaries per BPTT).  R % 4 == 0 and 16-byte aligned operands (checked by the caller).
__global__ void pack_recurrent_kernel(const float *__restrict__ lang_ih, const float *__restrict__ lang_hh,
                                      const float *__restrict__ att_ih, const float *__restrict__ att_hh,
                                      float *__restrict__ w_lang_cat, float *__restrict__ w_att_cat, int R, int ld_att_ih) {
    const int R4 = R >> 2;
    const int per_row = 5 * R4;                  // 3R + 2R floats of packed output per weight row, in 16-byte pieces
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 4 * R * per_row) return;
    const int row = i / per_row, q = i - row * per_row;
    typedef float f4 __attribute__((ext_vector_type(4)));
    if (q < 3 * R4) {
        const int c = q * 4;
        const f4 v = c < 2 * R ? *reinterpret_cast<const f4 *>(lang_ih + (size_t)row * 2 * R + c)
                               : *reinterpret_cast<const f4 *>(lang_hh + (size_t)row * R + (c - 2 * R));
        *reinterpret_cast<f4 *>(w_lang_cat + (size_t)row * 3 * R + c) = v;
    } else {
        const int c = (q - 3 * R4) * 4;
        const f4 v = c < R ? *reinterpret_cast<const f4 *>(att_ih + (size_t)row * ld_att_ih + c)
                           : *reinterpret_cast<const f4 *>(att_hh + (size_t)row * R + (c - R));
        *reinterpret_cast<f4 *>(w_att_cat + (size_t)row * 2 * R + c) = v;
    }
}

// Rows [0, Nb) of every time slab of up to 9 saved forward tensors ([slots][Nf][C] -> [slots][Nb][C]) in one launch: the fused
// SCST rollout carries its greedy-baseline rows (no gradient) behind the sampled ones, and the backward runs on the sampled
// rows only -- its time-batched GEMMs then need the activations without the gaps.  C in floats (int64 tokens: C = 2).
struct PackItem {
    const float *src;
    float *dst;
    int C, slots;
};
struct PackArgs {
    PackItem it[9];
    int n, Nf, Nb;
};
__global__ void pack_rows_kernel(const PackArgs a) {
    PackItem it = a.it[0];
#pragma unroll
    for (int i = 1; i < 9; ++i)
        if (blockIdx.y == i) it = a.it[i];          // static indices: no scratch copy of the argument table
    if ((int)blockIdx.y >= a.n) return;
    const size_t per_slot = (size_t)a.Nb * it.C, total = per_slot * it.slots;
    const bool vec = it.C % 4 == 0 && ((reinterpret_cast<uintptr_t>(it.src) | reinterpret_cast<uintptr_t>(it.dst)) & 15) == 0;
    typedef float f4 __attribute__((ext_vector_type(4)));
    if (vec) {
        for (size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x; q < total / 4; q += (size_t)gridDim.x * blockDim.x) {
            const size_t i = q * 4, slot = i / per_slot, rem = i - slot * per_slot;
            *reinterpret_cast<f4 *>(it.dst + i) = *reinterpret_cast<const f4 *>(it.src + slot * (size_t)a.Nf * it.C + rem);
        }
    } else {
        for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
            const size_t slot = i / per_slot, rem = i - slot * per_slot;
            it.dst[i] = it.src[slot * (size_t)a.Nf * it.C + rem];
        }
    }
}

// bookkeeping of the teacher-forced pass (what the per-step select kernel writes in mode 2 with no finish mask)
__global__ void teacher_bookkeep_kernel(const float *__restrict__ seq_logp, const int64_t *__restrict__ forced, int forced_ld,
                                        int64_t *__restrict__ seq, float *__restrict__ sel_logp, uint8_t *__restrict__ live,
                                        int N, int L, int V1) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N * L) return;
    const int r = i / L, t = i % L;
    const int64_t tok = forced[(size_t)r * forced_ld + t];
    seq[i] = tok;
    if (sel_logp) sel_logp[i] = seq_logp[(size_t)i * V1 + tok];
    if (live) live[i] = 1;
}

// teacher forcing: the token embeddings (+ ReLU + dropout) of ALL T steps in one launch: x[t][n][:] = relu(E[forced[n][t]]) * mask
__global__ void embed_fwd_all_steps_kernel(const int64_t *__restrict__ forced, int forced_ld, int64_t *__restrict__ it_all,
                                           const float *__restrict__ E, const float *__restrict__ mask, float *__restrict__ x,
                                           int T, int N, int Ed) {
    const size_t total = (size_t)T * N * Ed;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t row = i / Ed;
        const int c = (int)(i - row * Ed), t = (int)(row / N), n = (int)(row - (size_t)t * N);
        const int64_t tok = forced[(size_t)n * forced_ld + t];
        if (it_all && c == 0) it_all[row] = tok;
        float v = fmaxf(E[(size_t)tok * Ed + c], 0.f);
        if (mask) v *= mask[i];
        x[i] = v;
    }
}

}  // namespace

extern "C" {

int capmi_version(void) { return 1; }

int64_t capmi_updown_planes_bytes(int R, int E) {
    const int64_t nR = pl_chunks(R), nE = pl_chunks(E);
    return (4 * nR + nE + (nR > nE ? nR : nE)) * PL_CHUNK;
}
int64_t capmi_updown_bwd_planes_bytes(int R) { return (2 * pl_chunks(4 * R) + 1) * PL_CHUNK; }
const char *capmi_arch(void) { return "gfx950"; }

int capmi_updown_rollout_fwd(const capmi_updown_weights *w, capmi_updown_rollout *r, void *stream) {
    if (!w || !r) return CAPMI_EINVAL;
    const int B = r->B, n = r->n, N = r->N, K = r->K, A = r->A, R = r->R, E = r->E, V1 = r->V1, T = r->T, L = r->L;
    const int B_feat = r->B_feat > 0 ? r->B_feat : B;
    if (B <= 0 || n <= 0 || N <= 0 || T <= 0 || L < T || !r->partial || B_feat < B) return CAPMI_EINVAL;
    if (!r->row_img && (N != B * n || B_feat != B)) return CAPMI_EINVAL;
    if (r->row_img && N < B * n) return CAPMI_EINVAL;
    if ((r->mode == 2 || r->teacher) && !r->forced) return CAPMI_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const size_t NR = (size_t)N * R;
    const int ld_att_ih = 2 * R + E;
    // teacher forcing knows every input token up front: the vocabulary projection + log-softmax of all T steps run as ONE
    // fat GEMM over T*N rows after the loop (20 launches of a 320-row GEMM + 20 select launches otherwise)
    // scheduled sampling: the next input may be a draw from this step's distribution, so the steps stay sequential
    const bool sched = r->teacher && r->ss_mode;
    const bool batched_logit = r->teacher && !sched && T == L && r->seq_logp &&
                               (int64_t)N * T * R <= r->partial_capacity - CAPMI_WS_COUNTER_FLOATS;

    // A planes of the step's activations (round 3, N <= 64): [h_att | h_lang | ctx | h_drop | xt | zero]; the zero image also
    // stands in for the all-zero state of step 0
    unsigned char *pl_h_att = nullptr, *pl_h_lang = nullptr, *pl_ctx = nullptr, *pl_h_drop = nullptr, *pl_xt = nullptr,
                  *pl_zero = nullptr;
    if (r->planes && N <= 64 && r->planes_bytes >= capmi_updown_planes_bytes(R, E) &&
        (reinterpret_cast<uintptr_t>(r->planes) & 15) == 0) {
        unsigned char *q = static_cast<unsigned char *>(r->planes);
        const int64_t nR = pl_chunks(R) * PL_CHUNK, nE = pl_chunks(E) * PL_CHUNK;
        pl_h_att = q; pl_h_lang = q + nR; pl_ctx = q + 2 * nR; pl_h_drop = q + 3 * nR; pl_xt = q + 4 * nR;
        pl_zero = q + 4 * nR + nE;
    }


    // r4: the h_lang / h_att segments of the NEXT step's attention-LSTM gates ride in the idle workgroups of this step's select launch
    // (capmi.h capmi_updown_rollout.pre_partial, capmi_logsoftmax_select_partial_gemm)
    const bool use_pre = pl_zero && r->pre_partial && !r->teacher &&
                         r->pre_capacity >= CAPMI_WS_COUNTER_FLOATS + (int64_t)8 * N * 4 * R;
    const int raw_flag = (r->raw_logits && !r->teacher) ? CAPMI_SELECT_RAW : 0;      // output_logsoftmax = 0 (AttModel.py:265)
    float *preA = use_pre ? r->pre_partial : nullptr;
    int preA_splits = 0;
    bool preA_valid = false;                          // slabs of the ahead part for the step about to run

    // initial state (slot 0) and flags
    RC(capmi_rollout_init(r->h_att, r->c_att, r->h_lang, r->c_lang, (int64_t)NR, r->it, r->unfinished, N, stream));   // bos = 0

    // fc term of the attention LSTM, once: fc_gates[B,4R] = fc W_ih[:, R:2R]^T
    {
        SegSpec s{r->fc, R, w->att_w_ih + R, ld_att_ih, R, 1};
        RC(gemm(stream, 0, 0, B_feat, 4 * R, r->fc_gates, 4 * R, &s, 1, r->partial, r->partial_capacity, 0, nullptr));
    }

    // Early exit (AttModel.py:349-350).  The host is hundreds of launches ahead of the device, so it cannot simply ask "is anyone
    // left?": the select kernel of step t stores 1 in alive_host[t] (pinned host memory) for every row that goes on, an event is
    // recorded behind every k-th step, and the host reads that word two steps later -- by then the device has two more steps queued
    // and never waits for the host.  The steps queued behind the decisive one see finished rows only.
    const int ee = (!r->teacher && r->early_exit > 0 && r->alive_host) ? r->early_exit : 0;
    // (events belong to the device they were created on: one set per thread AND device)
    constexpr int EE_MAX_DEV = 16;
    static thread_local hipEvent_t ee_evs[EE_MAX_DEV][4];
    static thread_local bool ee_inits[EE_MAX_DEV] = {};
    int ee_dev = 0;
    if (ee && (hipGetDevice(&ee_dev) != hipSuccess || ee_dev < 0 || ee_dev >= EE_MAX_DEV)) return CAPMI_EINVAL;
    hipEvent_t *ee_ev = ee_evs[ee_dev];
    if (ee && !ee_inits[ee_dev]) {
        for (int i = 0; i < 4; ++i)
            if (hipEventCreateWithFlags(&ee_ev[i], hipEventDisableTiming) != hipSuccess) return CAPMI_EINVAL;
        ee_inits[ee_dev] = true;
    }
    if (ee) for (int t = 0; t < L; ++t) r->alive_host[t] = 0;
    int ee_pending = -1, ee_slot = 0, steps_run = T;
    const int ee_from = r->early_exit_from > 0 ? r->early_exit_from : 0;

    // r5: teacher forcing knows every INPUT token up front too (AttModel.py:140-164): the token-embedding third of the attention-LSTM
    // gates, xt(t) . W_ih[:, 2R:]^T, of all T steps is ONE fat GEMM over T*N rows before the loop instead of a K segment of each of the
    // T per-step gate GEMMs (at N = 320 those are 192 units on 256 CUs; at N <= 64 each streamed the 16 MB weight slice again).  The
    // product lands in gates_att[t] -- the slot the cell of step t overwrites with the activated gates -- and that cell reads it as a
    // second slab set (every thread reads exactly the elements it then writes).
    static const int env_bxt = capmi::knob("CAPMI_BATCHED_XT", 1);
    const bool batched_xt = env_bxt && r->teacher && !sched && r->gates_att && r->xt &&
                            (int64_t)T * N * 4 * R < ((int64_t)1 << 31);
    if (batched_xt) {
        {
            const size_t total = (size_t)T * N * E;
            int blocks = (int)((total + 255) / 256);
            if (blocks > 8192) blocks = 8192;
            hipLaunchKernelGGL(embed_fwd_all_steps_kernel, dim3(blocks), dim3(256), 0, st, r->forced, r->forced_ld, r->it_all, w->embed,
                               r->drop_xt, r->xt, T, N, E);
            CAPMI_CHECK_LAUNCH();
        }
        SegSpec sx{r->xt, E, w->att_w_ih + 2 * R, ld_att_ih, E, 1};
        RC(gemm(stream, 0, 0, T * N, 4 * R, r->gates_att, 4 * R, &sx, 1, r->partial, r->partial_capacity, 0, nullptr));
    }

    static const bool ee_trace = capmi::research("CAPMI_EE_TRACE", 0) != 0;
    const auto ee_t0 = std::chrono::steady_clock::now();
    double ee_wait_us = 0;
    for (int t = 0; t < T; ++t) {
        if (ee_pending >= 0 && t - ee_pending >= 2) {
            const auto w0 = std::chrono::steady_clock::now();
            if (hipEventSynchronize(ee_ev[ee_slot]) != hipSuccess) return CAPMI_EINVAL;
            ee_wait_us += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - w0).count();
            if (*const_cast<volatile int32_t *>(r->alive_host + ee_pending) == 0) {    // nobody went on after that step
                steps_run = t;
                break;
            }
            ee_pending = -1;
            ee_slot = (ee_slot + 1) & 3;
        }
        float *xt = r->xt + (size_t)t * N * E;
        const float *h_att_prev = r->h_att + (size_t)t * NR, *c_att_prev = r->c_att + (size_t)t * NR;
        const float *h_lang_prev = r->h_lang + (size_t)t * NR, *c_lang_prev = r->c_lang + (size_t)t * NR;
        float *h_att = r->h_att + (size_t)(t + 1) * NR, *c_att = r->c_att + (size_t)(t + 1) * NR;
        float *h_lang = r->h_lang + (size_t)(t + 1) * NR, *c_lang = r->c_lang + (size_t)(t + 1) * NR;
        float *att_h = r->att_h + (size_t)t * N * A;
        float *alpha = r->alpha + (size_t)t * N * K;
        float *ctx = r->ctx + (size_t)t * NR;
        float *h_drop = r->h_drop + (size_t)t * NR;
        int splits = 1;

        // 1. token embedding (+ReLU +dropout).  Free-running rollouts: only step 0 launches it (BOS); afterwards the
        //    select kernel of step t-1 has already written xt (the workgroup that chose the token embeds it).
        if (batched_xt) {
        } else if (r->teacher && (!sched || t == 0))
            RC(capmi_embed_fwd_pl(r->forced + t, r->forced_ld, r->it_all ? r->it_all + (size_t)t * N : nullptr, w->embed,
                                  r->drop_xt ? r->drop_xt + (size_t)t * N * E : nullptr, xt, N, E, 1, pl_xt, stream));
        else if (t == 0)
            RC(capmi_embed_fwd_pl(r->it, 1, r->it_all ? r->it_all + (size_t)t * N : nullptr, w->embed,
                                  r->drop_xt ? r->drop_xt + (size_t)t * N * E : nullptr, xt, N, E, 1, pl_xt, stream));

        // 2-3. attention LSTM: gates = [h_lang_prev | xt | h_att_prev] . [W_ih(:, 0:R) | W_ih(:, 2R:) | W_hh]
        {
            SegSpec s[3] = {{h_lang_prev, R, w->att_w_ih, ld_att_ih, R, 1, t ? pl_h_lang : pl_zero},
                            {xt, E, w->att_w_ih + 2 * R, ld_att_ih, E, 1, pl_xt},
                            {h_att_prev, R, w->att_w_hh, R, R, 1, t ? pl_h_att : pl_zero}};
            const float *slabs2 = preA_valid ? preA + CAPMI_WS_COUNTER_FLOATS : nullptr;
            int splits2 = preA_valid ? preA_splits : 0;
            if (batched_xt) {   // the token-embedding segment of every step was multiplied before the loop (into gates_att[t])
                SegSpec s2[2] = {s[0], s[2]};
                RC(gemm(stream, 0, 0, N, 4 * R, r->partial, 4 * R, s2, 2, r->partial, r->partial_capacity, 1, &splits, nullptr,
                        nullptr, 0, pl_zero));
                slabs2 = r->gates_att + (size_t)t * N * 4 * R;
                splits2 = 1;
            } else if (preA_valid)     // the h_lang / h_att segments were computed inside the previous step's select launch
                RC(gemm(stream, 0, 0, N, 4 * R, r->partial, 4 * R, s + 1, 1, r->partial, r->partial_capacity, 1, &splits, nullptr,
                        nullptr, 0, pl_zero));
            else
                RC(gemm(stream, 0, 0, N, 4 * R, r->partial, 4 * R, s, 3, r->partial, r->partial_capacity, 1, &splits, nullptr,
                        nullptr, 0, pl_zero));
            RC(capmi_lstm_cell_fwd_pl2(r->partial + CAPMI_WS_COUNTER_FLOATS, splits, slabs2, splits2, w->att_b_ih, w->att_b_hh,
                                       r->fc_gates, n, r->row_img, c_att_prev, h_att, c_att,
                                       r->gates_att + (size_t)t * N * 4 * R, nullptr, nullptr, N, R, pl_h_att, nullptr, stream));
        }
        // 4-5. att_h = h_att W_h2att^T + b left as K-slice slabs; the fused region attention finishes the reduction
        //      (+ bias), keeps att_h for the backward pass and runs score + softmax + context
        {
            SegSpec s{h_att, R, w->h2att_w, R, R, 1, pl_h_att};
            RC(gemm(stream, 0, 0, N, A, r->partial, A, &s, 1, r->partial, r->partial_capacity, 1, &splits, nullptr, nullptr, 0,
                    pl_zero));
        }
        RC(capmi_attention_fwd_partial_pl(r->partial + CAPMI_WS_COUNTER_FLOATS, splits, (int64_t)N * A, w->h2att_b, att_h,
                                          r->p_att, r->att, r->att_mask, w->alpha_w, w->alpha_b, ctx, alpha, B_feat, n, K, A, R,
                                          r->row_img, N, pl_ctx, stream));
        // 6-7. language LSTM: gates = [ctx | h_att | h_lang_prev] . [W_ih(:, 0:R) | W_ih(:, R:2R) | W_hh]
        {
            SegSpec s[3] = {{ctx, R, w->lang_w_ih, 2 * R, R, 1, pl_ctx},
                            {h_att, R, w->lang_w_ih + R, 2 * R, R, 1, pl_h_att},
                            {h_lang_prev, R, w->lang_w_hh, R, R, 1, t ? pl_h_lang : pl_zero}};
            RC(gemm(stream, 0, 0, N, 4 * R, r->partial, 4 * R, s, 3, r->partial, r->partial_capacity, 1, &splits,
                    nullptr, nullptr, 0, pl_zero));
            RC(capmi_lstm_cell_fwd_pl2(r->partial + CAPMI_WS_COUNTER_FLOATS, splits, nullptr, 0, w->lang_b_ih, w->lang_b_hh, nullptr, 1, nullptr, c_lang_prev,
                                       h_lang, c_lang, r->gates_lang + (size_t)t * N * 4 * R,
                                       r->drop_out ? r->drop_out + (size_t)t * NR : nullptr, h_drop, N, R, pl_h_lang,
                                       batched_logit ? nullptr : pl_h_drop, stream));
        }
        if (batched_logit) continue;
        // 8-9. vocabulary projection left as K-slice slabs; log-softmax + choice + bookkeeping assemble the row
        //      (slabs + bias) in registers: no split-K reduce launch, no logits round trip
        {
            SegSpec s{h_drop, R, w->logit_w, R, R, 1, pl_h_drop};
            RC(gemm(stream, 0, 0, N, V1, r->partial, V1, &s, 1, r->partial, r->partial_capacity, 1, &splits, nullptr, nullptr, 0,
                    pl_zero));
        }
        capmi_sample_filter flt{r->top_k, r->top_p};
        capmi_next_embed ne{};
        if ((!r->teacher || sched) && t + 1 < T) {
            ne.E = w->embed; ne.Edim = E; ne.relu = 1;
            ne.mask = r->drop_xt ? r->drop_xt + (size_t)(t + 1) * N * E : nullptr;
            ne.x = r->xt + (size_t)(t + 1) * N * E;
            ne.it_save = r->it_all ? r->it_all + (size_t)(t + 1) * N : nullptr;
            ne.x_planes = pl_xt;
        }
        if (ee) ne.alive = r->alive_host + t;
        if (sched && t + 1 < T) {
            // AttModel.py:145-154: the token chosen here is the INPUT of step t+1 -- forced[:, t+1] (mode 2 rows) or a
            // categorical draw from this step's log-probs (mode 1 rows, temperature 1); it is embedded by the same launch
            RC(capmi_logsoftmax_select_partial(r->partial + CAPMI_WS_COUNTER_FLOATS, splits, (int64_t)N * V1, w->logit_b, N, V1,
                                               t, L, 2, r->ss_mode + (size_t)(t + 1) * N, 1.f,
                                               r->gumbel ? r->gumbel + (size_t)t * N * V1 : nullptr, r->seed, r->forced + 1,
                                               r->forced_ld, 1, r->seq, L, r->it, r->unfinished, r->seq_logp, r->sel_logp,
                                               r->live, &ne, nullptr, stream));
            continue;
        }
        preA_valid = false;
        if (use_pre && t + 1 < T) {
            // select of step t + [h_lang(t) | h_att(t)] . [W_ih(:, 0:R) | W_hh]^T of step t+1's attention LSTM in ONE launch
            capmi_gemm_desc ah{};
            ah.nseg = 2;
            ah.seg[0].A = h_lang; ah.seg[0].lda = R; ah.seg[0].B = w->att_w_ih; ah.seg[0].ldb = ld_att_ih; ah.seg[0].K = R; ah.seg[0].a_row_div = 1;
            ah.seg[1].A = h_att; ah.seg[1].lda = R; ah.seg[1].B = w->att_w_hh; ah.seg[1].ldb = R; ah.seg[1].K = R; ah.seg[1].a_row_div = 1;
            ah.a_planes[0] = pl_h_lang; ah.a_planes[1] = pl_h_att;
            ah.M = N; ah.N = 4 * R; ah.C = preA; ah.ldc = 4 * R;
            ah.partial = preA; ah.partial_capacity = r->pre_capacity;
            ah.splits = 6;                            // 32 column blocks x 6 K slices = 192 workgroups beside the N select rows
            ah.defer_reduce = 1;
            RC(capmi_logsoftmax_select_partial_gemm(r->partial + CAPMI_WS_COUNTER_FLOATS, splits, (int64_t)N * V1, w->logit_b, N, V1, t, L,
                                                    r->mode | raw_flag, r->row_mode, r->temperature,
                                                    r->gumbel ? r->gumbel + (size_t)t * N * V1 : nullptr, r->seed, r->forced,
                                                    r->forced_ld, 0, r->seq, L, r->it, r->unfinished, r->seq_logp, r->sel_logp,
                                                    r->live, &ne, (r->top_k > 0 || r->top_p > 0.f) ? &flt : nullptr, &ah, stream));
            preA_splits = ah.splits_used;
            preA_valid = true;
        } else
        RC(capmi_logsoftmax_select_partial(r->partial + CAPMI_WS_COUNTER_FLOATS, splits, (int64_t)N * V1, w->logit_b, N, V1, t,
                                           L, r->teacher ? 2 : (r->mode | raw_flag), r->teacher ? nullptr : r->row_mode, r->temperature,
                                           r->gumbel ? r->gumbel + (size_t)t * N * V1 : nullptr, r->seed, r->forced,
                                           r->forced_ld, r->teacher ? 1 : 0, r->seq, L, r->it, r->unfinished, r->seq_logp,
                                           r->sel_logp, r->live, &ne, (r->top_k > 0 || r->top_p > 0.f) ? &flt : nullptr, stream));
        // (no check in the last 8 steps: after its last wait the host needs a lead of several decode steps to enqueue the reward
        //  and the backward behind the rollout without the device running dry -- a check at step 15 of 20 cost 2.5 % end to end)
        if (ee && ee_pending < 0 && t >= ee_from && (t + 1 - ee_from) % ee == 0 && t + 9 <= T) {
            if (hipEventRecord(ee_ev[ee_slot], st) != hipSuccess) return CAPMI_EINVAL;
            ee_pending = t;
        }
    }
    if (ee_trace)
        fprintf(stderr, "capmi rollout: %d of %d steps enqueued in %.0f us of host time, of which %.0f us blocked on the alive flag\n",
                steps_run, T, std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - ee_t0).count(), ee_wait_us);
    r->steps_run = steps_run;
    if (steps_run < T) {     // the steps never run leave pad tokens and zero log-probs like the reference's untouched columns
        hipError_t e = hipSuccess;
        const size_t tail = (size_t)(L - steps_run);
        if (r->seq_logp) e = hipMemset2DAsync(r->seq_logp + (size_t)steps_run * V1, (size_t)L * V1 * sizeof(float), 0,
                                              tail * V1 * sizeof(float), N, st);
        if (e == hipSuccess) e = hipMemset2DAsync(r->seq + steps_run, (size_t)L * sizeof(int64_t), 0, tail * sizeof(int64_t), N, st);
        if (e == hipSuccess && r->sel_logp)
            e = hipMemset2DAsync(r->sel_logp + steps_run, (size_t)L * sizeof(float), 0, tail * sizeof(float), N, st);
        if (e == hipSuccess && r->live) e = hipMemset2DAsync(r->live + steps_run, (size_t)L, 0, tail, N, st);
        if (e != hipSuccess) return (int)e;
    }
    if (batched_logit) {
        float *hd_nt = r->partial + CAPMI_WS_COUNTER_FLOATS;          // the split-K workspace is idle here
        const size_t total = (size_t)N * T * R;
        int blocks = (int)((total + 255) / 256);
        if (blocks > 4096) blocks = 4096;
        hipLaunchKernelGGL(tn_to_nt_kernel, dim3(blocks), dim3(256), 0, st, r->h_drop, hd_nt, T, N, R);
        CAPMI_CHECK_LAUNCH();
        SegSpec s{hd_nt, R, w->logit_w, R, R, 1};
        RC(gemm(stream, 0, 0, N * T, V1, r->seq_logp, V1, &s, 1, nullptr, 0, 0, nullptr, w->logit_b));
        RC(capmi_log_softmax_rows(r->seq_logp, r->seq_logp, N * T, V1, stream));
        hipLaunchKernelGGL(teacher_bookkeep_kernel, dim3((N * L + 255) / 256), dim3(256), 0, st, r->seq_logp, r->forced,
                           r->forced_ld, r->seq, r->sel_logp, r->live, N, L, V1);
        CAPMI_CHECK_LAUNCH();
    }
    return 0;
}

int capmi_updown_rollout_bwd(const capmi_updown_weights *w, const capmi_updown_rollout *r, const float *g_seq_logp,
                             capmi_updown_bwd_scratch *s, capmi_updown_grads *g, void *stream) {
    return capmi_updown_rollout_bwd_phases(w, r, g_seq_logp, s, g, CAPMI_BWD_ALL, stream);
}

int capmi_updown_rollout_bwd_phases(const capmi_updown_weights *w, const capmi_updown_rollout *r, const float *g_seq_logp,
                                    capmi_updown_bwd_scratch *s, capmi_updown_grads *g, int phases, void *stream) {
    if (!w || !r || (!g_seq_logp && !(s && s->sparse)) || !s || !g || !(phases & CAPMI_BWD_ALL)) return CAPMI_EINVAL;
    const int B = r->B, n = r->n, Nf = r->N, K = r->K, A = r->A, R = r->R, E = r->E, V1 = r->V1, T = r->T, L = r->L;
    hipStream_t st = (hipStream_t)stream;
    // Rows [0, N) of the rollout carry gradient.  N = all Nf rows, or -- fused SCST rollout, s->n_grad_rows -- the sampled rows
    // only: the greedy-baseline rows behind them would ride through every time-batched GEMM as zeros (K or M = T*60 instead of
    // T*50).  The saved forward tensors keep their Nf-row time slabs (NRf); everything the backward produces is [T][N][..].
    const int N = (s->n_grad_rows > 0 && s->n_grad_rows < Nf) ? s->n_grad_rows : Nf;
    const bool compact = N < Nf;
    const size_t NR = (size_t)N * R, NRf = (size_t)Nf * R;
    const int TN = T * N;
    // saved activations the time-batched GEMMs read as [T*N, C] matrices: the tensors themselves, or their packed copies
    const float *a_hdrop = r->h_drop, *a_hlang = r->h_lang, *a_xt = r->xt, *a_hatt = r->h_att, *a_ctx = r->ctx,
                *a_dropxt = r->drop_xt, *a_atth = r->att_h, *a_alpha = r->alpha;
    const int64_t *a_it = r->it_all;
    if (compact) {
        float *q = s->pack;
        const int64_t need = (int64_t)N * ((int64_t)T * (4 * (int64_t)R + 2 * (int64_t)E + 2 + A + K) + R) + 64;
        if (!q || s->pack_capacity < need) return CAPMI_EINVAL;
        auto take = [&](size_t floats) { float *o = q; q += (floats + 3) & ~(size_t)3; return o; };
        float *p_hdrop = take((size_t)TN * R), *p_hlang = take((size_t)TN * R), *p_xt = take((size_t)TN * E),
              *p_hatt = take((size_t)(T + 1) * N * R), *p_ctx = take((size_t)TN * R),
              *p_dropxt = r->drop_xt ? take((size_t)TN * E) : nullptr, *p_it = take((size_t)TN * 2),
              *p_atth = take((size_t)TN * A), *p_alpha = take((size_t)TN * K);
        if (phases & CAPMI_BWD_LOGIT) {             // the first phase of a (possibly phased) backward packs for all of them
            PackArgs pa{};
            pa.Nf = Nf; pa.Nb = N;
            auto add = [&](const void *src, float *dst, int C_, int slots) {
                if (src) pa.it[pa.n++] = PackItem{static_cast<const float *>(src), dst, C_, slots};
            };
            add(r->h_drop, p_hdrop, R, T); add(r->h_lang, p_hlang, R, T); add(r->xt, p_xt, E, T);
            add(r->h_att, p_hatt, R, T + 1); add(r->ctx, p_ctx, R, T); add(r->drop_xt, p_dropxt, E, T);
            add(r->it_all, p_it, 2, T); add(r->att_h, p_atth, A, T); add(r->alpha, p_alpha, K, T);
            hipLaunchKernelGGL(pack_rows_kernel, dim3(64, pa.n), dim3(256), 0, st, pa);
            CAPMI_CHECK_LAUNCH();
        }
        a_hdrop = p_hdrop; a_hlang = p_hlang; a_xt = p_xt; a_hatt = p_hatt; a_ctx = p_ctx; a_dropxt = p_dropxt;
        a_it = r->it_all ? reinterpret_cast<const int64_t *>(p_it) : nullptr; a_atth = p_atth; a_alpha = p_alpha;
    }
    const int ld_att_ih = 2 * R + E;
    float *P = s->partial;
    const int64_t cap = s->partial_capacity;
    // bias gradients (column sums over all T*N rows): with every phase in this one call they are recorded and finished by ONE
    // launch at the end (4 launches + 3 zero-fills + 2 copies otherwise); a phased call (DDP bucket overlap) keeps each phase
    // complete when it returns
    // (r5: any call that covers two or more of the phases that produce column sums batches them -- the two-call form of the
    //  early-Adam experiment, logit | everything else, keeps one batched launch for the second call)
    const int col_phases = ((phases & CAPMI_BWD_LOGIT) ? 1 : 0) + ((phases & CAPMI_BWD_ATT_LSTM) ? 1 : 0) +
                           ((phases & CAPMI_BWD_LANG_LSTM) ? 1 : 0) + ((phases & CAPMI_BWD_ATTENTION) ? 1 : 0);
    const bool batch_cols = col_phases >= 2;
    capmi_colsum_item cols[CAPMI_COLSUM_ARGS_MAX];
    int n_cols = 0;
    // r5: weight-gradient GEMMs that nothing reads before the optimizer run on a SIDE STREAM beside the latency-bound time loop
    // (CAPMI_BWD_SIDE = k > 0, single-call backward only): dW_logit as soon as d(logits) exists, and the LSTM / h2att weight
    // gradients of the LAST (k - 1) of k time chunks as soon as the loop has walked past them; the first chunk follows on the
    // main stream behind the loop and accumulates.  OFF by default -- measured (profiles/r05_scst_overlap.md): k = 1 between -2.5 % and
    // 0 on the SCST step depending on the box, k = 2 / 3 +3 % / +14 % (a fat GEMM owns the CUs it lands on: 120-144 KB of LDS; the
    // loop's weight-streaming GEMMs then wait for whole persistent workgroups), UpDown XE 12.1 -> 15.8 ms.
    static const int env_side = capmi::knob("CAPMI_BWD_SIDE", 0);
    hipStream_t side_st = nullptr;
    hipEvent_t *side_ev = nullptr;
    bool side_used = false;
    if (env_side > 0 && (phases & CAPMI_BWD_ALL) == CAPMI_BWD_ALL && T >= 4) {
        int dev_i = 0;
        if (hipGetDevice(&dev_i) != hipSuccess || dev_i < 0 || dev_i >= 16) return CAPMI_EINVAL;
        static thread_local hipStream_t side_sts[16] = {};
        static thread_local hipEvent_t side_evs[16][2] = {};
        if (!side_sts[dev_i]) {
            if (hipStreamCreateWithFlags(&side_sts[dev_i], hipStreamNonBlocking) != hipSuccess) return CAPMI_EINVAL;
            for (int i = 0; i < 2; ++i)
                if (hipEventCreateWithFlags(&side_evs[dev_i][i], hipEventDisableTiming) != hipSuccess) return CAPMI_EINVAL;
        }
        side_st = side_sts[dev_i];
        side_ev = side_evs[dev_i];
    }
    const int side_chunks = side_st ? (env_side < 4 ? env_side : 4) : 1;       // time chunks of the LSTM weight gradients
    // LSTM / h2att weight gradients of the time steps [t0, t1): K = (t1 - t0) * N rows of the time-batched operands
    const float *x_hatt_same = a_hatt + (compact ? NR : NRf);                   // h_att of the SAME step (slots 1..T)
    // r6: in a single-call backward without the side stream the eight weight gradients nothing reads before the optimizer (dW_logit and
    // the LSTM / h2att ones) are only LISTED where they become computable and go out together at the end as ONE grouped persistent
    // launch (capmi_gemm_group_tn: 256 x 128 tiles, whole-K tiles of the full rounds straight into the gradient buffers, a K-sliced
    // tail) instead of eight sub-wave grids (profiles/r05_scst_kernel_stats.md: 10 launches of gemm_x3_kernel<false,false>, 446 us).
    // CAPMI_GEMM_GROUP=0 restores them.
    static const int env_group = capmi::knob("CAPMI_GEMM_GROUP", 1);
    capmi_group_gemm grp[9];
    int n_grp = 0;
    const bool grouped = env_group && (phases & CAPMI_BWD_ALL) == CAPMI_BWD_ALL && !side_st;
    auto dw_one = [&](void *strm, int Mw, int Nw, float *Cw, int ldcw, const float *Aw, int ldaw, const float *Bw, int ldbw, int Kw, int acc,
                      float *Pw, int64_t capw) -> int {
        if (grouped && !acc && n_grp < 9) {
            grp[n_grp++] = capmi_group_gemm{Aw, Bw, Cw, ldaw, ldbw, ldcw, Kw, Mw, Nw, 0, 0, nullptr};
            return 0;
        }
        SegSpec a{Aw, ldaw, Bw, ldbw, Kw, 1};
        return gemm(strm, 1, 1, Mw, Nw, Cw, ldcw, &a, 1, Pw, capw, 0, nullptr, nullptr, nullptr, acc);
    };
    auto dw_chunk = [&](void *strm, int t0, int t1, int acc, float *Pw, int64_t capw) -> int {
        const int Kc = (t1 - t0) * N;
        const size_t o4 = (size_t)t0 * N * 4 * R, oR = (size_t)t0 * N * R, oE = (size_t)t0 * N * E, oA = (size_t)t0 * N * A;
        if (phases & CAPMI_BWD_ATT_LSTM) {
            RC(dw_one(strm, 4 * R, R, g->att_w_ih, ld_att_ih, s->dg_att + o4, 4 * R, a_hlang + oR, R, Kc, acc, Pw, capw));          // x h_lang_prev (slots 0..T-1)
            RC(dw_one(strm, 4 * R, E, g->att_w_ih + 2 * R, ld_att_ih, s->dg_att + o4, 4 * R, a_xt + oE, E, Kc, acc, Pw, capw));      // x xt
            RC(dw_one(strm, 4 * R, R, g->att_w_hh, R, s->dg_att + o4, 4 * R, a_hatt + oR, R, Kc, acc, Pw, capw));                    // x h_att_prev
        }
        if (phases & CAPMI_BWD_LANG_LSTM) {
            RC(dw_one(strm, 4 * R, R, g->lang_w_ih, 2 * R, s->dg_lang + o4, 4 * R, a_ctx + oR, R, Kc, acc, Pw, capw));
            RC(dw_one(strm, 4 * R, R, g->lang_w_ih + R, 2 * R, s->dg_lang + o4, 4 * R, x_hatt_same + oR, R, Kc, acc, Pw, capw));
            RC(dw_one(strm, 4 * R, R, g->lang_w_hh, R, s->dg_lang + o4, 4 * R, a_hlang + oR, R, Kc, acc, Pw, capw));
        }
        if (phases & CAPMI_BWD_ATTENTION)
            RC(dw_one(strm, A, R, g->h2att_w, R, s->d_att_h_all + oA, A, x_hatt_same + oR, R, Kc, acc, Pw, capw));
        return 0;
    };
    int dw_done_from = T;           // time steps [dw_done_from, T) already have their weight gradients (on the side stream)
    auto chunk_boundary = [&](int t) {              // t = c * T / side_chunks for some c in [1, side_chunks)
        for (int c = 1; c < side_chunks; ++c)
            if ((c * T) / side_chunks == t) return true;
        return false;
    };
    // (r6) a column sum whose input is the A operand of a listed weight-gradient GEMM rides in that GEMM's staging waves
    // (capmi_group_gemm.colsum); its second copy (nn.LSTMCell's bias_hh gradient) is a 16-KB copy behind the grouped launch
    struct Copy2 { float *dst; const float *src; size_t bytes; } copies[4];
    int n_copies = 0;
    auto colsum = [&](const float *in, int rows, int ncol, float *out, float *out2) -> int {
        for (int i = 0; i < n_grp; ++i)
            if (grp[i].A == in && grp[i].K == rows && grp[i].M == ncol && grp[i].lda == ncol && !grp[i].colsum && n_copies < 4 &&
                (reinterpret_cast<uintptr_t>(out) & 15) == 0) {
                grp[i].colsum = out;
                if (out2) copies[n_copies++] = Copy2{out2, out, (size_t)ncol * sizeof(float)};
                return 0;
            }
        if (batch_cols) {
            cols[n_cols++] = capmi_colsum_item{in, out, out2, rows, ncol, ncol, 0};
            return 0;
        }
        int rc = capmi_colsum(in, rows, ncol, ncol, out, 0, stream);
        if (rc) return rc;
        if (out2) {
            hipError_t e = hipMemcpyAsync(out2, out, (size_t)ncol * sizeof(float), hipMemcpyDeviceToDevice, (hipStream_t)stream);
            if (e != hipSuccess) return (int)e;
        }
        return 0;
    };

    // ---- logit layer, batched over all T*N rows ----------------------------------------------
    if (phases & CAPMI_BWD_LOGIT) {
        if (r->raw_logits && !r->teacher) {
            // the rollout returned logits: d(logits) is the loss gradient (sparse and / or dense parts), no softmax Jacobian
            capmi_sparse_logp_grad sp = s->sparse ? *s->sparse : capmi_sparse_logp_grad{};
            sp.raw = 1;
            RC(capmi_logsoftmax_bwd_sparse(&sp, g_seq_logp, r->seq_logp, r->live, s->dlogits, N, L, T, V1, stream));
        } else if (s->sparse) RC(capmi_logsoftmax_bwd_sparse(s->sparse, g_seq_logp, r->seq_logp, r->live, s->dlogits, N, L, T, V1, stream));
        else RC(capmi_logsoftmax_bwd(g_seq_logp, r->seq_logp, r->live, s->dlogits, N, L, T, V1, stream));
        SegSpec a{s->dlogits, V1, w->logit_w, R, V1, 1};   // d_hdrop = dlogits W_logit          [TN,R]
        RC(gemm(stream, 0, 1, TN, R, s->d_hdrop, R, &a, 1, P, cap, 0, nullptr));
        SegSpec b{s->dlogits, V1, a_hdrop, R, TN, 1};       // dW_logit = dlogits^T h_drop         [V1,R]
        // r5 (profiles/r05_scst_overlap.md): nothing reads dW_logit before the optimizer: it runs on the side stream beside the
        // time loop (no split-K workspace there: the loop owns P)
        if (side_st) {
            if (hipEventRecord(side_ev[0], st) != hipSuccess) return CAPMI_EINVAL;          // dlogits, h_drop (packed) are ready
            if (hipStreamWaitEvent(side_st, side_ev[0], 0) != hipSuccess) return CAPMI_EINVAL;
            RC(gemm((void *)side_st, 1, 1, V1, R, g->logit_w, R, &b, 1, nullptr, 0, 0, nullptr));
            side_used = true;
        } else if (grouped) grp[n_grp++] = capmi_group_gemm{s->dlogits, a_hdrop, g->logit_w, V1, R, R, TN, V1, R, 0, 0, nullptr};
        else RC(gemm(stream, 1, 1, V1, R, g->logit_w, R, &b, 1, P, cap, 0, nullptr));
        RC(colsum(s->dlogits, TN, V1, g->logit_b, nullptr));
    }

    // ---- pack the recurrent weight slices once: [W_ih | W_hh] side by side, so that each step needs ONE
    //      dX GEMM per LSTM instead of two (+ their split-K reductions): 80 MB of copies per BPTT buys back
    //      ~80 launches.
    if (phases & CAPMI_BWD_RECURRENT) {
        const bool al = R % 4 == 0 && ld_att_ih % 4 == 0 &&
                        ((reinterpret_cast<uintptr_t>(w->lang_w_ih) | reinterpret_cast<uintptr_t>(w->lang_w_hh) |
                          reinterpret_cast<uintptr_t>(w->att_w_ih) | reinterpret_cast<uintptr_t>(w->att_w_hh) |
                          reinterpret_cast<uintptr_t>(s->w_lang_cat) | reinterpret_cast<uintptr_t>(s->w_att_cat)) & 15) == 0;
        if (al) {
            const int total = 4 * R * 5 * R;
            hipLaunchKernelGGL(pack_recurrent_kernel, dim3((total / 4 + 255) / 256), dim3(256), 0, st, w->lang_w_ih, w->lang_w_hh,
                               w->att_w_ih, w->att_w_hh, s->w_lang_cat, s->w_att_cat, R, ld_att_ih);
            CAPMI_CHECK_LAUNCH();
        } else {
            hipError_t e;
            const size_t fb = sizeof(float);
            if ((e = hipMemcpy2DAsync(s->w_lang_cat, 3 * R * fb, w->lang_w_ih, 2 * R * fb, 2 * R * fb, 4 * R,
                                      hipMemcpyDeviceToDevice, st)) != hipSuccess) return (int)e;
            if ((e = hipMemcpy2DAsync(s->w_lang_cat + 2 * R, 3 * R * fb, w->lang_w_hh, R * fb, R * fb, 4 * R,
                                      hipMemcpyDeviceToDevice, st)) != hipSuccess) return (int)e;
            if ((e = hipMemcpy2DAsync(s->w_att_cat, 2 * R * fb, w->att_w_ih, (size_t)ld_att_ih * fb, R * fb, 4 * R,
                                      hipMemcpyDeviceToDevice, st)) != hipSuccess) return (int)e;
            if ((e = hipMemcpy2DAsync(s->w_att_cat + R, 2 * R * fb, w->att_w_hh, R * fb, R * fb, 4 * R,
                                      hipMemcpyDeviceToDevice, st)) != hipSuccess) return (int)e;
        }
    }

    // ---- BPTT over the recurrent part ----------------------------------------------------------
    // Workspace carved in three: the d_x1 and dh_att(attention) GEMMs leave their K-slice slabs in regions of their own
    // and the LSTM-cell kernels of the following launches finish those reductions (2 of the 3 split-K reduce launches
    // per step disappear; d_x1 is never materialised).  d_x2 (read by three kernels and the batched pass) is finished and
    // published by the attention Jacobian kernel, its first consumer.
    const int64_t cap1 = (cap / 4) & ~(int64_t)1023, caph = (cap / 8) & ~(int64_t)1023, capm = cap - cap1 - caph;
    float *P1 = P + capm, *Ph = P + capm + cap1;
    if (phases & CAPMI_BWD_RECURRENT) {
    if (capm <= CAPMI_WS_COUNTER_FLOATS || cap1 <= CAPMI_WS_COUNTER_FLOATS || caph <= CAPMI_WS_COUNTER_FLOATS) return CAPMI_EINVAL;
    static const bool self_reduce = capmi::research("CAPMI_GEMM_SELF_REDUCE", 0) != 0;
    if (self_reduce) {   // ticket words of the carved regions start zeroed like the main one (only the in-launch reduction reads them)
        hipError_t e = hipMemsetAsync(P1, 0, CAPMI_WS_COUNTER_FLOATS * sizeof(float), st);
        if (e == hipSuccess) e = hipMemsetAsync(Ph, 0, CAPMI_WS_COUNTER_FLOATS * sizeof(float), st);
        if (e != hipSuccess) return (int)e;
    }
    // A planes of d_gates (round 3, <= 64 gradient rows): [dg_lang | dg_att | one zero image]
    unsigned char *pl_dg_lang = nullptr, *pl_dg_att = nullptr, *pl_zero = nullptr;
    if (s->planes && N <= 64 && s->planes_bytes >= capmi_updown_bwd_planes_bytes(R) &&
        (reinterpret_cast<uintptr_t>(s->planes) & 15) == 0) {
        unsigned char *q = static_cast<unsigned char *>(s->planes);
        const int64_t n4 = pl_chunks(4 * R) * PL_CHUNK;
        pl_dg_lang = q; pl_dg_att = q + n4; pl_zero = q + 2 * n4;
    }
    const float *x1_slabs = P1 + CAPMI_WS_COUNTER_FLOATS, *h_slabs = Ph + CAPMI_WS_COUNTER_FLOATS;
    const int64_t x1_stride = (int64_t)N * 2 * R, h_stride = (int64_t)NR;
    int x1_splits = 1, h_splits = 1;
    for (int t = T - 1; t >= 0; --t) {
        const bool last = (t == T - 1);
        float *d_x2 = s->d_x2 + (size_t)t * N * 3 * R;
        const float *d_x2_next = last ? nullptr : s->d_x2 + (size_t)(t + 1) * N * 3 * R;
        float *dg_lang = s->dg_lang + (size_t)t * N * 4 * R;
        float *dg_att = s->dg_att + (size_t)t * N * 4 * R;
        float *dc_lang_in = s->dc_lang + (size_t)((t + 1) & 1) * NR, *dc_lang_out = s->dc_lang + (size_t)(t & 1) * NR;
        float *dc_att_in = s->dc_att + (size_t)((t + 1) & 1) * NR, *dc_att_out = s->dc_att + (size_t)(t & 1) * NR;

        // language LSTM cell: dh = d_hdrop*mask + dh_lang(att-LSTM input of step t+1: d_x1 slabs, columns 0..R)
        //                          + dh_lang(own W_hh, t+1)
        RC(capmi_lstm_cell_bwd_partial_pl(s->d_hdrop + (size_t)t * NR, R, r->drop_out ? r->drop_out + (size_t)t * NRf : nullptr,
                                          last ? nullptr : x1_slabs, 2 * R, x1_splits, x1_stride,
                                          d_x2_next ? d_x2_next + 2 * R : nullptr, 3 * R, 1, 0, last ? nullptr : dc_lang_in,
                                          r->gates_lang + (size_t)t * Nf * 4 * R, r->c_lang + (size_t)t * NRf,
                                          r->c_lang + (size_t)(t + 1) * NRf, dg_lang, dc_lang_out, N, R, pl_dg_lang, stream));
        // d_x2 = dg_lang [W_ih | W_hh]  -> (d_ctx | dh_att | dh_lang_prev), left as slabs; the attention Jacobian
        // (d_ctx -> d_att_h, and d_e kept for the batched pass) finishes the reduction of its rows and publishes d_x2
        int x2_splits = 1;
        {
            SegSpec a{dg_lang, 4 * R, s->w_lang_cat, 3 * R, 4 * R, 1, pl_dg_lang};
            RC(gemm(stream, 0, 1, N, 3 * R, P, 3 * R, &a, 1, P, capm, 1, &x2_splits, nullptr, nullptr, 0, pl_zero));
        }
        RC(capmi_attention_bwd_partial(P + CAPMI_WS_COUNTER_FLOATS, x2_splits, (int64_t)N * 3 * R, 3 * R, d_x2,
                                       r->att_h + (size_t)t * Nf * A, r->alpha + (size_t)t * Nf * K, r->p_att, r->att,
                                       w->alpha_w, s->d_att_h_all + (size_t)t * N * A, s->d_e_all + (size_t)t * N * K,
                                       r->B_feat > 0 ? r->B_feat : B, n, K, A, R, r->row_img, N, stream));
        {
            SegSpec a{s->d_att_h_all + (size_t)t * N * A, A, w->h2att_w, R, A, 1};   // dh_att via h2att, left as slabs
            RC(gemm(stream, 0, 1, N, R, Ph, R, &a, 1, Ph, caph, 1, &h_splits));
        }
        // attention LSTM cell: dh = dh_att(lang input) + dh_att(attention: slabs) + dh_att(own W_hh, t+1: d_x1 slabs,
        // columns R..2R)
        RC(capmi_lstm_cell_bwd_partial_pl(d_x2 + R, 3 * R, nullptr, h_slabs, R, h_splits, h_stride,
                                          last ? nullptr : x1_slabs + R, 2 * R, x1_splits, x1_stride,
                                          last ? nullptr : dc_att_in, r->gates_att + (size_t)t * Nf * 4 * R,
                                          r->c_att + (size_t)t * NRf, r->c_att + (size_t)(t + 1) * NRf, dg_att, dc_att_out, N, R,
                                          t > 0 ? pl_dg_att : nullptr, stream));
        // d_x1 = dg_att [W_ih(:, 0:R) | W_hh] -> (dh_lang_prev | dh_att_prev) as slabs for step t-1; not needed at t = 0
        if (t > 0) {
            SegSpec a{dg_att, 4 * R, s->w_att_cat, 2 * R, 4 * R, 1, pl_dg_att};
            RC(gemm(stream, 0, 1, N, 2 * R, P1, 2 * R, &a, 1, P1, cap1, 1, &x1_splits, nullptr, nullptr, 0, pl_zero));
        }
        // chunk boundaries at t = c * T / side_chunks, c = side_chunks - 1 .. 1: steps [t, dw_done_from) are complete
        if (side_chunks > 1 && t > 0 && t < dw_done_from && chunk_boundary(t)) {
            if (hipEventRecord(side_ev[0], st) != hipSuccess) return CAPMI_EINVAL;
            if (hipStreamWaitEvent(side_st, side_ev[0], 0) != hipSuccess) return CAPMI_EINVAL;
            RC(dw_chunk((void *)side_st, t, dw_done_from, dw_done_from < T ? 1 : 0, nullptr, 0));
            dw_done_from = t;
            side_used = true;
        }
    }

    }   // CAPMI_BWD_RECURRENT

    // ---- time-batched parameter / feature gradients --------------------------------------------
    if (side_used) {                                   // the side stream's GEMMs wrote (parts of) the gradients the chunk below adds to
        if (hipEventRecord(side_ev[1], side_st) != hipSuccess) return CAPMI_EINVAL;
        if (hipStreamWaitEvent(st, side_ev[1], 0) != hipSuccess) return CAPMI_EINVAL;
    }
    RC(dw_chunk(stream, 0, dw_done_from, dw_done_from < T ? 1 : 0, P, cap));
    // attention LSTM
    if (phases & CAPMI_BWD_ATT_LSTM) {
        RC(colsum(s->dg_att, TN, 4 * R, g->att_b_ih, g->att_b_hh));
        hipError_t e;
        // fc columns: sum over time and over the n rows of an image first
        RC(capmi_group_rowsum(s->dg_att, T, (int64_t)N * 4 * R, B, n, 4 * R, s->sum_dg_att, stream));
        // (K = B rows: listed for the grouped launch like the time-batched ones -- any K since r6)
        RC(dw_one(stream, 4 * R, R, g->att_w_ih + R, ld_att_ih, s->sum_dg_att, 4 * R, r->fc, R, B, 0, P, cap));
        if (g->d_fc) {
            SegSpec f{s->sum_dg_att, 4 * R, w->att_w_ih + R, ld_att_ih, 4 * R, 1};
            RC(gemm(stream, 0, 1, B, R, g->d_fc, R, &f, 1, P, cap, 0, nullptr));
        }
        // token embedding: d_xt = dg_att W_ih(:, 2R:) then scatter through ReLU/dropout
        SegSpec x{s->dg_att, 4 * R, w->att_w_ih + 2 * R, ld_att_ih, 4 * R, 1};
        RC(gemm(stream, 0, 1, TN, E, s->d_xt_all, E, &x, 1, P, cap, 0, nullptr));
        e = hipMemsetAsync(g->embed, 0, (size_t)V1 * E * sizeof(float), st);
        if (e != hipSuccess) return (int)e;
        RC(capmi_embed_bwd(a_it, s->d_xt_all, a_xt, a_dropxt, g->embed, TN, E, 1, stream));
    }
    // language LSTM
    if (phases & CAPMI_BWD_LANG_LSTM) RC(colsum(s->dg_lang, TN, 4 * R, g->lang_b_ih, g->lang_b_hh));
    // attention parameters / features
    if (phases & CAPMI_BWD_ATTENTION) {
        RC(colsum(s->d_att_h_all, TN, A, g->h2att_b, nullptr));
        // alpha_net's weight gradient: one partial row per (image, region) in the workspace (free again: the GEMMs above have
        // finished their own slabs), summed by the batched column-sum launch below instead of 184 000 atomicAdds on 512 words
        float *dw_part = (batch_cols && n_cols < CAPMI_COLSUM_ARGS_MAX && cap >= CAPMI_WS_COUNTER_FLOATS + (int64_t)B * K * A)
                             ? P + CAPMI_WS_COUNTER_FLOATS : nullptr;
        RC(capmi_attention_bwd_batched_ws(s->d_x2, 3 * R, a_atth, a_alpha, s->d_e_all, r->p_att, w->alpha_w, g->d_att,
                                          g->d_p_att, g->alpha_w, g->alpha_b, T, B, n, N, K, A, R, dw_part, stream));
        if (dw_part) RC(colsum(dw_part, B * K, A, g->alpha_w, nullptr));
    }
    if (n_grp) {
        // the grouped weight gradients (+ the bias gradients folded into them) LAST: every operand is final, and the K-slice pieces go
        // behind alpha_net's partial rows, which the batched column sum below still has to read
        const int64_t skip = CAPMI_WS_COUNTER_FLOATS + (((int64_t)B * K * A + 1023) & ~(int64_t)1023);
        RC(capmi_gemm_group_tn(grp, n_grp, cap > skip ? P + skip : nullptr, cap > skip ? cap - skip : 0, stream));
        for (int i = 0; i < n_copies; ++i) {
            hipError_t e = hipMemcpyAsync(copies[i].dst, copies[i].src, copies[i].bytes, hipMemcpyDeviceToDevice, st);
            if (e != hipSuccess) return (int)e;
        }
    }
    if (n_cols) RC(capmi_colsum_batch_args(cols, n_cols, stream));
    return 0;
}

}  // extern "C"
