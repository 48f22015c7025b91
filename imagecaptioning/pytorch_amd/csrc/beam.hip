// Device-side batched beam search for the UpDown decoder (gfx950).
// Reference: AttModel._sample_beam (AttModel.py:218-256) + CaptionModel.beam_search
// (CaptionModel.py:35-209, group_size 1).  See include/capmi.h for what each kernel replaces.
#include "capmi_common.h"
#include "../../../include/capmi.h"

using namespace capmi;

namespace {

#define RC(x)                 \
    do {                      \
        int rc__ = (x);       \
        if (rc__) return rc__;\
    } while (0)

constexpr int SEL_T = 1024;
constexpr int BD_MAX = 16;

struct Cand {
    float v;
    int i;
};
__device__ __forceinline__ Cand best_of(Cand a, Cand b) {
    if (b.v > a.v || (b.v == a.v && b.i < a.i)) return b;
    return a;
}

// one workgroup per image.  ONE pass over the cur*V1 candidates (~46 per thread at beam 5, V1 = 9488): every thread keeps
// the best BDP >= bd of its own candidates in a sorted register list (compare-exchange chain, fully unrolled: no dynamic
// register indexing), then bd rounds of a block-wide arg-max over the list HEADS pick the winners in order; the thread
// that owned a winner pops it.  The previous version re-read all candidates from L2 in each of the bd rounds, with an
// integer division and a taken-list scan per candidate: 96 us per step at beam 5, a quarter of a beam-5 decode.
// Order: larger score first, equal scores -> smaller flat index (parent * V1 + token), as torch.sort(stable) gives.
template <int BDP>
__global__ __launch_bounds__(SEL_T) void beam_select_kernel(const float *__restrict__ logp, const float *__restrict__ sums,
                                                           int cur, int bd, int V1, int force_end,
                                                           int *__restrict__ parent, int64_t *__restrict__ token,
                                                           float *__restrict__ score, float *__restrict__ next_sums,
                                                           uint8_t *__restrict__ ended) {
    __shared__ float s_v[32];
    __shared__ int s_i[32];
    __shared__ int s_win;
    const int b = blockIdx.x;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const float *lp = logp + (size_t)b * cur * V1;
    Cand top[BDP];
#pragma unroll
    for (int q = 0; q < BDP; ++q) top[q] = Cand{-INFINITY, 0x7fffffff};
    for (int r = 0; r < cur; ++r) {
        const float base = sums[(size_t)b * bd + r];
        const float *row = lp + (size_t)r * V1;
        for (int v = threadIdx.x; v < V1; v += blockDim.x) {
            Cand c{base + row[v], r * V1 + v};
#pragma unroll
            for (int q = 0; q < BDP; ++q) {          // insertion: c sinks to its place, pushing worse entries down
                const Cand hi = best_of(top[q], c);
                c = (hi.i == top[q].i) ? c : top[q];
                top[q] = hi;
            }
        }
    }
    for (int round = 0; round < bd; ++round) {
        Cand best = top[0];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            Cand y{__shfl_xor(best.v, o, 64), __shfl_xor(best.i, o, 64)};
            best = best_of(best, y);
        }
        if (lane == 0) {
            s_v[wid] = best.v;
            s_i[wid] = best.i;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            Cand t{s_v[0], s_i[0]};
            for (int i = 1; i < nw; ++i) t = best_of(t, Cand{s_v[i], s_i[i]});
            s_win = t.i;
            const int par = t.i / V1, tok = t.i % V1;
            const size_t o = (size_t)b * bd + round;
            parent[o] = par;
            token[o] = tok;
            score[o] = t.v;
            const bool end = (tok == 0) || force_end;
            ended[o] = end ? 1 : 0;
            next_sums[o] = end ? t.v - 1000.f : t.v;      // CaptionModel.py:198
        }
        __syncthreads();
        if (top[0].i == s_win) {                          // flat indices are unique: exactly one thread pops
#pragma unroll
            for (int q = 0; q + 1 < BDP; ++q) top[q] = top[q + 1];
            top[BDP - 1] = Cand{-INFINITY, 0x7fffffff};
        }
        __syncthreads();                                  // s_win / s_v reused next round
    }
}

__global__ void beam_reorder_kernel(const float *__restrict__ src, float *__restrict__ dst, const int *__restrict__ parent,
                                    int arrays, int B, int cur, int bd, int R) {
    const size_t per = (size_t)B * bd * R;
    const size_t total = per * arrays;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int a = (int)(i / per);
        const size_t rem = i % per;
        const int row = (int)(rem / R), c = (int)(rem % R);
        const int b = row / bd;
        const int srow = b * cur + parent[row];
        // source arrays hold B*bd rows each (only the first B*cur are meaningful at step 0)
        dst[(size_t)a * per + rem] = src[(size_t)a * per + (size_t)srow * R + c];
    }
}

__global__ __launch_bounds__(SEL_T) void beam_logsoftmax_kernel(const float *__restrict__ logits, float *__restrict__ out,
                                                               int V1, float inv_t, int unk_col) {
    __shared__ float s_f[32];
    const size_t r = blockIdx.x;
    const float *x = logits + r * V1;
    float *o = out + r * V1;
    float m = -INFINITY;
    for (int v = threadIdx.x; v < V1; v += blockDim.x) m = fmaxf(m, x[v]);
    m = block_max(m, s_f);
    float s = 0.f;
    for (int v = threadIdx.x; v < V1; v += blockDim.x) s += __expf(x[v] - m);
    s = block_sum(s, s_f);
    const float lse = m + __logf(s);
    // second pass over (x - lse) * inv_t
    const float m2 = (m - lse) * inv_t;       // inv_t > 0: the max stays the max
    float s2 = 0.f;
    for (int v = threadIdx.x; v < V1; v += blockDim.x) s2 += __expf((x[v] - lse) * inv_t - m2);
    s2 = block_sum(s2, s_f);
    const float lse2 = m2 + __logf(s2);
    for (int v = threadIdx.x; v < V1; v += blockDim.x) {
        float y = (x[v] - lse) * inv_t - lse2;
        if (v == unk_col) y -= 1000.f;
        o[v] = y;
    }
}

struct SegSpec {
    const float *A; int lda; const float *B; int ldb; int K; int a_row_div;
};
int gemm(void *stream, int M, int N, float *C, int ldc, const SegSpec *segs, int nseg, float *partial, int64_t cap,
         int defer, int *splits_used, const float *bias = nullptr) {
    capmi_gemm_desc d{};
    d.nseg = nseg;
    for (int i = 0; i < nseg; ++i) {
        d.seg[i].A = segs[i].A; d.seg[i].lda = segs[i].lda; d.seg[i].B = segs[i].B; d.seg[i].ldb = segs[i].ldb;
        d.seg[i].K = segs[i].K; d.seg[i].a_row_div = segs[i].a_row_div > 0 ? segs[i].a_row_div : 1;
    }
    d.M = M; d.N = N; d.C = C; d.ldc = ldc; d.bias = bias;
    d.partial = partial; d.partial_capacity = cap; d.splits = 0; d.defer_reduce = defer;
    const int rc = capmi_gemm_f32(&d, stream);
    if (splits_used) *splits_used = d.splits_used;
    return rc;
}

// one decoder step on `rows` rows, `n` rows per image (get_logprobs_state, AttModel.py:166-176)
int decode_step(const capmi_updown_weights *w, capmi_updown_beam *b, int rows, int n, const float *st_in, float *st_out,
                float *logp_out, float temperature, void *stream) {
    const int B = b->B, K = b->K, A = b->A, R = b->R, E = b->E, V1 = b->V1;
    const size_t per = (size_t)B * b->bd * R;
    const float *h_att_p = st_in, *c_att_p = st_in + per, *h_lang_p = st_in + 2 * per, *c_lang_p = st_in + 3 * per;
    float *h_att = st_out, *c_att = st_out + per, *h_lang = st_out + 2 * per, *c_lang = st_out + 3 * per;
    const int ld_att_ih = 2 * R + E;
    float *slabs = b->partial + CAPMI_WS_COUNTER_FLOATS;
    int splits = 1;
    RC(capmi_embed_fwd(b->it, 1, nullptr, w->embed, nullptr, b->xt, rows, E, 1, stream));
    {
        SegSpec s[3] = {{h_lang_p, R, w->att_w_ih, ld_att_ih, R, 1}, {b->xt, E, w->att_w_ih + 2 * R, ld_att_ih, E, 1},
                        {h_att_p, R, w->att_w_hh, R, R, 1}};
        RC(gemm(stream, rows, 4 * R, b->partial, 4 * R, s, 3, b->partial, b->partial_capacity, 1, &splits));
        RC(capmi_lstm_cell_fwd(slabs, splits, w->att_b_ih, w->att_b_hh, b->fc_gates, n, nullptr, c_att_p, h_att, c_att,
                               b->gates, nullptr, nullptr, rows, R, stream));
    }
    {
        SegSpec s{h_att, R, w->h2att_w, R, R, 1};
        RC(gemm(stream, rows, A, b->att_h, A, &s, 1, b->partial, b->partial_capacity, 0, nullptr, w->h2att_b));
    }
    RC(capmi_attention_fwd(b->att_h, b->p_att, b->att, b->att_mask, w->alpha_w, w->alpha_b, b->ctx, b->alpha, B, n, K, A, R,
                           nullptr, rows, stream));
    {
        SegSpec s[3] = {{b->ctx, R, w->lang_w_ih, 2 * R, R, 1}, {h_att, R, w->lang_w_ih + R, 2 * R, R, 1},
                        {h_lang_p, R, w->lang_w_hh, R, R, 1}};
        RC(gemm(stream, rows, 4 * R, b->partial, 4 * R, s, 3, b->partial, b->partial_capacity, 1, &splits));
        RC(capmi_lstm_cell_fwd(slabs, splits, w->lang_b_ih, w->lang_b_hh, nullptr, 1, nullptr, c_lang_p, h_lang, c_lang,
                               b->gates, nullptr, nullptr, rows, R, stream));
    }
    {
        SegSpec s{h_lang, R, w->logit_w, R, R, 1};     // eval mode: no dropout on the output
        RC(gemm(stream, rows, V1, b->logits, V1, &s, 1, b->partial, b->partial_capacity, 0, nullptr, w->logit_b));
    }
    if (!logp_out) return 0;      // raw logits wanted (capmi_updown_decode_step)
    return capmi_beam_logsoftmax(b->logits, logp_out, rows, V1, temperature, b->unk_col, stream);
}

}  // namespace

extern "C" {

int capmi_beam_select(const float *logp, const float *sums, int B, int cur, int bd, int V1, int force_end,
                      int32_t *parent, int64_t *token, float *score, float *next_sums, uint8_t *ended, void *stream) {
    if (!logp || !sums || !parent || !token || !score || !next_sums || !ended) return CAPMI_EINVAL;
    if (B <= 0 || cur <= 0 || cur > bd || bd > BD_MAX || V1 <= 0 || (long long)cur * V1 < bd) return CAPMI_EINVAL;
#define CAPMI_BSEL(P_)                                                                                                 \
    hipLaunchKernelGGL(beam_select_kernel<P_>, dim3(B), dim3(SEL_T), 0, (hipStream_t)stream, logp, sums, cur, bd, V1,  \
                       force_end, parent, token, score, next_sums, ended)
    if (bd <= 2) CAPMI_BSEL(2);
    else if (bd <= 4) CAPMI_BSEL(4);
    else if (bd <= 8) CAPMI_BSEL(8);
    else CAPMI_BSEL(16);
#undef CAPMI_BSEL
    CAPMI_CHECK_LAUNCH();
    return 0;
}

int capmi_beam_reorder(const float *src, float *dst, const int32_t *parent, int arrays, int B, int cur, int bd, int R,
                       void *stream) {
    if (!src || !dst || !parent || arrays <= 0 || B <= 0 || cur <= 0 || bd <= 0 || R <= 0) return CAPMI_EINVAL;
    const size_t total = (size_t)arrays * B * bd * R;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(beam_reorder_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, src, dst, parent, arrays, B, cur,
                       bd, R);
    CAPMI_CHECK_LAUNCH();
    return 0;
}

int capmi_beam_logsoftmax(const float *logits, float *out, int N, int V1, float temperature, int unk_col, void *stream) {
    if (!logits || !out || N <= 0 || V1 <= 0 || !(temperature > 0.f)) return CAPMI_EINVAL;
    hipLaunchKernelGGL(beam_logsoftmax_kernel, dim3(N), dim3(SEL_T), 0, (hipStream_t)stream, logits, out, V1,
                       1.f / temperature, unk_col);
    CAPMI_CHECK_LAUNCH();
    return 0;
}

int capmi_updown_decode_step(const capmi_updown_weights *w, capmi_updown_beam *b, int rows, int rows_per_image,
                             const float *state_in, float *state_out, int first, void *stream) {
    if (!w || !b || b->B <= 0 || b->bd <= 0 || rows_per_image <= 0 || rows_per_image > b->bd || rows != b->B * rows_per_image ||
        !state_in || !state_out || state_in == state_out || !b->partial || !b->logits || !b->it)
        return CAPMI_EINVAL;
    if (first) {   // fc term of the attention LSTM, once per set of images
        SegSpec s{b->fc, b->R, w->att_w_ih + b->R, 2 * b->R + b->E, b->R, 1};
        RC(gemm(stream, b->B, 4 * b->R, b->fc_gates, 4 * b->R, &s, 1, b->partial, b->partial_capacity, 0, nullptr));
    }
    return decode_step(w, b, rows, rows_per_image, state_in, state_out, nullptr, 1.f, stream);
}

int capmi_updown_beam_search(const capmi_updown_weights *w, capmi_updown_beam *b, void *stream) {
    if (!w || !b || b->B <= 0 || b->bd <= 0 || b->bd > BD_MAX || b->L <= 0 || !b->partial) return CAPMI_EINVAL;
    const int B = b->B, bd = b->bd, R = b->R, E = b->E, V1 = b->V1, L = b->L;
    const int N = B * bd;
    hipStream_t st = (hipStream_t)stream;
    const size_t per = (size_t)N * R, st_sz = 4 * per;
    hipError_t e;
    if ((e = hipMemsetAsync(b->state, 0, 2 * st_sz * sizeof(float), st)) != hipSuccess) return (int)e;
    if ((e = hipMemsetAsync(b->it, 0, (size_t)N * sizeof(int64_t), st)) != hipSuccess) return (int)e;   // BOS
    if ((e = hipMemsetAsync(b->sums, 0, (size_t)2 * N * sizeof(float), st)) != hipSuccess) return (int)e;
    {   // fc term of the attention LSTM
        SegSpec s{b->fc, R, w->att_w_ih + R, 2 * R + E, R, 1};
        RC(gemm(stream, B, 4 * R, b->fc_gates, 4 * R, &s, 1, b->partial, b->partial_capacity, 0, nullptr));
    }
    // first step from BOS on B rows (AttModel.py:235-239): rows b of the [B*bd]-row arrays, one row per image
    float *st_a = b->state, *st_b = b->state + st_sz;
    // the first distribution is the model's own log_softmax: the temperature only enters at CaptionModel.py:203-204
    RC(decode_step(w, b, B, 1, st_a, st_b, b->logp_rows, 1.f, stream));
    // NOTE: after this call the B live rows of st_b / logp_rows[0] are rows 0..B-1 (cur = 1 per image)
    float *cur_state = st_b, *nxt_state = st_a;
    int cur = 1;
    for (int t = 0; t < L; ++t) {
        const size_t o = (size_t)t * N;
        float *sums_in = b->sums + (size_t)(t & 1) * N, *sums_out = b->sums + (size_t)((t + 1) & 1) * N;
        RC(capmi_beam_select(b->logp_rows + (size_t)t * N * V1, sums_in, B, cur, bd, V1, t == L - 1 ? 1 : 0, b->parent + o,
                             b->token + o, b->score + o, sums_out, b->ended + o, stream));
        if (t == L - 1) break;
        RC(capmi_beam_reorder(cur_state, nxt_state, b->parent + o, 4, B, cur, bd, R, stream));
        if ((e = hipMemcpyAsync(b->it, b->token + o, (size_t)N * sizeof(int64_t), hipMemcpyDeviceToDevice, st)) != hipSuccess)
            return (int)e;
        // step on B*bd rows, bd rows per image; writes the next state in place of the consumed one
        RC(decode_step(w, b, N, bd, nxt_state, cur_state, b->logp_rows + (size_t)(t + 1) * N * V1, b->temperature, stream));
        cur = bd;
        // cur_state now holds the new state; nxt_state is free
    }
    return 0;
}

}  // extern "C"
