// NewFC decoder on gfx950: maxout-LSTM cell kernels + whole-rollout drivers (forward and BPTT).
// Replaces NewFCModel.core / _prepare_feature (AttModel.py:904-945) over LSTMCore (FCModel.py:13-42)
// and the time loops of AttModel._forward / _sample for that model.  Same structure as the UpDown
// drivers (rollout.hip): one host call per rollout, no host sync, time-batched weight gradients.
#include "capmi_common.h"
#include "../../../include/capmi.h"

using namespace capmi;

namespace {

#define RC(x)                 \
    do {                      \
        int rc__ = (x);       \
        if (rc__) return rc__;\
    } while (0)

inline int grid_for(size_t work) {
    size_t b = (work + 255) / 256;
    if (b > 2048) b = 2048;
    return (int)(b < 1 ? 1 : b);
}

__global__ void maxout_cell_fwd_kernel(const float *__restrict__ partial, int splits, const float *__restrict__ b1,
                                       const float *__restrict__ b2, const float *__restrict__ c_prev,
                                       float *__restrict__ h, float *__restrict__ c, float *__restrict__ saved,
                                       const float *__restrict__ out_mask, float *__restrict__ h_drop, int N, int R) {
    const size_t total = (size_t)N * R, slab = (size_t)N * 5 * R;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int r = (int)(i / R), j = (int)(i % R);
        float s[5];
#pragma unroll
        for (int q = 0; q < 5; ++q) {
            const size_t col = (size_t)q * R + j;
            float v = 0.f;
            for (int k0 = 0; k0 < splits; k0 += 4) {
                float tv[4];
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    tv[u] = (k0 + u < splits) ? partial[(size_t)(k0 + u) * slab + (size_t)r * 5 * R + col] : 0.f;
                v += tv[0] + tv[1] + tv[2] + tv[3];
            }
            if (b1) v += b1[col];
            if (b2) v += b2[col];
            s[q] = v;
        }
        const float ig = sigmoid_f(s[0]), fg = sigmoid_f(s[1]), og = sigmoid_f(s[2]);
        const float cand = fmaxf(s[3], s[4]);
        const float cn = fg * c_prev[i] + ig * cand;
        const float hn = og * tanh_f(cn);
        c[i] = cn;
        h[i] = hn;
        float *sv = saved + (size_t)r * 5 * R + j;
        sv[0] = ig; sv[R] = fg; sv[2 * R] = og; sv[3 * (size_t)R] = s[3]; sv[4 * (size_t)R] = s[4];
        if (h_drop) h_drop[i] = out_mask ? hn * out_mask[i] : hn;
    }
}

__global__ void maxout_cell_bwd_kernel(const float *__restrict__ dh_a, const float *__restrict__ dh_a_mask,
                                       const float *__restrict__ dh_b, const float *__restrict__ dc_next,
                                       const float *__restrict__ saved, const float *__restrict__ c_prev,
                                       const float *__restrict__ c_new, float *__restrict__ d_sums,
                                       float *__restrict__ dc_prev, int N, int R) {
    const size_t total = (size_t)N * R;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int r = (int)(i / R), j = (int)(i % R);
        float dh = 0.f;
        if (dh_a) dh += dh_a_mask ? dh_a[i] * dh_a_mask[i] : dh_a[i];
        if (dh_b) dh += dh_b[i];
        const float *sv = saved + (size_t)r * 5 * R + j;
        const float ig = sv[0], fg = sv[R], og = sv[2 * R], ca = sv[3 * (size_t)R], cb = sv[4 * (size_t)R];
        const float cand = fmaxf(ca, cb);
        const float tc = tanh_f(c_new[i]);
        float dc = dh * og * (1.f - tc * tc);
        if (dc_next) dc += dc_next[i];
        float *ds = d_sums + (size_t)r * 5 * R + j;
        ds[0] = dc * cand * ig * (1.f - ig);
        ds[R] = dc * c_prev[i] * fg * (1.f - fg);
        ds[2 * R] = dh * tc * og * (1.f - og);
        const float dcand = dc * ig;
        ds[3 * (size_t)R] = ca >= cb ? dcand : 0.f;       // torch.max(a, b) routes the gradient to the larger chunk
        ds[4 * (size_t)R] = ca >= cb ? 0.f : dcand;
        dc_prev[i] = dc * fg;
    }
}

struct SegSpec {
    const float *A; int lda; const float *B; int ldb; int K; int a_row_div;
};
int gemm(void *stream, int al, int bl, int M, int N, float *C, int ldc, const SegSpec *segs, int nseg, float *partial,
         int64_t cap, int defer, int *splits_used, const float *bias = nullptr) {
    capmi_gemm_desc d{};
    d.nseg = nseg;
    for (int i = 0; i < nseg; ++i) {
        d.seg[i].A = segs[i].A; d.seg[i].lda = segs[i].lda; d.seg[i].B = segs[i].B; d.seg[i].ldb = segs[i].ldb;
        d.seg[i].K = segs[i].K; d.seg[i].a_row_div = segs[i].a_row_div > 0 ? segs[i].a_row_div : 1;
    }
    d.a_layout = al; d.b_layout = bl; d.M = M; d.N = N; d.C = C; d.ldc = ldc; d.bias = bias;
    d.partial = partial; d.partial_capacity = cap; d.splits = 0; d.defer_reduce = defer;
    const int rc = capmi_gemm_f32(&d, stream);
    if (splits_used) *splits_used = d.splits_used;
    return rc;
}

}  // namespace

extern "C" {

int capmi_maxout_cell_fwd(const float *partial, int splits, const float *b_i2h, const float *b_h2h, const float *c_prev,
                          float *h, float *c, float *saved, const float *out_mask, float *h_drop, int N, int R,
                          void *stream) {
    if (!partial || splits < 1 || !c_prev || !h || !c || !saved || N <= 0 || R <= 0) return CAPMI_EINVAL;
    hipLaunchKernelGGL(maxout_cell_fwd_kernel, dim3(grid_for((size_t)N * R)), dim3(256), 0, (hipStream_t)stream, partial,
                       splits, b_i2h, b_h2h, c_prev, h, c, saved, out_mask, h_drop, N, R);
    CAPMI_CHECK_LAUNCH();
    return 0;
}

int capmi_maxout_cell_bwd(const float *dh_a, const float *dh_a_mask, const float *dh_b, const float *dc_next,
                          const float *saved, const float *c_prev, const float *c_new, float *d_sums, float *dc_prev,
                          int N, int R, void *stream) {
    if (!saved || !c_prev || !c_new || !d_sums || !dc_prev || N <= 0 || R <= 0) return CAPMI_EINVAL;
    hipLaunchKernelGGL(maxout_cell_bwd_kernel, dim3(grid_for((size_t)N * R)), dim3(256), 0, (hipStream_t)stream, dh_a,
                       dh_a_mask, dh_b, dc_next, saved, c_prev, c_new, d_sums, dc_prev, N, R);
    CAPMI_CHECK_LAUNCH();
    return 0;
}

int capmi_newfc_rollout_fwd(const capmi_newfc_weights *w, capmi_newfc_rollout *r, void *stream) {
    if (!w || !r) return CAPMI_EINVAL;
    const int B = r->B, n = r->n, N = r->N, R = r->R, E = r->E, V1 = r->V1, T = r->T, L = r->L;
    if (B <= 0 || n <= 0 || N != B * n || T <= 0 || L < T || !r->partial) return CAPMI_EINVAL;
    // (r5: mode may carry CAPMI_SELECT_RAW -- a free-running rollout that stores the LOGITS, AttModel._sample(output_logsoftmax=0))
    if (((r->mode & 255) == 2 || r->teacher) && !r->forced) return CAPMI_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const size_t NR = (size_t)N * R;
    float *slabs = r->partial + CAPMI_WS_COUNTER_FLOATS;
    hipError_t e;
    if ((e = hipMemsetAsync(r->h, 0, NR * sizeof(float), st)) != hipSuccess) return (int)e;
    if ((e = hipMemsetAsync(r->c, 0, NR * sizeof(float), st)) != hipSuccess) return (int)e;
    if ((e = hipMemsetAsync(r->it, 0, (size_t)N * sizeof(int64_t), st)) != hipSuccess) return (int)e;
    if ((e = hipMemsetAsync(r->unfinished, 1, (size_t)N, st)) != hipSuccess) return (int)e;
    int splits = 1;
    // step "-1": the image (AttModel.py:925-927); h = c = 0 so only the i2h term matters but keep the general form
    {
        SegSpec s[2] = {{r->fc_emb, E, w->i2h_w, E, E, n}, {r->h, R, w->h2h_w, R, R, 1}};
        RC(gemm(stream, 0, 0, N, 5 * R, r->partial, 5 * R, s, 2, r->partial, r->partial_capacity, 1, &splits));
        RC(capmi_maxout_cell_fwd(slabs, splits, w->i2h_b, w->h2h_b, r->c, r->h + NR, r->c + NR, r->saved, nullptr, nullptr, N,
                                 R, stream));
    }
    for (int t = 0; t < T; ++t) {
        float *x = r->x + (size_t)t * N * E;
        const float *h_prev = r->h + (size_t)(t + 1) * NR, *c_prev = r->c + (size_t)(t + 1) * NR;
        float *h = r->h + (size_t)(t + 2) * NR, *c = r->c + (size_t)(t + 2) * NR;
        float *h_drop = r->h_drop + (size_t)t * NR;
        if (r->teacher)
            RC(capmi_embed_fwd(r->forced + t, r->forced_ld, r->it_all + (size_t)t * N, w->embed, nullptr, x, N, E, 0, stream));
        else
            RC(capmi_embed_fwd(r->it, 1, r->it_all + (size_t)t * N, w->embed, nullptr, x, N, E, 0, stream));
        SegSpec s[2] = {{x, E, w->i2h_w, E, E, 1}, {h_prev, R, w->h2h_w, R, R, 1}};
        RC(gemm(stream, 0, 0, N, 5 * R, r->partial, 5 * R, s, 2, r->partial, r->partial_capacity, 1, &splits));
        RC(capmi_maxout_cell_fwd(slabs, splits, w->i2h_b, w->h2h_b, c_prev, h, c, r->saved + (size_t)(t + 1) * N * 5 * R,
                                 r->drop_out ? r->drop_out + (size_t)t * NR : nullptr, h_drop, N, R, stream));
        SegSpec sl{h_drop, R, w->logit_w, R, R, 1};
        RC(gemm(stream, 0, 0, N, V1, r->partial, V1, &sl, 1, r->partial, r->partial_capacity, 1, &splits));
        RC(capmi_logsoftmax_select_partial(slabs, splits, (int64_t)N * V1, w->logit_b, N, V1, t, L, r->teacher ? 2 : r->mode,
                                           nullptr, r->temperature, r->gumbel ? r->gumbel + (size_t)t * N * V1 : nullptr,
                                           r->seed, r->forced, r->forced_ld, r->teacher ? 1 : 0, r->seq, L, r->it,
                                           r->unfinished, r->seq_logp, r->sel_logp, r->live, nullptr, nullptr, stream));
    }
    return 0;
}

int capmi_newfc_rollout_bwd(const capmi_newfc_weights *w, const capmi_newfc_rollout *r, const float *g_seq_logp,
                            capmi_newfc_bwd_scratch *s, capmi_newfc_grads *g, void *stream) {
    if (!w || !r || (!g_seq_logp && !(s && s->sparse)) || !s || !g) return CAPMI_EINVAL;
    const int B = r->B, n = r->n, N = r->N, R = r->R, E = r->E, V1 = r->V1, T = r->T, L = r->L;
    hipStream_t st = (hipStream_t)stream;
    const size_t NR = (size_t)N * R;
    const int TN = T * N;
    float *P = s->partial;
    const int64_t cap = s->partial_capacity;
    if ((r->mode & CAPMI_SELECT_RAW) && !r->teacher) {
        // the rollout returned logits: d(logits) is the loss gradient itself (sparse and / or dense part), no softmax Jacobian
        capmi_sparse_logp_grad sp = s->sparse ? *s->sparse : capmi_sparse_logp_grad{};
        sp.raw = 1;
        RC(capmi_logsoftmax_bwd_sparse(&sp, g_seq_logp, r->seq_logp, r->live, s->dlogits, N, L, T, V1, stream));
    } else if (s->sparse) RC(capmi_logsoftmax_bwd_sparse(s->sparse, g_seq_logp, r->seq_logp, r->live, s->dlogits, N, L, T, V1, stream));
    else RC(capmi_logsoftmax_bwd(g_seq_logp, r->seq_logp, r->live, s->dlogits, N, L, T, V1, stream));
    static const int env_group = capmi::knob("CAPMI_GEMM_GROUP", 1);
    const bool grouped = env_group != 0;
    capmi_group_gemm grp[3];
    int n_grp = 0;
    {
        SegSpec a{s->dlogits, V1, w->logit_w, R, V1, 1};
        RC(gemm(stream, 0, 1, TN, R, s->d_hdrop, R, &a, 1, P, cap, 0, nullptr));
        // r6: the three time-batched weight gradients (logit, i2h, h2h: K = T * N rows -- 1 050 at bs10 x 5, not a multiple of 4, which
        // sent them to the exact-fp32 tile kernel: 263 us of a 1.9-ms step) are listed and go out as ONE grouped launch at the end;
        // the logit bias gradient rides in it.  CAPMI_GEMM_GROUP=0: one launch each, as before.
        if (grouped) {
            grp[n_grp++] = capmi_group_gemm{s->dlogits, r->h_drop, g->logit_w, V1, R, R, TN, V1, R, 0, 0,
                                            (reinterpret_cast<uintptr_t>(g->logit_b) & 15) == 0 ? g->logit_b : nullptr};
            if (!grp[n_grp - 1].colsum) RC(capmi_colsum(s->dlogits, TN, V1, V1, g->logit_b, 0, stream));
        } else {
            SegSpec b{s->dlogits, V1, r->h_drop, R, TN, 1};
            RC(gemm(stream, 1, 1, V1, R, g->logit_w, R, &b, 1, P, cap, 0, nullptr));
            RC(capmi_colsum(s->dlogits, TN, V1, V1, g->logit_b, 0, stream));
        }
    }
    for (int t = T - 1; t >= -1; --t) {
        const bool last = (t == T - 1);
        const int slot = t + 1;                       // saved / d_sums slot (0 = image step)
        float *d_sums = s->d_sums + (size_t)slot * N * 5 * R;
        const float *dh_next = last ? nullptr : s->dh_prev + (size_t)((t + 1) & 1) * NR;
        float *dh_out = s->dh_prev + (size_t)(t & 1) * NR;
        const float *dc_in = last ? nullptr : s->dc + (size_t)((t + 1) & 1) * NR;
        float *dc_out = s->dc + (size_t)(t & 1) * NR;
        RC(capmi_maxout_cell_bwd(t >= 0 ? s->d_hdrop + (size_t)t * NR : nullptr,
                                 (t >= 0 && r->drop_out) ? r->drop_out + (size_t)t * NR : nullptr, dh_next, dc_in,
                                 r->saved + (size_t)slot * N * 5 * R, r->c + (size_t)slot * NR, r->c + (size_t)(slot + 1) * NR,
                                 d_sums, dc_out, N, R, stream));
        if (t >= 0) {   // dh_prev = d_sums W_h2h (the image step's predecessor state is the constant zero)
            SegSpec a{d_sums, 5 * R, w->h2h_w, R, 5 * R, 1};
            RC(gemm(stream, 0, 1, N, R, dh_out, R, &a, 1, P, cap, 0, nullptr));
        }
    }
    // time-batched gradients.  d_sums slots 1..T belong to the word steps, slot 0 to the image step.
    const float *ds_words = s->d_sums + (size_t)N * 5 * R;
    {
        SegSpec a{ds_words, 5 * R, r->x, E, TN, 1};                       // dW_i2h (words)
        if (!grouped) RC(gemm(stream, 1, 1, 5 * R, E, g->i2h_w, E, &a, 1, P, cap, 0, nullptr));
        // + image step: x = fc_emb[row / n]  -> materialise d_ximg and use the row-shared operand through a_row_div
        // (grouped: the image step WRITES the gradient here and the words' product is added to it by the grouped launch)
        capmi_gemm_desc d{};
        d.nseg = 1; d.a_layout = 1; d.b_layout = 1; d.M = 5 * R; d.N = E; d.C = g->i2h_w; d.ldc = E; d.accumulate = grouped ? 0 : 1;
        d.partial = P; d.partial_capacity = cap;
        // A = d_sums(image) [N,5R] stored [K=N][M]; B must be [K=N][E] = fc_emb repeated: expand once into d_x scratch
        // (N*E floats, tiny) with the embed kernel's gather: rows r -> fc_emb[r / n]
        RC(capmi_group_rowsum(s->d_sums, 1, 0, B, n, 5 * R, s->d_x_all /* reuse as [B,5R] sum */, stream));
        d.seg[0].A = s->d_x_all; d.seg[0].lda = 5 * R; d.seg[0].B = r->fc_emb; d.seg[0].ldb = E; d.seg[0].K = B;
        d.seg[0].a_row_div = 1;
        RC(capmi_gemm_f32(&d, stream));
        // d_fc_emb [B,E] = (sum over the n rows of the image of d_sums_img) W_i2h
        if (g->d_fc_emb) {
            SegSpec f{s->d_x_all, 5 * R, w->i2h_w, E, 5 * R, 1};
            RC(gemm(stream, 0, 1, B, E, g->d_fc_emb, E, &f, 1, P, cap, 0, nullptr));
        }
        // bias gradients: all T+1 steps
        RC(capmi_colsum(s->d_sums, (T + 1) * N, 5 * R, 5 * R, g->i2h_b, 0, stream));
        hipError_t e = hipMemcpyAsync(g->h2h_b, g->i2h_b, (size_t)5 * R * sizeof(float), hipMemcpyDeviceToDevice, st);
        if (e != hipSuccess) return (int)e;
        // dW_h2h: h_prev of word step t is slot t+1; of the image step it is slot 0 (zeros) -> words only
        SegSpec c{ds_words, 5 * R, r->h + NR, R, TN, 1};
        if (grouped) {
            grp[n_grp++] = capmi_group_gemm{ds_words, r->x, g->i2h_w, 5 * R, E, E, TN, 5 * R, E, 1, 0, nullptr};
            grp[n_grp++] = capmi_group_gemm{ds_words, r->h + NR, g->h2h_w, 5 * R, R, R, TN, 5 * R, R, 0, 0, nullptr};
        } else RC(gemm(stream, 1, 1, 5 * R, R, g->h2h_w, R, &c, 1, P, cap, 0, nullptr));
        // word embeddings (plain Embedding: no ReLU, no dropout)
        SegSpec x{ds_words, 5 * R, w->i2h_w, E, 5 * R, 1};
        RC(gemm(stream, 0, 1, TN, E, s->d_x_all, E, &x, 1, P, cap, 0, nullptr));
        e = hipMemsetAsync(g->embed, 0, (size_t)V1 * E * sizeof(float), st);
        if (e != hipSuccess) return (int)e;
        RC(capmi_embed_bwd(r->it_all, s->d_x_all, nullptr, nullptr, g->embed, TN, E, 0, stream));
    }
    if (n_grp) RC(capmi_gemm_group_tn(grp, n_grp, cap > CAPMI_WS_COUNTER_FLOATS ? P + CAPMI_WS_COUNTER_FLOATS : nullptr,
                                      cap > CAPMI_WS_COUNTER_FLOATS ? cap - CAPMI_WS_COUNTER_FLOATS : 0, stream));
    return 0;
}

}  // extern "C"
