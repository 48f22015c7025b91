// Elementwise / small-reduction kernels of the decode step for gfx950: token embedding, the LSTM
// cell pointwise stage (fused with the split-K reduction of the gate GEMM), their backward,
// dropout-mask generation, column sums and the fused clip+Adam update.  All are HBM/latency bound:
// 16-byte accesses where the layout allows, grid-stride loops capped at 2048 workgroups.
#include "capmi_common.h"
#include <cstdlib>
#include "profile.h"
#include "embed_bwd_det.h"
#include "../../../include/capmi.h"

using namespace capmi;

#include <atomic>
// the epoch word of the random streams (capmi_rng_bind_epoch): process-wide, one process per GPU
static std::atomic<const uint64_t *> g_rng_epoch{nullptr};
namespace capmi {
const uint64_t *rng_epoch() { return g_rng_epoch.load(std::memory_order_relaxed); }
}

namespace {

inline int grid_for(size_t work, int per_block = 256, int cap = 2048) {
    size_t b = (work + per_block - 1) / per_block;
    if (b > (size_t)cap) b = cap;
    if (b < 1) b = 1;
    return (int)b;
}

// all (possibly null) pointers 16-byte aligned
template <typename... P>
inline bool aligned16(P... p) {
    return ((... | reinterpret_cast<uintptr_t>(p)) & 15) == 0;
}

// ---------------------------------------------------------------- embedding
__global__ void embed_fwd_kernel(const int64_t *__restrict__ it, int it_stride, int64_t *__restrict__ it_save,
                                 const float *__restrict__ E, const float *__restrict__ mask, float *__restrict__ x,
                                 int N, int Ed, int relu, unsigned char *__restrict__ pl) {
    const size_t total = (size_t)N * Ed;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int r = (int)(i / Ed), c = (int)(i % Ed);
        const int64_t tok = it[(size_t)r * it_stride];
        if (it_save && c == 0) it_save[r] = tok;
        float v = E[(size_t)tok * Ed + c];
        if (relu) v = fmaxf(v, 0.f);
        if (mask) v *= mask[i];
        x[i] = v;
        if (pl) pl_store1(pl, r, c, v);          // A planes of x for the attention-LSTM gate GEMM (N <= 64)
    }
}

__global__ void embed_bwd_kernel(const int64_t *__restrict__ it, const float *__restrict__ dx,
                                 const float *__restrict__ x_saved, const float *__restrict__ mask,
                                 float *__restrict__ dE, int rows, int Ed, int relu) {
    const size_t total = (size_t)rows * Ed;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int r = (int)(i / Ed), c = (int)(i % Ed);
        float g = dx[i];
        if (mask) g *= mask[i];
        if (relu && !(x_saved[i] > 0.f)) {
            // x_saved = relu(E)*mask: zero either because relu clipped or the unit was dropped; in
            // both cases the gradient is zero (dropped units already have g*mask == 0).
            g = 0.f;
        }
        if (g != 0.f) atomicAdd(&dE[(size_t)it[r] * Ed + c], g);
    }
}

struct EmbedGrad {       // gradient reaching x = relu(E[tok]) * mask at (position, 4 columns)
    const float *dx, *x_saved, *mask; int Ed, relu;
    __device__ __forceinline__ f32x4 operator()(size_t pos, int c) const {
        const size_t o = pos * Ed + c;
        f32x4 g = *reinterpret_cast<const f32x4 *>(dx + o);
        if (mask) g *= *reinterpret_cast<const f32x4 *>(mask + o);
        if (relu) {
            const f32x4 xs = *reinterpret_cast<const f32x4 *>(x_saved + o);
#pragma unroll
            for (int e = 0; e < 4; ++e) if (!(xs[e] > 0.f)) g[e] = 0.f;
        }
        return g;
    }
};
__global__ __launch_bounds__(EBD_THREADS) void embed_bwd_det_kernel(const int64_t *__restrict__ it, int rows, int Ed,
                                                                   float *__restrict__ dE, const EmbedGrad g) {
    embed_bwd_det_body(it, rows, 0, rows, Ed, dE, g);
}

// ---------------------------------------------------------------- LSTM cell
__global__ void lstm_cell_fwd_kernel(const float *__restrict__ partial, int splits, const float *__restrict__ b_ih,
                                     const float *__restrict__ b_hh, const float *__restrict__ row_bias,
                                     int row_bias_div, const int *__restrict__ row_bias_idx,
                                     const float *__restrict__ c_prev, float *__restrict__ h,
                                     float *__restrict__ c, float *gates_act,
                                     const float *__restrict__ out_mask, float *__restrict__ h_drop, int N, int R,
                                     unsigned char *__restrict__ pl_h, unsigned char *__restrict__ pl_hd,
                                     const float *partial2, int splits2) {
    // (gates_act / partial2 carry no __restrict__: the batched-xt rollout hands the SAME buffer in as the second slab set and as the
    //  activated-gates output -- every thread reads its own elements of all slabs before it stores them)
    const size_t total = (size_t)N * R;
    const size_t slab = (size_t)N * 4 * R;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int r = (int)(i / R), j = (int)(i % R);
        float g[4] = {0.f, 0.f, 0.f, 0.f};
        // second slab set (the h-dependent K segments of the gate GEMM, computed ahead on the side stream)
        for (int s2 = 0; s2 < splits2; ++s2)
#pragma unroll
            for (int q = 0; q < 4; ++q) g[q] += partial2[(size_t)s2 * slab + (size_t)r * 4 * R + (size_t)q * R + j];
        // K-slice reduction: issue 8 slabs x 4 gates of independent loads per trip (the rolled form
        // waits for every load before the next one: 36 serial HBM latencies, 18 us per launch)
        const float *pb = partial + (size_t)r * 4 * R + j;
        for (int s0 = 0; s0 < splits; s0 += 8) {
            float tv[8][4];
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    tv[u][q] = (s0 + u < splits) ? pb[(size_t)(s0 + u) * slab + (size_t)q * R] : 0.f;
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int q = 0; q < 4; ++q) g[q] += tv[u][q];
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const size_t col = (size_t)q * R + j;
            float v = g[q];
            if (b_ih) v += b_ih[col];
            if (b_hh) v += b_hh[col];
            if (row_bias) v += row_bias[(size_t)(row_bias_idx ? row_bias_idx[r] : r / row_bias_div) * 4 * R + col];
            g[q] = v;
        }
        const float ig = sigmoid_f(g[0]), fg = sigmoid_f(g[1]), gg = tanh_f(g[2]), og = sigmoid_f(g[3]);
        const float cn = fg * c_prev[i] + ig * gg;
        const float hn = og * tanh_f(cn);
        c[i] = cn;
        h[i] = hn;
        if (gates_act) {
            float *ga = gates_act + (size_t)r * 4 * R + j;
            ga[0] = ig; ga[R] = fg; ga[2 * R] = gg; ga[3 * (size_t)R] = og;
        }
        const float hd = out_mask ? hn * out_mask[i] : hn;
        if (h_drop) h_drop[i] = hd;
        if (pl_h) pl_store1(pl_h, r, j, hn);
        if (pl_hd) pl_store1(pl_hd, r, j, hd);
    }
}

// The same cell, 16 bytes at a time, for the rollouts that also want the A PLANES of h / h_drop (round 3; R % 4 == 0, aligned
// operands, N <= 64): one thread per 4 hidden units, every slab load of a trip issued before the first is consumed
// (branch-free, clamped + multiplied by 0/1 like slab_sum8), and the planes leave as one 8-byte store per plane instead of
// three 2-byte stores per element.  Same summation order per element as the scalar kernel: slabs in order, 8 per trip.
__device__ __forceinline__ f32x4 slab_seq8(const float *p, int s0, int splits, size_t stride, f32x4 acc) {
    f32x4 part[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) part[u] = *reinterpret_cast<const f32x4 *>(p + (size_t)min(s0 + u, splits - 1) * stride);
#pragma unroll
    for (int u = 0; u < 8; ++u) acc += part[u] * ((s0 + u < splits) ? 1.f : 0.f);
    return acc;
}

__global__ __launch_bounds__(64) void lstm_cell_fwd_vec_kernel(
    const float *__restrict__ partial, int splits, const float *__restrict__ b_ih, const float *__restrict__ b_hh,
    const float *__restrict__ row_bias, int row_bias_div, const int *__restrict__ row_bias_idx,
    const float *__restrict__ c_prev, float *__restrict__ h, float *__restrict__ c, float *gates_act,
    const float *__restrict__ out_mask, float *__restrict__ h_drop, int N, int R, unsigned char *__restrict__ pl_h,
    unsigned char *__restrict__ pl_hd, const float *partial2, int splits2) {            // (gates_act may alias partial2, see above)
    const int R4 = R >> 2;
    const int i4 = blockIdx.x * blockDim.x + threadIdx.x;
    if (i4 >= N * R4) return;
    const int r = i4 / R4, j = (i4 - r * R4) * 4;
    const size_t i = (size_t)r * R + j;
    const size_t slab = (size_t)N * 4 * R;
    const f32x4 cp = *reinterpret_cast<const f32x4 *>(c_prev + i);
    f32x4 om = {1.f, 1.f, 1.f, 1.f};
    if (out_mask) om = *reinterpret_cast<const f32x4 *>(out_mask + i);
    f32x4 g[4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    // all 8 slabs x 4 gates of a trip are requested before the first is consumed (one memory round trip per 8 slabs; gate by gate
    // it was four), clamped + multiplied by 0/1 so that the trip is a fixed run of loads
    // The slabs of BOTH sets (this launch's gate GEMM, and -- when given -- the h-dependent K segments computed ahead on the side
    // stream) form one list of splits + splits2 slabs, walked 8 at a time.
    const float *pb = partial + (size_t)r * 4 * R + j;
    const float *pb2 = partial2 ? partial2 + (size_t)r * 4 * R + j : pb;
    const int tot = splits + splits2;
    // r4: the bias rows are requested in the same round trip as the slabs (they used to be requested behind the slab sum: a
    // second trip); added afterwards in the old order, so the results are bit-identical
    f32x4 bi[4], bh[4], br[4];
    {
        const float *rb = row_bias ? row_bias + (size_t)(row_bias_idx ? row_bias_idx[r] : r / row_bias_div) * 4 * R : nullptr;
        const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const size_t col = (size_t)q * R + j;
            bi[q] = b_ih ? *reinterpret_cast<const f32x4 *>(b_ih + col) : z4;
            bh[q] = b_hh ? *reinterpret_cast<const f32x4 *>(b_hh + col) : z4;
            br[q] = rb ? *reinterpret_cast<const f32x4 *>(rb + col) : z4;
        }
    }
    for (int s0 = 0; s0 < tot; s0 += 8) {
        f32x4 tv[8][4];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int si = min(s0 + u, tot - 1);
            const float *ps = si < splits ? pb + (size_t)si * slab : pb2 + (size_t)(si - splits) * slab;
#pragma unroll
            for (int q = 0; q < 4; ++q) tv[u][q] = *reinterpret_cast<const f32x4 *>(ps + (size_t)q * R);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const float on = (s0 + u < tot) ? 1.f : 0.f;
#pragma unroll
            for (int q = 0; q < 4; ++q) g[q] += tv[u][q] * on;
        }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        if (b_ih) g[q] += bi[q];
        if (b_hh) g[q] += bh[q];
        if (row_bias) g[q] += br[q];
    }
    f32x4 ig, fg, gg, og, cn, hn, hd;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        ig[e] = sigmoid_f(g[0][e]); fg[e] = sigmoid_f(g[1][e]); gg[e] = tanh_f(g[2][e]); og[e] = sigmoid_f(g[3][e]);
        cn[e] = fg[e] * cp[e] + ig[e] * gg[e];
        hn[e] = og[e] * tanh_f(cn[e]);
        hd[e] = out_mask ? hn[e] * om[e] : hn[e];
    }
    *reinterpret_cast<f32x4 *>(c + i) = cn;
    *reinterpret_cast<f32x4 *>(h + i) = hn;
    if (gates_act) {
        float *ga = gates_act + (size_t)r * 4 * R + j;
        *reinterpret_cast<f32x4 *>(ga) = ig;
        *reinterpret_cast<f32x4 *>(ga + R) = fg;
        *reinterpret_cast<f32x4 *>(ga + 2 * (size_t)R) = gg;
        *reinterpret_cast<f32x4 *>(ga + 3 * (size_t)R) = og;
    }
    if (h_drop) *reinterpret_cast<f32x4 *>(h_drop + i) = hd;
    if (pl_h) pl_store4(pl_h, r, j, hn);
    if (pl_hd) pl_store4(pl_hd, r, j, hd);
}

// dh_b / dh_c may arrive as split-K slabs of the dX GEMMs of the previous BPTT step (b_splits / c_splits > 1,
// slab s at + s * stride): the cell finishes those reductions itself, so the dX GEMMs need no reduce launch.
__global__ void lstm_cell_bwd_kernel(const float *__restrict__ dh_a, int ld_a, const float *__restrict__ dh_a_mask,
                                     const float *__restrict__ dh_b, int ld_b, int b_splits, size_t b_stride,
                                     const float *__restrict__ dh_c, int ld_c, int c_splits, size_t c_stride,
                                     const float *__restrict__ dc_next, const float *__restrict__ gates_act,
                                     const float *__restrict__ c_prev, const float *__restrict__ c_new,
                                     float *__restrict__ d_gates, float *__restrict__ dc_prev, int N, int R,
                                     unsigned char *__restrict__ pl_dg) {
    const size_t total = (size_t)N * R;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int r = (int)(i / R), j = (int)(i % R);
        float dh = 0.f;
        if (dh_a) {
            const float v = dh_a[(size_t)r * ld_a + j];
            dh += dh_a_mask ? v * dh_a_mask[i] : v;
        }
        if (dh_b) {
            const float *p = dh_b + (size_t)r * ld_b + j;
            for (int s0 = 0; s0 < b_splits; s0 += 8) {     // 8 independent slab loads in flight
                float part[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) part[u] = (s0 + u < b_splits) ? p[(s0 + u) * b_stride] : 0.f;
                dh += ((part[0] + part[1]) + (part[2] + part[3])) + ((part[4] + part[5]) + (part[6] + part[7]));
            }
        }
        if (dh_c) {
            const float *p = dh_c + (size_t)r * ld_c + j;
            for (int s0 = 0; s0 < c_splits; s0 += 8) {     // 8 independent slab loads in flight
                float part[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) part[u] = (s0 + u < c_splits) ? p[(s0 + u) * c_stride] : 0.f;
                dh += ((part[0] + part[1]) + (part[2] + part[3])) + ((part[4] + part[5]) + (part[6] + part[7]));
            }
        }
        const float *ga = gates_act + (size_t)r * 4 * R + j;
        const float ig = ga[0], fg = ga[R], gg = ga[2 * R], og = ga[3 * (size_t)R];
        const float tc = tanh_f(c_new[i]);
        float dc = dh * og * (1.f - tc * tc);
        if (dc_next) dc += dc_next[i];
        float *dg = d_gates + (size_t)r * 4 * R + j;
        const float d0 = dc * gg * ig * (1.f - ig), d1 = dc * c_prev[i] * fg * (1.f - fg);
        const float d2 = dc * ig * (1.f - gg * gg), d3 = dh * tc * og * (1.f - og);
        dg[0] = d0; dg[R] = d1; dg[2 * R] = d2; dg[3 * (size_t)R] = d3;
        if (pl_dg) {                             // A planes of d_gates [N, 4R] for the dX GEMM of this BPTT step
            pl_store1(pl_dg, r, j, d0); pl_store1(pl_dg, r, R + j, d1);
            pl_store1(pl_dg, r, 2 * R + j, d2); pl_store1(pl_dg, r, 3 * R + j, d3);
        }
        dc_prev[i] = dc * fg;
    }
}

// 16-byte variant of the backward cell (R % 4 == 0, aligned operands: every BASELINE size): one thread per 4 outputs, one
// wave per workgroup, slab loads BRANCH-FREE (index clamped, contribution multiplied by 0/1) so that a trip is a fixed run
// of loads with exact vmcnt waits; same pairwise summation order as the scalar kernel (bit-identical).  5.5 -> 5.1 us; the
// same treatment of the forward cell measured no gain in round 1 (7.2 vs 7.5 us) and again in round 2 with all 32 slab
// loads of a thread issued up front (7.1 vs 7.4 us): launch + one first-touch round trip is the floor there.
__device__ __forceinline__ f32x4 slab_sum8(const float *p, int s0, int splits, size_t stride) {
    f32x4 part[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) part[u] = *reinterpret_cast<const f32x4 *>(p + (size_t)min(s0 + u, splits - 1) * stride);
#pragma unroll
    for (int u = 0; u < 8; ++u) part[u] *= (s0 + u < splits) ? 1.f : 0.f;
    return ((part[0] + part[1]) + (part[2] + part[3])) + ((part[4] + part[5]) + (part[6] + part[7]));
}

__global__ __launch_bounds__(64) void lstm_cell_bwd_vec_kernel(
    const float *__restrict__ dh_a, int ld_a, const float *__restrict__ dh_a_mask, const float *__restrict__ dh_b, int ld_b,
    int b_splits, size_t b_stride, const float *__restrict__ dh_c, int ld_c, int c_splits, size_t c_stride,
    const float *__restrict__ dc_next, const float *__restrict__ gates_act, const float *__restrict__ c_prev,
    const float *__restrict__ c_new, float *__restrict__ d_gates, float *__restrict__ dc_prev, int N, int R,
    unsigned char *__restrict__ pl_dg) {
    const int R4 = R >> 2;
    const int i4 = blockIdx.x * blockDim.x + threadIdx.x;
    if (i4 >= N * R4) return;
    const int r = i4 / R4, j = (i4 - r * R4) * 4;
    const size_t i = (size_t)r * R + j;
    // the always-present operands first: their loads are in flight while the slabs are summed
    const float *ga = gates_act + (size_t)r * 4 * R + j;
    const f32x4 ig = *reinterpret_cast<const f32x4 *>(ga), fg = *reinterpret_cast<const f32x4 *>(ga + R);
    const f32x4 gg = *reinterpret_cast<const f32x4 *>(ga + 2 * (size_t)R), og = *reinterpret_cast<const f32x4 *>(ga + 3 * (size_t)R);
    const f32x4 cnew = *reinterpret_cast<const f32x4 *>(c_new + i), cprev = *reinterpret_cast<const f32x4 *>(c_prev + i);
    f32x4 dh = {0.f, 0.f, 0.f, 0.f};
    if (dh_a) {
        f32x4 v = *reinterpret_cast<const f32x4 *>(dh_a + (size_t)r * ld_a + j);
        if (dh_a_mask) v *= *reinterpret_cast<const f32x4 *>(dh_a_mask + i);
        dh += v;
    }
    if (dh_b)
        for (int s0 = 0; s0 < b_splits; s0 += 8) dh += slab_sum8(dh_b + (size_t)r * ld_b + j, s0, b_splits, b_stride);
    if (dh_c)
        for (int s0 = 0; s0 < c_splits; s0 += 8) dh += slab_sum8(dh_c + (size_t)r * ld_c + j, s0, c_splits, c_stride);
    f32x4 dcn = {0.f, 0.f, 0.f, 0.f};
    if (dc_next) dcn = *reinterpret_cast<const f32x4 *>(dc_next + i);
    f32x4 d0, d1, d2, d3, dcp;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float tc = tanh_f(cnew[e]);
        float dc = dh[e] * og[e] * (1.f - tc * tc);
        if (dc_next) dc += dcn[e];
        d0[e] = dc * gg[e] * ig[e] * (1.f - ig[e]);
        d1[e] = dc * cprev[e] * fg[e] * (1.f - fg[e]);
        d2[e] = dc * ig[e] * (1.f - gg[e] * gg[e]);
        d3[e] = dh[e] * tc * og[e] * (1.f - og[e]);
        dcp[e] = dc * fg[e];
    }
    float *dg = d_gates + (size_t)r * 4 * R + j;
    *reinterpret_cast<f32x4 *>(dg) = d0;
    *reinterpret_cast<f32x4 *>(dg + R) = d1;
    *reinterpret_cast<f32x4 *>(dg + 2 * (size_t)R) = d2;
    *reinterpret_cast<f32x4 *>(dg + 3 * (size_t)R) = d3;
    *reinterpret_cast<f32x4 *>(dc_prev + i) = dcp;
    if (pl_dg) {                                 // A planes of d_gates [N, 4R] for the dX GEMM of this BPTT step
        pl_store4(pl_dg, r, j, d0); pl_store4(pl_dg, r, R + j, d1);
        pl_store4(pl_dg, r, 2 * R + j, d2); pl_store4(pl_dg, r, 3 * R + j, d3);
    }
}

// ---------------------------------------------------------------- misc
__global__ void dropout_mask_kernel(float *__restrict__ mask, size_t count, float p, uint64_t seed, uint64_t offset,
                                    const uint64_t *__restrict__ epoch) {
    const Philox rng(epoch_seed(seed, epoch));
    const float scale = 1.f / (1.f - p);
    const size_t quads = (count + 3) / 4;
    for (size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x; q < quads; q += (size_t)gridDim.x * blockDim.x) {
        uint32_t o[4];
        rng.gen(offset + q, 0x6d61736bULL /* "mask" stream */, o);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const size_t i = q * 4 + k;
            if (i < count) mask[i] = (u01(o[k]) < p) ? 0.f : scale;
        }
    }
}

// All dropout masks of one rollout in ONE launch (round 2: four launches of the kernel above, 5 us + a kernel boundary each,
// sat in front of every training rollout).  Segment i covers mask_i[count_i] with Philox offset off_i; elements whose
// row index within a slab (`row_len` floats per row, `rows` rows per slab) is >= keep_from are written as 1.0 -- the
// greedy-baseline rows of the fused SCST rollout run in eval mode (loss_wrapper.py:57-60).
struct MaskSegs {
    float *mask[CAPMI_MAX_MASKS];
    unsigned long long count[CAPMI_MAX_MASKS], offset[CAPMI_MAX_MASKS];
    int row_len[CAPMI_MAX_MASKS], rows[CAPMI_MAX_MASKS], keep_from[CAPMI_MAX_MASKS];
    int n;
};
__global__ void dropout_masks_kernel(const MaskSegs sg, float p, uint64_t seed, const uint64_t *__restrict__ epoch) {
    const Philox rng(epoch_seed(seed, epoch));
    const float scale = 1.f / (1.f - p);
    for (int i = 0; i < sg.n; ++i) {
        const size_t count = sg.count[i], quads = (count + 3) / 4;
        float *mask = sg.mask[i];
        const int row_len = sg.row_len[i], rows = sg.rows[i], keep_from = sg.keep_from[i];
        const bool vec = keep_from >= rows && (reinterpret_cast<uintptr_t>(mask) & 15) == 0;      // no eval-mode rows, aligned
        for (size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x; q < quads; q += (size_t)gridDim.x * blockDim.x) {
            uint32_t o[4];
            rng.gen(sg.offset[i] + q, 0x6d61736bULL, o);
            if (vec && q * 4 + 3 < count) {       // one 16-byte store (r4: four 4-byte stores 16 bytes apart ran at 2.4 TB/s)
                f32x4 v;
#pragma unroll
                for (int k = 0; k < 4; ++k) v[k] = (u01(o[k]) < p) ? 0.f : scale;
                *reinterpret_cast<f32x4 *>(mask + q * 4) = v;
                continue;
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const size_t e = q * 4 + k;
                if (e < count) {
                    const bool eval_row = keep_from < rows && (int)((e / row_len) % rows) >= keep_from;
                    mask[e] = eval_row ? 1.f : ((u01(o[k]) < p) ? 0.f : scale);
                }
            }
        }
    }
}

// initial state of a rollout (slot 0 of h / c, BOS tokens, unfinished flags) in one launch instead of six memsets
__global__ void rollout_init_kernel(float *__restrict__ a, float *__restrict__ b, float *__restrict__ c, float *__restrict__ d,
                                    size_t count, int64_t *__restrict__ it, uint8_t *__restrict__ unfinished, int N) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (size_t)gridDim.x * blockDim.x) {
        a[i] = 0.f; b[i] = 0.f;
        if (c) c[i] = 0.f;
        if (d) d[i] = 0.f;
        if (i < (size_t)N) { it[i] = 0; unfinished[i] = 1; }
    }
}

// Column sums of a [rows, cols] matrix (bias gradients over all T*N rows).  1024 threads = 16 column quads x 64 row
// slices: a wave reads four 256-byte row segments per instruction, every thread keeps <= rows/64 INDEPENDENT 16-byte
// loads in flight (the old one-column-per-thread loop was a serial chain of 150 scalar loads: 45 us for 19 MB), and
// the 64 slices meet in LDS in a fixed order (deterministic, no atomics).
constexpr int CS_Q = 16, CS_R = 64;
__global__ __launch_bounds__(CS_Q * CS_R) void colsum_kernel(const float *__restrict__ in, int rows, int cols, int ld,
                                                            float *__restrict__ out, int accumulate, int vec, int rsplit) {
    // gridDim.y = rsplit row ranges per column block (narrow matrices: d_model-wide bias gradients over thousands of rows
    // would otherwise run on cols/64 workgroups); with rsplit > 1 the partial sums meet in `out` by atomicAdd (zeroed by
    // the host unless accumulating)
    __shared__ f32x4 red[CS_R][CS_Q];
    const int cq = threadIdx.x % CS_Q, ry = threadIdx.x / CS_Q;
    const int col = (blockIdx.x * CS_Q + cq) * 4;
    const int r_lo = (int)(((long long)rows * blockIdx.y) / rsplit), r_hi = (int)(((long long)rows * (blockIdx.y + 1)) / rsplit);
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    if (col < cols) {
        if (vec) {
            const float *p = in + col;
            int r = r_lo + ry;
            for (; r + 3 * CS_R < r_hi; r += 4 * CS_R) {
                const f32x4 a = *reinterpret_cast<const f32x4 *>(p + (size_t)r * ld);
                const f32x4 b = *reinterpret_cast<const f32x4 *>(p + (size_t)(r + CS_R) * ld);
                const f32x4 c = *reinterpret_cast<const f32x4 *>(p + (size_t)(r + 2 * CS_R) * ld);
                const f32x4 d = *reinterpret_cast<const f32x4 *>(p + (size_t)(r + 3 * CS_R) * ld);
                s += (a + b) + (c + d);
            }
            for (; r < r_hi; r += CS_R) s += *reinterpret_cast<const f32x4 *>(p + (size_t)r * ld);
        } else {
            for (int r = r_lo + ry; r < r_hi; r += CS_R)
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (col + k < cols) s[k] += in[(size_t)r * ld + col + k];
        }
    }
    red[ry][cq] = s;
    __syncthreads();
    // 64 slices -> 4 (threads ry < 4 each fold 16 slices), then thread ry == 0 folds the 4
    if (ry < 4) {
        f32x4 t = red[ry * 16][cq];
#pragma unroll
        for (int k = 1; k < 16; ++k) t += red[ry * 16 + k][cq];
        red[ry * 16][cq] = t;
    }
    __syncthreads();
    if (ry == 0 && col < cols) {
        const f32x4 t = (red[0][cq] + red[16][cq]) + (red[32][cq] + red[48][cq]);
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (col + k < cols) {
                if (rsplit > 1) atomicAdd(&out[col + k], t[k]);
                else out[col + k] = accumulate ? out[col + k] + t[k] : t[k];
            }
    }
}

// Many column sums in one launch (all bias gradients of a backward): blockIdx.y = item, blockIdx.x walks the item's
// 64-column blocks; every block sums ALL rows of its columns (deterministic, no atomics, no zero-fill launch).  1024 threads =
// 16 column quads x 64 row slices.  (r4: 2.7 TB/s on the Transformer step's 1.5 GB; 128-column blocks -- 512-byte row segments, half
// the workgroups -- were measured and are SLOWER, 790 vs 550 us: the launch lives on the number of CUs that have a block to walk.)
constexpr int CB_Q = 16, CB_R = 64;
__device__ __forceinline__ void colsum_item_body(const capmi_colsum_item &it) {
    __shared__ f32x4 red[CB_R][CB_Q];
    const int cq = threadIdx.x % CB_Q, ry = threadIdx.x / CB_Q;
    const int cblocks = (it.cols + 4 * CB_Q - 1) / (4 * CB_Q);
    const bool vec = (reinterpret_cast<uintptr_t>(it.in) & 15) == 0 && it.ld % 4 == 0 && it.cols % 4 == 0;
    for (int cb = blockIdx.x; cb < cblocks; cb += gridDim.x) {
        const int col = (cb * CB_Q + cq) * 4;
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
        if (col < it.cols) {
            if (vec) {
                const float *p = it.in + col;
                int r = ry;
                for (; r + 7 * CB_R < it.rows; r += 8 * CB_R) {          // 8 rows in flight per thread
                    f32x4 v[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const f32x4 *>(p + (size_t)(r + u * CB_R) * it.ld);
                    s += ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
                }
                for (; r + 3 * CB_R < it.rows; r += 4 * CB_R) {
                    const f32x4 a = *reinterpret_cast<const f32x4 *>(p + (size_t)r * it.ld);
                    const f32x4 b = *reinterpret_cast<const f32x4 *>(p + (size_t)(r + CB_R) * it.ld);
                    const f32x4 c = *reinterpret_cast<const f32x4 *>(p + (size_t)(r + 2 * CB_R) * it.ld);
                    const f32x4 d = *reinterpret_cast<const f32x4 *>(p + (size_t)(r + 3 * CB_R) * it.ld);
                    s += (a + b) + (c + d);
                }
                for (; r < it.rows; r += CB_R) s += *reinterpret_cast<const f32x4 *>(p + (size_t)r * it.ld);
            } else {
                for (int r = ry; r < it.rows; r += CB_R)
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        if (col + k < it.cols) s[k] += it.in[(size_t)r * it.ld + col + k];
            }
        }
        red[ry][cq] = s;
        __syncthreads();
        if (ry < CB_R / 16) {
            f32x4 t = red[ry * 16][cq];
#pragma unroll
            for (int k = 1; k < 16; ++k) t += red[ry * 16 + k][cq];
            red[ry * 16][cq] = t;
        }
        __syncthreads();
        if (ry == 0 && col < it.cols) {
            static_assert(CB_R == 64, "four folded groups");
            const f32x4 t = (red[0][cq] + red[16][cq]) + (red[32][cq] + red[48][cq]);
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (col + k < it.cols) {
                    const float v = it.accumulate ? it.out[col + k] + t[k] : t[k];
                    it.out[col + k] = v;
                    if (it.out2) it.out2[col + k] = v;
                }
        }
        __syncthreads();
    }
}
__global__ __launch_bounds__(CB_Q * CB_R) void colsum_batch_kernel(const capmi_colsum_item *__restrict__ items) {
    colsum_item_body(items[blockIdx.y]);
}
struct ColsumArgs {
    capmi_colsum_item items[CAPMI_COLSUM_ARGS_MAX];
};
__global__ __launch_bounds__(CB_Q * CB_R) void colsum_batch_args_kernel(const ColsumArgs a) {
    // (static indices: a dynamically indexed kernel-argument array would be copied to scratch)
    capmi_colsum_item it = a.items[0];
#pragma unroll
    for (int i = 1; i < CAPMI_COLSUM_ARGS_MAX; ++i)
        if (blockIdx.y == i) it = a.items[i];
    colsum_item_body(it);
}

__global__ void group_rowsum_kernel(const float *__restrict__ in, int T, size_t slab, int groups, int group, int cols,
                                    float *__restrict__ out) {
    const size_t total = (size_t)groups * cols;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int g = (int)(i / cols), c = (int)(i % cols);
        float s = 0.f;
        for (int t = 0; t < T; ++t)
            for (int j = 0; j < group; ++j) s += in[t * slab + (size_t)(g * group + j) * cols + c];
        out[i] = s;
    }
}

// The same sum for cols % 4 == 0 and 16-byte aligned operands: one thread per (group, column quad, quarter of the T * group
// rows), its rows fetched 8 at a time with all loads of a batch in flight, the four quarters combined through LDS.  The
// scalar kernel above walks its 100 rows (SCST: T = 20, group = 5) one dependent round trip after the other: 31 us for 16 MB.
constexpr int GRS_PARTS = 4;
__global__ __launch_bounds__(256) void group_rowsum_v4_kernel(const float *__restrict__ in, int T, size_t slab, int groups,
                                                              int group, int cols, float *__restrict__ out) {
    __shared__ f32x4 sh[256];
    const int q4 = cols >> 2;
    const int part = threadIdx.x & (GRS_PARTS - 1);
    const size_t qi = (size_t)blockIdx.x * (256 / GRS_PARTS) + (threadIdx.x >> 2);       // (group, column quad)
    const size_t nq = (size_t)groups * q4;
    const int rows = T * group, per = (rows + GRS_PARTS - 1) / GRS_PARTS;
    const int r0 = part * per, r1 = min(rows, r0 + per);
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    if (qi < nq) {
        const int g = (int)(qi / q4), c = (int)(qi % q4) * 4;
        const float *base = in + (size_t)g * group * cols + c;
        for (int i0 = r0; i0 < r1; i0 += 8) {
            f32x4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = min(i0 + u, r1 - 1);
                v[u] = *reinterpret_cast<const f32x4 *>(base + (size_t)(i / group) * slab + (size_t)(i % group) * cols);
            }
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (i0 + u < r1) s += v[u];
        }
    }
    sh[threadIdx.x] = s;
    __syncthreads();
    if (part == 0 && qi < nq) {
        const f32x4 r = ((sh[threadIdx.x] + sh[threadIdx.x + 1]) + sh[threadIdx.x + 2]) + sh[threadIdx.x + 3];
        *reinterpret_cast<f32x4 *>(out + qi * 4) = r;
    }
}

__global__ void relu_mask_bwd_kernel(const float *__restrict__ dy, const float *__restrict__ y_ref,
                                     const float *__restrict__ mask, float *__restrict__ dx, size_t count) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (size_t)gridDim.x * blockDim.x) {
        float g = dy[i];
        if (mask) g *= mask[i];
        if (y_ref && !(y_ref[i] > 0.f)) g = 0.f;
        dx[i] = g;
    }
}

// dx = y_ref > 0 ? dy * scale : 0 -- the ReLU + dropout Jacobian when y_ref is the layer's output AFTER the mask (y = relu(pre) * mask:
// y > 0 exactly where the mask kept the element and the ReLU passed it, and there the mask's value is the constant 1 / (1 - p)): the
// [rows x d_ff] mask is not read again (r6: a quarter of the traffic of the Transformer's 43 launches, 1.04 ms per XE step)
__global__ void relu_scale_bwd_kernel(const f32x4 *__restrict__ dy, const f32x4 *__restrict__ y_ref, float scale, f32x4 *__restrict__ dx,
                                      size_t quads) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < quads; i += (size_t)gridDim.x * blockDim.x) {
        const f32x4 g = dy[i], y = y_ref[i];
        f32x4 o;
#pragma unroll
        for (int k = 0; k < 4; ++k) o[k] = y[k] > 0.f ? g[k] * scale : 0.f;
        dx[i] = o;
    }
}

__global__ void adam_kernel(float *__restrict__ p, const float *__restrict__ g, float *__restrict__ m,
                            float *__restrict__ v, size_t count, float lr, float b1, float b2, float eps, float wd,
                            float clip, float gscale, float bc1, float bc2_sqrt, const capmi_step_state *__restrict__ dyn) {
    // torch.optim.Adam (non-amsgrad): m = b1 m + (1-b1) g; v = b2 v + (1-b2) g^2;
    // p -= lr/bc1 * m / (sqrt(v)/sqrt(bc2) + eps)
    if (dyn) { lr = dyn->lr; bc1 = dyn->bc1; bc2_sqrt = dyn->bc2_sqrt; }      // capmi_adam_step_dyn: the step record in device memory
    const size_t quads = count / 4;
    const f32x4 *g4 = reinterpret_cast<const f32x4 *>(g);
    f32x4 *p4 = reinterpret_cast<f32x4 *>(p), *m4 = reinterpret_cast<f32x4 *>(m), *v4 = reinterpret_cast<f32x4 *>(v);
    const float step_size = lr / bc1;
    for (size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x; q < quads; q += (size_t)gridDim.x * blockDim.x) {
        f32x4 gg = g4[q], pp = p4[q], mm = m4[q], vv = v4[q];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float x = gg[k] * gscale;
            if (clip > 0.f) x = fminf(fmaxf(x, -clip), clip);
            if (wd != 0.f) x += wd * pp[k];
            mm[k] = b1 * mm[k] + (1.f - b1) * x;
            vv[k] = b2 * vv[k] + (1.f - b2) * x * x;
            pp[k] -= step_size * mm[k] / (sqrtf(vv[k]) / bc2_sqrt + eps);
        }
        p4[q] = pp; m4[q] = mm; v4[q] = vv;
    }
    // tail
    const size_t base = quads * 4;
    const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (tid < count - base) {
        const size_t i = base + tid;
        float x = g[i] * gscale;
        if (clip > 0.f) x = fminf(fmaxf(x, -clip), clip);
        if (wd != 0.f) x += wd * p[i];
        const float mi = b1 * m[i] + (1.f - b1) * x;
        const float vi = b2 * v[i] + (1.f - b2) * x * x;
        m[i] = mi; v[i] = vi;
        p[i] -= step_size * mi / (sqrtf(vi) / bc2_sqrt + eps);
    }
}

// Round 3 variant: two independent quads per trip (8 x 16-byte loads in flight per thread instead of 4) and non-temporal accesses
// for the streams nobody re-reads before the next optimizer step (g, m, v: 3/4 of the 1.46 GB), so that the 208 MB of updated
// parameters -- the first thing the next forward streams -- are what stays in the 256 MiB Infinity Cache.
template <bool NT>
__global__ void adam2_kernel(float *__restrict__ p, const float *__restrict__ g, float *__restrict__ m,
                             float *__restrict__ v, size_t quads, float lr, float b1, float b2, float eps, float wd,
                             float clip, float gscale, float bc1, float bc2_sqrt, const capmi_step_state *__restrict__ dyn) {
    const f32x4 *g4 = reinterpret_cast<const f32x4 *>(g);
    f32x4 *p4 = reinterpret_cast<f32x4 *>(p), *m4 = reinterpret_cast<f32x4 *>(m), *v4 = reinterpret_cast<f32x4 *>(v);
    if (dyn) { lr = dyn->lr; bc1 = dyn->bc1; bc2_sqrt = dyn->bc2_sqrt; }      // capmi_adam_step_dyn: the step record in device memory
    const float step_size = lr / bc1;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t q0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x; q0 < quads; q0 += 2 * stride) {
        const size_t q1 = q0 + stride;
        const bool two = q1 < quads;
        const size_t q1c = two ? q1 : q0;
        f32x4 gg[2], pp[2], mm[2], vv[2];
        if (NT) {
            gg[0] = __builtin_nontemporal_load(g4 + q0); gg[1] = __builtin_nontemporal_load(g4 + q1c);
            mm[0] = __builtin_nontemporal_load(m4 + q0); mm[1] = __builtin_nontemporal_load(m4 + q1c);
            vv[0] = __builtin_nontemporal_load(v4 + q0); vv[1] = __builtin_nontemporal_load(v4 + q1c);
        } else {
            gg[0] = g4[q0]; gg[1] = g4[q1c]; mm[0] = m4[q0]; mm[1] = m4[q1c]; vv[0] = v4[q0]; vv[1] = v4[q1c];
        }
        pp[0] = p4[q0]; pp[1] = p4[q1c];
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float x = gg[u][k] * gscale;
                if (clip > 0.f) x = fminf(fmaxf(x, -clip), clip);
                if (wd != 0.f) x += wd * pp[u][k];
                mm[u][k] = b1 * mm[u][k] + (1.f - b1) * x;
                vv[u][k] = b2 * vv[u][k] + (1.f - b2) * x * x;
                pp[u][k] -= step_size * mm[u][k] / (sqrtf(vv[u][k]) / bc2_sqrt + eps);
            }
        p4[q0] = pp[0];
        if (NT) { __builtin_nontemporal_store(mm[0], m4 + q0); __builtin_nontemporal_store(vv[0], v4 + q0); }
        else { m4[q0] = mm[0]; v4[q0] = vv[0]; }
        if (two) {
            p4[q1] = pp[1];
            if (NT) { __builtin_nontemporal_store(mm[1], m4 + q1); __builtin_nontemporal_store(vv[1], v4 + q1); }
            else { m4[q1] = mm[1]; v4[q1] = vv[1]; }
        }
    }
}

// the per-iteration record of a captured training step (capmi.h capmi_step_state): one thread, first launch of every iteration
__global__ void step_advance_kernel(capmi_step_state *st, float b1, float b2) {
    st->epoch += 1;
    const int step = st->adam_step + 1;
    st->adam_step = step;
    st->bc1 = (float)(1.0 - pow((double)b1, (double)step));
    st->bc2_sqrt = (float)sqrt(1.0 - pow((double)b2, (double)step));
}
__global__ void step_set_lr_kernel(capmi_step_state *st, float lr) { st->lr = lr; }

// one workgroup: N <= a few thousand rows.  reward[N] (written when N > 0 rows exist and mean_out is given) = mean advantage, what
// LossWrapper reports as out['reward'] (loss_wrapper.py:72) -- an ATen mean launch otherwise
__global__ void scst_advantage_kernel(const double *__restrict__ scores, int N, int n, float *__restrict__ reward,
                                      float *__restrict__ mean_out) {
    __shared__ float scratch[32];
    float s = 0.f;
    for (int r = threadIdx.x; r < N; r += blockDim.x) {
        const float a = (float)(scores[r] - scores[N + r / n]);
        reward[r] = a;
        s += a;
    }
    if (mean_out) {
        s = block_sum(s, scratch);
        if (threadIdx.x == 0) mean_out[0] = s / (float)N;
    }
}

}  // namespace

extern "C" {

int capmi_embed_fwd_pl(const int64_t *it, int it_stride, int64_t *it_save, const float *E, const float *mask, float *x,
                       int N, int Edim, int relu, void *x_planes, void *stream) {
    if (!it || !E || !x || N <= 0 || Edim <= 0 || it_stride < 1 || (x_planes && N > 64)) return CAPMI_EINVAL;
    hipLaunchKernelGGL(embed_fwd_kernel, dim3(grid_for((size_t)N * Edim)), dim3(256), 0, (hipStream_t)stream, it,
                       it_stride, it_save, E, mask, x, N, Edim, relu, static_cast<unsigned char *>(x_planes));
    CAPMI_CHECK_LAUNCH();
    return 0;
}

int capmi_embed_fwd(const int64_t *it, int it_stride, int64_t *it_save, const float *E, const float *mask, float *x,
                    int N, int Edim, int relu, void *stream) {
    return capmi_embed_fwd_pl(it, it_stride, it_save, E, mask, x, N, Edim, relu, nullptr, stream);
}

int capmi_embed_bwd(const int64_t *it, const float *dx, const float *x_saved, const float *mask, float *dE, int rows,
                    int Edim, int relu, void *stream) {
    if (!it || !dx || !dE || rows <= 0 || Edim <= 0 || (relu && !x_saved)) return CAPMI_EINVAL;
    static const int det = capmi::knob("CAPMI_EMBED_BWD_DET", 1);
    if (det && Edim % 4 == 0 && rows <= EBD_MAX_ROWS &&
        ((reinterpret_cast<uintptr_t>(dx) | reinterpret_cast<uintptr_t>(x_saved) | reinterpret_cast<uintptr_t>(mask) |
          reinterpret_cast<uintptr_t>(dE)) & 15) == 0) {
        // r6: ordered sums instead of atomicAdd -- the same bits on every run (embed_bwd_det.h)
        hipLaunchKernelGGL(embed_bwd_det_kernel, dim3(rows, (Edim + 255) / 256), dim3(EBD_THREADS), embed_bwd_det_lds(rows),
                           (hipStream_t)stream, it, rows, Edim, dE, EmbedGrad{dx, x_saved, mask, Edim, relu});
        CAPMI_CHECK_LAUNCH();
        return 0;
    }
    hipLaunchKernelGGL(embed_bwd_kernel, dim3(grid_for((size_t)rows * Edim)), dim3(256), 0, (hipStream_t)stream, it, dx,
                       x_saved, mask, dE, rows, Edim, relu);
    CAPMI_CHECK_LAUNCH();
    return 0;
}

int capmi_lstm_cell_fwd_pl(const float *partial, int splits, const float *b_ih, const float *b_hh, const float *row_bias,
                           int row_bias_div, const int32_t *row_bias_idx, const float *c_prev, float *h, float *c,
                           float *gates_act, const float *out_mask, float *h_drop, int N, int R, void *h_planes,
                           void *h_drop_planes, void *stream) {
    return capmi_lstm_cell_fwd_pl2(partial, splits, nullptr, 0, b_ih, b_hh, row_bias, row_bias_div, row_bias_idx, c_prev, h, c,
                                   gates_act, out_mask, h_drop, N, R, h_planes, h_drop_planes, stream);
}

int capmi_lstm_cell_fwd_pl2(const float *partial, int splits, const float *partial2, int splits2, const float *b_ih,
                            const float *b_hh, const float *row_bias, int row_bias_div, const int32_t *row_bias_idx,
                            const float *c_prev, float *h, float *c, float *gates_act, const float *out_mask, float *h_drop,
                            int N, int R, void *h_planes, void *h_drop_planes, void *stream) {
    if (!partial || splits < 1 || !c_prev || !h || !c || N <= 0 || R <= 0 || splits2 < 0 || (splits2 > 0 && !partial2))
        return CAPMI_EINVAL;
    if (splits2 == 0) partial2 = nullptr;
    if ((h_planes || h_drop_planes) && N > 64) return CAPMI_EINVAL;
    unsigned char *pl_h = static_cast<unsigned char *>(h_planes), *pl_hd = static_cast<unsigned char *>(h_drop_planes);
    const int rbd = row_bias_div > 0 ? row_bias_div : 1;
    // r4: the 16-byte kernel for every aligned call (it used to serve only the rollouts that also want planes; the teacher-forced
    // XE steps at 320 rows ran the scalar kernel: 11.5 us against ~7); same summation order per element, bit-identical
    if (R % 4 == 0 && aligned16(partial, partial2, b_ih, b_hh, row_bias, c_prev, h, c, gates_act, out_mask, h_drop)) {
        const int quads = N * (R / 4);
        hipLaunchKernelGGL(lstm_cell_fwd_vec_kernel, dim3((quads + 63) / 64), dim3(64), 0, (hipStream_t)stream, partial, splits,
                           b_ih, b_hh, row_bias, rbd, row_bias_idx, c_prev, h, c, gates_act, out_mask, h_drop, N, R, pl_h, pl_hd,
                           partial2, splits2);
        CAPMI_CHECK_LAUNCH();
        return 0;
    }
    hipLaunchKernelGGL(lstm_cell_fwd_kernel, dim3(grid_for((size_t)N * R)), dim3(256), 0, (hipStream_t)stream, partial,
                       splits, b_ih, b_hh, row_bias, rbd, row_bias_idx, c_prev, h, c, gates_act, out_mask, h_drop, N, R, pl_h,
                       pl_hd, partial2, splits2);
    CAPMI_CHECK_LAUNCH();
    return 0;
}

int capmi_lstm_cell_fwd(const float *partial, int splits, const float *b_ih, const float *b_hh, const float *row_bias,
                        int row_bias_div, const int32_t *row_bias_idx, const float *c_prev, float *h, float *c, float *gates_act,
                        const float *out_mask, float *h_drop, int N, int R, void *stream) {
    return capmi_lstm_cell_fwd_pl(partial, splits, b_ih, b_hh, row_bias, row_bias_div, row_bias_idx, c_prev, h, c, gates_act,
                                  out_mask, h_drop, N, R, nullptr, nullptr, stream);
}

int capmi_lstm_cell_bwd_partial(const float *dh_a, int ld_a, const float *dh_a_mask, const float *dh_b, int ld_b,
                                int b_splits, int64_t b_stride, const float *dh_c, int ld_c, int c_splits,
                                int64_t c_stride, const float *dc_next, const float *gates_act, const float *c_prev,
                                const float *c_new, float *d_gates, float *dc_prev, int N, int R, void *stream) {
    return capmi_lstm_cell_bwd_partial_pl(dh_a, ld_a, dh_a_mask, dh_b, ld_b, b_splits, b_stride, dh_c, ld_c, c_splits, c_stride,
                                          dc_next, gates_act, c_prev, c_new, d_gates, dc_prev, N, R, nullptr, stream);
}

int capmi_lstm_cell_bwd_partial_pl(const float *dh_a, int ld_a, const float *dh_a_mask, const float *dh_b, int ld_b,
                                   int b_splits, int64_t b_stride, const float *dh_c, int ld_c, int c_splits,
                                   int64_t c_stride, const float *dc_next, const float *gates_act, const float *c_prev,
                                   const float *c_new, float *d_gates, float *dc_prev, int N, int R, void *d_gates_planes,
                                   void *stream) {
    if (!gates_act || !c_prev || !c_new || !d_gates || !dc_prev || N <= 0 || R <= 0) return CAPMI_EINVAL;
    if ((dh_b && b_splits < 1) || (dh_c && c_splits < 1) || (d_gates_planes && N > 64)) return CAPMI_EINVAL;
    unsigned char *pl_dg = static_cast<unsigned char *>(d_gates_planes);
    if (R % 4 == 0 && ld_a % 4 == 0 && ld_b % 4 == 0 && ld_c % 4 == 0 && b_stride % 4 == 0 && c_stride % 4 == 0 &&
        aligned16(dh_a, dh_a_mask, dh_b, dh_c, dc_next, gates_act, c_prev, c_new, d_gates, dc_prev)) {
        const int quads = N * (R / 4);
        hipLaunchKernelGGL(lstm_cell_bwd_vec_kernel, dim3((quads + 63) / 64), dim3(64), 0, (hipStream_t)stream, dh_a, ld_a,
                           dh_a_mask, dh_b, ld_b, b_splits, (size_t)b_stride, dh_c, ld_c, c_splits, (size_t)c_stride, dc_next,
                           gates_act, c_prev, c_new, d_gates, dc_prev, N, R, pl_dg);
        CAPMI_CHECK_LAUNCH();
        return 0;
    }
    hipLaunchKernelGGL(lstm_cell_bwd_kernel, dim3(grid_for((size_t)N * R)), dim3(256), 0, (hipStream_t)stream, dh_a,
                       ld_a, dh_a_mask, dh_b, ld_b, b_splits, (size_t)b_stride, dh_c, ld_c, c_splits, (size_t)c_stride,
                       dc_next, gates_act, c_prev, c_new, d_gates, dc_prev, N, R, pl_dg);
    CAPMI_CHECK_LAUNCH();
    return 0;
}

int capmi_lstm_cell_bwd(const float *dh_a, int ld_a, const float *dh_a_mask, const float *dh_b, int ld_b,
                        const float *dh_c, int ld_c, const float *dc_next, const float *gates_act,
                        const float *c_prev, const float *c_new, float *d_gates, float *dc_prev, int N, int R,
                        void *stream) {
    return capmi_lstm_cell_bwd_partial(dh_a, ld_a, dh_a_mask, dh_b, ld_b, 1, 0, dh_c, ld_c, 1, 0, dc_next, gates_act,
                                       c_prev, c_new, d_gates, dc_prev, N, R, stream);
}

int capmi_dropout_mask(float *mask, int64_t count, float p, uint64_t seed, uint64_t offset, void *stream) {
    if (!mask || count <= 0 || p < 0.f || p >= 1.f) return CAPMI_EINVAL;
    hipLaunchKernelGGL(dropout_mask_kernel, dim3(grid_for((size_t)(count + 3) / 4)), dim3(256), 0, (hipStream_t)stream,
                       mask, (size_t)count, p, seed, offset, capmi::rng_epoch());
    CAPMI_CHECK_LAUNCH();
    return 0;
}

int capmi_dropout_masks(const capmi_mask_desc *descs, int n, float p, uint64_t seed, void *stream) {
    if (!descs || n < 1 || n > CAPMI_MAX_MASKS || p < 0.f || p >= 1.f) return CAPMI_EINVAL;
    MaskSegs sg{};
    sg.n = n;
    size_t most = 0;
    for (int i = 0; i < n; ++i) {
        const capmi_mask_desc &d = descs[i];
        if (!d.mask || d.count <= 0) return CAPMI_EINVAL;
        sg.mask[i] = d.mask; sg.count[i] = (unsigned long long)d.count; sg.offset[i] = d.offset;
        sg.row_len[i] = d.row_len > 0 ? d.row_len : 1;
        sg.rows[i] = d.rows > 0 ? d.rows : 1;
        sg.keep_from[i] = (d.keep_from >= 0 && d.rows > 0) ? d.keep_from : sg.rows[i];
        if ((size_t)d.count > most) most = (size_t)d.count;
    }
    hipLaunchKernelGGL(dropout_masks_kernel, dim3(grid_for((most + 3) / 4)), dim3(256), 0, (hipStream_t)stream, sg, p, seed, capmi::rng_epoch());
    CAPMI_CHECK_LAUNCH();
    return 0;
}

int capmi_rollout_init(float *h0, float *c0, float *h1, float *c1, int64_t count, int64_t *it, uint8_t *unfinished, int N,
                       void *stream) {
    if (!h0 || !c0 || !it || !unfinished || count < N || N <= 0) return CAPMI_EINVAL;
    hipLaunchKernelGGL(rollout_init_kernel, dim3(grid_for((size_t)count)), dim3(256), 0, (hipStream_t)stream, h0, c0, h1, c1,
                       (size_t)count, it, unfinished, N);
    CAPMI_CHECK_LAUNCH();
    return 0;
}

int capmi_colsum(const float *in, int rows, int cols, int ld, float *out, int accumulate, void *stream) {
    if (!in || !out || rows <= 0 || cols <= 0) return CAPMI_EINVAL;
    const int vec = ((reinterpret_cast<uintptr_t>(in) & 15) == 0 && ld % 4 == 0 && cols % 4 == 0) ? 1 : 0;
    const int cblocks = (cols + 4 * CS_Q - 1) / (4 * CS_Q);
    int rsplit = 1;
    if (cblocks < 64 && rows >= 4 * CS_R * 4) {            // narrow and tall: spread the rows over more workgroups
        rsplit = 128 / cblocks;
        const int max_by_rows = rows / (CS_R * 4);
        if (rsplit > max_by_rows) rsplit = max_by_rows;
        if (rsplit < 1) rsplit = 1;
    }
    if (rsplit > 1 && !accumulate) {
        hipError_t e = hipMemsetAsync(out, 0, (size_t)cols * sizeof(float), (hipStream_t)stream);
        if (e != hipSuccess) return (int)e;
    }
    hipLaunchKernelGGL(colsum_kernel, dim3(cblocks, rsplit), dim3(CS_Q * CS_R), 0, (hipStream_t)stream, in, rows, cols, ld, out,
                       accumulate, vec, rsplit);
    CAPMI_CHECK_LAUNCH();
    return 0;
}

int capmi_colsum_batch(const capmi_colsum_item *items, int n_items, void *stream) {
    if (!items || n_items <= 0) return CAPMI_EINVAL;
    // the column counts live on the device: 160 column blocks per item cover a vocabulary-wide bias (9 488 columns = 149 blocks) in
    // one round; workgroups past an item's last block leave at once.  (r4: with 32 the logit bias of a Transformer XE step --
    // 255 MB -- was walked by 32 workgroups while the chip idled: 520 us for the launch)
    hipLaunchKernelGGL(colsum_batch_kernel, dim3(160, n_items), dim3(CB_Q * CB_R), 0, (hipStream_t)stream, items);
    CAPMI_CHECK_LAUNCH();
    return 0;
}

int capmi_colsum_batch_args(const capmi_colsum_item *host_items, int n_items, void *stream) {
    if (!host_items || n_items <= 0 || n_items > CAPMI_COLSUM_ARGS_MAX) return CAPMI_EINVAL;
    ColsumArgs a{};
    int max_cb = 1;
    for (int i = 0; i < n_items; ++i) {
        a.items[i] = host_items[i];
        if (!a.items[i].in || !a.items[i].out || a.items[i].rows <= 0 || a.items[i].cols <= 0) return CAPMI_EINVAL;
        const int cb = (a.items[i].cols + 4 * CB_Q - 1) / (4 * CB_Q);
        if (cb > max_cb) max_cb = cb;
    }
    hipLaunchKernelGGL(colsum_batch_args_kernel, dim3(max_cb < 64 ? max_cb : 64, n_items), dim3(CB_Q * CB_R), 0, (hipStream_t)stream, a);
    CAPMI_CHECK_LAUNCH();
    return 0;
}

int capmi_group_rowsum(const float *in, int T, int64_t slab, int groups, int group, int cols, float *out,
                       void *stream) {
    if (!in || !out || T <= 0 || groups <= 0 || group <= 0 || cols <= 0) return CAPMI_EINVAL;
    if (cols % 4 == 0 && slab % 4 == 0 && ((reinterpret_cast<uintptr_t>(in) | reinterpret_cast<uintptr_t>(out)) & 15) == 0) {
        const size_t nq = (size_t)groups * (cols / 4);
        hipLaunchKernelGGL(group_rowsum_v4_kernel, dim3((unsigned)((nq + 63) / 64)), dim3(256), 0, (hipStream_t)stream, in, T,
                           (size_t)slab, groups, group, cols, out);
    } else {
        hipLaunchKernelGGL(group_rowsum_kernel, dim3(grid_for((size_t)groups * cols)), dim3(256), 0, (hipStream_t)stream, in,
                           T, (size_t)slab, groups, group, cols, out);
    }
    CAPMI_CHECK_LAUNCH();
    return 0;
}

int capmi_relu_mask_bwd(const float *dy, const float *y_ref, const float *mask, float *dx, int64_t count,
                        void *stream) {
    if (!dy || !dx || count <= 0) return CAPMI_EINVAL;
    hipLaunchKernelGGL(relu_mask_bwd_kernel, dim3(grid_for((size_t)count)), dim3(256), 0, (hipStream_t)stream, dy, y_ref,
                       mask, dx, (size_t)count);
    CAPMI_CHECK_LAUNCH();
    return 0;
}

int capmi_relu_scale_bwd(const float *dy, const float *y_ref, float scale, float *dx, int64_t count, void *stream) {
    if (!dy || !y_ref || !dx || count <= 0 || count % 4 != 0) return CAPMI_EINVAL;
    if ((reinterpret_cast<uintptr_t>(dy) | reinterpret_cast<uintptr_t>(y_ref) | reinterpret_cast<uintptr_t>(dx)) & 15) return CAPMI_EINVAL;
    hipLaunchKernelGGL(relu_scale_bwd_kernel, dim3(grid_for((size_t)count / 4)), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const f32x4 *>(dy), reinterpret_cast<const f32x4 *>(y_ref), scale, reinterpret_cast<f32x4 *>(dx),
                       (size_t)count / 4);
    CAPMI_CHECK_LAUNCH();
    return 0;
}

static int adam_launch(float *p, const float *g, float *m, float *v, int64_t count, float lr, float beta1, float beta2, float eps,
                       float weight_decay, float clip, float grad_scale, float bc1, float bc2_sqrt, const capmi_step_state *dyn,
                       void *stream) {
    if ((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(m) |
         reinterpret_cast<uintptr_t>(v)) & 15)
        return CAPMI_EINVAL;
    if (count % 4 == 0) {                           // the flat buffers are padded to 64 floats (flat.py): always this branch on the path
        const size_t quads = (size_t)count / 4;
        hipLaunchKernelGGL(adam2_kernel<true>, dim3(grid_for(quads / 2 + 1)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, quads, lr,
                           beta1, beta2, eps, weight_decay, clip, grad_scale, bc1, bc2_sqrt, dyn);
    } else {                                        // any other count: plain quads + scalar tail
        hipLaunchKernelGGL(adam_kernel, dim3(grid_for((size_t)count / 4 + 4)), dim3(256), 0, (hipStream_t)stream, p, g, m, v,
                           (size_t)count, lr, beta1, beta2, eps, weight_decay, clip, grad_scale, bc1, bc2_sqrt, dyn);
    }
    CAPMI_CHECK_LAUNCH();
    return 0;
}

int capmi_adam_step(float *p, const float *g, float *m, float *v, int64_t count, float lr, float beta1, float beta2,
                    float eps, float weight_decay, float clip, float grad_scale, int step, void *stream) {
    if (!p || !g || !m || !v || count <= 0 || step < 1) return CAPMI_EINVAL;
    const double bc1 = 1.0 - pow((double)beta1, step);
    const double bc2 = 1.0 - pow((double)beta2, step);
    return adam_launch(p, g, m, v, count, lr, beta1, beta2, eps, weight_decay, clip, grad_scale, (float)bc1, (float)sqrt(bc2), nullptr,
                       stream);
}

int capmi_adam_step_dyn(float *p, const float *g, float *m, float *v, int64_t count, const capmi_step_state *state, float beta1,
                        float beta2, float eps, float weight_decay, float clip, float grad_scale, void *stream) {
    if (!p || !g || !m || !v || count <= 0 || !state) return CAPMI_EINVAL;
    return adam_launch(p, g, m, v, count, 0.f, beta1, beta2, eps, weight_decay, clip, grad_scale, 1.f, 1.f, state, stream);
}

int capmi_upload_async(void *dst, const void *src_pinned, int64_t bytes, void *stream) {
    if (!dst || !src_pinned || bytes <= 0) return CAPMI_EINVAL;
    const hipError_t e = hipMemcpyAsync(dst, src_pinned, (size_t)bytes, hipMemcpyHostToDevice, (hipStream_t)stream);
    return e == hipSuccess ? 0 : (int)e;
}

int capmi_step_advance(capmi_step_state *state, float beta1, float beta2, void *stream) {
    if (!state) return CAPMI_EINVAL;
    hipLaunchKernelGGL(step_advance_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, state, beta1, beta2);
    CAPMI_CHECK_LAUNCH();
    return 0;
}

int capmi_step_set_lr(capmi_step_state *state, float lr, void *stream) {
    if (!state) return CAPMI_EINVAL;
    hipLaunchKernelGGL(step_set_lr_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, state, lr);
    CAPMI_CHECK_LAUNCH();
    return 0;
}

int capmi_rng_bind_epoch(const uint64_t *epoch, const uint64_t **prev) {
    const uint64_t *old = g_rng_epoch.exchange(epoch);
    if (prev) *prev = old;
    return 0;
}

int capmi_scst_advantage_mean(const double *scores, int N, int n, float *reward, float *mean_out, void *stream) {
    if (!scores || !reward || N <= 0 || n <= 0) return CAPMI_EINVAL;
    hipLaunchKernelGGL(scst_advantage_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, scores, N, n, reward, mean_out);
    CAPMI_CHECK_LAUNCH();
    return 0;
}

int capmi_scst_advantage(const double *scores, int N, int n, float *reward, void *stream) {
    return capmi_scst_advantage_mean(scores, N, n, reward, nullptr, stream);
}

}  // extern "C"
