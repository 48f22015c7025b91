// fp32 MFMA GEMM for gfx950 (v_mfma_f32_32x32x2_f32: f32 in, f32 accumulate, bit-equal to an fmaf
// chain -- the only matrix path on CDNA4 that keeps the reference's fp32 numerics; there is no
// xf32/TF32).  Replaces the addmm/mm behind nn.Linear / nn.LSTMCell and their backward on the
// caption-decoding hot path (see include/capmi.h for the reference call sites).
//
// Design (MI355X-first, not a CUDA tiling):
//  * 256-thread workgroups = 4 wave64s; each wave owns a (TM x TN) grid of 32x32 MFMA tiles.
//  * operands are staged k-major in LDS (As[k][m], Bs[k][n]) so that an MFMA operand fetch is one
//    conflict-free ds_read_b32 of 32 consecutive floats per half-wave (lane l reads k = l>>5,
//    m = l&31): no swizzle needed, any leading dimension.
//  * K-contiguous sources (activations [M][K], nn.Linear weights [N][K]) are read from HBM as full
//    128-byte lines (8 lanes x 16 B per row) and transposed on the LDS write (row pitch BM+1 makes
//    the 4 scattered ds_write_b32 conflict-free); M/N-contiguous sources ([K][M] for dY^T, [K][N]
//    for W in dX = dY W) are written with ds_write_b128 (pitch BM+4).
//  * register prefetch of tile t+1 is in flight while tile t runs on the matrix pipe.
//  * the decode GEMMs are skinny (M = 10..64 rows against 12-38 MB of weights), i.e. HBM-bound
//    weight streaming: split-K spreads one weight matrix over >= 2 workgroups per CU; partial sums
//    go to a workspace that the *consumer* kernel (LSTM cell, bias/ReLU epilogue) reduces, so the
//    gate pre-activations never make a round trip of their own.
//  * multi-segment K loop: [h_lang | xt | h_att] x [W_ih slices | W_hh] are walked in place -- the
//    reference's torch.cat (AttModel.py:626,632) and repeat_tensors (a_row_div) copies disappear.
#include "capmi_common.h"
#include "profile.h"
#include "../../../include/capmi.h"
#include "gemm_common.h"
#include <atomic>
#include <stdio.h>
#include <stdlib.h>
#include <hip/hip_ext.h>

using namespace capmi_gemm;

namespace {

// ---- global -> registers ---------------------------------------------------------------------
// KC = true : source stored [rows][K] (K contiguous).  thread -> (row, 4 consecutive k)
// KC = false: source stored [K][rows] (rows contiguous). thread -> (k, 4 consecutive rows)
template <int ROWS, bool KC>
__device__ __forceinline__ void g2r(f32x4 (&r)[ROWS * BK / 4 / NT], const float *__restrict__ src, int ld,
                                    int row0, int nrows, int k0, int K, int row_div, int vec) {
    constexpr int NV = ROWS * BK / 4 / NT;
#pragma unroll
    for (int p = 0; p < NV; ++p) {
        const int idx = p * NT + threadIdx.x;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (KC) {
            const int rr = idx / (BK / 4);
            const int kq = (idx % (BK / 4)) * 4;
            const int row = row0 + rr;
            const int k = k0 + kq;
            if (row < nrows && k < K) {
                const float *ptr = src + (size_t)(row / row_div) * ld + k;
                if (vec && k + 3 < K) {
                    v = *reinterpret_cast<const f32x4 *>(ptr);
                } else {
                    v[0] = ptr[0];
                    if (k + 1 < K) v[1] = ptr[1];
                    if (k + 2 < K) v[2] = ptr[2];
                    if (k + 3 < K) v[3] = ptr[3];
                }
            }
        } else {
            const int kk = idx / (ROWS / 4);
            const int mq = (idx % (ROWS / 4)) * 4;
            const int k = k0 + kk;
            const int row = row0 + mq;
            if (k < K && row < nrows) {
                const float *ptr = src + (size_t)k * ld + row;
                if (vec && row + 3 < nrows) {
                    v = *reinterpret_cast<const f32x4 *>(ptr);
                } else {
                    v[0] = ptr[0];
                    if (row + 1 < nrows) v[1] = ptr[1];
                    if (row + 2 < nrows) v[2] = ptr[2];
                    if (row + 3 < nrows) v[3] = ptr[3];
                }
            }
        }
        r[p] = v;
    }
}

// LDS images.  K-contiguous sources keep their row-major shape [rows][BK+4] (pitch 36 floats: 16-byte
// aligned rows, and 36 = 4*9 makes the 16 rows of a ds_read_b128 lane group hit 16 distinct 4-bank slots
// => conflict free); the MFMA k index is free to permute (a sum), so a lane fetches FOUR consecutive
// k values of its row with one ds_read_b128 and feeds four MFMAs from it.  M/N-contiguous sources are
// stored k-major [BK][rows+4] and read with ds_read_b32 at the same (permuted) k.
template <int ROWS, bool KC>
struct Pitch {
    static constexpr int value = KC ? BK + 4 : ROWS + 4;
    static constexpr int size = KC ? ROWS * (BK + 4) : BK * (ROWS + 4);
};

template <int ROWS, bool KC>
__device__ __forceinline__ void r2s(const f32x4 (&r)[ROWS * BK / 4 / NT], float *dst) {
    constexpr int NV = ROWS * BK / 4 / NT;
    constexpr int LD = Pitch<ROWS, KC>::value;
#pragma unroll
    for (int p = 0; p < NV; ++p) {
        const int idx = p * NT + threadIdx.x;
        if (KC) {
            const int rr = idx / (BK / 4);
            const int kq = (idx % (BK / 4)) * 4;
            *reinterpret_cast<f32x4 *>(dst + rr * LD + kq) = r[p];
        } else {
            const int kk = idx / (ROWS / 4);
            const int mq = (idx % (ROWS / 4)) * 4;
            *reinterpret_cast<f32x4 *>(dst + kk * LD + mq) = r[p];
        }
    }
}

template <int BM, int BN, int WM, int WN, bool AKC, bool BKC, int PF>
__global__ __launch_bounds__(NT) void gemm_kernel(const KArgs a) {
    constexpr int TM = BM / WM / 32;
    constexpr int TN = BN / WN / 32;
    constexpr int LDA = Pitch<BM, AKC>::value;
    constexpr int LDB = Pitch<BN, BKC>::value;
    static_assert(WM * WN == 4, "4 waves per workgroup");
    __shared__ __attribute__((aligned(16))) float smem[Pitch<BM, AKC>::size + Pitch<BN, BKC>::size];
    float *As = smem;
    float *Bs = smem + Pitch<BM, AKC>::size;

    const int m0 = blockIdx.y * BM;
    const int n0 = blockIdx.x * BN;
    const int z = blockIdx.z;
    // contiguous, balanced range of flat K tiles for this split
    const int t_begin = (int)(((long long)a.tiles_total * z) / a.splits);
    const int t_end = (int)(((long long)a.tiles_total * (z + 1)) / a.splits);

    const int lane = threadIdx.x & 63;
    const int wid = threadIdx.x >> 6;
    const int wm0 = (wid / WN) * (TM * 32);
    const int wn0 = (wid % WN) * (TN * 32);
    const int l31 = lane & 31;
    const int khalf = lane >> 5;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // PF register sets keep PF K-tiles in flight from HBM while one tile is consumed from LDS: at decode
    // sizes the loop is HBM-latency bound (one 16 KB tile per ~2 us round trip per workgroup otherwise).
    f32x4 ra[PF][BM * BK / 4 / NT];
    f32x4 rb[PF][BN * BK / 4 / NT];

#pragma unroll
    for (int u = 0; u < PF; ++u) {
        if (t_begin + u < t_end) {
            int s, k0;
            locate(a, t_begin + u, s, k0);
            g2r<BM, AKC>(ra[u], a.seg[s].A, a.seg[s].lda, m0, a.M, k0, a.seg[s].K, AKC ? a.seg[s].a_row_div : 1, a.seg[s].vecA);
            g2r<BN, BKC>(rb[u], a.seg[s].B, a.seg[s].ldb, n0, a.N, k0, a.seg[s].K, 1, a.seg[s].vecB);
        }
    }
    for (int t0 = t_begin; t0 < t_end; t0 += PF) {
#pragma unroll
        for (int u = 0; u < PF; ++u) {
            const int t = t0 + u;
            if (t < t_end) {
                __syncthreads();   // previous tile fully consumed
                if (!(a.ablate & 4)) {
                    r2s<BM, AKC>(ra[u], As);
                    r2s<BN, BKC>(rb[u], Bs);
                }
                __syncthreads();
                if (t + PF < t_end && !(a.ablate & 2)) {
                    int s, k0;
                    locate(a, t + PF, s, k0);
                    g2r<BM, AKC>(ra[u], a.seg[s].A, a.seg[s].lda, m0, a.M, k0, a.seg[s].K, AKC ? a.seg[s].a_row_div : 1, a.seg[s].vecA);
                    g2r<BN, BKC>(rb[u], a.seg[s].B, a.seg[s].ldb, n0, a.N, k0, a.seg[s].K, 1, a.seg[s].vecB);
                }
                if (a.ablate & 1) continue;
#pragma unroll
                for (int q = 0; q < BK / 8; ++q) {
                    // lane (l31, khalf) supplies k = 8q + 4*khalf + e to MFMA e = 0..3 of this group
                    const int kb = 8 * q + 4 * khalf;
                    f32x4 av[TM], bv[TN];
#pragma unroll
                    for (int i = 0; i < TM; ++i) {
                        const int row = wm0 + 32 * i + l31;
                        if (AKC) av[i] = *reinterpret_cast<const f32x4 *>(As + row * LDA + kb);
                        else av[i] = f32x4{As[(kb + 0) * LDA + row], As[(kb + 1) * LDA + row], As[(kb + 2) * LDA + row],
                                           As[(kb + 3) * LDA + row]};
                    }
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        const int col = wn0 + 32 * j + l31;
                        if (BKC) bv[j] = *reinterpret_cast<const f32x4 *>(Bs + col * LDB + kb);
                        else bv[j] = f32x4{Bs[(kb + 0) * LDB + col], Bs[(kb + 1) * LDB + col], Bs[(kb + 2) * LDB + col],
                                           Bs[(kb + 3) * LDB + col]};
                    }
#pragma unroll
                    for (int e = 0; e < 4; ++e)
#pragma unroll
                        for (int i = 0; i < TM; ++i)
#pragma unroll
                            for (int j = 0; j < TN; ++j)
                                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i][e], bv[j][e], acc[i][j], 0, 0, 0);
                }
            }
        }
    }

    // ---- epilogue: C/D layout of 32x32 MFMA: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
    const bool to_partial = a.to_partial != 0;
    const size_t MN = (size_t)a.M * a.N;
    float *out = to_partial ? a.partial + (size_t)z * MN : a.C;
    const int ldo = to_partial ? a.N : a.ldc;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int col = n0 + wn0 + 32 * j + l31;
            if (col >= a.N) continue;
            float cb = 0.f;
            if (!to_partial) {
                if (a.bias) cb += a.bias[col];
                if (a.bias2) cb += a.bias2[col];
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm0 + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * khalf;
                if (row >= a.M) continue;
                float v = acc[i][j][r];
                if (!to_partial) {
                    v += cb;
                    if (a.row_bias) v += a.row_bias[(size_t)(row / a.row_bias_div) * a.N + col];
                    if (a.relu) v = fmaxf(v, 0.f);
                    if (a.mul_mask) v *= a.mul_mask[(size_t)row * a.N + col];
                    if (a.accumulate) v += a.addend[(size_t)row * a.ldc + col];      // (not to_partial: out == C)
                    out[(size_t)row * ldo + col] = v;
                } else if (a.self_reduce) {
                    // write-through (sc1) slab store: visible to the last-arriving workgroup without a release fence
                    __hip_atomic_store(out + (size_t)row * ldo + col, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                } else {
                    out[(size_t)row * ldo + col] = v;
                }
            }
        }
    }
    if (!(to_partial && a.self_reduce)) return;
    // in-launch split-K reduction (cdna_hip_programming.md G16): drain, ticket, last arriver acquires + reduces
    __shared__ int s_last;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int *ticket = a.counters + (blockIdx.y * gridDim.x + blockIdx.x);
    if (threadIdx.x == 0) {
        const int t = __hip_atomic_fetch_add(ticket, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_last = (t == a.splits - 1);
    }
    __syncthreads();
    if (!s_last) return;
    if (threadIdx.x == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    __syncthreads();
    for (int e = threadIdx.x; e < BM * BN; e += NT) {
        const int rr = e / BN, cc = e % BN;
        const int row = m0 + rr, col = n0 + cc;
        if (row < a.M && col < a.N) {
            const float *pp = a.partial + (size_t)row * a.N + col;
            float v = 0.f;
            for (int s0 = 0; s0 < a.splits; s0 += 8) {
                float tv[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) tv[u] = (s0 + u < a.splits) ? pp[(size_t)(s0 + u) * MN] : 0.f;
#pragma unroll
                for (int u = 0; u < 8; ++u) v += tv[u];
            }
            if (a.bias) v += a.bias[col];
            if (a.bias2) v += a.bias2[col];
            if (a.row_bias) v += a.row_bias[(size_t)(row / a.row_bias_div) * a.N + col];
            if (a.relu) v = fmaxf(v, 0.f);
            if (a.mul_mask) v *= a.mul_mask[(size_t)row * a.N + col];
            if (a.accumulate) v += a.addend[(size_t)row * a.ldc + col];
            a.C[(size_t)row * a.ldc + col] = v;
        }
    }
    if (threadIdx.x == 0) __hip_atomic_store(ticket, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // re-arm
}

// VEC: 4 outputs of a row per thread on 16-byte accesses (N, ldc % 4 == 0 and every operand 16-byte aligned: the host checks); the
// slab sums keep the scalar kernel's order per element (8 slabs requested at a time), so both give the same bits
template <bool VEC>
__global__ void splitk_reduce_kernel(const float *__restrict__ partial, int splits, float *__restrict__ C, int ldc,
                                     int M, int N, const float *bias, const float *bias2, const float *row_bias,
                                     int row_bias_div, const float *mul_mask, int relu, int accumulate,
                                     const float *__restrict__ addend) {
    const size_t total = (size_t)M * N;
    if (VEC) {
        const unsigned quads = (unsigned)(total >> 2), nq = (unsigned)N >> 2;          // (host: total < 2^32)
        for (unsigned q = blockIdx.x * blockDim.x + threadIdx.x; q < quads; q += gridDim.x * blockDim.x) {
            const unsigned row = q / nq, col = (q - row * nq) * 4;
            const size_t i = (size_t)q * 4;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            for (int s0 = 0; s0 < splits; s0 += 8) {
                f32x4 tv[8];
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    tv[u] = (s0 + u < splits) ? *reinterpret_cast<const f32x4 *>(partial + (size_t)(s0 + u) * total + i) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int u = 0; u < 8; ++u) v += tv[u];
            }
            if (bias) v += *reinterpret_cast<const f32x4 *>(bias + col);
            if (bias2) v += *reinterpret_cast<const f32x4 *>(bias2 + col);
            if (row_bias) v += *reinterpret_cast<const f32x4 *>(row_bias + (size_t)(row / row_bias_div) * N + col);
            if (relu) { v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f); }
            if (mul_mask) v *= *reinterpret_cast<const f32x4 *>(mul_mask + i);
            if (accumulate) v += *reinterpret_cast<const f32x4 *>(addend + (size_t)row * ldc + col);
            *reinterpret_cast<f32x4 *>(C + (size_t)row * ldc + col) = v;
        }
        return;
    }
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int row = (int)(i / N), col = (int)(i % N);
        float v = 0.f;
        for (int s0 = 0; s0 < splits; s0 += 8) {   // 8 independent loads in flight (a rolled loop serialises the latencies)
            float tv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) tv[u] = (s0 + u < splits) ? partial[(size_t)(s0 + u) * total + i] : 0.f;
#pragma unroll
            for (int u = 0; u < 8; ++u) v += tv[u];
        }
        if (bias) v += bias[col];
        if (bias2) v += bias2[col];
        if (row_bias) v += row_bias[(size_t)(row / row_bias_div) * N + col];
        if (relu) v = fmaxf(v, 0.f);
        if (mul_mask) v *= mul_mask[i];
        float *o = C + (size_t)row * ldc + col;
        if (accumulate) v += addend[(size_t)row * ldc + col];
        *o = v;
    }
}

// blockIdx.y = item; its M x N outputs are walked by the item's blockIdx.x workgroups, 16 bytes per thread when the shapes allow
__global__ __launch_bounds__(256) void splitk_reduce_batch_kernel(const capmi_reduce_item *__restrict__ items) {
    const capmi_reduce_item it = items[blockIdx.y];
    const size_t total = (size_t)it.M * it.N;
    const bool vec = it.N % 4 == 0 && it.ldc % 4 == 0 && ((reinterpret_cast<uintptr_t>(it.partial) | reinterpret_cast<uintptr_t>(it.C)) & 15) == 0 &&
                     (!it.bias || (reinterpret_cast<uintptr_t>(it.bias) & 15) == 0);
    if (vec) {
        const size_t quads = total / 4;
        for (size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x; q < quads; q += (size_t)gridDim.x * blockDim.x) {
            const size_t i = q * 4;
            const int row = (int)(i / it.N), col = (int)(i % it.N);
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            for (int s0 = 0; s0 < it.splits; s0 += 4) {        // 4 independent 16-byte loads in flight
                f32x4 tv[4];
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    tv[u] = (s0 + u < it.splits) ? *reinterpret_cast<const f32x4 *>(it.partial + (size_t)(s0 + u) * total + i)
                                                 : f32x4{0.f, 0.f, 0.f, 0.f};
                v += (tv[0] + tv[1]) + (tv[2] + tv[3]);
            }
            if (it.bias) v += *reinterpret_cast<const f32x4 *>(it.bias + col);
            f32x4 *o = reinterpret_cast<f32x4 *>(it.C + (size_t)row * it.ldc + col);
            if (it.accumulate) v += *o;
            *o = v;
        }
        return;
    }
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int row = (int)(i / it.N), col = (int)(i % it.N);
        float v = 0.f;
        for (int s0 = 0; s0 < it.splits; ++s0) v += it.partial[(size_t)s0 * total + i];
        if (it.bias) v += it.bias[col];
        float *o = it.C + (size_t)row * it.ldc + col;
        if (it.accumulate) v += *o;
        *o = v;
    }
}

struct ProfInfo {
    int cls;
    double bytes, flops;
};
template <typename K>
void launch_one(K kernel, dim3 grid, hipStream_t st, const KArgs &a, const ProfInfo &pi) {
    hipEvent_t e0, e1;
    if (capmi_prof::take_events(pi.cls, &e0, &e1, pi.bytes, pi.flops))
        hipExtLaunchKernelGGL(kernel, grid, dim3(NT), 0, st, e0, e1, 0, a);
    else
        hipLaunchKernelGGL(kernel, grid, dim3(NT), 0, st, a);
}
template <int BM, int BN, int WM, int WN, int PF>
int launch_cfg(const KArgs &a, int al, int bl, dim3 grid, hipStream_t st, const ProfInfo &pi) {
    if (al == 0 && bl == 0) launch_one(gemm_kernel<BM, BN, WM, WN, true, true, PF>, grid, st, a, pi);
    else if (al == 0 && bl == 1) launch_one(gemm_kernel<BM, BN, WM, WN, true, false, PF>, grid, st, a, pi);
    else if (al == 1 && bl == 1) launch_one(gemm_kernel<BM, BN, WM, WN, false, false, PF>, grid, st, a, pi);
    else launch_one(gemm_kernel<BM, BN, WM, WN, false, true, PF>, grid, st, a, pi);
    CAPMI_CHECK_LAUNCH();
    return 0;
}

inline bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

static int splitk_reduce_addend(const float *partial, int splits, float *C, int ldc, int M, int N, const float *bias, const float *bias2,
                                const float *row_bias, int row_bias_div, const float *mul_mask, int relu, int accumulate,
                                const float *addend, void *stream) {
    if (!partial || !C || splits < 1 || M <= 0 || N <= 0) return CAPMI_EINVAL;
    const size_t total = (size_t)M * N;
    const float *ad = addend ? addend : C;
    const bool vec = N % 4 == 0 && ldc % 4 == 0 && total < (1ull << 32) &&
                     ((reinterpret_cast<uintptr_t>(partial) | reinterpret_cast<uintptr_t>(C) | reinterpret_cast<uintptr_t>(bias) |
                       reinterpret_cast<uintptr_t>(bias2) | reinterpret_cast<uintptr_t>(row_bias) | reinterpret_cast<uintptr_t>(mul_mask) |
                       reinterpret_cast<uintptr_t>(ad)) & 15) == 0;
    const size_t work = vec ? total / 4 : total;
    int blocks = (int)((work + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    if (vec)
        hipLaunchKernelGGL(splitk_reduce_kernel<true>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, partial, splits, C, ldc, M, N, bias,
                           bias2, row_bias, row_bias_div > 0 ? row_bias_div : 1, mul_mask, relu, accumulate, ad);
    else
        hipLaunchKernelGGL(splitk_reduce_kernel<false>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, partial, splits, C, ldc, M, N, bias,
                           bias2, row_bias, row_bias_div > 0 ? row_bias_div : 1, mul_mask, relu, accumulate, ad);
    CAPMI_CHECK_LAUNCH();
    return 0;
}

extern "C" int capmi_splitk_reduce(const float *partial, int splits, float *C, int ldc, int M, int N,
                                   const float *bias, const float *bias2, const float *row_bias, int row_bias_div,
                                   const float *mul_mask, int relu, int accumulate, void *stream) {
    return splitk_reduce_addend(partial, splits, C, ldc, M, N, bias, bias2, row_bias, row_bias_div, mul_mask, relu, accumulate, nullptr,
                                stream);
}

extern "C" int capmi_splitk_reduce_batch(const capmi_reduce_item *items, int n_items, void *stream) {
    if (!items || n_items <= 0) return CAPMI_EINVAL;
    hipLaunchKernelGGL(splitk_reduce_batch_kernel, dim3(64, n_items), dim3(256), 0, (hipStream_t)stream, items);
    CAPMI_CHECK_LAUNCH();
    return 0;
}

// r5: a wide (256 x 128) fat-GEMM workgroup owns its CU; planned for deferred-reduction GEMMs only when the caller says that no
// other stream runs beside them (capmi_gemm_set_policy)
static std::atomic<int> g_wide_deferred{0};

extern "C" int capmi_gemm_set_policy(int allow_wide_deferred) {
    return g_wide_deferred.exchange(allow_wide_deferred ? 1 : 0);
}

extern "C" int capmi_gemm_f32(capmi_gemm_desc *d, void *stream) {
    if (!d || d->nseg < 1 || d->nseg > CAPMI_MAX_SEG || d->M <= 0 || d->N <= 0 || !d->C) return CAPMI_EINVAL;
    if (d->a_layout < 0 || d->a_layout > 1 || d->b_layout < 0 || d->b_layout > 1) return CAPMI_EINVAL;
    KArgs a{};
    a.nseg = d->nseg;
    int tiles = 0;
    for (int s = 0; s < d->nseg; ++s) {
        const capmi_gemm_seg &g = d->seg[s];
        if (g.K <= 0) return CAPMI_EINVAL;
        if (!g.A || !g.B) return CAPMI_EINVAL;
        Seg &o = a.seg[s];
        o.A = g.A; o.B = g.B; o.lda = g.lda; o.ldb = g.ldb; o.K = g.K;
        o.a_row_div = g.a_row_div > 0 ? g.a_row_div : 1;
        if (d->a_layout == 1 && o.a_row_div != 1) return CAPMI_EINVAL;
        o.vecA = aligned16(g.A) && (g.lda % 4 == 0);
        o.vecB = aligned16(g.B) && (g.ldb % 4 == 0);
        o.rdiv = (65536 + o.a_row_div - 1) / o.a_row_div;
        o.tstart = tiles;
        o.Apl = static_cast<const unsigned char *>(d->a_planes[s]);
        tiles += (g.K + BK - 1) / BK;
    }
    for (int s = d->nseg; s < CAPMI_MAX_SEG; ++s) {      // unused slots: never selected, but always valid to read
        a.seg[s] = a.seg[0];
        a.seg[s].tstart = 0x7fffffff;
    }
    a.tiles_total = tiles;
    static const int env_ablate = capmi::ablate_env("CAPMI_GEMM_ABLATE");   // (variants builds only)
    a.ablate = env_ablate;
    a.M = d->M; a.N = d->N; a.C = d->C; a.ldc = d->ldc;
    a.bias = d->bias; a.bias2 = d->bias2; a.row_bias = d->row_bias;
    a.row_bias_div = d->row_bias_div > 0 ? d->row_bias_div : 1;
    a.mul_mask = d->mul_mask; a.relu = d->relu; a.accumulate = d->accumulate;
    a.addend = d->addend ? d->addend : d->C;
    // workspace layout: [CAPMI_WS_COUNTER_FLOATS ints of tile tickets (zero between launches)][K-slice slabs]
    const int64_t slab_cap = d->partial ? d->partial_capacity - CAPMI_WS_COUNTER_FLOATS : 0;
    a.partial = d->partial ? d->partial + CAPMI_WS_COUNTER_FLOATS : nullptr;
    a.counters = reinterpret_cast<int *>(d->partial);
    hipStream_t st = (hipStream_t)stream;
    double ksum = 0, abytes = 0;   // algorithmic traffic of this launch: every operand element once + the output once
    for (int s = 0; s < d->nseg; ++s) {
        ksum += d->seg[s].K;
        abytes += (double)d->seg[s].K * ((double)d->M / a.seg[s].a_row_div);
    }
    const double bytes = 4.0 * (abytes + ksum * d->N + (double)d->M * d->N);
    const double flops = 2.0 * d->M * (double)d->N * ksum;
    int pcls = (d->M <= 64 && d->a_layout == 0) ? (d->b_layout == 0 ? CAPMI_PROF_GEMM_DECODE : CAPMI_PROF_GEMM_BPTT)
                                                 : CAPMI_PROF_GEMM_FAT;
    // (r4: >= 24 MB.  The 17-MB token-embedding segment that is left of the attention-LSTM gate GEMM when its other two segments ride in
    //  the select launch is a short, latency-dominated launch: it is accounted with the small decode GEMMs, class 0)
    if (pcls == CAPMI_PROF_GEMM_DECODE && bytes >= 24e6) pcls = CAPMI_PROF_GEMM_DECODE_STREAM;

    static const int env_path = capmi::research("CAPMI_GEMM_PATH", 0);
    static const int env_blocks = capmi::research("CAPMI_GEMM_BLOCKS", 512);
    bool ares_ok = d->a_layout == 0 && d->M <= 64 && BK == 32;
    for (int s = 0; s < d->nseg && ares_ok; ++s)     // branch-free 16-byte operand fetch: aligned, K % 4 == 0
        ares_ok = a.seg[s].vecA && (a.seg[s].K % 4 == 0) && (d->b_layout == 1 || a.seg[s].vecB);
    // ---- loader / consumer kernel (gemm_lc.hip) when every segment is also delivered as A planes ----
    // CAPMI_LC=0 (documented knob): the A-resident kernel below serves those GEMMs too (tests/test_planes_gpu.py compares the two)
    static const int env_lc = capmi::knob("CAPMI_LC", 1);
    {
        bool lc_ok = env_lc && env_path != 3 && ares_ok && d->M <= 64;
        for (int s = 0; s < d->nseg && lc_ok; ++s)
            lc_ok = d->a_planes[s] != nullptr && a.seg[s].a_row_div == 1 && a.seg[s].vecB;
        if (lc_ok && d->b_layout == 1) lc_ok = d->N % 4 == 0 && d->N >= 4;
        if (lc_ok) {
            static const int env_ab = capmi::research("CAPMI_ARES_BLOCKS", 256);
            static const int env_opt = capmi::research("CAPMI_LC_OPT", 1);
            const int want = d->splits > 0 ? ((d->N + 127) / 128) * d->splits : env_ab;
            int splits = 0;
            a.sl = lc_plan(d->N, tiles, want, &splits);
            const bool partial = splits > 1 || d->defer_reduce;
            if (!partial || (d->partial && (int64_t)splits * d->M * d->N <= slab_cap)) {
                a.splits = splits;
                a.to_partial = partial ? 1 : 0;
                a.self_reduce = 0;
                a.ablate = env_opt;              // bit 0: XCD-aware workgroup map
                d->splits_used = splits;
                int rc = launch_lc(a, d->b_layout, st, pcls, bytes, flops);
                if (rc) return rc;
                if (splits > 1 && !d->defer_reduce)
                    return splitk_reduce_addend(a.partial, splits, d->C, d->ldc, d->M, d->N, d->bias, d->bias2, d->row_bias,
                                               a.row_bias_div, d->mul_mask, d->relu, d->accumulate, d->addend, stream);
                return 0;
            }
        }
    }
    if (env_path != 3 && ares_ok) {     // CAPMI_GEMM_PATH=3 forces the LDS-tiled kernel
        // ---- A-resident path (gemm_ares.hip): activations stay in LDS, weights stream straight to VGPRs ----
        static const int env_ab = capmi::research("CAPMI_ARES_BLOCKS", 256);
        // bf16x3 split (see gemm_x3.hip) for the decode GEMMs too: activations split once when staged, weights split in
        // registers by the wave that streams them.  Gate GEMM 24.9 -> 19.9 us; greedy decodes stay token-exact on
        // the reference fixtures.  CAPMI_ARES_X3=0 restores the exact-fp32 MFMA.
        static const int env_ax3 = capmi::knob("CAPMI_ARES_X3", 1);
        int use_x3 = env_ax3;
        const int want = d->splits > 0 ? ((d->N + 127) / 128) * d->splits : env_ab;
        int splits = 0;
        int ts_cap = ares_ts_cap(d->M, use_x3);
        int ts_max = ares_plan(d->N, tiles, want, ts_cap, &splits);
        if (use_x3 && ((d->N + 127) / 128) * splits > want) {
            // the bf16 planes of 64 rows cap a slice at 12 chunks; when that pushes the grid past one workgroup per CU
            // (a second, mostly empty round), first try HALF-size slices with two workgroups resident per CU (<= 6
            // chunks = 77 KB of LDS each, everything in one round: 21.0 vs 27.8 us for dX = dG [W_ih | W_hh] with
            // N = 3000), else the exact-fp32 image with its longer slices (26.3 us)
            const int nblk = (d->N + 127) / 128;
            int splits2 = 0;
            const int ts2 = ares_plan(d->N, tiles, 2 * want, 3, &splits2);
            if (nblk * splits2 <= 2 * want && (int64_t)splits2 * d->M * d->N <= slab_cap) {
                ts_cap = 3; ts_max = ts2; splits = splits2;
            } else {
                int splits32 = 0;
                const int ts32 = ares_plan(d->N, tiles, want, ares_ts_cap(d->M, 0), &splits32);
                if (nblk * splits32 <= want) {
                    use_x3 = 0; ts_cap = ares_ts_cap(d->M, 0); ts_max = ts32; splits = splits32;
                }
            }
        }
        if ((splits > 1 || d->defer_reduce) && (int64_t)splits * d->M * d->N > slab_cap) ts_max = 99;   // slabs do not fit
        if (ts_max <= ts_cap) {
            a.splits = splits;
            a.to_partial = (splits > 1 || d->defer_reduce) ? 1 : 0;
            a.self_reduce = 0;
            static const int env_aopt = capmi::research("CAPMI_ARES_OPT", 2);
            a.ablate = env_aopt;             // speed-only switches of the A-resident kernel (see gemm_ares.hip)
            if (a.to_partial && (!d->partial || (int64_t)splits * d->M * d->N > slab_cap)) return CAPMI_EINVAL;
            d->splits_used = splits;
            int rc = launch_ares(a, d->b_layout, ts_max, use_x3, st, pcls, bytes, flops);
            if (rc) return rc;
            if (splits > 1 && !d->defer_reduce)
                return splitk_reduce_addend(a.partial, splits, d->C, d->ldc, d->M, d->N, d->bias, d->bias2, d->row_bias,
                                           a.row_bias_div, d->mul_mask, d->relu, d->accumulate, d->addend, stream);
            return 0;
        }
    }

    // tile shape by M: decode batches are skinny.  Every configuration gives each wave >= 2 independent
    // accumulator chains (a lone dependent v_mfma_f32_32x32x2 chain loses ~40 % to issue gaps).
    static const int env_cfg = capmi::research("CAPMI_GEMM_CFG", 1);
    int BM, BN;
    (void)env_cfg;
    if (d->M <= 32) { BM = 32; BN = 128; }
    else if (d->M <= 64 || (long long)d->M * d->N < 256LL * 1024) {
        BM = 64;
        BN = (d->b_layout == 0 && tiles >= 64 && d->N >= 1024) ? 128 : 64;   // measured: wide tile only pays on long-K weight streams
        if (env_cfg == 64 || env_cfg == 128) BN = env_cfg;                    // experiments: CAPMI_GEMM_CFG=64|128
    } else { BM = 128; BN = 128; }
    const int gm = (d->M + BM - 1) / BM, gn = (d->N + BN - 1) / BN;
    // fat GEMMs (time-batched BPTT): fp32 through the bf16 pipe by exact 3-way splitting (gemm_x3.hip);
    // CAPMI_GEMM_X3=0 keeps them on the exact-fp32 MFMA
    static const int env_x3 = capmi::knob("CAPMI_GEMM_X3", 1);
    bool x3_ok = env_x3 && BM == 128 && BN == 128 && BK == 32 &&
                 (d->a_layout == 0 || d->M % 4 == 0) && (d->b_layout == 0 || d->N % 4 == 0);   // 4-row quads of [K][rows] operands
    for (int s = 0; s < d->nseg && x3_ok; ++s)      // branch-free 16-byte staging loads: aligned operands, K % 4 == 0
        x3_ok = a.seg[s].a_row_div == 1 && a.seg[s].vecA && a.seg[s].vecB && a.seg[s].K % 4 == 0 &&
                // 32-bit per-lane byte offsets in the staging loads
                (uint64_t)(d->a_layout == 0 ? d->M : 1) * (uint64_t)a.seg[s].lda * 4 < (1ull << 32) &&
                (uint64_t)(d->b_layout == 0 ? d->N : 1) * (uint64_t)a.seg[s].ldb * 4 < (1ull << 32);
    int splits = d->splits;
    // r5: the same arithmetic on 256 x 128 tiles (gemm_x3w.hip) when that tiling is cheaper: a wide unit costs X3W_COST / 100 of
    // two narrow ones per K tile (fewer staging instructions per MFMA) but rounds M up to 256 and halves the number of units.
    // CAPMI_X3_TILE = 128 / 256 forces one tiling (0: by cost).
    static const int env_tile = capmi::knob("CAPMI_X3_TILE", 0);
    static const int env_wcost = capmi::research("CAPMI_X3W_COST", 165);
    static const int env_swap = capmi::knob("CAPMI_X3_SWAP", 1);      // 0: never plan the swapped-operand (C^T) form
    int tiling = 0;                   // 0: 128 x 128, 1: 256 (rows of C) x 128, 2: 128 x 256 (columns of C): the wide kernel on C^T
    const int gmw = (d->M + 255) / 256, gnw = (d->N + 255) / 256;
    if (x3_ok) {
        // persistent kernel, one workgroup per CU: pick the K split that minimises (rounds x K tiles per unit) plus the
        // slab traffic it causes, in units of one K-tile step of the narrow kernel (~1.5 us; slabs move at ~4 TB/s)
        double best = 1e30;
        int best_sp = 1;
        const int sp_lo = splits > 0 ? splits : 1, sp_hi = splits > 0 ? splits : 16;
        for (int w = 0; w < 3; ++w) {
            if ((w == 0 && env_tile == 256) || (w >= 1 && env_tile == 128)) continue;
            // measured (profiles/r05_fat_gemm_wide.md): with fewer than four 256-row tiles the wide tiling loses -- its last tile is
            // mostly padding and takes the edge path of the split on every K tile ([320 x 4000]: 72 vs 69 us)
            if (w == 1 && env_tile != 256 && gmw < 4) continue;
            // ... and it is not used for GEMMs whose reduction is deferred: those are the weight gradients that run on a SIDE stream
            // beside the backward chain (ops.DeferredGrads).  A wide workgroup is 16 waves x 124 registers + 144 KB of LDS -- it
            // owns its CU -- and a persistent grid of them beside another stream's kernels starved both (Transformer XE 14.3 ->
            // 36.5 ms); the 128 x 128 kernel leaves room for the chain's small kernels.  (Re-measured at the end of r5 with the final
            // kernels, scripts/r5_ab7.sh: wide deferred GEMMs 13.44-13.71 ms vs 13.31-13.35 -- still a loss, with 4 or 8 staging waves.)
            if (w == 1 && env_tile != 256 && d->defer_reduce && !d->allow_wide_deferred && !g_wide_deferred.load(std::memory_order_relaxed)) continue;
            // w == 2: SKINNY products (few rows, many columns: the per-step gate / dX GEMMs of a teacher-forced XE step at bs64,
            // [320 x 4000]) on the wide kernel with the operands SWAPPED -- the 256-row side of the tile runs along the WEIGHTS, the
            // activations are the 128-row operand, the epilogue writes C^T back as C in 16-byte pieces (x3_epilogue_t).  Row-major
            // activations only ([K][rows] A operands are the side-stream weight gradients, see above), at least four 256-column tiles.
            if (w == 2 && (env_swap == 0 || d->a_layout != 0 || gnw < 4 || gmw >= 4)) continue;
            const int out_tiles = w == 0 ? gm * gn : w == 1 ? gmw * gn : gnw * gm;
            const double step = w ? env_wcost / 100.0 : 1.0;
            for (int sp = sp_lo; sp <= sp_hi && sp <= tiles; ++sp) {
                if (splits == 0 && sp > 1 && (!d->partial || (int64_t)sp * d->M * d->N > slab_cap)) break;
                const double rounds = (double)((out_tiles * sp + 255) / 256);
                const double slab_us = sp > 1 ? (2.0 * sp + 1.0) * d->M * (double)d->N * 4.0 / 4.0e6 : 0.0;
                const double cost = rounds * ((tiles + sp - 1) / sp) * step + slab_us / 1.5;
                if (cost < best) { best = cost; best_sp = sp; tiling = w; }
            }
        }
        if (splits == 0) splits = best_sp;
    } else if (splits == 0) {
        // aim for ~2 workgroups per CU (512) but keep >= 4 K tiles per slice
        const int blocks = gm * gn;
        splits = 1;
        if (blocks < env_blocks * 3 / 4 && d->partial) {
            splits = (env_blocks + blocks - 1) / blocks;
            if (splits > tiles / 4) splits = tiles / 4;
            if (splits > 64) splits = 64;
            if (splits < 1) splits = 1;
            while (splits > 1 && (int64_t)splits * d->M * d->N > slab_cap) --splits;
        }
    }
    if (splits > tiles) splits = tiles;
    a.splits = splits;
    a.to_partial = (splits > 1 || d->defer_reduce) ? 1 : 0;
    // in-launch last-arriver reduction is implemented (G16 recipe) but measured SLOWER than the follow-up reduce
    // kernel at decode sizes (logit 33.5 vs 27.9 us, dX 34.2 vs 27-31 us): opt-in via CAPMI_GEMM_SELF_REDUCE=1.
    static const int env_self = capmi::research("CAPMI_GEMM_SELF_REDUCE", 0);
    a.self_reduce = (env_self && splits > 1 && !d->defer_reduce && gn * gm <= CAPMI_WS_COUNTER_FLOATS) ? 1 : 0;
    if (x3_ok) a.self_reduce = 0;      // the persistent kernel always leaves plain slabs
    if (a.to_partial && (!d->partial || (int64_t)splits * d->M * d->N > slab_cap)) return CAPMI_EINVAL;
    d->splits_used = splits;
    static const int env_log = capmi::knob("CAPMI_GEMM_LOG", 0);   // shape census on stderr (profiling)
    if (env_log)
        fprintf(stderr, "capmi_gemm M=%d N=%d tiles=%d al=%d bl=%d x3=%d wide=%d splits=%d defer=%d acc=%d\n", d->M, d->N, tiles,
                d->a_layout, d->b_layout, (int)x3_ok, x3_ok ? tiling : 0, splits, d->defer_reduce, d->accumulate);
    dim3 grid(gn, gm, splits);
    const ProfInfo pi{pcls, bytes, flops};
    int rc;
    if (x3_ok && tiling == 1) rc = launch_x3w(a, d->a_layout, d->b_layout, dim3(gn, gmw, splits), st, pcls, bytes, flops);
    else if (x3_ok && tiling == 2) {
        KArgs t = a;                                     // C^T = B A^T: operands, shapes and layouts swapped; C, its pitch and
        for (int s2 = 0; s2 < CAPMI_MAX_SEG; ++s2) {     // the epilogue operands keep C's orientation (KArgs.transposed)
            t.seg[s2].A = a.seg[s2].B; t.seg[s2].B = a.seg[s2].A;
            t.seg[s2].lda = a.seg[s2].ldb; t.seg[s2].ldb = a.seg[s2].lda;
            t.seg[s2].vecA = a.seg[s2].vecB; t.seg[s2].vecB = a.seg[s2].vecA;
            t.seg[s2].Apl = nullptr;
        }
        t.M = a.N; t.N = a.M;
        t.transposed = 1;
        rc = launch_x3w(t, d->b_layout, d->a_layout, dim3(gm, gnw, splits), st, pcls, bytes, flops);
    } else if (x3_ok) rc = launch_x3(a, d->a_layout, d->b_layout, grid, st, pcls, bytes, flops);
    else if (BM == 32 && BN == 128) rc = launch_cfg<32, 128, 1, 4, 3>(a, d->a_layout, d->b_layout, grid, st, pi);
    else if (BM == 64 && BN == 64) rc = launch_cfg<64, 64, 2, 2, 3>(a, d->a_layout, d->b_layout, grid, st, pi);
        else if (BM == 64 && BN == 128) rc = launch_cfg<64, 128, 1, 4, 2>(a, d->a_layout, d->b_layout, grid, st, pi);
    else rc = launch_cfg<128, 128, 2, 2, 2>(a, d->a_layout, d->b_layout, grid, st, pi);
    if (rc) return rc;
    if (splits > 1 && !d->defer_reduce && !a.self_reduce)
        return splitk_reduce_addend(a.partial, splits, d->C, d->ldc, d->M, d->N, d->bias, d->bias2, d->row_bias,
                                   a.row_bias_div, d->mul_mask, d->relu, d->accumulate, d->addend, stream);
    return 0;
}

// ---- r6: grouped weight-gradient GEMMs (capmi.h capmi_gemm_group_tn; kernel in gemm_x3w.hip) -------------------------------------
extern "C" int capmi_gemm_group_tn(capmi_group_gemm *items, int n, float *slabs, int64_t slab_floats, void *stream) {
    if (!items || n <= 0) return CAPMI_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    static const int env_group = capmi::knob("CAPMI_GEMM_GROUP", 1);      // 0: every item through capmi_gemm_f32 (A/B switch)
    constexpr int TILE = 256 * 128;
    // eligible items, longest K first (a round of the grid then holds units of one length; a stable order keeps the plan a
    // function of the shapes alone)
    int order[1024], n_ok = 0, rest[1024], n_rest = 0;
    if (n > 1024) return CAPMI_EINVAL;
    for (int i = 0; i < n; ++i) {
        capmi_group_gemm &g = items[i];
        if (!g.A || !g.B || !g.C || g.K <= 0 || g.M <= 0 || g.N <= 0) return CAPMI_EINVAL;
        const bool ok = env_group && BK == 32 && g.M % 4 == 0 && g.N % 4 == 0 && g.lda % 4 == 0 && g.ldb % 4 == 0 &&       // (any K, r6)
                        g.ldc % 4 == 0 && aligned16(g.A) && aligned16(g.B) && aligned16(g.C) && aligned16(g.colsum) &&
                        (uint64_t)g.lda * 16 * 4 + (uint64_t)g.M * 4 < (1ull << 32) && (uint64_t)g.ldb * 16 * 4 + (uint64_t)g.N * 4 < (1ull << 32);
        if (ok) order[n_ok++] = i;
        else rest[n_rest++] = i;
    }
    for (int i = 1; i < n_ok; ++i) {                     // insertion sort by K tiles, descending, stable
        const int v = order[i], kv = (items[v].K + BK - 1) / BK;
        int j = i - 1;
        while (j >= 0 && (items[order[j]].K + BK - 1) / BK < kv) { order[j + 1] = order[j]; --j; }
        order[j + 1] = v;
    }
    int pos = 0;
    while (pos < n_ok) {
        // one launch: as many items as fit the table, one entry kept free for the item the round boundary cuts in two
        int cnt = n_ok - pos < GROUP_MAX - 1 ? n_ok - pos : GROUP_MAX - 1;
        long long U = 0;
        for (int j = 0; j < cnt; ++j) {
            const capmi_group_gemm &g = items[order[pos + j]];
            U += (long long)((g.M + 255) / 256) * ((g.N + 127) / 128);
        }
        const long long F = (U / 256) * 256;             // tiles of the full rounds: whole-K, straight into C
        const long long R = U - F;                        // tiles of the last, partial round: K-sliced so that they fill it
        GTab t{};
        int ne = 0, unit = 0, rt = 0;
        long long seen = 0;
        int64_t slab_off = 0;
        double bytes = 0, flops = 0;
        for (int j = 0; j < cnt; ++j) {
            capmi_group_gemm &g = items[order[pos + j]];
            const int gm = (g.M + 255) / 256, gn = (g.N + 127) / 128, ot = gm * gn, kt = (g.K + BK - 1) / BK;
            bytes += 4.0 * ((double)g.K * (g.M + g.N) + (double)g.M * g.N);
            flops += 2.0 * g.M * (double)g.N * g.K;
            g.splits_used = 1;
            const int whole = seen >= F ? 0 : (int)(F - seen < ot ? F - seen : ot);     // tiles of this item inside the full rounds
            for (int part = 0; part < 2; ++part) {
                const int t0 = part ? whole : 0, nt = part ? ot - whole : whole;
                if (nt <= 0) continue;
                GItem &e = t.it[ne++];
                e.A = g.A; e.B = g.B; e.C = g.C; e.slab = nullptr; e.cs = g.colsum;
                e.lda = g.lda; e.ldb = g.ldb; e.ldc = g.ldc; e.K = g.K; e.M = g.M; e.N = g.N;
                e.kt = kt; e.gn = gn; e.tile0 = t0; e.ntiles = nt; e.accumulate = g.accumulate ? 1 : 0;
                int sp = 1;
                if (part) {
                    sp = (int)(256 / R);                 // R tiles x sp slices ~ one round of the grid
                    if (sp > kt / 4) sp = kt / 4;        // >= 4 K tiles per slice
                    if (sp > 32) sp = 32;
                    while (sp > 1 && (!slabs || slab_off + (int64_t)nt * sp * (TILE + 256) > slab_floats)) --sp;
                    if (sp < 1) sp = 1;
                }
                e.splits = sp;
                e.unit0 = unit;
                unit += nt * sp;
                if (sp > 1) {
                    e.slab = slabs + slab_off;
                    slab_off += (int64_t)nt * sp * (TILE + 256);      // tile pieces, then the [256]-float column-sum pieces
                    e.rtile0 = rt;
                    rt += nt;
                    g.splits_used = sp;
                }
            }
            seen += ot;
        }
        t.n = ne; t.units = unit; t.rtiles = rt;
        static const int env_abl = capmi::ablate_env("CAPMI_GROUP_ABLATE");      // (variants builds only)
        t.reserved = env_abl;
        int rc = launch_x3w_group(t, st, CAPMI_PROF_GEMM_FAT, bytes, flops);
        if (rc) return rc;
        pos += cnt;
    }
    for (int j = 0; j < n_rest; ++j) {
        capmi_group_gemm &g = items[rest[j]];
        capmi_gemm_desc d{};
        d.nseg = 1;
        d.seg[0].A = g.A; d.seg[0].B = g.B; d.seg[0].lda = g.lda; d.seg[0].ldb = g.ldb; d.seg[0].K = g.K; d.seg[0].a_row_div = 1;
        d.a_layout = 1; d.b_layout = 1; d.M = g.M; d.N = g.N; d.C = g.C; d.ldc = g.ldc; d.accumulate = g.accumulate;
        // (no K split here: `slabs` holds pieces of the group, not the zeroed ticket words a capmi_gemm_f32 workspace starts with)
        int rc = capmi_gemm_f32(&d, stream);
        if (rc) return rc;
        if (g.colsum) {
            const capmi_colsum_item ci{g.A, g.colsum, nullptr, g.K, g.M, g.lda, 0};
            if ((rc = capmi_colsum_batch_args(&ci, 1, stream)) != 0) return rc;
        }
        g.splits_used = -1;
    }
    return 0;
}
