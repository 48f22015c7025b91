// Skinny fp32 MFMA GEMM for the decode step on gfx950: M <= 64 activation rows against 2-48 MB of
// weights (C[M,N] = sum_s A_s[M,K_s] * op(B_s)), the shape of every per-timestep GEMM of the
// UpDown rollout and of its BPTT (dX = dG W).
//
// Why a second kernel: at M = 64 with exact-f32 MFMA (v_mfma_f32_32x32x2_f32, 256 flop/clk/CU) the
// matrix pipe and HBM are co-limiting (att-LSTM gates: 1.5 GFLOP vs 48 MB -> ~10 us each), so the
// LDS-tiled kernel's barrier/transposed-store phases (2 barriers + 16 ds_write per 16 MFMAs) cost
// half the achievable rate.  Here there is NO LDS staging and NO barrier in the main loop:
//  * the MFMA k-index is free to permute (a sum), so lane (l&31, l>>5) takes the 16 CONSECUTIVE k
//    values k0 + 16*(l>>5) .. +15 of its row: both operands are loaded straight into VGPRs as four
//    16-byte loads per row (K-contiguous sources: activations, nn.Linear weights), every 128-byte
//    line being consumed completely by the two half-wave lanes of a row within the same 4 loads;
//    [K][N]-stored weights (dX = dG W) are read as 16 coalesced 128-byte rows per chunk;
//  * a workgroup owns a 64x32 (or 32x32) output tile; its 4 waves take interleaved 32-wide K chunks
//    (split-K inside the workgroup) with a two-deep register prefetch, each wave running 2
//    independent accumulator chains, and combine through LDS once at the end;
//  * across workgroups K is split only as far as needed to put >= 2 workgroups on every CU; the
//    slices are either handed to the fused consumer (LSTM cell) or combined IN THE LAUNCH by the
//    last-arriving workgroup of a tile (write-through slab stores + agent-scope ticket/acquire,
//    cdna_hip_programming.md G16) -- no separate reduce launch.
#include "gemm_common.h"
#include "profile.h"

namespace capmi_gemm {
namespace {

constexpr int CHK = 16;   // k values per lane per chunk (chunk = 32 k = BK)

template <bool KC>
__device__ __forceinline__ void load_frag(float (&f)[CHK], const float *__restrict__ src, int ld, int row, int nrows,
                                          int row_div, int kb, int K, int vec) {
    // KC: element (row, k) at src[(row/row_div)*ld + k]; else (k, row) at src[k*ld + row]
    if (KC) {
        if (row < nrows && kb < K) {
            const float *p = src + (size_t)(row / row_div) * ld + kb;
            if (vec && kb + CHK <= K) {
#pragma unroll
                for (int j = 0; j < CHK / 4; ++j) {
                    const f32x4 v = *reinterpret_cast<const f32x4 *>(p + 4 * j);
                    f[4 * j] = v[0]; f[4 * j + 1] = v[1]; f[4 * j + 2] = v[2]; f[4 * j + 3] = v[3];
                }
            } else {
#pragma unroll
                for (int j = 0; j < CHK; ++j) f[j] = (kb + j < K) ? p[j] : 0.f;
            }
        } else {
#pragma unroll
            for (int j = 0; j < CHK; ++j) f[j] = 0.f;
        }
    } else {
        const bool ok = row < nrows;
        const float *p = src + (size_t)kb * ld + row;
#pragma unroll
        for (int j = 0; j < CHK; ++j) f[j] = (ok && kb + j < K) ? p[(size_t)j * ld] : 0.f;
    }
}

template <bool BKC, int TM>
__device__ __forceinline__ void load_chunk(const KArgs &a, int c, int m0, int n0, int l31, int half,
                                           float (&A)[TM][CHK], float (&B)[CHK]) {
    int s, k0;
    locate(a, c, s, k0);
    const Seg &sg = a.seg[s];
    const int kb = k0 + CHK * half;
#pragma unroll
    for (int i = 0; i < TM; ++i)
        load_frag<true>(A[i], sg.A, sg.lda, m0 + 32 * i + l31, a.M, sg.a_row_div, kb, sg.K, sg.vecA);
    load_frag<BKC>(B, sg.B, sg.ldb, n0 + l31, a.N, 1, kb, sg.K, sg.vecB);
}

template <int TM>
__device__ __forceinline__ void mma_chunk(const float (&A)[TM][CHK], const float (&B)[CHK], f32x16 (&acc)[TM]) {
#pragma unroll
    for (int j = 0; j < CHK; ++j)
#pragma unroll
        for (int i = 0; i < TM; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(A[i][j], B[j], acc[i], 0, 0, 0);
}

__device__ __forceinline__ float epilogue(const KArgs &a, float v, int row, int col) {
    if (a.bias) v += a.bias[col];
    if (a.bias2) v += a.bias2[col];
    if (a.row_bias) v += a.row_bias[(size_t)(row / a.row_bias_div) * a.N + col];
    if (a.relu) v = fmaxf(v, 0.f);
    if (a.mul_mask) v *= a.mul_mask[(size_t)row * a.N + col];
    if (a.accumulate) v += a.C[(size_t)row * a.ldc + col];
    return v;
}

template <bool BKC, int TM>
__global__ __launch_bounds__(256, 2) void gemm_skinny_kernel(const KArgs a) {
    constexpr int ROWS = 32 * TM;
    __shared__ float red[4][ROWS][33];
    __shared__ int s_last;
    const int n0 = blockIdx.x * 32, m0 = blockIdx.y * ROWS, z = blockIdx.z;
    const int c_begin = (int)(((long long)a.tiles_total * z) / a.splits);
    const int c_end = (int)(((long long)a.tiles_total * (z + 1)) / a.splits);
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int l31 = lane & 31, half = lane >> 5;

    f32x16 acc[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

    float A0[TM][CHK], B0[CHK], A1[TM][CHK], B1[CHK];
    int c = c_begin + wid;
    if (c < c_end) load_chunk<BKC, TM>(a, c, m0, n0, l31, half, A0, B0);
    while (c < c_end) {
        if (c + 4 < c_end) load_chunk<BKC, TM>(a, c + 4, m0, n0, l31, half, A1, B1);
        mma_chunk<TM>(A0, B0, acc);
        c += 4;
        if (c >= c_end) break;
        if (c + 4 < c_end) load_chunk<BKC, TM>(a, c + 4, m0, n0, l31, half, A0, B0);
        mma_chunk<TM>(A1, B1, acc);
        c += 4;
    }

    // combine the 4 waves' K slices through LDS (C/D map: col = lane&31, row = (r&3)+8*(r>>2)+4*(lane>>5))
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) red[wid][32 * i + (r & 3) + 8 * (r >> 2) + 4 * half][l31] = acc[i][r];
    __syncthreads();

    const size_t MN = (size_t)a.M * a.N;
    if (!a.to_partial) {
        for (int e = threadIdx.x; e < ROWS * 32; e += 256) {
            const int rr = e >> 5, cc = e & 31;
            const int row = m0 + rr, col = n0 + cc;
            if (row < a.M && col < a.N)
                a.C[(size_t)row * a.ldc + col] =
                    epilogue(a, red[0][rr][cc] + red[1][rr][cc] + red[2][rr][cc] + red[3][rr][cc], row, col);
        }
        return;
    }
    float *slab = a.partial + (size_t)z * MN;
    if (!a.self_reduce) {   // slices go to a fused consumer kernel (stream order makes them visible)
        for (int e = threadIdx.x; e < ROWS * 32; e += 256) {
            const int rr = e >> 5, cc = e & 31;
            const int row = m0 + rr, col = n0 + cc;
            if (row < a.M && col < a.N)
                slab[(size_t)row * a.N + col] = red[0][rr][cc] + red[1][rr][cc] + red[2][rr][cc] + red[3][rr][cc];
        }
        return;
    }
    // in-launch reduction: publish this slice write-through (sc1), take a ticket, last arriver reduces
    for (int e = threadIdx.x; e < ROWS * 32; e += 256) {
        const int rr = e >> 5, cc = e & 31;
        const int row = m0 + rr, col = n0 + cc;
        if (row < a.M && col < a.N)
            __hip_atomic_store(slab + (size_t)row * a.N + col,
                               red[0][rr][cc] + red[1][rr][cc] + red[2][rr][cc] + red[3][rr][cc], __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // every storing wave drains its write-through stores
    __syncthreads();
    int *ticket = a.counters + (blockIdx.y * gridDim.x + blockIdx.x);
    if (threadIdx.x == 0) {
        const int t = __hip_atomic_fetch_add(ticket, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_last = (t == a.splits - 1);
    }
    __syncthreads();
    if (!s_last) return;
    if (threadIdx.x == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");   // drop stale L1 lines once
    __syncthreads();
    for (int e = threadIdx.x; e < ROWS * 32; e += 256) {
        const int rr = e >> 5, cc = e & 31;
        const int row = m0 + rr, col = n0 + cc;
        if (row < a.M && col < a.N) {
            const float *p = a.partial + (size_t)row * a.N + col;
            float v = 0.f;
            for (int s0 = 0; s0 < a.splits; s0 += 8) {
                float t[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) t[u] = (s0 + u < a.splits) ? p[(size_t)(s0 + u) * MN] : 0.f;
#pragma unroll
                for (int u = 0; u < 8; ++u) v += t[u];
            }
            a.C[(size_t)row * a.ldc + col] = epilogue(a, v, row, col);
        }
    }
    if (threadIdx.x == 0) __hip_atomic_store(ticket, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // re-arm
}

}  // namespace

int launch_skinny(const KArgs &a, int b_layout, int tm, dim3 grid, hipStream_t st) {
    if (b_layout == 0) {
        if (tm == 1) hipLaunchKernelGGL((gemm_skinny_kernel<true, 1>), grid, dim3(256), 0, st, a);
        else hipLaunchKernelGGL((gemm_skinny_kernel<true, 2>), grid, dim3(256), 0, st, a);
    } else {
        if (tm == 1) hipLaunchKernelGGL((gemm_skinny_kernel<false, 1>), grid, dim3(256), 0, st, a);
        else hipLaunchKernelGGL((gemm_skinny_kernel<false, 2>), grid, dim3(256), 0, st, a);
    }
    CAPMI_CHECK_LAUNCH();
    return 0;
}

}  // namespace capmi_gemm
