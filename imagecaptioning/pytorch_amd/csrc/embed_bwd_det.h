// Deterministic token-embedding gradient: dE[tok(i), :] = sum over the positions i that fed token tok(i), in ascending i.
//
// The r1-r5 kernels scattered every (position, column) with atomicAdd: the sum order -- and so the last bits of the gradient -- changed
// from run to run, which made two training runs from the same seed drift apart (and a captured hipGraph step incomparable with the
// stepped one).  Here the workgroup of position i scans all token ids once, compacts the positions that hold ITS token into an LDS list
// in ascending order, and goes on only if it is the first of them (the token's leader).  The leader's sixteen waves take contiguous
// sixteenths of the list, sum their rows in list order with eight loads in flight, and the parts are added in order 0..15: the
// partition depends on the list length alone, the result is a pure function of the inputs.  No zero-fill is needed for the rows that
// occur; like the atomic kernels it ADDS into dE (every caller zeroes dE first), as the only writer of a row.
// (r6: 16 waves instead of 4 -- the BOS / padding token holds a third of all positions (2 000+ of the 6 720 of a Transformer XE step), and
//  its one leader summed them as 4 x 500 dependent round trips: 529 us for 14 MB, profiles/r06_txe_kernel_stats.md.)
// grid = (positions, ceil(D / 256)); 1024 threads; dynamic LDS = positions * 4 bytes (+ 16 KB for the partials).
#pragma once
#include "capmi_common.h"

namespace capmi {

constexpr int EBD_THREADS = 1024, EBD_WAVES = EBD_THREADS / 64;
constexpr int EBD_MAX_ROWS = 14336;          // 56 KB of list

// G: float4 g(size_t position, int column) -- the gradient reaching the embedding output at (position, column .. column + 3)
template <typename G>
__device__ __forceinline__ void embed_bwd_det_body(const int64_t *__restrict__ tok, int T, int tok_ld, int rows, int D,
                                                   float *__restrict__ dE, const G &g) {
    extern __shared__ unsigned char ebd_smem[];
    int *list = reinterpret_cast<int *>(ebd_smem);                       // [rows]
    f32x4 *part = reinterpret_cast<f32x4 *>(ebd_smem + (size_t)((rows + 3) & ~3) * sizeof(int));   // [waves][64 lanes]
    __shared__ int s_cnt[EBD_WAVES];
    const int i = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int64_t my = tok[(size_t)(i / T) * tok_ld + (i % T)];
    int n = 0;                                                             // list length so far (uniform)
    for (int base = 0; base < rows; base += EBD_THREADS) {
        const int j = base + tid;
        const bool m = j < rows && tok[(size_t)(j / T) * tok_ld + (j % T)] == my;
        const unsigned long long b = __ballot(m);
        if (lane == 0) s_cnt[wave] = __popcll(b);
        __syncthreads();
        int before = n;
        for (int w = 0; w < wave; ++w) before += s_cnt[w];
        if (m) list[before + __popcll(b & ((1ull << lane) - 1ull))] = j;
#pragma unroll
        for (int w = 0; w < EBD_WAVES; ++w) n += s_cnt[w];
        __syncthreads();
        if (n > 0 && list[0] != i) return;                                // an earlier position holds this token: it leads
    }
    // (n >= 1: position i itself)
    const int c = (blockIdx.y * 64 + lane) * 4;                            // this lane's four columns
    const int per = (n + EBD_WAVES - 1) / EBD_WAVES, lo = wave * per, hi = min(n, lo + per);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    if (c < D) {
        int k = lo;
        for (; k + 8 <= hi; k += 8) {                                      // eight rows in flight, added in list order
            f32x4 a[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) a[u] = g((size_t)list[k + u], c);
#pragma unroll
            for (int u = 0; u < 8; ++u) acc += a[u];
        }
        for (; k < hi; ++k) acc += g((size_t)list[k], c);
    }
    part[wave * 64 + lane] = acc;
    __syncthreads();
    if (wave == 0 && c < D) {
        f32x4 s = part[lane];
#pragma unroll
        for (int w = 1; w < EBD_WAVES; ++w) s += part[w * 64 + lane];
        f32x4 *o = reinterpret_cast<f32x4 *>(dE + (size_t)my * D + c);      // D % 4 == 0 (checked by the caller)
        *o = *o + s;                                                       // the one writer of this row: dE += like the atomic kernels
    }
}

static inline size_t embed_bwd_det_lds(int rows) { return (size_t)((rows + 3) & ~3) * sizeof(int) + EBD_WAVES * 64 * sizeof(f32x4); }

}  // namespace capmi
