// Deterministic token-embedding gradient: dE[tok(i), :] = sum over the positions i that fed token tok(i), in ascending i.
//
// The r1-r5 kernels scattered every (position, column) with atomicAdd: the sum order -- and so the last bits of the gradient -- changed
// from run to run, which made two training runs from the same seed drift apart (and a captured hipGraph step incomparable with the
// stepped one).  Here the workgroup of position i scans all token ids once, compacts the positions that hold ITS token into an LDS list
// in ascending order, and goes on only if it is the first of them (the token's leader).  The leader's four waves take contiguous
// quarters of the list, sum their rows in list order with several loads in flight, and the quarters are added in order 0..3: the
// partition depends on the list length alone, the result is a pure function of the inputs.  No zero-fill is needed for the rows that
// occur; like the atomic kernels it ADDS into dE (every caller zeroes dE first), as the only writer of a row.
// grid = (positions, ceil(D / 256)); 256 threads; dynamic LDS = positions * 4 bytes (+ 4 KB for the partials).
#pragma once
#include "capmi_common.h"

namespace capmi {

constexpr int EBD_THREADS = 256;
constexpr int EBD_MAX_ROWS = 14336;          // 56 KB of list

// G: float4 g(size_t position, int column) -- the gradient reaching the embedding output at (position, column .. column + 3)
template <typename G>
__device__ __forceinline__ void embed_bwd_det_body(const int64_t *__restrict__ tok, int T, int tok_ld, int rows, int D,
                                                   float *__restrict__ dE, const G &g) {
    extern __shared__ unsigned char ebd_smem[];
    int *list = reinterpret_cast<int *>(ebd_smem);                       // [rows]
    f32x4 *part = reinterpret_cast<f32x4 *>(ebd_smem + (size_t)((rows + 3) & ~3) * sizeof(int));   // [4 waves][64 lanes]
    __shared__ int s_cnt[5];
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
        n += s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
        __syncthreads();
        if (n > 0 && list[0] != i) return;                                // an earlier position holds this token: it leads
    }
    // (n >= 1: position i itself)
    const int c = (blockIdx.y * 64 + lane) * 4;                            // this lane's four columns
    const int per = (n + 3) >> 2, lo = wave * per, hi = min(n, lo + per);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    if (c < D) {
        int k = lo;
        for (; k + 4 <= hi; k += 4) {                                      // four rows in flight, added in list order
            const f32x4 a0 = g((size_t)list[k], c), a1 = g((size_t)list[k + 1], c), a2 = g((size_t)list[k + 2], c),
                        a3 = g((size_t)list[k + 3], c);
            acc += a0; acc += a1; acc += a2; acc += a3;
        }
        for (; k < hi; ++k) acc += g((size_t)list[k], c);
    }
    part[wave * 64 + lane] = acc;
    __syncthreads();
    if (wave == 0 && c < D) {
        f32x4 s = part[lane];
        s += part[64 + lane]; s += part[128 + lane]; s += part[192 + lane];
        f32x4 *o = reinterpret_cast<f32x4 *>(dE + (size_t)my * D + c);      // D % 4 == 0 (checked by the caller)
        *o = *o + s;                                                       // the one writer of this row: dE += like the atomic kernels
    }
}

static inline size_t embed_bwd_det_lds(int rows) { return (size_t)((rows + 3) & ~3) * sizeof(int) + 4 * 64 * sizeof(f32x4); }

}  // namespace capmi
