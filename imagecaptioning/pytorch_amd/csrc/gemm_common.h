// Shared argument block of the GEMM kernels (LDS-tiled kernel in gemm_f32.hip, A-resident decode kernel in gemm_ares.hip,
// bf16x3 fat kernel in gemm_x3.hip).
#pragma once
#include "capmi_common.h"
#include "../../../include/capmi.h"

namespace capmi_gemm {

#ifndef CAPMI_BK
#define CAPMI_BK 32
#endif
constexpr int BK = CAPMI_BK;
constexpr int NT = 256;

struct Seg {
    const float *A, *B;
    int lda, ldb, K, a_row_div;
    int vecA, vecB;   // 16-byte vector loads legal for this segment
    int rdiv;         // ceil(65536 / a_row_div): row / a_row_div == (row * rdiv) >> 16 for row < 64
    int tstart;       // index of this segment's first K tile in the flat tile list (INT_MAX for unused slots)
    const unsigned char *Apl;   // the same activations pre-split as "A planes" (capmi_common.h), or null
};

struct KArgs {
    Seg seg[CAPMI_MAX_SEG];
    int nseg;
    int M, N;
    float *C;
    int ldc;
    const float *bias, *bias2, *row_bias;
    int row_bias_div;
    const float *mul_mask;
    int relu, accumulate;
    const float *addend;   // what `accumulate` adds: the descriptor's addend, else C (row pitch ldc either way)
    float *partial;
    int splits;
    int to_partial;      // write raw K-slice sums to `partial` (split-K and/or fused consumer)
    int tiles_total;     // sum over segments of ceil(K/BK)
    int *counters;       // per-output-tile arrival tickets (in-launch split-K reduction), zero between launches
    int self_reduce;     // last-arriving K slice of a tile reduces all slices and applies the epilogue
    int sl;              // loader / consumer kernel: K chunks per workgroup slice (runtime)
    int ablate;          // experiments only (CAPMI_GEMM_ABLATE): 1 = skip MFMA phase, 2 = skip global loads, 4 = skip LDS writes
    int transposed;      // r5 (gemm_x3w only): the kernel computes C^T -- seg A/B, M/N and the layouts arrive SWAPPED, C / ldc / bias /
                         // row_bias / mask / addend keep C's own orientation ([N rows of this struct][M columns]); see x3_epilogue_t
};

// flat K-tile index -> (segment, k0)
__device__ __forceinline__ void locate(const KArgs &a, int tile, int &s, int &k0) {
    s = 0;
    int t = tile;
#pragma unroll
    for (int i = 0; i < CAPMI_MAX_SEG; ++i) {
        if (i < a.nseg - 1 && s == i) {
            const int nt = (a.seg[i].K + BK - 1) / BK;
            if (t >= nt) {
                t -= nt;
                s = i + 1;
            }
        }
    }
    k0 = t * BK;
}


// A-resident skinny path (M <= 64, A stored [M][K]); defined in gemm_ares.hip
int ares_plan(int N, int tiles, int want_blocks, int ts_cap, int *splits);
int ares_ts_cap(int M, int x3);
int launch_ares(const KArgs &a, int b_layout, int ts_max, int x3, hipStream_t st, int pcls, double bytes, double flops);
// r4: a caller that wants to run the loader / consumer GEMM INSIDE another launch (sampler.hip: select + GEMM in one grid) sets this
// pointer around capmi_gemm_f32(); launch_lc() then stores its arguments here instead of launching.  filled stays false when the
// dispatcher took another path (that GEMM was launched normally).
struct LcCapture {
    KArgs a;
    int b_layout, grid_x, grid_y, tm;
    bool filled;
};
extern thread_local LcCapture *g_lc_capture;
// loader / consumer kernel on A planes (gemm_lc.hip): lc_plan picks a.sl and a.splits
int lc_plan(int N, int tiles, int want_blocks, int *splits);
int launch_lc(const KArgs &a, int b_layout, hipStream_t st, int pcls, double bytes, double flops);

// fat GEMMs through the bf16 pipe by exact 3-way operand splitting; defined in gemm_x3.hip
int launch_x3(const KArgs &a, int a_layout, int b_layout, dim3 grid, hipStream_t st, int pcls, double bytes, double flops);
// ... on 256 x 128 tiles (gemm_x3w.hip, r5); grid = (gn, gm of the 256-row tiling, splits)
int launch_x3w(const KArgs &a, int a_layout, int b_layout, dim3 grid, hipStream_t st, int pcls, double bytes, double flops);

// ---- r6: grouped launch of independent weight-gradient GEMMs C_i = A_i^T B_i ([K][M] x [K][N] operands) on the 256 x 128 kernel ----
// One table entry = a run of output tiles of one GEMM (row-major tile order of its 256 x 128 tiling).  Entries with splits == 1 write
// (or add to) C directly; the tail entries of a launch are cut into K slices that leave [256 x 128] pieces in `slab`
// (piece (tile_local, z) at slab + (tile_local * splits + z) * 32768 floats), summed into C by group_reduce.
struct GItem {
    const float *A, *B;
    float *C;
    float *slab;
    float *cs;         // optional: cs[m] = sum_k A[k][m] (the bias gradient beside the weight gradient), taken by the staging waves of the
                       // entry's first column tiles from the operand they stage anyway; K-sliced entries leave [256]-float pieces
                       // behind their tile pieces (slab + ntiles * splits * 32768 + (tile_local * splits + z) * 256)
    int lda, ldb, ldc, K, M, N;
    int kt;            // ceil(K / BK)
    int gn;            // column tiles of the GEMM
    int tile0, ntiles; // tile range of this entry
    int splits;
    int unit0;         // first unit of this entry in the launch's unit list (ntiles * splits units, z-major)
    int rtile0;        // first piece-set of this entry in the reduction launch's list (splits > 1 only)
    int accumulate;    // C += instead of C =
};
constexpr int GROUP_MAX = 42;      // entries per launch: the table travels BY VALUE in the kernel arguments (42 x 96 B + 16 < 4 KB)
struct GTab {
    GItem it[GROUP_MAX];
    int n, units, rtiles, reserved;
};
int launch_x3w_group(const GTab &t, hipStream_t st, int pcls, double bytes, double flops);

}  // namespace capmi_gemm
