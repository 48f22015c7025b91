// Loader / consumer decode GEMM for gfx950 (round 3): C[M <= 64, N] = sum_s A_s[M, K_s] op(B_s) with the activations
// delivered as A PLANES (capmi_common.h) -- the weight-streaming GEMMs of the decode step (LSTM gates, vocabulary
// projection, dX = dG W of the BPTT).
//
// What the phase traces of the two earlier kernels (gemm_ares.hip) say about this shape (48 MB of fp32 weights against 60
// rows, one workgroup per CU):
//  * a wave that loads weights into VGPRs stalls AT ISSUE once the CU's memory queue is full (issuing a chunk's loads takes as
//    long as a chunk takes to arrive) and an in-order wave that is stalled cannot issue its MFMAs: with a deep ring the whole
//    weight slice is requested up front and the 9.2k cycles of MFMAs run AFTER the stream (W-only 9.8 us, + MFMA 5.8 us,
//    + epilogue 2.9 us, + activations 1.4 us = 19.2 us: nothing overlaps); with a 2-deep ring they overlap partly (16.0 us);
//  * no MFMA can start before the WHOLE activation slice (147 KB) has landed and been published by a barrier: ~10k cycles;
//  * the two K halves of a column group meet in LDS behind the loop: 4-5k cycles more.
// Here the roles are split (MI355X_MICROARCH.md, rows ldsdma-fill / prefetch-credit):
//  * waves 8-11 are LOADERS: they copy one K chunk per stage -- its activation image (12 KB, three bf16 planes of 64 rows) and
//    its weight tile (128 columns x 32 k fp32 = 16 KB) -- into a 5-stage LDS ring with global_load_lds_dwordx4.  LDS-DMA has no
//    register destination: a stalled loader holds up nobody, and 4 stages (112 KB) are in flight per CU from the first cycle;
//  * waves 0-7 are CONSUMERS, two per SIMD: column group cg = w & 3 (32 columns, all 64 rows) x stage parity w >> 2 (even / odd
//    stages).  A consumer ds_reads the weight fragment of its stage, splits it to bf16x3 in registers, ds_reads the
//    activation fragments and issues 24 MFMAs; it never issues a global load.  One wave per SIMD was tried first: its
//    ds_read -> split (VALU) -> MFMA chain is serial (1.68k cycles per stage against 768 of MFMA; hipcc does not interleave the
//    next stage's split with the MFMAs even under sched_group_barrier), so two waves in opposite phases share the SIMD: a
//    stage is computed across TWO barrier intervals (split + first k-step | second k-step) and in every interval the SIMD has
//    one wave's VALU beside the other wave's MFMAs;
//  * one s_barrier per stage hands stage k to its consumers and slot k-2 back to the loaders; stage 0 is consumed as soon as IT
//    has landed, not the slice, and the ring fills up behind it;
//  * the two parities of a column group meet in LDS behind the loop (the ring is dead by then);
//  * the weight tile is stored [column][32 k] with its eight 16-byte pieces XOR-swizzled by (column >> 1) & 7 -- applied on the
//    SOURCE address of each lane, LDS-DMA writes linearly -- so the consumers' ds_read_b128 are conflict-free; [K][N] weights
//    (dX GEMMs) are stored [k][128 columns] and read with conflict-free ds_read_b32.
// A stage's chunk is located (segment, pointer, rows left) on the scalar unit from SGPR-pinned segment fields.  The slice
// length is a runtime value (no LDS-resident slice any more), so one instance serves every K.
#include "gemm_lc_body.h"
#include "profile.h"
#include <hip/hip_ext.h>

namespace capmi_gemm {
namespace {

template <bool BKC, int TM, int ABL = 0, int WAUX = 0>
__global__ __launch_bounds__(LC_NT) void gemm_lc_kernel(const KArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char ldsb[];
    gemm_lc_body<BKC, TM, ABL, WAUX>(a, blockIdx.x, blockIdx.y, gridDim.x, gridDim.y, ldsb);
}

}  // namespace

// K chunks per workgroup slice and the number of slices for ~want_blocks workgroups (one per CU: the ring takes 140 KB of LDS)
int lc_plan(int N, int tiles, int want_blocks, int *splits) {
    const int nblk = (N + LC_BN - 1) / LC_BN;
    int s = want_blocks / nblk;
    if (s < 1) s = 1;
    if (s > tiles) s = tiles;
    int sl = (tiles + s - 1) / s;
    if (sl < 2 && tiles >= 2) sl = 2;               // 1-chunk slices are all prologue
    *splits = (tiles + sl - 1) / sl;
    return sl;
}

template <bool BKC, int TM>
static int launch_lc_t(const KArgs &a, hipStream_t st, int pcls, double bytes, double flops) {
    constexpr size_t lds = (size_t)LC_NS * LC_STAGE;
    static_assert(lds <= 160 * 1024, "ring does not fit the CU's LDS");
    dim3 grid((a.N + LC_BN - 1) / LC_BN, a.splits);
    hipEvent_t e0, e1;
    const bool prof = capmi_prof::take_events(pcls, &e0, &e1, bytes, flops);
    static const int abl = capmi::ablate_env("CAPMI_LC_ABLATE");
#define CAPMI_LC_GO2(A_, W_)                                                                                               \
    do {                                                                                                                   \
        static bool set = false;                                                                                           \
        if (!set) {                                                                                                        \
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&gemm_lc_kernel<BKC, TM, A_, W_>),                    \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);                             \
            set = true;                                                                                                    \
        }                                                                                                                  \
        if (prof) hipExtLaunchKernelGGL((gemm_lc_kernel<BKC, TM, A_, W_>), grid, dim3(LC_NT), lds, st, e0, e1, 0, a);      \
        else hipLaunchKernelGGL((gemm_lc_kernel<BKC, TM, A_, W_>), grid, dim3(LC_NT), lds, st, a);                         \
    } while (0)
#define CAPMI_LC_GO(A_) CAPMI_LC_GO2(A_, LC_WAUX)
#ifdef CAPMI_VARIANTS
    static const int env_nt = capmi::research("CAPMI_LC_NT", LC_WAUX);
    if (env_nt != LC_WAUX && abl == 0) {
        if (env_nt == 2) CAPMI_LC_GO2(0, 2); else CAPMI_LC_GO2(0, 0);
    } else if constexpr (BKC && TM == 2) {
        switch (abl) {
            case 1: CAPMI_LC_GO(1); break;
            case 2: CAPMI_LC_GO(2); break;
            case 4: CAPMI_LC_GO(4); break;
            case 8: CAPMI_LC_GO(8); break;
            case 3: CAPMI_LC_GO(3); break;
            case 10: CAPMI_LC_GO(10); break;
            case 11: CAPMI_LC_GO(11); break;
            case 15: CAPMI_LC_GO(15); break;
            case 16: CAPMI_LC_GO(16); break;
            case 32: CAPMI_LC_GO(32); break;
            case 64: CAPMI_LC_GO(64); break;
            case 128: CAPMI_LC_GO(128); break;
            case 192: CAPMI_LC_GO(192); break;
            case 160: CAPMI_LC_GO(160); break;
            default: CAPMI_LC_GO(0); break;
        }
    } else {
        CAPMI_LC_GO(0);
    }
#else
    (void)abl;
    CAPMI_LC_GO(0);
#endif
#undef CAPMI_LC_GO
#undef CAPMI_LC_GO2
    CAPMI_CHECK_LAUNCH();
    return 0;
}

// a.splits * a.sl must cover a.tiles_total with no empty slice; every segment carries planes
thread_local LcCapture *g_lc_capture = nullptr;

int launch_lc(const KArgs &a, int b_layout, hipStream_t st, int pcls, double bytes, double flops) {
    if (a.sl < 1 || a.M > 64 || (long long)a.splits * a.sl < a.tiles_total || (long long)(a.splits - 1) * a.sl >= a.tiles_total)
        return CAPMI_EINVAL;
    if (g_lc_capture) {                              // the caller launches the body itself (fused select + GEMM)
        LcCapture &c = *g_lc_capture;
        c.a = a; c.b_layout = b_layout; c.grid_x = (a.N + LC_BN - 1) / LC_BN; c.grid_y = a.splits; c.tm = a.M <= 32 ? 1 : 2;
        c.filled = true;
        return 0;
    }
    if (b_layout == 0) return a.M <= 32 ? launch_lc_t<true, 1>(a, st, pcls, bytes, flops) : launch_lc_t<true, 2>(a, st, pcls, bytes, flops);
    return a.M <= 32 ? launch_lc_t<false, 1>(a, st, pcls, bytes, flops) : launch_lc_t<false, 2>(a, st, pcls, bytes, flops);
}

}  // namespace capmi_gemm
