// bf16x3 fat GEMM, WIDE tile (round 5): 256 x 128 x 32 per workgroup -- the same exact 3-way operand split and the same six
// v_mfma_f32_32x32x16_bf16 per fp32 MAC tile as gemm_x3.hip (see its header for the arithmetic), other proportions.
//
// Why: the 128 x 128 kernel's K loop runs at ~2 500 cycles per K tile against 1 536 of MFMA per SIMD (profiles/r03_x3pl_probe.md,
// r04_fat_gemm_epilogue.md).  Its SIMDs are ISSUE-bound, not pipe-bound: beside the 48 MFMAs of a K tile a SIMD also has to issue
// its share of the operand split -- 256 rows x 32 k of fp32 -> three bf16 planes = 5.5 VALU per element, ~230 instructions of the
// two staging waves -- plus 24 ds_read_b128 of the MFMA wave: ~5.5 issues per MFMA gap where ~5 hide (MI355X_MICROARCH.md,
// "single-issue instructions HIDDEN per v_mfma gap").  A 256 x 128 tile stages (256 + 128) rows for TWICE the MFMAs:
//     per SIMD and K tile: 96 MFMAs (two MFMA waves) | 12 loads + 264 split VALU + 36 ds_write_b64 (staging) | 48 ds_read_b128
//     = ~4.0 issues per MFMA gap, and a fragment set read from LDS still feeds 24 MFMAs.
// Two MFMA waves share each SIMD: while one waits for its ds_reads the other's MFMAs keep the pipe busy, so the loop needs no
// software pipelining (and no second fragment register set).  Staging: one wave per SIMD (768 threads) or -- the default, measured
// 3-7 % faster -- two (1 024 threads, 124 registers): the first splits the two A blocks, the second B.
//
// LDS: a stage is 3 planes x (256 + 128) rows x 32 k bf16.  With the 80-byte padded rows of gemm_x3.hip two stages would take
// 184 KB; here a row is exactly 64 bytes and its four 16-byte pieces are XOR-swizzled by (row >> 2) & 3, which makes the 16 lanes
// of every ds_read_b128 lane group (rows {0-3, 12-15, 20-27} / {4-11, 16-19, 28-31} of a 32-row block) hit 16 distinct bank quads:
// 2 stages = 144 KB, one workgroup per CU, persistent over a unit list like the 128 x 128 kernel.
//
// Edges: rows beyond M / N are clamped to a valid address at fetch time and zeroed when the registers are split; the facts needed
// for that (valid rows of the three 128-row operand blocks, valid k of the tile) are workgroup-uniform and ride along with the
// register set as scalars -- no per-quad keep-factor registers.
#include "gemm_x3_common.h"
#include "profile.h"
#include <hip/hip_ext.h>
#include <stdlib.h>
#include <stdio.h>

namespace capmi_gemm {
namespace {

constexpr int WBM = 256, WBN = 128;
constexpr int WPL_A = WBM * 32, WPL_B = WBN * 32;          // bf16 elements per plane
constexpr int WSTAGE = 3 * (WPL_A + WPL_B);                // 36 864 elements = 72 KB
static_assert(BK == 32, "gemm_x3w assumes 32-wide K tiles");

// One 128-row block of an operand: HBM -> 16 floats per thread (4 quads of 4 consecutive k of one row), branch-free with a fixed
// number of loads (see gemm_x3.hip Stager for why); 256 threads.
template <bool KC>
struct WStager {
    gcb src;
    unsigned voff[4];     // KC: byte offset of (clamped row, in-tile k) per quad; k-major: voff[0] = (4kg * ld + clamped column) * 4
    unsigned voff0;       // k-major: the kg = 0 variant of voff[0]
    int ld, K, kq0, valid_rows;
    __device__ __forceinline__ void init(const float *src_, int ld_, int K_) {
        src = (gcb)(uintptr_t)src_;
        ld = ld_; K = K_;
    }
    __device__ __forceinline__ void set_tile(int row0, int nrows, int tid) {
        valid_rows = min(max(nrows - row0, 0), 128);
        if (KC) {
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const int idx = p * NT + tid;
                const int row = row0 + idx / 8;
                voff[p] = ((unsigned)min(row, nrows - 1) * (unsigned)ld + (unsigned)(idx % 8) * 4u) * 4u;
            }
            kq0 = (tid & 7) * 4;
        } else {
            const int lane = tid & 63, kg = lane >> 3, mq = (tid >> 6) * 8 + (lane & 7);
            const int row = row0 + 4 * mq;
            voff0 = (unsigned)min(row, nrows - 4) * 4u;
            voff[0] = (unsigned)(4 * kg) * (unsigned)ld * 4u + voff0;
            kq0 = 4 * kg;
        }
    }
    __device__ __forceinline__ void fetch(float (&r)[16], int k0) const {
        const bool inside = k0 + kq0 < K;                      // per lane; false only in the last tile of a segment
        if (KC) {
            const unsigned back = inside ? 0u : (unsigned)kq0 * 4u;      // -> the tile's first quad of the same row
            gcb b = src + (size_t)k0 * 4;
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const f32x4 v = *(gcf4)(b + (voff[p] - back));
                r[4 * p] = v[0]; r[4 * p + 1] = v[1]; r[4 * p + 2] = v[2]; r[4 * p + 3] = v[3];
            }
        } else {
            const unsigned off = inside ? voff[0] : voff0;
            const size_t ld4 = (size_t)ld * 4;
            gcb b = src + (size_t)k0 * ld4;
            f32x4 v[4];
            if ((K & 3) && k0 + BK > K) {
                // r6: K is not a multiple of 4 and this is its last tile (workgroup-uniform branch; the one register set of this kernel
                // is waited for as a whole anyway): k rows at or past K are redirected to row K - 1 -- always valid memory -- and zeroed
                // element by element when the registers are split.  (K = T * N rows of a rollout: 50 rows x an odd step count.)
                const int kb = inside ? k0 + kq0 : k0;
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = *(gcf4)(src + (size_t)min(kb + j, K - 1) * ld4 + voff0);
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    v[j] = *(gcf4)(b + off);
                    b += ld4;
                }
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) r[4 * i + j] = v[j][i];       // quad i = row 4mq+i, k = 4kg..4kg+3
        }
    }
};

// registers of one 128-row block -> three bf16 planes (plane pitch PLANE elements) at dst = plane 0, first row of the block
template <bool KC, bool EDGE, int PLANE>
__device__ __forceinline__ void w_r2s(const float (&r)[16], unsigned short *dst, int tid, int valid_rows, int valid_k) {
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        int row, kq;
        if (KC) {
            const int idx = p * NT + tid;
            row = idx / 8;
            kq = (idx % 8) * 4;
        } else {
            const int lane = tid & 63;
            row = 4 * ((tid >> 6) * 8 + (lane & 7)) + p;
            kq = 4 * (lane >> 3);
        }
        // element j of the quad is k = kq + j (K % 4 != 0 is allowed for [K][rows] operands: the last quad may be cut)
        uint32_t h[4], m[4], l[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float fac = (!EDGE || (row < valid_rows && kq + j < valid_k)) ? 1.f : 0.f;
            const float x = EDGE ? r[4 * p + j] * fac : r[4 * p + j];
            h[j] = fbits(x);
            const float r1 = x - bfloat(h[j] & 0xffff0000u);            // exact
            m[j] = fbits(r1);
            l[j] = fbits(r1 - bfloat(m[j] & 0xffff0000u));              // exact, <= 8 significant bits
        }
        unsigned short *o = dst + wswz(row, kq);
        *reinterpret_cast<u32x2 *>(o) = u32x2{pack2(h[0], h[1]), pack2(h[2], h[3])};
        *reinterpret_cast<u32x2 *>(o + PLANE) = u32x2{pack2(m[0], m[1]), pack2(m[2], m[3])};
        *reinterpret_cast<u32x2 *>(o + 2 * PLANE) = u32x2{pack2(l[0], l[1]), pack2(l[2], l[3])};
    }
}

__device__ __forceinline__ Unit wunit_of(const KArgs &a, int u, int gm, int gn) {
    const int per = gm * gn;
    {   // XCD-aware order (see gemm_x3.hip unit_of): every XCD walks a contiguous run of output tiles
        const int G = gridDim.x, total = per * a.splits;
        const int b = u % G, r = u / G;
        if ((G & 7) == 0 && (r + 1) * G <= total) u = (b & 7) * (G >> 3) + (b >> 3) + r * G;
    }
    const int z = u / per, tile = u - z * per;
    Unit r;
    r.z = z;
    r.m0 = (tile / gn) * WBM;
    r.n0 = (tile % gn) * WBN;
    r.t_begin = (int)(((long long)a.tiles_total * z) / a.splits);
    r.nt = (int)(((long long)a.tiles_total * (z + 1)) / a.splits) - r.t_begin;
    return r;
}

// uniform facts of one staged K tile, carried beside its register set
struct WEdge {
    int va0, va1, vb, vk;     // valid rows of A block 0 / 1 and of B, valid k of the tile
    bool edge;                // any of them short of a full block
};

// DOA / DOB: this wave stages the A blocks / the B block (NSW = 4: one staging wave per SIMD does both; NSW = 8: two per SIMD, the
// first takes A -- two thirds of the split -- the second B)
template <bool AKC, bool BKC, bool DOA, bool DOB>
__device__ __forceinline__ void w_staging(const KArgs &a, int gm, int gn, int units, int tid, unsigned short *smem) {
    // ONE register set (A block 0, A block 1, B: 48 floats per thread): the loads of K tile g+2 are issued right behind the split of
    // tile g+1 and consumed a whole K-tile period later -- >= 3 072 MFMA cycles per SIMD, the prefetch distance the 128 x 128
    // kernel gets from two sets (2 x 1 536).  Two sets (96 floats + 12 offsets + the split's temporaries) do not fit the 168
    // registers that 3 waves per SIMD leave: the compiler spilled, and every wait degenerated to vmcnt(0).
    float ra[16], rc[16], rb[16];
    WEdge e{};
    WStager<AKC> sa0, sa1;
    WStager<BKC> sb;
    auto bind = [&](int sidx, int m0_, int n0_) {
#define CAPMI_X3W_SEG(I)                                                  \
    case I:                                                               \
        sa0.init(a.seg[I].A, a.seg[I].lda, a.seg[I].K);                   \
        sa1.init(a.seg[I].A, a.seg[I].lda, a.seg[I].K);                   \
        sb.init(a.seg[I].B, a.seg[I].ldb, a.seg[I].K);                    \
        break;
        switch (sidx) {
            CAPMI_X3W_SEG(0) CAPMI_X3W_SEG(1) CAPMI_X3W_SEG(2) CAPMI_X3W_SEG(3)
        }
#undef CAPMI_X3W_SEG
        sa0.set_tile(m0_, a.M, tid);
        sa1.set_tile(m0_ + 128, a.M, tid);
        sb.set_tile(n0_, a.N, tid);
    };
    int fu = blockIdx.x, f_left = 0, f_sleft = 0, f_k0 = 0, f_s = 0, f_m0 = 0, f_n0 = 0;
    auto open_unit = [&]() {
        const Unit un = wunit_of(a, fu, gm, gn);
        int sidx, k0;
        locate(a, un.t_begin, sidx, k0);
        f_s = __builtin_amdgcn_readfirstlane(sidx);
        f_k0 = k0; f_left = un.nt; f_m0 = un.m0; f_n0 = un.n0;
        bind(f_s, f_m0, f_n0);
        f_sleft = (sa0.K - k0 + BK - 1) / BK;
    };
    if (fu < units) open_unit();
    else bind(0, 0, 0);                                  // (never fetched for real: steps == 0)
    auto fetch = [&](float (&xa0)[16], float (&xa1)[16], float (&xb)[16], WEdge &ed) {
        if (DOA) sa0.fetch(xa0, f_k0);
        if (DOA) sa1.fetch(xa1, f_k0);
        if (DOB) sb.fetch(xb, f_k0);
        ed.va0 = sa0.valid_rows; ed.va1 = sa1.valid_rows; ed.vb = sb.valid_rows;
        ed.vk = min(sa0.K - f_k0, BK);
        ed.edge = (DOA && (ed.va0 < 128 || ed.va1 < 128)) || (DOB && ed.vb < 128) || ed.vk < BK;
        if (f_left > 0) {                              // workgroup-uniform
            if (--f_left == 0) {
                fu += gridDim.x;
                if (fu < units) open_unit();
            } else if (--f_sleft == 0) {
                bind(++f_s, f_m0, f_n0);
                f_k0 = 0;
                f_sleft = (sa0.K + BK - 1) / BK;
            } else {
                f_k0 += BK;
            }
        }
    };
    int steps = 0;
    for (int u = blockIdx.x; u < units; u += gridDim.x) steps += wunit_of(a, u, gm, gn).nt;
    auto store = [&](const float (&xa0)[16], const float (&xa1)[16], const float (&xb)[16], const WEdge &ed, int g) {
        unsigned short *st = smem + (g & 1) * WSTAGE;
        if (ed.edge) {
            if (DOA) w_r2s<AKC, true, WPL_A>(xa0, st, tid, ed.va0, ed.vk);
            if (DOA) w_r2s<AKC, true, WPL_A>(xa1, st + 128 * 32, tid, ed.va1, ed.vk);
            if (DOB) w_r2s<BKC, true, WPL_B>(xb, st + 3 * WPL_A, tid, ed.vb, ed.vk);
        } else {
            if (DOA) w_r2s<AKC, false, WPL_A>(xa0, st, tid, 128, BK);
            if (DOA) w_r2s<AKC, false, WPL_A>(xa1, st + 128 * 32, tid, 128, BK);
            if (DOB) w_r2s<BKC, false, WPL_B>(xb, st + 3 * WPL_A, tid, 128, BK);
        }
    };
    fetch(ra, rc, rb, e);                              // step 0
    if (steps > 0) store(ra, rc, rb, e, 0);
    fetch(ra, rc, rb, e);                              // step 1
    __syncthreads();                                   // stage 0 ready
    for (int g = 0; g < steps; ++g) {
        if (g + 1 < steps) store(ra, rc, rb, e, g + 1);   // into the stage the MFMA waves released at the previous barrier
        fetch(ra, rc, rb, e);                             // step g + 2
        __syncthreads();
    }
}

// One K tile of an MFMA wave: two halves (ks) of 12 ds_read_b128 + 24 MFMAs.  six of the nine cross terms, small ones first (planes:
// 0 = h, 1 = m, 2 = l) -- the order of gemm_x3.hip, so that a K slice sums to the same bits in every fat kernel.
// r6: the workgroup barrier of the K tile sits BETWEEN the second half's reads and its MFMAs (as in gemm_x3.hip's pipelined loop), not
// behind them.  Two MFMA waves share a SIMD and alternate -- one multiplies while the other waits for its fragments -- so they are
// half a phase apart; a barrier at the END of the tile made the leading wave wait a whole MFMA phase (768 cycles of an idle matrix
// pipe per K tile and SIMD, ~20 % of the tile: CAPMI_GROUP_ABLATE staging-off run 2.08 us per tile against 1.28 of MFMA work).  At the
// early barrier every wave still holds 24 MFMAs to issue, so the pipe has work queued while the waves wait for each other; the stage
// is released to the staging waves 24 MFMAs sooner too.  `late` (research switch) restores the barrier at the end.
__device__ __forceinline__ void w_ktile(f32x16 (&acc)[2][2], const unsigned short *As, const unsigned short *Bs, const int (&offA)[2][2],
                                        const int (&offB)[2][2], bool late, bool no_mfma) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        bf16x8 av[2][3], bv[2][3];
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) {
                av[q][pl] = *reinterpret_cast<const bf16x8 *>(As + pl * WPL_A + offA[q][ks]);
                bv[q][pl] = *reinterpret_cast<const bf16x8 *>(Bs + pl * WPL_B + offB[q][ks]);
            }
        if (ks == 1 && !late) __syncthreads();             // every read of this stage has landed (the barrier's lgkmcnt(0)): stage released,
                                                           // next stage ready
        if (no_mfma) continue;                             // (profiling ablation)
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                acc[q][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[q][2], bv[j][0], acc[q][j], 0, 0, 0);
                acc[q][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[q][0], bv[j][2], acc[q][j], 0, 0, 0);
            }
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                acc[q][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[q][1], bv[j][1], acc[q][j], 0, 0, 0);
                acc[q][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[q][1], bv[j][0], acc[q][j], 0, 0, 0);
            }
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                acc[q][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[q][0], bv[j][1], acc[q][j], 0, 0, 0);
                acc[q][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[q][0], bv[j][0], acc[q][j], 0, 0, 0);
            }
    }
    if (late) __syncthreads();                             // stage g&1 released, stage (g+1)&1 ready
}

// Waves 0-7 are MFMA waves (64 x 64 each: 4 along M x 2 along N), waves 8.. staging waves.  Waves are dealt to the SIMDs round
// robin, so every SIMD holds two MFMA waves and NSW / 4 staging waves.
template <bool AKC, bool BKC, int NSW>
__global__ __launch_bounds__(512 + 64 * NSW) void gemm_x3w_kernel(const KArgs a, int gm, int gn, int prio) {
    extern __shared__ __attribute__((aligned(16))) unsigned short smem[];      // 2 stages = 144 KB
    const int units = gm * gn * a.splits;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;

    if (wid >= 8) {
        // (experiment knob CAPMI_X3W_PRIO: the staging waves are the youngest of their SIMD and lose the VALU arbitration by age)
        if ((prio & 3) == 1) __builtin_amdgcn_s_setprio(1);
        else if ((prio & 3) == 2) __builtin_amdgcn_s_setprio(2);
        else if ((prio & 3) == 3) __builtin_amdgcn_s_setprio(3);
        if (NSW == 4) w_staging<AKC, BKC, true, true>(a, gm, gn, units, threadIdx.x - 512, smem);
        else if (wid < 12) w_staging<AKC, BKC, true, false>(a, gm, gn, units, threadIdx.x - 512, smem);
        else w_staging<AKC, BKC, false, true>(a, gm, gn, units, threadIdx.x - 768, smem);
        return;
    }
    const int wm0 = (wid >> 1) * 64, wn0 = (wid & 1) * 64;
    const int l31 = lane & 31, half = lane >> 5;
    // fragment offsets inside a plane (elements), fixed for the whole kernel: [q][ks]
    int offA[2][2], offB[2][2];
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            offA[q][ks] = wswz(wm0 + 32 * q + l31, 16 * ks + 8 * half);
            offB[q][ks] = wswz(wn0 + 32 * q + l31, 16 * ks + 8 * half);
        }
    __syncthreads();                                       // stage 0 ready
    int g = 0;
    for (int u = blockIdx.x; u < units; u += gridDim.x) {
        const Unit un = wunit_of(a, u, gm, gn);
        f32x16 acc[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        for (int i = 0; i < un.nt; ++i, ++g) {
            const unsigned short *As = smem + (g & 1) * WSTAGE, *Bs = As + 3 * WPL_A;
            w_ktile(acc, As, Bs, offA, offB, (prio & 4) ? true : (prio & 8) ? wid < 4 : false, false);
        }
        if (a.transposed) x3_epilogue_t<2>(a, un, acc, wm0, wn0, l31, half);
        else x3_epilogue<2>(a, un, acc, wm0, wn0, l31, half);
    }
}


// ============================================================ r6: GROUPED launch ============================================================
// N independent GEMMs (the weight gradients dW_i = dY_i^T X_i that nothing reads before the optimizer) as ONE persistent grid: the
// units of all table entries form one list, a workgroup walks every gridDim.x-th unit and its staging waves run ahead across entry
// boundaries exactly as across the units of one GEMM.  Why (profiles/r05_scst_kernel_stats.md, r05_txe_kernel_stats.md): launched one
// by one these GEMMs are sub-wave grids -- [4000 x 1000] is 128 wide tiles on 256 CUs, [512 x 512] is 8 -- so each either ran on the
// 128 x 128 kernel (MFMA busy 0.45 against 0.59) or needed a 16-32-way K split with its slab traffic, and each paid its own
// prologue / tail.  In a group the full rounds run whole-K tiles straight into C; only the LAST partial round of a launch is cut
// into K slices (<= 256 pieces of 128 KB) so that it spreads over all CUs, and one small launch sums those pieces.
// The table travels by value in the kernel arguments (s_load with a computed offset: no device table, no upload, capturable).

// The table is read THROUGH THE KERNEL-ARGUMENT SEGMENT (address space 4, scalar loads with a computed offset): taken from the by-value
// parameter the compiler copied all 3.9 KB of it into scratch first (a dynamically indexed aggregate; .private_segment_fixed_size 3896).
typedef const GTab __attribute__((address_space(4))) *GTabK;
typedef const GItem __attribute__((address_space(4))) *GItemK;

#ifdef CAPMI_VARIANTS
// CAPMI_GROUP_TRACE=1 (variants build): workgroup 0's first MFMA wave stamps s_memtime behind every K tile, before and behind every
// epilogue; launch_x3w_group synchronises and prints the gaps (where does a unit spend time outside its K loop?)
__device__ unsigned long long g_group_trace[4096];
#define GROUP_STAMP(i) do { const int i_ = (i); if (trace && i_ < 4096) g_group_trace[i_] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define GROUP_STAMP(i) do { } while (0)
#endif

struct GUnit {
    int e;                  // table entry
    int m0, n0, t_begin, nt, z, tl;
};

// the unit a workgroup handles in round r (XCD-aware inside full rounds, see wunit_of)
__device__ __forceinline__ int g_unit_index(int r, int units) {
    const int G = gridDim.x, b = blockIdx.x;
    if ((G & 7) == 0 && (r + 1) * G <= units) return (b & 7) * (G >> 3) + (b >> 3) + r * G;
    return b + r * G;
}

// `e` only moves forward: a workgroup's unit indices grow with the round
__device__ __forceinline__ GUnit g_unit_of(GTabK t, int u, int &e) {
    while (e + 1 < t->n && u >= t->it[e + 1].unit0) ++e;
    GItemK it = &t->it[e];
    const int local = u - it->unit0;
    GUnit r;
    r.e = e;
    r.z = local / it->ntiles;
    r.tl = local - r.z * it->ntiles;
    const int tile = it->tile0 + r.tl;
    r.m0 = (tile / it->gn) * WBM;
    r.n0 = (tile % it->gn) * WBN;
    r.t_begin = (int)(((long long)it->kt * r.z) / it->splits);
    r.nt = (int)(((long long)it->kt * (r.z + 1)) / it->splits) - r.t_begin;
    return r;
}

// uniform facts of one staged K tile of a GROUPED launch: WEdge + where the tile's column sums go
struct GEdge {
    WEdge w;
    float *cs;                // null: no column sums from this tile; else the destination of the unit's 256 sums (tile origin)
    int cs_cols;              // valid columns from the tile origin
    bool last;                // last K tile of its unit: the sums are complete
};

template <bool AKC, bool BKC, bool DOA, bool DOB>
__device__ __forceinline__ void wg_staging(GTabK t, int tid, unsigned short *smem) {
    float ra[16], rc[16], rb[16];
    GEdge e{};
    WStager<AKC> sa0, sa1;
    WStager<BKC> sb;
    const int G = gridDim.x;
    int steps = 0, rounds = 0;
    {
        int ent = 0;
        for (int r = 0; blockIdx.x + r * G < t->units; ++r) {
            steps += g_unit_of(t, g_unit_index(r, t->units), ent).nt;
            ++rounds;
        }
    }
    int f_ent = 0, f_round = 0, f_left = 0, f_k0 = 0, f_cols = 0;
    float *f_cs = nullptr;
    auto open_unit = [&]() {
        const GUnit un = g_unit_of(t, g_unit_index(f_round, t->units), f_ent);
        GItemK it = &t->it[un.e];
        sa0.init(it->A, it->lda, it->K);
        sa1.init(it->A, it->lda, it->K);
        sb.init(it->B, it->ldb, it->K);
        sa0.set_tile(un.m0, it->M, tid);
        sa1.set_tile(un.m0 + 128, it->M, tid);
        sb.set_tile(un.n0, it->N, tid);
        f_k0 = un.t_begin * BK;
        f_left = un.nt;
        // column sums of A: by the units of the FIRST column tile only (every column tile stages the same A panel)
        f_cs = nullptr;
        if (DOA && !AKC && it->cs && un.n0 == 0) {
            f_cs = it->splits > 1 ? it->slab + (size_t)it->ntiles * it->splits * (size_t)(WBM * WBN) + ((size_t)un.tl * it->splits + un.z) * WBM
                                  : it->cs + un.m0;
            f_cols = it->M - un.m0;
        }
    };
    if (rounds > 0) open_unit();
    else {                                                 // (never fetched for real: steps == 0)
        GItemK it = &t->it[0];
        sa0.init(it->A, it->lda, it->K); sa1.init(it->A, it->lda, it->K); sb.init(it->B, it->ldb, it->K);
        sa0.set_tile(0, it->M, tid); sa1.set_tile(0, it->M, tid); sb.set_tile(0, it->N, tid);
    }
    auto fetch = [&](float (&xa0)[16], float (&xa1)[16], float (&xb)[16], GEdge &ed) {
        if (DOA) sa0.fetch(xa0, f_k0);
        if (DOA) sa1.fetch(xa1, f_k0);
        if (DOB) sb.fetch(xb, f_k0);
        ed.w.va0 = sa0.valid_rows; ed.w.va1 = sa1.valid_rows; ed.w.vb = sb.valid_rows;
        ed.w.vk = min(sa0.K - f_k0, BK);
        ed.w.edge = (DOA && (ed.w.va0 < 128 || ed.w.va1 < 128)) || (DOB && ed.w.vb < 128) || ed.w.vk < BK;
        ed.cs = f_cs; ed.cs_cols = f_cols; ed.last = f_left == 1;
        if (f_left > 0) {                              // workgroup-uniform
            if (--f_left == 0) {
                if (++f_round < rounds) open_unit();
            } else {
                f_k0 += BK;
            }
        }
    };
    // running column sums of this thread's 2 x 4 columns (k-major A: thread (kg, mq) holds k = 4kg..4kg+3 of columns 4mq..4mq+3 of each
    // 128-column block, r[4 i + j] = A[k0 + 4kg + j][4mq + i]); summed over the unit's K tiles in order, then over kg by lane shuffles
    float cs0[4] = {0.f, 0.f, 0.f, 0.f}, cs1[4] = {0.f, 0.f, 0.f, 0.f};
    auto colsum_step = [&](const float (&xa0)[16], const float (&xa1)[16], const GEdge &ed) {
        const int lane = tid & 63, kq = 4 * (lane >> 3);
        float f[4];                                                          // k = kq + j inside K?  (all ones except in the last tile)
#pragma unroll
        for (int j = 0; j < 4; ++j) f[j] = kq + j < ed.w.vk ? 1.f : 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            cs0[i] += (xa0[4 * i] * f[0] + xa0[4 * i + 1] * f[1]) + (xa0[4 * i + 2] * f[2] + xa0[4 * i + 3] * f[3]);
            cs1[i] += (xa1[4 * i] * f[0] + xa1[4 * i + 1] * f[1]) + (xa1[4 * i + 2] * f[2] + xa1[4 * i + 3] * f[3]);
        }
        if (!ed.last) return;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
            for (int sh = 8; sh < 64; sh <<= 1) {
                cs0[i] += __shfl_xor(cs0[i], sh);
                cs1[i] += __shfl_xor(cs1[i], sh);
            }
        }
        if (lane < 8) {
            const int c = 4 * ((tid >> 6) * 8 + lane);                        // first of this thread's four columns inside a block
            if (c < ed.cs_cols) *reinterpret_cast<f32x4 *>(ed.cs + c) = f32x4{cs0[0], cs0[1], cs0[2], cs0[3]};       // M % 4 == 0
            if (128 + c < ed.cs_cols) *reinterpret_cast<f32x4 *>(ed.cs + 128 + c) = f32x4{cs1[0], cs1[1], cs1[2], cs1[3]};
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) cs0[i] = cs1[i] = 0.f;
    };
    auto store = [&](const float (&xa0)[16], const float (&xa1)[16], const float (&xb)[16], const GEdge &ed, int g) {
        unsigned short *st = smem + (g & 1) * WSTAGE;
        if (DOA && !AKC && ed.cs) colsum_step(xa0, xa1, ed);
        if (t->reserved & 4) return;                      // (ablation: no split, no LDS stores)
        if (ed.w.edge) {
            if (DOA) w_r2s<AKC, true, WPL_A>(xa0, st, tid, ed.w.va0, ed.w.vk);
            if (DOA) w_r2s<AKC, true, WPL_A>(xa1, st + 128 * 32, tid, ed.w.va1, ed.w.vk);
            if (DOB) w_r2s<BKC, true, WPL_B>(xb, st + 3 * WPL_A, tid, ed.w.vb, ed.w.vk);
        } else {
            if (DOA) w_r2s<AKC, false, WPL_A>(xa0, st, tid, 128, BK);
            if (DOA) w_r2s<AKC, false, WPL_A>(xa1, st + 128 * 32, tid, 128, BK);
            if (DOB) w_r2s<BKC, false, WPL_B>(xb, st + 3 * WPL_A, tid, 128, BK);
        }
    };
    fetch(ra, rc, rb, e);                              // step 0
    if (steps > 0) store(ra, rc, rb, e, 0);
    fetch(ra, rc, rb, e);                              // step 1
    __syncthreads();                                   // stage 0 ready
    for (int g = 0; g < steps; ++g) {
        if (g + 1 < steps) store(ra, rc, rb, e, g + 1);
        fetch(ra, rc, rb, e);
        __syncthreads();
    }
}

// one wave's 64 x 64 block of a finished unit -> memory (C/D layout of the 32x32 MFMA: column = lane & 31, rows (x & 3) + 8 (x >> 2) +
// 4 (lane >> 5)); out points at the tile's origin
template <bool ADD, bool EDGE, bool NT = false>
__device__ __forceinline__ void g_store(const f32x16 (&acc)[2][2], float *out, int ldo, int vr, int vc, int wm0, int wn0, int l31, int half) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = wn0 + 32 * j + l31;
            const int ccol = EDGE ? min(col, vc - 1) : col;
            float *o = out + (size_t)(wm0 + 32 * i + 4 * half) * ldo + ccol;
            const int r0 = wm0 + 32 * i + 4 * half;
#pragma unroll
            for (int xg = 0; xg < 16; xg += 4) {                 // (four rows at a time: 16 more live registers spilled)
                float old[4];
                if (ADD) {
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int dr = u + 2 * xg;                          // = ((xg + u) & 3) + 8 * ((xg + u) >> 2)
                        const int cdr = EDGE ? min(r0 + dr, vr - 1) - r0 : dr;          // (clamped: always a valid address)
                        old[u] = o[(ptrdiff_t)cdr * ldo];
                    }
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int dr = u + 2 * xg;
                    const float v = ADD ? acc[i][j][xg + u] + old[u] : acc[i][j][xg + u];
                    if (!EDGE || (col < vc && r0 + dr < vr)) {
                        if (NT) __builtin_nontemporal_store(v, o + (size_t)dr * ldo);
                        else o[(size_t)dr * ldo] = v;
                    }
                }
            }
        }
}

template <bool AKC, bool BKC>
__global__ __launch_bounds__(1024) void gemm_x3w_group_kernel(const GTab table) {
    GTabK t = (GTabK)__builtin_amdgcn_kernarg_segment_ptr();              // = &table, where the dispatch put it
    extern __shared__ __attribute__((aligned(16))) unsigned short smem[];      // 2 stages = 144 KB
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    if (wid >= 8) {
        if (wid < 12) wg_staging<AKC, BKC, true, false>(t, threadIdx.x - 512, smem);
        else wg_staging<AKC, BKC, false, true>(t, threadIdx.x - 768, smem);
        return;
    }
    const int wm0 = (wid >> 1) * 64, wn0 = (wid & 1) * 64;
    const int l31 = lane & 31, half = lane >> 5;
    int offA[2][2], offB[2][2];
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            offA[q][ks] = wswz(wm0 + 32 * q + l31, 16 * ks + 8 * half);
            offB[q][ks] = wswz(wn0 + 32 * q + l31, 16 * ks + 8 * half);
        }
    __syncthreads();                                       // stage 0 ready
    const int abl = t->reserved;                            // (CAPMI_GROUP_ABLATE, variants builds: profiling ablations; 0 otherwise)
#ifdef CAPMI_VARIANTS
    const bool trace = (abl & 64) && blockIdx.x == 0 && threadIdx.x == 0;
    int ts = 0;
#endif
    int g = 0, ent = 0;
    for (int r = 0; blockIdx.x + r * (int)gridDim.x < t->units; ++r) {
        const GUnit un = g_unit_of(t, g_unit_index(r, t->units), ent);
        f32x16 acc[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int x = 0; x < 16; ++x) acc[i][j][x] = 0.f;
        GROUP_STAMP(ts++);                                  // unit start
        for (int i = 0; i < un.nt; ++i, ++g) {
            const unsigned short *As = smem + (g & 1) * WSTAGE, *Bs = As + 3 * WPL_A;
            w_ktile(acc, As, Bs, offA, offB, (abl & 8) ? true : (abl & 16) ? wid < 4 : false, (abl & 2) != 0);
            GROUP_STAMP(ts++);                              // behind K tile i
        }
        // ---- epilogue: whole-K units write (or add to) C; K slices leave their [256 x 128] piece in the entry's slab
        GItemK it = &t->it[un.e];
        const bool piece = it->splits > 1;
        float *out = piece ? it->slab + ((size_t)un.tl * it->splits + un.z) * (size_t)(WBM * WBN)
                           : it->C + (size_t)un.m0 * it->ldc + un.n0;
        const int ldo = piece ? WBN : it->ldc;
        const int vr = it->M - un.m0, vc = it->N - un.n0;                 // valid rows / columns of the tile (may exceed 256 / 128)
        const bool add = !piece && it->accumulate;
        // four straight-line variants behind workgroup-uniform branches (a per-element `add ? load : -` and per-element bounds tests
        // compiled to a branch and an s_waitcnt vmcnt(0) in front of every one of the 64 stores)
        const bool edge = vr < WBM || vc < WBN;
        if (abl & 1) continue;                            // (ablation: the K loop without the epilogue)
        if (add) {
            if (edge) g_store<true, true>(acc, out, ldo, vr, vc, wm0, wn0, l31, half);
            else g_store<true, false>(acc, out, ldo, vr, vc, wm0, wn0, l31, half);
        } else if (abl & 32) {                            // (research: streaming stores)
            if (edge) g_store<false, true, true>(acc, out, ldo, vr, vc, wm0, wn0, l31, half);
            else g_store<false, false, true>(acc, out, ldo, vr, vc, wm0, wn0, l31, half);
        } else {
            if (edge) g_store<false, true>(acc, out, ldo, vr, vc, wm0, wn0, l31, half);
            else g_store<false, false>(acc, out, ldo, vr, vc, wm0, wn0, l31, half);
        }
        GROUP_STAMP(ts++);                                  // behind the epilogue's stores (issued, not landed)
    }
}

// blockIdx.x = one split tile of the launch: C tile (+)= sum of its `splits` pieces in slice order -- the order of splitk_reduce_kernel
// (gemm_f32.hip), so that a grouped GEMM and the same GEMM launched alone through capmi_gemm_f32 with the same K split agree bit for bit.
__global__ __launch_bounds__(256) void x3w_group_reduce_kernel(const GTab table) {
    GTabK t = (GTabK)__builtin_amdgcn_kernarg_segment_ptr();
    int e = 0;
    while (e < t->n && !(t->it[e].splits > 1 && (int)blockIdx.x >= t->it[e].rtile0 && (int)blockIdx.x < t->it[e].rtile0 + t->it[e].ntiles)) ++e;
    if (e >= t->n) return;
    GItemK it = &t->it[e];
    const int tl = blockIdx.x - it->rtile0, tile = it->tile0 + tl;
    const int m0 = (tile / it->gn) * WBM, n0 = (tile % it->gn) * WBN;
    const int vr = min(it->M - m0, WBM), vc = min(it->N - n0, WBN);
    const float *p = it->slab + (size_t)tl * it->splits * (size_t)(WBM * WBN);
    for (int q = threadIdx.x; q < WBM * WBN / 4; q += 256) {
        const int row = q / (WBN / 4), c4 = (q % (WBN / 4)) * 4;
        if (row >= vr || c4 >= vc) continue;                 // (N % 4 == 0: a quad of columns is entirely inside or outside)
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        for (int s0 = 0; s0 < it->splits; s0 += 8) {         // 8 independent loads in flight, summed in slice order
            f32x4 tv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u)
                tv[u] = (s0 + u < it->splits) ? *reinterpret_cast<const f32x4 *>(p + (size_t)(s0 + u) * (WBM * WBN) + (size_t)row * WBN + c4)
                                              : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int u = 0; u < 8; ++u) v += tv[u];
        }
        f32x4 *o = reinterpret_cast<f32x4 *>(it->C + (size_t)(m0 + row) * it->ldc + n0 + c4);
        if (it->accumulate) v += *o;
        *o = v;
    }    // the column sums of a K-sliced first-column tile: [splits][256] pieces behind the entry's tile pieces, summed in slice order
    if (it->cs && n0 == 0 && (int)threadIdx.x < vr) {
        const float *q = it->slab + (size_t)it->ntiles * it->splits * (size_t)(WBM * WBN) + (size_t)tl * it->splits * WBM + threadIdx.x;
        float v = 0.f;
        for (int z = 0; z < it->splits; ++z) v += q[(size_t)z * WBM];
        it->cs[m0 + threadIdx.x] = v;
    }
}

}  // namespace

int launch_x3w(const KArgs &a, int a_layout, int b_layout, dim3 tiles, hipStream_t st, int pcls, double bytes, double flops) {
    // tiles = (gn, gm, splits) of the 256 x 128 tiling; persistent grid: one workgroup per CU walks the unit list
    const int gn = tiles.x, gm = tiles.y;
    const int units = gn * gm * a.splits;
    const dim3 grid(units < 256 ? units : 256);
    hipEvent_t e0, e1;
    const bool prof = capmi_prof::take_events(pcls, &e0, &e1, bytes, flops);
    constexpr size_t lds = 2 * (size_t)WSTAGE * sizeof(unsigned short);
    static_assert(lds <= 160 * 1024, "two stages must fit the CU's LDS");
    // (research switches of a -DCAPMI_VARIANTS build; the product build compiles the measured best: 8 staging waves, priority 0)
    static const int env_nsw = capmi::research("CAPMI_X3W_NSW", 8), env_prio = capmi::research("CAPMI_X3W_PRIO", 0);
#define CAPMI_X3W_N(AK, BK_, NSW_)                                                                              \
    do {                                                                                                        \
        static bool attr_set = false;                                                                           \
        if (!attr_set) {                                                                                        \
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&gemm_x3w_kernel<AK, BK_, NSW_>),          \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                    \
            attr_set = true;                                                                                    \
        }                                                                                                       \
        const dim3 blk(512 + 64 * NSW_);                                                                        \
        if (prof) hipExtLaunchKernelGGL((gemm_x3w_kernel<AK, BK_, NSW_>), grid, blk, lds, st, e0, e1, 0, a, gm, gn, env_prio); \
        else hipLaunchKernelGGL((gemm_x3w_kernel<AK, BK_, NSW_>), grid, blk, lds, st, a, gm, gn, env_prio);     \
    } while (0)
#define CAPMI_X3W_GO(AK, BK_)                                                                                   \
    do {                                                                                                        \
        if (env_nsw == 8) CAPMI_X3W_N(AK, BK_, 8);                                                              \
        else CAPMI_X3W_N(AK, BK_, 4);                                                                           \
    } while (0)
    if (a_layout == 0 && b_layout == 0) CAPMI_X3W_GO(true, true);
    else if (a_layout == 0 && b_layout == 1) CAPMI_X3W_GO(true, false);
    else if (a_layout == 1 && b_layout == 1) CAPMI_X3W_GO(false, false);
    else CAPMI_X3W_GO(false, true);
#undef CAPMI_X3W_GO
#undef CAPMI_X3W_N
    CAPMI_CHECK_LAUNCH();
    return 0;
}

int launch_x3w_group(const GTab &t, hipStream_t st, int pcls, double bytes, double flops) {
    if (t.n < 1 || t.n > GROUP_MAX || t.units < 1) return CAPMI_EINVAL;
    const dim3 grid(t.units < 256 ? t.units : 256);
    hipEvent_t e0, e1;
    const bool prof = capmi_prof::take_events(pcls, &e0, &e1, bytes, flops);
    constexpr size_t lds = 2 * (size_t)WSTAGE * sizeof(unsigned short);
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&gemm_x3w_group_kernel<false, false>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    if (prof) hipExtLaunchKernelGGL((gemm_x3w_group_kernel<false, false>), grid, dim3(1024), lds, st, e0, e1, 0, t);
    else hipLaunchKernelGGL((gemm_x3w_group_kernel<false, false>), grid, dim3(1024), lds, st, t);
    CAPMI_CHECK_LAUNCH();
#ifdef CAPMI_VARIANTS
    if (t.reserved & 64) {
        (void)hipStreamSynchronize(st);
        static unsigned long long h[4096];
        (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_group_trace), sizeof(h));
        // unit layout of workgroup 0: [start, K tiles..., after epilogue] per unit; units of workgroup 0 = rounds
        int idx = 0;
        fprintf(stderr, "capmi group trace (workgroup 0, s_memtime ticks):\n");
        for (int r = 0; r * (int)grid.x < t.units && idx < 4000; ++r) {
            // nt of this unit is not known here: print until the next start marker by scanning for the largest gap pattern is fragile --
            // so print raw deltas, 16 per line, with the index
            (void)r;
        }
        for (int i = 1; i < 4096 && h[i]; ++i) {
            if ((i - 1) % 16 == 0) fprintf(stderr, "\n%5d:", i);
            fprintf(stderr, " %6llu", h[i] - h[i - 1]);
        }
        fprintf(stderr, "\n");
    }
#endif
    if (t.rtiles > 0) {
        hipLaunchKernelGGL(x3w_group_reduce_kernel, dim3(t.rtiles), dim3(256), 0, st, t);
        CAPMI_CHECK_LAUNCH();
    }
    return 0;
}

}  // namespace capmi_gemm
