// fp32 GEMM on the bf16 matrix pipe of gfx950 by exact 3-way operand splitting ("bf16x3"), for the FAT GEMMs of the
// BPTT (time-batched weight gradients dW = dG^T X and batched input gradients dX = dG W: M, N >= 128, K >= 1000).
//
// Why: CDNA4 has no xf32/TF32; v_mfma_f32_32x32x2_f32 runs at the f32 VECTOR rate (157 TFLOP/s, 1/16 of the bf16
// rate).  Every fp32 number is EXACTLY the sum of three bf16 numbers obtained by truncation,
//     x = h + m + l,   h = top 8 mantissa bits of x,  m = top 8 bits of (x - h),  l = x - h - m  (<= 8 bits left),
// and a product of two bf16 values is exact in fp32, so
//     a*b = ah*bh + (ah*bm + am*bh) + (am*bm + ah*bl + al*bh) + [am*bl + al*bm + al*bl],
// where the bracket is <= 3 * 2^-24 |a||b| -- the size of ONE fp32 rounding of the product.  Dropping it and feeding
// the other six terms to v_mfma_f32_32x32x16_bf16 (fp32 accumulate) gives fp32-grade results: against an fp64 reference
// the error is the same as the exact-fp32 MFMA kernel's and below the vendor fp32 matmul's, also on operands with a
// wide dynamic range (tests/test_kernels_gpu.py::test_gemm_fat_bf16x3_is_fp32_grade, tools_x3_accuracy.py) -- at 6/16
// of the matrix-pipe time.  The split costs ~5 VALU ops per element and is done ONCE per element per workgroup, on the
// way from HBM to LDS; the decode-step GEMMs (each weight used by one wave only) stay on the exact fp32 pipe.
// CAPMI_GEMM_X3=0 routes the fat GEMMs back to the exact-fp32 kernel.
//
// Layout: 128x128x32 tile per workgroup.  LDS holds three bf16 planes per operand, [128 rows][32 k]; r5: rows are 64 bytes, their
// 16-byte pieces XOR-swizzled (wswz, gemm_x3_common.h) -- the 80-byte padded rows of rounds 2-4 kept the ds_read_b128 of a lane's
// MFMA operand conflict-free but not the staging stores (PMC, profiles/r05_fat_gemm_wide.md: SQ_LDS_BANK_CONFLICT = 46 % of the
// LDS-active cycles against 13 % for the swizzled 256 x 128 kernel).  A staged quad (4 consecutive k of one row, 3 planes) is
// three ds_write_b64.  K-major sources ([K][M] gradients / activations
// of dW = dG^T X) are fetched as 4x4 blocks (16-byte loads along the contiguous dimension) and transposed in
// registers, so the same row-quad store applies without an LDS transpose.
#include "gemm_x3_common.h"
#include "profile.h"
#include <hip/hip_ext.h>
#include <stdlib.h>

namespace capmi_gemm {
namespace {

constexpr int XBM = 128, XBN = 128, XP = 32;       // r5: 64-byte rows, XOR-swizzled (wswz, gemm_x3_common.h) instead of padded to 80
constexpr int XPLANE = 128 * XP;                    // bf16 elements per plane

static_assert(BK == 32, "gemm_x3 assumes 32-wide K tiles");

// ---- HBM -> registers: 4 quads (row, 4 consecutive k) per thread and operand ----------------------------
// BRANCH-FREE: out-of-range rows / k are clamped to valid addresses and zeroed by a multiply, so every thread issues
// exactly 4 (or 16) loads per tile and operand and the compiler keeps them all in flight (guarded loads become
// branches with an s_waitcnt vmcnt(0) behind each).  Requires 16-byte aligned operands and K % 4 == 0 (host checks).
// The staging waves share their SIMD's issue slots with the MFMA waves (~5 filler slots per 32-cycle MFMA), so the
// per-tile instruction count matters: row clamps / validity are computed ONCE (Stager), interior tiles (the common
// case) skip the zeroing multiplies entirely.

// Addresses are (workgroup-uniform base: src + k0 ...) + (per-lane 32-bit byte offset fixed per output tile), which the
// compiler turns into SGPR-base global loads: no per-load 64-bit VALU address arithmetic.
template <bool KC>
struct Stager {
    gcb src;
    unsigned voff[4];     // KC: byte offset of (clamped row, in-tile k) per quad; k-major: voff[0] = (4kg * ld + clamped column) * 4
    unsigned voff0;       // k-major: the kg = 0 variant of voff[0] (always a valid k row of the tile)
    float keep[4];        // row validity (1 / 0) per quad (k-major: keep[0])
    int ld, K, kq0;
    bool edge_rows;       // workgroup-uniform: some rows of this operand tile are out of range
    __device__ __forceinline__ void init(const float *src_, int ld_, int K_) {
        src = (gcb)(uintptr_t)src_;      // kernel-argument pointers copied around lose their address space otherwise
        ld = ld_; K = K_;
    }
    __device__ __forceinline__ void set_tile(int row0, int nrows, int tid) {
        edge_rows = row0 + 128 > nrows;
        if (KC) {
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const int idx = p * NT + tid;
                const int row = row0 + idx / 8;
                voff[p] = ((unsigned)min(row, nrows - 1) * (unsigned)ld + (unsigned)(idx % 8) * 4u) * 4u;
                keep[p] = row < nrows ? 1.f : 0.f;
            }
            kq0 = (tid & 7) * 4;      // NT % 8 == 0: every quad of a thread sits at the same in-tile k
        } else {
            // [K][rows] source: thread (kg, mq) takes the 4x4 block k = 4kg..4kg+3, rows 4mq..4mq+3 as four 16-byte loads
            // along the contiguous dimension (8 lanes = 128 contiguous bytes per k row) and transposes it in registers
            // into four (row, 4 k) quads.  rows % 4 == 0 (host checks): a quad of rows is entirely valid or invalid.
            const int lane = tid & 63, kg = lane >> 3, mq = (tid >> 6) * 8 + (lane & 7);
            const int row = row0 + 4 * mq;
            voff0 = (unsigned)min(row, nrows - 4) * 4u;
            voff[0] = (unsigned)(4 * kg) * (unsigned)ld * 4u + voff0;
            keep[0] = row < nrows ? 1.f : 0.f;
            kq0 = 4 * kg;
        }
    }
    // ONE straight-line path with a fixed number of loads (4 per operand): tiles in the interior of the operand take the
    // same clamps (all zero) and keep-factors (all one) as edge tiles.  Any branch around a load -- even a workgroup-uniform
    // interior/edge one -- makes the number of YOUNGER loads in flight unknown to the compiler at the point where an
    // older register set is consumed, and it then waits with s_waitcnt vmcnt(0): the 2-tile prefetch distance of the
    // staging pipeline collapsed to "wait for everything", i.e. one exposed HBM latency per K tile.
    // Every address is (workgroup-uniform 64-bit base of the tile) + (32-bit per-lane byte offset): SGPR-base loads, no
    // per-load 64-bit VALU arithmetic.  K % 4 == 0 (host checks), so a thread's quad of k is entirely inside or outside K:
    // an outside quad is redirected to an inside one of the same tile and zeroed by its factor.
    // r: the raw 16 floats; f: the factor (1 = keep, 0 = out of range) of quad p (KC) / of the thread's k quad (k-major, all
    // four f equal).  The factors are applied when the registers are SPLIT (x3_r2s), not here: touching a loaded value at
    // fetch time would park the wave on that load and there would be no prefetch at all.
    // Returns (workgroup-uniform) whether any factor can be 0, so that interior tiles skip the multiplies.
    __device__ __forceinline__ bool fetch(float (&r)[16], float (&f)[4], int k0) const {
        const bool inside = k0 + kq0 < K;                      // per lane; false only in the last tile of a segment
        if (KC) {
            const unsigned back = inside ? 0u : (unsigned)kq0 * 4u;      // -> the tile's first quad of the same row
            gcb b = src + (size_t)k0 * 4;
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const f32x4 v = *(gcf4)(b + (voff[p] - back));
                f[p] = inside ? keep[p] : 0.f;
                r[4 * p] = v[0]; r[4 * p + 1] = v[1]; r[4 * p + 2] = v[2]; r[4 * p + 3] = v[3];
            }
        } else {
            const unsigned off = inside ? voff[0] : voff0;
            const size_t ld4 = (size_t)ld * 4;
            gcb b = src + (size_t)k0 * ld4;
            f32x4 v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                v[j] = *(gcf4)(b + off);
                b += ld4;
                f[j] = inside ? keep[0] : 0.f;
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) r[4 * i + j] = v[j][i];       // quad i = row 4mq+i, k = 4kg..4kg+3
        }
        return edge_rows || k0 + BK > K;
    }
};

// registers -> three bf16 planes in LDS
template <bool KC, bool EDGE>
__device__ __forceinline__ void x3_r2s(const float (&r)[16], const float (&f)[4], unsigned short *dst, int tid) {
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
        // 5.5 VALU ops per element: and, sub, and, sub + three packs per pair.  The packs take the UPPER halves of x, x - h
        // and x - h - m directly (the upper half of x is h's, of x - h is m's; l has <= 8 significant bits).
        uint32_t h[4], m[4], l[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float x = EDGE ? r[4 * p + j] * (KC ? f[p] : f[j]) : r[4 * p + j];   // out-of-range rows / k -> 0
            h[j] = fbits(x);
            const float r1 = x - bfloat(h[j] & 0xffff0000u);            // exact
            m[j] = fbits(r1);
            l[j] = fbits(r1 - bfloat(m[j] & 0xffff0000u));              // exact, <= 8 significant bits: truncation is lossless
        }
        unsigned short *o = dst + wswz(row, kq);
        *reinterpret_cast<u32x2 *>(o) = u32x2{pack2(h[0], h[1]), pack2(h[2], h[3])};
        *reinterpret_cast<u32x2 *>(o + XPLANE) = u32x2{pack2(m[0], m[1]), pack2(m[2], m[3])};
        *reinterpret_cast<u32x2 *>(o + 2 * XPLANE) = u32x2{pack2(l[0], l[1]), pack2(l[2], l[3])};
    }
}

// 512 threads: waves 0-3 are MFMA waves (64x64 each), waves 4-7 are STAGING waves (HBM -> split -> LDS).  One of
// each kind sits on every SIMD, so the staging VALU work runs in the shadow of the other wave's MFMAs; two LDS stages
// ping-pong and ONE workgroup barrier per K tile hands a stage over.  The workgroups are PERSISTENT (one per CU, 120 KB
// of LDS): each walks a list of (output tile, K slice) units and the staging waves run ahead across unit boundaries,
// so the HBM latency of a unit's first tiles and the MFMA waves' epilogue stores overlap with useful work.
[[maybe_unused]] constexpr int XNT = 512;   // (ablation builds)
constexpr int XSTAGE = 6 * XPLANE;            // bf16 elements per stage: 3 planes x (A, B) = 60 KB

__device__ __forceinline__ Unit unit_of(const KArgs &a, int u, int gm, int gn) {
    const int per = gm * gn;
    // XCD-aware order: workgroup b runs on XCD b % 8 (each XCD has its own L2).  Within a full round of the grid,
    // hand every XCD a CONTIGUOUS run of output tiles (same A panel, neighbouring B panels) instead of every 8th one:
    // the dW GEMMs fetched their [K x 128] gradient panel through all 8 L2s (139 MB per launch for 25 MB of operands).
    {
        const int G = gridDim.x, total = per * a.splits;
        const int b = u % G, r = u / G;
        if ((G & 7) == 0 && (r + 1) * G <= total) u = (b & 7) * (G >> 3) + (b >> 3) + r * G;
    }
    const int z = u / per, tile = u - z * per;
    Unit r;
    r.z = z;
    r.m0 = (tile / gn) * XBM;
    r.n0 = (tile % gn) * XBN;
    r.t_begin = (int)(((long long)a.tiles_total * z) / a.splits);
    r.nt = (int)(((long long)a.tiles_total * (z + 1)) / a.splits) - r.t_begin;
    return r;
}

// ---------------- staging waves: HBM -> registers -> split -> LDS planes ----------------
// DOA / DOB: this wave stages operand A / B (one wave per SIMD stages both; with two staging waves per SIMD one takes each).
template <bool AKC, bool BKC, int ABL, bool DOA, bool DOB>
__device__ __forceinline__ void x3_staging(const KArgs &a, int gm, int gn, int units, int tid, unsigned short *smem) {
    float ra0[16], rb0[16], ra1[16], rb1[16];          // pipeline step g lives in register set g & 1
    float fa0[4], fb0[4], fa1[4], fb1[4];              // ... with its keep-factors
    bool e0 = false, e1 = false;                       // ... and whether any of them can be zero (workgroup-uniform)
    // K segments ([h | x | ...] x [W slices]) are walked in place.  The segment of a K tile is workgroup-uniform; its
    // fields are picked with STATIC indices behind a uniform switch (a dynamically indexed kernel-argument table
    // would be copied to scratch), and the per-lane offsets are rebuilt only when the segment or the unit changes.
    Stager<AKC> sa;
    Stager<BKC> sb;
    auto bind = [&](int sidx, int m0_, int n0_) {
#define CAPMI_X3_SEG(I)                                                   \
    case I:                                                               \
        sa.init(a.seg[I].A, a.seg[I].lda, a.seg[I].K);                    \
        sb.init(a.seg[I].B, a.seg[I].ldb, a.seg[I].K);                    \
        break;
        switch (sidx) {
            CAPMI_X3_SEG(0) CAPMI_X3_SEG(1) CAPMI_X3_SEG(2) CAPMI_X3_SEG(3)
        }
#undef CAPMI_X3_SEG
        if (DOA) sa.set_tile(m0_, a.M, tid);
        if (DOB) sb.set_tile(n0_, a.N, tid);
    };
    // fetch cursor: runs up to 3 steps ahead of the step being consumed, across unit boundaries.  It is advanced
    // incrementally (k0 += BK; next segment / next unit are rare uniform branches): a flat tile index decoded per step
    // cost ~150 scalar instructions per K tile, a third of the staging wave's issue slots.
    int fu = blockIdx.x, f_left = 0, f_sleft = 0, f_k0 = 0, f_s = 0, f_m0 = 0, f_n0 = 0;
    auto open_unit = [&]() {
        const Unit un = unit_of(a, fu, gm, gn);
        int sidx, k0;
        locate(a, un.t_begin, sidx, k0);
        f_s = __builtin_amdgcn_readfirstlane(sidx);
        f_k0 = k0; f_left = un.nt; f_m0 = un.m0; f_n0 = un.n0;
        bind(f_s, f_m0, f_n0);
        f_sleft = (sa.K - k0 + BK - 1) / BK;
    };
    if (fu < units) open_unit();
    // Every call issues exactly the same loads (see Stager::fetch); past the last unit the cursor stays on the last
    // tile and the (few) extra fetches are thrown away.
    auto fetch = [&](float (&xa)[16], float (&xb)[16], float (&ya)[4], float (&yb)[4], bool &edge) {
        if (!(ABL & 2)) {
            bool ea = false, eb = false;
            if (DOA) ea = sa.fetch(xa, ya, f_k0);
            if (DOB) eb = sb.fetch(xb, yb, f_k0);
            edge = ea || eb;
        }
        if (f_left > 0) {                              // workgroup-uniform
            if (--f_left == 0) {
                fu += gridDim.x;
                if (fu < units) open_unit();
            } else if (--f_sleft == 0) {
                bind(++f_s, f_m0, f_n0);
                f_k0 = 0;
                f_sleft = (sa.K + BK - 1) / BK;
            } else {
                f_k0 += BK;
            }
        }
    };
    int steps = 0;
    for (int u = blockIdx.x; u < units; u += gridDim.x) steps += unit_of(a, u, gm, gn).nt;
    // the interior/edge branch sits HERE, around VALU + LDS work only (a branch around the loads would cost the exact waits)
    auto store = [&](const float (&xa)[16], const float (&xb)[16], const float (&ya)[4], const float (&yb)[4], bool edge, int g) {
        if (ABL & 4) return;
        unsigned short *st = smem + (g & 1) * XSTAGE;
        if (edge) {
            if (DOA) x3_r2s<AKC, true>(xa, ya, st, tid);
            if (DOB) x3_r2s<BKC, true>(xb, yb, st + 3 * XPLANE, tid);
        } else {
            if (DOA) x3_r2s<AKC, false>(xa, ya, st, tid);
            if (DOB) x3_r2s<BKC, false>(xb, yb, st + 3 * XPLANE, tid);
        }
    };
    // Two register sets alternate: while the MFMA waves consume step g (stage g&1) the staging waves publish step g+1 and
    // fetch step g+3 into the registers step g+1 just left.  The fetches are unconditional (fixed load count per
    // half-iteration, see Stager::fetch), only stores and barriers are guarded, so the compiler waits with exact
    // s_waitcnt vmcnt(8..15) for the OLDER set while the younger set's loads stay in flight.
    fetch(ra0, rb0, fa0, fb0, e0);                     // step 0
    fetch(ra1, rb1, fa1, fb1, e1);                     // step 1
    if (steps > 0) store(ra0, rb0, fa0, fb0, e0, 0);
    fetch(ra0, rb0, fa0, fb0, e0);                     // step 2
    __syncthreads();                                   // stage 0 ready
    for (int g = 0; g < steps; g += 2) {
        if (g + 1 < steps) store(ra1, rb1, fa1, fb1, e1, g + 1);
        fetch(ra1, rb1, fa1, fb1, e1);
        __syncthreads();
        const bool second = g + 1 < steps;
        if (second && g + 2 < steps) store(ra0, rb0, fa0, fb0, e0, g + 2);
        fetch(ra0, rb0, fa0, fb0, e0);
        if (second) __syncthreads();
    }
}

// ABL: compile-time ablations for profiling (CAPMI_GEMM_ABLATE, same-layout shapes only): 1 no MFMAs, 2 no global
// fetch, 4 no split/store.  Compile-time because a run-time test around the loads costs the exact vmcnt waits.
// NSW: staging waves.  4: one per SIMD stages both operands (512 threads, 256 registers: the MFMA waves software-pipeline
// their LDS reads).  8: two per SIMD, one per operand (768 threads, 168 registers: plain read-then-multiply MFMA loop).
template <bool AKC, bool BKC, int ABL = 0, int NSW = 4>
__global__ __launch_bounds__(256 + 64 * NSW) void gemm_x3_kernel(const KArgs a, int gm, int gn) {
    extern __shared__ __attribute__((aligned(16))) unsigned short smem[];      // 2 stages = 120 KB
    const int units = gm * gn * a.splits;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;

    if (wid >= 4) {
        if (a.ablate & 8) __builtin_amdgcn_s_setprio(1);
        if (NSW == 4) x3_staging<AKC, BKC, ABL, true, true>(a, gm, gn, units, threadIdx.x - NT, smem);
        else if (wid < 8) x3_staging<AKC, BKC, ABL, true, false>(a, gm, gn, units, threadIdx.x - NT, smem);
        else x3_staging<AKC, BKC, ABL, false, true>(a, gm, gn, units, threadIdx.x - 2 * NT, smem);
        return;
    }

    // ---------------- MFMA waves ----------------
    const int wm0 = ((wid >> 1) & 1) * 64, wn0 = (wid & 1) * 64;
    const int l31 = lane & 31, half = lane >> 5;
    // LDS reads are software-pipelined against the MFMAs: the fragments of K half 1 are requested before the 24 MFMAs of half 0
    // are issued, and the fragments of the NEXT tile's half 0 (other stage, possibly the next unit's) before the MFMAs of
    // half 1, so the matrix pipe never waits for ds_read latency (measured 0.90 -> ... us per K tile with staging ablated; the
    // pipe needs 0.64).  The one barrier per K tile sits between the two halves: by then every read of this stage has
    // completed (the barrier's lgkmcnt(0)) and the staging waves have published the next one.
    auto read_frag = [&](bf16x8 (&av)[2][3], bf16x8 (&bv)[2][3], int g_, int ks) {
        const unsigned short *As = smem + (g_ & 1) * XSTAGE, *Bs = As + 3 * XPLANE;
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) {
                av[q][pl] = *reinterpret_cast<const bf16x8 *>(As + pl * XPLANE + wswz(wm0 + 32 * q + l31, 16 * ks + 8 * half));
                bv[q][pl] = *reinterpret_cast<const bf16x8 *>(Bs + pl * XPLANE + wswz(wn0 + 32 * q + l31, 16 * ks + 8 * half));
            }
    };
    __syncthreads();                                       // stage 0 ready
    int g = 0;
    bf16x8 av0[2][3], bv0[2][3], av1[2][3], bv1[2][3];
    if (NSW == 4) read_frag(av0, bv0, 0, 0);
    for (int u = blockIdx.x; u < units; u += gridDim.x) {
        const Unit un = unit_of(a, u, gm, gn);
        f32x16 acc[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        // six of the nine cross terms, small ones first (planes: 0 = h, 1 = m, 2 = l)
        auto mfma24 = [&](const bf16x8 (&av)[2][3], const bf16x8 (&bv)[2][3]) {
            if (ABL & 1) return;
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
        };
        if (NSW == 4) {
            for (int i = 0; i < un.nt; ++i, ++g) {
                // (sched_barrier: the scheduler otherwise sinks the reads behind most of the MFMAs to shorten live ranges)
                read_frag(av1, bv1, g, 1);
                __builtin_amdgcn_sched_barrier(0);
                mfma24(av0, bv0);
                __builtin_amdgcn_sched_barrier(0);
                __syncthreads();                               // stage g&1 released, stage (g+1)&1 ready
                read_frag(av0, bv0, g + 1, 0);                 // (past the last step: harmless read of a stale stage)
                __builtin_amdgcn_sched_barrier(0);
                mfma24(av1, bv1);
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
            for (int i = 0; i < un.nt; ++i, ++g) {
                read_frag(av0, bv0, g, 0);
                mfma24(av0, bv0);
                read_frag(av0, bv0, g, 1);
                mfma24(av0, bv0);
                __syncthreads();                               // stage g&1 released, stage (g+1)&1 ready
            }
        }

        x3_epilogue<2>(a, un, acc, wm0, wn0, l31, half);
    }
}

}  // namespace

int launch_x3(const KArgs &a, int a_layout, int b_layout, dim3 tiles, hipStream_t st, int pcls, double bytes, double flops) {
    // tiles = (gn, gm, splits) of the 128x128 tiling; persistent grid: one workgroup per CU walks the unit list
    const int gn = tiles.x, gm = tiles.y;
    const int units = gn * gm * a.splits;
    static const int env_wg = capmi::research("CAPMI_X3_WGS", 256);
    const dim3 grid(units < env_wg ? units : env_wg);
    hipEvent_t e0, e1;
    const bool prof = capmi_prof::take_events(pcls, &e0, &e1, bytes, flops);
    constexpr size_t lds = 2 * (size_t)XSTAGE * sizeof(unsigned short);
#define CAPMI_X3N(AK, BK_, NSW_)                                                                                \
    do {                                                                                                        \
        static bool attr_set = false;                                                                           \
        if (!attr_set) {                                                                                        \
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&gemm_x3_kernel<AK, BK_, 0, NSW_>),        \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                    \
            attr_set = true;                                                                                    \
        }                                                                                                       \
        const dim3 blk(256 + 64 * NSW_);                                                                        \
        if (prof) hipExtLaunchKernelGGL((gemm_x3_kernel<AK, BK_, 0, NSW_>), grid, blk, lds, st, e0, e1, 0, a, gm, gn);  \
        else hipLaunchKernelGGL((gemm_x3_kernel<AK, BK_, 0, NSW_>), grid, blk, lds, st, a, gm, gn);                     \
    } while (0)
#define CAPMI_X3(AK, BK_)                                                                                       \
    do {                                                                                                        \
        CAPMI_X3N(AK, BK_, 8);                                                                                  \
    } while (0)
#ifdef CAPMI_VARIANTS
#define CAPMI_X3A(AK, BK_, ABL_)                                                                                \
    do {                                                                                                        \
        static bool attr_set = false;                                                                           \
        if (!attr_set) {                                                                                        \
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&gemm_x3_kernel<AK, BK_, ABL_>),           \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                    \
            attr_set = true;                                                                                    \
        }                                                                                                       \
        hipLaunchKernelGGL((gemm_x3_kernel<AK, BK_, ABL_>), grid, dim3(XNT), lds, st, a, gm, gn);               \
    } while (0)
    if ((a.ablate & 7) && a_layout == b_layout) {            // profiling builds: [K][rows] x [K][rows] and [rows][K] x [rows][K]
        if (a_layout == 1) {
            if ((a.ablate & 7) == 1) CAPMI_X3A(false, false, 1);
            else if ((a.ablate & 7) == 2) CAPMI_X3A(false, false, 2);
            else if ((a.ablate & 7) == 4) CAPMI_X3A(false, false, 4);
            else CAPMI_X3A(false, false, 6);
        } else {
            if ((a.ablate & 7) == 1) CAPMI_X3A(true, true, 1);
            else if ((a.ablate & 7) == 2) CAPMI_X3A(true, true, 2);
            else if ((a.ablate & 7) == 4) CAPMI_X3A(true, true, 4);
            else CAPMI_X3A(true, true, 6);
        }
        CAPMI_CHECK_LAUNCH();
        return 0;
    }
#undef CAPMI_X3A
#endif
    if (a_layout == 0 && b_layout == 0) CAPMI_X3(true, true);
    else if (a_layout == 0 && b_layout == 1) CAPMI_X3(true, false);
    else if (a_layout == 1 && b_layout == 1) CAPMI_X3(false, false);
    else CAPMI_X3(false, true);
#undef CAPMI_X3
    CAPMI_CHECK_LAUNCH();
    return 0;
}

}  // namespace capmi_gemm
