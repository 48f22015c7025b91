// "A-resident" skinny fp32 MFMA GEMM for gfx950: C[M<=64, N] = sum_s A_s[M,K_s] * op(B_s), the decode-step
// shape (60 caption rows against 12-64 MB of weights).
//
// Why: the LDS-tiled kernel (gemm_f32.hip) spends a barrier pair, an LDS write pass and an LDS read pass on the
// WEIGHT tile of every 32-wide K step although each weight element is used by exactly one wave, and its MFMA
// phase never overlaps its loads (ablation: memory-only 15 us, MFMA-only 22 us, both 28 us for the 48 MB gate
// GEMM whose co-limit is ~12 us).  Here
//  * a workgroup (8 waves) owns ONE K slice of the activations for all 64 rows and keeps it RESIDENT in LDS
//    (<= 145 KB of the CU's 160 KB: 64 x (18*32 + 4) floats), staged once, read with conflict-free ds_read_b128;
//  * weights never touch LDS: each wave streams its own 32 output columns x half of the slice's K chunks straight
//    into VGPRs, 3 chunks (12 x 16 B per lane) in flight, so the main loop has NO barrier and the only LDS
//    traffic is the A operand; the two K halves of a column group sit on the same SIMD and hide each other's
//    waits, and meet through LDS at the end (half the slabs of a pure split-K);
//  * the MFMA k index is a free permutation: lane (l&31, l>>5) takes k = 16*(l>>5) + 4q + e of each 32-wide
//    chunk, i.e. 64 contiguous bytes of its weight row per chunk ([N][K] weights), or 16 coalesced 128-byte
//    rows per half-wave ([K][N] weights of dX = dG W);
//  * 2 independent accumulator chains per wave (rows 0-31 / 32-63) keep the matrix pipe issuing back to back;
//  * the grid is (N/128 column blocks) x (K slices) ~ 256 workgroups = one per CU; K slices leave as slabs for the
//    fused consumer / reduce kernel exactly like the tiled kernel.
#include "gemm_common.h"
#include "profile.h"
#include <hip/hip_ext.h>

namespace capmi_gemm {
namespace {

constexpr int AR_BN = 128;        // 4 waves x 32 columns
#ifndef CAPMI_AR_PF
#define CAPMI_AR_PF 3
#endif
constexpr int AR_PF = CAPMI_AR_PF;          // weight chunks in flight per wave
constexpr int AR_TSMAX = 9;       // K chunks (of 32) per wave; the 2*9-chunk slice of 64 rows is 145 KB of LDS
constexpr int AR_NT = 512;        // 8 waves: 4 column groups x 2 K halves

// One 32-wide K chunk of the slice, resolved once per workgroup into LDS so neither the staging loop nor the
// weight stream indexes the kernel-argument segment table per lane (that became dependent global loads).
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

struct TileRef {
    const float *A, *B;     // segment base + k0 (A: column offset; B: column offset [N][K] / row offset [K][N])
    int lda, ldb;
    int rdiv;               // ceil(65536 / a_row_div): row / a_row_div == (row * rdiv) >> 16 for row < 64
    int arem;               // activations valid for k < arem within the chunk (<= 0: padding slot, all zero)
    int brem;               // weights: K_s - k0 (>= 4), offsets clamp to brem - 4 / brem - 1
};

// pointers that round-trip through LDS lose their address space; without this the loads become flat_load (which
// also ticks lgkmcnt and forces vmcnt(0)/lgkmcnt(0) pairs)
typedef const float __attribute__((address_space(1))) *gcf;
typedef const f32x4 __attribute__((address_space(1))) *gcf4;
__device__ __forceinline__ gcf as_global(const float *p) { return (gcf)(uintptr_t)p; }

__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ const float *uni(const float *p) {
    const uint64_t u = reinterpret_cast<uint64_t>(p);
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)u), hi = __builtin_amdgcn_readfirstlane((uint32_t)(u >> 32));
    return reinterpret_cast<const float *>(((uint64_t)hi << 32) | lo);
}

// Branch-free weight fetch of one 32-wide K chunk for this lane: always exactly 4 x 16-byte (or 16 x 4-byte) loads,
// out-of-range rows / k clamped to valid memory.  Static load counts keep the compiler's s_waitcnt vmcnt(n) precise:
// with guarded loads every wait degenerated to vmcnt(0) and the prefetch ring was serialised (43 us for 48 MB).
// Clamped k positions meet a ZERO activation in LDS, clamped columns are never stored.
template <bool BKC>
__device__ __forceinline__ void load_b(float (&b)[16], const TileRef *tp, int colc, int half) {
    const float *B = uni(tp->B);
    const int ldb = uni(tp->ldb), brem = uni(tp->brem);
    if (BKC) {
        gcf p = as_global(B) + (size_t)colc * ldb;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int kk = min(16 * half + 4 * q, brem - 4);
            const f32x4 v = *(gcf4)(p + kk);
            b[4 * q] = v[0]; b[4 * q + 1] = v[1]; b[4 * q + 2] = v[2]; b[4 * q + 3] = v[3];
        }
    } else {
        gcf p = as_global(B) + colc;
#pragma unroll
        for (int j = 0; j < 16; ++j) b[j] = p[(size_t)min(16 * half + j, brem - 1) * ldb];
    }
}

// ABL = 16: phase trace (s_memtime) of waves 0 and 4 of every workgroup into the ticket words of the workspace
#define CAPMI_STAMP(slot)                                                                                          \
    do {                                                                                                           \
        if ((ABL & 16) && lane == 0 && (wid & 3) == 0) {                                                           \
            __builtin_amdgcn_sched_barrier(0);                                                                     \
            reinterpret_cast<unsigned long long *>(a.counters)[((blockIdx.y * gridDim.x + blockIdx.x) * 2 + (wid >> 2)) * 10 + (slot)] = \
                __builtin_readcyclecounter();                                                                      \
            __builtin_amdgcn_sched_barrier(0);                                                                     \
        }                                                                                                          \
    } while (0)

// TM = 1: M <= 32 (one accumulator chain, 32 staged rows); TM = 2: M <= 64
// ABL (profiling builds of the <true,6,2,x3> shape only, CAPMI_ARES_ABLATE): 1 = no activation loads, 2 = no split / MFMA,
// 4 = no weight loads, 8 = no K-half reduction / stores.  Never for results.
// DIRECT (round 2, bf16x3 path).  The phase trace showed the tile table costing 2.4k cycles before the activation loads -- which
// then need another 5-10k cycles to arrive -- could even be issued: per-lane indexing of the kernel-argument segment array had
// become a chain of dependent GLOBAL loads plus an integer division.  Here the segment fields are pinned in SGPRs (s_load), the
// table entries are resolved by value selects against host-computed tile starts / reciprocals, and only ONE weight chunk per
// wave is requested ahead of the activations; the rest of the ring follows once they have landed (the full ring ahead of them
// delayed their arrival from 5k to 10-13k cycles).
template <bool BKC, int TS, int TM, bool X3, int ABL = 0, bool DIRECT = false>
__global__ __launch_bounds__(AR_NT) void gemm_ares_kernel(const KArgs a) {
    constexpr int ROWS = 32 * TM;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int SL = 2 * TS;                  // K chunks per workgroup slice: waves 0-3 take [0,TS), waves 4-7 [TS,2TS)
    constexpr int pitch = SL * 32 + 4;              // fp32 image: row pitch in floats
    constexpr int pitchH = SL * 32 + 8;             // bf16x3 image: row pitch in bf16 elements (16 bytes of padding)
    constexpr int planeH = ROWS * pitchH;           // bf16 elements per plane (h, m, l)
    float *As = lds;                                            // [ROWS][pitch]            (exact fp32 path)
    unsigned short *Ah = reinterpret_cast<unsigned short *>(lds);   // [3][ROWS][pitchH]    (bf16x3 path)
    TileRef *tiles = X3 ? reinterpret_cast<TileRef *>(Ah + 3 * planeH) : reinterpret_cast<TileRef *>(lds + ROWS * pitch);   // [SL]
    // XCD-aware workgroup -> (column block, K slice) map: workgroups are dispatched round-robin over the 8 XCDs (id % 8),
    // each with a private L2.  All column blocks of one K slice read the SAME activation slice, so slices are dealt to
    // XCDs (slice-major rank p, contiguous p per XCD): an XCD's L2 then fetches 1/8 of the activations once instead of all of
    // them, and 31 of 32 workgroups find their slice in L2.  Pure speed: any placement computes the same result.
    int bx = blockIdx.x, z = blockIdx.y;
    if (a.ablate & 1) {
        const int L = blockIdx.y * gridDim.x + blockIdx.x, total = gridDim.x * gridDim.y;
        const int q = total >> 3, r = total & 7, xcd = L & 7;
        const int p = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (L >> 3);
        z = p / (int)gridDim.x;
        bx = p - z * (int)gridDim.x;
    }
    const int n0 = bx * AR_BN;
    const int t0 = z * SL;          // every slice runs exactly SL chunks; slots past the last K tile are zero padded
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int cg = wid & 3, kh = wid >> 2;
    const int l31 = lane & 31, half = lane >> 5;
    const int col = n0 + 32 * cg + l31;
    const int colc = min(col, a.N - 1);
    CAPMI_STAMP(0);

    // segment fields as wave-uniform values (s_load from the kernel arguments, pinned in SGPRs)
    const float *sgA[CAPMI_MAX_SEG], *sgB[CAPMI_MAX_SEG];
    int sgLda[CAPMI_MAX_SEG], sgLdb[CAPMI_MAX_SEG], sgK[CAPMI_MAX_SEG], sgRdiv[CAPMI_MAX_SEG], sgT[CAPMI_MAX_SEG];
    if (DIRECT) {
#pragma unroll
        for (int i = 0; i < CAPMI_MAX_SEG; ++i) {
            sgA[i] = a.seg[i].A; sgB[i] = a.seg[i].B; sgLda[i] = a.seg[i].lda; sgLdb[i] = a.seg[i].ldb; sgK[i] = a.seg[i].K;
            sgRdiv[i] = a.seg[i].rdiv; sgT[i] = a.seg[i].tstart;
            asm volatile("" : "+s"(sgA[i]), "+s"(sgB[i]), "+s"(sgLda[i]), "+s"(sgLdb[i]), "+s"(sgK[i]), "+s"(sgRdiv[i]), "+s"(sgT[i]));
        }
    }
    // tile (flat index) -> TileRef, by value selects (works per lane and, on uniform input, on the scalar unit)
    auto resolve = [&](int tile) {
        TileRef t;
        const bool real = tile < a.tiles_total;
        const int tl = real ? tile : 0;
        const int s = (tl >= sgT[1] ? 1 : 0) + (tl >= sgT[2] ? 1 : 0) + (tl >= sgT[3] ? 1 : 0);
#define CAPMI_SEL(F) (s == 3 ? F[3] : s == 2 ? F[2] : s == 1 ? F[1] : F[0])
        const int k0 = (tl - CAPMI_SEL(sgT)) * 32;
        const int K = CAPMI_SEL(sgK);
        t.lda = CAPMI_SEL(sgLda); t.ldb = CAPMI_SEL(sgLdb); t.rdiv = CAPMI_SEL(sgRdiv);
        t.A = CAPMI_SEL(sgA) + k0;
        t.B = BKC ? CAPMI_SEL(sgB) + k0 : CAPMI_SEL(sgB) + (size_t)k0 * t.ldb;
#undef CAPMI_SEL
        t.arem = real ? K - k0 : 0;
        t.brem = K - k0;
        return t;
    };
    if (DIRECT && threadIdx.x < SL) tiles[threadIdx.x] = resolve(t0 + (int)threadIdx.x);
    if (!DIRECT && threadIdx.x < SL) {
        int s = 0, k0 = 0;
        const bool real = t0 + (int)threadIdx.x < a.tiles_total;
        if (real) locate(a, t0 + threadIdx.x, s, k0);
        TileRef t;
        // per-lane dynamic segment index resolved with selects over the (<= 4) segments
        const float *sA = a.seg[0].A, *sB = a.seg[0].B;
        int lda = a.seg[0].lda, ldb = a.seg[0].ldb, K = a.seg[0].K, div = a.seg[0].a_row_div;
#pragma unroll
        for (int i = 1; i < CAPMI_MAX_SEG; ++i)
            if (s == i) { sA = a.seg[i].A; sB = a.seg[i].B; lda = a.seg[i].lda; ldb = a.seg[i].ldb; K = a.seg[i].K; div = a.seg[i].a_row_div; }
        t.A = sA + k0;
        t.B = BKC ? sB + k0 : sB + (size_t)k0 * ldb;
        t.lda = lda; t.ldb = ldb;
        t.arem = real ? K - k0 : 0;
        t.brem = K - k0;
        t.rdiv = (65536 + div - 1) / div;
        tiles[threadIdx.x] = t;
    }
    __syncthreads();
    CAPMI_STAMP(1);
    const TileRef *mine = tiles + kh * TS;
    // this wave's weight chunk u: from the LDS table, or resolved on the scalar unit (kh is wave-uniform)
    const int khu = __builtin_amdgcn_readfirstlane(kh);
    auto load_w = [&](float (&bb)[16], int u) {
        load_b<BKC>(bb, &mine[u], colc, half);
    };
    (void)khu;

    // Loads return IN ORDER (vmcnt counts them so): the activation loads (L2 hits, all workgroups of a K slice read the
    // same rows) are issued FIRST and the weight prefetch ring right behind them, so the staging below waits for the
    // activations only while the first weight chunks are still on their way from HBM.  (Ring first made the staging
    // wait for 24 MB of weights to land before the first MFMA could start.)
    constexpr int PF = AR_PF < TS ? AR_PF : TS;
    float b[PF + 1][16];
    // stage the activation slice: 64 rows x SL*32 k in 16-byte pieces (rows >= M and k >= K_s zero filled): SL pieces
    // per thread, every load issued before the first LDS store = one L2 round trip for the whole slice.
    constexpr int quads = SL * 8;               // ROWS * quads pieces / 512 threads = TS * TM per thread
    constexpr int NP = TS * TM;
    {
        f32x4 v[NP];
        float okf[NP];
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            const int idx = j * AR_NT + (int)threadIdx.x;
            const int row = idx / quads, c4 = idx - row * quads;
            const int k = (c4 & 7) * 4;
            if (DIRECT) {
                const TileRef t = tiles[c4 >> 3];
                okf[j] = (row < a.M && k < t.arem) ? 1.f : 0.f;
                const int off = (int)okf[j] * (((row * t.rdiv) >> 16) * t.lda + k);
                v[j] = *(gcf4)(as_global(t.A) + off);
                continue;
            }
            const TileRef *tp = &tiles[c4 >> 3];
            const bool ok = row < a.M && k < tp->arem;
            gcf p = as_global(tp->A) + (ok ? (size_t)((row * tp->rdiv) >> 16) * tp->lda + k : 0);
            if (ABL & 1) v[j] = f32x4{1.f, 2.f, 3.f, (float)j}; else
            v[j] = *(gcf4)p;
        }
        __builtin_amdgcn_sched_barrier(0);
        constexpr int PF0 = DIRECT ? 1 : PF;       // weight chunks requested ahead of the activations' arrival
#pragma unroll
        for (int u = 0; u < PF0; ++u) {
            if (ABL & 4) { for (int q = 0; q < 16; ++q) b[u][q] = (float)(q + u + lane); }
            else load_w(b[u], u);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            if (DIRECT) { v[j] *= okf[j]; continue; }
            const int idx = j * AR_NT + (int)threadIdx.x;
            const int row = idx / quads, c4 = idx - row * quads;
            const bool ok = row < a.M && (c4 & 7) * 4 < tiles[c4 >> 3].arem;
            v[j] *= ok ? 1.f : 0.f;                // multiply (not select) so the load stays unconditional
        }
        if (DIRECT) {                               // the activations are here: now fill the rest of the weight ring
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = PF0; u < PF; ++u) {
                if (ABL & 4) { for (int q = 0; q < 16; ++q) b[u][q] = (float)(q + u + lane); }
                else load_w(b[u], u);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (ABL & 16) {
#pragma unroll
            for (int j = 0; j < NP; ++j) asm volatile("" ::"v"(v[j][0]));
            CAPMI_STAMP(9);
        }
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            const int idx = j * AR_NT + (int)threadIdx.x;
            const int row = idx / quads, c4 = idx - row * quads;
            if (!X3) {
                *reinterpret_cast<f32x4 *>(As + row * pitch + c4 * 4) = v[j];
            } else {
                // exact 3-way split (see gemm_x3.hip): x = h + m + l with three truncated bf16 values
                uint32_t h[4], m[4], l[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float x = v[j][e];
                    h[e] = __builtin_bit_cast(uint32_t, x) & 0xffff0000u;
                    const float r1 = x - __builtin_bit_cast(float, h[e]);
                    m[e] = __builtin_bit_cast(uint32_t, r1) & 0xffff0000u;
                    l[e] = __builtin_bit_cast(uint32_t, r1 - __builtin_bit_cast(float, m[e]));
                }
                unsigned short *o = Ah + row * pitchH + c4 * 4;
                *reinterpret_cast<u32x2 *>(o) = u32x2{(h[0] >> 16) | (h[1] & 0xffff0000u), (h[2] >> 16) | (h[3] & 0xffff0000u)};
                *reinterpret_cast<u32x2 *>(o + planeH) = u32x2{(m[0] >> 16) | (m[1] & 0xffff0000u), (m[2] >> 16) | (m[3] & 0xffff0000u)};
                *reinterpret_cast<u32x2 *>(o + 2 * planeH) = u32x2{(l[0] >> 16) | (l[1] & 0xffff0000u), (l[2] >> 16) | (l[3] & 0xffff0000u)};
            }
        }
    }
    CAPMI_STAMP(2);
    __syncthreads();
    CAPMI_STAMP(3);

    f32x16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc0[r] = acc1[r] = 0.f;
    const float *a_lo = As + l31 * pitch + 16 * half + kh * TS * 32;
    const float *a_hi = a_lo + 32 * pitch;

    const unsigned short *ah_lo = Ah + l31 * pitchH + 16 * half + kh * TS * 32;
    auto mma = [&](const float (&bb)[16], int c) {
        if (!X3) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 x0 = *reinterpret_cast<const f32x4 *>(a_lo + c * 32 + 4 * q);
                f32x4 x1 = x0;
                if (TM == 2) x1 = *reinterpret_cast<const f32x4 *>(a_hi + c * 32 + 4 * q);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x0[e], bb[4 * q + e], acc0, 0, 0, 0);
                    if (TM == 2) acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(x1[e], bb[4 * q + e], acc1, 0, 0, 0);
                }
            }
        }
    };
    // bf16x3: lane (l31, half) owns k = 16*half + 8*ks + 0..7 of the chunk for k-step ks.  The weights are split in
    // registers (each is used by this wave only), ONE CHUNK AHEAD of their MFMAs so that the split's dependent VALU
    // chains sit in the shadow of the previous chunk's MFMAs (PMC: 39 % of the wave cycles were issue stalls with the
    // split directly in front of its own MFMAs); the activations were split once when they were staged.
    auto wsplit = [&](const float (&bb)[16], u32x4 (&wb)[2][3]) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int e2 = 0; e2 < 4; ++e2) {
                uint32_t hh[2], mm[2], ll[2];
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const float x = bb[8 * ks + 2 * e2 + t];
                    hh[t] = __builtin_bit_cast(uint32_t, x) & 0xffff0000u;
                    const float r1 = x - __builtin_bit_cast(float, hh[t]);
                    mm[t] = __builtin_bit_cast(uint32_t, r1) & 0xffff0000u;
                    ll[t] = __builtin_bit_cast(uint32_t, r1 - __builtin_bit_cast(float, mm[t]));
                }
                wb[ks][0][e2] = (hh[0] >> 16) | (hh[1] & 0xffff0000u);
                wb[ks][1][e2] = (mm[0] >> 16) | (mm[1] & 0xffff0000u);
                wb[ks][2][e2] = (ll[0] >> 16) | (ll[1] & 0xffff0000u);
            }
    };
    auto mma3 = [&](const u32x4 (&wb)[2][3], int c) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 bw[3], x0[3], x1[3];
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) {
                bw[pl] = __builtin_bit_cast(bf16x8, wb[ks][pl]);
                x0[pl] = *reinterpret_cast<const bf16x8 *>(ah_lo + pl * planeH + c * 32 + 8 * ks);
                if (TM == 2) x1[pl] = *reinterpret_cast<const bf16x8 *>(ah_lo + pl * planeH + 32 * pitchH + c * 32 + 8 * ks);
            }
            // six of nine cross terms, small ones first (0 = h, 1 = m, 2 = l)
            constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
            for (int t = 0; t < 6; ++t) {
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x0[PA[t]], bw[PB[t]], acc0, 0, 0, 0);
                if (TM == 2) acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x1[PA[t]], bw[PB[t]], acc1, 0, 0, 0);
            }
        }
    };
    // fully unrolled ring of PF + 1 slots: chunk c's MFMAs read slot c % (PF+1) while the slot chunk c-1 just
    // released is refilled with chunk c + PF.  Straight-line code with static load counts keeps s_waitcnt vmcnt(n)
    // exact (inside a loop the compiler parked a vmcnt(0) at the loop head); the sched_barriers stop the machine
    // scheduler from sinking each refill down to its consumer (which exposed the full HBM latency).
    if (!X3) {
#pragma unroll
        for (int c = 0; c < TS; ++c) {
            if (c + PF < TS) load_w(b[(c + PF) % (PF + 1)], c + PF);
            __builtin_amdgcn_sched_barrier(0);
            mma(b[c % (PF + 1)], c);
            __builtin_amdgcn_sched_barrier(0);
        }
    } else {
        u32x4 w0[2][3], w1[2][3];                      // split weights of the even / odd chunks
        wsplit(b[0], w0);
#pragma unroll
        for (int c = 0; c < TS; ++c) {
            if (c + PF < TS) {
                if (ABL & 4) { for (int q = 0; q < 16; ++q) b[(c + PF) % (PF + 1)][q] = (float)(q + c + lane); }
                else load_w(b[(c + PF) % (PF + 1)], c + PF);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (ABL & 2) {
                // keep the loads live without the split / MFMA work
#pragma unroll
                for (int q = 0; q < 16; ++q) asm volatile("" ::"v"(b[c % (PF + 1)][q]));
                continue;
            }
            // the scheduler may interleave these two: MFMAs of chunk c, VALU split of chunk c + 1
            if (c & 1) {
                if (c + 1 < TS) wsplit(b[(c + 1) % (PF + 1)], w0);
                mma3(w1, c);
            } else {
                if (c + 1 < TS) wsplit(b[(c + 1) % (PF + 1)], w1);
                mma3(w0, c);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (ABL & 16) {
                if (c == 0) { asm volatile("" ::"v"(acc0[0]), "v"(acc1[0])); CAPMI_STAMP(4); }
                if (c == 2 % TS && TS > 1) { asm volatile("" ::"v"(acc0[0]), "v"(acc1[0])); CAPMI_STAMP(5); }
                if (c == TS - 1) { asm volatile("" ::"v"(acc0[15]), "v"(acc1[15])); CAPMI_STAMP(6); }
            }
        }
    }

    // K halves meet in LDS (the activation slice is dead): waves 4-7 park their 64x32 sums, waves 0-3 add and store.
    // C/D map of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5).
    constexpr int RP = AR_BN + 4;
    if (ABL & 8) {
#pragma unroll
        for (int r = 0; r < 16; ++r) { asm volatile("" ::"v"(acc0[r])); asm volatile("" ::"v"(acc1[r])); }
        return;
    }
    __syncthreads();
    float *red = lds;                                           // [64][RP]
    if (kh == 1) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
            red[row * RP + 32 * cg + l31] = acc0[r];
            if (TM == 2) red[(32 + row) * RP + 32 * cg + l31] = acc1[r];
        }
    }
    __syncthreads();
    CAPMI_STAMP(7);
    if (kh == 1 || col >= a.N) return;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
        acc0[r] += red[row * RP + 32 * cg + l31];
        if (TM == 2) acc1[r] += red[(32 + row) * RP + 32 * cg + l31];
    }
    if (a.to_partial) {
        float *out = a.partial + (size_t)z * a.M * a.N + col;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
            if (row < a.M) out[(size_t)row * a.N] = acc0[r];
            if (TM == 2 && row + 32 < a.M) out[(size_t)(row + 32) * a.N] = acc1[r];
        }
        if (ABL & 16) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); CAPMI_STAMP(8); }
        return;
    }
    float cb = 0.f;
    if (a.bias) cb += a.bias[col];
    if (a.bias2) cb += a.bias2[col];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = 32 * i + (r & 3) + 8 * (r >> 2) + 4 * half;
            if (row >= a.M) continue;
            float v = (i == 0 ? acc0[r] : acc1[r]) + cb;
            if (a.row_bias) v += a.row_bias[(size_t)(row / a.row_bias_div) * a.N + col];
            if (a.relu) v = fmaxf(v, 0.f);
            if (a.mul_mask) v *= a.mul_mask[(size_t)row * a.N + col];
            if (a.accumulate) v += a.addend[(size_t)row * a.ldc + col];
            a.C[(size_t)row * a.ldc + col] = v;
        }
    }
}



// X[M, K] (row pitch ld) -> A planes.  Producers on the hot path write their planes themselves (pl_store4); this kernel serves
// operands that have no fused producer and the tests.
__global__ void planes_from_f32_kernel(const float *__restrict__ X, int ld, int M, int K, unsigned char *__restrict__ pl) {
    const int K4 = (K + 3) >> 2;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= M * K4) return;
    const int row = i / K4, k = (i - row * K4) * 4;
    const float *p = X + (size_t)row * ld + k;
    if (k + 4 <= K && ((reinterpret_cast<uintptr_t>(p) & 15) == 0)) {
        capmi::pl_store4(pl, row, k, *reinterpret_cast<const f32x4 *>(p));
    } else {
        for (int e = 0; e < 4 && k + e < K; ++e) capmi::pl_store1(pl, row, k + e, p[e]);
    }
}

}  // namespace

// plan: K chunks per wave (1..ts_cap); a workgroup slice is 2 of those; *splits slices cover `tiles`
int ares_plan(int N, int tiles, int want_blocks, int ts_cap, int *splits) {
    const int nblk = (N + AR_BN - 1) / AR_BN;
    int s = want_blocks / nblk;
    if (s < 1) s = 1;
    if (s > tiles) s = tiles;
    int ts = ((tiles + s - 1) / s + 1) / 2;
    if (ts < 1) ts = 1;
    if (ts > ts_cap) ts = ts_cap;
    *splits = (tiles + 2 * ts - 1) / (2 * ts);
    return ts;
}

// largest K chunks per wave whose activation slice fits LDS
int ares_ts_cap(int M, int x3) { return (x3 && M > 32) ? 6 : AR_TSMAX; }

template <bool BKC, int TS, int TM, bool X3>
static int launch_ts(const KArgs &a, hipStream_t st, int pcls, double bytes, double flops) {
    static bool attr_set = false;
    constexpr size_t slice = (X3 ? (size_t)3 * 32 * TM * (2 * TS * 32 + 8) * sizeof(unsigned short)
                                 : (size_t)32 * TM * (2 * TS * 32 + 4) * sizeof(float)) + (size_t)2 * TS * sizeof(TileRef);
    constexpr size_t red = (size_t)32 * TM * (AR_BN + 4) * sizeof(float);
    constexpr size_t lds = slice > red ? slice : red;
    static_assert(lds <= 160 * 1024, "activation slice does not fit the CU's LDS");
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&gemm_ares_kernel<BKC, TS, TM, X3>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    dim3 grid((a.N + AR_BN - 1) / AR_BN, a.splits);
    hipEvent_t e0, e1;
    const bool prof = capmi_prof::take_events(pcls, &e0, &e1, bytes, flops);
    if constexpr (X3) {
        if (a.ablate & 2) {                 // table-free direct addressing (CAPMI_ARES_OPT bit 1)
            static bool dset = false;
            if (!dset) {
                (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&gemm_ares_kernel<BKC, TS, TM, X3, 0, true>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                dset = true;
            }
#ifdef CAPMI_VARIANTS
            if constexpr (BKC && TS == 6 && TM == 2) {
                static const int abl = capmi::ablate_env("CAPMI_ARES_ABLATE");
                if (abl == 16) {
                    static bool tset = false;
                    if (!tset) {
                        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&gemm_ares_kernel<BKC, TS, TM, X3, 16, true>),
                                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                        tset = true;
                    }
                    hipLaunchKernelGGL((gemm_ares_kernel<BKC, TS, TM, X3, 16, true>), grid, dim3(AR_NT), lds, st, a);
                    return 0;
                }
            }
#endif
            if (prof) hipExtLaunchKernelGGL((gemm_ares_kernel<BKC, TS, TM, X3, 0, true>), grid, dim3(AR_NT), lds, st, e0, e1, 0, a);
            else hipLaunchKernelGGL((gemm_ares_kernel<BKC, TS, TM, X3, 0, true>), grid, dim3(AR_NT), lds, st, a);
            CAPMI_CHECK_LAUNCH();
            return 0;
        }
    }
#ifdef CAPMI_VARIANTS
    if constexpr (BKC && TS == 6 && TM == 2 && X3) {
        static const int abl = capmi::ablate_env("CAPMI_ARES_ABLATE");
        if (abl) {
#define CAPMI_ABL(V) case V: { static bool set = false; if (!set) { (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&gemm_ares_kernel<BKC, TS, TM, X3, V>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); set = true; } \
            hipLaunchKernelGGL((gemm_ares_kernel<BKC, TS, TM, X3, V>), grid, dim3(AR_NT), lds, st, a); return 0; }
            switch (abl) { CAPMI_ABL(1) CAPMI_ABL(2) CAPMI_ABL(3) CAPMI_ABL(4) CAPMI_ABL(6) CAPMI_ABL(7) CAPMI_ABL(8) CAPMI_ABL(10) CAPMI_ABL(15) CAPMI_ABL(16) CAPMI_ABL(20) default: break; }
#undef CAPMI_ABL
        }
    }
#endif
    if (prof) hipExtLaunchKernelGGL((gemm_ares_kernel<BKC, TS, TM, X3>), grid, dim3(AR_NT), lds, st, e0, e1, 0, a);
    else hipLaunchKernelGGL((gemm_ares_kernel<BKC, TS, TM, X3>), grid, dim3(AR_NT), lds, st, a);
    CAPMI_CHECK_LAUNCH();
    return 0;
}

template <bool BKC, bool X3>
static int launch_layout(const KArgs &a, int ts, hipStream_t st, int pcls, double bytes, double flops) {
    switch (ts) {
#define CAPMI_TS(T) case T: return a.M <= 32 ? launch_ts<BKC, T, 1, X3>(a, st, pcls, bytes, flops) : launch_ts<BKC, T, (X3 && T > 6) ? 1 : 2, X3>(a, st, pcls, bytes, flops);
        CAPMI_TS(1) CAPMI_TS(2) CAPMI_TS(3) CAPMI_TS(4) CAPMI_TS(5) CAPMI_TS(6) CAPMI_TS(7) CAPMI_TS(8) CAPMI_TS(9)
#undef CAPMI_TS
    }
    return CAPMI_EINVAL;
}

// a.splits * 2 * ts must cover a.tiles_total
int launch_ares(const KArgs &a, int b_layout, int ts, int x3, hipStream_t st, int pcls, double bytes, double flops) {
    if (ts < 1 || ts > ares_ts_cap(a.M, x3) || (long long)a.splits * 2 * ts < a.tiles_total) return CAPMI_EINVAL;
    if (x3) return b_layout == 0 ? launch_layout<true, true>(a, ts, st, pcls, bytes, flops)
                                 : launch_layout<false, true>(a, ts, st, pcls, bytes, flops);
    return b_layout == 0 ? launch_layout<true, false>(a, ts, st, pcls, bytes, flops)
                         : launch_layout<false, false>(a, ts, st, pcls, bytes, flops);
}


}  // namespace capmi_gemm

extern "C" int64_t capmi_planes_bytes(int K) { return (int64_t)((K + 31) / 32) * CAPMI_PL_CHUNK_BYTES; }

extern "C" int capmi_planes_from_f32(const float *X, int ld, int M, int K, void *planes, void *stream) {
    if (!X || !planes || M < 1 || M > 64 || K < 1 || ld < K) return CAPMI_EINVAL;
    const int work = M * ((K + 3) / 4);
    hipLaunchKernelGGL(capmi_gemm::planes_from_f32_kernel, dim3((work + 255) / 256), dim3(256), 0, (hipStream_t)stream, X, ld, M,
                       K, static_cast<unsigned char *>(planes));
    CAPMI_CHECK_LAUNCH();
    return 0;
}

