// gemm_lc_body.h -- the loader / consumer decode GEMM of gemm_lc.hip as a device function (see that file for the design notes).
#pragma once
#include "gemm_common.h"

namespace capmi_gemm {
namespace {

constexpr int LC_BN = 128;
constexpr int LC_NT = 768;                          // waves 0-7 consume, waves 8-11 load
constexpr int LC_NS = 5;                            // ring stages
constexpr int LC_AB = CAPMI_PL_CHUNK_BYTES;         // activation image of a chunk
constexpr int LC_WB = LC_BN * 32 * 4;               // weight tile of a chunk
constexpr int LC_STAGE = LC_AB + LC_WB;             // 28 KB

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef const void __attribute__((address_space(1))) *gvoid;
typedef void __attribute__((address_space(3))) *lvoid;

// s_waitcnt vmcnt(n) with expcnt / lgkmcnt left alone (gfx9 encoding: vmcnt[3:0] | expcnt << 4 | lgkmcnt << 8 | vmcnt[5:4] << 14)
constexpr int LC_WAUX = 0;     // default cache policy of the weight DMAs (A/B: profiles/r04_gemm_lc_nt.md)
#define CAPMI_VMCNT(n) __builtin_amdgcn_s_waitcnt((((n) >> 4) << 14) | ((n) & 15) | 0x0f70)

// ABL (profiling builds, CAPMI_LC_ABLATE; never for results): 1 = loaders copy no activations, 2 = consumers skip split / MFMA,
// 4 = loaders copy no weights, 8 = no stores, 16 = s_memtime stamps of waves 0 and 4 into the ticket words of the workspace
// WAUX: cache policy of the WEIGHT DMAs (aux field of global_load_lds): 0 = default, 2 = nt (MI355X_MICROARCH.md row nt-weights);
// the activation planes keep the default policy (32 column blocks re-read them from L2)
// The kernel's body as a device function of (workgroup coordinates, grid extents, LDS base): `gemm_lc_kernel` runs it on its own
// grid; the fused select + GEMM launch (sampler.hip, r4) runs it in the workgroups behind the select rows of the same launch.
// Only threads 0 .. LC_NT - 1 of a workgroup may enter.
template <bool BKC, int TM, int ABL = 0, int WAUX = 0>
__device__ __forceinline__ void gemm_lc_body(const KArgs &a, const int blk_x, const int blk_y, const int grd_x, const int grd_y,
                                             unsigned char *ldsb) {
    int bx = blk_x, z = blk_y;
    if (a.ablate & 1) {                              // XCD-aware map: the column blocks of one K slice share an XCD's L2
        const int L = blk_y * grd_x + blk_x, total = grd_x * grd_y;
        const int q = total >> 3, r = total & 7, xcd = L & 7;
        const int p = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (L >> 3);
        z = p / grd_x;
        bx = p - z * grd_x;
    }
    const int n0 = bx * LC_BN;
    const int SL = a.sl;
    const int t0 = z * SL;
    const int nst = min(SL, a.tiles_total - t0);      // stages of this workgroup (>= 1: splits * sl covers the tiles exactly once)
    const int lane = threadIdx.x & 63;
    const int widu = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
#define CAPMI_LC_STAMP(slot)                                                                                               \
    do {                                                                                                                   \
        if ((ABL & 16) && lane == 0 && (widu & 7) == 0) {                                                                  \
            __builtin_amdgcn_sched_barrier(0);                                                                             \
            reinterpret_cast<unsigned long long *>(a.counters)[((blk_y * grd_x + blk_x) * 2 + (widu >> 3)) * 10 + (slot)] = \
                __builtin_readcyclecounter();                                                                              \
            __builtin_amdgcn_sched_barrier(0);                                                                             \
        }                                                                                                                  \
    } while (0)
    CAPMI_LC_STAMP(0);

    if (widu >= 8) {
        // =================================================== loaders ======================================================
        const int j = widu - 8;
        static_assert(CAPMI_MAX_SEG == 4, "segment selects are written out for 4 segments");
#define CAPMI_PIN(i)                                                                                                       \
        const unsigned char *sgP##i = a.seg[i].Apl; const float *sgB##i = a.seg[i].B;                                      \
        int sgL##i = a.seg[i].ldb, sgK##i = a.seg[i].K, sgT##i = a.seg[i].tstart;                                          \
        asm volatile("" : "+s"(sgP##i), "+s"(sgB##i), "+s"(sgL##i), "+s"(sgK##i), "+s"(sgT##i));
        CAPMI_PIN(0) CAPMI_PIN(1) CAPMI_PIN(2) CAPMI_PIN(3)
#undef CAPMI_PIN
#define CAPMI_SEL(F, tl) ((tl) >= sgT3 ? F##3 : (tl) >= sgT2 ? F##2 : (tl) >= sgT1 ? F##1 : F##0)
        const int N = a.N;
        // per-lane constants of the weight pieces this loader copies (pieces j, j + 4, j + 8, j + 12 of the 16 of a tile)
        // [N][K] weights: piece p = columns 8p .. 8p+7, lane l -> column 8p + (l >> 3), LDS slot l & 7 holds 16-byte piece
        //                 (l & 7) ^ ((column >> 1) & 7) of the column's 128 bytes
        // [K][N] weights: piece p = k rows 2p, 2p+1, lane l -> row 2p + (l >> 5), columns 4 (l & 31) .. +3
        int wrow[4], wk[4];                         // BKC: clamped global column, k offset | !BKC: chunk row, clamped global column
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int p = j + 4 * u;
            if (BKC) {
                const int cl = 8 * p + (lane >> 3);
                wrow[u] = min(n0 + cl, N - 1);
                wk[u] = 4 * ((lane & 7) ^ ((cl >> 1) & 7));
            } else {
                wrow[u] = 2 * p + (lane >> 5);
                wk[u] = min(n0 + 4 * (lane & 31), N - 4);
            }
        }
        // DMAs of one stage by this loader: 3 (rows < 32: 2) activation pieces + 4 weight pieces, a fixed count so that the
        // counted waits below are exact
        constexpr int DPS = (TM == 2 ? 3 : 2) + 4;
        // (a macro, not a lambda: routed through a by-reference capture hipcc turns the segment selects into loads through
        //  selected POINTERS and the pinned scalars end up in scratch memory)
#define CAPMI_LC_ISSUE(i_, slot_idx_)                                                                                      \
        do {                                                                                                               \
            const int tl_ = t0 + (i_);                                                                                     \
            unsigned char *slot_ = ldsb + (slot_idx_) * LC_STAGE;                                                          \
            const int tb_ = CAPMI_SEL(sgT, tl_);                                                                           \
            if (!(ABL & 1)) {                                                                                              \
                const unsigned char *img_ = CAPMI_SEL(sgP, tl_) + (size_t)(tl_ - tb_) * LC_AB + lane * 16;                 \
                _Pragma("unroll") for (int u_ = 0; u_ < (TM == 2 ? 3 : 2); ++u_) {                                         \
                    /* TM = 2: 1-KB pieces j, j + 4, j + 8 of the 12; TM = 1: rows 0-31 = pieces {0,1,4,5,8,9}: two per loader  \
                       (the last one twice: same bytes to the same place) */                                              \
                    int pc_;                                                                                               \
                    if (TM == 2) pc_ = j + 4 * u_;                                                                         \
                    else { const int q_ = min(j + 4 * u_, 5); pc_ = (q_ >> 1) * 4 + (q_ & 1); }                            \
                    __builtin_amdgcn_global_load_lds((gvoid)(uintptr_t)(img_ + pc_ * 1024), (lvoid)(slot_ + pc_ * 1024), 16, 0, 0); \
                }                                                                                                          \
            }                                                                                                              \
            if (!(ABL & 4)) {                                                                                              \
                const int k0_ = (tl_ - tb_) * 32;                                                                          \
                const int ldb_ = CAPMI_SEL(sgL, tl_), brem_ = CAPMI_SEL(sgK, tl_) - k0_;                                   \
                const float *B_ = CAPMI_SEL(sgB, tl_);                                                                     \
                _Pragma("unroll") for (int u_ = 0; u_ < 4; ++u_) {                                                         \
                    const float *src_ = BKC ? B_ + (size_t)wrow[u_] * ldb_ + k0_ + min(wk[u_], brem_ - 4)                  \
                                            : B_ + (size_t)(k0_ + min(wrow[u_], brem_ - 1)) * ldb_ + wk[u_];               \
                    __builtin_amdgcn_global_load_lds((gvoid)(uintptr_t)src_, (lvoid)(slot_ + LC_AB + (j + 4 * u_) * 1024), 16, 0, WAUX); \
                }                                                                                                          \
            }                                                                                                              \
        } while (0)
        // Ring protocol.  Stage k is computed between barrier k and barrier k + 2 (its consumers pass barrier k + 1 in the
        // middle), so after barrier k the slots of stages <= k - 2 are free: stage f may be requested once f <= k - 2 + NS.  Only
        // two stages are requested before the first hand-over (a loader stalls AT ISSUE while the CU's queue is full: the
        // hand-over of stage 0 must not sit behind the whole ring) and the ring is topped up two stages per interval.
        const int pre = min(nst, 2);
        for (int s = 0; s < pre; ++s) CAPMI_LC_ISSUE(s, s);
        CAPMI_LC_STAMP(1);
        int fill = pre;                              // next stage to request, into slot fslot = fill % LC_NS
        int fslot = pre;
        unsigned long long q_land = 0, q_bar = 0, q_issue = 0;      // ABL & 128: where a loader's cycles go
        for (int k = 0; k <= nst; ++k) {             // nst + 1 barriers (the last one ends the second half of the last stage)
            unsigned long long q0 = 0, q1 = 0, q2 = 0;
            if (ABL & 128) q0 = clock64();
            if (k < nst) {
                // stage k has landed when at most the DMAs of the younger stages are outstanding (a wave's LDS-DMAs retire in order)
                const int younger = fill - 1 - k;    // 0 .. LC_NS - 1
                if ((ABL & 5) == 5) { }
                else if (ABL & 5) {                  // ablation builds issue fewer DMAs per stage: wait for all of them
                    CAPMI_VMCNT(0);
                } else if (younger >= 4) CAPMI_VMCNT(4 * DPS);
                else if (younger == 3) CAPMI_VMCNT(3 * DPS);
                else if (younger == 2) CAPMI_VMCNT(2 * DPS);
                else if (younger == 1) CAPMI_VMCNT(DPS);
                else CAPMI_VMCNT(0);
            }
            if (k == 0) CAPMI_LC_STAMP(2);
            if (ABL & 128) q1 = clock64();
            __builtin_amdgcn_s_barrier();
            if (ABL & 128) q2 = clock64();
#pragma unroll
            for (int r = 0; r < 2; ++r)
                if (fill < nst && fill <= k - 2 + LC_NS) {
                    CAPMI_LC_ISSUE(fill, fslot);
                    ++fill;
                    fslot = fslot == LC_NS - 1 ? 0 : fslot + 1;
                }
            if (ABL & 128) { q_land += q1 - q0; q_bar += q2 - q1; q_issue += clock64() - q2; }
        }
        if ((ABL & 128) && blk_x == 0 && blk_y == 0 && lane == 0 && j == 0)
            printf("lc loader 0: %d stages; cycles per stage: waiting for its DMAs to land %.0f, in the barrier %.0f, issuing %.0f\n", nst,
                   (double)q_land / (nst + 1), (double)q_bar / (nst + 1), (double)q_issue / (nst + 1));
        __builtin_amdgcn_s_barrier();                // the two barriers of the parity merge below
        __builtin_amdgcn_s_barrier();
        CAPMI_LC_STAMP(3);
#undef CAPMI_LC_ISSUE
#undef CAPMI_SEL
        return;
    }

    // ===================================================== consumers ======================================================
    const int cg = widu & 3, par = widu >> 2;
    const int l31 = lane & 31, half = lane >> 5;
    const int cl = 32 * cg + l31;                    // column within the workgroup's tile
    const int col = n0 + cl;
    f32x16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc0[r] = acc1[r] = 0.f;
    // activation fragment of lane (row l31 [+32], half), k-step ks: piece 2 half + ks of the row, swizzled by (row >> 2) & 3
    const int sw = (l31 >> 2) & 3;
    const int ao[2] = {l31 * 64 + (((2 * half) ^ sw) << 4), l31 * 64 + (((2 * half + 1) ^ sw) << 4)};
    // weight fragment: [N][K]: 4 pieces 4 half + q of column cl, swizzled by (cl >> 1) & 7; [K][N]: 16 dwords k = 16 half + jj
    const int wsw = (cl >> 1) & 7;
    int wo[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) wo[q] = LC_AB + (BKC ? cl * 128 + (((4 * half + q) ^ wsw) << 4) : (16 * half + 4 * q) * 512 + cl * 4);
    // weight fragment of a stage: ds_read + exact 3-way split (x = h + m + l, truncated bf16 values; see gemm_x3.hip)
    auto wfrag = [&](const unsigned char *slot, u32x4 (&wb)[2][3]) {
        if (ABL & 64) return;                        // (probe: MFMAs on stale registers -- no LDS reads, no split)
        float bb[16];
        if (BKC) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 v = *reinterpret_cast<const f32x4 *>(slot + wo[q]);
                bb[4 * q] = v[0]; bb[4 * q + 1] = v[1]; bb[4 * q + 2] = v[2]; bb[4 * q + 3] = v[3];
            }
        } else {
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int e = 0; e < 4; ++e) bb[4 * q + e] = *reinterpret_cast<const float *>(slot + wo[q] + e * 512);
        }
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
    auto mma = [&](const unsigned char *slot, const u32x4 (&wb)[2][3], int ks) {
        bf16x8 bw[3], x0[3] = {}, x1[3] = {};
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
            bw[pl] = __builtin_bit_cast(bf16x8, wb[ks][pl]);
            if (ABL & 64) continue;
            const unsigned char *q = slot + pl * CAPMI_PL_PLANE_BYTES + ao[ks];
            x0[pl] = *reinterpret_cast<const bf16x8 *>(q);
            if (TM == 2) x1[pl] = *reinterpret_cast<const bf16x8 *>(q + 32 * 64);
        }
        if (ABL & 32) {                              // (probe: split + LDS reads stay, nothing is multiplied)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) asm volatile("" ::"v"(bw[pl]), "v"(x0[pl]), "v"(x1[pl]));
            return;
        }
        constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};      // six of nine cross terms, small ones first
#pragma unroll
        for (int t = 0; t < 6; ++t) {
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x0[PA[t]], bw[PB[t]], acc0, 0, 0, 0);
            if (TM == 2) acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x1[PA[t]], bw[PB[t]], acc1, 0, 0, 0);
        }
    };
    // (s_barrier is IntrNoMem for the compiler: the asm memory clobbers keep the LDS reads on their side of it)
    unsigned long long c_bar = 0, c_t0 = 0;
    if (ABL & 128) c_t0 = clock64();
#define CAPMI_LC_BARRIER() do { unsigned long long b0_ = 0; if (ABL & 128) b0_ = clock64(); asm volatile("" ::: "memory"); \
                                __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); if (ABL & 128) c_bar += clock64() - b0_; } while (0)
    int bars = 0;                                    // ring barriers passed (every wave passes nst + 1 of them)
    if (par) { CAPMI_LC_BARRIER(); ++bars; }         // barrier 0 hands over an even stage
    if (par == 0) CAPMI_LC_STAMP(1);
    {
        u32x4 wb[2][3] = {};
        int sidx = par;                              // slot of stage j
        for (int j = par; j < nst; j += 2) {
            CAPMI_LC_BARRIER();                      // barrier j: stage j has landed
            ++bars;
            if (j == 0) CAPMI_LC_STAMP(4);
            const unsigned char *slot = ldsb + sidx * LC_STAGE;
            sidx = sidx >= LC_NS - 2 ? sidx + 2 - LC_NS : sidx + 2;
            // Interval j: this wave fetches and SPLITS the stage's weights (VALU) while its SIMD partner of the other parity
            // multiplies the previous stage (matrix pipe); interval j + 1: the roles swap.  (With the first k-step's MFMAs in
            // interval j as well -- CAPMI_LC_OPT bit 1 clear... the r3a arrangement -- the interval was this wave's serial
            // read -> split -> read -> 12 MFMAs, 1.69k cycles, and the partner idled behind its 12: consumers, not the weight
            // stream, set the pace; the copy rate it happened to equal, 5.1 TB/s, is not the Infinity Cache's.)
            const bool late = (a.ablate & 2) != 0;
            if (!(ABL & 2)) { wfrag(slot, wb); if (!late) mma(slot, wb, 0); }
            CAPMI_LC_BARRIER();                      // barrier j + 1 (the other parity's hand-over, or the closing one)
            ++bars;
            if (!(ABL & 2)) { if (late) mma(slot, wb, 0); mma(slot, wb, 1); }
        }
    }
    while (bars <= nst) { CAPMI_LC_BARRIER(); ++bars; }
    if ((ABL & 128) && blk_x == 0 && blk_y == 0 && lane == 0 && cg == 0)
        printf("lc consumer parity %d: %d ring barriers, %.0f cycles per stage interval, of which %.0f in the barrier\n", par, bars,
               (double)(clock64() - c_t0) / bars, (double)c_bar / bars);
    // ---- the two parities of a column group meet in LDS (the ring is dead behind the first barrier) -------------------------
    constexpr int RP = LC_BN + 4;
    float *red = reinterpret_cast<float *>(ldsb);   // [64][RP]
    CAPMI_LC_BARRIER();
    if (par == 1) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
            red[row * RP + cl] = acc0[r];
            if (TM == 2) red[(32 + row) * RP + cl] = acc1[r];
        }
    }
    CAPMI_LC_BARRIER();
#undef CAPMI_LC_BARRIER
    if (par == 1) return;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
        acc0[r] += red[row * RP + cl];
        if (TM == 2) acc1[r] += red[(32 + row) * RP + cl];
    }
    if (ABL & 16) { asm volatile("" ::"v"(acc0[15]), "v"(acc1[15])); CAPMI_LC_STAMP(2); }
    if ((ABL & 8) || col >= a.N) {
#pragma unroll
        for (int r = 0; r < 16; ++r) { asm volatile("" ::"v"(acc0[r])); asm volatile("" ::"v"(acc1[r])); }
        return;
    }
    // C/D map of the 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
    if (a.to_partial) {
        float *out = a.partial + (size_t)z * a.M * a.N + col;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
            if (row < a.M) out[(size_t)row * a.N] = acc0[r];
            if (TM == 2 && row + 32 < a.M) out[(size_t)(row + 32) * a.N] = acc1[r];
        }
        if (ABL & 16) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); CAPMI_LC_STAMP(3); }
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
#undef CAPMI_LC_STAMP
}

}  // namespace
}  // namespace capmi_gemm
