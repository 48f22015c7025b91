// Pieces shared by the two bf16x3 fat-GEMM kernels (gemm_x3.hip: 128 x 128 tile; gemm_x3w.hip: 256 x 128 tile): operand-split
// bit helpers, the unit descriptor of the persistent grids and the epilogue of one output unit.
#pragma once
#include "gemm_common.h"

namespace capmi_gemm {
namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef const float __attribute__((address_space(1))) *gcf;
typedef const f32x4 __attribute__((address_space(1))) *gcf4;
typedef const char __attribute__((address_space(1))) *gcb;

__device__ __forceinline__ uint32_t fbits(float x) { return __builtin_bit_cast(uint32_t, x); }
__device__ __forceinline__ float bfloat(uint32_t b) { return __builtin_bit_cast(float, b); }
// upper halves of two fp32 bit patterns -> one dword of two bf16 (v_perm_b32: bytes 2,3 of lo, bytes 2,3 of hi)
__device__ __forceinline__ uint32_t pack2(uint32_t lo, uint32_t hi) { return __builtin_amdgcn_perm(hi, lo, 0x07060302u); }

// element index of (row, k) inside a plane: 64-byte rows, 16-byte pieces swizzled by s = (row >> 2) & 3 (conflict-free ds_read_b128: the 16 lanes of a hardware lane group -- rows {0-3, 12-15, 20-27} / {4-11, 16-19, 28-31} of a 32-row block -- hit 16 distinct bank quads), and the row
// itself stored in slot row ^ s of its aligned group of four: a [K][rows] operand is staged as 4 x 4 blocks, so the 8 lanes that
// write one k quad hold rows 4i + p -- 256 bytes apart, i.e. the SAME 16 banks (8-way conflict on every ds_write_b64); with the
// slot XOR they spread over the four 64-byte windows (2-way, what the padded rows of gemm_x3.hip get).  Reads are unaffected: an
// aligned group of four rows still covers its own 256 bytes.
__device__ __forceinline__ int wswz(int row, int k) {
    const int s = (row >> 2) & 3;
    return (row ^ s) * 32 + ((((k >> 3) ^ s) & 3) << 3) + (k & 7);
}

struct Unit {
    int m0, n0, z, t_begin, nt;
};

// ---- epilogue of a unit of the TRANSPOSED product (r5, KArgs.transposed): the kernel multiplied with the operands swapped, so a
// unit's rows are C's COLUMNS.  In the 32x32 MFMA's C/D layout a lane holds 4 CONSECUTIVE rows of the product per register group
// (rows (r&3) + 8*(r>>2) + 4*(lane>>5)) = 4 consecutive columns of C: every access -- store, column bias, row bias, mask, addend -- is
// one 16-byte piece per lane (lanes differ in C's row: 64 rows x 16 bytes per instruction; L2 assembles the 128-byte lines from the
// 8 pieces the same wave writes).  Here a.M = columns of C, a.N = rows of C.
template <int NJ>
__device__ __forceinline__ void x3_epilogue_t(const KArgs &a, const Unit &un, const f32x16 (&acc)[2][NJ], int wm0, int wn0, int l31,
                                              int half) {
    const int Cr = a.N, Cc = a.M;                          // C's own shape
    const bool to_partial = a.to_partial != 0;
    float *out = to_partial ? a.partial + (size_t)un.z * Cr * Cc : a.C;
    const int ldo = to_partial ? Cc : a.ldc;
    const bool plain = to_partial || !(a.bias || a.bias2 || a.row_bias || a.relu || a.mul_mask || a.accumulate);
    // 16-byte accesses need Cc % 4 == 0 (columns come in aligned quads: tile offsets are multiples of 4), 16-byte aligned pointers
    // and pitches; otherwise element by element
    auto al16 = [](const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    const bool v4 = (Cc % 4 == 0) && (ldo % 4 == 0) && al16(out) &&
                    (plain || ((!a.bias || al16(a.bias)) && (!a.bias2 || al16(a.bias2)) && (!a.row_bias || al16(a.row_bias)) &&
                               (!a.mul_mask || al16(a.mul_mask)) && (!a.accumulate || (al16(a.addend) && a.ldc % 4 == 0))));
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int m = un.n0 + wn0 + 32 * j + l31;      // C's row
            if (m >= Cr) continue;
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                const int n = un.m0 + wm0 + 32 * i + 8 * rg + 4 * half;        // first of 4 consecutive columns of C
                if (n >= Cc) continue;
                f32x4 v = {acc[i][j][4 * rg], acc[i][j][4 * rg + 1], acc[i][j][4 * rg + 2], acc[i][j][4 * rg + 3]};
                if (v4) {
                    if (!plain) {
                        if (a.bias) v += *reinterpret_cast<const f32x4 *>(a.bias + n);
                        if (a.bias2) v += *reinterpret_cast<const f32x4 *>(a.bias2 + n);
                        if (a.row_bias) v += *reinterpret_cast<const f32x4 *>(a.row_bias + (size_t)(m / a.row_bias_div) * Cc + n);
                        if (a.relu) { v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f); }
                        if (a.mul_mask) v *= *reinterpret_cast<const f32x4 *>(a.mul_mask + (size_t)m * Cc + n);
                        if (a.accumulate) v += *reinterpret_cast<const f32x4 *>(a.addend + (size_t)m * a.ldc + n);
                    }
                    *reinterpret_cast<f32x4 *>(out + (size_t)m * ldo + n) = v;
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        if (n + e >= Cc) continue;
                        float x = v[e];
                        if (!plain) {
                            if (a.bias) x += a.bias[n + e];
                            if (a.bias2) x += a.bias2[n + e];
                            if (a.row_bias) x += a.row_bias[(size_t)(m / a.row_bias_div) * Cc + n + e];
                            if (a.relu) x = fmaxf(x, 0.f);
                            if (a.mul_mask) x *= a.mul_mask[(size_t)m * Cc + n + e];
                            if (a.accumulate) x += a.addend[(size_t)m * a.ldc + n + e];
                        }
                        out[(size_t)m * ldo + n + e] = x;
                    }
                }
            }
        }
}

// ---- epilogue of one output unit: C/D layout of the 32x32 MFMA: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
template <int NJ>
__device__ __forceinline__ void x3_epilogue(const KArgs &a, const Unit &un, const f32x16 (&acc)[2][NJ], int wm0, int wn0, int l31,
                                            int half) {
    const bool to_partial = a.to_partial != 0;
    float *out = to_partial ? a.partial + (size_t)un.z * a.M * a.N : a.C;
    const int ldo = to_partial ? a.N : a.ldc;
    const bool plain = to_partial || !(a.bias || a.bias2 || a.row_bias || a.relu || a.mul_mask || a.accumulate);
    // the column biases of all NJ column blocks are requested up front: read inside the (i, j) loop each was a memory round trip
    // in front of its block's stores (r4, scripts/tools_epilogue_bench.py: bias + ReLU cost an FFN GEMM 104 -> 134 us)
    float cbv[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int col = min(un.n0 + wn0 + 32 * j + l31, a.N - 1);
        cbv[j] = 0.f;
        if (!plain) {
            if (a.bias) cbv[j] += a.bias[col];
            if (a.bias2) cbv[j] += a.bias2[col];
        }
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int col = un.n0 + wn0 + 32 * j + l31;
            if (col >= a.N) continue;
            if (plain) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = un.m0 + wm0 + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * half;
                    if (row < a.M) out[(size_t)row * ldo + col] = acc[i][j][r];
                }
                continue;
            }
            const float cb = cbv[j];
            if (!(a.row_bias || a.mul_mask || a.accumulate)) {          // bias / ReLU only: nothing to fetch per element
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = un.m0 + wm0 + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * half;
                    if (row >= a.M) continue;
                    float v = acc[i][j][r] + cb;
                    if (a.relu) v = fmaxf(v, 0.f);
                    out[(size_t)row * ldo + col] = v;
                }
                continue;
            }
            if (!a.row_bias) continue;       // mask / addend only: handled below, pipelined over all blocks
            // (with a per-row bias: 8 rows per round trip, as in r4)
#pragma unroll
            for (int r0 = 0; r0 < 16; r0 += 8) {
                float mk[8], ad[8], rb[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int r = r0 + u;
                    const int row = min(un.m0 + wm0 + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * half, a.M - 1);
                    rb[u] = a.row_bias[(size_t)(row / a.row_bias_div) * a.N + col];
                    mk[u] = a.mul_mask ? a.mul_mask[(size_t)row * a.N + col] : 1.f;
                    ad[u] = a.accumulate ? a.addend[(size_t)row * a.ldc + col] : 0.f;      // (not to_partial: out == C)
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int r = r0 + u;
                    const int row = un.m0 + wm0 + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * half;
                    if (row >= a.M) continue;
                    float v = acc[i][j][r] + cb + rb[u];
                    if (a.relu) v = fmaxf(v, 0.f);
                    out[(size_t)row * ldo + col] = v * mk[u] + ad[u];
                }
            }
        }
    if (plain || a.row_bias || !(a.mul_mask || a.accumulate)) return;
    // Dropout mask / residual addend: one value per output element, a memory round trip per group of loads with the wave's share
    // of the matrix pipe idle (r4: 8 rows at a time = 8 round trips per 64 x 64 wave tile; mask + addend took an FFN GEMM from 83
    // to 111 us).  r5: the groups are software-pipelined -- the operands of group g+1 are requested BEFORE group g is combined and
    // stored (the compiler may not move a load across a store to a plain pointer in either direction, so the written order is the
    // issued order), which leaves one exposed round trip per tile.  Two register sets of 8 rows x {mask, addend}.
    constexpr int G = 4 * NJ;                              // (i, j, row half) groups of the wave tile
    auto issue = [&](int gi, float (&mk)[8], float (&ad)[8]) {
        const int i = gi / (2 * NJ), j = (gi / 2) % NJ, r0 = (gi & 1) * 8;
        const int col = min(un.n0 + wn0 + 32 * j + l31, a.N - 1);
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int r = r0 + u;
            const int row = min(un.m0 + wm0 + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * half, a.M - 1);
            mk[u] = a.mul_mask ? a.mul_mask[(size_t)row * a.N + col] : 1.f;
            ad[u] = a.accumulate ? a.addend[(size_t)row * a.ldc + col] : 0.f;      // (not to_partial: out == C)
        }
    };
    auto finish = [&](int gi, const float (&mk)[8], const float (&ad)[8]) {
        const int i = gi / (2 * NJ), j = (gi / 2) % NJ, r0 = (gi & 1) * 8;
        const int col = un.n0 + wn0 + 32 * j + l31;
        if (col >= a.N) return;
        const float cb = cbv[j];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int r = r0 + u;
            const int row = un.m0 + wm0 + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * half;
            if (row >= a.M) continue;
            float v = acc[i][j][r] + cb;
            if (a.relu) v = fmaxf(v, 0.f);
            out[(size_t)row * ldo + col] = v * mk[u] + ad[u];
        }
    };
    float mk0[8], ad0[8], mk1[8], ad1[8];
    issue(0, mk0, ad0);
#pragma unroll
    for (int gi = 0; gi < G; gi += 2) {
        issue(gi + 1, mk1, ad1);
        finish(gi, mk0, ad0);
        if (gi + 2 < G) issue(gi + 2, mk0, ad0);
        finish(gi + 1, mk1, ad1);
    }
}

}  // namespace
}  // namespace capmi_gemm
