// Fused additive region attention for gfx950 (AttModel.py:728-748 Attention.forward and its
// autograd backward).
//
// Forward: a workgroup owns `rpb` (1..8) caption rows of ONE image.  The image's projected tile
// p_att[b] (K x A) and feature tile att[b] (K x R) are read once per workgroup and reused, in
// registers, by its rows; the reference's repeat_tensors copy (models/utils.py:3-14) does not exist.
// rpb is chosen by the host so that the grid covers the chip: at SCST sizes (N = 60 rows) one row per
// workgroup (60 CUs busy instead of 10), at eval/XE sizes several rows per workgroup.  Workgroups of
// the same image are mapped to the SAME XCD (blockIdx % 8 is the observed XCD), so the n-fold reuse of
// an image tile is served by that XCD's L2 and HBM still sees each tile once:
// HBM traffic per launch = B*K*(A+R)*4 + N*(A+R+K)*4 bytes (SURVEY.md 8d "unique bytes").
// score -> softmax (-> mask renorm) -> context run in one launch; only att_h and the score matrix
// live in LDS.
//   phase 1  wave w owns regions k = w, w+8, ...: lanes stride A with 16-byte loads, tanh on the
//            VALU, wave64 shuffle reduction per (row, region)
//   phase 2  wave j normalises row j (K <= a few hundred): shuffle max/sum
//   phase 3  each lane owns 2 feature columns and streams the K rows of att[b] with 8-byte loads
#include "capmi_common.h"
#include "profile.h"
#include <hip/hip_ext.h>
#include "../../../include/capmi.h"

using namespace capmi;

namespace {

constexpr int NMAX = 8;       // max rows per image handled by one workgroup
constexpr int ATT_THREADS = 512;

// XCD-aware workgroup -> (image, row chunk): ids congruent mod 8 run on the same XCD, so all chunks
// of image b live on XCD b % 8.  Returns false for the padding ids when B % 8 != 0.
__device__ __forceinline__ bool decode_block(int B, int chunks, int &b, int &chunk) {
    const int id = blockIdx.x;
    const int xcd = id & 7, slot = id >> 3;
    b = (slot / chunks) * 8 + xcd;
    chunk = slot % chunks;
    return b < B;
}
inline int grid_blocks(int B, int chunks) { return ((B + 7) / 8) * 8 * chunks; }
inline int pick_rpb(int B, int n) {
    int rpb = (int)(((long long)B * n) / 256);
    if (rpb < 1) rpb = 1;
    if (rpb > n) rpb = n;
    if (rpb > NMAX) rpb = NMAX;
    return rpb;
}
// forward (r4): round UP, so that B * n in (256, 512] rows (XE at bs64 x 5 = 320) becomes <= 256 workgroups of two rows and stays on
// the register-resident kernel instead of 320 one-row workgroups of the streaming one
inline int pick_rpb_fwd(int B, int n) {
    int rpb = (int)(((long long)B * n + 255) / 256);
    if (rpb < 1) rpb = 1;
    if (rpb > n) rpb = n;
    if (rpb > NMAX) rpb = NMAX;
    return rpb;
}

// Latency note: at decode sizes these kernels are a chain of dependent memory round trips, so every
// phase issues ALL its loads for a group of regions before touching the data (a rolled `for k` loop
// exposes one ~1-2 us HBM/L2 latency per region: 36 of them per launch).
constexpr int SG = 5;     // regions a wave keeps in flight in the score phase
constexpr int CG = 12;    // att rows a lane keeps in flight in the context phase

__global__ __launch_bounds__(ATT_THREADS) void attention_fwd_kernel(
    const float *__restrict__ att_h, const float *__restrict__ p_att, const float *__restrict__ att,
    const float *__restrict__ mask, const float *__restrict__ w, const float *__restrict__ bptr,
    float *__restrict__ ctx, float *__restrict__ alpha, int B, int n_img, int rpb, int chunks, int K, int A, int R,
    const int *__restrict__ row_img, int h_splits, size_t h_stride, const float *__restrict__ h_bias,
    float *__restrict__ att_h_out, unsigned char *__restrict__ pl_ctx) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int AH = A > 1024 ? A : 1024;  // s_h doubles as the [NMAX][1024] combine buffer of the context phase
    float *s_h = lds;                    // [NMAX][AH]
    float *s_e = lds + (size_t)NMAX * AH; // [NMAX][K]
    int b, chunk, row0, n;
    if (row_img) {                       // ragged grouping: one row per workgroup, image from the map
        row0 = blockIdx.x;
        n = 1;
        b = row_img[row0];
    } else {
        if (!decode_block(B, chunks, b, chunk)) return;
        row0 = b * n_img + chunk * rpb;          // first caption row of this workgroup
        n = min(rpb, n_img - chunk * rpb);         // rows handled here (<= NMAX)
    }
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = blockDim.x >> 6;

    if (h_splits > 0) {
        // att_h arrives as the h2att GEMM's K-slice slabs: finish the reduction (+ bias) here and keep the
        // finished rows for the backward pass (no split-K reduce launch between the GEMM and this kernel)
        for (int i = threadIdx.x; i < n * A; i += blockDim.x) {
            const float *p = att_h + (size_t)row0 * A + i;
            float v = 0.f;
            for (int s0 = 0; s0 < h_splits; s0 += 8) {      // 8 independent slab loads in flight
                float part[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) part[u] = (s0 + u < h_splits) ? p[(s0 + u) * h_stride] : 0.f;
                v += ((part[0] + part[1]) + (part[2] + part[3])) + ((part[4] + part[5]) + (part[6] + part[7]));
            }
            if (h_bias) v += h_bias[i % A];
            s_h[i] = v;
            if (att_h_out) att_h_out[(size_t)row0 * A + i] = v;
        }
    } else {
        for (int i = threadIdx.x; i < n * A; i += blockDim.x) s_h[i] = att_h[(size_t)row0 * A + i];
    }

    const float bias = bptr ? bptr[0] : 0.f;
    const float *pb = p_att + (size_t)b * K * A;
    const bool fastA = (A % 4 == 0) && (A <= 512) && ((reinterpret_cast<uintptr_t>(p_att) & 15) == 0) &&
                       ((reinterpret_cast<uintptr_t>(w) & 15) == 0);
    if (fastA) {
        // lane covers a = 4*lane + 256*q, q in {0,1}
        const int a0 = lane * 4, a1 = a0 + 256;
        const bool v0 = a0 < A, v1 = a1 < A;
        f32x4 w0 = {0.f, 0.f, 0.f, 0.f}, w1 = w0;
        if (v0) w0 = *reinterpret_cast<const f32x4 *>(w + a0);
        if (v1) w1 = *reinterpret_cast<const f32x4 *>(w + a1);
#pragma unroll 1
        for (int base = 0; base < K; base += nw * SG) {   // block-uniform trip count (barrier inside)
            const int kb = base + wid;
            f32x4 p0[SG], p1[SG];
#pragma unroll
            for (int g = 0; g < SG; ++g) {
                const int k = kb + g * nw;
                p0[g] = p1[g] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (k < K) {
                    if (v0) p0[g] = *reinterpret_cast<const f32x4 *>(pb + (size_t)k * A + a0);
                    if (v1) p1[g] = *reinterpret_cast<const f32x4 *>(pb + (size_t)k * A + a1);
                }
            }
            __syncthreads();   // s_h visible (first trip); harmless afterwards (uniform trip count per block)
#pragma unroll
            for (int g = 0; g < SG; ++g) {
                const int k = kb + g * nw;
#pragma unroll
                for (int j = 0; j < NMAX; ++j) {
                    if (j < n) {
                        float acc = 0.f;
                        if (v0) {
                            const f32x4 h = *reinterpret_cast<const f32x4 *>(s_h + j * A + a0);
                            acc += w0[0] * tanh_f(p0[g][0] + h[0]) + w0[1] * tanh_f(p0[g][1] + h[1]) +
                                   w0[2] * tanh_f(p0[g][2] + h[2]) + w0[3] * tanh_f(p0[g][3] + h[3]);
                        }
                        if (v1) {
                            const f32x4 h = *reinterpret_cast<const f32x4 *>(s_h + j * A + a1);
                            acc += w1[0] * tanh_f(p1[g][0] + h[0]) + w1[1] * tanh_f(p1[g][1] + h[1]) +
                                   w1[2] * tanh_f(p1[g][2] + h[2]) + w1[3] * tanh_f(p1[g][3] + h[3]);
                        }
                        const float e = wave_sum(acc) + bias;
                        if (lane == 0 && k < K) s_e[j * K + k] = e;
                    }
                }
            }
        }
    } else {
        __syncthreads();
        for (int k = wid; k < K; k += nw) {
            float acc[NMAX];
#pragma unroll
            for (int j = 0; j < NMAX; ++j) acc[j] = 0.f;
            const float *pk = pb + (size_t)k * A;
            for (int a = lane; a < A; a += 64) {
                const float p = pk[a], wv = w[a];
#pragma unroll
                for (int j = 0; j < NMAX; ++j)
                    if (j < n) acc[j] += wv * tanh_f(p + s_h[j * A + a]);
            }
#pragma unroll
            for (int j = 0; j < NMAX; ++j) {
                if (j < n) {
                    const float e = wave_sum(acc[j]) + bias;
                    if (lane == 0) s_e[j * K + k] = e;
                }
            }
        }
    }
    __syncthreads();

    // softmax over regions, one wave per row
    for (int j = wid; j < n; j += nw) {
        float *e = s_e + j * K;
        float m = -INFINITY;
        for (int k = lane; k < K; k += 64) m = fmaxf(m, e[k]);
        m = wave_max(m);
        float s = 0.f;
        for (int k = lane; k < K; k += 64) {
            const float x = __expf(e[k] - m);
            e[k] = x;
            s += x;
        }
        s = wave_sum(s);
        const float inv = 1.f / s;
        if (mask) {
            const float *mb = mask + (size_t)b * K;
            float s2 = 0.f;
            for (int k = lane; k < K; k += 64) {
                const float x = e[k] * inv * mb[k];
                e[k] = x;
                s2 += x;
            }
            s2 = wave_sum(s2);
            for (int k = lane; k < K; k += 64) e[k] = e[k] / s2;
        } else {
            for (int k = lane; k < K; k += 64) e[k] = e[k] * inv;
        }
        for (int k = lane; k < K; k += 64) alpha[(size_t)(row0 + j) * K + k] = e[k];
    }
    __syncthreads();

    // context: 16-byte loads (8-byte accesses run at 0.54-0.70x the 16-byte rate on gfx950): the two halves of the
    // workgroup take alternate regions for the same 4-column group, then combine through LDS.
    const float *ab = att + (size_t)b * K * R;
    const bool vecR = (R % 4 == 0) && (R <= 1024) && ((reinterpret_cast<uintptr_t>(att) & 15) == 0) &&
                      ((reinterpret_cast<uintptr_t>(ctx) & 15) == 0);
    if (vecR) {
        const int cg = threadIdx.x & 255, half = threadIdx.x >> 8;      // ATT_THREADS == 512
        const int r = cg * 4;
        float c0[NMAX], c1[NMAX], c2[NMAX], c3[NMAX];
#pragma unroll
        for (int j = 0; j < NMAX; ++j) c0[j] = c1[j] = c2[j] = c3[j] = 0.f;
        if (r < R) {
            constexpr int CG4 = 6;               // 6 x 16 B in flight per lane
#pragma unroll 1
            for (int k0 = half; k0 < K; k0 += 2 * CG4) {
                f32x4 v0, v1, v2, v3, v4, v5;
                const f32x4 zz = {0.f, 0.f, 0.f, 0.f};
                v0 = (k0 + 0 < K) ? *reinterpret_cast<const f32x4 *>(ab + (size_t)(k0 + 0) * R + r) : zz;
                v1 = (k0 + 2 < K) ? *reinterpret_cast<const f32x4 *>(ab + (size_t)(k0 + 2) * R + r) : zz;
                v2 = (k0 + 4 < K) ? *reinterpret_cast<const f32x4 *>(ab + (size_t)(k0 + 4) * R + r) : zz;
                v3 = (k0 + 6 < K) ? *reinterpret_cast<const f32x4 *>(ab + (size_t)(k0 + 6) * R + r) : zz;
                v4 = (k0 + 8 < K) ? *reinterpret_cast<const f32x4 *>(ab + (size_t)(k0 + 8) * R + r) : zz;
                v5 = (k0 + 10 < K) ? *reinterpret_cast<const f32x4 *>(ab + (size_t)(k0 + 10) * R + r) : zz;
#define CAPMI_CTX_ACC(V, KK)                                                            \
    if ((KK) < K) {                                                                     \
        _Pragma("unroll") for (int j = 0; j < NMAX; ++j) {                              \
            if (j < n) {                                                                \
                const float al = s_e[j * K + (KK)];                                     \
                c0[j] += al * V[0]; c1[j] += al * V[1]; c2[j] += al * V[2]; c3[j] += al * V[3]; \
            }                                                                           \
        }                                                                               \
    }
                CAPMI_CTX_ACC(v0, k0 + 0) CAPMI_CTX_ACC(v1, k0 + 2) CAPMI_CTX_ACC(v2, k0 + 4)
                CAPMI_CTX_ACC(v3, k0 + 6) CAPMI_CTX_ACC(v4, k0 + 8) CAPMI_CTX_ACC(v5, k0 + 10)
#undef CAPMI_CTX_ACC
            }
        }
        __syncthreads();                       // all reads of s_h done; reuse it as the combine buffer [NMAX][1024]
        float *s_c = s_h;
        if (half == 1 && r < R) {
#pragma unroll
            for (int j = 0; j < NMAX; ++j)
                if (j < n) *reinterpret_cast<f32x4 *>(s_c + j * 1024 + r) = f32x4{c0[j], c1[j], c2[j], c3[j]};
        }
        __syncthreads();
        if (half == 0 && r < R) {
#pragma unroll
            for (int j = 0; j < NMAX; ++j) {
                if (j < n) {
                    const f32x4 o = *reinterpret_cast<const f32x4 *>(s_c + j * 1024 + r);
                    const f32x4 cv = f32x4{c0[j] + o[0], c1[j] + o[1], c2[j] + o[2], c3[j] + o[3]};
                    *reinterpret_cast<f32x4 *>(ctx + (size_t)(row0 + j) * R + r) = cv;
                    if (pl_ctx) pl_store4(pl_ctx, row0 + j, r, cv);     // A planes of ctx for the language-LSTM gate GEMM
                }
            }
        }
    } else {
        for (int r = threadIdx.x; r < R; r += blockDim.x) {
            float a0[NMAX];
#pragma unroll
            for (int j = 0; j < NMAX; ++j) a0[j] = 0.f;
            for (int k = 0; k < K; ++k) {
                const float v = ab[(size_t)k * R + r];
#pragma unroll
                for (int j = 0; j < NMAX; ++j)
                    if (j < n) a0[j] += s_e[j * K + k] * v;
            }
#pragma unroll
            for (int j = 0; j < NMAX; ++j)
                if (j < n) {
                    ctx[(size_t)(row0 + j) * R + r] = a0[j];
                    if (pl_ctx) pl_store1(pl_ctx, row0 + j, r, a0[j]);
                }
        }
    }
}


// ---- forward, round 2: every global load of the launch is issued before the first one is consumed ------------------------
// The kernel above is a chain of dependent round trips (h2att slabs -> barrier -> 5 regions of p_att -> barrier -> softmax ->
// three trips over att): 14 us for 4.7 MB at the SCST shape, every trip paying 1-2 us of L2 / Infinity-Cache latency.  None of
// those loads depends on a computed value: the addresses of the slabs, of the p_att tile and of the att tile are known at
// entry.  Here a thread issues, in the order it will need them,
//     <= 4 x 16 B of h2att K-slice slabs | 5 regions x 2 x 16 B of p_att | 18 regions x 16 B of att      (K <= 36 ... 40)
// branch-free (clamped addresses, invalid lanes zeroed afterwards, so the compiler's s_waitcnt vmcnt(n) stays exact) and only
// then reduces the slabs, scores, normalises and accumulates the context out of registers: ONE memory round trip per launch.
// Shapes outside the fast path (A > 512, R > 1024, K > 40, unaligned) keep the kernel above.
constexpr int V2_KMAX = 40;          // regions held in registers: 5 score regions per wave x 8 waves, 20 att rows per half
constexpr int V2_SREG = 5;
// CS (r3): the context's R columns of a row are split over CS workgroups (blockIdx.y): each scores and normalises
// the row itself (p_att in full) but fetches and accumulates only its R / CS columns of the image's att tile -- a CU pulls at most
// ~250 cache lines at a time, so a 217 KB working set per workgroup is three or four queue-fulls.
template <int NR, int CS = 1>      // rows per workgroup held in registers (1 or 2; larger groups use the kernel above)
__global__ __launch_bounds__(ATT_THREADS) void attention_fwd_v2_kernel(
    const float *__restrict__ att_h, const float *__restrict__ p_att, const float *__restrict__ att,
    const float *__restrict__ mask, const float *__restrict__ w, const float *__restrict__ bptr,
    float *__restrict__ ctx, float *__restrict__ alpha, int B, int n_img, int rpb, int chunks, int K, int A, int R,
    const int *__restrict__ row_img, int h_splits, size_t h_stride, const float *__restrict__ h_bias,
    float *__restrict__ att_h_out, unsigned char *__restrict__ pl_ctx) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *s_h = lds;                         // [NR][512]; later the context combine buffer [NR][1024]
    float *s_e = lds + (size_t)NR * 1024;     // [NR][V2_KMAX]
    float *s_p = s_e + NR * V2_KMAX;          // [4][NR * 512] slab partial sums of the four slab groups
    int b, chunk, row0, n;
    if (row_img) {
        row0 = blockIdx.x;
        n = 1;
        b = row_img[row0];
    } else {
        if (!decode_block(B, chunks, b, chunk)) return;
        row0 = b * n_img + chunk * rpb;
        n = min(rpb, n_img - chunk * rpb);
    }
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int nA4 = (n * A) >> 2;                                   // 16-byte pieces of this workgroup's att_h rows

    // ---- issue everything -------------------------------------------------------------------------------------------
    // (a) att_h: thread t owns piece t % 128 (+128 i) of the rows and slab group t / 128 (4 groups): <= 4 loads per piece
    const int sgrp = threadIdx.x >> 7, pc0 = threadIdx.x & 127;
    const int per_grp = h_splits > 0 ? (h_splits + 3) >> 2 : 1;     // slabs per group (<= 4 handled in registers)
    f32x4 hv[NR][4];
    const f32x4 zz = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < NR; ++i) {
        const int pc = pc0 + 128 * i;
        const int pcc = min(pc, max(nA4 - 1, 0));
        const float *p = att_h + (size_t)row0 * A + (size_t)pcc * 4;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            hv[i][u] = zz;
            if (i < n) {      // n is workgroup-uniform: no divergence, and unused rows cost no loads
                const int sl = h_splits > 0 ? min(sgrp * per_grp + u, h_splits - 1) : 0;
                hv[i][u] = *reinterpret_cast<const f32x4 *>(p + (size_t)sl * h_stride);
            }
        }
    }
    // (b) p_att: wave w scores regions w, w + 8, ...; lane covers a = 4 lane and 4 lane + 256
    const float *pb = p_att + (size_t)b * K * A;
    const int a0 = lane * 4, a1c = min(a0 + 256, A - 4);
    const int a0c = min(a0, A - 4);
    f32x4 p0[V2_SREG], p1[V2_SREG];
#pragma unroll
    for (int g = 0; g < V2_SREG; ++g) {
        const int k = min(wid + 8 * g, K - 1);
        p0[g] = *reinterpret_cast<const f32x4 *>(pb + (size_t)k * A + a0c);
        p1[g] = *reinterpret_cast<const f32x4 *>(pb + (size_t)k * A + a1c);
    }
    // (c) att: thread (cg, half) owns columns 4 cg .. 4 cg + 3 and regions half, half + GR, ...  (CS = 1: 256 column quads x 2
    //     region groups; CS workgroups per row: 256 / CS quads x 2 CS groups, the workgroup's quads start at blockIdx.y * QPW)
    constexpr int QD = 256 / CS, GR = 2 * CS, CREG = (V2_KMAX + GR - 1) / GR;
    const float *ab = att + (size_t)b * K * R;
    const int qpw = CS == 1 ? 256 : ((R >> 2) + CS - 1) / CS;       // column quads per workgroup
    const int q_lo = CS == 1 ? 0 : (int)blockIdx.y * qpw, q_hi = CS == 1 ? (R >> 2) : min((R >> 2), q_lo + qpw);
    const int cg = threadIdx.x & (QD - 1), half = threadIdx.x / QD;
    const int r = (q_lo + cg) * 4, rc = min(r, R - 4);
    const bool own = q_lo + cg < q_hi;
    f32x4 av[CREG];
#pragma unroll
    for (int g = 0; g < CREG; ++g) {
        const int k = min(half + GR * g, K - 1);
        av[g] = *reinterpret_cast<const f32x4 *>(ab + (size_t)k * R + rc);
    }
    const f32x4 w0 = *reinterpret_cast<const f32x4 *>(w + a0c), w1 = *reinterpret_cast<const f32x4 *>(w + a1c);
    const float bias = bptr ? bptr[0] : 0.f;
    __builtin_amdgcn_sched_barrier(0);

    // ---- att_h: finish the split-K reduction (+ bias), keep the rows for the backward pass ------------------------------
    if (h_splits > 0) {
#pragma unroll
        for (int i = 0; i < NR; ++i) {
            const int pc = pc0 + 128 * i;
            if (i < n && pc < nA4) {
                f32x4 v = zz;
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (u < per_grp && sgrp * per_grp + u < h_splits) v += hv[i][u];
                *reinterpret_cast<f32x4 *>(s_p + (size_t)sgrp * NR * 512 + pc * 4) = v;
            }
        }
        __syncthreads();
        for (int pc = threadIdx.x; pc < nA4; pc += ATT_THREADS) {
            f32x4 v = *reinterpret_cast<const f32x4 *>(s_p + pc * 4);
#pragma unroll
            for (int q = 1; q < 4; ++q) v += *reinterpret_cast<const f32x4 *>(s_p + (size_t)q * NR * 512 + pc * 4);
            if (h_bias) v += *reinterpret_cast<const f32x4 *>(h_bias + (pc * 4) % A);
            *reinterpret_cast<f32x4 *>(s_h + pc * 4) = v;
            if (att_h_out && (CS == 1 || blockIdx.y == 0)) *reinterpret_cast<f32x4 *>(att_h_out + (size_t)row0 * A + pc * 4) = v;
        }
    } else {
        if (sgrp == 0) {
#pragma unroll
            for (int i = 0; i < NR; ++i) {
                const int pc = pc0 + 128 * i;
                if (i < n && pc < nA4) *reinterpret_cast<f32x4 *>(s_h + pc * 4) = hv[i][0];
            }
        }
    }
    __syncthreads();

    // ---- scores -------------------------------------------------------------------------------------------------------
    const bool v0 = a0 < A, v1 = a0 + 256 < A;
#pragma unroll
    for (int g = 0; g < V2_SREG; ++g) {
        const int k = wid + 8 * g;
#pragma unroll
        for (int j = 0; j < NR; ++j) {
            if (j < n) {
                float acc = 0.f;
                const f32x4 h0 = *reinterpret_cast<const f32x4 *>(s_h + j * A + a0c);
                const f32x4 h1 = *reinterpret_cast<const f32x4 *>(s_h + j * A + a1c);
                const float t0 = w0[0] * tanh_f(p0[g][0] + h0[0]) + w0[1] * tanh_f(p0[g][1] + h0[1]) +
                                 w0[2] * tanh_f(p0[g][2] + h0[2]) + w0[3] * tanh_f(p0[g][3] + h0[3]);
                const float t1 = w1[0] * tanh_f(p1[g][0] + h1[0]) + w1[1] * tanh_f(p1[g][1] + h1[1]) +
                                 w1[2] * tanh_f(p1[g][2] + h1[2]) + w1[3] * tanh_f(p1[g][3] + h1[3]);
                acc = (v0 ? t0 : 0.f) + (v1 ? t1 : 0.f);
                const float e = wave_sum(acc) + bias;
                if (lane == 0 && k < K) s_e[j * V2_KMAX + k] = e;
            }
        }
    }
    __syncthreads();

    // ---- softmax over regions (+ mask renormalisation), one wave per row ---------------------------------------------------
    for (int j = wid; j < n; j += 8) {
        float *e = s_e + j * V2_KMAX;
        const float x = lane < K ? e[lane] : -INFINITY;            // K <= 40 < 64: one element per lane
        const float m = wave_max(x);
        float ex = lane < K ? __expf(x - m) : 0.f;
        const float inv = 1.f / wave_sum(ex);
        ex *= inv;
        if (mask) {
            ex *= lane < K ? mask[(size_t)b * K + lane] : 0.f;
            ex = ex / wave_sum(ex);
        }
        if (lane < K) {
            e[lane] = ex;
            if (CS == 1 || blockIdx.y == 0) alpha[(size_t)(row0 + j) * K + lane] = ex;
        }
    }
    __syncthreads();

    // ---- context out of registers ------------------------------------------------------------------------------------------
    float c0[NR], c1[NR], c2[NR], c3[NR];
#pragma unroll
    for (int j = 0; j < NR; ++j) c0[j] = c1[j] = c2[j] = c3[j] = 0.f;
#pragma unroll
    for (int g = 0; g < CREG; ++g) {
        const int k = half + GR * g;
        if (k < K) {              // uniform per region group
#pragma unroll
            for (int j = 0; j < NR; ++j) {
                if (j < n) {
                    const float al = s_e[j * V2_KMAX + k];
                    c0[j] += al * av[g][0]; c1[j] += al * av[g][1]; c2[j] += al * av[g][2]; c3[j] += al * av[g][3];
                }
            }
        }
    }
    __syncthreads();                          // s_h / s_p are dead: the combine buffer [GR - 1][NR][QD * 4] (<= 4 * NR * 512 floats)
    float *s_c = CS == 1 ? s_h : s_p;
    if (half > 0 && own) {
#pragma unroll
        for (int j = 0; j < NR; ++j)
            if (j < n) *reinterpret_cast<f32x4 *>(s_c + ((size_t)(half - 1) * NR + j) * (QD * 4) + cg * 4) = f32x4{c0[j], c1[j], c2[j], c3[j]};
    }
    __syncthreads();
    if (half == 0 && own) {
#pragma unroll
        for (int j = 0; j < NR; ++j) {
            if (j < n) {
                f32x4 cv = f32x4{c0[j], c1[j], c2[j], c3[j]};
#pragma unroll
                for (int q = 1; q < GR; ++q) cv += *reinterpret_cast<const f32x4 *>(s_c + ((size_t)(q - 1) * NR + j) * (QD * 4) + cg * 4);
                *reinterpret_cast<f32x4 *>(ctx + (size_t)(row0 + j) * R + r) = cv;
                if (pl_ctx) pl_store4(pl_ctx, row0 + j, r, cv);         // A planes of ctx for the language-LSTM gate GEMM
            }
        }
    }
}

// ---- backward, one step: d_ctx -> d_e, d_att_h ------------------------------------------------
// With or without the mask renormalisation the softmax-input gradient is
//   d_e[k] = alpha[k] * (dalpha[k] - sum_k' alpha[k'] dalpha[k'])   (alpha = the FINAL weights),
// because the renorm u_k = alpha_sm,k m_k / S composes with the softmax Jacobian to the same form
// (masked regions have alpha = 0 and receive 0).
constexpr int DG = 2;     // regions (x4 16-byte loads) a wave keeps in flight in the dalpha phase
__global__ __launch_bounds__(ATT_THREADS) void attention_bwd_kernel(
    const float *__restrict__ d_ctx, int ld_dctx, const float *__restrict__ att_h, const float *__restrict__ alpha,
    const float *__restrict__ p_att, const float *__restrict__ att, const float *__restrict__ w,
    float *__restrict__ d_att_h, float *__restrict__ d_e, int B, int n_img, int rpb, int chunks, int K, int A, int R,
    const int *__restrict__ row_img, const float *__restrict__ x_slabs, int x_splits, size_t x_stride, int x_cols,
    float *__restrict__ x_out) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *s_dc = lds;                        // [NMAX][R4]  (R rounded up to a multiple of 4, zero padded)
    const int R4 = (R + 3) & ~3;
    float *s_de = s_dc + (size_t)NMAX * R4;   // [NMAX][K]  dalpha, then d_e
    int b, chunk, row0, n;
    if (row_img) {
        row0 = blockIdx.x;
        n = 1;
        b = row_img[row0];
    } else {
        if (!decode_block(B, chunks, b, chunk)) return;
        row0 = b * n_img + chunk * rpb;
        n = min(rpb, n_img - chunk * rpb);
    }
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = blockDim.x >> 6;
    if (x_slabs) {
        // d_ctx is the first R of x_cols columns of a dX GEMM left as K-slice slabs: finish the reduction for this
        // workgroup's rows over ALL x_cols columns (the other columns feed the LSTM-cell backward), publish the rows to
        // x_out [N, x_cols] (= d_ctx, ld_dctx) and keep the d_ctx part in LDS.  x_cols % 4 == 0, 16-byte aligned.
        const int q4 = x_cols >> 2;
        for (int i = threadIdx.x; i < n * q4; i += blockDim.x) {
            const int j = i / q4, c = (i - j * q4) * 4;
            const float *p = x_slabs + (size_t)(row0 + j) * x_cols + c;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            for (int s0 = 0; s0 < x_splits; s0 += 8) {      // 8 independent 16-byte slab loads in flight
                f32x4 part[8];
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    part[u] = (s0 + u < x_splits) ? *reinterpret_cast<const f32x4 *>(p + (size_t)(s0 + u) * x_stride)
                                                  : f32x4{0.f, 0.f, 0.f, 0.f};
                v += ((part[0] + part[1]) + (part[2] + part[3])) + ((part[4] + part[5]) + (part[6] + part[7]));
            }
            *reinterpret_cast<f32x4 *>(x_out + (size_t)(row0 + j) * x_cols + c) = v;
            if (c < R) {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (c + e < R4) s_dc[j * R4 + c + e] = (c + e < R) ? v[e] : 0.f;
            }
        }
    } else {
        for (int i = threadIdx.x; i < n * R4; i += blockDim.x) {
            const int j = i / R4, r = i % R4;
            s_dc[i] = r < R ? d_ctx[(size_t)(row0 + j) * ld_dctx + r] : 0.f;
        }
    }
    __syncthreads();
    const float *ab = att + (size_t)b * K * R;
    const bool fastR = (R % 4 == 0) && (R <= 1024) && ((reinterpret_cast<uintptr_t>(att) & 15) == 0);
    if (fastR) {
        for (int kb = wid; kb < K; kb += nw * DG) {
            f32x4 v[DG][4];
#pragma unroll
            for (int g = 0; g < DG; ++g) {
                const int k = kb + g * nw;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int r = lane * 4 + 256 * q;
                    v[g][q] = (k < K && r < R) ? *reinterpret_cast<const f32x4 *>(ab + (size_t)k * R + r)
                                               : f32x4{0.f, 0.f, 0.f, 0.f};
                }
            }
#pragma unroll
            for (int g = 0; g < DG; ++g) {
                const int k = kb + g * nw;
#pragma unroll
                for (int j = 0; j < NMAX; ++j) {
                    if (j < n) {
                        float acc = 0.f;
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const int r = lane * 4 + 256 * q;
                            if (r < R) {
                                const f32x4 dcv = *reinterpret_cast<const f32x4 *>(s_dc + j * R4 + r);
                                acc += v[g][q][0] * dcv[0] + v[g][q][1] * dcv[1] + v[g][q][2] * dcv[2] + v[g][q][3] * dcv[3];
                            }
                        }
                        const float sum = wave_sum(acc);
                        if (lane == 0 && k < K) s_de[j * K + k] = sum;
                    }
                }
            }
        }
    } else {
        for (int k = wid; k < K; k += nw) {
            float acc[NMAX];
#pragma unroll
            for (int j = 0; j < NMAX; ++j) acc[j] = 0.f;
            for (int r = lane; r < R; r += 64) {
                const float v = ab[(size_t)k * R + r];
#pragma unroll
                for (int j = 0; j < NMAX; ++j)
                    if (j < n) acc[j] += v * s_dc[j * R4 + r];
            }
#pragma unroll
            for (int j = 0; j < NMAX; ++j) {
                if (j < n) {
                    const float sum = wave_sum(acc[j]);
                    if (lane == 0) s_de[j * K + k] = sum;
                }
            }
        }
    }
    __syncthreads();
    for (int j = wid; j < n; j += nw) {
        const float *al = alpha + (size_t)(row0 + j) * K;
        float c = 0.f;
        for (int k = lane; k < K; k += 64) c += al[k] * s_de[j * K + k];
        c = wave_sum(c);
        for (int k = lane; k < K; k += 64) {
            const float de = al[k] * (s_de[j * K + k] - c);
            s_de[j * K + k] = de;
            d_e[(size_t)(row0 + j) * K + k] = de;
        }
    }
    __syncthreads();
    const float *pb = p_att + (size_t)b * K * A;
    for (int a = threadIdx.x; a < A; a += blockDim.x) {
        float acc[NMAX], hh[NMAX];
#pragma unroll
        for (int j = 0; j < NMAX; ++j) {
            acc[j] = 0.f;
            hh[j] = j < n ? att_h[(size_t)(row0 + j) * A + a] : 0.f;
        }
        for (int k0 = 0; k0 < K; k0 += CG) {
            float pv[CG];
#pragma unroll
            for (int g = 0; g < CG; ++g) pv[g] = (k0 + g < K) ? pb[(size_t)(k0 + g) * A + a] : 0.f;
#pragma unroll
            for (int g = 0; g < CG; ++g) {
                if (k0 + g < K) {
#pragma unroll
                    for (int j = 0; j < NMAX; ++j) {
                        if (j < n) {
                            const float t = tanh_f(pv[g] + hh[j]);
                            acc[j] += s_de[j * K + k0 + g] * (1.f - t * t);
                        }
                    }
                }
            }
        }
        const float wa = w[a];
#pragma unroll
        for (int j = 0; j < NMAX; ++j)
            if (j < n) d_att_h[(size_t)(row0 + j) * A + a] = wa * acc[j];
    }
}

// ---- backward, one step, ONE ROW per workgroup (the fused SCST rollout's row_img mode, or rpb == 1) ----------------
// Same math as attention_bwd_kernel, restructured around its memory round trips (the old kernel walked slabs -> att ->
// p_att as three dependent phases, 408 KB through one CU per row):
//  * every global operand of the later phases (the image's att tile for the wave's regions, the thread's p_att column,
//    att_h, w) is requested BEFORE the slab reduction starts and sits in registers when its phase begins;
//  * the dX slab reduction is spread over gridDim.y workgroups per row: role 0 finishes only the d_ctx columns it needs,
//    roles 1.. finish the remaining columns (dh_att | dh_lang of d_x2) on other CUs and exit.
constexpr int BW2_KMAX = 40, BW2_KPW = 5;     // regions, regions per wave (8 waves)
__global__ __launch_bounds__(ATT_THREADS) void attention_bwd_v2_kernel(
    const float *__restrict__ d_ctx, int ld_dctx, const float *__restrict__ att_h, const float *__restrict__ alpha,
    const float *__restrict__ p_att, const float *__restrict__ att, const float *__restrict__ w,
    float *__restrict__ d_att_h, float *__restrict__ d_e, int B, int n_img, int chunks, int K, int A, int R,
    const int *__restrict__ row_img, const float *__restrict__ x_slabs, int x_splits, size_t x_stride, int x_cols,
    float *__restrict__ x_out) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *s_dc = lds;                        // [R]
    float *s_de = lds + R;                    // [K]
    int b, chunk, row0;
    if (row_img) {
        row0 = blockIdx.x;
        b = row_img[row0];
    } else {
        if (!decode_block(B, chunks, b, chunk)) return;
        row0 = b * n_img + chunk;
    }
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int role = blockIdx.y;
    if (role > 0) {
        // columns [R, x_cols) in (gridDim.y - 1) equal shares of whole quads
        const int q_all = (x_cols - R) >> 2, per = (q_all + gridDim.y - 2) / (gridDim.y - 1);
        const int q_lo = (role - 1) * per, q_hi = min(q_lo + per, q_all);
        for (int q = q_lo + tid; q < q_hi; q += blockDim.x) {
            const int c = R + 4 * q;
            const float *p = x_slabs + (size_t)row0 * x_cols + c;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            for (int s0 = 0; s0 < x_splits; s0 += 8) {
                f32x4 part[8];
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    part[u] = (s0 + u < x_splits) ? *reinterpret_cast<const f32x4 *>(p + (size_t)(s0 + u) * x_stride)
                                                  : f32x4{0.f, 0.f, 0.f, 0.f};
                v += ((part[0] + part[1]) + (part[2] + part[3])) + ((part[4] + part[5]) + (part[6] + part[7]));
            }
            *reinterpret_cast<f32x4 *>(x_out + (size_t)row0 * x_cols + c) = v;
        }
        return;
    }
    // ---- (1) request everything the later phases read
    const float *ab = att + (size_t)b * K * R;
    f32x4 av[BW2_KPW][4];
#pragma unroll
    for (int g = 0; g < BW2_KPW; ++g) {
        const int k = wid + 8 * g;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int r = lane * 4 + 256 * q;
            av[g][q] = (k < K && r < R) ? *reinterpret_cast<const f32x4 *>(ab + (size_t)k * R + r) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    }
    const float *pb = p_att + (size_t)b * K * A;
    float pv[BW2_KMAX];
#pragma unroll
    for (int k = 0; k < BW2_KMAX; ++k) pv[k] = (k < K && tid < A) ? pb[(size_t)k * A + tid] : 0.f;
    const float hh = tid < A ? att_h[(size_t)row0 * A + tid] : 0.f;
    const float wa = tid < A ? w[tid] : 0.f;
    const float al = tid < K ? alpha[(size_t)row0 * K + tid] : 0.f;
    // ---- (2) d_ctx of this row: finish the slab reduction of its columns (or read it)
    if (x_slabs) {
        for (int q = tid; q < (R >> 2); q += blockDim.x) {
            const int c = 4 * q;
            const float *p = x_slabs + (size_t)row0 * x_cols + c;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            for (int s0 = 0; s0 < x_splits; s0 += 8) {
                f32x4 part[8];
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    part[u] = (s0 + u < x_splits) ? *reinterpret_cast<const f32x4 *>(p + (size_t)(s0 + u) * x_stride)
                                                  : f32x4{0.f, 0.f, 0.f, 0.f};
                v += ((part[0] + part[1]) + (part[2] + part[3])) + ((part[4] + part[5]) + (part[6] + part[7]));
            }
            *reinterpret_cast<f32x4 *>(x_out + (size_t)row0 * x_cols + c) = v;
            *reinterpret_cast<f32x4 *>(s_dc + c) = v;
        }
    } else {
        for (int r = tid; r < R; r += blockDim.x) s_dc[r] = d_ctx[(size_t)row0 * ld_dctx + r];
    }
    __syncthreads();
    // ---- (3) dalpha[k] = att[b,k,:] . d_ctx from the registers
#pragma unroll
    for (int g = 0; g < BW2_KPW; ++g) {
        const int k = wid + 8 * g;
        float acc = 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int r = lane * 4 + 256 * q;
            if (r < R) {
                const f32x4 dcv = *reinterpret_cast<const f32x4 *>(s_dc + r);
                acc += av[g][q][0] * dcv[0] + av[g][q][1] * dcv[1] + av[g][q][2] * dcv[2] + av[g][q][3] * dcv[3];
            }
        }
        const float sum = wave_sum(acc);
        if (lane == 0 && k < K) s_de[k] = sum;
    }
    __syncthreads();
    // ---- (4) softmax Jacobian (K <= 40 <= 64: one wave)
    if (wid == 0) {
        const float da = lane < K ? s_de[lane] : 0.f;
        const float c = wave_sum(al * da);
        if (lane < K) {
            const float de = al * (da - c);
            s_de[lane] = de;
            d_e[(size_t)row0 * K + lane] = de;
        }
    }
    __syncthreads();
    // ---- (5) d_att_h[a] = w[a] sum_k d_e[k] (1 - tanh^2(p_att[k,a] + att_h[a])) from the registers
    if (tid < A) {
        float acc = 0.f;
#pragma unroll
        for (int k = 0; k < BW2_KMAX; ++k) {
            if (k < K) {
                const float t = tanh_f(pv[k] + hh);
                acc += s_de[k] * (1.f - t * t);
            }
        }
        d_att_h[(size_t)row0 * A + tid] = wa * acc;
    }
}

// ---- backward, time-batched feature / parameter gradients --------------------------------------
// d_att[b, k, :] = sum over time and over the image's n caption rows of alpha[row, k] * d_ctx[row, :].
// grid (B, ceil(K/KCH), ceil(R/128)): every workgroup owns KCH regions x 128 columns of one image and streams its
// T*n rows once with n independent loads in flight (the first version ran B*4 workgroups and re-read d_ctx K/KCH
// times: 193 us for 4.8 MB).
constexpr int KCH = 12;
constexpr int DATT_T = 128;
__global__ __launch_bounds__(DATT_T) void attn_datt_kernel(const float *__restrict__ d_ctx_all, int ld_dctx,
                                                          const float *__restrict__ alpha_all, float *__restrict__ d_att,
                                                          int T, int N, int n, int K, int R) {   // N = rows per time slab
    const int b = blockIdx.x, kc = blockIdx.y * KCH;
    const int r = blockIdx.z * DATT_T + threadIdx.x;
    const int total = T * n;
    // the image's alpha[T*n rows][KCH regions] goes through LDS once (every thread needs all of it)
    extern __shared__ float s_al[];          // [total][KCH]
    for (int i = threadIdx.x; i < total * KCH; i += DATT_T) {
        const int row_i = i / KCH, q = i - row_i * KCH;
        const size_t row = (size_t)(row_i / n) * N + b * n + (row_i % n);
        s_al[i] = (kc + q < K) ? alpha_all[row * K + kc + q] : 0.f;
    }
    __syncthreads();
    if (r >= R) return;
    float acc[KCH];
#pragma unroll
    for (int q = 0; q < KCH; ++q) acc[q] = 0.f;
    for (int i0 = 0; i0 < total; i0 += 4) {
        float d[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = min(i0 + u, total - 1);
            const size_t row = (size_t)(i / n) * N + b * n + (i % n);
            d[u] = (i0 + u < total) ? d_ctx_all[row * ld_dctx + r] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const float *al = s_al + min(i0 + u, total - 1) * KCH;
#pragma unroll
            for (int q = 0; q < KCH; ++q) acc[q] += al[q] * d[u];
        }
    }
#pragma unroll
    for (int q = 0; q < KCH; ++q)
        if (kc + q < K) d_att[((size_t)b * K + kc + q) * R + r] = acc[q];
}

// d_w: alpha_net weight gradient, summed over every (image, region) workgroup.  dw_part ([B*K, A], optional): each workgroup
// writes its own row there and the caller column-sums it -- 184 000 atomicAdds on 512 addresses (360 per address, serialised
// in L2) were most of this kernel's 52 us.
template <int KB>
__global__ void attn_dpatt_kernel(const float *__restrict__ att_h_all, const float *__restrict__ d_e_all,
                                  const float *__restrict__ p_att, const float *__restrict__ w,
                                  float *__restrict__ d_p_att, float *__restrict__ d_w, float *__restrict__ dw_part, int T, int N,
                                  int n, int K, int A) {
    // grid (B * ceil(K / KB)); threads over a; a workgroup serves KB regions of its image.  The T*n rows of an image are walked 8 at a
    // time: the loads of a group are issued before the first tanh (one memory round trip per 8 rows instead of one per row: 51 ->
    // 15 us at T*n = 100), the sums keep their order.  KB = 4 (r4) when the grid stays above two workgroups per CU: every region's
    // workgroup re-read the image's att_h rows -- 36 x 13.8 MB through L2 at the XE batch, 176 us.
    const int kblocks = (K + KB - 1) / KB;
    const int b = blockIdx.x / kblocks, k0 = (blockIdx.x % kblocks) * KB;
    const int total = T * n;
    for (int a = threadIdx.x; a < A; a += blockDim.x) {
        float p[KB], acc[KB], accw[KB];
#pragma unroll
        for (int q = 0; q < KB; ++q) {
            p[q] = p_att[((size_t)b * K + min(k0 + q, K - 1)) * A + a];
            acc[q] = accw[q] = 0.f;
        }
        for (int i0 = 0; i0 < total; i0 += 8) {
            float de[8][KB], ah[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = min(i0 + u, total - 1);
                const size_t row = (size_t)(i / n) * N + b * n + (i % n);
#pragma unroll
                for (int q = 0; q < KB; ++q) de[u][q] = d_e_all[row * K + min(k0 + q, K - 1)];
                ah[u] = att_h_all[row * A + a];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (i0 + u < total) {
#pragma unroll
                    for (int q = 0; q < KB; ++q) {
                        const float th = tanh_f(p[q] + ah[u]);
                        acc[q] += de[u][q] * (1.f - th * th);
                        accw[q] += de[u][q] * th;
                    }
                }
        }
#pragma unroll
        for (int q = 0; q < KB; ++q) {
            if (k0 + q >= K) break;
            d_p_att[((size_t)b * K + k0 + q) * A + a] = w[a] * acc[q];
            if (dw_part) dw_part[((size_t)b * K + k0 + q) * A + a] = accw[q];
            else atomicAdd(&d_w[a], accw[q]);
        }
    }
}

__global__ void sum_all_kernel(const float *__restrict__ in, size_t count, float *__restrict__ out) {
    // one workgroup (deterministic order); 8 independent loads in flight per thread -- a plain strided loop is one dependent
    // round trip per element (57 us for the 240 000 values of a bs64 XE step)
    __shared__ float scratch[32];
    float s = 0.f;
    const size_t stride = blockDim.x;
    for (size_t i0 = threadIdx.x; i0 < count; i0 += 8 * stride) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const size_t i = i0 + u * stride;
            v[u] = i < count ? in[i] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) s += v[u];
    }
    s = block_sum(s, scratch);
    if (threadIdx.x == 0) out[0] = s;
}

}  // namespace

extern "C" {

static int attention_fwd_launch(const float *att_h, int h_splits, int64_t h_stride, const float *h_bias, float *att_h_out,
                                const float *p_att, const float *att, const float *mask, const float *w, const float *b,
                                float *ctx, float *alpha, int B, int n, int K, int A, int R, const int32_t *row_img, int N,
                                void *stream, unsigned char *pl_ctx = nullptr) {
    if (!att_h || !p_att || !att || !w || !ctx || !alpha || B <= 0 || K <= 0 || A <= 0 || R <= 0) return CAPMI_EINVAL;
    if (row_img ? N <= 0 : n <= 0) return CAPMI_EINVAL;
    if (!row_img) N = B * n;
    if (pl_ctx && N > 64) return CAPMI_EINVAL;
    const size_t lds = ((size_t)NMAX * (A > 1024 ? A : 1024) + (size_t)NMAX * K) * sizeof(float);
    if (lds > 64 * 1024) return CAPMI_EINVAL;
    // unique (algorithmic) bytes: image tiles once + per-row att_h in, ctx and alpha out
    const double abytes = 4.0 * ((double)B * K * (A + R) + (double)N * (A + R + K));
    const int rpb = row_img ? 1 : pick_rpb_fwd(B, n), chunks = row_img ? 1 : (n + rpb - 1) / rpb;
    hipEvent_t e0, e1;
    // round-2 kernel: all loads up front, one memory round trip (BASELINE shapes: K = 36, A = 512, R = 1000)
    static const int env_v2 = capmi::research("CAPMI_ATT_V2", 1);
    const bool al16 = ((reinterpret_cast<uintptr_t>(att_h) | reinterpret_cast<uintptr_t>(p_att) | reinterpret_cast<uintptr_t>(att) |
                        reinterpret_cast<uintptr_t>(w) | reinterpret_cast<uintptr_t>(ctx) | reinterpret_cast<uintptr_t>(h_bias) |
                        reinterpret_cast<uintptr_t>(att_h_out)) & 15) == 0;
    // (latency regime only: with more workgroups than CUs the 188-VGPR kernel's single resident workgroup per CU loses to
    //  the occupancy of the kernel above -- measured 27.8 vs 21.7 us at B = 512, 51 vs 37 us at B = 1024)
    if (env_v2 && al16 && rpb <= 2 && (row_img ? N : grid_blocks(B, chunks)) <= 256 && K <= V2_KMAX && A % 4 == 0 && A >= 4 && A <= 512 && R % 4 == 0 && R >= 4 && R <= 1024 &&
        h_splits <= 16 && (h_stride & 3) == 0) {
        const size_t lds2 = ((size_t)rpb * 1024 + (size_t)rpb * V2_KMAX + (size_t)4 * rpb * 512) * sizeof(float);
        const bool prof = capmi_prof::take_events(CAPMI_PROF_ATTENTION_FWD, &e0, &e1, abytes, (double)N * K * (2.0 * A + 2.0 * R));
        const int gx = row_img ? N : grid_blocks(B, chunks);
        // default 4 (when every workgroup still gets its own CU): 10.3 -> 9.3 us per launch at the SCST shape; CAPMI_ATT_CS=1: one
        // workgroup per row as in round 2
        static const int env_cs = capmi::research("CAPMI_ATT_CS", 4);
        const int cs = (env_cs == 2 || env_cs == 4) && gx * env_cs <= 256 ? env_cs : 1;     // every workgroup on its own CU
        const dim3 grid(gx, cs);
#define CAPMI_ATT_V2(NR_, CS_)                                                                                              \
        if (prof) hipExtLaunchKernelGGL((attention_fwd_v2_kernel<NR_, CS_>), grid, dim3(ATT_THREADS), lds2, (hipStream_t)stream, e0, \
                                        e1, 0, att_h, p_att, att, mask, w, b, ctx, alpha, B, n, rpb, chunks, K, A, R, row_img,  \
                                        h_splits, (size_t)h_stride, h_bias, att_h_out, pl_ctx);                                \
        else hipLaunchKernelGGL((attention_fwd_v2_kernel<NR_, CS_>), grid, dim3(ATT_THREADS), lds2, (hipStream_t)stream, att_h, p_att, \
                                att, mask, w, b, ctx, alpha, B, n, rpb, chunks, K, A, R, row_img, h_splits, (size_t)h_stride,   \
                                h_bias, att_h_out, pl_ctx)
        if (rpb == 1) {
            if (cs == 4) { CAPMI_ATT_V2(1, 4); } else if (cs == 2) { CAPMI_ATT_V2(1, 2); } else { CAPMI_ATT_V2(1, 1); }
        } else {
            if (cs == 4) { CAPMI_ATT_V2(2, 4); } else if (cs == 2) { CAPMI_ATT_V2(2, 2); } else { CAPMI_ATT_V2(2, 1); }
        }
#undef CAPMI_ATT_V2
        CAPMI_CHECK_LAUNCH();
        return 0;
    }
    if (capmi_prof::take_events(CAPMI_PROF_ATTENTION_FWD, &e0, &e1, abytes, (double)N * K * (2.0 * A + 2.0 * R)))
        hipExtLaunchKernelGGL(attention_fwd_kernel, dim3(row_img ? N : grid_blocks(B, chunks)), dim3(ATT_THREADS), lds,
                              (hipStream_t)stream, e0, e1, 0, att_h, p_att, att, mask, w, b, ctx, alpha, B, n, rpb, chunks, K,
                              A, R, row_img, h_splits, (size_t)h_stride, h_bias, att_h_out, pl_ctx);
    else
        hipLaunchKernelGGL(attention_fwd_kernel, dim3(row_img ? N : grid_blocks(B, chunks)), dim3(ATT_THREADS), lds,
                           (hipStream_t)stream, att_h, p_att, att, mask, w, b, ctx, alpha, B, n, rpb, chunks, K, A, R, row_img,
                           h_splits, (size_t)h_stride, h_bias, att_h_out, pl_ctx);
    CAPMI_CHECK_LAUNCH();
    return 0;
}

int capmi_attention_fwd(const float *att_h, const float *p_att, const float *att, const float *mask, const float *w,
                        const float *b, float *ctx, float *alpha, int B, int n, int K, int A, int R,
                        const int32_t *row_img, int N, void *stream) {
    return attention_fwd_launch(att_h, 0, 0, nullptr, nullptr, p_att, att, mask, w, b, ctx, alpha, B, n, K, A, R, row_img, N,
                                stream);
}

int capmi_attention_fwd_partial(const float *h_partial, int h_splits, int64_t h_stride, const float *h_bias,
                                float *att_h_out, const float *p_att, const float *att, const float *mask, const float *w,
                                const float *b, float *ctx, float *alpha, int B, int n, int K, int A, int R,
                                const int32_t *row_img, int N, void *stream) {
    if (h_splits < 1) return CAPMI_EINVAL;
    return attention_fwd_launch(h_partial, h_splits, h_stride, h_bias, att_h_out, p_att, att, mask, w, b, ctx, alpha, B, n, K,
                                A, R, row_img, N, stream);
}

int capmi_attention_fwd_partial_pl(const float *h_partial, int h_splits, int64_t h_stride, const float *h_bias,
                                   float *att_h_out, const float *p_att, const float *att, const float *mask, const float *w,
                                   const float *b, float *ctx, float *alpha, int B, int n, int K, int A, int R,
                                   const int32_t *row_img, int N, void *ctx_planes, void *stream) {
    if (h_splits < 1) return CAPMI_EINVAL;
    return attention_fwd_launch(h_partial, h_splits, h_stride, h_bias, att_h_out, p_att, att, mask, w, b, ctx, alpha, B, n, K,
                                A, R, row_img, N, stream, static_cast<unsigned char *>(ctx_planes));
}

static int attention_bwd_launch(const float *d_ctx, int ld_dctx, const float *x_slabs, int x_splits, int64_t x_stride, int x_cols,
                                float *x_out, const float *att_h, const float *alpha, const float *p_att, const float *att,
                                const float *w, float *d_att_h, float *d_e, int B, int n, int K, int A, int R,
                                const int32_t *row_img, int N, void *stream) {
    if (!att_h || !alpha || !p_att || !att || !w || !d_att_h || !d_e || B <= 0) return CAPMI_EINVAL;
    if (row_img ? N <= 0 : n <= 0) return CAPMI_EINVAL;
    if (!row_img) N = B * n;
    const size_t lds = ((size_t)NMAX * ((R + 3) & ~3) + (size_t)NMAX * K) * sizeof(float);
    if (lds > 64 * 1024) return CAPMI_EINVAL;
    const int rpb = row_img ? 1 : pick_rpb(B, n), chunks = row_img ? 1 : (n + rpb - 1) / rpb;
    static const int env_v2 = capmi::research("CAPMI_ATT_BWD_V2", 1);
    if (env_v2 && rpb == 1 && K <= BW2_KMAX && A <= ATT_THREADS && R % 4 == 0 && R <= 1024 &&
        (reinterpret_cast<uintptr_t>(att) & 15) == 0 && (!x_slabs || (x_cols - R) % 4 == 0)) {
        int roles = 1;
        if (x_slabs && x_cols > R) {
            roles = 1 + (x_cols - R + 1023) / 1024;
            if (roles > 4) roles = 4;
        }
        hipLaunchKernelGGL(attention_bwd_v2_kernel, dim3(row_img ? N : grid_blocks(B, chunks), roles), dim3(ATT_THREADS),
                           (size_t)(R + K) * sizeof(float), (hipStream_t)stream, d_ctx, ld_dctx, att_h, alpha, p_att, att, w,
                           d_att_h, d_e, B, n, chunks, K, A, R, row_img, x_slabs, x_splits, (size_t)x_stride, x_cols, x_out);
        CAPMI_CHECK_LAUNCH();
        return 0;
    }
    hipLaunchKernelGGL(attention_bwd_kernel, dim3(row_img ? N : grid_blocks(B, chunks)), dim3(ATT_THREADS), lds,
                       (hipStream_t)stream, d_ctx, ld_dctx, att_h, alpha, p_att, att, w, d_att_h, d_e, B, n, rpb, chunks, K,
                       A, R, row_img, x_slabs, x_splits, (size_t)x_stride, x_cols, x_out);
    CAPMI_CHECK_LAUNCH();
    return 0;
}

int capmi_attention_bwd(const float *d_ctx, int ld_dctx, const float *att_h, const float *alpha, const float *p_att,
                        const float *att, const float *mask, const float *w, float *d_att_h, float *d_e, int B, int n,
                        int K, int A, int R, const int32_t *row_img, int N, void *stream) {
    (void)mask;   // the masked renorm folds into the same Jacobian (see kernel comment)
    if (!d_ctx || ld_dctx < R) return CAPMI_EINVAL;
    return attention_bwd_launch(d_ctx, ld_dctx, nullptr, 0, 0, 0, nullptr, att_h, alpha, p_att, att, w, d_att_h, d_e, B, n, K, A,
                                R, row_img, N, stream);
}

int capmi_attention_bwd_partial(const float *x_slabs, int x_splits, int64_t x_stride, int x_cols, float *x_out,
                                const float *att_h, const float *alpha, const float *p_att, const float *att,
                                const float *w, float *d_att_h, float *d_e, int B, int n, int K, int A, int R,
                                const int32_t *row_img, int N, void *stream) {
    if (!x_slabs || !x_out || x_splits < 1 || x_cols < R) return CAPMI_EINVAL;
    if ((x_cols & 3) || (x_stride & 3) || ((reinterpret_cast<uintptr_t>(x_slabs) | reinterpret_cast<uintptr_t>(x_out)) & 15)) {
        // sizes the 16-byte slab loads cannot take (hidden size not a multiple of 4, odd row counts): finish the K-slice
        // reduction with the generic reduce launch, then run the Jacobian on the finished rows
        const int rows = row_img ? N : B * n;
        if (x_stride != (int64_t)rows * x_cols) return CAPMI_EINVAL;
        const int rc = capmi_splitk_reduce(x_slabs, x_splits, x_out, x_cols, rows, x_cols, nullptr, nullptr, nullptr, 1, nullptr, 0,
                                           0, stream);
        if (rc) return rc;
        return attention_bwd_launch(x_out, x_cols, nullptr, 0, 0, 0, nullptr, att_h, alpha, p_att, att, w, d_att_h, d_e, B, n, K, A,
                                    R, row_img, N, stream);
    }
    return attention_bwd_launch(x_out, x_cols, x_slabs, x_splits, x_stride, x_cols, x_out, att_h, alpha, p_att, att, w, d_att_h,
                                d_e, B, n, K, A, R, row_img, N, stream);
}

int capmi_attention_bwd_batched(const float *d_ctx_all, int ld_dctx, const float *att_h_all, const float *alpha_all,
                                const float *d_e_all, const float *p_att, const float *w, float *d_att,
                                float *d_p_att, float *d_w, float *d_b, int T, int B, int n, int N_stride, int K, int A,
                                int R, void *stream) {
    return capmi_attention_bwd_batched_ws(d_ctx_all, ld_dctx, att_h_all, alpha_all, d_e_all, p_att, w, d_att, d_p_att, d_w, d_b, T,
                                          B, n, N_stride, K, A, R, nullptr, stream);
}

int capmi_attention_bwd_batched_ws(const float *d_ctx_all, int ld_dctx, const float *att_h_all, const float *alpha_all,
                                   const float *d_e_all, const float *p_att, const float *w, float *d_att,
                                   float *d_p_att, float *d_w, float *d_b, int T, int B, int n, int N_stride, int K, int A,
                                   int R, float *dw_partial, void *stream) {
    if (!d_ctx_all || !att_h_all || !alpha_all || !d_e_all || !p_att || !w || !d_att || !d_p_att || !d_b) return CAPMI_EINVAL;
    if (!d_w && !dw_partial) return CAPMI_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const int N = N_stride > 0 ? N_stride : B * n;
    hipLaunchKernelGGL(attn_datt_kernel, dim3(B, (K + KCH - 1) / KCH, (R + DATT_T - 1) / DATT_T), dim3(DATT_T),
                       (size_t)T * n * KCH * sizeof(float), st, d_ctx_all, ld_dctx, alpha_all, d_att, T, N, n, K, R);
    CAPMI_CHECK_LAUNCH();
    if (!dw_partial) {
        hipError_t e = hipMemsetAsync(d_w, 0, (size_t)A * sizeof(float), st);
        if (e != hipSuccess) return (int)e;
    }
    if ((B * ((K + 3) / 4)) >= 512)
        hipLaunchKernelGGL(attn_dpatt_kernel<4>, dim3(B * ((K + 3) / 4)), dim3(A >= 512 ? 512 : 256), 0, st, att_h_all, d_e_all, p_att, w,
                           d_p_att, d_w, dw_partial, T, N, n, K, A);
    else
        hipLaunchKernelGGL(attn_dpatt_kernel<1>, dim3(B * K), dim3(A >= 512 ? 512 : 256), 0, st, att_h_all, d_e_all, p_att, w,
                           d_p_att, d_w, dw_partial, T, N, n, K, A);
    CAPMI_CHECK_LAUNCH();
    hipLaunchKernelGGL(sum_all_kernel, dim3(1), dim3(1024), 0, st, d_e_all, (size_t)T * N * K, d_b);
    CAPMI_CHECK_LAUNCH();
    return 0;
}

}  // extern "C"
