// Device-side helpers shared by the gfx950 kernels of libcapmi.  CDNA4 only: wave = 64 lanes.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CAPMI_WAVE 64

#define CAPMI_CHECK_LAUNCH()                      \
    do {                                          \
        hipError_t e__ = hipGetLastError();       \
        if (e__ != hipSuccess) return (int)e__;   \
    } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace capmi {

// Wave-wide reductions on the DPP cross-lane path.  r4: the __shfl_xor butterflies these replace were six ds_bpermute round trips
// through the LDS crossbar each (about 100 cycles apiece, serialised by s_waitcnt): the softmax of the short-sequence MHA spent
// 2 000 cycles per query row in them (scripts/mha_ablate.py).  Here: xor 1 / xor 2 are quad permutes, xor 4 / xor 8 the half-row and
// row mirrors (after the quad steps every quad is uniform, so the mirrored lane holds the xor partner's value), and the four row
// results are combined through v_readlane.  Callers must be wave-converged, as __shfl required.  The result is wave-uniform and
// the reduction tree is the bottom-up butterfly (1, 2, 4, 8, 16, 32).
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
template <int CTRL>
__device__ __forceinline__ int dpp_i(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, true); }
constexpr int DPP_XOR1 = 0xB1, DPP_XOR2 = 0x4E, DPP_HALF_MIRROR = 0x141, DPP_ROW_MIRROR = 0x140;
__device__ __forceinline__ float lane_f(float v, int l) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l)); }

__device__ __forceinline__ float wave_sum(float v) {
    v += dpp_f<DPP_XOR1>(v);
    v += dpp_f<DPP_XOR2>(v);
    v += dpp_f<DPP_HALF_MIRROR>(v);
    v += dpp_f<DPP_ROW_MIRROR>(v);
    return (lane_f(v, 0) + lane_f(v, 16)) + (lane_f(v, 32) + lane_f(v, 48));
}
__device__ __forceinline__ float wave_max(float v) {
    v = fmaxf(v, dpp_f<DPP_XOR1>(v));
    v = fmaxf(v, dpp_f<DPP_XOR2>(v));
    v = fmaxf(v, dpp_f<DPP_HALF_MIRROR>(v));
    v = fmaxf(v, dpp_f<DPP_ROW_MIRROR>(v));
    return fmaxf(fmaxf(lane_f(v, 0), lane_f(v, 16)), fmaxf(lane_f(v, 32), lane_f(v, 48)));
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// block-wide reductions through a small LDS scratch (>= 32 floats); all threads get the result
__device__ __forceinline__ float block_sum(float v, float *scratch) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    __syncthreads();
    if (lane == 0) scratch[wid] = v;
    __syncthreads();
    float r = 0.f;
    for (int i = 0; i < nw; ++i) r += scratch[i];
    return r;
}
__device__ __forceinline__ float block_max(float v, float *scratch) {
    v = wave_max(v);
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    __syncthreads();
    if (lane == 0) scratch[wid] = v;
    __syncthreads();
    float r = scratch[0];
    for (int i = 1; i < nw; ++i) r = fmaxf(r, scratch[i]);
    return r;
}

// tanh / sigmoid from 1-ulp hardware building blocks: v_exp_f32 and v_rcp_f32 (an IEEE divide expands to ~10
// VALU instructions and made the attention score phase VALU-bound at XE sizes).  Absolute error ~2e-7, far inside
// the 1e-4 parity budget and below the fp32 accumulation-order noise of the GEMMs.
__device__ __forceinline__ float rcp_f(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float tanh_f(float x) {
    const float ax = fabsf(x);
    const float e = __expf(-2.0f * ax);           // in (0,1]
    const float t = (1.0f - e) * rcp_f(1.0f + e);
    return copysignf(t, x);
}
__device__ __forceinline__ float sigmoid_f(float x) {
    // stable for both signs: 1/(1+exp(-x))
    const float e = __expf(-fabsf(x));
    const float s = rcp_f(1.0f + e);              // sigmoid(|x|)
    return x >= 0.f ? s : 1.0f - s;
}

// Philox4x32-10 counter RNG (Salmon et al. 2011) -- in-kernel dropout masks and Gumbel noise.
struct Philox {
    uint32_t k0, k1;
    __device__ __forceinline__ Philox(uint64_t seed) : k0((uint32_t)seed), k1((uint32_t)(seed >> 32)) {}
    __device__ __forceinline__ static void round(uint32_t (&c)[4], uint32_t a, uint32_t b) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ a;
        const uint32_t n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ b;
        const uint32_t n3 = (uint32_t)p0;
        c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
    }
    __device__ __forceinline__ void gen(uint64_t ctr_lo, uint64_t ctr_hi, uint32_t (&out)[4]) const {
        uint32_t c[4] = {(uint32_t)ctr_lo, (uint32_t)(ctr_lo >> 32), (uint32_t)ctr_hi, (uint32_t)(ctr_hi >> 32)};
        uint32_t a = k0, b = k1;
#pragma unroll
        for (int i = 0; i < 10; ++i) {
            round(c, a, b);
            a += 0x9E3779B9u;
            b += 0xBB67AE85u;
        }
        out[0] = c[0]; out[1] = c[1]; out[2] = c[2]; out[3] = c[3];
    }
};
// The random stream of a kernel launched with `seed` while an epoch word is bound (capmi_rng_bind_epoch, capmi.h): a captured
// hipGraph freezes the seed argument, the epoch word in device memory moves on every replay (capmi_step_advance).
__device__ __forceinline__ uint64_t epoch_seed(uint64_t seed, const uint64_t *__restrict__ epoch) {
    return epoch ? seed + 0x9E3779B97F4A7C15ull * *epoch : seed;
}
// uniform in (0,1): never 0 so log() is finite
__device__ __forceinline__ float u01(uint32_t x) { return ((float)(x >> 8) + 0.5f) * (1.0f / 16777216.0f); }

// Environment switches of the library, all read once per process through these three helpers:
//  * knob():     the documented product switches: CAPMI_GEMM_X3, CAPMI_ARES_X3, CAPMI_LC, CAPMI_GEMM_LOG and -- r5 --
//                CAPMI_X3_TILE (fat-GEMM tiling: 0 by cost / 128 / 256), CAPMI_X3_SWAP, CAPMI_BATCHED_XT, CAPMI_BWD_SIDE.
//  * research(): tuning constants of experiments (grid sizes, kernel flavours).  The product build compiles them to their
//                measured-best defaults; only a -DCAPMI_VARIANTS build (scripts/build_variants.sh) reads the environment,
//                and says so on stderr for every variable it finds set.
//  * ablate_env(): profiling ablations (CAPMI_*_ABLATE) drop parts of a kernel's work to time the rest: results are WRONG
//                by design; -DCAPMI_VARIANTS builds only, loud on stderr.
// the epoch word bound by capmi_rng_bind_epoch (defined in pointwise.hip); every launch of a seed-taking kernel passes it on
const uint64_t *rng_epoch();
static inline int knob(const char *name, int dflt) {
    const char *e = getenv(name);
    return e ? atoi(e) : dflt;
}
#ifdef CAPMI_VARIANTS
static inline int research(const char *name, int dflt) {
    const char *e = getenv(name);
    if (e) fprintf(stderr, "*** capmi (variants build): research switch %s=%s ***\n", name, e);
    return e ? atoi(e) : dflt;
}
static inline int ablate_env(const char *name) {
    const char *e = getenv(name);
    int v = e ? atoi(e) : 0;
    if (v)
        fprintf(stderr, "\n*** capmi: %s=%d -- PROFILING ABLATION ACTIVE, kernel results are deliberately incomplete; "
                        "unset it for any real run ***\n\n", name, v);
    return v;
}
#else
static inline constexpr int research(const char *, int dflt) { return dflt; }
static inline constexpr int ablate_env(const char *) { return 0; }
#endif
}  // namespace capmi

// ---- "A planes": an activation matrix X[M <= 64, K] pre-split for the decode GEMMs (gemm_ares.hip, round 3) -------------
// The decode GEMMs run fp32 through the bf16 matrix pipe by an exact 3-way split x = h + m + l (three truncated bf16
// values).  Round 2 split the activation slice inside every one of the 32 column-block workgroups that share it (VGPR round
// trip + ~5.5 VALU ops per element + LDS writes before the first MFMA could start).  Now the PRODUCER of an activation (LSTM
// cell, attention, select/embed, cell backward) writes the three planes once, already in the LDS image the GEMM wants, and
// the GEMM copies its K slice with LDS-DMA (global_load_lds_dwordx4: no VGPRs, no VALU, no ds_write).
// Layout: K is cut in chunks of 32; chunk c = [plane 3][row 64][32 bf16 = 64 B] = CAPMI_PL_CHUNK_BYTES.  Inside a row the four
// 16-byte pieces (8 k each) are XOR-swizzled by (row >> 2) & 3 so that the MFMA fragment reads (ds_read_b128: 16-lane groups
// over 16 rows, same piece) touch all 64 banks exactly once.  Rows >= M and k >= K are never written and stay zero (the
// buffers are zero-filled once when they are allocated).
#define CAPMI_PL_CHUNK_BYTES 12288
#define CAPMI_PL_PLANE_BYTES 4096

namespace capmi {
__device__ __forceinline__ uint32_t f2u(float x) { return __builtin_bit_cast(uint32_t, x); }
__device__ __forceinline__ float u2f(uint32_t x) { return __builtin_bit_cast(float, x); }
// byte offset of element (row, k) inside plane 0
__device__ __forceinline__ size_t pl_offset(int row, int k) {
    const int kk = k & 31;
    return (size_t)(k >> 5) * CAPMI_PL_CHUNK_BYTES + row * 64 + ((((kk >> 3) ^ (row >> 2)) & 3) << 4) + ((kk & 7) << 1);
}
__device__ __forceinline__ void pl_store1(unsigned char *pl, int row, int k, float x) {
    const uint32_t h = f2u(x) & 0xffff0000u;
    const float r1 = x - u2f(h);
    const uint32_t m = f2u(r1) & 0xffff0000u;
    const uint32_t l = f2u(r1 - u2f(m));
    unsigned char *o = pl + pl_offset(row, k);
    *reinterpret_cast<unsigned short *>(o) = (unsigned short)(h >> 16);
    *reinterpret_cast<unsigned short *>(o + CAPMI_PL_PLANE_BYTES) = (unsigned short)(m >> 16);
    *reinterpret_cast<unsigned short *>(o + 2 * CAPMI_PL_PLANE_BYTES) = (unsigned short)(l >> 16);
}
// four consecutive k (k % 4 == 0): one 8-byte store per plane
__device__ __forceinline__ void pl_store4(unsigned char *pl, int row, int k, f32x4 v) {
    typedef unsigned int u32x2_ __attribute__((ext_vector_type(2)));
    uint32_t h[4], m[4], l[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        h[e] = f2u(v[e]) & 0xffff0000u;
        const float r1 = v[e] - u2f(h[e]);
        m[e] = f2u(r1) & 0xffff0000u;
        l[e] = f2u(r1 - u2f(m[e]));
    }
    unsigned char *o = pl + pl_offset(row, k);
    *reinterpret_cast<u32x2_ *>(o) = u32x2_{(h[0] >> 16) | (h[1] & 0xffff0000u), (h[2] >> 16) | (h[3] & 0xffff0000u)};
    *reinterpret_cast<u32x2_ *>(o + CAPMI_PL_PLANE_BYTES) = u32x2_{(m[0] >> 16) | (m[1] & 0xffff0000u), (m[2] >> 16) | (m[3] & 0xffff0000u)};
    *reinterpret_cast<u32x2_ *>(o + 2 * CAPMI_PL_PLANE_BYTES) = u32x2_{(l[0] >> 16) | (l[1] & 0xffff0000u), (l[2] >> 16) | (l[3] & 0xffff0000u)};
}
}  // namespace capmi
