// Non-GEMM kernels of the Transformer captioner on gfx950 (TransformerModel.py): the reference's own
// LayerNorm formula, short-sequence multi-head attention (Tq <= 32 queries x Tk <= 128 keys per head fit one
// workgroup's LDS; K = 36 regions / T <= 21 tokens in the BASELINE configs), token embedding + sinusoidal
// position, row log-softmax.  Cross-attention reads the per-image memory K/V once per image and serves the n
// caption rows of that image from LDS (no repeat_tensors, TransformerModel.py:330-334).
#include "capmi_common.h"
#include "embed_bwd_det.h"
#include "../../../include/capmi.h"

using namespace capmi;

namespace {

// ---------------------------------------------------------------- LayerNorm (one wave per row)
__global__ void layernorm_fwd_kernel(const float *__restrict__ x, const float *__restrict__ a, const float *__restrict__ b,
                                     float *__restrict__ y, float *__restrict__ mean, float *__restrict__ inv, int M, int D,
                                     float eps) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = blockDim.x >> 6;
    for (int r = blockIdx.x * nw + wid; r < M; r += gridDim.x * nw) {
        const float *xr = x + (size_t)r * D;
        float s = 0.f;
        for (int c = lane; c < D; c += 64) s += xr[c];
        const float mu = wave_sum(s) / D;
        float v = 0.f;
        for (int c = lane; c < D; c += 64) {
            const float d = xr[c] - mu;
            v += d * d;
        }
        const float sd = sqrtf(wave_sum(v) / (D - 1));      // torch.std: unbiased
        const float iv = 1.f / (sd + eps);                   // eps OUTSIDE the sqrt (TransformerModel.py:87)
        for (int c = lane; c < D; c += 64) y[(size_t)r * D + c] = a[c] * (xr[c] - mu) * iv + b[c];
        if (lane == 0) {
            mean[r] = mu;
            inv[r] = iv;
        }
    }
}

__global__ void layernorm_bwd_kernel(const float *__restrict__ dy, const float *__restrict__ x, const float *__restrict__ a,
                                     const float *__restrict__ mean, const float *__restrict__ inv, float *__restrict__ dx,
                                     int accumulate, float *__restrict__ g_scaled, int M, int D, float eps) {
    // y = a * xc * iv + b, iv = 1/(s+eps), s = sqrt(sum xc^2/(D-1)):
    // dx_j = iv (g_j - mean(g)) - iv^2 xc_j / ((D-1) s) * sum_i g_i xc_i,   g = dy * a
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = blockDim.x >> 6;
    for (int r = blockIdx.x * nw + wid; r < M; r += gridDim.x * nw) {
        const float mu = mean[r], iv = inv[r];
        const float sd = 1.f / iv - eps;
        float sg = 0.f, sgx = 0.f;
        for (int c = lane; c < D; c += 64) {
            const float g = dy[(size_t)r * D + c] * a[c];
            const float xc = x[(size_t)r * D + c] - mu;
            sg += g;
            sgx += g * xc;
        }
        sg = wave_sum(sg);
        sgx = wave_sum(sgx);
        const float mg = sg / D;
        const float k2 = sd > 0.f ? iv * iv / ((D - 1) * sd) * sgx : 0.f;
        for (int c = lane; c < D; c += 64) {
            const size_t i = (size_t)r * D + c;
            const float xc = x[i] - mu;
            const float v = iv * (dy[i] * a[c] - mg) - k2 * xc;
            dx[i] = accumulate ? dx[i] + v : v;
            if (g_scaled) g_scaled[i] = dy[i] * xc * iv;
        }
    }
}

// Register-resident variants (r4): D <= 64 * NE.  A lane owns the elements lane + 64 i of its row -- the assignment and the summation
// order of the kernels above, so the results are bit-identical -- and issues EVERY load of the row (x, gain, bias / dy, dx) before
// the first one is consumed: one memory round trip per row instead of two or three passes of D / 64 dependent loads each (the
// decode-step LayerNorms of AoA, [60, 1024] rows, were 10 us forward / 19 us backward of pure latency; now 3.7 / 4.9).
template <int NE>
__global__ __launch_bounds__(256) void layernorm_fwd_reg_kernel(const float *__restrict__ x, const float *__restrict__ a,
                                                                const float *__restrict__ b, float *__restrict__ y,
                                                                float *__restrict__ mean, float *__restrict__ inv, int M, int D,
                                                                float eps) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = blockDim.x >> 6;
    float av[NE], bv[NE];
#pragma unroll
    for (int i = 0; i < NE; ++i) {
        const int c = lane + 64 * i;
        av[i] = c < D ? a[c] : 0.f;
        bv[i] = c < D ? b[c] : 0.f;
    }
    for (int r = blockIdx.x * nw + wid; r < M; r += gridDim.x * nw) {
        const float *xr = x + (size_t)r * D;
        float xv[NE];
#pragma unroll
        for (int i = 0; i < NE; ++i) {
            const int c = lane + 64 * i;
            xv[i] = c < D ? xr[c] : 0.f;
        }
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < NE; ++i) s += xv[i];
        const float mu = wave_sum(s) / D;
        float v = 0.f;
#pragma unroll
        for (int i = 0; i < NE; ++i) {
            if (lane + 64 * i < D) {
                const float d = xv[i] - mu;
                v += d * d;
            }
        }
        const float sd = sqrtf(wave_sum(v) / (D - 1));      // torch.std: unbiased
        const float iv = 1.f / (sd + eps);                   // eps OUTSIDE the sqrt (TransformerModel.py:87)
#pragma unroll
        for (int i = 0; i < NE; ++i) {
            const int c = lane + 64 * i;
            if (c < D) y[(size_t)r * D + c] = av[i] * (xv[i] - mu) * iv + bv[i];
        }
        if (lane == 0) {
            mean[r] = mu;
            inv[r] = iv;
        }
    }
}

template <int NE>
__global__ __launch_bounds__(256) void layernorm_bwd_reg_kernel(const float *__restrict__ dy, const float *__restrict__ x,
                                                                const float *__restrict__ a, const float *__restrict__ mean,
                                                                const float *__restrict__ inv, float *__restrict__ dx,
                                                                int accumulate, float *__restrict__ g_scaled, int M, int D,
                                                                float eps, float *__restrict__ part_a = nullptr,
                                                                float *__restrict__ part_b = nullptr) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = blockDim.x >> 6;
    float av[NE];
    // r6 (capmi_layernorm_bwd_parts): the wave's share of the parameter gradients d_a = colsum(dy (x - mean) inv), d_b = colsum(dy),
    // summed over ITS rows in row order and left as one partial row per wave -- [waves, D] instead of an [M, D] g_scaled written here and
    // read back (with dy) by a column-sum launch
    float pa[NE], pb[NE];
#pragma unroll
    for (int i = 0; i < NE; ++i) {
        av[i] = lane + 64 * i < D ? a[lane + 64 * i] : 0.f;
        pa[i] = pb[i] = 0.f;
    }
    for (int r = blockIdx.x * nw + wid; r < M; r += gridDim.x * nw) {
        const float mu = mean[r], iv = inv[r];
        float dv[NE], xc[NE], ov[NE];
#pragma unroll
        for (int i = 0; i < NE; ++i) {
            const int c = lane + 64 * i;
            const size_t j = (size_t)r * D + c;
            dv[i] = c < D ? dy[j] : 0.f;
            xc[i] = c < D ? x[j] : 0.f;
            ov[i] = (accumulate && c < D) ? dx[j] : 0.f;
        }
        const float sd = 1.f / iv - eps;
        float sg = 0.f, sgx = 0.f;
#pragma unroll
        for (int i = 0; i < NE; ++i) {
            if (lane + 64 * i < D) {
                const float g = dv[i] * av[i];
                xc[i] -= mu;
                sg += g;
                sgx += g * xc[i];
            }
        }
        sg = wave_sum(sg);
        sgx = wave_sum(sgx);
        const float mg = sg / D;
        const float k2 = sd > 0.f ? iv * iv / ((D - 1) * sd) * sgx : 0.f;
#pragma unroll
        for (int i = 0; i < NE; ++i) {
            const int c = lane + 64 * i;
            if (c < D) {
                const size_t j = (size_t)r * D + c;
                const float v = iv * (dv[i] * av[i] - mg) - k2 * xc[i];
                dx[j] = accumulate ? ov[i] + v : v;
                if (g_scaled) g_scaled[j] = dv[i] * xc[i] * iv;
                if (part_a) {
                    pa[i] += dv[i] * xc[i] * iv;
                    pb[i] += dv[i];
                }
            }
        }
    }
    if (part_a) {
        // the four waves' sums -> ONE partial row per workgroup, added in wave order (through LDS, reusing it for b after a)
        __shared__ float s_part[4][64 * NE];
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
#pragma unroll
            for (int i = 0; i < NE; ++i) s_part[wid][lane + 64 * i] = pass ? pb[i] : pa[i];
            __syncthreads();
            float *dst = (pass ? part_b : part_a) + (size_t)blockIdx.x * D;
            for (int c = threadIdx.x; c < D; c += blockDim.x) dst[c] = ((s_part[0][c] + s_part[1][c]) + s_part[2][c]) + s_part[3][c];
            __syncthreads();
        }
    }
}

// The same backward for a decode step of the AoA core (AoAModel.py:163-186), where BOTH inputs are still K-slice slabs of the GEMMs
// that produced them: dy = sum_s dy_slabs[s] (the query projection's dX, also written out: its column sums are the b_2 gradient)
// and the running gradient dx = sum_s acc_slabs[s] (row pitch acc_ld: the query half of d_cat) + the LayerNorm term.  Slabs are
// added in slab order starting from 0.f like splitk_reduce_kernel / split_halves_kernel do, so the results carry the same bits as
// reduce + split + capmi_layernorm_bwd -- two launches fewer per step.
template <int NE>
__global__ __launch_bounds__(256) void layernorm_bwd_slabs_kernel(const float *__restrict__ dy_slabs, int dy_splits, size_t dy_stride,
                                                                  float *__restrict__ dy_out, const float *__restrict__ x,
                                                                  const float *__restrict__ a, const float *__restrict__ mean,
                                                                  const float *__restrict__ inv, const float *__restrict__ acc_slabs,
                                                                  int acc_splits, size_t acc_stride, int acc_ld,
                                                                  float *__restrict__ dx, float *__restrict__ g_scaled, int M, int D,
                                                                  float eps) {
    // One workgroup per row (50 rows in a decode step: waves, not workgroups, are scarce).  Phase 1, all 256 threads: a thread owns
    // 16-byte pieces of the row and requests 8 slabs of each input before it adds the first (a slab per round trip was 145 us per
    // launch, four per round trip with a wave per row 18 us); the finished dy and dx-input rows go to LDS, dy also to dy_out.
    // Phase 2, wave 0: layernorm_bwd_reg_kernel's arithmetic on the LDS rows with ITS lane -> column mapping and reduction order,
    // so dx / g_scaled carry its bits.  Missing slabs add 0.f like splitk_reduce_kernel does.
    __shared__ __attribute__((aligned(16))) float s_dy[64 * NE], s_acc[64 * NE];
    const int r = blockIdx.x, lane = threadIdx.x & 63;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    const float *dyp = dy_slabs + (size_t)r * D, *acp = acc_slabs + (size_t)r * acc_ld;
    for (int q = threadIdx.x; q < (D >> 2); q += blockDim.x) {
        f32x4 d = zero4, o = zero4;
        for (int s0 = 0; s0 < max(dy_splits, acc_splits); s0 += 8) {
            f32x4 td[8], ta[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                td[u] = s0 + u < dy_splits ? *reinterpret_cast<const f32x4 *>(dyp + (size_t)(s0 + u) * dy_stride + 4 * q) : zero4;
                ta[u] = s0 + u < acc_splits ? *reinterpret_cast<const f32x4 *>(acp + (size_t)(s0 + u) * acc_stride + 4 * q) : zero4;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                d += td[u];
                o += ta[u];
            }
        }
        *reinterpret_cast<f32x4 *>(s_dy + 4 * q) = d;
        *reinterpret_cast<f32x4 *>(s_acc + 4 * q) = o;
        if (dy_out) *reinterpret_cast<f32x4 *>(dy_out + (size_t)r * D + 4 * q) = d;
    }
    __syncthreads();
    if (threadIdx.x >= 64) return;
    const float mu = mean[r], iv = inv[r];
    float dv[NE], xc[NE], ov[NE], av[NE];
#pragma unroll
    for (int i = 0; i < NE; ++i) {
        const int c = lane + 64 * i;
        av[i] = c < D ? a[c] : 0.f;
        xc[i] = c < D ? x[(size_t)r * D + c] : 0.f;
        dv[i] = c < D ? s_dy[c] : 0.f;
        ov[i] = c < D ? s_acc[c] : 0.f;
    }
    const float sd = 1.f / iv - eps;
    float sg = 0.f, sgx = 0.f;
#pragma unroll
    for (int i = 0; i < NE; ++i) {
        if (lane + 64 * i < D) {
            const float g = dv[i] * av[i];
            xc[i] -= mu;
            sg += g;
            sgx += g * xc[i];
        }
    }
    sg = wave_sum(sg);
    sgx = wave_sum(sgx);
    const float mg = sg / D;
    const float k2 = sd > 0.f ? iv * iv / ((D - 1) * sd) * sgx : 0.f;
#pragma unroll
    for (int i = 0; i < NE; ++i) {
        const int c = lane + 64 * i;
        if (c < D) {
            const size_t j = (size_t)r * D + c;
            const float v = iv * (dv[i] * av[i] - mg) - k2 * xc[i];
            dx[j] = ov[i] + v;
            if (g_scaled) g_scaled[j] = dv[i] * xc[i] * iv;
        }
    }
}

// ---------------------------------------------------------------- short-sequence MHA
constexpr int MHA_T = 256, MHA_T_BIG = 512, MHA_T_MAX = 1024;
// phase ablation for profiling (scripts/mha_ablate.py): only the research build (-DCAPMI_VARIANTS) has the switch
#ifdef CAPMI_VARIANTS
__device__ int g_mha_abl = 0;
#define MHA_ABL(b) (g_mha_abl & (b))
#else
#define MHA_ABL(b) false
#endif
static void mha_sync_ablation() {
#ifdef CAPMI_VARIANTS
    static int last = 0;
    const int a = capmi::research("CAPMI_MHA_ABL", 0);
    if (a != last) { (void)hipMemcpyToSymbol(HIP_SYMBOL(g_mha_abl), &a, sizeof(int)); last = a; }
#endif
}
// Launch shape (profiles/r04_mha_phases.md; CAPMI_MHA_MFMA, CAPMI_MHA_THREADS, CAPMI_MHA_LDS_KB force them in the research build):
//  * the contraction phases go to the matrix pipe when a pass has at least one 16-row tile of query rows; a decode step's 5 rows
//    stay on the vector loops;
//  * 8 waves per workgroup from 1 536 scores per pass up (a cross-attention's 105 x 36), 4 below (2 560 workgroups of 21 x 21
//    are faster small); a grid with fewer workgroups than the chip has CUs (AoA: 10 images x 8 heads) gets 8 or, from 1 024
//    scores up, 16 -- there the waves of ONE workgroup are all the parallelism a CU has;
//  * the query rows of a pass take what fits in 80 KB next to K/V (two workgroups per CU); when they do not fit, occupancy is 1
//    anyway and the pass takes the whole 160 KB.
static int mha_on_mfma(int rows) {
    const int f = capmi::research("CAPMI_MHA_MFMA", -1);
    return f >= 0 ? f : (rows >= 16);
}
static int mha_threads(int rows, int Tk, int64_t wgs) {
    const int f = capmi::research("CAPMI_MHA_THREADS", 0);
    if (f > 0) return f;
    const int64_t scores = (int64_t)rows * Tk;
    if (wgs < 256) return scores >= 1024 ? MHA_T_MAX : MHA_T_BIG;
    return scores >= 1536 ? MHA_T_BIG : MHA_T;
}
// query rows per pass: all of them when they fit the budget next to K/V, else equal chunks of whole 16-row tiles
static int mha_chunk(int total, int64_t fixed_f, int64_t per_f) {
    const int kb = capmi::research("CAPMI_MHA_LDS_KB", 0);
    const int64_t two_per_cu = (kb > 0 ? kb : 80) * 1024 / 4;
    if (fixed_f + total * per_f <= two_per_cu) return total;
    const int64_t budget = (kb > 0 ? kb : 160) * 1024 / 4;
    const int64_t room = budget > fixed_f ? (budget - fixed_f) / per_f : 0;
    if (total <= room) return total;
    if (room < 1) return 1;
    const int64_t n = (total + room - 1) / room;
    int64_t ch = (total + n - 1) / n;
    if (((ch + 15) & ~15) <= room) ch = (ch + 15) & ~15;
    return (int)ch;
}

// workgroup = (kv row, head): the K/V tile of one image (or caption) and head is staged in LDS once and serves ALL
// q_per_kv * Tq query rows that attend to it.  The query rows are flattened (caption-major) and processed CH at a time --
// as many as fit in LDS next to K/V, normally all of them -- so a decode step (Tq = 1, 5 captions per image) or a
// cross-attention (5 x 21 rows per image) is ONE pass of four block-wide phases instead of q_per_kv sequential passes.
// r4 (scripts/mha_ablate.py, profiles/r04_mha_phases.md): with 4-8 waves per CU these phases are bound by the instructions a wave
// issues, not by a pipe: the global row / caption / position of a query row come from a table in LDS instead of two integer
// divisions per element, row-shaped loops give 16 lanes to a row (no division by Tk, reductions inside a DPP row), and the large
// contractions run on the f32 MFMA.  The vector contractions read 16 bytes of LDS per instruction along the head dimension
// (dk % 4 == 0, row pitch dk + 4 floats: the 16 lanes of a ds_read_b128 phase hit 64 distinct banks).
// LDS: K [Tk][dk+4], V [Tk][dk+4], Q [CH][dk+4], S [CH][Tk+1], row tables 2 x [CH] (position, mask row; the global row of a
// query is kvr * q_per_kv * Tq + its flat index: consecutive)
__device__ __forceinline__ float dot4(const f32x4 a, const f32x4 b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2] + a[3] * b[3]; }
#define LDS4(p) (*reinterpret_cast<const f32x4 *>(p))
__device__ __forceinline__ int pow2_shift(int d) { return (d & (d - 1)) ? -1 : __builtin_ctz(d); }
__device__ __forceinline__ int idiv(int i, int d, int sh) { return sh >= 0 ? i >> sh : i / d; }

// C[M,N] = sum_k A(i,k) B(k,j) with both operands in LDS behind arbitrary strides, on v_mfma_f32_16x16x4_f32 (f32 in, f32
// accumulate: bitwise an fmaf chain over k).  One 16x16 tile per wave at a time, tiles strided over the waves; the operands of
// 8 k-steps are requested before the 8 MFMAs that consume them.  Lane l feeds A[i0 + l%16][k + l/16] and B[k + l/16][j0 + l%16]
// and owns C[i0 + 4 (l/16) + r][j0 + l%16], r = 0..3.  pre(i, j) is evaluated for the owned entries BEFORE the k loop (a global
// or LDS value the epilogue needs: its latency hides under the tile), ep(i, j, value, pre) after it, both for in-range entries
// only.  Out-of-range rows / columns read a clamped (valid, finite) address and are dropped; k past K reads a clamped address
// and A is zeroed.
template <typename Pre, typename Ep>
__device__ __forceinline__ void lds_mfma16(const float *a, int sa_i, int sa_k, const float *b, int sb_k, int sb_j, int M, int N, int K,
                                           Pre pre, Ep ep) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const int tn = (N + 15) >> 4, tiles = ((M + 15) >> 4) * tn;
    const int li = lane & 15, lk = lane >> 4;
    for (int t = wid; t < tiles; t += nw) {
        const int ti = t / tn, i0 = ti << 4, j0 = (t - ti * tn) << 4;
        const int j = j0 + li;
        float pv[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int i = i0 + 4 * lk + r;
            pv[r] = (i < M && j < N) ? pre(i, j) : 0.f;
        }
        const float *ap = a + min(i0 + li, M - 1) * sa_i + lk * sa_k;
        const float *bp = b + min(j, N - 1) * sb_j + lk * sb_k;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int k0 = 0; k0 < K; k0 += 32) {
            float av[8], bv[8];
            if (k0 + 32 <= K) {
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    av[u] = ap[(k0 + 4 * u) * sa_k];
                    bv[u] = bp[(k0 + 4 * u) * sb_k];
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u], bv[u], acc, 0, 0, 0);
            } else {
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int k = k0 + 4 * u + lk, kc = min(k, K - 1) - lk;
                    av[u] = ap[kc * sa_k];
                    bv[u] = bp[kc * sb_k];
                    if (k >= K) av[u] = 0.f;
                }
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    if (k0 + 4 * u < K) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u], bv[u], acc, 0, 0, 0);
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int i = i0 + 4 * lk + r;
            if (i < M && j < N) ep(i, j, acc[r], pv[r]);
        }
    }
}
// reductions over the 16 lanes of a DPP row (every lane of the row gets the result); capmi_common.h: the cross-lane path
__device__ __forceinline__ float row16_sum(float v) {
    v += dpp_f<DPP_XOR1>(v);
    v += dpp_f<DPP_XOR2>(v);
    v += dpp_f<DPP_HALF_MIRROR>(v);
    return v + dpp_f<DPP_ROW_MIRROR>(v);
}
__device__ __forceinline__ float row16_max(float v) {
    v = fmaxf(v, dpp_f<DPP_XOR1>(v));
    v = fmaxf(v, dpp_f<DPP_XOR2>(v));
    v = fmaxf(v, dpp_f<DPP_HALF_MIRROR>(v));
    return fmaxf(v, dpp_f<DPP_ROW_MIRROR>(v));
}

// K and V (and, in the backward, zeroed dK / dV) of (kv row, head) -> LDS, four trips of loads in flight
template <bool ZERO>
__device__ __forceinline__ void mha_stage_kv(const float *__restrict__ k, const float *__restrict__ v, size_t base, int kstride, int Tk,
                                             int d4, int d4sh, int P1, float *sK, float *sV, float *sdK, float *sdV) {
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    for (int i0 = threadIdx.x; i0 < Tk * d4; i0 += 4 * blockDim.x) {
        f32x4 kk[4], vv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = min(i0 + u * (int)blockDim.x, Tk * d4 - 1), j = idiv(i, d4, d4sh), c = (i - j * d4) * 4;
            const size_t gi = base + (size_t)j * kstride + c;
            kk[u] = *reinterpret_cast<const f32x4 *>(k + gi);
            vv[u] = *reinterpret_cast<const f32x4 *>(v + gi);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = i0 + u * (int)blockDim.x;
            if (i < Tk * d4) {
                const int j = idiv(i, d4, d4sh), c = (i - j * d4) * 4;
                *reinterpret_cast<f32x4 *>(sK + j * P1 + c) = kk[u];
                *reinterpret_cast<f32x4 *>(sV + j * P1 + c) = vv[u];
                if (ZERO) {
                    *reinterpret_cast<f32x4 *>(sdK + j * P1 + c) = zero4;
                    *reinterpret_cast<f32x4 *>(sdV + j * P1 + c) = zero4;
                }
            }
        }
    }
}

template <bool QSLABS>
__global__ __launch_bounds__(MHA_T_MAX) void mha_fwd_kernel(const float *__restrict__ q, const float *__restrict__ k,
                                                           const float *__restrict__ v, int ldkv, int kstride,
                                                           const uint8_t *__restrict__ mask, int mask_tq, int mask_per_q,
                                                           int causal, int q_pos0, const float *__restrict__ drop,
                                                           float *__restrict__ o, float *__restrict__ p, int q_per_kv, int Tq,
                                                           int Tk, int h, int dk, int CH, int qstride, int mfma, int q_splits,
                                                           size_t q_slab_stride, const float *__restrict__ q_bias,
                                                           float *__restrict__ q_out) {
    // q_splits > 0 (r5): q is still `q_splits` K-slice slabs (row pitch qstride, q_slab_stride floats apart) of the query projection
    // of an AoA decode step; each workgroup finishes its head's columns -- slabs in order from 0.f, then the bias, the bits of
    // splitk_reduce_kernel -- and writes them to q_out (pitch D) for the backward
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int D = h * dk, P1 = dk + 4, S1 = Tk + 1, d4 = dk >> 2, d4sh = pow2_shift(d4);
    float *sK = lds, *sV = sK + Tk * P1, *sQ = sV + Tk * P1, *sS = sQ + CH * P1;
    int *sTT = reinterpret_cast<int *>(sS + CH * S1), *sMR = sTT + CH;       // query row -> position t, mask row
    const int kvr = blockIdx.x, hd = blockIdx.y;
    const float scale = rsqrtf((float)dk);
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = blockDim.x >> 6, l16 = lane & 15, sub = lane >> 4;
    if (!MHA_ABL(1)) mha_stage_kv<false>(k, v, (size_t)kvr * ldkv + hd * dk, kstride, Tk, d4, d4sh, P1, sK, sV, nullptr, nullptr);
    const int R_all = q_per_kv * Tq;
    for (int row0 = 0; row0 < R_all; row0 += CH) {
        const int rows = min(CH, R_all - row0);
        if (row0) __syncthreads();          // the previous chunk's readers are done
        for (int lr = threadIdx.x; lr < rows; lr += blockDim.x) {
            const int gr = row0 + lr, r = kvr * q_per_kv + gr / Tq, t = gr % Tq;
            sTT[lr] = t;
            sMR[lr] = (mask_per_q ? r : kvr) * mask_tq + (mask_tq > 1 ? t : 0);
        }
        // (the query rows of a kv row are consecutive rows of q: row = kvr * R_all + row0 + lr -- the staging needs no table, so its
        //  loads go out with the K / V loads and one barrier covers tables, K / V and Q)
        const size_t qrow0 = (size_t)kvr * R_all + row0;
        if (!MHA_ABL(2))
        for (int i0 = threadIdx.x; i0 < rows * d4; i0 += 4 * blockDim.x) {
            f32x4 qq[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = min(i0 + u * (int)blockDim.x, rows * d4 - 1), lr = idiv(i, d4, d4sh), c = (i - lr * d4) * 4;
                qq[u] = *reinterpret_cast<const f32x4 *>(q + (qrow0 + lr) * qstride + hd * dk + c);
            }
            if constexpr (QSLABS) {
                // (a decode step has fewer 16-byte pieces than threads: pieces one at a time, 8 slabs of a piece in flight)
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int i = i0 + u * (int)blockDim.x;
                    if (i >= rows * d4) continue;
                    const int lr = idiv(i, d4, d4sh), c = (i - lr * d4) * 4;
                    const float *src = q + (qrow0 + lr) * qstride + hd * dk + c;
                    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
                    for (int s0 = 0; s0 < q_splits; s0 += 8) {
                        f32x4 tv[8];
#pragma unroll
                        for (int w = 0; w < 8; ++w)
                            tv[w] = s0 + w < q_splits ? *reinterpret_cast<const f32x4 *>(src + (size_t)(s0 + w) * q_slab_stride) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                        for (int w = 0; w < 8; ++w) acc += tv[w];
                    }
                    if (q_bias) acc += *reinterpret_cast<const f32x4 *>(q_bias + hd * dk + c);
                    if (q_out) *reinterpret_cast<f32x4 *>(q_out + (qrow0 + lr) * D + hd * dk + c) = acc;
                    qq[u] = acc;
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = i0 + u * (int)blockDim.x;
                if (i < rows * d4) {
                    const int lr = idiv(i, d4, d4sh), c = (i - lr * d4) * 4;
                    *reinterpret_cast<f32x4 *>(sQ + lr * P1 + c) = qq[u];
                }
            }
        }
        __syncthreads();
        // scores = scale * Q K^T, masked
        if (mfma) {
            if (!MHA_ABL(4))
            lds_mfma16(sQ, P1, 1, sK, 1, P1, rows, Tk, dk,
                       [&](int lr, int j) { return (mask && !MHA_ABL(8)) ? (float)mask[(size_t)sMR[lr] * Tk + j] : 1.f; },
                       [&](int lr, int j, float acc, float mk) {
                           const bool ok = mk != 0.f && !(causal && j > q_pos0 + sTT[lr]);
                           sS[lr * S1 + j] = ok ? acc * scale : -INFINITY;
                       });
        } else {
            for (int lr = wid * 4 + sub; lr < rows; lr += nw * 4) {        // 16 lanes per query row
                const float *qa = sQ + lr * P1;
                const size_t mrow = (size_t)sMR[lr] * Tk;
                const int jmax = causal ? q_pos0 + sTT[lr] : Tk;
                for (int j = l16; j < Tk; j += 16) {
                    bool ok = j <= jmax;
                    if (mask && !MHA_ABL(8)) ok = ok && mask[mrow + j] != 0;          // (requested before the arithmetic)
                    const float *ka = sK + j * P1;
                    float s = 0.f;
                    if (!MHA_ABL(4))
                    for (int c = 0; c < dk; c += 4) s += dot4(LDS4(qa + c), LDS4(ka + c));
                    sS[lr * S1 + j] = ok ? s * scale : -INFINITY;
                }
            }
        }
        __syncthreads();
        if (Tk <= 64 && !MHA_ABL(16)) {
            // softmax with 16 lanes per query row: a lane holds keys l16 + 16 e (e < 4) in registers, four rows per wave instruction
            for (int lr0 = wid * 4; lr0 < rows; lr0 += nw * 4) {
                const int lr = lr0 + sub, lrc = min(lr, rows - 1), t = sTT[lrc];
                const size_t pi0 = ((qrow0 + lrc - t) * h + (size_t)hd * Tq + t) * Tk;
                float x[4], dr[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int j = l16 + 16 * e;
                    const bool ok = lr < rows && j < Tk;
                    x[e] = ok ? sS[lrc * S1 + j] : -INFINITY;
                    dr[e] = (drop && ok && !MHA_ABL(32)) ? drop[pi0 + j] : 1.f;
                }
                const float m = row16_max(fmaxf(fmaxf(x[0], x[1]), fmaxf(x[2], x[3])));
                float sum = 0.f;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    x[e] = (l16 + 16 * e < Tk) ? __expf(x[e] - m) : 0.f;
                    sum += x[e];
                }
                const float is = 1.f / row16_sum(sum);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int j = l16 + 16 * e;
                    if (lr < rows && j < Tk) {
                        const float pr = x[e] * is;
                        if (p && !MHA_ABL(32)) p[pi0 + j] = pr;
                        sS[lr * S1 + j] = pr * dr[e];
                    }
                }
            }
        } else if (!MHA_ABL(16))
        for (int lr = wid; lr < rows; lr += nw) {          // softmax: one wave per query row, any Tk
            float *row = sS + lr * S1;
            const int t = sTT[lr];
            const size_t pi0 = ((qrow0 + lr - t) * h + (size_t)hd * Tq + t) * Tk;
            float m = -INFINITY;
            for (int j = lane; j < Tk; j += 64) m = fmaxf(m, row[j]);
            m = wave_max(m);
            float sum = 0.f;
            for (int j = lane; j < Tk; j += 64) {
                const float e = __expf(row[j] - m);
                row[j] = e;
                sum += e;
            }
            sum = wave_sum(sum);
            const float is = 1.f / sum;
            for (int j = lane; j < Tk; j += 64) {
                float pr = row[j] * is;
                if (p) p[pi0 + j] = pr;
                if (drop) pr *= drop[pi0 + j];
                row[j] = pr;
            }
        }
        __syncthreads();
        // o = P V
        if (mfma) {
            if (!MHA_ABL(64))
            lds_mfma16(sS, S1, 1, sV, P1, 1, rows, dk, Tk, [](int, int) { return 0.f; },
                       [&](int lr, int c, float acc, float) { if (!MHA_ABL(128)) o[(qrow0 + lr) * D + hd * dk + c] = acc; });
        } else
        for (int i = threadIdx.x; i < rows * d4; i += blockDim.x) {      // 4 head columns per thread
            const int lr = idiv(i, d4, d4sh), c = (i - lr * d4) * 4;
            const float *pr = sS + lr * S1;
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            if (!MHA_ABL(64))
            for (int j = 0; j < Tk; ++j) acc += pr[j] * LDS4(sV + j * P1 + c);
            if (!MHA_ABL(128))
            *reinterpret_cast<f32x4 *>(o + (qrow0 + lr) * D + hd * dk + c) = acc;
        }
    }
}

// backward, same decomposition; dK/dV accumulate over the chunks in LDS and are written once.
// LDS: K, V, dK, dV [Tk][dk+4]; Q, dO [CH][dk+4]; P, P*drop, dS [CH][Tk+1]; row table [CH] (position)
// MAXT: 1024 threads cap a thread at 128 registers, which this kernel exceeds (17 spilled, 72 bytes of scratch per lane); only the
// small-grid launch shape needs them, every other shape runs the 512-thread instance
template <int MAXT, bool DOSLABS = false>
__global__ __launch_bounds__(MAXT) void mha_bwd_kernel(const float *__restrict__ d_o, const float *__restrict__ q,
                                                           const float *__restrict__ k, const float *__restrict__ v, int ldkv,
                                                           int kstride, const float *__restrict__ p,
                                                           const float *__restrict__ drop, float *__restrict__ dq,
                                                           float *__restrict__ dk_out, float *__restrict__ dv_out, int dkv_ld,
                                                           int dkv_stride, int accumulate, int q_per_kv, int Tq, int Tk, int h,
                                                           int dk, int CH, int qstride, int dq_stride, int mfma, int do_ld,
                                                           int do_splits, size_t do_stride) {
    // do_ld / do_splits / do_stride (r5): d_o has row pitch do_ld and may still be `do_splits` K-slice slabs `do_stride` floats apart
    // (the attention half of d_cat in an AoA decode step); summed in slab order from 0.f like split_halves_kernel
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int D = DOSLABS ? do_ld : h * dk, P1 = dk + 4, S1 = Tk + 1, d4 = dk >> 2, d4sh = pow2_shift(d4);      // D: row pitch of d_o
    float *sK = lds, *sV = sK + Tk * P1, *sdK = sV + Tk * P1, *sdV = sdK + Tk * P1;
    float *sQ = sdV + Tk * P1, *sdO = sQ + CH * P1, *sP = sdO + CH * P1, *sPd = sP + CH * S1, *sdS = sPd + CH * S1;
    int *sTT = reinterpret_cast<int *>(sdS + CH * S1);
    const int kvr = blockIdx.x, hd = blockIdx.y;
    const float scale = rsqrtf((float)dk);
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    mha_stage_kv<true>(k, v, (size_t)kvr * ldkv + hd * dk, kstride, Tk, d4, d4sh, P1, sK, sV, sdK, sdV);
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = blockDim.x >> 6, l16 = lane & 15, sub = lane >> 4;
    const int R_all = q_per_kv * Tq;
    for (int row0 = 0; row0 < R_all; row0 += CH) {
        const int rows = min(CH, R_all - row0);
        if (row0) __syncthreads();
        for (int lr = threadIdx.x; lr < rows; lr += blockDim.x) {
            sTT[lr] = (row0 + lr) % Tq;
        }
        // (as in the forward: the staging below addresses rows without the table -- one barrier per pass)
        const size_t qrow0 = (size_t)kvr * R_all + row0;
        for (int i0 = threadIdx.x; i0 < rows * d4; i0 += 4 * blockDim.x) {
            f32x4 qq[4], oo[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = min(i0 + u * (int)blockDim.x, rows * d4 - 1), lr = idiv(i, d4, d4sh), c = (i - lr * d4) * 4;
                const size_t rt = qrow0 + lr;
                qq[u] = *reinterpret_cast<const f32x4 *>(q + rt * qstride + hd * dk + c);
                oo[u] = *reinterpret_cast<const f32x4 *>(d_o + rt * D + hd * dk + c);
            }
            if constexpr (DOSLABS) {
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int i = i0 + u * (int)blockDim.x;
                    if (i >= rows * d4) continue;
                    const int lr = idiv(i, d4, d4sh), c = (i - lr * d4) * 4;
                    const float *src = d_o + (qrow0 + lr) * D + hd * dk + c;
                    f32x4 acc = zero4;
                    for (int s0 = 0; s0 < do_splits; s0 += 8) {
                        f32x4 tv[8];
#pragma unroll
                        for (int w = 0; w < 8; ++w)
                            tv[w] = s0 + w < do_splits ? *reinterpret_cast<const f32x4 *>(src + (size_t)(s0 + w) * do_stride) : zero4;
#pragma unroll
                        for (int w = 0; w < 8; ++w) acc += tv[w];
                    }
                    oo[u] = acc;
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = i0 + u * (int)blockDim.x;
                if (i < rows * d4) {
                    const int lr = idiv(i, d4, d4sh), c = (i - lr * d4) * 4;
                    *reinterpret_cast<f32x4 *>(sQ + lr * P1 + c) = qq[u];
                    *reinterpret_cast<f32x4 *>(sdO + lr * P1 + c) = oo[u];
                }
            }
        }
        // P, P * dropout and (parked in dS until dP is formed) the dropout mask: 16 lanes per row, two rows' loads in flight
        if (!MHA_ABL(4096))
        for (int lr0 = wid * 8; lr0 < rows; lr0 += nw * 8) {
            for (int j0 = 0; j0 < Tk; j0 += 32) {
                float pv[4], dv[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int lr = min(lr0 + sub + 4 * (u >> 1), rows - 1), j = min(j0 + l16 + 16 * (u & 1), Tk - 1), t = (row0 + lr) % Tq;
                    const size_t pi = ((qrow0 + lr - t) * h + (size_t)hd * Tq + t) * Tk + j;
                    pv[u] = p[pi];
                    dv[u] = drop ? drop[pi] : 1.f;
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int lr = lr0 + sub + 4 * (u >> 1), j = j0 + l16 + 16 * (u & 1);
                    if (lr < rows && j < Tk) {
                        sP[lr * S1 + j] = pv[u];
                        sPd[lr * S1 + j] = pv[u] * dv[u];
                        sdS[lr * S1 + j] = dv[u];
                    }
                }
            }
        }
        __syncthreads();
        // dP_drop = dO V^T ; dP = dP_drop * drop
        if (mfma) {
            if (!MHA_ABL(256))
            lds_mfma16(sdO, P1, 1, sV, 1, P1, rows, Tk, dk, [&](int lr, int j) { return sdS[lr * S1 + j]; },
                       [&](int lr, int j, float acc, float dm) { sdS[lr * S1 + j] = dm * acc; });
        } else {
            for (int lr = wid * 4 + sub; lr < rows; lr += nw * 4) {
                const float *da = sdO + lr * P1;
                for (int j = l16; j < Tk; j += 16) {
                    const float *va = sV + j * P1;
                    float acc = 0.f;
                    if (!MHA_ABL(256))
                    for (int c = 0; c < dk; c += 4) acc += dot4(LDS4(da + c), LDS4(va + c));
                    sdS[lr * S1 + j] *= acc;             // dP (w.r.t. the pre-dropout probabilities)
                }
            }
        }
        __syncthreads();
        // softmax backward per row: dS = P * (dP - sum_j P dP) * scale
        if (Tk <= 64 && !MHA_ABL(512)) {          // 16 lanes per row, as the forward softmax
            for (int lr0 = wid * 4; lr0 < rows; lr0 += nw * 4) {
                const int lr = lr0 + sub, lrc = min(lr, rows - 1);
                float pp[4], dp[4], sacc = 0.f;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int j = l16 + 16 * e;
                    const bool ok = lr < rows && j < Tk;
                    pp[e] = ok ? sP[lrc * S1 + j] : 0.f;
                    dp[e] = ok ? sdS[lrc * S1 + j] : 0.f;
                    sacc += pp[e] * dp[e];
                }
                sacc = row16_sum(sacc);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int j = l16 + 16 * e;
                    if (lr < rows && j < Tk) sdS[lr * S1 + j] = pp[e] * (dp[e] - sacc) * scale;
                }
            }
        } else if (!MHA_ABL(512))
        for (int lr = wid; lr < rows; lr += nw) {
            float s = 0.f;
            for (int j = lane; j < Tk; j += 64) s += sP[lr * S1 + j] * sdS[lr * S1 + j];
            s = wave_sum(s);
            for (int j = lane; j < Tk; j += 64) sdS[lr * S1 + j] = sP[lr * S1 + j] * (sdS[lr * S1 + j] - s) * scale;
        }
        __syncthreads();
        // dQ = dS K ; dK += dS^T Q ; dV += (P*drop)^T dO
        if (mfma) {
            if (!MHA_ABL(1024))
            lds_mfma16(sdS, S1, 1, sK, P1, 1, rows, dk, Tk, [](int, int) { return 0.f; },
                       [&](int lr, int c, float acc, float) { dq[(qrow0 + lr) * dq_stride + hd * dk + c] = acc; });
            if (!MHA_ABL(2048)) {
                lds_mfma16(sdS, 1, S1, sQ, P1, 1, Tk, dk, rows, [&](int j, int c) { return sdK[j * P1 + c]; },
                           [&](int j, int c, float acc, float old) { sdK[j * P1 + c] = old + acc; });
                lds_mfma16(sPd, 1, S1, sdO, P1, 1, Tk, dk, rows, [&](int j, int c) { return sdV[j * P1 + c]; },
                           [&](int j, int c, float acc, float old) { sdV[j * P1 + c] = old + acc; });
            }
            continue;
        }
        for (int i = threadIdx.x; i < rows * d4; i += blockDim.x) {     // 4 head columns per thread
            const int lr = idiv(i, d4, d4sh), c = (i - lr * d4) * 4;
            const float *ds = sdS + lr * S1;
            f32x4 acc = zero4;
            if (!MHA_ABL(1024))
            for (int j = 0; j < Tk; ++j) acc += ds[j] * LDS4(sK + j * P1 + c);
            *reinterpret_cast<f32x4 *>(dq + (qrow0 + lr) * dq_stride + hd * dk + c) = acc;
        }
        if (!MHA_ABL(2048))
        for (int i = threadIdx.x; i < Tk * d4; i += blockDim.x) {
            const int j = idiv(i, d4, d4sh), c = (i - j * d4) * 4;
            f32x4 ak = zero4, av = zero4;
            for (int lr = 0; lr < rows; ++lr) {
                ak += sdS[lr * S1 + j] * LDS4(sQ + lr * P1 + c);
                av += sPd[lr * S1 + j] * LDS4(sdO + lr * P1 + c);
            }
            *reinterpret_cast<f32x4 *>(sdK + j * P1 + c) += ak;
            *reinterpret_cast<f32x4 *>(sdV + j * P1 + c) += av;
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < Tk * d4; i += blockDim.x) {
        const int j = idiv(i, d4, d4sh), c = (i - j * d4) * 4;
        const size_t oi = (size_t)kvr * dkv_ld + (size_t)j * dkv_stride + hd * dk + c;
        f32x4 gk = LDS4(sdK + j * P1 + c), gv = LDS4(sdV + j * P1 + c);
        if (accumulate) {
            gk += *reinterpret_cast<const f32x4 *>(dk_out + oi);
            gv += *reinterpret_cast<const f32x4 *>(dv_out + oi);
        }
        *reinterpret_cast<f32x4 *>(dk_out + oi) = gk;
        *reinterpret_cast<f32x4 *>(dv_out + oi) = gv;
    }
}
#undef LDS4

__global__ void embed_pe_fwd_kernel(const int64_t *__restrict__ tok, int tok_ld, const float *__restrict__ E,
                                    const float *__restrict__ pe, const float *__restrict__ drop, float *__restrict__ x,
                                    int N, int T, int D, int pos0, float scale) {
    const size_t total = (size_t)N * T * D;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % D);
        const size_t rt = i / D;
        const int t = (int)(rt % T), r = (int)(rt / T);
        float v = E[(size_t)tok[(size_t)r * tok_ld + t] * D + c] * scale + pe[(size_t)(pos0 + t) * D + c];
        if (drop) v *= drop[i];
        x[i] = v;
    }
}

__global__ void embed_pe_bwd_kernel(const int64_t *__restrict__ tok, int tok_ld, const float *__restrict__ dx,
                                    const float *__restrict__ drop, float *__restrict__ dE, int N, int T, int D,
                                    float scale) {
    const size_t total = (size_t)N * T * D;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % D);
        const size_t rt = i / D;
        const int t = (int)(rt % T), r = (int)(rt / T);
        float g = dx[i] * scale;
        if (drop) g *= drop[i];
        if (g != 0.f) atomicAdd(&dE[(size_t)tok[(size_t)r * tok_ld + t] * D + c], g);
    }
}

struct EmbedPeGrad {     // gradient reaching x = (E[tok] * sqrt(D) + pe) * drop at (position, 4 columns)
    const float *dx, *drop; int D; float scale;
    __device__ __forceinline__ f32x4 operator()(size_t pos, int c) const {
        const size_t o = pos * D + c;
        f32x4 g = *reinterpret_cast<const f32x4 *>(dx + o) * scale;
        if (drop) g *= *reinterpret_cast<const f32x4 *>(drop + o);
        return g;
    }
};
__global__ __launch_bounds__(capmi::EBD_THREADS) void embed_pe_bwd_det_kernel(const int64_t *__restrict__ tok, int T, int tok_ld,
                                                                             int rows, int D, float *__restrict__ dE,
                                                                             const EmbedPeGrad g) {
    capmi::embed_bwd_det_body(tok, T, tok_ld, rows, D, dE, g);
}

__global__ __launch_bounds__(1024) void log_softmax_rows_kernel(const float *__restrict__ logits, float *__restrict__ out,
                                                                int V1) {
    __shared__ float s_f[32];
    const size_t r = blockIdx.x;
    const float *x = logits + r * V1;
    float m = -INFINITY;
    for (int v = threadIdx.x; v < V1; v += blockDim.x) m = fmaxf(m, x[v]);
    m = block_max(m, s_f);
    float s = 0.f;
    for (int v = threadIdx.x; v < V1; v += blockDim.x) s += __expf(x[v] - m);
    s = block_sum(s, s_f);
    const float lse = m + __logf(s);
    for (int v = threadIdx.x; v < V1; v += blockDim.x) out[r * V1 + v] = x[v] - lse;
}
// the same with the row in registers (V1 <= 1024 NE): one read of the logits, all of a thread's loads in flight at once; element
// assignment and summation order as above (bit-identical).  r4: the three-pass form above ran at 2.4 TB/s on [6 720, 9 488].
template <int NE>
__global__ __launch_bounds__(1024) void log_softmax_rows_reg_kernel(const float *__restrict__ logits, float *__restrict__ out,
                                                                    int V1) {
    __shared__ float s_f[32];
    const size_t r = blockIdx.x;
    const float *x = logits + r * V1;
    float xv[NE];
#pragma unroll
    for (int i = 0; i < NE; ++i) {
        const int v = threadIdx.x + 1024 * i;
        xv[i] = v < V1 ? x[v] : -INFINITY;
    }
    float m = -INFINITY;
#pragma unroll
    for (int i = 0; i < NE; ++i) m = fmaxf(m, xv[i]);
    m = block_max(m, s_f);
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NE; ++i)
        if (threadIdx.x + 1024 * i < V1) s += __expf(xv[i] - m);
    s = block_sum(s, s_f);
    const float lse = m + __logf(s);
#pragma unroll
    for (int i = 0; i < NE; ++i) {
        const int v = threadIdx.x + 1024 * i;
        if (v < V1) out[r * V1 + v] = xv[i] - lse;
    }
}

__global__ void glu_fwd_kernel(const float *__restrict__ pre, const float *__restrict__ mask,
                               const float *__restrict__ residual, float *__restrict__ out, int M, int R) {
    const size_t total = (size_t)M * R;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t r = i / R, c = i % R;
        float v = pre[r * 2 * R + c] * sigmoid_f(pre[r * 2 * R + R + c]);
        if (mask) v *= mask[i];
        if (residual) v += residual[i];
        out[i] = v;
    }
}

// AoA decode step (AoAModel.py:143-158 core: att2ctx Linear -> GLU -> dropouts): finishes the att2ctx GEMM's K-slice slabs
// (+ bias, in the order of splitk_reduce_kernel), applies the GLU and writes every consumer's operand in ONE launch --
//   pre [M,2R] (kept for the backward), out = pre[:, :R] * sigmoid(pre[:, R:]),
//   out_a = out * mask_a (the logit GEMM's input, F.dropout of the output), out_b = out * mask_b (the NEXT step's context input)
// -- the last two also as "A planes" (M <= 64) so that both GEMMs stage them by LDS-DMA.  Replaces a split-K reduce launch, the
// GLU launch and two mask launches per step.  4 columns per thread (R % 4 == 0, 16-byte aligned operands).
__global__ __launch_bounds__(256) void glu_fwd_fused_kernel(const float *__restrict__ slabs, int splits, size_t stride,
                                                            const float *__restrict__ bias, float *__restrict__ pre,
                                                            float *__restrict__ out, const float *__restrict__ mask_a,
                                                            float *__restrict__ out_a, unsigned char *__restrict__ pl_a,
                                                            const float *__restrict__ mask_b, float *__restrict__ out_b,
                                                            unsigned char *__restrict__ pl_b, int M, int R) {
    const int q4 = R >> 2;
    const size_t nq = (size_t)M * q4;
    for (size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x; q < nq; q += (size_t)gridDim.x * blockDim.x) {
        const int r = (int)(q / q4), c = (int)(q % q4) * 4;
        const size_t ia = (size_t)r * 2 * R + c, ig = ia + R;
        f32x4 a = {0.f, 0.f, 0.f, 0.f}, g = {0.f, 0.f, 0.f, 0.f};
        for (int s0 = 0; s0 < splits; s0 += 4) {               // 8 independent 16-byte loads in flight
            f32x4 ta[4], tg[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const size_t o = (size_t)min(s0 + u, splits - 1) * stride;
                ta[u] = *reinterpret_cast<const f32x4 *>(slabs + o + ia);
                tg[u] = *reinterpret_cast<const f32x4 *>(slabs + o + ig);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (s0 + u < splits) { a += ta[u]; g += tg[u]; }
        }
        if (bias) {
            a += *reinterpret_cast<const f32x4 *>(bias + c);
            g += *reinterpret_cast<const f32x4 *>(bias + R + c);
        }
        *reinterpret_cast<f32x4 *>(pre + ia) = a;
        *reinterpret_cast<f32x4 *>(pre + ig) = g;
        f32x4 o;
#pragma unroll
        for (int k = 0; k < 4; ++k) o[k] = a[k] * sigmoid_f(g[k]);
        const size_t io = (size_t)r * R + c;
        *reinterpret_cast<f32x4 *>(out + io) = o;
        if (out_a) {
            const f32x4 v = mask_a ? o * *reinterpret_cast<const f32x4 *>(mask_a + io) : o;
            *reinterpret_cast<f32x4 *>(out_a + io) = v;
            if (pl_a) capmi::pl_store4(pl_a, r, c, v);
        }
        if (out_b) {
            const f32x4 v = mask_b ? o * *reinterpret_cast<const f32x4 *>(mask_b + io) : o;
            *reinterpret_cast<f32x4 *>(out_b + io) = v;
            if (pl_b) capmi::pl_store4(pl_b, r, c, v);
        }
    }
}

__global__ void glu_bwd_kernel(const float *__restrict__ d_out, const float *__restrict__ mask, const float *__restrict__ pre,
                               float *__restrict__ d_pre, int M, int R, const float *__restrict__ add_slabs, int add_splits,
                               size_t add_stride, const float *__restrict__ add_mask) {
    const size_t total = (size_t)M * R;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t r = i / R, c = i % R;
        float g = d_out[i];
        if (mask) g *= mask[i];
        if (add_slabs) {         // + add_mask * sum_s add_slabs[s]: a second gradient source still in K-slice slabs (splitk_reduce's order)
            float v = 0.f;
            for (int s0 = 0; s0 < add_splits; s0 += 8) {
                float tv[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) tv[u] = s0 + u < add_splits ? add_slabs[(size_t)(s0 + u) * add_stride + i] : 0.f;
#pragma unroll
                for (int u = 0; u < 8; ++u) v += tv[u];
            }
            if (add_mask) v *= add_mask[i];
            g = v + g;
        }
        const float a = pre[r * 2 * R + c], sg = sigmoid_f(pre[r * 2 * R + R + c]);
        d_pre[r * 2 * R + c] = g * sg;
        d_pre[r * 2 * R + R + c] = g * a * sg * (1.f - sg);
    }
}

// [lo | hi] halves of a [M, 2R] product that is still K-slice slabs (or a finished matrix: splits = 1), each times its mask
__global__ void split_halves_kernel(const float *__restrict__ slabs, int splits, size_t stride, const float *__restrict__ mask_lo,
                                    const float *__restrict__ mask_hi, float *__restrict__ out_lo, float *__restrict__ out_hi,
                                    int M, int R) {
    const size_t total = (size_t)M * 2 * R;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t r = i / (2 * R), c = i % (2 * R);
        float v = 0.f;
        for (int sp = 0; sp < splits; ++sp) v += slabs[(size_t)sp * stride + i];
        if (c < (size_t)R) {
            const size_t o = r * R + c;
            out_lo[o] = mask_lo ? v * mask_lo[o] : v;
        } else {
            const size_t o = r * R + (c - R);
            out_hi[o] = mask_hi ? v * mask_hi[o] : v;
        }
    }
}

__global__ void meanpool_fwd_kernel(const float *__restrict__ x, const float *__restrict__ mask, float *__restrict__ mean,
                                    int K, int D) {
    const int b = blockIdx.x;
    float cnt = 0.f;
    for (int k = 0; k < K; ++k) cnt += mask ? mask[(size_t)b * K + k] : 1.f;
    for (int c = threadIdx.x; c < D; c += blockDim.x) {
        float s = 0.f;
        for (int k = 0; k < K; ++k) s += (mask ? mask[(size_t)b * K + k] : 1.f) * x[((size_t)b * K + k) * D + c];
        mean[(size_t)b * D + c] = s / cnt;
    }
}

__global__ void meanpool_bwd_kernel(const float *__restrict__ dmean, const float *__restrict__ mask, float *__restrict__ dx,
                                    int accumulate, int B, int K, int D) {
    const size_t total = (size_t)B * K * D;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % D);
        const size_t bk = i / D;
        const int b = (int)(bk / K);
        float cnt = 0.f;
        for (int k = 0; k < K; ++k) cnt += mask ? mask[(size_t)b * K + k] : 1.f;
        const float v = (mask ? mask[bk] : 1.f) / cnt * dmean[(size_t)b * D + c];
        dx[i] = accumulate ? dx[i] + v : v;
    }
}

inline int grid_for(size_t work) {
    size_t b = (work + 255) / 256;
    if (b > 4096) b = 4096;
    return (int)(b < 1 ? 1 : b);
}

}  // namespace

// ---- caption statistics of the evaluation loop (reference captioning/utils/eval_utils.py:173-174) --------------------------------
//   entropy[r]    = -sum_t sum_v softmax(lp[r,t,:])_v * lp[r,t,v] / (#tokens(r) + 1)
//   perplexity[r] = -sum_t lp[r,t,seq[r,t]] / (#tokens(r) + 1),          #tokens = count(seq[r,:] > 0)
// from the dense [N, L, V1] log-probs a decode returns: one workgroup per (row, step), the step's row read ONCE into registers
// (max, sum of exponentials and sum of e * lp in one pass over them), the per-step terms left in [N, L] scratch and folded per row by
// the second kernel -- instead of torch.softmax + mul + nan_to_num + two sums + gather over three dense temporaries.
// A log-prob of -inf (a token the decoding constraints removed) contributes 0 * -inf := 0, as the ATen path's nan_to_num did.
template <int NE>
__global__ __launch_bounds__(1024) void caption_step_stats_kernel(const float *__restrict__ lp, const int64_t *__restrict__ seq, int V1,
                                                                  float *__restrict__ ent_t, float *__restrict__ sel_t) {
    __shared__ float s_f[32];
    const size_t rt = blockIdx.x;
    const float *x = lp + rt * V1;
    float xv[NE];
#pragma unroll
    for (int i = 0; i < NE; ++i) {
        const int v = threadIdx.x + 1024 * i;
        xv[i] = v < V1 ? x[v] : -INFINITY;
    }
    float m = -INFINITY;
#pragma unroll
    for (int i = 0; i < NE; ++i) m = fmaxf(m, xv[i]);
    m = block_max(m, s_f);
    float z = 0.f, a = 0.f;
#pragma unroll
    for (int i = 0; i < NE; ++i) {
        if (xv[i] > -INFINITY) {
            const float e = __expf(xv[i] - m);
            z += e;
            a += e * xv[i];
        }
    }
    z = block_sum(z, s_f);
    a = block_sum(a, s_f);
    if (threadIdx.x == 0) {
        ent_t[rt] = z > 0.f ? -a / z : 0.f;
        const int64_t tok = seq[rt];
        sel_t[rt] = (tok >= 0 && tok < V1) ? x[tok] : 0.f;
    }
}
__global__ void caption_row_stats_kernel(const float *__restrict__ ent_t, const float *__restrict__ sel_t, const int64_t *__restrict__ seq,
                                         int N, int L, float *__restrict__ entropy, float *__restrict__ perplexity) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= N) return;
    float e = 0.f, p = 0.f, steps = 1.f;
    for (int t = 0; t < L; ++t) {
        e += ent_t[(size_t)r * L + t];
        p += sel_t[(size_t)r * L + t];
        steps += seq[(size_t)r * L + t] > 0 ? 1.f : 0.f;
    }
    entropy[r] = e / steps;
    perplexity[r] = -p / steps;
}

extern "C" {

int capmi_layernorm_fwd(const float *x, const float *a, const float *b, float *y, float *mean, float *inv, int M, int D,
                        float eps, void *stream) {
    if (!x || !a || !b || !y || !mean || !inv || M <= 0 || D < 2) return CAPMI_EINVAL;
    int blocks = (M + 3) / 4;
    if (blocks > 4096) blocks = 4096;
#define CAPMI_LNF(NE_) hipLaunchKernelGGL(layernorm_fwd_reg_kernel<NE_>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, a, b, y, mean, inv, M, D, eps)
    if (D <= 256) CAPMI_LNF(4);
    else if (D <= 512) CAPMI_LNF(8);
    else if (D <= 1024) CAPMI_LNF(16);
    else if (D <= 2048) CAPMI_LNF(32);
    else hipLaunchKernelGGL(layernorm_fwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, a, b, y, mean, inv, M, D, eps);
#undef CAPMI_LNF
    CAPMI_CHECK_LAUNCH();
    return 0;
}

int capmi_layernorm_bwd(const float *dy, const float *x, const float *a, const float *mean, const float *inv, float *dx,
                        int accumulate, float *g_scaled, int M, int D, float eps, void *stream) {
    if (!dy || !x || !a || !mean || !inv || !dx || M <= 0 || D < 2) return CAPMI_EINVAL;
    int blocks = (M + 3) / 4;
    if (blocks > 4096) blocks = 4096;
#define CAPMI_LNB(NE_) hipLaunchKernelGGL(layernorm_bwd_reg_kernel<NE_>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, dy, x, a, mean, inv, dx, accumulate, g_scaled, M, D, eps)
    if (D <= 256) CAPMI_LNB(4);
    else if (D <= 512) CAPMI_LNB(8);
    else if (D <= 1024) CAPMI_LNB(16);
    else if (D <= 2048) CAPMI_LNB(32);
    else hipLaunchKernelGGL(layernorm_bwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, dy, x, a, mean, inv, dx,
                            accumulate, g_scaled, M, D, eps);
#undef CAPMI_LNB
    CAPMI_CHECK_LAUNCH();
    return 0;
}

int capmi_layernorm_bwd_parts_rows(int M) {
    // one partial row per workgroup of four waves; a workgroup takes ~8 rows (a wave per 16 rows, the first cut, left 180 workgroups
    // for 11 520 rows -- fewer than CUs, and the launch got slower than the two column-sum passes it saves)
    int blocks = (M + 7) / 8;
    if (blocks < 1) blocks = 1;
    if (blocks > 4096) blocks = 4096;
    return blocks;
}

int capmi_layernorm_bwd_parts(const float *dy, const float *x, const float *a, const float *mean, const float *inv, float *dx,
                              int accumulate, float *part_a, float *part_b, int M, int D, float eps, void *stream) {
    if (!dy || !x || !a || !mean || !inv || !dx || !part_a || !part_b || M <= 0 || D < 2 || D > 2048) return CAPMI_EINVAL;
    const int blocks = capmi_layernorm_bwd_parts_rows(M);
#define CAPMI_LNP(NE_) hipLaunchKernelGGL(layernorm_bwd_reg_kernel<NE_>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, dy, x, a, mean, inv, dx, accumulate, (float *)nullptr, M, D, eps, part_a, part_b)
    if (D <= 256) CAPMI_LNP(4);
    else if (D <= 512) CAPMI_LNP(8);
    else if (D <= 1024) CAPMI_LNP(16);
    else CAPMI_LNP(32);
#undef CAPMI_LNP
    CAPMI_CHECK_LAUNCH();
    return 0;
}

int capmi_layernorm_bwd_slabs(const float *dy_slabs, int dy_splits, int64_t dy_stride, float *dy_out, const float *x, const float *a,
                              const float *mean, const float *inv, const float *acc_slabs, int acc_splits, int64_t acc_stride,
                              int acc_ld, float *dx, float *g_scaled, int M, int D, float eps, void *stream) {
    if (!dy_slabs || !x || !a || !mean || !inv || !acc_slabs || !dx || M <= 0 || D < 2 || D > 2048 || dy_splits < 1 || acc_splits < 1 ||
        acc_ld < D)
        return CAPMI_EINVAL;
    if ((dy_splits > 1 && dy_stride < (int64_t)M * D) || (acc_splits > 1 && acc_stride < (int64_t)(M - 1) * acc_ld + D)) return CAPMI_EINVAL;
    // 16-byte pieces of both inputs
    if (D % 4 || acc_ld % 4 || dy_stride % 4 || acc_stride % 4 ||
        ((reinterpret_cast<uintptr_t>(dy_slabs) | reinterpret_cast<uintptr_t>(acc_slabs) | reinterpret_cast<uintptr_t>(dy_out)) & 15))
        return CAPMI_EINVAL;
    const int blocks = M;
#define CAPMI_LNS(NE_) hipLaunchKernelGGL(layernorm_bwd_slabs_kernel<NE_>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, dy_slabs, dy_splits, (size_t)dy_stride, dy_out, x, a, mean, inv, acc_slabs, acc_splits, (size_t)acc_stride, acc_ld, dx, g_scaled, M, D, eps)
    if (D <= 256) CAPMI_LNS(4);
    else if (D <= 512) CAPMI_LNS(8);
    else if (D <= 1024) CAPMI_LNS(16);
    else CAPMI_LNS(32);
#undef CAPMI_LNS
    CAPMI_CHECK_LAUNCH();
    return 0;
}

int capmi_mha_fwd(const float *q, const float *k, const float *v, int ldkv, int kstride, const uint8_t *mask, int mask_tq,
                  int mask_per_q, int causal, int q_pos0, const float *drop, float *o, float *p, int Nq, int q_per_kv,
                  int Tq, int Tk, int h, int dk, void *stream) {
    return capmi_mha_fwd_s(q, 0, k, v, ldkv, kstride, mask, mask_tq, mask_per_q, causal, q_pos0, drop, o, p, Nq, q_per_kv, Tq, Tk, h, dk,
                           stream);
}

int capmi_mha_fwd_s(const float *q, int qstride, const float *k, const float *v, int ldkv, int kstride, const uint8_t *mask,
                    int mask_tq, int mask_per_q, int causal, int q_pos0, const float *drop, float *o, float *p, int Nq,
                    int q_per_kv, int Tq, int Tk, int h, int dk, void *stream) {
    return capmi_mha_fwd_qslabs(q, qstride, 0, 0, nullptr, nullptr, k, v, ldkv, kstride, mask, mask_tq, mask_per_q, causal, q_pos0, drop,
                                o, p, Nq, q_per_kv, Tq, Tk, h, dk, stream);
}

int capmi_mha_fwd_qslabs(const float *q, int qstride, int q_splits, int64_t q_slab_stride, const float *q_bias, float *q_out,
                         const float *k, const float *v, int ldkv, int kstride, const uint8_t *mask, int mask_tq, int mask_per_q,
                         int causal, int q_pos0, const float *drop, float *o, float *p, int Nq, int q_per_kv, int Tq, int Tk, int h,
                         int dk, void *stream) {
    if (q_splits < 0 || (q_splits > 1 && (q_slab_stride % 4 || q_slab_stride < (int64_t)Nq * Tq * (qstride > 0 ? qstride : h * dk))))
        return CAPMI_EINVAL;
    if (q_splits > 0 && ((reinterpret_cast<uintptr_t>(q_bias) | reinterpret_cast<uintptr_t>(q_out)) & 15)) return CAPMI_EINVAL;
    if (qstride <= 0) qstride = h * dk;
    if (qstride % 4) return CAPMI_EINVAL;
    if (kstride <= 0) kstride = h * dk;
    if (!q || !k || !v || !o || Nq <= 0 || q_per_kv <= 0 || Nq % q_per_kv || Tq <= 0 || Tk <= 0 || h <= 0 || dk <= 0)
        return CAPMI_EINVAL;
    if (mask && mask_tq != 1 && mask_tq != Tq) return CAPMI_EINVAL;
    // 16-byte accesses along the head dimension
    if (dk % 4 || ldkv % 4 || kstride % 4 || ((reinterpret_cast<uintptr_t>(q) | reinterpret_cast<uintptr_t>(k) |
                                                reinterpret_cast<uintptr_t>(v) | reinterpret_cast<uintptr_t>(o)) & 15))
        return CAPMI_EINVAL;
    const int64_t fixed_f = (int64_t)2 * Tk * (dk + 4), per_f = (int64_t)(dk + 4) + (Tk + 1) + 2;
    const int64_t wgs = (int64_t)(Nq / q_per_kv) * h;
    const int CH = mha_chunk(q_per_kv * Tq, fixed_f, per_f);
    const size_t lds = (size_t)(fixed_f + (int64_t)CH * per_f) * sizeof(float);
    if (lds > 160 * 1024) return CAPMI_EINVAL;
    static bool attr_f = false;      // more than 64 KB of dynamic LDS needs the opt-in (36 regions x 64 dims already do in bwd)
    if (!attr_f) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&mha_fwd_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&mha_fwd_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_f = true;
    }
    mha_sync_ablation();
    if (q_splits > 0)
        hipLaunchKernelGGL(mha_fwd_kernel<true>, dim3(Nq / q_per_kv, h), dim3(mha_threads(CH, Tk, wgs)), lds, (hipStream_t)stream, q, k, v, ldkv,
                           kstride, mask, mask_tq, mask_per_q, causal, q_pos0, drop, o, p, q_per_kv, Tq, Tk, h, dk, CH, qstride,
                           mha_on_mfma(CH), q_splits, (size_t)q_slab_stride, q_bias, q_out);
    else
        hipLaunchKernelGGL(mha_fwd_kernel<false>, dim3(Nq / q_per_kv, h), dim3(mha_threads(CH, Tk, wgs)), lds, (hipStream_t)stream, q, k, v, ldkv,
                           kstride, mask, mask_tq, mask_per_q, causal, q_pos0, drop, o, p, q_per_kv, Tq, Tk, h, dk, CH, qstride,
                           mha_on_mfma(CH), 0, (size_t)0, nullptr, nullptr);
    CAPMI_CHECK_LAUNCH();
    return 0;
}

int capmi_mha_bwd(const float *d_o, const float *q, const float *k, const float *v, int ldkv, int kstride, const float *p,
                  const float *drop, float *dq, float *dk_out, float *dv_out, int dkv_ld, int dkv_stride, int accumulate,
                  int Nq, int q_per_kv, int Tq, int Tk, int h, int dk, void *stream) {
    return capmi_mha_bwd_s(d_o, q, 0, k, v, ldkv, kstride, p, drop, dq, 0, dk_out, dv_out, dkv_ld, dkv_stride, accumulate, Nq, q_per_kv,
                           Tq, Tk, h, dk, stream);
}

int capmi_mha_bwd_s(const float *d_o, const float *q, int qstride, const float *k, const float *v, int ldkv, int kstride,
                    const float *p, const float *drop, float *dq, int dq_stride, float *dk_out, float *dv_out, int dkv_ld,
                    int dkv_stride, int accumulate, int Nq, int q_per_kv, int Tq, int Tk, int h, int dk, void *stream) {
    return capmi_mha_bwd_slabs(d_o, 1, 0, h * dk, q, qstride, k, v, ldkv, kstride, p, drop, dq, dq_stride, dk_out, dv_out, dkv_ld,
                               dkv_stride, accumulate, Nq, q_per_kv, Tq, Tk, h, dk, stream);
}

int capmi_mha_bwd_slabs(const float *d_o, int do_splits, int64_t do_stride, int do_ld, const float *q, int qstride, const float *k,
                        const float *v, int ldkv, int kstride, const float *p, const float *drop, float *dq, int dq_stride,
                        float *dk_out, float *dv_out, int dkv_ld, int dkv_stride, int accumulate, int Nq, int q_per_kv, int Tq,
                        int Tk, int h, int dk, void *stream) {
    if (do_splits < 1 || do_ld < h * dk || do_ld % 4 || (do_splits > 1 && (do_stride % 4 || do_stride < (int64_t)Nq * Tq * do_ld)))
        return CAPMI_EINVAL;
    if (qstride <= 0) qstride = h * dk;
    if (dq_stride <= 0) dq_stride = h * dk;
    if (qstride % 4 || dq_stride % 4) return CAPMI_EINVAL;
    if (kstride <= 0) kstride = h * dk;
    if (dkv_stride <= 0) dkv_stride = h * dk;
    if (dkv_ld <= 0) dkv_ld = Tk * dkv_stride;
    if (!d_o || !q || !k || !v || !p || !dq || !dk_out || !dv_out || Nq <= 0 || q_per_kv <= 0 || Nq % q_per_kv)
        return CAPMI_EINVAL;
    if (dk % 4 || ldkv % 4 || kstride % 4 || dkv_ld % 4 || dkv_stride % 4 ||
        ((reinterpret_cast<uintptr_t>(d_o) | reinterpret_cast<uintptr_t>(q) | reinterpret_cast<uintptr_t>(k) |
          reinterpret_cast<uintptr_t>(v) | reinterpret_cast<uintptr_t>(dq) | reinterpret_cast<uintptr_t>(dk_out) |
          reinterpret_cast<uintptr_t>(dv_out)) & 15))
        return CAPMI_EINVAL;
    const int64_t fixed_b = (int64_t)4 * Tk * (dk + 4), per_b = (int64_t)2 * (dk + 4) + 3 * (Tk + 1) + 1;
    const int64_t wgs = (int64_t)(Nq / q_per_kv) * h;
    const int CH = mha_chunk(q_per_kv * Tq, fixed_b, per_b);
    const size_t lds = (size_t)(fixed_b + (int64_t)CH * per_b) * sizeof(float);
    if (lds > 160 * 1024) return CAPMI_EINVAL;
    static bool attr_b = false;
    if (!attr_b) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&mha_bwd_kernel<MHA_T_BIG>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&mha_bwd_kernel<MHA_T_MAX>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&mha_bwd_kernel<MHA_T_BIG, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&mha_bwd_kernel<MHA_T_MAX, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_b = true;
    }
    const int threads = mha_threads(CH, Tk, wgs);
    mha_sync_ablation();
    if (do_splits > 1 || do_ld != h * dk) {             // d_o in slabs / with a pitch: instances of their own, the plain ones keep their code
        if (threads > MHA_T_BIG)
            hipLaunchKernelGGL((mha_bwd_kernel<MHA_T_MAX, true>), dim3(Nq / q_per_kv, h), dim3(threads), lds, (hipStream_t)stream, d_o, q, k, v,
                               ldkv, kstride, p, drop, dq, dk_out, dv_out, dkv_ld, dkv_stride, accumulate, q_per_kv, Tq, Tk, h, dk, CH,
                               qstride, dq_stride, mha_on_mfma(CH), do_ld, do_splits, (size_t)do_stride);
        else
            hipLaunchKernelGGL((mha_bwd_kernel<MHA_T_BIG, true>), dim3(Nq / q_per_kv, h), dim3(threads), lds, (hipStream_t)stream, d_o, q, k, v,
                               ldkv, kstride, p, drop, dq, dk_out, dv_out, dkv_ld, dkv_stride, accumulate, q_per_kv, Tq, Tk, h, dk, CH,
                               qstride, dq_stride, mha_on_mfma(CH), do_ld, do_splits, (size_t)do_stride);
    } else if (threads > MHA_T_BIG)
        hipLaunchKernelGGL(mha_bwd_kernel<MHA_T_MAX>, dim3(Nq / q_per_kv, h), dim3(threads), lds, (hipStream_t)stream, d_o, q, k, v, ldkv,
                           kstride, p, drop, dq, dk_out, dv_out, dkv_ld, dkv_stride, accumulate, q_per_kv, Tq, Tk, h, dk, CH, qstride,
                           dq_stride, mha_on_mfma(CH), do_ld, do_splits, (size_t)do_stride);
    else
        hipLaunchKernelGGL(mha_bwd_kernel<MHA_T_BIG>, dim3(Nq / q_per_kv, h), dim3(threads), lds, (hipStream_t)stream, d_o, q, k, v, ldkv,
                           kstride, p, drop, dq, dk_out, dv_out, dkv_ld, dkv_stride, accumulate, q_per_kv, Tq, Tk, h, dk, CH, qstride,
                           dq_stride, mha_on_mfma(CH), do_ld, do_splits, (size_t)do_stride);
    CAPMI_CHECK_LAUNCH();
    return 0;
}

int capmi_embed_pe_fwd(const int64_t *tok, int tok_ld, const float *E, const float *pe, const float *drop, float *x, int N,
                       int T, int D, int pos0, void *stream) {
    if (!tok || !E || !pe || !x || N <= 0 || T <= 0 || D <= 0 || pos0 < 0) return CAPMI_EINVAL;
    hipLaunchKernelGGL(embed_pe_fwd_kernel, dim3(grid_for((size_t)N * T * D)), dim3(256), 0, (hipStream_t)stream, tok, tok_ld,
                       E, pe, drop, x, N, T, D, pos0, sqrtf((float)D));
    CAPMI_CHECK_LAUNCH();
    return 0;
}

int capmi_embed_pe_bwd(const int64_t *tok, int tok_ld, const float *dx, const float *drop, float *dE, int N, int T, int D,
                       void *stream) {
    if (!tok || !dx || !dE || N <= 0 || T <= 0 || D <= 0) return CAPMI_EINVAL;
    static const int det = capmi::knob("CAPMI_EMBED_BWD_DET", 1);
    if (det && D % 4 == 0 && (int64_t)N * T <= capmi::EBD_MAX_ROWS &&
        ((reinterpret_cast<uintptr_t>(dx) | reinterpret_cast<uintptr_t>(drop) | reinterpret_cast<uintptr_t>(dE)) & 15) == 0) {
        // r6: ordered sums instead of atomicAdd -- the same bits on every run (embed_bwd_det.h)
        hipLaunchKernelGGL(embed_pe_bwd_det_kernel, dim3(N * T, (D + 255) / 256), dim3(capmi::EBD_THREADS),
                           capmi::embed_bwd_det_lds(N * T), (hipStream_t)stream, tok, T, tok_ld, N * T, D, dE,
                           EmbedPeGrad{dx, drop, D, sqrtf((float)D)});
        CAPMI_CHECK_LAUNCH();
        return 0;
    }
    hipLaunchKernelGGL(embed_pe_bwd_kernel, dim3(grid_for((size_t)N * T * D)), dim3(256), 0, (hipStream_t)stream, tok, tok_ld,
                       dx, drop, dE, N, T, D, sqrtf((float)D));
    CAPMI_CHECK_LAUNCH();
    return 0;
}

int capmi_glu_fwd(const float *pre, const float *mask, const float *residual, float *out, int M, int R, void *stream) {
    if (!pre || !out || M <= 0 || R <= 0) return CAPMI_EINVAL;
    hipLaunchKernelGGL(glu_fwd_kernel, dim3(grid_for((size_t)M * R)), dim3(256), 0, (hipStream_t)stream, pre, mask, residual, out,
                       M, R);
    CAPMI_CHECK_LAUNCH();
    return 0;
}

int capmi_glu_fwd_fused(const float *slabs, int splits, int64_t stride, const float *bias, float *pre, float *out,
                        const float *mask_a, float *out_a, void *planes_a, const float *mask_b, float *out_b, void *planes_b,
                        int M, int R, void *stream) {
    if (!slabs || !pre || !out || splits < 1 || M <= 0 || R <= 0 || R % 4 || stride % 4) return CAPMI_EINVAL;
    if ((planes_a || planes_b) && M > 64) return CAPMI_EINVAL;
    if ((planes_a && !out_a) || (planes_b && !out_b)) return CAPMI_EINVAL;
    if ((reinterpret_cast<uintptr_t>(slabs) | reinterpret_cast<uintptr_t>(bias) | reinterpret_cast<uintptr_t>(pre) |
         reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(mask_a) | reinterpret_cast<uintptr_t>(out_a) |
         reinterpret_cast<uintptr_t>(mask_b) | reinterpret_cast<uintptr_t>(out_b) | reinterpret_cast<uintptr_t>(planes_a) |
         reinterpret_cast<uintptr_t>(planes_b)) & 15)
        return CAPMI_EINVAL;
    hipLaunchKernelGGL(glu_fwd_fused_kernel, dim3(grid_for((size_t)M * (R / 4))), dim3(256), 0, (hipStream_t)stream, slabs, splits,
                       (size_t)stride, bias, pre, out, mask_a, out_a, static_cast<unsigned char *>(planes_a), mask_b, out_b,
                       static_cast<unsigned char *>(planes_b), M, R);
    CAPMI_CHECK_LAUNCH();
    return 0;
}

int capmi_glu_bwd(const float *d_out, const float *mask, const float *pre, float *d_pre, int M, int R, void *stream) {
    return capmi_glu_bwd_add(d_out, mask, nullptr, 0, 0, nullptr, pre, d_pre, M, R, stream);
}

int capmi_glu_bwd_add(const float *d_out, const float *mask, const float *add_slabs, int add_splits, int64_t add_stride,
                      const float *add_mask, const float *pre, float *d_pre, int M, int R, void *stream) {
    if (!d_out || !pre || !d_pre || M <= 0 || R <= 0) return CAPMI_EINVAL;
    if (add_slabs && (add_splits < 1 || (add_splits > 1 && add_stride < (int64_t)M * R))) return CAPMI_EINVAL;
    hipLaunchKernelGGL(glu_bwd_kernel, dim3(grid_for((size_t)M * R)), dim3(256), 0, (hipStream_t)stream, d_out, mask, pre, d_pre,
                       M, R, add_slabs, add_splits, (size_t)add_stride, add_mask);
    CAPMI_CHECK_LAUNCH();
    return 0;
}

int capmi_split_halves(const float *slabs, int splits, int64_t stride, const float *mask_lo, const float *mask_hi, float *out_lo,
                       float *out_hi, int M, int R, void *stream) {
    if (!slabs || !out_lo || !out_hi || splits < 1 || M <= 0 || R <= 0) return CAPMI_EINVAL;
    hipLaunchKernelGGL(split_halves_kernel, dim3(grid_for((size_t)M * 2 * R)), dim3(256), 0, (hipStream_t)stream, slabs, splits,
                       (size_t)stride, mask_lo, mask_hi, out_lo, out_hi, M, R);
    CAPMI_CHECK_LAUNCH();
    return 0;
}

int capmi_meanpool_fwd(const float *x, const float *mask, float *mean, int B, int K, int D, void *stream) {
    if (!x || !mean || B <= 0 || K <= 0 || D <= 0) return CAPMI_EINVAL;
    hipLaunchKernelGGL(meanpool_fwd_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, x, mask, mean, K, D);
    CAPMI_CHECK_LAUNCH();
    return 0;
}

int capmi_meanpool_bwd(const float *dmean, const float *mask, float *dx, int accumulate, int B, int K, int D, void *stream) {
    if (!dmean || !dx || B <= 0 || K <= 0 || D <= 0) return CAPMI_EINVAL;
    hipLaunchKernelGGL(meanpool_bwd_kernel, dim3(grid_for((size_t)B * K * D)), dim3(256), 0, (hipStream_t)stream, dmean, mask, dx,
                       accumulate, B, K, D);
    CAPMI_CHECK_LAUNCH();
    return 0;
}

int capmi_caption_stats(const float *seq_logp, const int64_t *seq, int N, int L, int V1, float *scratch, float *entropy, float *perplexity,
                        void *stream) {
    if (!seq_logp || !seq || !scratch || !entropy || !perplexity || N <= 0 || L <= 0 || V1 <= 0 || V1 > 32 * 1024) return CAPMI_EINVAL;
    float *ent_t = scratch, *sel_t = scratch + (size_t)N * L;
#define CAPMI_CS(NE_) hipLaunchKernelGGL(caption_step_stats_kernel<NE_>, dim3(N * L), dim3(1024), 0, (hipStream_t)stream, seq_logp, seq, V1, ent_t, sel_t)
    if (V1 <= 10 * 1024) CAPMI_CS(10);
    else if (V1 <= 16 * 1024) CAPMI_CS(16);
    else CAPMI_CS(32);
#undef CAPMI_CS
    CAPMI_CHECK_LAUNCH();
    hipLaunchKernelGGL(caption_row_stats_kernel, dim3((N + 255) / 256), dim3(256), 0, (hipStream_t)stream, ent_t, sel_t, seq, N, L, entropy,
                       perplexity);
    CAPMI_CHECK_LAUNCH();
    return 0;
}

int capmi_log_softmax_rows(const float *logits, float *out, int rows, int V1, void *stream) {
    if (!logits || !out || rows <= 0 || V1 <= 0) return CAPMI_EINVAL;
    if (V1 <= 10 * 1024)
        hipLaunchKernelGGL(log_softmax_rows_reg_kernel<10>, dim3(rows), dim3(1024), 0, (hipStream_t)stream, logits, out, V1);
    else if (V1 <= 16 * 1024)
        hipLaunchKernelGGL(log_softmax_rows_reg_kernel<16>, dim3(rows), dim3(1024), 0, (hipStream_t)stream, logits, out, V1);
    else
        hipLaunchKernelGGL(log_softmax_rows_kernel, dim3(rows), dim3(1024), 0, (hipStream_t)stream, logits, out, V1);
    CAPMI_CHECK_LAUNCH();
    return 0;
}

}  // extern "C"
