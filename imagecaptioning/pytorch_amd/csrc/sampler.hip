// log-softmax + next-token choice + rollout bookkeeping in one launch per step (gfx950).
// Replaces F.log_softmax (AttModel.py:172), CaptionModel.sample_next_word (CaptionModel.py:370-407:
// torch.max / Categorical.sample + gather) and the per-step host logic of AttModel._sample
// (AttModel.py:340-350), including its device->host `unfinished.sum()==0` sync: the finished flags
// stay on the device.
//
// One workgroup (1024 threads = 16 wave64s) per caption row: the V1 logits of the row are read once
// (16-byte loads), max / sum-exp / arg-max are wave-shuffle + LDS reductions, and the dense log-prob
// row is written once.  Arg-max ties resolve to the LOWEST index like torch.max on CPU.
#include "capmi_common.h"
#include "profile.h"
#include "../../../include/capmi.h"

using namespace capmi;

namespace {

constexpr int SEL_THREADS = 1024;

struct ArgMax {
    float v;
    int i;
};
__device__ __forceinline__ ArgMax better(ArgMax a, ArgMax b) {
    // larger value wins; equal values -> smaller index (first occurrence)
    if (b.v > a.v || (b.v == a.v && b.i < a.i)) return b;
    return a;
}
__device__ __forceinline__ ArgMax wave_argmax(ArgMax x) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        ArgMax y;
        y.v = __shfl_xor(x.v, o, 64);
        y.i = __shfl_xor(x.i, o, 64);
        x = better(x, y);
    }
    return x;
}

__global__ __launch_bounds__(SEL_THREADS) void logsoftmax_select_kernel(
    const float *__restrict__ logits, int V1, int step, int L, int mode, const uint8_t *__restrict__ row_mode,
    float temperature, const float *__restrict__ gumbel, uint64_t seed, const int64_t *__restrict__ forced,
    int forced_ld, int no_finish_mask, int64_t *__restrict__ seq, int seq_ld, int64_t *__restrict__ it_next,
    uint8_t *__restrict__ unfinished, float *__restrict__ seq_logp, float *__restrict__ sel_logp,
    uint8_t *__restrict__ live) {
    __shared__ float s_f[32];
    __shared__ int s_i[32];
    const int r = blockIdx.x;
    const float *x = logits + (size_t)r * V1;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const int my_mode = row_mode ? (int)row_mode[r] : mode;

    float m = -INFINITY;
    for (int v = threadIdx.x; v < V1; v += blockDim.x) m = fmaxf(m, x[v]);
    m = block_max(m, s_f);
    float s = 0.f;
    for (int v = threadIdx.x; v < V1; v += blockDim.x) s += __expf(x[v] - m);
    s = block_sum(s, s_f);
    const float lse = m + __logf(s);

    // choose
    int token;
    if (my_mode == 2) {
        token = (int)forced[(size_t)r * forced_ld + step];
    } else {
        ArgMax best{-INFINITY, 0x7fffffff};
        if (my_mode == 0) {
            for (int v = threadIdx.x; v < V1; v += blockDim.x) best = better(best, ArgMax{x[v], v});
        } else {
            const float invT = 1.f / temperature;
            if (gumbel) {
                const float *g = gumbel + (size_t)r * V1;
                for (int v = threadIdx.x; v < V1; v += blockDim.x)
                    best = better(best, ArgMax{(x[v] - lse) * invT + g[v], v});
            } else {
                const Philox rng(seed);
                // one Philox call yields 4 uniforms: thread handles quads of vocabulary entries
                for (int q = threadIdx.x; q * 4 < V1; q += blockDim.x) {
                    uint32_t o[4];
                    rng.gen(((uint64_t)step << 32) | (uint32_t)r, (uint64_t)q, o);
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const int v = q * 4 + k;
                        if (v < V1) {
                            const float gn = -__logf(-__logf(u01(o[k])));
                            best = better(best, ArgMax{(x[v] - lse) * invT + gn, v});
                        }
                    }
                }
            }
        }
        best = wave_argmax(best);
        __syncthreads();
        if (lane == 0) {
            s_f[wid] = best.v;
            s_i[wid] = best.i;
        }
        __syncthreads();
        ArgMax t{s_f[0], s_i[0]};
        for (int i = 1; i < nw; ++i) t = better(t, ArgMax{s_f[i], s_i[i]});
        token = t.i;
    }

    // bookkeeping (AttModel.py:340-347)
    const bool was_unf = (step == 0 || no_finish_mask) ? true : (unfinished[r] != 0);
    if (!was_unf) token = 0;
    const float keep = was_unf ? 1.f : 0.f;
    float *out = seq_logp ? seq_logp + ((size_t)r * L + step) * V1 : nullptr;
    if (out) {
        if (was_unf)
            for (int v = threadIdx.x; v < V1; v += blockDim.x) out[v] = x[v] - lse;
        else
            for (int v = threadIdx.x; v < V1; v += blockDim.x) out[v] = 0.f;
    }
    __syncthreads();   // all reads of unfinished[r] done before thread 0 rewrites it
    if (threadIdx.x == 0) {
        seq[(size_t)r * seq_ld + step] = token;
        it_next[r] = token;
        if (sel_logp) sel_logp[(size_t)r * L + step] = keep * (x[token] - lse);
        if (live) live[(size_t)r * L + step] = was_unf ? 1 : 0;
        if (!no_finish_mask) unfinished[r] = (was_unf && token != 0) ? 1 : 0;
    }
}

__global__ __launch_bounds__(SEL_THREADS) void logsoftmax_bwd_kernel(const float *__restrict__ g,
                                                                      const float *__restrict__ seq_logp,
                                                                      const uint8_t *__restrict__ live,
                                                                      float *__restrict__ dlogits, int N, int L,
                                                                      int V1) {
    __shared__ float s_f[32];
    const int t = blockIdx.x / N, n = blockIdx.x % N;     // output row (time-major) = blockIdx.x
    const size_t r = (size_t)n * L + t;                    // input row (caption-major)
    const float *gr = g + r * V1;
    const float *lp = seq_logp + r * V1;
    float *o = dlogits + (size_t)blockIdx.x * V1;
    if (live && !live[r]) {
        for (int v = threadIdx.x; v < V1; v += blockDim.x) o[v] = 0.f;
        return;
    }
    float s = 0.f;
    for (int v = threadIdx.x; v < V1; v += blockDim.x) s += gr[v];
    s = block_sum(s, s_f);
    for (int v = threadIdx.x; v < V1; v += blockDim.x) o[v] = gr[v] - __expf(lp[v]) * s;
}

}  // namespace

extern "C" {

int capmi_logsoftmax_select(const float *logits, int N, int V1, int step, int L, int mode, const uint8_t *row_mode,
                            float temperature, const float *gumbel, uint64_t seed, const int64_t *forced,
                            int forced_ld, int no_finish_mask, int64_t *seq, int seq_ld, int64_t *it_next,
                            uint8_t *unfinished, float *seq_logp, float *sel_logp, uint8_t *live, void *stream) {
    if (!logits || N <= 0 || V1 <= 0 || step < 0 || step >= L || !seq || !it_next) return CAPMI_EINVAL;
    if (!no_finish_mask && !unfinished) return CAPMI_EINVAL;
    if ((mode == 2 || row_mode) && !forced && mode == 2) return CAPMI_EINVAL;
    if (mode == 1 && !(temperature > 0.f)) return CAPMI_EINVAL;
    hipLaunchKernelGGL(logsoftmax_select_kernel, dim3(N), dim3(SEL_THREADS), 0, (hipStream_t)stream, logits, V1, step, L,
                       mode, row_mode, temperature, gumbel, seed, forced, forced_ld, no_finish_mask, seq, seq_ld,
                       it_next, unfinished, seq_logp, sel_logp, live);
    CAPMI_CHECK_LAUNCH();
    return 0;
}

int capmi_logsoftmax_bwd(const float *g, const float *seq_logp, const uint8_t *live, float *dlogits, int N, int L,
                         int T, int V1, void *stream) {
    if (!g || !seq_logp || !dlogits || N <= 0 || T <= 0 || T > L || V1 <= 0) return CAPMI_EINVAL;
    hipLaunchKernelGGL(logsoftmax_bwd_kernel, dim3(N * T), dim3(SEL_THREADS), 0, (hipStream_t)stream, g, seq_logp, live,
                       dlogits, N, L, V1);
    CAPMI_CHECK_LAUNCH();
    return 0;
}

}  // extern "C"
