// Decode-time options of the reference's samplers, applied on the device to the [rows, V1] log-probability rows of
// one step (gfx950).  They replace the host-side tensor edits (and their `.item()` / `.cpu()` syncs) of
//   AttModel._sample            AttModel.py:293-330   decoding_constraint, remove_bad_endings, block_trigrams
//   AttModel._diverse_sample    AttModel.py:391-432   + the column penalty of earlier groups
//   CaptionModel.beam_search    CaptionModel.py:38-57, 152-157   add_diversity and the same two constraints
// All of this is a handful of scattered writes per row: one thread per row does them in the reference's order so that
// overlapping edits (e.g. both constraints hitting column 0) resolve exactly as they do there.
#include "capmi_common.h"
#include "../../../include/capmi.h"

using namespace capmi;

namespace {

__global__ void decode_constrain_kernel(float *__restrict__ logp, int N, int V1, const int64_t *__restrict__ prev,
                                        int prev_stride, int flags, const int64_t *__restrict__ bad, int n_bad,
                                        const int64_t *__restrict__ seq, int seq_ld, int t, int tri_rows) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= N) return;
    float *x = logp + (size_t)r * V1;
    if (flags & (CAPMI_DECODE_NO_REPEAT | CAPMI_DECODE_NO_BAD_ENDING)) {
        const int64_t p = prev[(size_t)r * prev_stride];
        // logprobs + tmp with tmp = -inf at one entry (AttModel.py:293-296)
        if ((flags & CAPMI_DECODE_NO_REPEAT) && p >= 0 && p < V1) x[p] += -INFINITY;
        if (flags & CAPMI_DECODE_NO_BAD_ENDING) {      // :298-303: the end token may not follow a bad ending
            bool is_bad = false;
            for (int i = 0; i < n_bad; ++i) is_bad |= (bad[i] == p);
            if (is_bad) x[0] += -INFINITY;
        }
    }
    if ((flags & CAPMI_DECODE_BLOCK_TRIGRAMS) && t >= 3 && r < tri_rows) {
        // :307-330: every earlier trigram (s[e-2], s[e-1], s[e]) whose first two tokens equal the last two tokens
        // (s[t-2], s[t-1]) adds 1 to mask[s[e]]; logprobs += (mask * -0.693) * 2.0
        const int64_t *s = seq + (size_t)r * seq_ld;
        const int64_t a = s[t - 2], b = s[t - 1];
        for (int e = 2; e < t; ++e) {
            if (s[e - 2] != a || s[e - 1] != b) continue;
            const int64_t tok = s[e];
            bool first = true;
            int count = 0;
            for (int f = 2; f < t; ++f) {
                if (s[f - 2] == a && s[f - 1] == b && s[f] == tok) {
                    if (f < e) first = false;
                    ++count;
                }
            }
            if (first && tok >= 0 && tok < V1) x[tok] += ((float)count * -0.693f) * 2.0f;
        }
    }
}

// out[b*cur + r, :] = logp[b*cur + r, :] - change[b, :] * lambda, change[b, v] = how many of the n_prev tokens the earlier
// groups chose for image b at this local time equal v (CaptionModel.py:38-57)
__global__ __launch_bounds__(256) void beam_diversity_kernel(const float *__restrict__ logp, float *__restrict__ out, int cur,
                                                              int V1, const int64_t *__restrict__ prev_tok, int prev_stride,
                                                              int n_prev, float lambda) {
    const int b = blockIdx.x;
    const size_t base = (size_t)b * cur * V1;
    const size_t cnt = (size_t)cur * V1;
    for (size_t i = threadIdx.x; i < cnt; i += blockDim.x) out[base + i] = logp[base + i];
    __syncthreads();
    const int64_t *pt = prev_tok + (size_t)b * prev_stride;
    for (int p = threadIdx.x; p < n_prev; p += blockDim.x) {
        const int64_t tok = pt[p];
        bool first = true;
        int count = 0;
        for (int q = 0; q < n_prev; ++q) {
            if (pt[q] == tok) {
                if (q < p) first = false;
                ++count;
            }
        }
        if (!first || tok < 0 || tok >= V1) continue;
        const float pen = (float)count * lambda;
        for (int r = 0; r < cur; ++r) {
            const size_t o = base + (size_t)r * V1 + tok;
            out[o] = logp[o] - pen;
        }
    }
}

// logprobs[:, tokens] = logprobs[:, tokens] - lambda (AttModel.py:395-397): EVERY row loses lambda once in each column
// that any row of the earlier group chose (advanced-index assignment: duplicates write the same value)
__global__ __launch_bounds__(256) void column_penalty_kernel(float *__restrict__ logp, int V1, const int64_t *__restrict__ tok,
                                                              int n_tok, int tok_stride, float lambda) {
    float *x = logp + (size_t)blockIdx.x * V1;
    for (int p = threadIdx.x; p < n_tok; p += blockDim.x) {
        const int64_t v = tok[(size_t)p * tok_stride];
        bool first = true;
        for (int q = 0; q < p; ++q) first &= (tok[(size_t)q * tok_stride] != v);
        if (first && v >= 0 && v < V1) x[v] = x[v] - lambda;
    }
}

}  // namespace

extern "C" {

int capmi_decode_constrain(float *logp, int N, int V1, const int64_t *prev, int prev_stride, int flags,
                           const int64_t *bad_endings, int n_bad, const int64_t *seq, int seq_ld, int t, int trigram_rows,
                           void *stream) {
    if (!logp || N <= 0 || V1 <= 0 || t < 0) return CAPMI_EINVAL;
    if ((flags & (CAPMI_DECODE_NO_REPEAT | CAPMI_DECODE_NO_BAD_ENDING)) && (!prev || prev_stride <= 0)) return CAPMI_EINVAL;
    if ((flags & CAPMI_DECODE_NO_BAD_ENDING) && n_bad > 0 && !bad_endings) return CAPMI_EINVAL;
    if ((flags & CAPMI_DECODE_BLOCK_TRIGRAMS) && (!seq || seq_ld < t)) return CAPMI_EINVAL;
    if (!flags) return 0;
    hipLaunchKernelGGL(decode_constrain_kernel, dim3((N + 63) / 64), dim3(64), 0, (hipStream_t)stream, logp, N, V1, prev,
                       prev_stride, flags, bad_endings, n_bad < 0 ? 0 : n_bad, seq, seq_ld, t, trigram_rows);
    CAPMI_CHECK_LAUNCH();
    return 0;
}

int capmi_beam_diversity(const float *logp, float *out, int B, int cur, int V1, const int64_t *prev_tokens, int prev_stride,
                         int n_prev, float diversity_lambda, void *stream) {
    if (!logp || !out || B <= 0 || cur <= 0 || V1 <= 0 || n_prev < 0 || (n_prev > 0 && (!prev_tokens || prev_stride < n_prev)))
        return CAPMI_EINVAL;
    hipLaunchKernelGGL(beam_diversity_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, logp, out, cur, V1, prev_tokens,
                       prev_stride, n_prev, diversity_lambda);
    CAPMI_CHECK_LAUNCH();
    return 0;
}

int capmi_column_penalty(float *logp, int N, int V1, const int64_t *tokens, int n_tokens, int token_stride,
                         float diversity_lambda, void *stream) {
    if (!logp || N <= 0 || V1 <= 0 || n_tokens < 0 || (n_tokens > 0 && (!tokens || token_stride <= 0))) return CAPMI_EINVAL;
    if (n_tokens == 0) return 0;
    hipLaunchKernelGGL(column_penalty_kernel, dim3(N), dim3(256), 0, (hipStream_t)stream, logp, V1, tokens, n_tokens,
                       token_stride, diversity_lambda);
    CAPMI_CHECK_LAUNCH();
    return 0;
}

}  // extern "C"
