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
#include "gemm_lc_body.h"      // (r4: the decode GEMM's body, for the fused select + GEMM launch)

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
template <int CTRL>
__device__ __forceinline__ ArgMax dpp_better(ArgMax x) { return better(x, ArgMax{dpp_f<CTRL>(x.v), dpp_i<CTRL>(x.i)}); }
__device__ __forceinline__ ArgMax lane_am(ArgMax x, int l) { return ArgMax{lane_f(x.v, l), __builtin_amdgcn_readlane(x.i, l)}; }
// better() is associative, commutative and breaks ties by index, so any reduction tree gives the same winner (capmi_common.h: DPP path)
__device__ __forceinline__ ArgMax wave_argmax(ArgMax x) {
    x = dpp_better<DPP_XOR1>(x);
    x = dpp_better<DPP_XOR2>(x);
    x = dpp_better<DPP_HALF_MIRROR>(x);
    x = dpp_better<DPP_ROW_MIRROR>(x);
    return better(better(lane_am(x, 0), lane_am(x, 16)), better(lane_am(x, 32), lane_am(x, 48)));
}

// order-preserving map float -> uint32 (a < b  <=>  key(a) < key(b); -inf -> smallest)
__device__ __forceinline__ uint32_t order_key(float f) {
    const uint32_t b = __builtin_bit_cast(uint32_t, f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}

// next step's token embedding written by the workgroup that just chose the token (saves the embed launch of every
// step but the first): x_next[r,:] = relu?(E[token,:]) * mask[r,:], it_save[r] = token
struct NextEmbed {
    const float *E, *mask;
    float *x;
    int64_t *it_save;
    int Edim, relu;
    unsigned char *pl;      // A planes of x (rows <= 64) for the next step's gate GEMM, or null
    int *alive;             // set to 1 by every row that is still unfinished after this step (early exit of rollouts), or null
};
__device__ __forceinline__ void emit_next_embed(const NextEmbed &ne, int r, int token) {
    if (!ne.x) return;
    const float *e = ne.E + (size_t)token * ne.Edim;
    for (int c = threadIdx.x; c < ne.Edim; c += blockDim.x) {
        float v = e[c];
        if (ne.relu) v = fmaxf(v, 0.f);
        if (ne.mask) v *= ne.mask[(size_t)r * ne.Edim + c];
        ne.x[(size_t)r * ne.Edim + c] = v;
        if (ne.pl) capmi::pl_store1(ne.pl, r, c, v);
    }
    if (threadIdx.x == 0 && ne.it_save) ne.it_save[r] = token;
}

__global__ __launch_bounds__(SEL_THREADS) void logsoftmax_select_kernel(
    const float *__restrict__ logits, int splits, size_t slab_stride, const float *__restrict__ bias, int V1, int step,
    int L, int mode, const uint8_t *__restrict__ row_mode,
    float temperature, const float *__restrict__ gumbel, uint64_t seed, const int64_t *__restrict__ forced,
    int forced_ld, int no_finish_mask, int64_t *__restrict__ seq, int seq_ld, int64_t *__restrict__ it_next,
    uint8_t *__restrict__ unfinished, float *__restrict__ seq_logp, float *__restrict__ sel_logp,
    uint8_t *__restrict__ live, const NextEmbed ne, int top_k, float top_p, int prenorm, const uint64_t *__restrict__ epoch) {
    __shared__ float s_f[32];
    __shared__ int s_i[32];
    const int r = blockIdx.x;
    // any vocabulary size / alignment: the row is re-assembled (slabs + bias) on every pass
    struct Row {
        const float *p, *bias;
        int splits;
        size_t stride;
        __device__ __forceinline__ float operator[](int v) const {
            float a = p[v];
            for (int s = 1; s < splits; ++s) a += p[s * stride + v];
            return bias ? a + bias[v] : a;
        }
    };
    const Row x{logits + (size_t)r * V1, bias, splits, slab_stride};
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const int my_mode = row_mode ? (int)row_mode[r] : mode;

    float m = -INFINITY;
    for (int v = threadIdx.x; v < V1; v += blockDim.x) m = fmaxf(m, x[v]);
    m = block_max(m, s_f);
    float s = 0.f;
    for (int v = threadIdx.x; v < V1; v += blockDim.x) s += __expf(x[v] - m);
    s = block_sum(s, s_f);
    // prenorm: the row already holds (possibly constrained: -inf / penalised entries) log-probabilities that must be
    // stored and gathered as they are (AttModel.py:293-330 edits them AFTER log_softmax); Categorical / arg-max only
    // need them up to a constant, so nothing is renormalised
    const float lse = prenorm ? 0.f : m + __logf(s);

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
            // top-k / nucleus threshold: same radix descent as the register-resident kernel (see there)
            uint32_t thr_key = 0;
            if (top_k > 0 || top_p > 0.f) {
                float lse_t = 0.f;
                if (top_p > 0.f) {
                    float mt = -INFINITY, st = 0.f;
                    for (int v = threadIdx.x; v < V1; v += blockDim.x) mt = fmaxf(mt, (x[v] - lse) * invT);
                    mt = block_max(mt, s_f);
                    for (int v = threadIdx.x; v < V1; v += blockDim.x) st += __expf((x[v] - lse) * invT - mt);
                    st = block_sum(st, s_f);
                    lse_t = mt + __logf(st);
                }
                uint32_t cur = 0;
                for (int bit = 31; bit >= 0; --bit) {
                    const uint32_t cand = cur | (1u << bit);
                    float acc = 0.f;
                    for (int v = threadIdx.x; v < V1; v += blockDim.x) {
                        const float xt_ = (x[v] - lse) * invT;
                        const uint32_t key = order_key(xt_);
                        if (top_k > 0) acc += key >= cand ? 1.f : 0.f;
                        else acc += key > cand ? __expf(xt_ - lse_t) : 0.f;
                    }
                    acc = block_sum(acc, s_f);
                    if (top_k > 0 ? acc >= (float)top_k : acc >= top_p) cur = cand;
                }
                thr_key = top_k > 0 ? cur : cur + 1;
            }
            if (gumbel) {
                const float *g = gumbel + (size_t)r * V1;
                for (int v = threadIdx.x; v < V1; v += blockDim.x) {
                    const float xt_ = (x[v] - lse) * invT;
                    if (order_key(xt_) >= thr_key) best = better(best, ArgMax{xt_ + g[v], v});
                }
            } else {
                const Philox rng(capmi::epoch_seed(seed, epoch));
                // one Philox call yields 4 uniforms: thread handles quads of vocabulary entries
                for (int q = threadIdx.x; q * 4 < V1; q += blockDim.x) {
                    uint32_t o[4];
                    rng.gen(((uint64_t)step << 32) | (uint32_t)r, (uint64_t)q, o);
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const int v = q * 4 + k;
                        if (v < V1) {
                            const float gn = -__logf(-__logf(u01(o[k])));
                            const float xt_ = (x[v] - lse) * invT;
                            if (order_key(xt_) >= thr_key) best = better(best, ArgMax{xt_ + gn, v});
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
    const int chosen = token;
    if (!was_unf) token = 0;
    const float keep = was_unf ? 1.f : 0.f;
    float *out = seq_logp ? seq_logp + ((size_t)r * L + step) * V1 : nullptr;
    if (out) {
        if (was_unf)
            for (int v = threadIdx.x; v < V1; v += blockDim.x) out[v] = x[v] - lse;
        else if (prenorm)   // logprobs * unfinished (AttModel.py:345): -inf * 0 = NaN is kept as the reference produces it
            for (int v = threadIdx.x; v < V1; v += blockDim.x) out[v] = x[v] * keep;
        else
            for (int v = threadIdx.x; v < V1; v += blockDim.x) out[v] = 0.f;
    }
    emit_next_embed(ne, r, token);
    __syncthreads();   // all reads of unfinished[r] done before thread 0 rewrites it
    if (threadIdx.x == 0) {
        seq[(size_t)r * seq_ld + step] = token;
        it_next[r] = token;
        // prenorm 2 (AttModel._diverse_sample, AttModel.py:436-447): the log-prob of the token the sampler picked, also
        // for rows that had already finished (their token is overwritten by the pad, the stored value is not)
        if (sel_logp) sel_logp[(size_t)r * L + step] = prenorm == 2 ? x[chosen] : keep * (x[token] - lse);
        if (live) live[(size_t)r * L + step] = was_unf ? 1 : 0;
        if (!no_finish_mask) unfinished[r] = (was_unf && token != 0) ? 1 : 0;
        // early exit (AttModel.py:349-350): a row that goes on tells the host so (a word of pinned host memory, only ever set)
        if (ne.alive && (no_finish_mask || (was_unf && token != 0))) *ne.alive = 1;
    }
}

// Register-resident variant: the row (<= NQ*4096 logits, V1 % 4 == 0) is assembled ONCE into registers as
// sum of the logit GEMM's K-slice slabs + bias (so the split-K reduce launch and the logits round trip disappear),
// then max / sum-exp / choice / dense log-prob write all run from registers.  Same thread->vocabulary-quad map as
// the streaming kernel above, hence identical Philox counters and identical samples.
// The arguments of one select launch (one struct so that the fused select + GEMM launch can carry them next to the GEMM's).
struct SelArgs {
    const float *src; int splits; size_t slab_stride; const float *bias; int V1, step, L, mode; const uint8_t *row_mode;
    float temperature; const float *gumbel; uint64_t seed; const int64_t *forced; int forced_ld, no_finish_mask; int64_t *seq;
    int seq_ld; int64_t *it_next; uint8_t *unfinished; float *seq_logp, *sel_logp; uint8_t *live; NextEmbed ne; int top_k;
    float top_p; int abl; int raw_out;     // raw_out: store the logits themselves (AttModel.py:172-175 output_logsoftmax = 0)
    const uint64_t *epoch;                 // capmi_rng_bind_epoch (NULL: the seed alone)
};

// Body of the register-resident select for caption row r; s_f [32] / s_i [32] / s_tok [1] are workgroup scratch in LDS (static in the
// stand-alone kernel, carved out of the dynamic region in the fused launch, where static LDS would misalign the GEMM's ring).
template <int NQ>
__device__ __forceinline__ void select_reg_body(const SelArgs &A, const int r, float *s_f, int *s_i, float *s_tok_p) {
    const float *__restrict__ src = A.src; const int splits = A.splits; const size_t slab_stride = A.slab_stride;
    const float *__restrict__ bias = A.bias; const int V1 = A.V1, step = A.step, L = A.L, mode = A.mode;
    const uint8_t *__restrict__ row_mode = A.row_mode; const float temperature = A.temperature;
    const float *__restrict__ gumbel = A.gumbel; const uint64_t seed = A.seed; const int64_t *__restrict__ forced = A.forced;
    const int forced_ld = A.forced_ld, no_finish_mask = A.no_finish_mask; int64_t *__restrict__ seq = A.seq; const int seq_ld = A.seq_ld;
    int64_t *__restrict__ it_next = A.it_next; uint8_t *__restrict__ unfinished = A.unfinished; float *__restrict__ seq_logp = A.seq_logp;
    float *__restrict__ sel_logp = A.sel_logp; uint8_t *__restrict__ live = A.live; const NextEmbed &ne = A.ne; const int top_k = A.top_k;
    const float top_p = A.top_p; const int abl = A.abl; const int raw_out = A.raw_out;
#define s_tok (*s_tok_p)
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = SEL_THREADS >> 6;
    const int my_mode = row_mode ? (int)row_mode[r] : mode;
    const int nq = V1 >> 2;

    f32x4 x[NQ];
    {
        // every load of the row assembly -- the first four K-slice slabs and the bias -- is issued before the first add,
        // branch-free (quad index clamped, surplus slabs re-read slab 0 and are multiplied away): ONE memory round trip
        // (round 1 waited for slab 0 + bias, then for the other slabs; 14.2 us per launch at 60 rows x 9488)
        f32x4 p4[4][NQ], bq[NQ];
        const f32x4 zz = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < NQ; ++j) {
            const int qc = min((int)threadIdx.x + j * SEL_THREADS, nq - 1);
            const float *base = src + (size_t)r * V1 + 4 * qc;
#pragma unroll
            for (int u = 0; u < 4; ++u) p4[u][j] = *reinterpret_cast<const f32x4 *>(base + (size_t)(u < splits ? u : 0) * slab_stride);
            bq[j] = bias ? *reinterpret_cast<const f32x4 *>(bias + 4 * qc) : zz;
        }
#pragma unroll
        for (int j = 0; j < NQ; ++j) {
            x[j] = p4[0][j] + bq[j];
#pragma unroll
            for (int u = 1; u < 4; ++u) x[j] += p4[u][j] * (u < splits ? 1.f : 0.f);
        }
        for (int s0 = 4; s0 < splits; s0 += 3) {          // further slabs: 3 x NQ quads of independent loads in flight
            f32x4 p3[3][NQ];
#pragma unroll
            for (int u = 0; u < 3; ++u)
#pragma unroll
                for (int j = 0; j < NQ; ++j) {
                    const int qc = min((int)threadIdx.x + j * SEL_THREADS, nq - 1);
                    p3[u][j] = *reinterpret_cast<const f32x4 *>(src + (size_t)min(s0 + u, splits - 1) * slab_stride + (size_t)r * V1 + 4 * qc);
                    p3[u][j] *= (s0 + u < splits) ? 1.f : 0.f;
                }
#pragma unroll
            for (int j = 0; j < NQ; ++j) x[j] += (p3[0][j] + p3[1][j]) + p3[2][j];
        }
#pragma unroll
        for (int j = 0; j < NQ; ++j)
            if ((int)threadIdx.x + j * SEL_THREADS >= nq) x[j] = f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    }

    float m = -INFINITY;
#pragma unroll
    for (int j = 0; j < NQ; ++j) m = fmaxf(m, fmaxf(fmaxf(x[j][0], x[j][1]), fmaxf(x[j][2], x[j][3])));
    m = block_max(m, s_f);
    float sum = 0.f;
#pragma unroll
    for (int j = 0; j < NQ; ++j)
#pragma unroll
        for (int k = 0; k < 4; ++k) sum += __expf(x[j][k] - m);      // padding lanes hold -inf -> exp = 0
    sum = block_sum(sum, s_f);
    const float lse = m + __logf(sum);

    int token;
    if (my_mode == 2) {
        token = (int)forced[(size_t)r * forced_ld + step];
    } else {
        ArgMax best{-INFINITY, 0x7fffffff};
        if (my_mode == 0) {
#pragma unroll
            for (int j = 0; j < NQ; ++j) {
                const int q = threadIdx.x + j * SEL_THREADS;
                if (q < nq)
#pragma unroll
                    for (int k = 0; k < 4; ++k) best = better(best, ArgMax{x[j][k], 4 * q + k});
            }
        } else {
            const float invT = 1.f / temperature;
            // top-k / nucleus filtering (CaptionModel.py:388-404): both keep the tokens whose tempered logit is >= a
            // threshold, then sample among them -- a restricted Gumbel-max (renormalisation does not move the arg-max).
            // The threshold is found by a 32-step radix descent over the order-preserving integer image of the floats,
            // one block-wide count (top-k) or probability mass (nucleus) per bit.
            uint32_t thr_key = 0;
            if (top_k > 0 || top_p > 0.f) {
                float lse_t = 0.f;
                if (top_p > 0.f) {                 // softmax of the TEMPERED log-probs
                    float mt = -INFINITY, st = 0.f;
#pragma unroll
                    for (int j = 0; j < NQ; ++j)
#pragma unroll
                        for (int k = 0; k < 4; ++k) mt = fmaxf(mt, (x[j][k] - lse) * invT);
                    mt = block_max(mt, s_f);
#pragma unroll
                    for (int j = 0; j < NQ; ++j)
#pragma unroll
                        for (int k = 0; k < 4; ++k) st += __expf((x[j][k] - lse) * invT - mt);
                    st = block_sum(st, s_f);
                    lse_t = mt + __logf(st);
                }
                uint32_t cur = 0;
                for (int bit = 31; bit >= 0; --bit) {
                    const uint32_t cand = cur | (1u << bit);
                    float acc = 0.f;
#pragma unroll
                    for (int j = 0; j < NQ; ++j) {
                        const int q = threadIdx.x + j * SEL_THREADS;
                        if (q < nq)
#pragma unroll
                            for (int k = 0; k < 4; ++k) {
                                const float xt_ = (x[j][k] - lse) * invT;
                                const uint32_t key = order_key(xt_);
                                if (top_k > 0) acc += key >= cand ? 1.f : 0.f;            // how many are >= cand
                                else acc += key > cand ? __expf(xt_ - lse_t) : 0.f;       // mass strictly above cand
                            }
                    }
                    acc = block_sum(acc, s_f);
                    if (top_k > 0 ? acc >= (float)top_k : acc >= top_p) cur = cand;
                }
                // top-k: cur = key of the k-th largest.  nucleus: cur = largest key whose strictly-greater mass is still
                // >= p, so the kept set starts right above it ... unless even the largest token alone has it (cur stays 0)
                thr_key = top_k > 0 ? cur : cur + 1;
            }
            const Philox rng(capmi::epoch_seed(seed, A.epoch));
#pragma unroll
            for (int j = 0; j < NQ; ++j) {
                const int q = threadIdx.x + j * SEL_THREADS;
                if (q < nq) {
                    float gn[4];
                    if (abl & 4) {
                        gn[0] = gn[1] = gn[2] = gn[3] = 0.f;
                    } else if (gumbel) {
                        const f32x4 g = *reinterpret_cast<const f32x4 *>(gumbel + (size_t)r * V1 + 4 * q);
                        gn[0] = g[0]; gn[1] = g[1]; gn[2] = g[2]; gn[3] = g[3];
                    } else {
                        uint32_t o[4];
                        rng.gen(((uint64_t)step << 32) | (uint32_t)r, (uint64_t)q, o);
#pragma unroll
                        for (int k = 0; k < 4; ++k) gn[k] = -__logf(-__logf(u01(o[k])));
                    }
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const float xt_ = (x[j][k] - lse) * invT;
                        if (order_key(xt_) >= thr_key) best = better(best, ArgMax{xt_ + gn[k], 4 * q + k});
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
    float *out = (seq_logp && !(abl & 1)) ? seq_logp + ((size_t)r * L + step) * V1 : nullptr;
#pragma unroll
    for (int j = 0; j < NQ; ++j) {
        const int q = threadIdx.x + j * SEL_THREADS;
        if (q < nq) {
            if (out) {
                f32x4 o = x[j] - (raw_out ? 0.f : lse);
                if (!was_unf) o = f32x4{0.f, 0.f, 0.f, 0.f};
                *reinterpret_cast<f32x4 *>(out + 4 * q) = o;
            }
            if (q == (token >> 2)) s_tok = x[j][token & 3];     // the owner of the chosen logit publishes it
        }
    }
    if (!(abl & 2)) emit_next_embed(ne, r, token);
    __syncthreads();   // s_tok visible; all reads of unfinished[r] done before thread 0 rewrites it
    if (threadIdx.x == 0) {
        seq[(size_t)r * seq_ld + step] = token;
        it_next[r] = token;
        if (sel_logp) sel_logp[(size_t)r * L + step] = keep * (s_tok - (raw_out ? 0.f : lse));
        if (live) live[(size_t)r * L + step] = was_unf ? 1 : 0;
        if (!no_finish_mask) unfinished[r] = (was_unf && token != 0) ? 1 : 0;
        // early exit (AttModel.py:349-350): a row that goes on tells the host so (a word of pinned host memory, only ever set)
        if (ne.alive && (no_finish_mask || (was_unf && token != 0))) *ne.alive = 1;
    }
}
#undef s_tok

// r4 -- select + GEMM in ONE grid.  The select of step t keeps 60 CUs busy for ~14.5 us (a latency chain: slabs -> max / sum-exp ->
// Gumbel arg-max -> embedding gather) while the other 196 idle; two thirds of the NEXT step's attention-LSTM gate GEMM -- the K
// segments fed by h_lang(t) and h_att(t), 32 MB of weights -- depend on nothing this select produces.  Workgroups 0 .. n_sel-1 run the
// select body, the rest run gemm_lc's body on that part (threads >= 768 leave at once: s_barrier counts surviving waves only); the
// following gate GEMM is left with the token-embedding segment and the LSTM cell sums both slab sets.  One launch, no stream, no
// event, no flag: the two halves never talk.  The select's scratch sits behind the ring in the dynamic LDS (static LDS would
// misalign the ring's ds_read_b128).
template <int NQ>
__global__ __launch_bounds__(SEL_THREADS) void select_gemm_kernel(const SelArgs A, const capmi_gemm::KArgs g, int n_sel, int gx, int gy) {
    extern __shared__ __attribute__((aligned(16))) unsigned char fused_lds[];
    if ((int)blockIdx.x < n_sel) {
        unsigned char *scr = fused_lds + capmi_gemm::LC_NS * capmi_gemm::LC_STAGE;
        select_reg_body<NQ>(A, blockIdx.x, reinterpret_cast<float *>(scr), reinterpret_cast<int *>(scr + 128),
                            reinterpret_cast<float *>(scr + 256));
        return;
    }
    if ((int)threadIdx.x >= capmi_gemm::LC_NT) return;
    const int L = (int)blockIdx.x - n_sel;
    capmi_gemm::gemm_lc_body<true, 2, 0, capmi_gemm::LC_WAUX>(g, L % gx, L / gx, gx, gy, fused_lds);
}

template <int NQ>
__global__ __launch_bounds__(SEL_THREADS) void logsoftmax_select_reg_kernel(const SelArgs A) {
    __shared__ float s_f[32];
    __shared__ int s_i[32];
    __shared__ float s_tok;
    select_reg_body<NQ>(A, blockIdx.x, s_f, s_i, &s_tok);
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

// The same gradient when the loss touches the log-probs only through (i) the entry of ONE token per (row, step) and (ii)
// the row sum -- every criterion of the hot path (losses.py:18-37 RewardCriterion, :168-187 new_self_critical, :204-224
// LanguageModelCriterion: only (i); :227-265 LabelSmoothing: (i) and (ii)).  With a = dL/d logp[r,t,tok] and b = dL/d sum_v logp[r,t,v]
//     dlogits[v] = a (1[v == tok] - p_v) + b (1 - V1 p_v)            p = exp(logp)
// plus the dense form above when a dense gradient exists too.  Reads the saved log-probs once and writes dlogits once: the
// dense [N,L,V1] gradient the reference's autograd materialises (gather backward: zero fill + scatter, 2 x 45 MB at bs10 x 5,
// 2 x 255 MB at bs64) never exists.
__global__ __launch_bounds__(SEL_THREADS) void logsoftmax_bwd_sparse_kernel(
    const float *__restrict__ g_sel, const float *__restrict__ g_sum, const int64_t *__restrict__ tok, int tok_ld,
    const float *__restrict__ g, const float *__restrict__ seq_logp, const uint8_t *__restrict__ live,
    float *__restrict__ dlogits, int N, int L, int V1, const float *__restrict__ scale, int raw) {
    __shared__ float s_f[32];
    const int t = blockIdx.x / N, n = blockIdx.x % N;
    const size_t r = (size_t)n * L + t;
    const float *lp = seq_logp + r * V1;
    float *o = dlogits + (size_t)blockIdx.x * V1;
    if (live && !live[r]) {
        for (int v = threadIdx.x; v < V1; v += blockDim.x) o[v] = 0.f;
        return;
    }
    const float sc = scale ? scale[0] : 1.f;          // upstream gradient of the scalar loss (sparse parts only)
    const float a = g_sel ? sc * g_sel[r] : 0.f;
    const float b = g_sum ? sc * g_sum[r] : 0.f;
    const int token = g_sel ? (int)tok[(size_t)n * tok_ld + t] : -1;
    float s = a + b * (float)V1;                      // sum over v of the implied dense gradient
    const float *gr = g ? g + r * V1 : nullptr;
    if (raw) {                                         // the rollout returned the LOGITS (output_logsoftmax = 0): their gradient is
        for (int v = threadIdx.x; v < V1; v += blockDim.x) {      // the loss gradient itself, no softmax Jacobian
            float d = b;
            if (v == token) d += a;
            if (gr) d += gr[v];
            o[v] = d;
        }
        return;
    }
    if (gr) {
        float sd = 0.f;
        for (int v = threadIdx.x; v < V1; v += blockDim.x) sd += gr[v];
        s += block_sum(sd, s_f);
    }
    for (int v = threadIdx.x; v < V1; v += blockDim.x) {
        float d = b - __expf(lp[v]) * s;
        if (v == token) d += a;
        if (gr) d += gr[v];
        o[v] = d;
    }
}

// RewardCriterion (losses.py:18-37) on the selected log-probs a rollout already wrote: mask = (seq > 0) shifted right with a
// leading 1, loss = -sum(sel * reward * mask) / sum(mask) (or per row), and the coefficient d loss / d sel for the backward --
// one launch instead of the reference's gather + ~12 elementwise / reduction launches over [N,L] values.
__global__ __launch_bounds__(1024) void reward_criterion_kernel(const float *__restrict__ sel, int sel_ld,
                                                               const int64_t *__restrict__ seq, int seq_ld,
                                                               const float *__restrict__ reward, int rw_rs, int rw_cs,
                                                               int N_used, int N_all, int L, int per_row,
                                                               float *__restrict__ loss, float *__restrict__ gcoef) {
    __shared__ float s_f[32];
    __shared__ float s_rows[2048];             // per-row mask counts (N_used <= 2048 checked by the host)
    float num = 0.f, den = 0.f;
    for (int i = threadIdx.x; i < N_used * L; i += blockDim.x) {
        const int n = i / L, t = i - n * L;
        const float m = (t == 0 || seq[(size_t)n * seq_ld + t - 1] > 0) ? 1.f : 0.f;
        num -= sel[(size_t)n * sel_ld + t] * reward[(size_t)n * rw_rs + (size_t)t * rw_cs] * m;
        den += m;
    }
    if (per_row) {
        for (int n = threadIdx.x; n < N_used; n += blockDim.x) {
            float rn = 0.f, rd = 0.f;
            for (int t = 0; t < L; ++t) {
                const float m = (t == 0 || seq[(size_t)n * seq_ld + t - 1] > 0) ? 1.f : 0.f;
                rn -= sel[(size_t)n * sel_ld + t] * reward[(size_t)n * rw_rs + (size_t)t * rw_cs] * m;
                rd += m;
            }
            loss[n] = rn / rd;
            s_rows[n] = rd;
        }
        __syncthreads();
    } else {
        num = block_sum(num, s_f);
        den = block_sum(den, s_f);
        if (threadIdx.x == 0) loss[0] = num / den;
    }
    for (int i = threadIdx.x; i < N_all * L; i += blockDim.x) {
        const int n = i / L, t = i - n * L;
        float gc = 0.f;
        if (n < N_used) {
            const float m = (t == 0 || seq[(size_t)n * seq_ld + t - 1] > 0) ? 1.f : 0.f;
            gc = -reward[(size_t)n * rw_rs + (size_t)t * rw_cs] * m / (per_row ? s_rows[n] : den);
        }
        gcoef[i] = gc;
    }
}

}  // namespace

extern "C" {

int capmi_reward_criterion(const float *sel, int sel_ld, const int64_t *seq, int seq_ld, const float *reward, int reward_row_stride,
                           int reward_col_stride, int N_used, int N_all, int L, int per_row, float *loss, float *gcoef,
                           void *stream) {
    if (!sel || !seq || !reward || !loss || !gcoef || N_used <= 0 || N_all < N_used || L <= 0 || sel_ld < L || seq_ld < L)
        return CAPMI_EINVAL;
    if (per_row && N_used > 2048) return CAPMI_EINVAL;
    hipLaunchKernelGGL(reward_criterion_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, sel, sel_ld, seq, seq_ld, reward,
                       reward_row_stride, reward_col_stride, N_used, N_all, L, per_row, loss, gcoef);
    CAPMI_CHECK_LAUNCH();
    return 0;
}

int capmi_logsoftmax_select_partial(const float *partial, int splits, int64_t slab_stride, const float *bias, int N,
                                    int V1, int step, int L, int mode, const uint8_t *row_mode, float temperature,
                                    const float *gumbel, uint64_t seed, const int64_t *forced, int forced_ld,
                                    int no_finish_mask, int64_t *seq, int seq_ld, int64_t *it_next, uint8_t *unfinished,
                                    float *seq_logp, float *sel_logp, uint8_t *live, const capmi_next_embed *next,
                                    const capmi_sample_filter *filter, void *stream) {
    const int raw_out = (mode & CAPMI_SELECT_RAW) ? 1 : 0;       // store logits, not log-probs (output_logsoftmax = 0)
    mode &= ~CAPMI_SELECT_RAW;
    if (!partial || splits < 1 || N <= 0 || V1 <= 0 || step < 0 || step >= L || !seq || !it_next) return CAPMI_EINVAL;
    if (!no_finish_mask && !unfinished) return CAPMI_EINVAL;
    if ((mode == 2 || row_mode) && !forced && mode == 2) return CAPMI_EINVAL;
    if (mode == 1 && !(temperature > 0.f)) return CAPMI_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const int top_k = filter ? filter->top_k : 0;
    const float top_p = filter ? filter->top_p : 0.f;
    if (top_k < 0 || top_p < 0.f || top_p >= 1.f || (top_k > 0 && top_p > 0.f)) return CAPMI_EINVAL;
    NextEmbed ne{};
    if (next && next->x) {
        if (!next->E || next->Edim <= 0) return CAPMI_EINVAL;
        if (next->x_planes && N > 64) return CAPMI_EINVAL;
        ne = NextEmbed{next->E, next->mask, next->x, next->it_save, next->Edim, next->relu,
                       static_cast<unsigned char *>(next->x_planes), nullptr};
    }
    if (next) ne.alive = next->alive;
    const bool al = ((reinterpret_cast<uintptr_t>(partial) | reinterpret_cast<uintptr_t>(bias) |
                      reinterpret_cast<uintptr_t>(gumbel) | reinterpret_cast<uintptr_t>(seq_logp)) & 15) == 0 &&
                    (slab_stride % 4 == 0);
    static const int env_abl = capmi::ablate_env("CAPMI_SEL_ABLATE");   // profiling only
    if (al && V1 % 4 == 0 && V1 <= 3 * 4 * SEL_THREADS) {
        const SelArgs sa{partial, splits, (size_t)slab_stride, bias, V1, step, L, mode, row_mode, temperature, gumbel, seed, forced,
                         forced_ld, no_finish_mask, seq, seq_ld, it_next, unfinished, seq_logp, sel_logp, live, ne, top_k, top_p, env_abl,
                         raw_out, capmi::rng_epoch()};
#define CAPMI_SEL(NQ) hipLaunchKernelGGL(logsoftmax_select_reg_kernel<NQ>, dim3(N), dim3(SEL_THREADS), 0, st, sa)
        if (V1 <= 4 * SEL_THREADS) CAPMI_SEL(1);
        else if (V1 <= 8 * SEL_THREADS) CAPMI_SEL(2);
        else CAPMI_SEL(3);
#undef CAPMI_SEL
        CAPMI_CHECK_LAUNCH();
        return 0;
    }
    hipLaunchKernelGGL(logsoftmax_select_kernel, dim3(N), dim3(SEL_THREADS), 0, st, partial, splits, (size_t)slab_stride, bias,
                       V1, step, L, mode, row_mode,
                       temperature, gumbel, seed, forced, forced_ld, no_finish_mask, seq, seq_ld, it_next, unfinished,
                       seq_logp, sel_logp, live, ne, top_k, top_p, raw_out, capmi::rng_epoch());      // (prenorm = 1 stores and gathers the row as it is)
    CAPMI_CHECK_LAUNCH();
    return 0;
}

int capmi_logsoftmax_select_partial_gemm(const float *partial, int splits, int64_t slab_stride, const float *bias, int N,
                                         int V1, int step, int L, int mode, const uint8_t *row_mode, float temperature,
                                         const float *gumbel, uint64_t seed, const int64_t *forced, int forced_ld,
                                         int no_finish_mask, int64_t *seq, int seq_ld, int64_t *it_next, uint8_t *unfinished,
                                         float *seq_logp, float *sel_logp, uint8_t *live, const capmi_next_embed *next,
                                         const capmi_sample_filter *filter, capmi_gemm_desc *ahead, void *stream) {
    if (!ahead) return CAPMI_EINVAL;
    const int mode_in = mode;
    const int raw_out = (mode & CAPMI_SELECT_RAW) ? 1 : 0;
    mode &= ~CAPMI_SELECT_RAW;
    // what the stand-alone select would run on these arguments
    const int top_k = filter ? filter->top_k : 0;
    const float top_p = filter ? filter->top_p : 0.f;
    const bool al = ((reinterpret_cast<uintptr_t>(partial) | reinterpret_cast<uintptr_t>(bias) | reinterpret_cast<uintptr_t>(gumbel) |
                      reinterpret_cast<uintptr_t>(seq_logp)) & 15) == 0 && (slab_stride % 4 == 0);
    const bool reg_path = partial && splits >= 1 && N > 0 && al && V1 > 0 && V1 % 4 == 0 && V1 <= 3 * 4 * SEL_THREADS &&
                          (!next || !next->x || (next->E && next->Edim > 0 && !(next->x_planes && N > 64)));
    capmi_gemm::LcCapture cap{};
    capmi_gemm::g_lc_capture = &cap;
    const int rc = capmi_gemm_f32(ahead, stream);          // plans the GEMM; launches it only if it is not a loader / consumer GEMM
    capmi_gemm::g_lc_capture = nullptr;
    if (rc) return rc;
    const bool fuse = cap.filled && reg_path && cap.tm == 2 && cap.b_layout == 0 && N + cap.grid_x * cap.grid_y <= 256 &&
                      step >= 0 && step < L && seq && it_next && (no_finish_mask || unfinished) && !(mode == 2 && !forced) &&
                      !(mode == 1 && !(temperature > 0.f)) && top_k >= 0 && top_p >= 0.f && top_p < 1.f && !(top_k > 0 && top_p > 0.f);
    if (!fuse) {
        if (cap.filled) {                                   // planned but not fusable here: launch it the ordinary way
            const int rc2 = capmi_gemm_f32(ahead, stream);
            if (rc2) return rc2;
        }
        return capmi_logsoftmax_select_partial(partial, splits, slab_stride, bias, N, V1, step, L, mode_in, row_mode, temperature, gumbel,
                                               seed, forced, forced_ld, no_finish_mask, seq, seq_ld, it_next, unfinished, seq_logp,
                                               sel_logp, live, next, filter, stream);
    }
    NextEmbed ne{};
    if (next && next->x)
        ne = NextEmbed{next->E, next->mask, next->x, next->it_save, next->Edim, next->relu, static_cast<unsigned char *>(next->x_planes),
                       nullptr};
    if (next) ne.alive = next->alive;
    const SelArgs sa{partial, splits, (size_t)slab_stride, bias, V1, step, L, mode, row_mode, temperature, gumbel, seed, forced,
                     forced_ld, no_finish_mask, seq, seq_ld, it_next, unfinished, seq_logp, sel_logp, live, ne, top_k, top_p, 0, raw_out, capmi::rng_epoch()};
    const size_t lds = (size_t)capmi_gemm::LC_NS * capmi_gemm::LC_STAGE + 512;
    const dim3 grid(N + cap.grid_x * cap.grid_y);
#define CAPMI_FUSED(NQ)                                                                                                     \
    do {                                                                                                                    \
        static bool set = false;                                                                                            \
        if (!set) {                                                                                                         \
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&select_gemm_kernel<NQ>),                              \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);                              \
            set = true;                                                                                                     \
        }                                                                                                                   \
        hipLaunchKernelGGL(select_gemm_kernel<NQ>, grid, dim3(SEL_THREADS), lds, (hipStream_t)stream, sa, cap.a, N, cap.grid_x, \
                           cap.grid_y);                                                                                     \
    } while (0)
    if (V1 <= 4 * SEL_THREADS) CAPMI_FUSED(1);
    else if (V1 <= 8 * SEL_THREADS) CAPMI_FUSED(2);
    else CAPMI_FUSED(3);
#undef CAPMI_FUSED
    CAPMI_CHECK_LAUNCH();
    return 0;
}

int capmi_select_logp(const float *logp, int N, int V1, int step, int L, int mode, float temperature, const float *gumbel,
                      uint64_t seed, int64_t *seq, int seq_ld, int64_t *it_next, uint8_t *unfinished, float *seq_logp,
                      float *sel_logp, int sel_unmasked, const capmi_sample_filter *filter, void *stream) {
    if (!logp || N <= 0 || V1 <= 0 || step < 0 || step >= L || !seq || !it_next || !unfinished) return CAPMI_EINVAL;
    if (mode != 0 && mode != 1) return CAPMI_EINVAL;
    if (mode == 1 && !(temperature > 0.f)) return CAPMI_EINVAL;
    const int top_k = filter ? filter->top_k : 0;
    const float top_p = filter ? filter->top_p : 0.f;
    if (top_k < 0 || top_p < 0.f || top_p >= 1.f || (top_k > 0 && top_p > 0.f)) return CAPMI_EINVAL;
    hipLaunchKernelGGL(logsoftmax_select_kernel, dim3(N), dim3(SEL_THREADS), 0, (hipStream_t)stream, logp, 1, (size_t)0,
                       (const float *)nullptr, V1, step, L, mode, (const uint8_t *)nullptr, temperature, gumbel, seed,
                       (const int64_t *)nullptr, 0, 0, seq, seq_ld, it_next, unfinished, seq_logp, sel_logp,
                       (uint8_t *)nullptr, NextEmbed{}, top_k, top_p, sel_unmasked ? 2 : 1, capmi::rng_epoch());
    CAPMI_CHECK_LAUNCH();
    return 0;
}

int capmi_logsoftmax_select(const float *logits, int N, int V1, int step, int L, int mode, const uint8_t *row_mode,
                            float temperature, const float *gumbel, uint64_t seed, const int64_t *forced,
                            int forced_ld, int no_finish_mask, int64_t *seq, int seq_ld, int64_t *it_next,
                            uint8_t *unfinished, float *seq_logp, float *sel_logp, uint8_t *live, void *stream) {
    return capmi_logsoftmax_select_partial(logits, 1, 0, nullptr, N, V1, step, L, mode, row_mode, temperature, gumbel,
                                           seed, forced, forced_ld, no_finish_mask, seq, seq_ld, it_next, unfinished,
                                           seq_logp, sel_logp, live, nullptr, nullptr, stream);
}

int capmi_logsoftmax_bwd_sparse(const capmi_sparse_logp_grad *sp, const float *g, const float *seq_logp, const uint8_t *live,
                                float *dlogits, int N, int L, int T, int V1, void *stream) {
    if (!sp || !seq_logp || !dlogits || N <= 0 || T <= 0 || T > L || V1 <= 0) return CAPMI_EINVAL;
    if (!sp->g_sel && !sp->g_sum && !g) return CAPMI_EINVAL;
    if (sp->g_sel && (!sp->tok || sp->tok_ld < T)) return CAPMI_EINVAL;
    hipLaunchKernelGGL(logsoftmax_bwd_sparse_kernel, dim3(N * T), dim3(SEL_THREADS), 0, (hipStream_t)stream, sp->g_sel,
                       sp->g_sum, sp->tok, sp->tok_ld, g, seq_logp, live, dlogits, N, L, V1, sp->scale, sp->raw);
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
