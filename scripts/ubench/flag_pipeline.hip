// Can the decode step's  pointwise kernel -> weight-streaming GEMM -> pointwise kernel ...  chain run as TWO free-running streams
// that hand over through device flags instead of kernel boundaries?  (VERDICT r3 next #5c, the "persistent step" question asked
// in the form that needs no megakernel: the GEMM of step k+1 is launched EARLY on its own stream, its loader waves pull the first
// ring-full of WEIGHTS into LDS while the pointwise kernel it depends on still runs, then wait for that kernel's flag; the next
// pointwise kernel is resident and spinning when the GEMM finishes.)
//
// Stub kernels with the real kernels' geometry and memory behaviour (no arithmetic that matters):
//   G  = gemm_lc_kernel's skeleton: 256 workgroups x 768 threads, 5-stage 28-KB LDS ring filled by 4 loader waves with
//        global_load_lds (16 KB of fp32 weights + 12 KB of activation planes per stage), 8 consumer waves that read the stage,
//        split the weights to bf16x3 and issue the 24 MFMAs per stage, 32 KB of slab stores per workgroup: 48 MB per launch.
//   P  = an LSTM-cell-like pointwise kernel: 250 x 256 threads, reads the 8 K-slice slabs (7.7 MB), writes 0.5 MB.
// Modes:  A  one stream, P G P G ...                      (today)
//         B  stream 1: P P P ..., stream 2: G G G ...; G_k waits for P_k's flag (weights prefetched before), P_k+1 waits for G_k's
//         C  like B but G does not prefetch before the flag    (what the early launch alone buys)
// Every spin is bounded (a give-up word is counted and printed); flags are monotonic counters zeroed before each run.
//   hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize flag_pipeline.hip -o flag_pipeline && ./flag_pipeline
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

typedef const void __attribute__((address_space(1))) *gvoid;
typedef void __attribute__((address_space(3))) *lvoid;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#define VMCNT(n) __builtin_amdgcn_s_waitcnt((((n) >> 4) << 14) | ((n) & 15) | 0x0f70)

constexpr int NS = 5, AB = 12288, WB = 16384, STAGE = AB + WB, NT = 768;

struct Sync {
    unsigned *wait_flag; unsigned wait_val;            // poll until *wait_flag >= wait_val (nullptr: no wait)
    unsigned *ticket; unsigned *done_flag; unsigned done_val;   // last arriving workgroup stores done_val (nullptr: no signal)
    unsigned *err;
    int prefetch;                                      // G: weights of the first ring-full before the wait
};
struct GArgs { const unsigned char *W; const unsigned char *A; float *slab; int nst; Sync s; };
struct PArgs { const float *slab; float *out; int n4; Sync s; };

__device__ __forceinline__ bool spin_ge(unsigned *f, unsigned v, unsigned *err) {
    for (unsigned i = 0; i < (1u << 16); ++i) {
        if (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= v) return true;
        __builtin_amdgcn_s_sleep(4);
    }
    atomicAdd(err, 1u);
    return false;
}
// after every wave's stores were drained (asm vmcnt(0)) and a workgroup barrier: ONE lane publishes
__device__ __forceinline__ void signal_done(const Sync &s, unsigned nwg) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned old = __hip_atomic_fetch_add(s.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (old == nwg - 1) __hip_atomic_store(s.done_flag, s.done_val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__global__ __launch_bounds__(NT) void g_kernel(const GArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char ldsb[];
    unsigned *s_gop = reinterpret_cast<unsigned *>(ldsb + NS * STAGE);      // (no static LDS: it would misalign the ring)
#define s_go (*s_gop)
    const int lane = threadIdx.x & 63;
    const int widu = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int nst = a.nst;
    const int b = blockIdx.x;
    if (threadIdx.x == 0) s_go = 0;
    __syncthreads();
    if (widu >= 8) {
        const int j = widu - 8;
        const unsigned char *Wb = a.W + ((size_t)b * nst) * WB + lane * 16;
        const unsigned char *Ab = a.A + ((size_t)(b >> 5) * nst) * AB + lane * 16;
#define ISSUE_W(s_, slot_) do { _Pragma("unroll") for (int u_ = 0; u_ < 4; ++u_) \
            __builtin_amdgcn_global_load_lds((gvoid)(uintptr_t)(Wb + (size_t)(s_) * WB + (j + 4 * u_) * 1024), \
                                             (lvoid)(ldsb + (slot_) * STAGE + AB + (j + 4 * u_) * 1024), 16, 0, 0); } while (0)
#define ISSUE_A(s_, slot_) do { _Pragma("unroll") for (int u_ = 0; u_ < 3; ++u_) \
            __builtin_amdgcn_global_load_lds((gvoid)(uintptr_t)(Ab + (size_t)(s_) * AB + (j + 4 * u_) * 1024), \
                                             (lvoid)(ldsb + (slot_) * STAGE + (j + 4 * u_) * 1024), 16, 0, 0); } while (0)
        int wpre = 0;                               // stages whose weights were requested before the flag
        if (a.s.wait_flag) {
            if (a.s.prefetch) {
                wpre = min(nst, NS);
                for (int s = 0; s < wpre; ++s) ISSUE_W(s, s);
            }
            if (j == 0) {
                if (lane == 0) spin_ge(a.s.wait_flag, a.s.wait_val, a.s.err);
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                if (lane == 0) __hip_atomic_store(&s_go, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            } else {
                while (__hip_atomic_load(&s_go, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == 0) __builtin_amdgcn_s_sleep(1);
            }
        }
        const int pre = min(nst, 2);
        for (int s = 0; s < pre; ++s) { if (s >= wpre) ISSUE_W(s, s); ISSUE_A(s, s); }
        int fill = pre, fslot = pre;
        for (int k = 0; k <= nst; ++k) {
            if (k < nst) {
                int cnt = 0;                        // DMAs of this wave issued after stage k's last one (they retire in order)
                for (int f = k + 1; f < fill; ++f) cnt += f < wpre ? 3 : 7;
                switch (cnt) {
#define C(n) case n: VMCNT(n); break;
                    C(0) C(3) C(6) C(7) C(9) C(10) C(12) C(13) C(14) C(16) C(17) C(19) C(20) C(21) C(23) C(24) C(26) C(27) C(28)
#undef C
                    default: VMCNT(0); break;
                }
            }
            __builtin_amdgcn_s_barrier();
#pragma unroll
            for (int r = 0; r < 2; ++r)
                if (fill < nst && fill <= k - 2 + NS) {
                    if (fill >= wpre) ISSUE_W(fill, fslot);
                    ISSUE_A(fill, fslot);
                    ++fill;
                    fslot = fslot == NS - 1 ? 0 : fslot + 1;
                }
        }
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_s_barrier();               // the closing barrier before the signal
        return;
    }
    // ---- consumers: the real kernel's per-stage work (weight fragment from LDS, bf16x3 split, 12 plane reads, 24 MFMAs) ----------
    const int cg = widu & 3, par = widu >> 2;
    f32x16 acc0 = {}, acc1 = {};
#define BAR() do { asm volatile("" ::: "memory"); __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); } while (0)
    int bars = 0;
    if (par) { BAR(); ++bars; }
    {
        u32x4 wb[2][3] = {};
        int sidx = par;
        for (int jn = par; jn < nst; jn += 2) {
            BAR(); ++bars;
            const unsigned char *slot = ldsb + sidx * STAGE;
            sidx = sidx >= NS - 2 ? sidx + 2 - NS : sidx + 2;
            float bb[16];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 v = *reinterpret_cast<const f32x4 *>(slot + AB + cg * 4096 + q * 1024 + lane * 16);
                bb[4 * q] = v[0]; bb[4 * q + 1] = v[1]; bb[4 * q + 2] = v[2]; bb[4 * q + 3] = v[3];
            }
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int e2 = 0; e2 < 4; ++e2) {
                    unsigned hh[2], mm[2], ll[2];
#pragma unroll
                    for (int t = 0; t < 2; ++t) {
                        const float x = bb[8 * ks + 2 * e2 + t];
                        hh[t] = __builtin_bit_cast(unsigned, x) & 0xffff0000u;
                        const float r1 = x - __builtin_bit_cast(float, hh[t]);
                        mm[t] = __builtin_bit_cast(unsigned, r1) & 0xffff0000u;
                        ll[t] = __builtin_bit_cast(unsigned, r1 - __builtin_bit_cast(float, mm[t]));
                    }
                    wb[ks][0][e2] = (hh[0] >> 16) | (hh[1] & 0xffff0000u);
                    wb[ks][1][e2] = (mm[0] >> 16) | (mm[1] & 0xffff0000u);
                    wb[ks][2][e2] = (ll[0] >> 16) | (ll[1] & 0xffff0000u);
                }
            auto mma = [&](int ks) {
                bf16x8 bw[3], x0[3], x1[3];
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) {
                    bw[pl] = __builtin_bit_cast(bf16x8, wb[ks][pl]);
                    const unsigned char *q = slot + pl * 4096 + ks * 2048 + (lane & 31) * 64 + (lane >> 5) * 16;
                    x0[pl] = *reinterpret_cast<const bf16x8 *>(q);
                    x1[pl] = *reinterpret_cast<const bf16x8 *>(q + 32);
                }
                constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
                for (int t = 0; t < 6; ++t) {
                    acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x0[PA[t]], bw[PB[t]], acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x1[PA[t]], bw[PB[t]], acc1, 0, 0, 0);
                }
            };
            mma(0);
            BAR(); ++bars;
            mma(1);
        }
    }
    while (bars <= nst) { BAR(); ++bars; }
    BAR();
    BAR();
    if (par == 0) {                                  // 32 KB of slab per workgroup
        float *o = a.slab + (size_t)b * 64 * 128 + cg * 32 + (lane & 31);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            o[(size_t)row * 128] = acc0[r] + acc1[r];
            o[(size_t)(32 + row) * 128] = acc1[r];
        }
    } else {
        asm volatile("" ::"v"(acc0), "v"(acc1));
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    BAR();                                            // (the loaders' closing barrier)
    if (a.s.done_flag && threadIdx.x == 0) signal_done(a.s, gridDim.x);
}

__global__ __launch_bounds__(256) void p_kernel(const PArgs a) {
    __shared__ int dummy;
    if (a.s.wait_flag) {
        if (threadIdx.x == 0) {
            spin_ge(a.s.wait_flag, a.s.wait_val, a.s.err);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            dummy = 1;
        }
        __syncthreads();
    }
    const int i = blockIdx.x * blockDim.x + threadIdx.x;          // one 16-byte piece of the [60, 4000]-float gate rows per thread
    if (i < a.n4) {
        f32x4 v[8];
#pragma unroll
        for (int s = 0; s < 8; ++s) v[s] = *reinterpret_cast<const f32x4 *>(a.slab + (size_t)s * a.n4 * 4 + (size_t)i * 4);
        f32x4 t = ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
#pragma unroll
        for (int e = 0; e < 4; ++e) t[e] = 1.f / (1.f + __expf(-t[e]));
        *reinterpret_cast<f32x4 *>(a.out + (size_t)i * 4) = t;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (a.s.done_flag && threadIdx.x == 0) signal_done(a.s, gridDim.x);
}

__global__ __launch_bounds__(768) void barrier_kernel(unsigned *counter, unsigned *err, int nb) {
    for (int i = 1; i <= nb; ++i) {
        __syncthreads();
        if (threadIdx.x == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            spin_ge(counter, (unsigned)i * gridDim.x, err);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        __syncthreads();
    }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

int main() {
    const int NWG = 256, nst = 12, PAIRS = 40, NWSET = 3;
    const size_t wbytes = (size_t)NWG * nst * WB;                     // 48 MB per weight set
    unsigned char *W[NWSET], *A; float *slab, *pout; unsigned *sync;
    for (auto &w : W) { CK(hipMalloc(&w, wbytes)); CK(hipMemset(w, 0x11, wbytes)); }
    CK(hipMalloc(&A, (size_t)8 * nst * AB)); CK(hipMemset(A, 0x22, (size_t)8 * nst * AB));
    CK(hipMalloc(&slab, (size_t)NWG * 64 * 128 * 4)); CK(hipMemset(slab, 0, (size_t)NWG * 64 * 128 * 4));
    const int n4 = 60 * 4000 / 4;                                     // 60000 pieces, 8 slabs of them fit the 8 MB slab buffer
    CK(hipMalloc(&pout, (size_t)n4 * 16));
    CK(hipMalloc(&sync, 4096));
    unsigned *pflag = sync, *gflag = sync + 16, *err = sync + 32, *tick = sync + 64;      // tick[0 .. 2 PAIRS)
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&g_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, NS * STAGE + 16));
    hipStream_t sp, sg; CK(hipStreamCreateWithFlags(&sp, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&sg, hipStreamNonBlocking));
    hipEvent_t e0, e1, ep; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreateWithFlags(&ep, hipEventDisableTiming));
    const int PG = (n4 + 255) / 256;

    auto run = [&](int mode, int prefetch, bool empty = false) -> double {
        CK(hipMemsetAsync(sync, 0, 4096, sg));
        CK(hipStreamSynchronize(sg));
        CK(hipEventRecord(e0, sg));
        if (mode != 0) { CK(hipStreamWaitEvent(sp, e0, 0)); }
        for (int k = 0; k < PAIRS; ++k) {
            PArgs p{slab, pout, empty ? 0 : n4, {}};
            GArgs g{W[k % NWSET], A, slab, empty ? 0 : nst, {}};
            if (mode == 0) {
                hipLaunchKernelGGL(p_kernel, dim3(PG), dim3(256), 0, sg, p);
                hipLaunchKernelGGL(g_kernel, dim3(NWG), dim3(NT), NS * STAGE + 16, sg, g);
            } else {
                p.s = Sync{k ? gflag : nullptr, (unsigned)k, tick + 2 * k, pflag, (unsigned)(k + 1), err, 0};
                g.s = Sync{pflag, (unsigned)(k + 1), tick + 2 * k + 1, gflag, (unsigned)(k + 1), err, prefetch};
                hipLaunchKernelGGL(p_kernel, dim3(PG), dim3(256), 0, sp, p);
                hipLaunchKernelGGL(g_kernel, dim3(NWG), dim3(NT), NS * STAGE + 16, sg, g);
            }
        }
        if (mode != 0) { CK(hipEventRecord(ep, sp)); CK(hipStreamWaitEvent(sg, ep, 0)); }
        CK(hipEventRecord(e1, sg));
        CK(hipStreamSynchronize(sg));
        CK(hipStreamSynchronize(sp));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        return ms * 1e3 / PAIRS;
    };
    // standalone durations
    {
        PArgs p{slab, pout, n4, {}}; GArgs g{W[0], A, slab, nst, {}};
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipEventRecord(e0, sg));
            for (int k = 0; k < 60; ++k) { g.W = W[k % NWSET]; hipLaunchKernelGGL(g_kernel, dim3(NWG), dim3(NT), NS * STAGE + 16, sg, g); }
            CK(hipEventRecord(e1, sg)); CK(hipStreamSynchronize(sg));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep) printf("G alone, back to back: %.2f us per launch (48 MB of weights, 3 rotating sets)\n", ms * 1e3 / 60);
            CK(hipEventRecord(e0, sg));
            for (int k = 0; k < 60; ++k) hipLaunchKernelGGL(p_kernel, dim3(PG), dim3(256), 0, sg, p);
            CK(hipEventRecord(e1, sg)); CK(hipStreamSynchronize(sg));
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep) printf("P alone, back to back: %.2f us per launch (7.7 MB of slabs in, 1 MB out)\n", ms * 1e3 / 60);
        }
    }
    // D: the hand-offs alone (both kernels with empty bodies): what two flag hand-offs per pair cost vs two kernel boundaries
    {
        std::vector<double> ta, tb;
        for (int i = 0; i < 7; ++i) { ta.push_back(run(0, 0, true)); tb.push_back(run(1, 0, true)); }
        std::sort(ta.begin(), ta.end()); std::sort(tb.begin(), tb.end());
        printf("D  EMPTY bodies: one stream %.2f us per pair (two kernel boundaries); two streams + flags %.2f us per pair (two flag hand-offs)\n",
               ta[3], tb[3]);
    }
    // E: a persistent launch of 256 workgroups that does nothing but grid barriers (monotonic counter, lane-0 release fence before
    //    arriving, relaxed poll + s_sleep, acquire fence after): the price of one in-launch phase boundary
    {
        const int NB = 200;
        std::vector<double> t;
        for (int i = 0; i < 7; ++i) {
            CK(hipMemsetAsync(sync, 0, 4096, sg));
            CK(hipEventRecord(e0, sg));
            hipLaunchKernelGGL(barrier_kernel, dim3(NWG), dim3(768), 0, sg, sync + 128, err, NB);
            CK(hipEventRecord(e1, sg)); CK(hipStreamSynchronize(sg));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            t.push_back(ms * 1e3 / NB);
        }
        std::sort(t.begin(), t.end());
        printf("E  persistent launch, 256 workgroups x 768 threads, %d counter grid barriers: %.2f us per barrier (median of 7)\n", NB, t[3]);
    }
    const char *names[] = {"A  one stream, P G P G ...", "B  two streams + flags, weights prefetched before the flag", "C  two streams + flags, no prefetch"};
    for (int rep = 0; rep < 3; ++rep)
        for (int m = 0; m < 3; ++m) {
            std::vector<double> t;
            for (int i = 0; i < 7; ++i) t.push_back(run(m == 0 ? 0 : 1, m == 1 ? 1 : 0));
            std::sort(t.begin(), t.end());
            unsigned herr = 0; CK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
            if (rep) printf("%-62s %.2f us per (P, G) pair (median of 7; min %.2f)   give-ups %u\n", names[m], t[3], t[0], herr);
        }
    return 0;
}
