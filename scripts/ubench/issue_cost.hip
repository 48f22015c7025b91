// How long does a wave take to ISSUE its weight loads, by address pattern?  (The wave is in-order: while the CU's address
// pipeline digests a load that touches 64 separate lines the wave cannot issue anything else.)
//   pattern 0: gemm_ares [N][K] pattern: lane (n = l&31, half) reads 16 B of row n at k0 + 16*half + 4q    (64 lines / instr)
//   pattern 1: packed tiles: lane l reads 16 B at base + 16*l                                               (8 lines / instr)
//   pattern 2: 16-row pattern (MFMA 16x16x32 operand): lane (n = l&15, g = l>>4) reads 16 B of row n at k0 + 4g (+16) (16 segments of 64 B)
//   pattern 3: [K][N] dword pattern: lane (n, half) reads 4 B at row k0+16*half+j, col n                   (2 lines / instr, 16 instr per chunk)
// 256 workgroups x 8 waves, each wave issues NCH chunks (32 cols x 32 k of fp32 = 4 KB) back to back, then consumes.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int PAT, int NCH>
__global__ __launch_bounds__(512) void k(const float *__restrict__ w, int N, int K, float *out, unsigned long long *stamps) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int wave = blockIdx.x * 8 + wid;               // global wave id: owns NCH consecutive chunks
    f32x4 buf[NCH][4];
    const unsigned long long t0 = __builtin_readcyclecounter();
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const size_t chunk = (size_t)wave * NCH + c;
        if (PAT == 1) {
            const float *p = w + chunk * 1024 + lane * 4;
#pragma unroll
            for (int q = 0; q < 4; ++q) buf[c][q] = *reinterpret_cast<const f32x4 *>(p + q * 256);
        } else if (PAT == 0 || PAT == 2) {
            const int kchunks = K / 32;
            const int rb = (int)(chunk / kchunks), kc = (int)(chunk % kchunks);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                int r, kk;
                if (PAT == 0) { r = lane & 31; kk = 16 * (lane >> 5) + 4 * q; }
                else { r = (lane & 15) + 16 * (q >> 1); kk = 4 * (lane >> 4) + 16 * (q & 1); }
                buf[c][q] = *reinterpret_cast<const f32x4 *>(w + (size_t)(rb * 32 + r) * K + kc * 32 + kk);
            }
        } else {
            const int nb = N / 32;
            const int kc = (int)(chunk / nb), cb = (int)(chunk % nb);
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    buf[c][q][e] = w[(size_t)(kc * 32 + 16 * (lane >> 5) + 4 * q + e) * N + cb * 32 + (lane & 31)];
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    __builtin_amdgcn_sched_barrier(0);
    f32x4 acc = {0, 0, 0, 0};
#pragma unroll
    for (int c = 0; c < NCH; ++c)
#pragma unroll
        for (int q = 0; q < 4; ++q) acc += buf[c][q];
    const unsigned long long t2 = __builtin_readcyclecounter();
    if (lane == 0) { stamps[wave * 2] = t1 - t0; stamps[wave * 2 + 1] = t2 - t0; }
    if (acc[0] + acc[1] + acc[2] + acc[3] == 123.456f) out[0] = 1.f;
}

int main() {
    const int N = 4096, K = 3072;             // 48 MB; 256 WGs x 8 waves x 6 chunks x 4 KB = 48 MB exactly
    const size_t elems = (size_t)N * K;
    float *w[3]; for (auto &b : w) { CK(hipMalloc(&b, elems * 4)); CK(hipMemset(b, 1, elems * 4)); }
    float *out; CK(hipMalloc(&out, 64));
    unsigned long long *st; CK(hipMalloc(&st, 2048 * 2 * 8));
    std::vector<unsigned long long> h(2048 * 2);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto run = [&](const char *name, auto launch) {
        for (int i = 0; i < 9; ++i) launch(w[i % 3]);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        for (int i = 0; i < 60; ++i) launch(w[i % 3]);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        CK(hipMemcpy(h.data(), st, h.size() * 8, hipMemcpyDeviceToHost));
        std::vector<unsigned long long> iss, tot;
        for (int i = 0; i < 2048; ++i) { iss.push_back(h[2 * i]); tot.push_back(h[2 * i + 1]); }
        std::sort(iss.begin(), iss.end()); std::sort(tot.begin(), tot.end());
        printf("%-34s %6.2f us/launch | issue 24 loads: median %6llu max %6llu cycles | all data landed: median %6llu max %6llu\n", name,
               ms * 1e3 / 60, iss[1024], iss[2047], tot[1024], tot[2047]);
    };
    run("P0 [N][K] 64 lines/instr", [&](float *b) { hipLaunchKernelGGL((k<0, 6>), dim3(256), dim3(512), 0, 0, b, N, K, out, st); });
    run("P1 packed 1 KB/instr", [&](float *b) { hipLaunchKernelGGL((k<1, 6>), dim3(256), dim3(512), 0, 0, b, N, K, out, st); });
    run("P2 16 rows x 64 B /instr", [&](float *b) { hipLaunchKernelGGL((k<2, 6>), dim3(256), dim3(512), 0, 0, b, N, K, out, st); });
    run("P3 [K][N] dword 2 lines/instr", [&](float *b) { hipLaunchKernelGGL((k<3, 6>), dim3(256), dim3(512), 0, 0, b, N, K, out, st); });
    return 0;
}
