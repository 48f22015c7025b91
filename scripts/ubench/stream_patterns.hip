// Micro-benchmark: how fast can 256 workgroups x 8 waves pull a [N][K] fp32 weight matrix (the decode-step GEMM's
// operand) from HBM / Infinity Cache into registers, by lane->address pattern?  No MFMA: every loaded value is summed so
// the loads stay live.  Build: hipcc --offload-arch=gfx950 -O3 -o /tmp/stream_patterns stream_patterns.hip
//   P0  flat float4 grid-stride copy-like read (ceiling)
//   P1  gemm_ares pattern: wave = 32 weight rows, lane (n = l&31, half = l>>5) reads 64 contiguous bytes of row n per
//       32-wide K chunk (4 x 16 B), PF chunks in flight
//   P2  full-line pattern: wave = 32 rows, one instruction = 8 rows x 128 B (lane l: row l/8, 16-B piece l%8)
//   P3  full-line pattern, 64-wide K chunk: one instruction = 4 rows x 256 B
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__global__ __launch_bounds__(512) void p0_flat(const f32x4 *__restrict__ w, size_t n4, float *out) {
    f32x4 acc = {0, 0, 0, 0};
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + 3 * stride < n4; i += 4 * stride) {
        f32x4 a = w[i], b = w[i + stride], c = w[i + 2 * stride], d = w[i + 3 * stride];
        acc += a + b + c + d;
    }
    for (; i < n4; i += stride) acc += w[i];
    if (acc[0] + acc[1] + acc[2] + acc[3] == 123.456f) out[0] = 1.f;
}

// grid = (N/128 column blocks, splits); workgroup = 8 waves = 4 column groups x 2 K halves, TS chunks of CH k per wave
template <int MODE, int CH, int PF>
__global__ __launch_bounds__(512) void p_rows(const float *__restrict__ w, int N, int K, int TS, float *out) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int cg = wid & 3, kh = wid >> 2;
    const int row0 = blockIdx.x * 128 + 32 * cg;
    const int k_begin = (blockIdx.y * 2 + kh) * TS * CH;
    f32x4 acc = {0, 0, 0, 0};
    constexpr int NL = CH / 8;                      // 16-byte loads per lane per chunk (32 rows x CH floats / 64 lanes / 4)
    f32x4 buf[PF][NL];
    auto issue = [&](int c, f32x4 (&b)[NL]) {
        const int k0 = k_begin + c * CH;
#pragma unroll
        for (int j = 0; j < NL; ++j) {
            int r, k;
            if (MODE == 1) { r = lane & 31; k = (CH / 2) * (lane >> 5) + 4 * j; }                 // 64 (or 128) contiguous bytes per lane
            else if (CH == 32) { r = (lane >> 3) + 8 * j; k = 4 * (lane & 7); }                   // 8 rows x 128 B per instruction
            else { r = (lane >> 4) + 4 * j; k = 4 * (lane & 15); }                                // 4 rows x 256 B per instruction
            const int rr = min(row0 + r, N - 1), kk = min(k0 + k, K - 4);
            b[j] = *reinterpret_cast<const f32x4 *>(w + (size_t)rr * K + kk);
        }
    };
#pragma unroll
    for (int u = 0; u < PF; ++u) issue(u, buf[u]);
    for (int c = 0; c < TS; c += PF) {
#pragma unroll
        for (int u = 0; u < PF; ++u) {
#pragma unroll
            for (int j = 0; j < NL; ++j) acc += buf[u][j];
            if (c + u + PF < TS) issue(c + u + PF, buf[u]);
        }
    }
    if (acc[0] + acc[1] + acc[2] + acc[3] == 123.456f) out[0] = 1.f;
}

int main(int argc, char **argv) {
    const int N = argc > 1 ? atoi(argv[1]) : 4000, K = argc > 2 ? atoi(argv[2]) : 3000;
    const int NBUF = argc > 3 ? atoi(argv[3]) : 3;           // rotate over NBUF matrices (3 x 48 MB ~ the 136 MB a decode step streams)
    const int reps = 30;
    std::vector<float *> bufs(NBUF);
    const size_t elems = (size_t)N * K;
    for (auto &b : bufs) { CK(hipMalloc(&b, elems * 4 + 4096)); CK(hipMemset(b, 1, elems * 4 + 4096)); }
    float *out; CK(hipMalloc(&out, 64));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int nblk = (N + 127) / 128;
    auto run = [&](const char *name, auto launch) {
        for (int i = 0; i < 3 * NBUF; ++i) launch(bufs[i % NBUF]);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        for (int i = 0; i < reps * NBUF; ++i) launch(bufs[i % NBUF]);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        const double us = ms * 1e3 / (reps * NBUF);
        printf("%-44s %7.2f us  %6.2f TB/s\n", name, us, elems * 4.0 / us / 1e6);
    };
    printf("N=%d K=%d (%.1f MB), %d rotating buffers\n", N, K, elems * 4.0 / 1e6, NBUF);
    run("P0 flat float4, 1024 WGs", [&](float *b) { hipLaunchKernelGGL(p0_flat, dim3(1024), dim3(512), 0, 0, (const f32x4 *)b, elems / 4, out); });
    run("P0 flat float4, 256 WGs", [&](float *b) { hipLaunchKernelGGL(p0_flat, dim3(256), dim3(512), 0, 0, (const f32x4 *)b, elems / 4, out); });
    for (int splits : {8, 16}) {
        const int tiles32 = (K + 31) / 32, tiles64 = (K + 63) / 64;
        const int ts32 = (tiles32 + 2 * splits - 1) / (2 * splits), ts64 = (tiles64 + 2 * splits - 1) / (2 * splits);
        char nm[128];
        dim3 g(nblk, splits);
        snprintf(nm, sizeof nm, "P1 64B/lane rows, CH32 PF3, %dx%d WGs", nblk, splits);
        run(nm, [&](float *b) { hipLaunchKernelGGL((p_rows<1, 32, 3>), g, dim3(512), 0, 0, b, N, K, ts32, out); });
        snprintf(nm, sizeof nm, "P1 64B/lane rows, CH32 PF6, %dx%d WGs", nblk, splits);
        run(nm, [&](float *b) { hipLaunchKernelGGL((p_rows<1, 32, 6>), g, dim3(512), 0, 0, b, N, K, ts32, out); });
        snprintf(nm, sizeof nm, "P1 128B/lane rows, CH64 PF3, %dx%d WGs", nblk, splits);
        run(nm, [&](float *b) { hipLaunchKernelGGL((p_rows<1, 64, 3>), g, dim3(512), 0, 0, b, N, K, ts64, out); });
        snprintf(nm, sizeof nm, "P2 full lines 8x128B, CH32 PF3, %dx%d WGs", nblk, splits);
        run(nm, [&](float *b) { hipLaunchKernelGGL((p_rows<2, 32, 3>), g, dim3(512), 0, 0, b, N, K, ts32, out); });
        snprintf(nm, sizeof nm, "P2 full lines 8x128B, CH32 PF6, %dx%d WGs", nblk, splits);
        run(nm, [&](float *b) { hipLaunchKernelGGL((p_rows<2, 32, 6>), g, dim3(512), 0, 0, b, N, K, ts32, out); });
        snprintf(nm, sizeof nm, "P3 full lines 4x256B, CH64 PF3, %dx%d WGs", nblk, splits);
        run(nm, [&](float *b) { hipLaunchKernelGGL((p_rows<2, 64, 3>), g, dim3(512), 0, 0, b, N, K, ts64, out); });
    }
    return 0;
}
