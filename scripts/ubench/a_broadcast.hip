// How fast can all 256 workgroups (512 threads) pull a 98 KB activation slice that 32 of them SHARE (8 slices of a [64 x 3000]
// fp32 matrix: the gemm_ares staging pattern), compared with 256 private slices, and with the shared slice pre-loaded into
// the XCD's L2?  Cycles from first load issue to last data landed, per workgroup (median / max), and wall time.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

// MODE 0: slice = blockIdx.y (8 slices x 32 column blocks, dispatch order x fastest => a slice's 32 WGs spread over all XCDs)
// MODE 1: private slice per workgroup (no sharing)
// MODE 2: slice = XCD (linear id % 8): the 32 sharers sit on one XCD
template <int MODE>
__global__ __launch_bounds__(512) void k(const float *__restrict__ a, int lda, float *out, unsigned long long *stamps) {
    const int L = blockIdx.y * gridDim.x + blockIdx.x;
    const int slice = MODE == 0 ? blockIdx.y : MODE == 1 ? L : (L & 7);
    const float *base = MODE == 1 ? a + (size_t)slice * 64 * 384 : a + (size_t)slice * 384;
    const int ld = MODE == 1 ? 384 : lda;
    f32x4 v[12];
    const unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll
    for (int j = 0; j < 12; ++j) {
        const int idx = j * 512 + threadIdx.x;
        const int row = idx / 96, c4 = idx - row * 96;
        v[j] = *reinterpret_cast<const f32x4 *>(base + (size_t)row * ld + c4 * 4);
    }
    f32x4 acc = {0, 0, 0, 0};
#pragma unroll
    for (int j = 0; j < 12; ++j) acc += v[j];
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (acc[0] + acc[1] + acc[2] + acc[3] == 123.456f) out[0] = 1.f;
    if ((threadIdx.x & 63) == 0) stamps[L * 8 + (threadIdx.x >> 6)] = t1 - t0;
}
__global__ void touch(float *a, size_t n) {          // "previous kernel" writes the activations
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) a[i] = (float)(i & 1023);
}

int main() {
    const int lda = 3000 + 72;                   // 64 rows x 3072 floats
    float *a; CK(hipMalloc(&a, (size_t)256 * 64 * 384 * 4));
    float *out; CK(hipMalloc(&out, 64));
    unsigned long long *st; CK(hipMalloc(&st, 2048 * 8));
    std::vector<unsigned long long> h(2048);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto run = [&](const char *name, auto launch, size_t touch_n) {
        float ms_sum = 0;
        for (int i = 0; i < 20; ++i) {
            hipLaunchKernelGGL(touch, dim3(64), dim3(256), 0, 0, a, touch_n);    // fresh data from another kernel each time
            CK(hipEventRecord(e0));
            launch();
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (i >= 5) ms_sum += ms;
        }
        CK(hipMemcpy(h.data(), st, 2048 * 8, hipMemcpyDeviceToHost));
        std::sort(h.begin(), h.end());
        printf("%-52s %6.2f us | per-wave cycles: median %6llu  p90 %6llu  max %6llu\n", name, ms_sum / 15 * 1e3, h[1024], h[1843], h[2047]);
    };
    run("shared slices, sharers spread over XCDs (gemm_ares)", [&] { hipLaunchKernelGGL(k<0>, dim3(32, 8), dim3(512), 0, 0, a, lda, out, st); }, (size_t)64 * lda);
    run("shared slices, sharers on one XCD", [&] { hipLaunchKernelGGL(k<2>, dim3(32, 8), dim3(512), 0, 0, a, lda, out, st); }, (size_t)64 * lda);
    run("private slices (25 MB total)", [&] { hipLaunchKernelGGL(k<1>, dim3(32, 8), dim3(512), 0, 0, a, lda, out, st); }, (size_t)256 * 64 * 384);
    return 0;
}
