// What does one dependent launch cost on this box?  Empty kernels of several geometries, back to back on one stream:
// wall time per launch (host-paired) and in-dispatch event duration.   hipcc --offload-arch=gfx950 -O3 launch_cost.hip -o launch_cost
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
struct Args { float *p[40]; int n[40]; };       // 480-byte argument block like the GEMM's
__global__ void k_empty(Args a) { if (a.n[0] == 12345) a.p[0][threadIdx.x] = 1.f; }
__global__ void k_lds(Args a) { extern __shared__ float l[]; if (a.n[0] == 12345) { l[threadIdx.x] = 1.f; a.p[0][threadIdx.x] = l[0]; } }
__global__ void k_small(float *p, int n) { if (n == 12345) p[threadIdx.x] = 1.f; }
int main() {
    hipStream_t st; hipStreamCreate(&st);
    float *d; hipMalloc(&d, 1 << 20);
    Args a{}; a.p[0] = d;
    hipFuncSetAttribute(reinterpret_cast<const void *>(&k_lds), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    struct Cfg { const char *name; int kind, grid, block, lds; } cfgs[] = {
        {"args480 256x512 lds0", 0, 256, 512, 0}, {"args480 256x512 lds150K", 1, 256, 512, 150 * 1024},
        {"args480 256x256 lds0", 0, 256, 256, 0}, {"args480 512x256 lds70K", 1, 512, 256, 70 * 1024},
        {"args480 60x1024 lds0", 0, 60, 1024, 0}, {"args480 1024x256 lds0", 0, 1024, 256, 0},
        {"args16 256x512 lds0", 2, 256, 512, 0}, {"args16 60x256 lds0", 2, 60, 256, 0}, {"args16 2048x256 lds0", 2, 2048, 256, 0}};
    for (auto &c : cfgs) {
        auto launch = [&](bool ev) {
            if (c.kind == 0) { if (ev) hipExtLaunchKernelGGL(k_empty, dim3(c.grid), dim3(c.block), 0, st, e0, e1, 0, a); else hipLaunchKernelGGL(k_empty, dim3(c.grid), dim3(c.block), 0, st, a); }
            if (c.kind == 1) { if (ev) hipExtLaunchKernelGGL(k_lds, dim3(c.grid), dim3(c.block), c.lds, st, e0, e1, 0, a); else hipLaunchKernelGGL(k_lds, dim3(c.grid), dim3(c.block), c.lds, st, a); }
            if (c.kind == 2) { if (ev) hipExtLaunchKernelGGL(k_small, dim3(c.grid), dim3(c.block), 0, st, e0, e1, 0, d, 1); else hipLaunchKernelGGL(k_small, dim3(c.grid), dim3(c.block), 0, st, d, 1); }
        };
        for (int i = 0; i < 200; ++i) launch(false);
        hipStreamSynchronize(st);
        const int N = 2000;
        auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < N; ++i) launch(false);
        hipStreamSynchronize(st);
        const double wall = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / N;
        double evs = 0;
        for (int i = 0; i < 50; ++i) { launch(true); hipStreamSynchronize(st); float ms; hipEventElapsedTime(&ms, e0, e1); evs += ms * 1e3; }
        printf("%-28s wall/launch %.2f us   in-dispatch event duration %.2f us\n", c.name, wall, evs / 50);
    }
    return 0;
}
