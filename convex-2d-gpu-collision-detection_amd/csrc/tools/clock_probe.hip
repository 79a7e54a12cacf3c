// clock_probe.hip — developer tool: is s_memtime a shader-cycle counter, and what clock does the chip hold under a pure
// VALU load?  Every wave issues ITER x 16 independent v_fma_f32 and stamps s_memtime / s_memrealtime (100 MHz) around
// them.  With W waves per SIMD a saturated SIMD spends 2 cycles per wave64 FMA, so one wave sees 2 W ticks per own
// instruction IF a tick is a shader cycle; ticks per microsecond is then the clock.  Usage: clock_probe
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); return 1; } } while (0)
struct Stamp { unsigned long long c0, c1, r0, r1; };
constexpr int ITER = 20000;

__global__ __launch_bounds__(64) void k_fma(float* sink, Stamp* st)
{
    float a[16];
#pragma unroll
    for (int i = 0; i < 16; i++) a[i] = (float)threadIdx.x + i;
    const float m = 1.0000001f, c = 1e-9f;
    Stamp s;
    s.c0 = __builtin_amdgcn_s_memtime();
    s.r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < ITER; it++) {
#pragma unroll
        for (int i = 0; i < 16; i++) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
    }
    s.c1 = __builtin_amdgcn_s_memtime();
    s.r1 = __builtin_amdgcn_s_memrealtime();
    float t = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) t += a[i];
    if (t == 123.456f) sink[0] = t;
    if (threadIdx.x == 0) st[blockIdx.x] = s;
}

int main()
{
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    float* sink; Stamp* st;
    CK(hipMalloc(&sink, 4));
    const int max_grid = prop.multiProcessorCount * 4 * 8;
    CK(hipMalloc(&st, sizeof(Stamp) * max_grid));
    for (int w : {1, 2, 4, 8}) {
        const int grid = prop.multiProcessorCount * 4 * w;
        for (int rep = 0; rep < 3; rep++) hipLaunchKernelGGL(k_fma, dim3(grid), dim3(64), 0, 0, sink, st);
        CK(hipDeviceSynchronize());
        std::vector<Stamp> h(grid);
        CK(hipMemcpy(h.data(), st, sizeof(Stamp) * grid, hipMemcpyDeviceToHost));
        std::vector<double> tpi, ghz;
        for (auto& s : h) {
            tpi.push_back((double)(s.c1 - s.c0) / (ITER * 16.0));
            ghz.push_back((double)(s.c1 - s.c0) / ((s.r1 - s.r0) * 10.0));
        }
        std::sort(tpi.begin(), tpi.end());
        std::sort(ghz.begin(), ghz.end());
        printf("%d waves/SIMD (%d waves): s_memtime ticks per own v_fma_f32: median %.2f (expected %d if tick = shader cycle);  ticks per ns: p10 %.3f median %.3f p90 %.3f;  "
               "median wave time %.1f us\n", w, grid, tpi[grid / 2], w == 1 ? 4 : 2 * w, ghz[grid / 10], ghz[grid / 2], ghz[grid * 9 / 10],
               (double)(h[grid / 2].r1 - h[grid / 2].r0) / 100.0);
    }
    return 0;
}
