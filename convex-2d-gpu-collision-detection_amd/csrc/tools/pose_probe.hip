// pose_probe.hip — developer tool (not part of libc2d.so): when do the waves of the pose-format kernel start and
// end, and at which shader clock do they run?  Each wave stamps s_memtime (shader cycles) and s_memrealtime
// (100 MHz) at entry and exit; the host prints the distribution.  Build: make tools.  Usage: pose_probe [pairs]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../c2d_math.hpp"

using namespace c2d;
typedef float f32x4 __attribute__((ext_vector_type(4)));
struct Planes10 { const float* p[10]; };
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

C2D_DEV uint32_t pose_pair_collides(const float (&v)[10])
{
    float r1[8], r2[8], s, c;
    sincos_(v[4], s, c);
    rect_from_half_extents(v[2] / 2, v[3] / 2, c, s, v[0], v[1], r1);
    sincos_(v[9], s, c);
    rect_from_half_extents(v[7] / 2, v[8] / 2, c, s, v[5], v[6], r2);
    return rect_collide(r1, r2) ? 1u : 0u;
}
C2D_DEV void load10(const Planes10& P, size_t g, f32x4 (&q)[10])
{
#pragma unroll
    for (int k = 0; k < 10; k++) q[k] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(P.p[k]) + g);
}
C2D_DEV uint32_t eval4(const f32x4 (&q)[10])
{
    uint32_t packed = 0;
#pragma unroll
    for (int e = 0; e < 4; e++) {
        float v[10];
#pragma unroll
        for (int k = 0; k < 10; k++) v[k] = q[k][e];
        packed |= pose_pair_collides(v) << (8 * e);
    }
    return packed;
}
struct Stamp { unsigned long long c0, c1, r0, r1; };
C2D_DEV void stamp_in(Stamp& s) { s.c0 = __builtin_amdgcn_s_memtime(); s.r0 = __builtin_amdgcn_s_memrealtime(); }
C2D_DEV void stamp_out(Stamp& s, Stamp* out)
{
    s.c1 = __builtin_amdgcn_s_memtime();
    s.r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) out[blockIdx.x] = s;
}

__global__ __launch_bounds__(64) void k_plain(Planes10 P, size_t n_groups, uint8_t* __restrict__ out, Stamp* st)
{
    Stamp s;
    stamp_in(s);
    const size_t g = (size_t)blockIdx.x * 64 + threadIdx.x;
    if (g < n_groups) {
        f32x4 q[10];
        load10(P, g, q);
        __builtin_nontemporal_store(eval4(q), reinterpret_cast<uint32_t*>(out) + g);
    }
    stamp_out(s, st);
}

__global__ __launch_bounds__(64) void k_pf(Planes10 P, size_t n_groups, uint8_t* __restrict__ out, Stamp* st)
{
    Stamp s;
    stamp_in(s);
    const size_t stride = (size_t)gridDim.x * 64;
    size_t g = (size_t)blockIdx.x * 64 + threadIdx.x;
    f32x4 a[10], b[10];
    if (g < n_groups) load10(P, g, a);
    while (g < n_groups) {
        const size_t g1 = g + stride;
        if (g1 < n_groups) load10(P, g1, b);
        __builtin_nontemporal_store(eval4(a), reinterpret_cast<uint32_t*>(out) + g);
        if (g1 >= n_groups) break;
        const size_t g2 = g1 + stride;
        if (g2 < n_groups) load10(P, g2, a);
        __builtin_nontemporal_store(eval4(b), reinterpret_cast<uint32_t*>(out) + g1);
        g = g2;
    }
    stamp_out(s, st);
}

static void report(const char* name, std::vector<Stamp>& st, float ms)
{
    unsigned long long r_min = ~0ull, r_max = 0;
    for (auto& s : st) { r_min = std::min(r_min, s.r0); r_max = std::max(r_max, s.r1); }
    std::vector<double> start, life, clk;
    for (auto& s : st) {
        start.push_back((s.r0 - r_min) / 100.0);   // us
        life.push_back((s.r1 - s.r0) / 100.0);
        if (s.r1 > s.r0 + 50) clk.push_back((double)(s.c1 - s.c0) / ((s.r1 - s.r0) * 10.0));  // GHz
    }
    auto q = [](std::vector<double> v, double f) { std::sort(v.begin(), v.end()); return v[(size_t)(f * (v.size() - 1))]; };
    printf("%-8s waves %zu  event time %.1f us  first start .. last end %.1f us\n", name, st.size(), ms * 1e3, (r_max - r_min) / 100.0);
    printf("         start offset us: p0 %.1f p10 %.1f p50 %.1f p90 %.1f p100 %.1f\n", q(start, 0), q(start, .1), q(start, .5), q(start, .9), q(start, 1));
    printf("         lifetime     us: p0 %.1f p10 %.1f p50 %.1f p90 %.1f p100 %.1f\n", q(life, 0), q(life, .1), q(life, .5), q(life, .9), q(life, 1));
    if (!clk.empty()) printf("         shader clock GHz (s_memtime / s_memrealtime): p10 %.3f p50 %.3f p90 %.3f\n", q(clk, .1), q(clk, .5), q(clk, .9));
}

int main(int argc, char** argv)
{
    const size_t n = argc > 1 ? strtoull(argv[1], nullptr, 0) : 10000000;
    const size_t n4 = n / 4;
    std::vector<float> h(10 * n);
    unsigned long long x = 88172645463325252ull;
    auto u = [&]() { x ^= x << 13; x ^= x >> 7; x ^= x << 17; return (float)((x >> 11) * (1.0 / 9007199254740992.0)); };
    for (size_t i = 0; i < n; i++)
        for (int r = 0; r < 2; r++) {
            h[(5 * r + 0) * n + i] = u() * 16 - 8; h[(5 * r + 1) * n + i] = u() * 16 - 8;
            h[(5 * r + 2) * n + i] = 0.1f + u() * 4.9f; h[(5 * r + 3) * n + i] = 0.1f + u() * 4.9f; h[(5 * r + 4) * n + i] = u() * 6.2831853f;
        }
    float* d = nullptr; uint8_t* out = nullptr; Stamp* st = nullptr;
    CK(hipMalloc(&d, 10 * n * 4)); CK(hipMalloc(&out, n)); 
    CK(hipMemcpy(d, h.data(), 10 * n * 4, hipMemcpyHostToDevice));
    Planes10 P; for (int k = 0; k < 10; k++) P.p[k] = d + k * n;
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int plain_grid = (int)((n4 + 63) / 64);
    CK(hipMalloc(&st, sizeof(Stamp) * plain_grid));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int pass = 0; pass < 2; pass++)
        for (int w : {0, 2, 4}) {
            const int grid = w == 0 ? plain_grid : prop.multiProcessorCount * 4 * w;
            for (int rep = 0; rep < 30; rep++) {  // warm
                if (w == 0) hipLaunchKernelGGL(k_plain, dim3(grid), dim3(64), 0, 0, P, n4, out, st);
                else hipLaunchKernelGGL(k_pf, dim3(grid), dim3(64), 0, 0, P, n4, out, st);
            }
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0));
            if (w == 0) hipLaunchKernelGGL(k_plain, dim3(grid), dim3(64), 0, 0, P, n4, out, st);
            else hipLaunchKernelGGL(k_pf, dim3(grid), dim3(64), 0, 0, P, n4, out, st);
            CK(hipEventRecord(e1));
            CK(hipDeviceSynchronize());
            float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
            std::vector<Stamp> hs(grid);
            CK(hipMemcpy(hs.data(), st, sizeof(Stamp) * grid, hipMemcpyDeviceToHost));
            char name[32]; snprintf(name, sizeof name, w ? "pf x%d" : "plain", w);
            if (pass == 1) report(name, hs, ms);
        }
    return 0;
}
