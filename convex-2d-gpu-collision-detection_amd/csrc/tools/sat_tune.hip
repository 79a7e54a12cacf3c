// sat_tune.hip — developer tool (not part of libc2d.so): A/B timing of launch-shape and
// cache-policy variants of the rectangle SAT kernel on one device, interleaved rounds in one
// process (cdna_hip_programming.md §5.4 rule 24).  Build: make tools.  Usage: sat_tune [pairs] [rounds]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "../c2d_math.hpp"

using namespace c2d;

typedef float f32x4 __attribute__((ext_vector_type(4)));
struct Planes16 { const float* p[16]; };

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

template <bool NT_LD, bool NT_ST, int BLOCK, int MINW>
__global__ __launch_bounds__(BLOCK, MINW) void k_sat(Planes16 P, size_t n_groups, uint8_t* __restrict__ out)
{
    const size_t stride = (size_t)gridDim.x * BLOCK;
    for (size_t g = (size_t)blockIdx.x * BLOCK + threadIdx.x; g < n_groups; g += stride) {
        f32x4 v[16];
#pragma unroll
        for (int k = 0; k < 16; k++) {
            const f32x4* src = reinterpret_cast<const f32x4*>(P.p[k]) + g;
            v[k] = NT_LD ? __builtin_nontemporal_load(src) : *src;
        }
        uint32_t packed = 0;
#pragma unroll
        for (int e = 0; e < 4; e++) {
            float r1[8], r2[8];
#pragma unroll
            for (int k = 0; k < 8; k++) { r1[k] = v[k][e]; r2[k] = v[8 + k][e]; }
            packed |= (rect_collide(r1, r2) ? 1u : 0u) << (8 * e);
        }
        uint32_t* dst = reinterpret_cast<uint32_t*>(out) + g;
        if (NT_ST) __builtin_nontemporal_store(packed, dst); else *dst = packed;
    }
}

// software-pipelined variant: the next group's 16 loads are in flight while the current group is
// evaluated (ping-pong register buffers, 2 groups per loop trip)
C2D_DEV uint32_t eval4(const f32x4 (&v)[16])
{
    uint32_t packed = 0;
#pragma unroll
    for (int e = 0; e < 4; e++) {
        float r1[8], r2[8];
#pragma unroll
        for (int k = 0; k < 8; k++) { r1[k] = v[k][e]; r2[k] = v[8 + k][e]; }
        packed |= (rect_collide(r1, r2) ? 1u : 0u) << (8 * e);
    }
    return packed;
}
C2D_DEV void load16(const Planes16& P, size_t g, f32x4 (&v)[16])
{
#pragma unroll
    for (int k = 0; k < 16; k++) v[k] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(P.p[k]) + g);
}
template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void k_sat_pf(Planes16 P, size_t n_groups, uint8_t* __restrict__ out)
{
    const size_t stride = (size_t)gridDim.x * BLOCK;
    size_t g = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    f32x4 a[16], b[16];
    if (g < n_groups) load16(P, g, a);
    while (g < n_groups) {
        const size_t g1 = g + stride;
        if (g1 < n_groups) load16(P, g1, b);
        __builtin_nontemporal_store(eval4(a), reinterpret_cast<uint32_t*>(out) + g);
        if (g1 >= n_groups) break;
        const size_t g2 = g1 + stride;
        if (g2 < n_groups) load16(P, g2, a);
        __builtin_nontemporal_store(eval4(b), reinterpret_cast<uint32_t*>(out) + g1);
        g = g2;
    }
}

// 2 pairs per lane (8-byte loads): half the data registers, more waves per SIMD
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void k_sat_v2(Planes16 P, size_t n_groups2, uint8_t* __restrict__ out)
{
    const size_t stride = (size_t)gridDim.x * BLOCK;
    for (size_t g = (size_t)blockIdx.x * BLOCK + threadIdx.x; g < n_groups2; g += stride) {
        f32x2 v[16];
#pragma unroll
        for (int k = 0; k < 16; k++) v[k] = __builtin_nontemporal_load(reinterpret_cast<const f32x2*>(P.p[k]) + g);
        uint32_t packed = 0;
#pragma unroll
        for (int e = 0; e < 2; e++) {
            float r1[8], r2[8];
#pragma unroll
            for (int k = 0; k < 8; k++) { r1[k] = v[k][e]; r2[k] = v[8 + k][e]; }
            packed |= (rect_collide(r1, r2) ? 1u : 0u) << (8 * e);
        }
        __builtin_nontemporal_store((uint16_t)packed, reinterpret_cast<uint16_t*>(out) + g);
    }
}

// 32-bit byte offsets: lets hipcc use the scalar-base + 32-bit VGPR offset form of global_load
// (no 64-bit address arithmetic per plane, 30 fewer VGPRs)
template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void k_sat_saddr(Planes16 P, size_t n_groups, uint8_t* __restrict__ out)
{
    const uint32_t g = blockIdx.x * BLOCK + threadIdx.x;
    if (g >= n_groups) return;
    const uint32_t off = g * 16u;
    f32x4 v[16];
#pragma unroll
    for (int k = 0; k < 16; k++)
        v[k] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(P.p[k]) + off));
    __builtin_nontemporal_store(eval4(v), reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(out) + g * 4u));
}

// counting variants -------------------------------------------------------------------------
// MODE 0: one 64-bit atomic per block on a single word (LDS block reduce)
// MODE 1: per-block partial stored to a workspace, summed by a finishing kernel
template <int BLOCK, int MODE>
__global__ __launch_bounds__(BLOCK) void k_sat_count(Planes16 P, size_t n_groups, uint8_t* __restrict__ out,
                                                     unsigned long long* __restrict__ d_count, uint32_t* __restrict__ partial)
{
    __shared__ uint32_t wave_sums[BLOCK / 64 > 0 ? BLOCK / 64 : 1];
    uint32_t my = 0;
    const size_t stride = (size_t)gridDim.x * BLOCK;
    for (size_t g = (size_t)blockIdx.x * BLOCK + threadIdx.x; g < n_groups; g += stride) {
        f32x4 v[16];
#pragma unroll
        for (int k = 0; k < 16; k++) v[k] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(P.p[k]) + g);
        uint32_t packed = 0;
#pragma unroll
        for (int e = 0; e < 4; e++) {
            float r1[8], r2[8];
#pragma unroll
            for (int k = 0; k < 8; k++) { r1[k] = v[k][e]; r2[k] = v[8 + k][e]; }
            packed |= (rect_collide(r1, r2) ? 1u : 0u) << (8 * e);
        }
        my += __popc(packed);
        __builtin_nontemporal_store(packed, reinterpret_cast<uint32_t*>(out) + g);
    }
    uint32_t v = my;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    if (BLOCK > 64) {
        if ((threadIdx.x & 63) == 0) wave_sums[threadIdx.x >> 6] = v;
        __syncthreads();
        if (threadIdx.x == 0) {
            uint32_t s = 0;
            for (int w = 0; w < BLOCK / 64; w++) s += wave_sums[w];
            v = s;
        }
    }
    if (threadIdx.x == 0) {
        if (MODE == 0) { if (v) atomicAdd(d_count, (unsigned long long)v); }
        else partial[blockIdx.x] = v;
    }
}

__global__ __launch_bounds__(1024) void k_finish(const uint32_t* __restrict__ partial, uint32_t n, unsigned long long* __restrict__ d_count)
{
    __shared__ unsigned long long ws[16];
    unsigned long long s = 0;
    for (uint32_t i = blockIdx.x * 1024 + threadIdx.x; i < n; i += gridDim.x * 1024) s += partial[i];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long t = 0;
        for (int w = 0; w < 16; w++) t += ws[w];
        if (t) atomicAdd(d_count, t);
    }
}

// MODE 2: one returning 64-bit atomic per wave on one of 256 line-separated words; the word packs
// (arrivals << 40 | sum); the last arriver of a word flushes it into *d_count and clears it.
template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void k_sat_count_tk(Planes16 P, size_t n_groups, uint8_t* __restrict__ out,
                                                        unsigned long long* __restrict__ d_count,
                                                        unsigned long long* __restrict__ words)
{
    uint32_t my = 0;
    const size_t g = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    if (g < n_groups) {
        f32x4 v[16];
        load16(P, g, v);
        const uint32_t packed = eval4(v);
        my = __popc(packed);
        __builtin_nontemporal_store(packed, reinterpret_cast<uint32_t*>(out) + g);
    }
    uint32_t v = my;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    if ((threadIdx.x & 63) == 0) {
        const uint32_t wave_id = blockIdx.x * (BLOCK / 64) + (threadIdx.x >> 6);
        const uint32_t n_waves = gridDim.x * (BLOCK / 64);
        const uint32_t slot = wave_id & 255u;
        const uint32_t expected = (n_waves >> 8) + (slot < (n_waves & 255u) ? 1u : 0u);
        unsigned long long* w = words + (size_t)slot * 16;  // 128-B apart
        const unsigned long long old = atomicAdd(w, (1ull << 40) | (unsigned long long)v);
        if ((uint32_t)(old >> 40) + 1u == expected) {
            const unsigned long long total = (old & ((1ull << 40) - 1)) + v;
            *w = 0;  // next launch is stream-ordered behind this kernel
            if (total) atomicAdd(d_count, total);
        }
    }
}

// copy-only ceiling with the same access pattern (16 float4 streams in, 1 dword out)
template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void k_stream(Planes16 P, size_t n_groups, uint8_t* __restrict__ out)
{
    const size_t stride = (size_t)gridDim.x * BLOCK;
    for (size_t g = (size_t)blockIdx.x * BLOCK + threadIdx.x; g < n_groups; g += stride) {
        f32x4 acc = {0, 0, 0, 0};
#pragma unroll
        for (int k = 0; k < 16; k++) acc += __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(P.p[k]) + g);
        uint32_t packed = (acc[0] > 0) | ((acc[1] > 0) << 8) | ((acc[2] > 0) << 16) | ((acc[3] > 0) << 24);
        __builtin_nontemporal_store(packed, reinterpret_cast<uint32_t*>(out) + g);
    }
}

struct Variant { std::string name; void (*launch)(Planes16, size_t, uint8_t*, int grid, hipStream_t); int block; int grid_cap; std::vector<float> ms; };

template <bool L, bool S, int B, int W>
void launch_sat(Planes16 P, size_t ng, uint8_t* out, int grid, hipStream_t s) { hipLaunchKernelGGL((k_sat<L, S, B, W>), dim3(grid), dim3(B), 0, s, P, ng, out); }
static unsigned long long* g_count; static uint32_t* g_partial;
template <int B, int MODE>
void launch_count(Planes16 P, size_t ng, uint8_t* out, int grid, hipStream_t s)
{
    hipLaunchKernelGGL((k_sat_count<B, MODE>), dim3(grid), dim3(B), 0, s, P, ng, out, g_count, g_partial);
    if (MODE == 1) hipLaunchKernelGGL(k_finish, dim3((grid + 8191) / 8192), dim3(1024), 0, s, g_partial, (uint32_t)grid, g_count);
}
static unsigned long long* g_words;
template <int B>
void launch_tk(Planes16 P, size_t ng, uint8_t* out, int grid, hipStream_t s) { hipLaunchKernelGGL((k_sat_count_tk<B>), dim3(grid), dim3(B), 0, s, P, ng, out, g_count, g_words); }
template <int B>
void launch_saddr(Planes16 P, size_t ng, uint8_t* out, int grid, hipStream_t s) { hipLaunchKernelGGL((k_sat_saddr<B>), dim3(grid), dim3(B), 0, s, P, ng, out); }
template <int B>
void launch_pf(Planes16 P, size_t ng, uint8_t* out, int grid, hipStream_t s) { hipLaunchKernelGGL((k_sat_pf<B>), dim3(grid), dim3(B), 0, s, P, ng, out); }
template <int B>
void launch_v2(Planes16 P, size_t ng, uint8_t* out, int grid, hipStream_t s) { hipLaunchKernelGGL((k_sat_v2<B>), dim3(grid), dim3(B), 0, s, P, ng * 2, out); }
template <int B>
void launch_stream(Planes16 P, size_t ng, uint8_t* out, int grid, hipStream_t s) { hipLaunchKernelGGL((k_stream<B>), dim3(grid), dim3(B), 0, s, P, ng, out); }

int main(int argc, char** argv)
{
    size_t n = argc > 1 ? strtoull(argv[1], nullptr, 0) : 10000000;
    int rounds = argc > 2 ? atoi(argv[2]) : 15;
    n &= ~(size_t)3;
    const size_t ng = n / 4;
    float* d = nullptr;
    CK(hipMalloc(&d, 16 * n * sizeof(float)));
    std::vector<float> h(16 * n);
    srand(1);
    for (auto& x : h) x = (float)rand() / RAND_MAX * 16.f - 8.f;
    CK(hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    uint8_t* out = nullptr;
    CK(hipMalloc(&out, n));
    Planes16 P;
    for (int k = 0; k < 16; k++) P.p[k] = d + (size_t)k * n;
    hipStream_t s;
    CK(hipStreamCreate(&s));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    std::vector<Variant> V;
    const int full = 1 << 30;
    auto add = [&](std::string nm, auto fn, int block, int cap) { V.push_back(Variant{nm, fn, block, cap, {}}); };
    add("nt/nt b256 cap2048", launch_sat<true, true, 256, 1>, 256, 2048);
    add("nt/nt b256 cap4096", launch_sat<true, true, 256, 1>, 256, 4096);
    add("nt/nt b256 cap1024", launch_sat<true, true, 256, 1>, 256, 1024);
    add("nt/nt b256 cap1280", launch_sat<true, true, 256, 1>, 256, 1280);
    add("nt/nt b256 cap2560", launch_sat<true, true, 256, 1>, 256, 2560);
    add("nt/nt b256 full", launch_sat<true, true, 256, 1>, 256, full);
    add("ld/st b256 cap2048", launch_sat<false, false, 256, 1>, 256, 2048);
    add("ld/st b256 full", launch_sat<false, false, 256, 1>, 256, full);
    add("nt/st b256 cap2048", launch_sat<true, false, 256, 1>, 256, 2048);
    add("ld/nt b256 cap2048", launch_sat<false, true, 256, 1>, 256, 2048);
    add("nt/nt b512 cap1024", launch_sat<true, true, 512, 1>, 512, 1024);
    add("nt/nt b512 full", launch_sat<true, true, 512, 1>, 512, full);
    add("nt/nt b128 cap4096", launch_sat<true, true, 128, 1>, 128, 4096);
    add("nt/nt b128 full", launch_sat<true, true, 128, 1>, 128, full);
    add("nt/nt b64 full", launch_sat<true, true, 64, 1>, 64, full);
    add("nt/nt b256 w6 cap2048", launch_sat<true, true, 256, 6>, 256, 2048);
    add("nt/nt b256 w8 cap2048", launch_sat<true, true, 256, 8>, 256, 2048);
    add("nt/nt b256 w8 full", launch_sat<true, true, 256, 8>, 256, full);
    CK(hipMalloc(&g_count, 8)); CK(hipMemset(g_count, 0, 8)); CK(hipMalloc(&g_partial, 4 << 20));
    add("count atomic b256 cap2048", launch_count<256, 0>, 256, 2048);
    add("count atomic b256 cap3072", launch_count<256, 0>, 256, 3072);
    add("count atomic b256 cap4096", launch_count<256, 0>, 256, 4096);
    add("count atomic b256 cap5120", launch_count<256, 0>, 256, 5120);
    add("count atomic b512 cap2048", launch_count<512, 0>, 512, 2048);
    add("count atomic b512 cap2560", launch_count<512, 0>, 512, 2560);
    add("count atomic b1024 cap1280", launch_count<1024, 0>, 1024, 1280);
    add("count atomic b256 full", launch_count<256, 0>, 256, full);
    add("count partial b256 cap2048", launch_count<256, 1>, 256, 2048);
    add("count partial b256 full", launch_count<256, 1>, 256, full);
    add("count partial b64 full", launch_count<64, 1>, 64, full);
    add("count partial b512 full", launch_count<512, 1>, 512, full);
    CK(hipMalloc(&g_words, 256 * 128)); CK(hipMemset(g_words, 0, 256 * 128));
    add("count ticket b64 full", launch_tk<64>, 64, full);
    add("count ticket b256 full", launch_tk<256>, 256, full);
    add("saddr b64 full", launch_saddr<64>, 64, full);
    add("saddr b128 full", launch_saddr<128>, 128, full);
    add("saddr b256 full", launch_saddr<256>, 256, full);
    add("prefetch b256 cap512", launch_pf<256>, 256, 512);
    add("prefetch b256 cap768", launch_pf<256>, 256, 768);
    add("prefetch b256 cap1024", launch_pf<256>, 256, 1024);
    add("prefetch b256 cap1536", launch_pf<256>, 256, 1536);
    add("prefetch b64 cap3072", launch_pf<64>, 64, 3072);
    add("prefetch b64 cap4096", launch_pf<64>, 64, 4096);
    add("vec2 b256 full", launch_v2<256>, 256, full);
    add("vec2 b256 cap4096", launch_v2<256>, 256, 4096);
    add("stream-only b256 cap2048", launch_stream<256>, 256, 2048);
    add("stream-only b256 full", launch_stream<256>, 256, full);
    const bool sustained = argc > 3 && std::string(argv[3]) == "sustained";
    if (sustained) {
        // steady-state clocks: 200 untimed launches, then 100 timed, per variant, two passes
        printf("%-28s %10s %10s %8s   (sustained)\n", "variant", "pass1_us", "pass2_us", "frac8T");
        for (auto& v : V) {
            size_t blocks = ((v.name.rfind("vec2", 0) == 0 ? 2 * ng : ng) + v.block - 1) / v.block;
            int grid = (int)std::min<size_t>(blocks, (size_t)v.grid_cap);
            float res[2];
            for (int pass = 0; pass < 2; pass++) {
                for (int i = 0; i < 200; i++) v.launch(P, ng, out, grid, s);
                CK(hipEventRecord(e0, s));
                for (int i = 0; i < 100; i++) v.launch(P, ng, out, grid, s);
                CK(hipEventRecord(e1, s));
                CK(hipEventSynchronize(e1));
                float ms;
                CK(hipEventElapsedTime(&ms, e0, e1));
                res[pass] = ms / 100 * 1e3f;
            }
            double gbs = 65.0 * n / (res[1] * 1e-6) / 1e9;
            printf("%-28s %10.2f %10.2f %8.3f\n", v.name.c_str(), res[0], res[1], gbs / 8000.0);
        }
        return 0;
    }
    const int inner = 10;
    for (int r = 0; r < rounds + 1; r++) {
        for (auto& v : V) {
            size_t blocks = (ng + v.block - 1) / v.block;
            int grid = (int)std::min<size_t>(blocks, (size_t)v.grid_cap);
            CK(hipEventRecord(e0, s));
            for (int i = 0; i < inner; i++) v.launch(P, ng, out, grid, s);
            CK(hipEventRecord(e1, s));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (r > 0) v.ms.push_back(ms / inner);
        }
    }
    printf("%-28s %10s %10s %10s %8s\n", "variant", "med_us", "min_us", "GB/s(med)", "frac8T");
    for (auto& v : V) {
        std::sort(v.ms.begin(), v.ms.end());
        float med = v.ms[v.ms.size() / 2], mn = v.ms[0];
        double gbs = 65.0 * n / (med * 1e-3) / 1e9;
        printf("%-28s %10.2f %10.2f %10.1f %8.3f\n", v.name.c_str(), med * 1e3, mn * 1e3, gbs, gbs / 8000.0);
    }
    return 0;
}
