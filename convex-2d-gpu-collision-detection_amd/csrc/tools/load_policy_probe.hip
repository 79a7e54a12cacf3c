// load_policy_probe — does a cache-policy bit on the loads move the rate at which one MI355X streams the rectangle-pair batch?
// Developer tool behind DESIGN.md §5.1 ("6.1-6.4 TB/s, the rate at which this chip copies float4 streams").  The access pattern is
// sat_rect_verts_kernel's: sixteen f32 planes of n values, a single-wave block takes 256 consecutive pairs, every lane loads one
// 16-byte vector per plane (sixteen loads in flight) and stores four result bytes.  No arithmetic beyond a sum that keeps the loads
// alive: what is measured is the memory system.  Variants:
//   global        plain global_load_dwordx4
//   global nt     __builtin_nontemporal_load (what the product's kernel uses)
//   buffer aux=k  raw buffer loads with the gfx940+ cache-policy bits of the instruction: sc0 = 1, nt = 2, sc1 = 16 (all eight combinations)
// and for the best of them the stores with and without the non-temporal hint, the grid as resident blocks that stride, and the same
// stream with arithmetic on the loaded values (0 .. 2048 VALU instructions per lane; the product kernel issues 928).
// A second part does the same for the polygon kernels' shape: one pair per lane, 4-byte loads of 16 / 38 / 64 rows, one result byte.
// Usage: load_policy_probe [pairs]     (default 1e7: 650 MB per pass; prints microseconds per pass and GB/s, median of 5 timings of 40 passes)
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x)                                                                                  \
    do {                                                                                          \
        hipError_t e__ = (x);                                                                     \
        if (e__ != hipSuccess) {                                                                  \
            fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e__));   \
            exit(1);                                                                              \
        }                                                                                         \
    } while (0)

typedef float v4f __attribute__((ext_vector_type(4)));
typedef int v4i __attribute__((ext_vector_type(4)));

// MODE 0: global, 1: global nt, 2: raw buffer with AUX; WORK: arithmetic per lane on the loaded values, in rounds of 128 VALU
// instructions (a multiply and an add on each of the 64 loaded floats: nothing can start before the loads land)
template <int MODE, int AUX, bool NT_STORE, int WORK = 0>
__global__ __launch_bounds__(64) void stream_kernel(const float* __restrict__ planes, size_t n, uint32_t* __restrict__ out, size_t groups)
{
    const size_t stride = (size_t)gridDim.x * 64;
    for (size_t g = (size_t)blockIdx.x * 64 + threadIdx.x; g < groups; g += stride) {   // g: group of four pairs
        v4f acc = {0.f, 0.f, 0.f, 0.f};
        v4f v[16];
        if constexpr (MODE == 2) {
            // one descriptor per plane (a plane is 40 MB; offsets stay in 32 bits)
#pragma unroll
            for (int p = 0; p < 16; p++) {
                const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(planes + (size_t)p * n), 0, (int)(n * 4), 0x00020000);
                const v4i w = __builtin_amdgcn_raw_buffer_load_b128(r, (int)(g * 16), 0, AUX);
                v[p] = __builtin_bit_cast(v4f, w);
            }
        } else {
#pragma unroll
            for (int p = 0; p < 16; p++) {
                const v4f* src = reinterpret_cast<const v4f*>(planes + (size_t)p * n) + g;
                v[p] = MODE == 1 ? __builtin_nontemporal_load(src) : *src;
            }
        }
#pragma unroll 1
        for (int w = 0; w < WORK; w++) {   // one round: 64 v_mul_f32 + 64 v_add_f32 (built with -ffp-contract=off -fno-slp-vectorize: neither fused nor packed)
#pragma unroll
            for (int i = 0; i < 64; i++) v[i >> 2][i & 3] = v[i >> 2][i & 3] * 1.0000001f + v[((i >> 2) + 1) & 15][i & 3];
        }
#pragma unroll
        for (int p = 0; p < 16; p++) acc += v[p];
        const uint32_t bits = (acc.x > 0.f ? 1u : 0u) | (acc.y > 0.f ? 0x100u : 0u) | (acc.z > 0.f ? 0x10000u : 0u) | (acc.w > 0.f ? 0x1000000u : 0u);
        if (NT_STORE) __builtin_nontemporal_store(bits, out + g);
        else out[g] = bits;
    }
}

// The polygon kernels' shape: a single-wave block takes 64 pairs (TILES x 64 with TILES > 1), one pair per lane, and reads ROWS
// f32 rows of n values with 4-byte loads (a wave's request for a row is 256 contiguous bytes); one result byte per pair.
// BUF: raw buffer loads with the row offset in an SGPR (what sat_poly_binned_kernel does) instead of global loads.
template <int ROWS, int TILES, bool BUF>
__global__ __launch_bounds__(64) void row_stream_kernel(const float* __restrict__ rows, size_t n, uint8_t* __restrict__ out)
{
#pragma unroll 1
    for (int t = 0; t < TILES; t++) {
        const size_t i = ((size_t)blockIdx.x * TILES + t) * 64 + threadIdx.x;
        if (i >= n) return;
        float v[ROWS];
        if constexpr (BUF) {
            const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(rows) + ((size_t)blockIdx.x * TILES + t) * 64, 0, 0x7fffffff, 0x00020000);
            uint32_t soff = 0;
#pragma unroll
            for (int k = 0; k < ROWS; k++) {
                v[k] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)(threadIdx.x * 4), (int)soff, 2));
                soff += (uint32_t)(n * 4);
            }
        } else {
#pragma unroll
            for (int k = 0; k < ROWS; k++) v[k] = __builtin_nontemporal_load(rows + (size_t)k * n + i);
        }
        float acc = 0.f;
#pragma unroll
        for (int k = 0; k < ROWS; k++) acc += v[k];
        __builtin_nontemporal_store((uint8_t)(acc > 0.f ? 1 : 0), out + i);
    }
}

template <typename F>
static double time_us(F launch, hipStream_t s)
{
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (int i = 0; i < 60; i++) launch();   // clocks ramp for a few milliseconds after idle
    CHECK(hipStreamSynchronize(s));
    std::vector<double> t;
    for (int rep = 0; rep < 5; rep++) {
        CHECK(hipEventRecord(e0, s));
        for (int i = 0; i < 40; i++) launch();
        CHECK(hipEventRecord(e1, s));
        CHECK(hipEventSynchronize(e1));
        float ms = 0.f;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        t.push_back(ms * 1000.0 / 40);
    }
    std::sort(t.begin(), t.end());
    CHECK(hipEventDestroy(e0));
    CHECK(hipEventDestroy(e1));
    return t[2];
}

int main(int argc, char** argv)
{
    const size_t n = argc > 1 ? (size_t)atof(argv[1]) : 10000000;
    if (n % 4 || n * 4 > 0x7fffffffull) { fprintf(stderr, "pairs: a multiple of 4, below 2^29\n"); return 1; }
    const size_t groups = n / 4;
    float* planes = nullptr;
    uint32_t* out = nullptr;
    CHECK(hipMalloc(&planes, 16 * n * sizeof(float)));
    CHECK(hipMalloc(&out, groups * sizeof(uint32_t)));
    CHECK(hipMemset(planes, 0x3c, 16 * n * sizeof(float)));
    hipStream_t s;
    CHECK(hipStreamCreate(&s));
    const double bytes = 65.0 * (double)n;
    const unsigned blocks = (unsigned)((groups + 63) / 64);
    printf("# %zu pairs, %.1f MB per pass, %u single-wave blocks (one group of four pairs per lane)\n", n, bytes / 1e6, blocks);
    auto report = [&](const char* name, double us) { printf("%-44s %8.1f us  %7.1f GB/s  %.3f of 8 TB/s\n", name, us, bytes / us / 1e3, bytes / us / 1e3 / 8000.0); fflush(stdout); };
#define RUN(name, MODE, AUX, NTS, GRID) report(name, time_us([&] { hipLaunchKernelGGL((stream_kernel<MODE, AUX, NTS>), dim3(GRID), dim3(64), 0, s, planes, n, out, groups); }, s))
#define RUNW(name, WORK) report(name, time_us([&] { hipLaunchKernelGGL((stream_kernel<1, 0, true, WORK>), dim3(blocks), dim3(64), 0, s, planes, n, out, groups); }, s))
    RUN("global, nt store", 0, 0, true, blocks);
    RUN("global nt, nt store  (the product's form)", 1, 0, true, blocks);
    RUN("buffer aux 0, nt store", 2, 0, true, blocks);
    RUN("buffer sc0, nt store", 2, 1, true, blocks);
    RUN("buffer nt, nt store", 2, 2, true, blocks);
    RUN("buffer sc0 nt, nt store", 2, 3, true, blocks);
    RUN("buffer sc1, nt store", 2, 16, true, blocks);
    RUN("buffer sc0 sc1, nt store", 2, 17, true, blocks);
    RUN("buffer sc1 nt, nt store", 2, 18, true, blocks);
    RUN("buffer sc0 sc1 nt, nt store", 2, 19, true, blocks);
    RUN("global nt, plain store", 1, 0, false, blocks);
    RUN("buffer nt, plain store", 2, 2, false, blocks);
    RUN("global nt, nt store, 2 groups per block", 1, 0, true, (blocks + 1) / 2);
    RUN("global nt, nt store, 3 groups per block", 1, 0, true, (blocks + 2) / 3);
    RUN("global nt, nt store, 8192 resident blocks", 1, 0, true, 8192);
    RUN("global nt, nt store, 4096 resident blocks", 1, 0, true, 4096);
    RUN("buffer sc1 nt, nt store, 8192 resident", 2, 18, true, 8192);
    RUN("global nt, nt store  (again)", 1, 0, true, blocks);
    // the same stream with arithmetic on the loaded values: sat_rect_verts_kernel<4, 64> issues 232 VALU instructions per pair = 928 per lane
    RUNW("  + 128 VALU instructions per lane", 1);
    RUNW("  + 256", 2);
    RUNW("  + 512", 4);
    RUNW("  + 768", 6);
    RUNW("  + 896  (the product kernel: 928)", 7);
    RUNW("  + 1024", 8);
    RUNW("  + 1536", 12);
    RUNW("  + 2048", 16);
    RUN("global nt, nt store  (once more)", 1, 0, true, blocks);
    CHECK(hipGetLastError());
    CHECK(hipFree(planes));
    CHECK(hipFree(out));
    // ---- the polygon kernels' shape (sat_poly_binned_kernel: 64 pairs per wave, 4-byte loads of about 38 rows of 256 bytes)
    {
        constexpr int kMaxRows = 64;
        // (the raw-buffer form keeps the whole batch below 2 GiB from a tile's base: rows x pairs x 4 bytes)
        const size_t np = n > 8000000 ? 8000000 : n;
        float* rows = nullptr;
        uint8_t* o8 = nullptr;
        CHECK(hipMalloc(&rows, (size_t)kMaxRows * np * sizeof(float)));
        CHECK(hipMalloc(&o8, np));
        CHECK(hipMemset(rows, 0x3c, (size_t)kMaxRows * np * sizeof(float)));
        printf("# polygon shape: %zu pairs, one pair per lane, ROWS 4-byte loads per lane, one result byte\n", np);
        auto report_rows = [&](const char* name, int nrows, double us) {
            const double b = ((double)nrows * 4 + 1) * (double)np;
            printf("%-44s %8.1f us  %7.1f GB/s  %.3f of 8 TB/s\n", name, us, b / us / 1e3, b / us / 1e3 / 8000.0);
            fflush(stdout);
        };
#define RUNR(name, ROWS, TILES, BUF) report_rows(name, ROWS, time_us([&] { hipLaunchKernelGGL((row_stream_kernel<ROWS, TILES, BUF>), dim3((unsigned)((np + 64 * TILES - 1) / (64 * TILES))), dim3(64), 0, s, rows, np, o8); }, s))
        RUNR("16 rows, global nt, 1 tile per wave", 16, 1, false);
        RUNR("38 rows, global nt, 1 tile per wave", 38, 1, false);
        RUNR("38 rows, global nt, 2 tiles per wave", 38, 2, false);
        RUNR("38 rows, buffer nt + soffset, 1 tile", 38, 1, true);
        RUNR("38 rows, buffer nt + soffset, 2 tiles", 38, 2, true);
        RUNR("38 rows, buffer nt + soffset, 4 tiles", 38, 4, true);
        RUNR("64 rows, global nt, 1 tile per wave", 64, 1, false);
        RUNR("64 rows, buffer nt + soffset, 1 tile", 64, 1, true);
        CHECK(hipGetLastError());
        CHECK(hipFree(rows));
        CHECK(hipFree(o8));
    }
    return 0;
}
