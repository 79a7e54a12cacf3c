// store_pattern_probe — what the write side of a counting sort can reach on one MI355X, as a function of how its stores are laid out
// in memory and in time.  Developer tool behind profiles/notes_r04_bin_move.md (the binning pass of c2d_poly_binned.hip); no input is
// read, every variant writes the same 1.5 GB: n = 1e7 "pairs" x RP = 38 row planes of 4 bytes, partitioned into B bins.
//
// A block owns one tile of P consecutive pairs (tiles of an XCD contiguous, as poly_bin_move_kernel assigns them); the tile holds
// c = P / B pairs of every bin, already in bin order, so the run a (tile, bin, plane) contributes is c * 4 bytes.
//   layout  SoA     bin b = RP planes of S slots each (what sat_poly_binned_kernel reads): a run lands at plane * S * 4 + slot * 4
//           AoSoA   bin b = chunks of 64 slots, a chunk = RP rows of 256 B: the RP runs of a (tile, bin) fall into one or two chunks
//   order   plane-outer  the block writes plane 0 of all bins, then plane 1 ... (the move kernel: one row of the tile is staged at a time)
//           bin-outer    the block writes all planes of bin 0, then bin 1 ...   (only possible with the whole tile on chip)
//   nt      the stores carry the non-temporal hint
// Usage: store_pattern_probe            (prints one line per variant: microseconds per pass, GB/s)
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <utility>
#include <vector>

#define CHECK(x)                                                                                  \
    do {                                                                                          \
        hipError_t e__ = (x);                                                                     \
        if (e__ != hipSuccess) {                                                                  \
            fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e__));   \
            exit(1);                                                                              \
        }                                                                                         \
    } while (0)

struct Shape {
    int T, c, B, RP, tiles_per_xcd;
    int segs;           // plane-outer only: a block owns `segs` consecutive tiles and writes plane p of all of them before plane p + 1 (0 or 1: one tile)
    int plain_map;      // 1: tile = blockIdx.x (neighbouring tiles on different XCDs); 0: the tiles of an XCD are contiguous
    unsigned magic_c;   // ceil(2^32 / c): (s * magic) >> 32 == s / c for the ranges used here (checked on the host)
    long long S64;      // slots of a bin, rounded up to whole chunks of 64
};

template <bool AOSOA>
__device__ __forceinline__ long long element_of(const Shape& sh, int b, int plane, long long slot)
{
    const long long bin_base = (long long)b * sh.RP * sh.S64;
    if (AOSOA) return bin_base + (slot >> 6) * ((long long)sh.RP * 64) + (long long)plane * 64 + (slot & 63);
    return bin_base + (long long)plane * sh.S64 + slot;
}

template <bool AOSOA, bool BIN_OUTER, bool NT>
__global__ __launch_bounds__(512) void scatter_kernel(float* __restrict__ out, Shape sh)
{
    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    const int segs = sh.segs > 1 ? sh.segs : 1;
    const int t_first = (sh.plain_map ? (int)blockIdx.x : xcd * sh.tiles_per_xcd + idx) * segs;   // (tiles_per_xcd counts ranges when segs > 1)
    if (idx >= sh.tiles_per_xcd || t_first >= sh.T) return;
    auto put = [&](long long e, float v) {
        if (NT) __builtin_nontemporal_store(v, out + e);
        else out[e] = v;
    };
    if (!BIN_OUTER && segs > 1) {
        const int per_plane = sh.B * sh.c;
        for (int plane = 0; plane < sh.RP; plane++)
            for (int g = 0; g < segs && t_first + g < sh.T; g++) {
                const long long slot0 = (long long)(t_first + g) * sh.c;
                for (int s = threadIdx.x; s < per_plane; s += 512) {
                    const int b = (int)(((unsigned long long)(unsigned)s * sh.magic_c) >> 32);
                    const int i = s - b * sh.c;
                    put(element_of<AOSOA>(sh, b, plane, slot0 + i), (float)s);
                }
            }
        return;
    }
    const int t = t_first;
    const long long slot0 = (long long)t * sh.c;
    if (!BIN_OUTER) {
        const int per_plane = sh.B * sh.c;
        for (int plane = 0; plane < sh.RP; plane++)
            for (int s = threadIdx.x; s < per_plane; s += 512) {
                const int b = (int)(((unsigned long long)(unsigned)s * sh.magic_c) >> 32);
                const int i = s - b * sh.c;
                put(element_of<AOSOA>(sh, b, plane, slot0 + i), (float)s);
            }
    } else {
        const int per_bin = sh.RP * sh.c;
        for (int b = 0; b < sh.B; b++)
            for (int e = threadIdx.x; e < per_bin; e += 512) {
                const int plane = (int)(((unsigned long long)(unsigned)e * sh.magic_c) >> 32);
                const int i = e - plane * sh.c;
                put(element_of<AOSOA>(sh, b, plane, slot0 + i), (float)e);
            }
    }
}

// the store pattern of a sort that stages through a per-bin ring in LDS and only ever stores whole 64-byte sectors: a block owns a range
// of `segs` tiles; for each plane and each tile, a group of 16 lanes takes bin b = group, group + 32, ... and stores the sectors of that
// bin's row which the tile completes (the tile adds c slots to the bin; sector s is complete once slot 16 s + 15 has arrived)
__global__ __launch_bounds__(512) void ring_kernel(float* __restrict__ out, Shape sh)
{
    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    const int t_first = (xcd * sh.tiles_per_xcd + idx) * sh.segs;
    if (idx >= sh.tiles_per_xcd || t_first >= sh.T) return;
    const int group = threadIdx.x >> 4, k = threadIdx.x & 15;
    for (int plane = 0; plane < sh.RP; plane++)
        for (int g = 0; g < sh.segs && t_first + g < sh.T; g++) {
            const long long f0 = (long long)(t_first + g) * sh.c, f1 = f0 + sh.c;   // the tile's slots of every bin
            // (the first tile of a range also owns the sector its first slot lies in, the last one the sector its last slot lies in:
            //  range boundaries are the only shared sectors, and they are few)
            const long long s0 = f0 >> 4, s1 = (g == sh.segs - 1 || t_first + g == sh.T - 1) ? ((f1 + 15) >> 4) : (f1 >> 4);
            for (int b = group; b < sh.B; b += 32)
                for (long long sec = s0; sec < s1; sec++) {
                    const long long slot = sec * 16 + k;
                    if (slot < sh.S64) out[element_of<false>(sh, b, plane, slot)] = (float)slot;
                }
        }
}

// unaligned runs as in scatter_kernel (SoA, plane-outer), but a run is never cut between two waves: wave w stores the runs w, w + 8, ...
// with its first c lanes (c <= 64)
__global__ __launch_bounds__(512) void run_per_wave_kernel(float* __restrict__ out, Shape sh)
{
    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    const int t = xcd * sh.tiles_per_xcd + idx;
    if (idx >= sh.tiles_per_xcd || t >= sh.T) return;
    const long long slot0 = (long long)t * sh.c;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int plane = 0; plane < sh.RP; plane++)
        for (int b = wave; b < sh.B; b += 8)
            if (lane < sh.c) out[element_of<false>(sh, b, plane, slot0 + lane)] = (float)lane;
}

// the same bytes as one stream: every thread writes consecutive float4
__global__ __launch_bounds__(512) void stream_kernel(float4* __restrict__ out, long long n4)
{
    for (long long i = (long long)blockIdx.x * 512 + threadIdx.x; i < n4; i += (long long)gridDim.x * 512) out[i] = make_float4(1.f, 2.f, 3.f, 4.f);
}

int main()
{
    const long long N = 10'000'000;
    const int RP = 38;
    const long long cap_elems = N * RP + 64ll * RP * 256;   // every variant stays below this (checked per variant)
    float* d = nullptr;
    CHECK(hipMalloc(&d, (size_t)cap_elems * 4));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    auto timed = [&](auto launch) {
        launch();
        CHECK(hipDeviceSynchronize());
        float best = 1e30f, sum = 0.f;
        const int reps = 5;
        for (int r = 0; r < reps; r++) {
            CHECK(hipEventRecord(e0));
            launch();
            CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1));
            float ms;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            best = ms < best ? ms : best;
            sum += ms;
        }
        CHECK(hipGetLastError());
        return std::pair<float, float>(best, sum / reps);
    };
    {
        const long long n4 = N * RP / 4;
        auto r = timed([&] { hipLaunchKernelGGL(stream_kernel, dim3(256 * 8), dim3(512), 0, 0, (float4*)d, n4); });
        printf("stream float4, %lld bytes: min %.1f us avg %.1f us = %.2f TB/s\n", n4 * 16, r.first * 1e3, r.second * 1e3, n4 * 16 / (r.first * 1e-3) / 1e12);
    }
    printf("%-6s %-6s %-7s %-11s %-3s | run B | bytes      | min us  | avg us  | TB/s\n", "bins", "tile", "layout", "order", "nt");
    const int bins_list[] = {196, 49, 16};
    const int tile_list[] = {8192};
    for (int B : bins_list)
        for (int P : tile_list) {
            Shape sh;
            sh.B = B;
            sh.RP = RP;
            sh.c = P / B;
            sh.T = (int)(N / P);
            sh.tiles_per_xcd = (sh.T + 7) / 8;
            sh.plain_map = 0;
            sh.segs = 0;
            const long long S = (long long)sh.T * sh.c;
            sh.S64 = (S + 63) / 64 * 64;
            sh.magic_c = (unsigned)(((1ull << 32) + sh.c - 1) / sh.c);
            // host checks: the division trick is exact over both index ranges, and the last element written is inside the buffer
            const long long max_index = (long long)(B > RP ? B : RP) * sh.c;
            bool ok = true;
            for (long long s = 0; s < max_index; s++)
                if ((long long)(((unsigned long long)s * sh.magic_c) >> 32) != s / sh.c) { ok = false; break; }
            const long long total_elems = (long long)B * RP * sh.S64;
            if (!ok || total_elems > cap_elems) {
                printf("%-6d %-6d skipped (division check %d, elements %lld of %lld)\n", B, P, (int)ok, total_elems, cap_elems);
                continue;
            }
            const long long bytes = (long long)sh.T * B * sh.c * RP * 4;
            const unsigned grid = 8u * (unsigned)sh.tiles_per_xcd;
            auto row = [&](const char* layout, const char* order, int nt, std::pair<float, float> r) {
                printf("%-6d %-6d %-7s %-11s %-3d | %5d | %10lld | %7.1f | %7.1f | %.2f\n", B, P, layout, order, nt, sh.c * 4, bytes, r.first * 1e3, r.second * 1e3,
                       bytes / (r.first * 1e-3) / 1e12);
                fflush(stdout);
            };
            row("SoA", "plane-outer", 0, timed([&] { hipLaunchKernelGGL((scatter_kernel<false, false, false>), dim3(grid), dim3(512), 0, 0, d, sh); }));
            row("SoA", "bin-outer", 0, timed([&] { hipLaunchKernelGGL((scatter_kernel<false, true, false>), dim3(grid), dim3(512), 0, 0, d, sh); }));
            row("AoSoA", "plane-outer", 0, timed([&] { hipLaunchKernelGGL((scatter_kernel<true, false, false>), dim3(grid), dim3(512), 0, 0, d, sh); }));
            row("AoSoA", "bin-outer", 0, timed([&] { hipLaunchKernelGGL((scatter_kernel<true, true, false>), dim3(grid), dim3(512), 0, 0, d, sh); }));
            row("SoA", "plane-outer", 1, timed([&] { hipLaunchKernelGGL((scatter_kernel<false, false, true>), dim3(grid), dim3(512), 0, 0, d, sh); }));
            row("AoSoA", "bin-outer", 1, timed([&] { hipLaunchKernelGGL((scatter_kernel<true, true, true>), dim3(grid), dim3(512), 0, 0, d, sh); }));
        }
    // second table: bins x pairs of a bin per tile, chosen so that runs are line-aligned (c a multiple of 32) or not, SoA, plane-outer
    printf("\n%-6s %-6s %-9s %-5s | run B | aligned | bytes      | min us  | avg us  | TB/s\n", "bins", "c", "tile", "map");
    const int matrix[][2] = {{196, 32}, {196, 36}, {196, 40}, {196, 41}, {196, 48}, {196, 64}, {196, 128}, {49, 83}, {49, 128}, {49, 256}, {49, 334}, {16, 41}, {16, 64}, {16, 250},
                             {16, 256}, {4, 41}, {4, 256}, {784, 32}, {784, 41}};
    for (auto& m : matrix)
        for (int plain = 0; plain < 2; plain++) {
            Shape sh;
            sh.B = m[0];
            sh.c = m[1];
            sh.RP = RP;
            const int P = sh.B * sh.c;
            sh.T = (int)(N / P);
            sh.tiles_per_xcd = (sh.T + 7) / 8;
            sh.plain_map = plain;
            sh.segs = 0;
            const long long S = (long long)sh.T * sh.c;
            sh.S64 = (S + 63) / 64 * 64;
            sh.magic_c = (unsigned)(((1ull << 32) + sh.c - 1) / sh.c);
            const long long max_index = (long long)(sh.B > RP ? sh.B : RP) * sh.c;
            bool ok = max_index < (1ll << 31);
            for (long long s2 = 0; ok && s2 < max_index; s2++)
                if ((long long)(((unsigned long long)s2 * sh.magic_c) >> 32) != s2 / sh.c) ok = false;
            const long long total_elems = (long long)sh.B * RP * sh.S64;
            if (!ok || total_elems > cap_elems) {
                printf("%-6d %-6d skipped (division check %d, elements %lld of %lld)\n", sh.B, sh.c, (int)ok, total_elems, cap_elems);
                continue;
            }
            const long long bytes = (long long)sh.T * sh.B * sh.c * RP * 4;
            const unsigned grid = 8u * (unsigned)sh.tiles_per_xcd;
            auto r = timed([&] { hipLaunchKernelGGL((scatter_kernel<false, false, false>), dim3(grid), dim3(512), 0, 0, d, sh); });
            printf("%-6d %-6d %-9d %-5s | %5d | %7s | %10lld | %7.1f | %7.1f | %.2f\n", sh.B, sh.c, P, plain ? "plain" : "xcd", sh.c * 4, sh.c % 32 == 0 ? "128 B" : sh.c % 16 == 0 ? "64 B" : sh.c % 8 == 0 ? "32 B" : sh.c % 4 == 0 ? "16 B" : "no", bytes,
                   r.first * 1e3, r.second * 1e3, bytes / (r.first * 1e-3) / 1e12);
            fflush(stdout);
        }
    // third table: a block owns a RANGE of consecutive tiles and writes plane p of all of them before plane p + 1, so that the two halves
    // of a line shared by the runs of neighbouring tiles come from the same CU a few microseconds apart
    printf("\n%-6s %-6s %-6s %-6s | run B | ranges | min us  | avg us  | TB/s\n", "bins", "c", "tile", "segs");
    const int ranges_list[][3] = {{196, 41, 1}, {196, 41, 5}, {196, 21, 10}, {196, 21, 20}, {196, 10, 20}, {196, 41, 2}, {16, 250, 1}, {16, 250, 10}};
    for (auto& m : ranges_list) {
        Shape sh;
        sh.B = m[0];
        sh.c = m[1];
        sh.segs = m[2];
        sh.RP = RP;
        sh.plain_map = 0;
        const int P = sh.B * sh.c;
        sh.T = (int)(N / P);
        const int ranges = (sh.T + sh.segs - 1) / sh.segs;
        sh.tiles_per_xcd = (ranges + 7) / 8;
        const long long S = (long long)sh.T * sh.c;
        sh.S64 = (S + 63) / 64 * 64;
        sh.magic_c = (unsigned)(((1ull << 32) + sh.c - 1) / sh.c);
        const long long max_index = (long long)(sh.B > RP ? sh.B : RP) * sh.c;
        bool ok = true;
        for (long long s2 = 0; ok && s2 < max_index; s2++)
            if ((long long)(((unsigned long long)s2 * sh.magic_c) >> 32) != s2 / sh.c) ok = false;
        const long long total_elems = (long long)sh.B * RP * sh.S64;
        if (!ok || total_elems > cap_elems) {
            printf("%-6d %-6d skipped\n", sh.B, sh.c);
            continue;
        }
        const long long bytes = (long long)sh.T * sh.B * sh.c * RP * 4;
        const unsigned grid = 8u * (unsigned)sh.tiles_per_xcd;
        auto r = timed([&] { hipLaunchKernelGGL((scatter_kernel<false, false, false>), dim3(grid), dim3(512), 0, 0, d, sh); });
        printf("%-6d %-6d %-6d %-6d | %5d | %6d | %7.1f | %7.1f | %.2f\n", sh.B, sh.c, P, sh.segs, sh.c * 4, ranges, r.first * 1e3, r.second * 1e3,
               bytes / (r.first * 1e-3) / 1e12);
        fflush(stdout);
    }
    // a run per wave: does it matter that scatter_kernel cuts runs at wave boundaries?
    {
        Shape sh;
        sh.B = 196; sh.c = 41; sh.RP = RP; sh.segs = 0; sh.plain_map = 0;
        sh.T = (int)(N / (sh.B * sh.c));
        sh.tiles_per_xcd = (sh.T + 7) / 8;
        const long long S = (long long)sh.T * sh.c;
        sh.S64 = (S + 63) / 64 * 64;
        sh.magic_c = (unsigned)(((1ull << 32) + sh.c - 1) / sh.c);
        const long long bytes = (long long)sh.T * sh.B * sh.c * RP * 4;
        const unsigned grid = 8u * (unsigned)sh.tiles_per_xcd;
        auto r0 = timed([&] { hipLaunchKernelGGL((scatter_kernel<false, false, false>), dim3(grid), dim3(512), 0, 0, d, sh); });
        auto r1 = timed([&] { hipLaunchKernelGGL(run_per_wave_kernel, dim3(grid), dim3(512), 0, 0, d, sh); });
        printf("\n196 bins, c = 41: runs cut at wave boundaries %.1f us, one run per wave %.1f us (%.2f / %.2f TB/s)\n", r0.first * 1e3, r1.first * 1e3,
               bytes / (r0.first * 1e-3) / 1e12, bytes / (r1.first * 1e-3) / 1e12);
    }
    // fourth table: whole-sector stores out of a ring (ring_kernel)
    printf("\n%-6s %-6s %-6s %-6s | ranges | min us  | avg us  | TB/s   (ring: 16-lane groups store whole 64-byte sectors)\n", "bins", "c", "tile", "segs");
    const int ring_list[][3] = {{196, 21, 10}, {196, 21, 5}, {196, 41, 5}, {196, 10, 20}, {49, 83, 10}};
    for (auto& m : ring_list) {
        Shape sh;
        sh.B = m[0];
        sh.c = m[1];
        sh.segs = m[2];
        sh.RP = RP;
        sh.plain_map = 0;
        const int P = sh.B * sh.c;
        sh.T = (int)(N / P);
        const int ranges = (sh.T + sh.segs - 1) / sh.segs;
        sh.tiles_per_xcd = (ranges + 7) / 8;
        const long long S = (long long)sh.T * sh.c;
        sh.S64 = (S + 63) / 64 * 64;
        sh.magic_c = 0;
        const long long total_elems = (long long)sh.B * RP * sh.S64;
        if (total_elems > cap_elems) { printf("%-6d %-6d skipped\n", sh.B, sh.c); continue; }
        const long long bytes = (long long)sh.T * sh.B * sh.c * RP * 4;
        const unsigned grid = 8u * (unsigned)sh.tiles_per_xcd;
        auto r = timed([&] { hipLaunchKernelGGL(ring_kernel, dim3(grid), dim3(512), 0, 0, d, sh); });
        printf("%-6d %-6d %-6d %-6d | %6d | %7.1f | %7.1f | %.2f\n", sh.B, sh.c, P, sh.segs, ranges, r.first * 1e3, r.second * 1e3, bytes / (r.first * 1e-3) / 1e12);
        fflush(stdout);
    }
    CHECK(hipFree(d));
    return 0;
}
