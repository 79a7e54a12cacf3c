// c2d_poly_binned.hip — polygon SAT over BINNED batches (include/c2d.h, "binned polygon batches"; BASELINE config 5).
//
// Why bins.  In the padded layout f32[2][16][n] the pairs of a wave have unrelated vertex counts, so every 64-byte
// segment of every vertex row holds a vertex somebody needs: 259 bytes move per pair for 155 bytes of real vertices
// (K ~ U{3..16}), whatever the kernel does (sat_poly_kernel sits at the HBM ceiling on those 259 bytes).  A bin holds
// pairs whose polygons have (about) the same size in a plane layout of its own: the bytes that move are the bytes that
// are vertices, and the bin's row counts are known from the launch table — the vertex rows are requested at once,
// without first waiting for the count bytes as sat_poly_kernel must (one memory round trip per tile instead of two).
//
// One launch covers every bin: block = one wave = one tile of 64 pairs of one bin; a u32 per tile names the bin, whose
// descriptor arrives in scalar registers.  Arithmetic, phases and the early-out are those of sat_poly_kernel
// (c2d_poly.hip): phase 1 tests ONE axis of polygon A per pair in registers — the edge whose normal points best at B —
// phase 2 gives every undecided pair the full evaluation, lane = axis, vertices broadcast from LDS.  New here: the
// lanes of phase 2 are dealt by the bin's axis count (rows_a + rows_b lanes per pair, as many pairs side by side as
// fit the wave), not by a compile-time power of two.
#include <algorithm>
#include <chrono>
#include <cstdlib>
#include <new>
#include <type_traits>
#include <vector>

#include "c2d_internal.hpp"
#include "c2d_math.hpp"
#include "c2d_count.hpp"

namespace c2d {

constexpr int kBinSlots = 12;            // LDS slots for parked (undecided) pairs; a round parks whole trips only (see `slots`)
constexpr int kSlotF2 = 2 * C2D_POLY_KMAX + 2;  // float2 per slot: 32 vertices + 16 bytes, so that the slots start on different banks

// launch-table entry (device memory, read through the scalar cache).  Polygon "A" of the table is the one with MORE
// vertex rows (the host swaps the two polygons of a bin when rows_a < rows_b: SAT is symmetric and phase 1 draws its axis
// from A), which also halves the number of phase-1 instances below.
struct alignas(8) BinDesc {
    const float* ax;
    const float* ay;
    const float* bx;
    const float* by;
    const uint8_t* ka;   // may be NULL: every polygon A of the bin has rows_a vertices
    const uint8_t* kb;
    uint8_t* out;
    uint32_t n;          // pairs
    uint32_t stride;     // elements between vertex rows
    uint32_t tile0;      // first tile of the bin within the batch
    uint16_t rows_a, rows_b;
};
static_assert(sizeof(BinDesc) == 72, "BinDesc layout");

C2D_DEV void binned_minmax(float nx, float ny, float x, float y, float& mn, float& mx)
{
    const float p = nx * x + ny * y;  // unfused (translation unit is -ffp-contract=off), utils.cu:173
    mn = __builtin_fminf(mn, p);
    mx = __builtin_fmaxf(mx, p);
}

// One plane of a bin as a raw buffer: a row is addressed as (lane offset in a VGPR) + (row offset in ONE SGPR), so stepping to
// the next row costs one scalar add for both coordinates (a flat 64-bit row pointer costs two per plane).  Planes are below
// 4 GiB (checked when the table is built).
C2D_DEV __amdgpu_buffer_rsrc_t plane_rsrc(const float* base, uint32_t bytes)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, (int)bytes, 0x00020000);
}
C2D_DEV float plane_load(__amdgpu_buffer_rsrc_t r, uint32_t voff, uint32_t soff)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)voff, (int)soff, 2 /* nt: streamed once */));
}

// ---- phase 1 of one tile, for bins with rows_a in (CA - 4, CA] and rows_b in (CB - 4, CB]: straight-line code over CA + CB
// vertex slots.  Slots at and above a polygon's vertex count hold vertex 0, which is exactly neutral (a zero-length edge
// never separates, a repeated projection changes no extreme), so only the LOADS of the last three rows are guarded.
// Returns "separated by the one axis tested"; the vertices stay in ax .. by for the parking step of phase 2.
template <int CA, int CB>
C2D_DEV bool binned_phase1(const BinDesc& D, uint32_t p0, uint32_t cl, bool& bad, int& ka, int& kb, float (&ax)[C2D_POLY_KMAX],
                           float (&ay)[C2D_POLY_KMAX], float (&bx)[C2D_POLY_KMAX], float (&by)[C2D_POLY_KMAX], uint32_t* __restrict__ async_err)
{
    const int rows_a = D.rows_a, rows_b = D.rows_b;
    const uint32_t row_bytes = D.stride * 4u;
    const uint32_t voff = cl * 4u;
    {
        const __amdgpu_buffer_rsrc_t rx = plane_rsrc(D.ax, (uint32_t)rows_a * row_bytes), ry = plane_rsrc(D.ay, (uint32_t)rows_a * row_bytes);
        uint32_t soff = p0 * 4u;
#pragma unroll
        for (int r = 0; r < CA; r++) {
            if (r <= CA - 4 || r < rows_a) {  // (rows_a > CA - 4: the first CA - 3 rows exist)
                ax[r] = plane_load(rx, voff, soff);
                ay[r] = plane_load(ry, voff, soff);
            } else {
                ax[r] = 0.0f;
                ay[r] = 0.0f;
            }
            soff += row_bytes;
        }
    }
    {
        const __amdgpu_buffer_rsrc_t rx = plane_rsrc(D.bx, (uint32_t)rows_b * row_bytes), ry = plane_rsrc(D.by, (uint32_t)rows_b * row_bytes);
        uint32_t soff = p0 * 4u;
#pragma unroll
        for (int r = 0; r < CB; r++) {
            if (r <= CB - 4 || r < rows_b) {
                bx[r] = plane_load(rx, voff, soff);
                by[r] = plane_load(ry, voff, soff);
            } else {
                bx[r] = 0.0f;
                by[r] = 0.0f;
            }
            soff += row_bytes;
        }
    }
    ka = rows_a;  // (per lane; wave-uniform in a bin without count arrays)
    kb = rows_b;
    bad = false;
    if (D.ka != nullptr) {
        ka = D.ka[p0 + cl];
        kb = D.kb[p0 + cl];
        bad = ka < 1 || ka > rows_a || kb < 1 || kb > rows_b;
        ka = ka < 1 ? 1 : (ka > rows_a ? rows_a : ka);
        kb = kb < 1 ? 1 : (kb > rows_b ? rows_b : kb);
        if (__ballot(bad) != 0 && threadIdx.x == 0) __hip_atomic_fetch_or(async_err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    // ---- neutral padding + vertex sums (for the direction between the vertex means)
    float sax = ax[0], say = ay[0], sbx = bx[0], sby = by[0];
#pragma unroll
    for (int r = 1; r < CA; r++) {
        const bool u = r < ka;
        ax[r] = u ? ax[r] : ax[0];
        ay[r] = u ? ay[r] : ay[0];
        sax += ax[r];
        say += ay[r];
    }
#pragma unroll
    for (int r = 1; r < CB; r++) {
        const bool u = r < kb;
        bx[r] = u ? bx[r] : bx[0];
        by[r] = u ? by[r] : by[0];
        sbx += bx[r];
        sby += by[r];
    }
    // mean of the real vertices: the sums hold (C - k) extra copies of vertex 0
    const float ia = __builtin_amdgcn_rcpf((float)ka), ib = __builtin_amdgcn_rcpf((float)kb);
    float dX = (sbx - (float)(CB - kb) * bx[0]) * ib - (sax - (float)(CA - ka) * ax[0]) * ia;
    float dY = (sby - (float)(CB - kb) * by[0]) * ib - (say - (float)(CA - ka) * ay[0]) * ia;
    {   // orientation of A: clockwise polygons have inward-pointing (-ey, ex), so the preferred direction flips
        const float c = (ax[1] - ax[0]) * (ay[2] - ay[0]) - (ay[1] - ay[0]) * (ax[2] - ax[0]);
        const uint32_t sgn = __float_as_uint(c) & 0x80000000u;
        dX = __uint_as_float(__float_as_uint(dX) ^ sgn);
        dY = __uint_as_float(__float_as_uint(dY) ^ sgn);
    }
    // ---- A's edge whose normal points best towards B: largest n.d / |n|, compared as t / l2 with t = (n.d) |n.d| and
    // cross-multiplied (no reciprocal square root: a quarter-rate instruction per edge).  A heuristic in fast arithmetic —
    // it only chooses WHICH canonical axis is evaluated; a zero-length edge gives 0 > -0 = false and is never chosen.
    float bt = -1e30f, bl = 1.0f, nx1 = 0.0f, ny1 = 0.0f;
#pragma unroll
    for (int r = 0; r < CA; r++) {
        const int r1 = (r + 1) % CA;    // (the slot after the last vertex holds vertex 0: the closing edge)
        const float nx = -(ay[r1] - ay[r]);   // true normal (-ey, ex), exactly as phase 2 and the oracle
        const float ny = ax[r1] - ax[r];
        const float nd = fma_(nx, dX, ny * dY);
        const float l2 = fma_(nx, nx, ny * ny);
        const float t = nd * __builtin_fabsf(nd);
        const bool better = t * bl > bt * l2;
        bt = better ? t : bt;
        bl = better ? l2 : bl;
        nx1 = better ? nx : nx1;
        ny1 = better ? ny : ny1;
    }
    float mnA = __builtin_inff(), mxA = -__builtin_inff(), mnB = __builtin_inff(), mxB = -__builtin_inff();
#pragma unroll
    for (int r = 0; r < CA; r++) binned_minmax(nx1, ny1, ax[r], ay[r], mnA, mxA);
#pragma unroll
    for (int r = 0; r < CB; r++) binned_minmax(nx1, ny1, bx[r], by[r], mnB, mxB);
    // (a NaN first projection keeps an axis from separating: first_projections_ordered, c2d_math.hpp)
    return ((mxA < mnB) || (mxB < mnA)) && first_projections_ordered(nx1 * ax[0] + ny1 * ay[0], nx1 * bx[0] + ny1 * by[0]);
}

// parking of one pair's vertices in an LDS slot: A's CA slots, then B's CB (both even), two vertices per ds_write_b128
template <int CA, int CB>
C2D_DEV void binned_park(float2* slot, const float (&ax)[C2D_POLY_KMAX], const float (&ay)[C2D_POLY_KMAX], const float (&bx)[C2D_POLY_KMAX],
                         const float (&by)[C2D_POLY_KMAX])
{
    float4* S4 = reinterpret_cast<float4*>(slot);
#pragma unroll
    for (int r = 0; r < CA / 2; r++) S4[r] = make_float4(ax[2 * r], ay[2 * r], ax[2 * r + 1], ay[2 * r + 1]);
#pragma unroll
    for (int r = 0; r < CB / 2; r++) S4[CA / 2 + r] = make_float4(bx[2 * r], by[2 * r], bx[2 * r + 1], by[2 * r + 1]);
}

// the phase-1 / parking instance of a bin: rows rounded up to 4, 8, 12, 16 on either side, A the larger
#define C2D_BIN_VARIANTS(X) X(4, 4) X(8, 4) X(8, 8) X(12, 4) X(12, 8) X(12, 12) X(16, 4) X(16, 8) X(16, 12) X(16, 16)
C2D_DEV int bin_variant(int rows_a, int rows_b) { return ((rows_a + 3) >> 2) * 4 + ((rows_b + 3) >> 2); }   // ca4 * 4 + cb4, ca4 >= cb4 >= 1

// One tile (64 pairs) of one bin.  SYM: only the instances with equal row classes on both sides are compiled (the padded
// layouts of c2d_sat_poly_pairs_rows are one bin with rows_a == rows_b).
template <bool SYM>
C2D_DEV uint32_t binned_tile(const BinDesc& D, uint32_t tile_in_bin, uint32_t* __restrict__ async_err)   // -> colliding pairs of the tile (wave-uniform)
{
    constexpr int KM = C2D_POLY_KMAX;
    __shared__ __attribute__((aligned(16))) float2 s_slot[kBinSlots][kSlotF2];
    const uint32_t lane = threadIdx.x;
    const int rows_a = D.rows_a, rows_b = D.rows_b;
    const uint32_t p0 = tile_in_bin * 64u;
    const uint32_t here = (D.n - p0) < 64u ? (D.n - p0) : 64u;  // wave-uniform
    const bool in = lane < here;
    const uint32_t cl = in ? lane : here - 1;  // lanes past the end re-read the last pair (never stored)
    const int variant = bin_variant(rows_a, rows_b);
    // ---- phase 1: one axis of A per pair, everything in registers ------------------------------------------------------
    float ax[KM], ay[KM], bx[KM], by[KM];
    bool sep = false, bad = false;
    int ka = 0, kb = 0;
    const bool counted = D.ka != nullptr;  // wave-uniform
    switch (variant) {
#define C2D_BIN_CASE(CA, CB) case (CA / 4) * 4 + CB / 4: if constexpr (!SYM || CA == CB) sep = binned_phase1<CA, CB>(D, p0, cl, bad, ka, kb, ax, ay, bx, by, async_err); break;
        C2D_BIN_VARIANTS(C2D_BIN_CASE)
#undef C2D_BIN_CASE
    default: break;
    }
    sep = sep || bad;  // out-of-range vertex count: reported, result 0
    // ---- phase 2: full evaluation of the pairs that are still undecided -------------------------------------------------
    // Up to kBinSlots undecided lanes park their vertices in LDS; then PP pairs are evaluated side by side, LP = rows_a +
    // rows_b lanes each: lane a of a pair's lanes owns axis a (A's edges first) and projects every vertex of both
    // polygons, two per broadcast ds_read_b128; "some axis separates" is one slice of a ballot.
    unsigned long long todo = __ballot(in && !sep);
    if (todo) {
        const int ca = ((rows_a + 3) >> 2) * 4;                        // float2 of A in a slot (B's follow)
        const int ra2 = (rows_a + 1) >> 1, rb2 = (rows_b + 1) >> 1;   // vertex pairs (float4) of A / of B that hold real vertices
        const int LP = rows_a + rows_b;
        const int pp_fit = 64 / LP;
        const int PP = pp_fit < 8 ? pp_fit : 8;                        // pairs per trip: >= 2 because LP <= 32
        // pairs parked per round: whole trips only (8 parked pairs at 3 per trip left the third trip a third empty: dense
        // scenes 0.957 -> 0.893 ms with 12); 8 where that already is a multiple of the pairs per trip
        const int slots = (8 % PP) == 0 ? 8 : (kBinSlots / PP) * PP;
        uint32_t lane2 = lane;
        asm volatile("" : "+v"(lane2));  // phase-2 lane roles are derived here, not hoisted into phase 1's register peak
        const uint32_t magic = (65536u + (uint32_t)LP - 1u) / (uint32_t)LP;  // lane / LP for lane < 64 (exact: LP <= 32)
        const int sub = (int)((lane2 * magic) >> 16);
        const int a = (int)lane2 - sub * LP;
        // the lane's edge: vertex indices within a slot (A at 0, B at ca)
        const bool edge_of_a = a < rows_a;
        const int eb = edge_of_a ? 0 : ca, ei = edge_of_a ? a : a - rows_a, en = edge_of_a ? rows_a : rows_b;
        const int i0 = eb + ei, i1 = eb + (ei + 1 == en ? 0 : ei + 1);
        const unsigned long long pair_mask = (1ull << LP) - 1ull;
        while (todo) {
            const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(todo >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)todo, 0u));
            const bool park = ((todo >> lane) & 1ull) && rank < (uint32_t)slots;
            __syncthreads();  // single-wave block: a wave-level fence (no s_barrier is emitted); earlier reads are done
            if (park) {
                float2* slot = &s_slot[rank][0];
                switch (variant) {
#define C2D_BIN_CASE(CA, CB) case (CA / 4) * 4 + CB / 4: if constexpr (!SYM || CA == CB) binned_park<CA, CB>(slot, ax, ay, bx, by); break;
                    C2D_BIN_VARIANTS(C2D_BIN_CASE)
#undef C2D_BIN_CASE
                default: break;
                }
            }
            __syncthreads();
            const int left = __popcll(todo);
            const int g = left < slots ? left : slots;
            for (int i = 0; i < g; i += PP) {
                const int cnt = (g - i) < PP ? (g - i) : PP;  // pairs of this trip (wave-uniform)
                const float2* S = &s_slot[i + (sub < cnt ? sub : 0)][0];  // lanes without a pair of their own repeat the first
                const float4* SA = reinterpret_cast<const float4*>(S);
                const float4* SB = reinterpret_cast<const float4*>(S + ca);
                const float2 e0 = S[i0], e1 = S[i1];
                const float nx = -(e1.y - e0.y), ny = e1.x - e0.x;
                float mn1 = __builtin_inff(), mx1 = -__builtin_inff(), mn2 = __builtin_inff(), mx2 = -__builtin_inff();
                // slots past a polygon's count repeat its vertex 0, so the scans need no masking
                // In a bin that holds several sizes the scans stop at the largest count among the trip's pairs (slots above a
                // polygon's own count repeat its vertex 0, so a longer scan is harmless).  (A one-read-ahead form of these loops
                // was measured: 5 more VGPRs, 0.264 -> 0.279 ms on the bench bins, nothing gained on dense scenes.)
                int na2 = ra2, nb2 = rb2;
                if (counted && LP > 16) {  // (small bins: the bound costs more than the few iterations it saves)
                    unsigned long long t2 = todo;
                    int kA = 0, kB = 0;
                    for (int q = 0; q < cnt; q++) {
                        const int j = __ffsll((long long)t2) - 1;
                        t2 &= t2 - 1;
                        const int kaq = __builtin_amdgcn_readlane(ka, j), kbq = __builtin_amdgcn_readlane(kb, j);
                        kA = kaq > kA ? kaq : kA;
                        kB = kbq > kB ? kbq : kB;
                    }
                    na2 = (kA + 1) >> 1;
                    nb2 = (kB + 1) >> 1;
                }
                {   // four vertices per trip of the loop: two LDS reads in flight, half the loop overhead (dense scenes 1.00 -> 0.955 ms)
                    int r2 = 0;
                    for (; r2 + 1 < na2; r2 += 2) {
                        const float4 qa = SA[r2], qb = SA[r2 + 1];
                        binned_minmax(nx, ny, qa.x, qa.y, mn1, mx1);
                        binned_minmax(nx, ny, qa.z, qa.w, mn1, mx1);
                        binned_minmax(nx, ny, qb.x, qb.y, mn1, mx1);
                        binned_minmax(nx, ny, qb.z, qb.w, mn1, mx1);
                    }
                    if (r2 < na2) {
                        const float4 q4 = SA[r2];
                        binned_minmax(nx, ny, q4.x, q4.y, mn1, mx1);
                        binned_minmax(nx, ny, q4.z, q4.w, mn1, mx1);
                    }
                    r2 = 0;
                    for (; r2 + 1 < nb2; r2 += 2) {
                        const float4 qa = SB[r2], qb = SB[r2 + 1];
                        binned_minmax(nx, ny, qa.x, qa.y, mn2, mx2);
                        binned_minmax(nx, ny, qa.z, qa.w, mn2, mx2);
                        binned_minmax(nx, ny, qb.x, qb.y, mn2, mx2);
                        binned_minmax(nx, ny, qb.z, qb.w, mn2, mx2);
                    }
                    if (r2 < nb2) {
                        const float4 q4 = SB[r2];
                        binned_minmax(nx, ny, q4.x, q4.y, mn2, mx2);
                        binned_minmax(nx, ny, q4.z, q4.w, mn2, mx2);
                    }
                }
                const float pa0 = nx * S[0].x + ny * S[0].y, pb0 = nx * S[ca].x + ny * S[ca].y;  // first projections
                const unsigned long long bal = (__builtin_amdgcn_ballot_w64(mx1 < mn2) | __builtin_amdgcn_ballot_w64(mx2 < mn1)) &
                                               __builtin_amdgcn_ballot_w64(first_projections_ordered(pa0, pb0));
                for (int q = 0; q < cnt; q++) {
                    const int j = __ffsll((long long)todo) - 1;
                    todo &= todo - 1;
                    const bool any = ((bal >> (q * LP)) & pair_mask) != 0ull;
                    sep = ((int)lane == j) ? any : sep;
                }
            }
        }
    }
    const bool collide = in && !sep;
    if (in) D.out[p0 + lane] = collide ? (uint8_t)1 : (uint8_t)0;
    return (uint32_t)__popcll(__ballot(collide));
}

// every bin of a batch in one launch: a u32 per tile names the bin
// A wave takes kTilesPerWave consecutive tiles and arrives at the count once: with one tile per wave the returning atomic of the
// count — a round trip to the memory side at the end of a life of a few microseconds — cost 5 % of the launch on the config-5 bins
// and 14 % on bins of triangles (tests/tools/binned_bench.py: 0.279 against 0.265 ms, 0.144 against 0.126 ms).  Two tiles per
// wave, interleaved A/B on one box, 1e7 pairs with the count: config-5 bins 0.269 -> 0.258 ms, K = 3 / 8 / 12 / 16 only 0.156 ->
// 0.144 / 0.229 -> 0.205 / 0.326 -> 0.310 / 0.429 -> 0.421 ms (faster even without the count: the second tile's loads start while
// the first one's stores drain); four tiles are no better (0.256-0.270 / 0.141 / 0.208 / 0.325 / 0.412-0.426), eight are worse.
#ifndef C2D_POLY_TILES_PER_WAVE
#define C2D_POLY_TILES_PER_WAVE 2
#endif
constexpr uint32_t kTilesPerWave = C2D_POLY_TILES_PER_WAVE;
__global__ __launch_bounds__(64, 5) void sat_poly_binned_kernel(const BinDesc* __restrict__ bins, const uint32_t* __restrict__ tile_bin,
                                                               uint32_t tile_begin, uint32_t tile_end, unsigned long long* __restrict__ d_count,
                                                               CountWs words, uint32_t* __restrict__ async_err)
{
    uint32_t total = 0;
#pragma nounroll
    for (uint32_t i = 0; i < kTilesPerWave; i++) {
        const uint32_t tile = tile_begin + blockIdx.x * kTilesPerWave + i;
        if (tile >= tile_end) break;
        const uint32_t bin = __builtin_amdgcn_readfirstlane(tile_bin[tile]);
        const BinDesc D = bins[bin];
        total += binned_tile<false>(D, tile - D.tile0, async_err);
        if (kTilesPerWave > 1) __syncthreads();  // (single-wave block: a wave-level fence) the next tile reuses the LDS slots
    }
    if (d_count) wave_count_arrive_total2(total, d_count, words);
}

// ONE bin whose descriptor travels in the kernel arguments: the padded layouts of c2d_sat_poly_pairs_rows (no table, no upload)
__global__ __launch_bounds__(64, 5) void sat_poly_onebin_kernel(BinDesc D, uint32_t tile_offset, unsigned long long* __restrict__ d_count,
                                                               CountWs words, uint32_t* __restrict__ async_err)
{
    const uint32_t c = binned_tile<true>(D, blockIdx.x + tile_offset, async_err);
    if (d_count) wave_count_arrive_total2(c, d_count, words);
}

// host entry used by c2d_poly.hip: rows_a == rows_b == rows, planes rows * n apart
int launch_poly_onebin(c2d_ctx* ctx, hipStream_t s, const float* d_vx, const float* d_vy, const uint8_t* d_k, size_t n, int rows, uint8_t* d_out,
                       unsigned long long* d_count, uint32_t* async_err)
{
    if (n > 0xffffffffull || (uint64_t)rows * n * 4 > 0xffffffffull) return C2D_ERR_UNSUPPORTED;
    BinDesc D;
    D.ax = d_vx; D.ay = d_vy; D.bx = d_vx + (size_t)rows * n; D.by = d_vy + (size_t)rows * n;
    D.ka = d_k; D.kb = d_k + n; D.out = d_out;
    D.n = (uint32_t)n; D.stride = (uint32_t)n; D.tile0 = 0; D.rows_a = (uint16_t)rows; D.rows_b = (uint16_t)rows;
    const size_t tiles = (n + 63) / 64;
    for (size_t t0 = 0; t0 < tiles; t0 += (size_t)kMaxGrid) {
        const size_t grid = std::min(tiles - t0, (size_t)kMaxGrid);
        hipLaunchKernelGGL(sat_poly_onebin_kernel, dim3((unsigned)grid), dim3(64), 0, s, D, (uint32_t)t0, d_count,
                           workspace_count_ticket2(ctx, s, grid, d_count != nullptr), async_err);
    }
    return C2D_OK;
}

// ---- binning a padded batch -------------------------------------------------------------------------------------------
// class of a pair: (ceil(ka / g) - 1) * 16 + (ceil(kb / g) - 1); counts outside 1..rows go to class 255 = "bad" which is
// reported and whose pairs read 0
C2D_DEV uint32_t bin_class(int ka, int kb, int rows, int g)
{
    if (ka < 1 || ka > rows || kb < 1 || kb > rows) return 0xffffffffu;
    return (uint32_t)(((ka + g - 1) / g - 1) * 16 + ((kb + g - 1) / g - 1));
}

// Binning is a stable counting sort by class in three passes, so that every bin keeps its pairs in input order and the
// result does not depend on scheduling:
//   count  a block takes a TILE of 8192 consecutive pairs (kMoveTile) and counts them per class in an LDS histogram (256 words, LDS
//          atomics), writes the tile's 256 counts and adds them to the sums of its CHUNK of 32 tiles;
//   scan   one block per chunk, one thread per class (coalesced rows): the chunk's base is the sum of the earlier chunks'
//          sums, then the tile counts become exclusive prefixes; the last chunk also leaves the class totals for the host;
//   move   below.
constexpr int kBinBlock = 1024, kBinWaves = kBinBlock / 64;
#ifndef C2D_MOVE_TILE
#define C2D_MOVE_TILE 8192
#endif
constexpr int kMoveTile = C2D_MOVE_TILE, kMoveSub = kMoveTile / kBinBlock;
constexpr int kScanChunk = 32;  // tiles per chunk of the scan

// rank of the lane among the wave's lanes of its class (lanes with ok == false take no part); s_wc[class] gets the wave's count
C2D_DEV uint32_t wave_class_rank(uint32_t c, bool ok, uint16_t* s_wc)
{
    const uint32_t lane = threadIdx.x & 63;
    uint32_t rank = 0;
    unsigned long long rest = __ballot(ok);
    while (rest) {
        const int first = __ffsll((long long)rest) - 1;
        const uint32_t c0 = (uint32_t)__builtin_amdgcn_readlane((int)c, first);
        const unsigned long long m = __ballot(ok && c == c0);
        if (ok && c == c0) rank = (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
        if ((int)lane == first) s_wc[c0] = (uint16_t)__popcll(m);
        rest &= ~m;
    }
    return rank;
}

__global__ __launch_bounds__(kBinBlock) void poly_bin_count_kernel(const uint8_t* __restrict__ k, size_t n, int rows, int g,
                                                                    uint32_t* __restrict__ tile_hist, uint32_t* __restrict__ chunk_sums,
                                                                    uint32_t* __restrict__ n_bad)
{
    __shared__ uint32_t s_h[256];
    if (threadIdx.x < 256) s_h[threadIdx.x] = 0;
    __syncthreads();
    const size_t tile0 = (size_t)blockIdx.x * kMoveTile;
    bool bad = false;
#pragma unroll
    for (int sub = 0; sub < kMoveSub; sub++) {
        const size_t i = tile0 + (size_t)sub * kBinBlock + threadIdx.x;
        if (i < n) {
            const uint32_t c = bin_class(k[i], k[n + i], rows, g);
            if (c != 0xffffffffu) atomicAdd(&s_h[c], 1u);
            else bad = true;
        }
    }
    if (__ballot(bad) != 0 && (threadIdx.x & 63) == 0) atomicAdd(n_bad, 1u);
    __syncthreads();
    if (threadIdx.x < 256) {
        const uint32_t v = s_h[threadIdx.x];
        tile_hist[(size_t)blockIdx.x * 256 + threadIdx.x] = v;
        if (v) atomicAdd(&chunk_sums[(size_t)(blockIdx.x / kScanChunk) * 256 + threadIdx.x], v);
    }
}

// tile counts -> exclusive prefixes over the tiles, per class; totals[c] = pairs of the class
__global__ __launch_bounds__(256) void poly_bin_scan_kernel(uint32_t* __restrict__ tile_hist, uint32_t n_tiles, const uint32_t* __restrict__ chunk_sums,
                                                             uint32_t* __restrict__ totals)
{
    const uint32_t c = threadIdx.x, chunk = blockIdx.x;
    uint32_t run = 0;
    for (uint32_t q = 0; q < chunk; q++) run += chunk_sums[(size_t)q * 256 + c];
    if (chunk == gridDim.x - 1) totals[c] = run + chunk_sums[(size_t)chunk * 256 + c];
    const uint32_t t0 = chunk * kScanChunk, t1 = (t0 + kScanChunk) < n_tiles ? (t0 + kScanChunk) : n_tiles;
    for (uint32_t t = t0; t < t1; t++) {
        const uint32_t v = tile_hist[(size_t)t * 256 + c];
        tile_hist[(size_t)t * 256 + c] = run;
        run += v;
    }
}

struct BinMoveArgs {
    const float* vx;
    const float* vy;
    const uint8_t* k;
    size_t n;
    int rows, g;
    const BinDesc* table;          // one entry per bin, polygons as in the INPUT (not the test kernel's swapped view)
    const uint16_t* class_to_bin;  // [256], 0xffff = empty class
    const uint32_t* pair_base;     // [bins]: first position of the bin in the concatenated order
    const uint32_t* tile_prefix;   // [tiles of kMoveTile pairs][256] from the scan
    const uint32_t* totals;        // [256] pairs per class
    uint32_t n_tiles;
    uint32_t tiles_per_xcd;        // ceil(n_tiles / 8): block b takes tile (b % 8) * tiles_per_xcd + b / 8
    uint32_t* index;               // [n] out: position of input pair i; 0xffffffff = bad counts
    float* base;                   // the block every plane of every bin lives in: a plane position is an element offset from here
#ifdef C2D_MOVE_CHECK  // (developer build: every index of the move kernel is checked against its array; the first offender is recorded and skipped)
    unsigned long long lim_in, lim_k, lim_block, lim_prefix, lim_bins;
    unsigned long long* chk;  // [0] = kind (0 none), [1] = index, [2] = limit, [3] = tile << 32 | thread
#endif
};
#ifdef C2D_MOVE_CHECK
#define C2D_MOVE_OK(kind, idx, lim) move_check(A.chk, kind, (unsigned long long)(idx), (unsigned long long)(lim), tile, t)
C2D_DEV bool move_check(unsigned long long* chk, unsigned kind, unsigned long long idx, unsigned long long lim, uint32_t tile, uint32_t t)
{
    if (idx < lim) return true;
    if (atomicCAS(&chk[0], 0ull, (unsigned long long)kind) == 0ull) { chk[1] = idx; chk[2] = lim; chk[3] = ((unsigned long long)tile << 32) | t; }
    return false;
}
#else
#define C2D_MOVE_OK(kind, idx, lim) true
#endif

// The move: a block owns a TILE of 8192 consecutive pairs (kMoveTile).  Lanes that write one pair each scatter 4-byte stores over as many
// cache lines as there are classes in a wave (196 bins: 7.5 ms per 1e7 pairs), so the tile is first sorted by destination:
// every pair gets its slot (prefix of its tile and class + pairs of the class earlier in the tile + rank in its wave), the tile's
// pairs are ordered by (class, slot) in LDS, and then every vertex row of every plane passes through an LDS stage — read from the
// padded batch as it lies (coalesced), written in destination order, where neighbouring lanes hold neighbouring slots of one class.
//
// Round 4 (the form of round 3 moved 4.14 GB in 1.355 ms = 0.38 of the HBM peak, profiles/r03h_*; with the loads alone or the
// stores alone the first form of this round took 0.73 and 0.71 ms, profiles/notes_r04_bin_move.md):
//  * the waves of a block are SPECIALISED: the producer waves only load (16-byte loads, two rows in flight in registers, written
//    to one of two LDS stages), the consumer waves only read the stage and store — fire and forget, they never wait for memory
//    (a wave with loads AND stores in flight can only wait for "everything": one counter counts both); one barrier per row
//    hands a stage over;
//  * a consumer keeps everything about its destination positions in REGISTERS (source index in the tile, class, row counts packed
//    in one word; stride; the running plane offsets of the polygon in hand): a row costs it one 8-byte LDS read and two stores
//    per position instead of nine LDS look-ups;
//  * the stage holds (x, y) side by side, so that the random-bank read of a position is ONE ds_read_b64;
//  * the slots come from an eight-ballot class match (wave_match_class) and one prefix pass over the waves per class, two
//    barriers per 1024 pairs, where round 3 looped over the distinct classes of a wave and summed the earlier waves per lane;
//  * consecutive tiles go to the SAME XCD (block b of a launch runs on XCD b % 8): the run a tile contributes to a bin's row is
//    about 42 floats and ends inside a cache line that the NEXT tile continues — written from the same L2 the two partial lines
//    merge before they leave for HBM (WRITE_SIZE 2.07 -> 1.69 GB for 1.55 GB of vertices);
//  * plane positions are element offsets from the bins' block in 32 bits (Off = uint32_t; a block of 16 GiB or more takes the
//    64-bit instance);
//  * the tile is 8192 pairs, one block per CU (134 KB of LDS): what limits the pass now is the STORES — with the loads switched off they
//    alone take 0.70 ms for 1.6 GB (2.3 TB/s), folded into a 32 MB window that L2 absorbs 0.35 ms.  csrc/tools/store_pattern_probe.hip
//    shows what they are slow at: a run of whole 128-byte lines streams at 5-6 TB/s however many bins there are, the same run at an
//    arbitrary 4-byte offset costs 2-3 times as much whatever the layout or the order — the 64-byte sectors at the two ends of a run are
//    shared with the neighbouring tiles' runs, which other blocks write tens of microseconds earlier or later, and a tile contributes
//    42 +- 6 pairs to a bin's row.  Hence the longer run per (tile, bin, row) is worth more than the second block per CU that a
//    4096-pair tile allows (same box: 1.27 -> 1.15 ms; whole call 1.70 -> 1.38 ms).  The form that removes the shared sectors — a ring
//    per bin in LDS that carries every write front's unfinished line from tile to tile, whole lines stored — was built and measured:
//    correct, 1.29 ms at best against 1.03-1.11 ms for this one; its 640 small steps per block cost more than the stores save
//    (profiles/notes_r04_bin_move.md, profiles/r04_ring_move_kernel.patch).
typedef float v4f_u __attribute__((ext_vector_type(4), aligned(4)));  // four floats at any 4-byte boundary (a row of the padded batch starts anywhere)
#ifndef C2D_MOVE_PRODUCER_WAVES
#define C2D_MOVE_PRODUCER_WAVES 8
#endif
constexpr int kProducerWaves = C2D_MOVE_PRODUCER_WAVES, kConsumerThreads = kBinBlock - 64 * kProducerWaves;
constexpr int kMovePos = (kMoveTile + kConsumerThreads - 1) / kConsumerThreads;  // destination positions per consumer thread
constexpr int kMoveVec = kMoveTile / 4 / (64 * kProducerWaves);                  // 16-byte loads per producer thread, row and plane
static_assert(kMoveVec * 4 * 64 * kProducerWaves == kMoveTile, "the producers' loads tile the row");

// the lanes of the wave (with ok) that hold the same 8-bit class as this lane: eight ballots, no loop over classes
C2D_DEV unsigned long long wave_match_class(uint32_t c, bool ok)
{
    unsigned long long peers = __ballot(ok);
#pragma unroll
    for (int b = 0; b < 8; b++) {
        const bool bit = ((c >> b) & 1u) != 0;
        const unsigned long long m = __ballot(bit);
        peers &= bit ? m : ~m;
    }
    return peers;
}

template <typename Off>
__global__ __launch_bounds__(kBinBlock, (kMoveTile <= 4096 ? 2 : 1) * kBinWaves / 4) void poly_bin_move_kernel(BinMoveArgs A)
{
    // two stages, each one vertex row of the tile as (x, y) pairs.  Until the rows start they hold the tables of the slot
    // computation: per-wave class counts and prefixes in stage 0, s_slot / s_cls / s_sorted (32 KB) in stage 1
    __shared__ __attribute__((aligned(16))) float2 s_stage[2][kMoveTile];
    uint16_t (*s_wcnt)[256] = reinterpret_cast<uint16_t (*)[256]>(&s_stage[0][0]);               // [wave][class] pairs of the class in the wave
    uint16_t (*s_wpre)[256] = reinterpret_cast<uint16_t (*)[256]>(&s_stage[0][0]) + kBinWaves;   // [wave][class] ... in the tile before the wave
    static_assert(2 * sizeof(uint16_t) * kBinWaves * 256 <= sizeof(float2) * kMoveTile, "the wave tables fit stage 0");
    uint32_t* s_slot = reinterpret_cast<uint32_t*>(&s_stage[1][0]);                       // slot of local pair li within its bin
    uint16_t* s_cls = reinterpret_cast<uint16_t*>(&s_stage[1][kMoveTile / 2]);            // its class, 0xffff = bad counts
    uint16_t* s_sorted = reinterpret_cast<uint16_t*>(&s_stage[1][kMoveTile / 2]) + kMoveTile;  // local pair at position sp of the tile's (class, slot) order
    __shared__ uint32_t s_cbase[257];             // first position of class c in that order; [256] = valid pairs of the tile
    __shared__ uint32_t s_first[256];             // slot of the tile's first pair of class c
    __shared__ uint32_t s_run[256];               // pairs of class c in the tile's earlier sub-blocks
    __shared__ Off s_plane[4][256];               // ax, ay, bx, by of the class's bin, as element offsets from A.base
    __shared__ uint32_t s_stride[256];
    __shared__ uint8_t s_rows[2][256];
    const uint32_t t = threadIdx.x, wave = t >> 6, lane = t & 63;
    const uint32_t tile = (blockIdx.x & 7u) * A.tiles_per_xcd + (blockIdx.x >> 3);
    if (tile >= A.n_tiles) return;  // (block-uniform)
    const size_t tile0 = (size_t)tile * kMoveTile;
    // ---- per-class tables of this tile
    if (t < 256) {
        const uint32_t c = t;
        const uint32_t bin = A.class_to_bin[c];
        uint32_t cnt = 0, first = 0;
        if (bin != 0xffffu && C2D_MOVE_OK(1, bin, A.lim_bins) && C2D_MOVE_OK(2, (size_t)tile * 256 + c + (tile + 1 < A.n_tiles ? 256 : 0), A.lim_prefix)) {
            const BinDesc D = A.table[bin];
            s_plane[0][c] = (Off)(D.ax - A.base); s_plane[1][c] = (Off)(D.ay - A.base);
            s_plane[2][c] = (Off)(D.bx - A.base); s_plane[3][c] = (Off)(D.by - A.base);
            s_stride[c] = D.stride;
            s_rows[0][c] = (uint8_t)D.rows_a; s_rows[1][c] = (uint8_t)D.rows_b;
            first = A.tile_prefix[(size_t)tile * 256 + c];
            const uint32_t end = tile + 1 < A.n_tiles ? A.tile_prefix[(size_t)(tile + 1) * 256 + c] : A.totals[c];
            cnt = end - first;
        } else {
            s_rows[0][c] = s_rows[1][c] = 0;
        }
        s_first[c] = first;
        s_run[c] = 0;
        s_cbase[c + 1] = cnt;  // counts for now
    }
    for (int i = t; i < kBinWaves * 256; i += kBinBlock) (&s_wcnt[0][0])[i] = 0;
    __syncthreads();
    if (t < 64) {  // counts -> first positions (s_cbase[c + 1] holds the count of class c): one wave, four classes per lane
        uint32_t v[4], sum = 0;
#pragma unroll
        for (int q = 0; q < 4; q++) { v[q] = s_cbase[4 * t + q + 1]; sum += v[q]; }
        uint32_t incl = sum;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t o = (uint32_t)__shfl_up((int)incl, off, 64);
            if (t >= (uint32_t)off) incl += o;
        }
        uint32_t run = incl - sum;  // classes before this lane's four
        if (t == 0) s_cbase[0] = 0;
#pragma unroll
        for (int q = 0; q < 4; q++) { run += v[q]; s_cbase[4 * t + q + 1] = run; }
    }
    // ---- slots, index, and the tile's (class, slot) order: 1024 pairs at a time
    for (int sub = 0; sub < kMoveSub; sub++) {
        const uint32_t li = sub * kBinBlock + t;
        const size_t i = tile0 + li;
        const bool in = i < A.n;
        int ka = 0, kb = 0;
        uint32_t c = 0xffffffffu;
        if (in && C2D_MOVE_OK(3, A.n + i, A.lim_k)) {
            ka = A.k[i];
            kb = A.k[A.n + i];
            c = bin_class(ka, kb, A.rows, A.g);
        }
        const bool ok = c != 0xffffffffu;
        const unsigned long long peers = wave_match_class(c, ok);
        const uint32_t rank = (uint32_t)__popcll(peers & ((1ull << lane) - 1ull));
        if (ok && rank == 0) s_wcnt[wave][c] = (uint16_t)__popcll(peers);  // the class's first lane in the wave
        __syncthreads();  // (also orders the s_cbase scan above before its first use)
        if (t < 256) {  // per class: counts of the waves -> pairs of the class in the tile before each wave; the counts are cleared for the next round
            uint32_t run = s_run[t];
#pragma unroll
            for (int w = 0; w < kBinWaves; w++) {
                const uint32_t v = s_wcnt[w][t];
                s_wcnt[w][t] = 0;
                s_wpre[w][t] = (uint16_t)run;
                run += v;
            }
            s_run[t] = run;
        }
        __syncthreads();
        s_cls[li] = ok ? (uint16_t)c : (uint16_t)0xffff;
        if (in) {
            uint32_t pos = 0xffffffffu;
            if (ok) {
                const uint32_t in_tile = (uint32_t)s_wpre[wave][c] + rank;  // pairs of the class earlier in the tile
                const uint32_t slot = s_first[c] + in_tile;
                s_slot[li] = slot;
                if (C2D_MOVE_OK(4, s_cbase[c] + in_tile, kMoveTile)) s_sorted[s_cbase[c] + in_tile] = (uint16_t)li;
                const uint32_t bin = A.class_to_bin[c];
                (void)C2D_MOVE_OK(5, bin, A.lim_bins);
                pos = A.pair_base[bin] + slot;
                const BinDesc D = A.table[bin];
                if (D.ka) {  // (two bytes per pair: written by the pair's own lane)
                    const_cast<uint8_t*>(D.ka)[slot] = (uint8_t)ka;
                    const_cast<uint8_t*>(D.kb)[slot] = (uint8_t)kb;
                }
            }
            A.index[i] = pos;
        }
    }
    __syncthreads();
    const uint32_t n_valid = s_cbase[256];
    // pairs of this tile.  In 32 bits on purpose (n < 2^32 - 64, checked by the host): written as a 64-bit compare and select, hipcc 7.2
    // compiled the partial-tile path below with here = kMoveTile — v_cmp_lt_u64 into vcc, s_cbranch_vccz, then an s_cselect_b32 on an
    // SCC that an unrelated s_add_i32 had set — so the last tile's loads were not guarded and ran up to 8191 floats past the end of
    // vx / vy (read into registers nobody uses: results unaffected, but a fault when the batch ends at the end of a mapping;
    // profiles/notes_r05_move_kernel_overread.md).  tests/test_gpu_poly_binned.py runs the index-checked build over this.
    const uint32_t left = (uint32_t)A.n - (uint32_t)tile0;
    const uint32_t here = left < (uint32_t)kMoveTile ? left : (uint32_t)kMoveTile;
    const int n_rows = 2 * A.rows;  // rows of polygon A, then of polygon B
    if (wave >= (uint32_t)kProducerWaves) {
        // ---- consumer: this thread's destination positions, in registers.  meta = li | class << 16 | rows_a << 24 | rows_b << 28
        // (row counts 1..16 stored as 0..15; a position without a pair has stride 0 and never stores: see `live`)
        const uint32_t tc = t - 64 * kProducerWaves;
        uint32_t d_meta[kMovePos], d_stride[kMovePos], d_slot[kMovePos];
        unsigned live = 0;  // bit j: position j holds a pair
#pragma unroll
        for (int j = 0; j < kMovePos; j++) {
            const uint32_t sp = (uint32_t)j * kConsumerThreads + tc;
            d_meta[j] = 0; d_stride[j] = 0; d_slot[j] = 0;
            if (sp < n_valid && C2D_MOVE_OK(6, sp, kMoveTile) && C2D_MOVE_OK(7, s_sorted[sp], kMoveTile)) {
                const uint32_t li = s_sorted[sp];
                const uint32_t c = s_cls[li];
                d_slot[j] = s_slot[li];
                d_stride[j] = s_stride[c];
                d_meta[j] = li | (c << 16) | ((uint32_t)(s_rows[0][c] - 1) << 24) | ((uint32_t)(s_rows[1][c] - 1) << 28);
                live |= 1u << j;
            }
        }
        __syncthreads();  // (0) the tables in the stages have been read: the producers may fill them
#pragma unroll
        for (int poly = 0; poly < 2; poly++) {
            Off d_x[kMovePos], d_y[kMovePos];
#pragma unroll
            for (int j = 0; j < kMovePos; j++) {
                const uint32_t c = (d_meta[j] >> 16) & 0xffu;
                d_x[j] = s_plane[2 * poly][c] + d_slot[j];
                d_y[j] = s_plane[2 * poly + 1][c] + d_slot[j];
            }
#pragma nounroll
            for (int r = 0; r < A.rows; r++) {
                const int it = poly * A.rows + r;
                __syncthreads();  // (it + 1) stage it & 1 holds row it
                const float2* sv = &s_stage[it & 1][0];
#pragma unroll
                for (int j = 0; j < kMovePos; j++) {
                    const uint32_t rows_here = ((d_meta[j] >> (24 + 4 * poly)) & 0xfu) + 1u;
#ifdef C2D_MOVE_NO_STORE  // (experiment: how long do the loads alone take)
                    const float2 v = sv[d_meta[j] & 0xffffu];
                    if (((live >> j) & 1u) && (uint32_t)r < rows_here && v.x == 1.2345e-30f && v.y == 5.4321e-30f) A.base[d_x[j]] = 0.0f;
#else
                    if (((live >> j) & 1u) && (uint32_t)r < rows_here && C2D_MOVE_OK(8, d_x[j], A.lim_block) && C2D_MOVE_OK(9, d_y[j], A.lim_block)) {
                        const float2 v = sv[d_meta[j] & 0xffffu];
#if defined(C2D_MOVE_STORE_WINDOW)      // (experiment: the same stores folded into a 32 MB window: is it DRAM or the way there?)
                        A.base[d_x[j] & 0x7fffffu] = v.x;
                        A.base[d_y[j] & 0x7fffffu] = v.y;
#elif defined(C2D_MOVE_NT_STORE)
                        __builtin_nontemporal_store(v.x, A.base + d_x[j]);
                        __builtin_nontemporal_store(v.y, A.base + d_y[j]);
#else
                        A.base[d_x[j]] = v.x;
                        A.base[d_y[j]] = v.y;
#endif
                    }
#endif
                    d_x[j] += d_stride[j];
                    d_y[j] += d_stride[j];
                }
            }
        }
        return;
    }
    // ---- producer: rows it, it + 1 in flight in registers; row it goes to stage it & 1 once it has arrived.  The loop exists
    // twice: for a whole tile (plain 16-byte loads, nothing to decide per lane) and for the batch's last, partial tile
    auto produce = [&](auto full_const) {
        constexpr bool FULL = decltype(full_const)::value;
        float4 bx_[2][kMoveVec], by_[2][kMoveVec];
        auto fetch = [&](int it, float4 (&fx)[kMoveVec], float4 (&fy)[kMoveVec]) {
            const size_t row = (size_t)it * A.n + tile0;   // (it = poly * rows + r: the planes are [2][rows][n])
#pragma unroll
            for (int e = 0; e < kMoveVec; e++) {
                const uint32_t li = 4u * ((uint32_t)e * 64 * kProducerWaves + t);
#ifdef C2D_MOVE_NO_LOAD  // (experiment: how long do the stores alone take)
                if (true) {
                    fx[e] = make_float4((float)li, 1.0f, 2.0f, (float)it);
                    fy[e] = make_float4((float)li, 3.0f, 4.0f, (float)it);
                } else
#endif
                if constexpr (FULL) {
                    if (!C2D_MOVE_OK(10, row + li + 3, A.lim_in)) { fx[e] = fy[e] = make_float4(0, 0, 0, 0); continue; }
                    const v4f_u lx = __builtin_nontemporal_load(reinterpret_cast<const v4f_u*>(A.vx + row + li));
                    const v4f_u ly = __builtin_nontemporal_load(reinterpret_cast<const v4f_u*>(A.vy + row + li));
                    fx[e] = make_float4(lx.x, lx.y, lx.z, lx.w);
                    fy[e] = make_float4(ly.x, ly.y, ly.z, ly.w);
                } else {
                    float ax[4], ay[4];
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        const bool inside = li + q < here && C2D_MOVE_OK(11, row + li + q, A.lim_in);
                        ax[q] = inside ? A.vx[row + li + q] : 0.0f;
                        ay[q] = inside ? A.vy[row + li + q] : 0.0f;
                    }
                    fx[e] = make_float4(ax[0], ax[1], ax[2], ax[3]);
                    fy[e] = make_float4(ay[0], ay[1], ay[2], ay[3]);
                }
            }
        };
        auto put = [&](int it, const float4 (&fx)[kMoveVec], const float4 (&fy)[kMoveVec]) {
            float4* sv = reinterpret_cast<float4*>(&s_stage[it & 1][0]);  // two (x, y) pairs per float4
#pragma unroll
            for (int e = 0; e < kMoveVec; e++) {
                const uint32_t li = 4u * ((uint32_t)e * 64 * kProducerWaves + t);
                sv[li / 2] = make_float4(fx[e].x, fy[e].x, fx[e].y, fy[e].y);
                sv[li / 2 + 1] = make_float4(fx[e].z, fy[e].z, fx[e].w, fy[e].w);
            }
        };
        // n_rows is even (two polygons).  Every step issues its fetch — past the end it re-reads the last row into registers nobody
        // reads — so that the number of loads in flight at each wait is the same on every path and the wait for row `it` can
        // leave row it + 1's loads outstanding (a conditional fetch makes the compiler wait for everything)
        const int last = n_rows - 1;
        fetch(0, bx_[0], by_[0]);
        fetch(1, bx_[1], by_[1]);
        __syncthreads();  // (0)
#pragma nounroll
        for (int it = 0; it < n_rows; it += 2) {
            put(it, bx_[0], by_[0]);
            fetch(it + 2 < last ? it + 2 : last, bx_[0], by_[0]);
            __syncthreads();  // (it + 1)
            put(it + 1, bx_[1], by_[1]);
            fetch(it + 3 < last ? it + 3 : last, bx_[1], by_[1]);
            __syncthreads();  // (it + 2)
        }
    };
    if (here == (uint32_t)kMoveTile) produce(std::true_type{});
    else produce(std::false_type{});
}

__global__ __launch_bounds__(256) void poly_bins_results_kernel(const uint8_t* __restrict__ out_all, const uint32_t* __restrict__ index, size_t n,
                                                               uint8_t* __restrict__ out)
{
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
        const uint32_t p = index[i];
        out[i] = p == 0xffffffffu ? (uint8_t)0 : out_all[p];
    }
}

}  // namespace c2d

using namespace c2d;

struct c2d_poly_bins {
    int device = 0;
    std::vector<c2d_poly_bin> bins;   // as given (strides resolved)
    std::vector<uint32_t> strides;
    BinDesc* d_table = nullptr;
    uint32_t* d_tile_bin = nullptr;
    size_t n_tiles = 0, pairs = 0, bytes = 0;
    bool table_in_block = false;      // d_table / d_tile_bin live inside d_block (c2d_poly_bins_from_padded): nothing to free
    // handles made by c2d_poly_bins_from_padded own their data
    void* d_block = nullptr;
    uint8_t* d_out_all = nullptr;     // inside d_block: every bin's results, in bin order
    uint32_t* d_index = nullptr;      // inside d_block
    size_t n_input = 0;
    bool had_bad_counts = false;
};

namespace {

// the launch table as the test kernel wants it: polygon A of an entry is the one with more rows
// in_block: where the table and the tile list go (device memory inside the handle's own block, table first, tile list at
// in_block + tile_list_offset; see table_bytes below), or nullptr = two allocations of their own (c2d_poly_bins_create);
// s: the stream of the uploads (synchronised before this returns: the host vectors are locals)
size_t table_bytes(size_t n_bins) { return (n_bins * sizeof(BinDesc) + 255) / 256 * 256; }

int upload_table(c2d_ctx* ctx, c2d_poly_bins* B, char* in_block = nullptr, hipStream_t s = nullptr)
{
    std::vector<BinDesc> table(B->bins.size());
    std::vector<uint32_t> tile_bin;
    size_t tiles = 0;
    B->pairs = 0;
    B->bytes = 0;
    for (size_t b = 0; b < B->bins.size(); b++) {
        const c2d_poly_bin& s = B->bins[b];
        BinDesc& d = table[b];
        d.ax = s.d_ax; d.ay = s.d_ay; d.bx = s.d_bx; d.by = s.d_by; d.ka = s.d_ka; d.kb = s.d_kb; d.out = s.d_out;
        d.n = (uint32_t)s.n;
        d.stride = B->strides[b];
        d.tile0 = (uint32_t)tiles;
        d.rows_a = (uint16_t)s.rows_a;
        d.rows_b = (uint16_t)s.rows_b;
        if (d.rows_a < d.rows_b) {  // SAT is symmetric: the kernel takes its phase-1 axis from the polygon with more edges
            std::swap(d.ax, d.bx);
            std::swap(d.ay, d.by);
            std::swap(d.ka, d.kb);
            std::swap(d.rows_a, d.rows_b);
        }
        if ((uint64_t)d.rows_a * d.stride * 4 > 0xffffffffull) return fail_arg(ctx, "c2d_poly_bins: a vertex plane of a bin exceeds 4 GiB (split the bin)");
        const size_t t = (s.n + 63) / 64;
        tiles += t;
        if (tiles > 0xffffffffull) return fail_arg(ctx, "c2d_poly_bins: more than 2^32 tiles in one batch");
        tile_bin.insert(tile_bin.end(), t, (uint32_t)b);
        B->pairs += s.n;
        B->bytes += s.n * ((size_t)(s.rows_a + s.rows_b) * 8 + (s.d_ka ? 2 : 0) + 1);
    }
    B->n_tiles = tiles;
    if (in_block) {
        B->table_in_block = true;
        B->d_table = reinterpret_cast<BinDesc*>(in_block);
        B->d_tile_bin = reinterpret_cast<uint32_t*>(in_block + table_bytes(table.size()));
        if (!table.empty()) C2D_HIP(ctx, hipMemcpyAsync(B->d_table, table.data(), table.size() * sizeof(BinDesc), hipMemcpyHostToDevice, s));
        if (tiles) C2D_HIP(ctx, hipMemcpyAsync(B->d_tile_bin, tile_bin.data(), tiles * sizeof(uint32_t), hipMemcpyHostToDevice, s));
        C2D_HIP(ctx, hipStreamSynchronize(s));
        return C2D_OK;
    }
    if (!table.empty()) {
        C2D_HIP(ctx, hipMalloc(&B->d_table, table.size() * sizeof(BinDesc)));
        C2D_HIP(ctx, hipMemcpy(B->d_table, table.data(), table.size() * sizeof(BinDesc), hipMemcpyHostToDevice));
    }
    if (tiles) {
        C2D_HIP(ctx, hipMalloc(&B->d_tile_bin, tiles * sizeof(uint32_t)));
        C2D_HIP(ctx, hipMemcpy(B->d_tile_bin, tile_bin.data(), tiles * sizeof(uint32_t), hipMemcpyHostToDevice));
    }
    return C2D_OK;
}

void release(c2d_poly_bins* B)
{
    if (!B) return;
    if (B->d_table && !B->table_in_block) (void)hipFree(B->d_table);
    if (B->d_tile_bin && !B->table_in_block) (void)hipFree(B->d_tile_bin);
    if (B->d_block) (void)hipFree(B->d_block);
    delete B;
}

}  // namespace

extern "C" {

int c2d_poly_bins_create(c2d_ctx* ctx, const c2d_poly_bin* bins, size_t n_bins, c2d_poly_bins** out)
{
    if (!ctx || !out) return C2D_ERR_INVALID_ARG;
    *out = nullptr;
    if (n_bins && !bins) return fail_arg(ctx, "c2d_poly_bins_create: NULL bins");
    if (n_bins > 65535) return fail_arg(ctx, "c2d_poly_bins_create: more than 65535 bins");
    c2d_poly_bins* B = new (std::nothrow) c2d_poly_bins();
    if (!B) return C2D_ERR_NOMEM;
    B->device = ctx->device;
    for (size_t b = 0; b < n_bins; b++) {
        const c2d_poly_bin& s = bins[b];
        const char* why = nullptr;
        if (s.rows_a < 1 || s.rows_a > C2D_POLY_KMAX || s.rows_b < 1 || s.rows_b > C2D_POLY_KMAX) why = "c2d_poly_bins_create: rows_a / rows_b must be 1..C2D_POLY_KMAX";
        else if (s.n > 0xffffffffull) why = "c2d_poly_bins_create: a bin holds at most 2^32 - 1 pairs";
        else if (s.stride != 0 && s.stride < s.n) why = "c2d_poly_bins_create: stride must be 0 or >= n";
        else if (s.stride > 0xffffffffull) why = "c2d_poly_bins_create: stride must fit 32 bits";
        else if (s.n && (!s.d_ax || !s.d_ay || !s.d_bx || !s.d_by || !s.d_out)) why = "c2d_poly_bins_create: NULL plane or result pointer";
        else if ((s.d_ka == nullptr) != (s.d_kb == nullptr)) why = "c2d_poly_bins_create: d_ka and d_kb must both be given or both be NULL";
        if (why) { release(B); return fail_arg(ctx, why); }
        B->bins.push_back(s);  // (an empty bin stays in the list — c2d_poly_bins_get(i) is the caller's bin i — and takes no tiles)
        B->strides.push_back((uint32_t)(s.stride ? s.stride : s.n));
    }
    DeviceGuard g(ctx->device);
    const int st = upload_table(ctx, B);
    if (st != C2D_OK) { release(B); return st; }
    *out = B;
    return C2D_OK;
}

int c2d_poly_bins_destroy(c2d_ctx* ctx, c2d_poly_bins* bins)
{
    if (!bins) return C2D_OK;
    DeviceGuard g(ctx ? ctx->device : bins->device);
    release(bins);
    return C2D_OK;
}

size_t c2d_poly_bins_size(const c2d_poly_bins* bins) { return bins ? bins->bins.size() : 0; }
size_t c2d_poly_bins_pairs(const c2d_poly_bins* bins) { return bins ? bins->pairs : 0; }
size_t c2d_poly_bins_bytes(const c2d_poly_bins* bins) { return bins ? bins->bytes : 0; }

int c2d_poly_bins_get(const c2d_poly_bins* bins, size_t i, c2d_poly_bin* out)
{
    if (!bins || !out || i >= bins->bins.size()) return C2D_ERR_INVALID_ARG;
    *out = bins->bins[i];
    out->stride = bins->strides[i];
    return C2D_OK;
}

int c2d_sat_poly_pairs_binned(c2d_ctx* ctx, const c2d_poly_bins* bins, unsigned long long* d_count, c2d_stream stream)
{
    if (!ctx) return C2D_ERR_INVALID_ARG;
    if (!bins) return fail_arg(ctx, "c2d_sat_poly_pairs_binned: NULL bins");
    if (bins->device != ctx->device) return fail_arg(ctx, "c2d_sat_poly_pairs_binned: the bins belong to another device");
    if (bins->n_tiles == 0) return C2D_OK;
    DeviceGuard g(ctx->device);
    hipStream_t s = (hipStream_t)stream;
    if (int rc = workspace_acquire(ctx, s, d_count != nullptr)) return rc;
    const size_t per_launch = (size_t)kMaxGrid * kTilesPerWave;
    for (size_t t0 = 0; t0 < bins->n_tiles; t0 += per_launch) {
        const size_t t1 = std::min(bins->n_tiles, t0 + per_launch);
        const size_t grid = (t1 - t0 + kTilesPerWave - 1) / kTilesPerWave;
        hipLaunchKernelGGL(sat_poly_binned_kernel, dim3((unsigned)grid), dim3(64), 0, s, bins->d_table, bins->d_tile_bin, (uint32_t)t0, (uint32_t)t1, d_count,
                           workspace_count_ticket2(ctx, s, grid, d_count != nullptr), ctx->d_async_err);
    }
    C2D_LAUNCH_CHECK(ctx);
    return C2D_OK;
}

int c2d_poly_bins_from_padded(c2d_ctx* ctx, const float* d_vx, const float* d_vy, const uint8_t* d_k, size_t n, int rows, int granularity,
                              c2d_poly_bins** out, c2d_stream stream)
{
    // developer aid: C2D_TRACE_BINS=1 prints the wall time of each phase to stderr (synchronising after each: not for timing the whole)
    static const bool trace = getenv("C2D_TRACE_BINS") != nullptr;
    auto t_last = std::chrono::steady_clock::now();
    auto lap = [&](const char* what) {
        if (!trace) return;
        (void)hipDeviceSynchronize();
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[c2d bins] %-28s %8.1f us\n", what, std::chrono::duration<double, std::micro>(now - t_last).count());
        t_last = std::chrono::steady_clock::now();
    };
    if (!ctx || !out) return C2D_ERR_INVALID_ARG;
    *out = nullptr;
    if (rows < 1 || rows > C2D_POLY_KMAX) return fail_arg(ctx, "c2d_poly_bins_from_padded: rows must be 1..C2D_POLY_KMAX");
    if (granularity < 1 || granularity > C2D_POLY_KMAX) return fail_arg(ctx, "c2d_poly_bins_from_padded: granularity must be 1..C2D_POLY_KMAX");
    if (n > 0xffffffffull - 64) return fail_arg(ctx, "c2d_poly_bins_from_padded: at most 2^32 - 65 pairs");
    if (n && (!d_vx || !d_vy || !d_k)) return fail_arg(ctx, "c2d_poly_bins_from_padded: NULL argument");
    DeviceGuard g(ctx->device);
    hipStream_t s = (hipStream_t)stream;
    c2d_poly_bins* B = new (std::nothrow) c2d_poly_bins();
    if (!B) return C2D_ERR_NOMEM;
    B->device = ctx->device;
    B->n_input = n;
    if (n == 0) { *out = B; return C2D_OK; }
    // the pass uses the ctx scratch and ends synchronised; a failure after its first enqueue drains the stream before it
    // returns, so nothing of the call stays in flight on the scratch (no ticket needed: c2d_internal.hpp)
    bool enqueued = false;
    auto fail = [&](int st) {
        if (enqueued) { (void)hipStreamSynchronize(s); (void)hipGetLastError(); }
        release(B);
        return st;
    };
#define C2D_BIN_HIP(call)                                                                        \
    do {                                                                                         \
        hipError_t e__ = (call);                                                                 \
        if (e__ != hipSuccess) return fail(c2d::fail_hip(ctx, e__, #call, __FILE__, __LINE__));  \
    } while (0)
    // ---- 1. pairs per (tile, class), prefixes over the tiles, class totals
    const size_t n_tiles = (n + kMoveTile - 1) / kMoveTile;
    const size_t n_chunks = (n_tiles + kScanChunk - 1) / kScanChunk;
    // scratch of the pass, from the ctx (grown on demand, kept between calls; this call is ordered on `s` and ends synchronised):
    // [n_tiles][256] counts -> prefixes | [256] totals + 1 word of bad-count waves | [n_chunks][256] chunk sums | the move kernel's table
    const size_t hist_words = n_tiles * 256 + 257 + n_chunks * 256;
    const size_t off_plain = (hist_words * sizeof(uint32_t) + 255) / 256 * 256;
    const size_t scratch_need = off_plain + 256 * sizeof(BinDesc);
    if (int rc = workspace_acquire(ctx, s, true)) return fail(rc);
    if (ctx->scratch_bytes < scratch_need) {
        if (ctx->d_scratch) (void)hipFree(ctx->d_scratch);
        ctx->d_scratch = nullptr;
        ctx->scratch_bytes = 0;
        C2D_BIN_HIP(hipMalloc(&ctx->d_scratch, scratch_need));
        ctx->scratch_bytes = scratch_need;
    }
    uint32_t* d_hist = static_cast<uint32_t*>(ctx->d_scratch);
    uint32_t* d_totals = d_hist + n_tiles * 256;
    uint32_t* d_chunk_sums = d_totals + 257;
    uint32_t hist[257];
    enqueued = true;
    C2D_BIN_HIP(hipMemsetAsync(d_totals, 0, (257 + n_chunks * 256) * sizeof(uint32_t), s));
    hipLaunchKernelGGL(poly_bin_count_kernel, dim3((unsigned)n_tiles), dim3(kBinBlock), 0, s, d_k, n, rows, granularity, d_hist, d_chunk_sums, d_totals + 256);
    hipLaunchKernelGGL(poly_bin_scan_kernel, dim3((unsigned)n_chunks), dim3(256), 0, s, d_hist, (uint32_t)n_tiles, d_chunk_sums, d_totals);
    C2D_BIN_HIP(hipMemcpyAsync(hist, d_totals, sizeof hist, hipMemcpyDeviceToHost, s));
    C2D_BIN_HIP(hipStreamSynchronize(s));
    lap("count + scan + read-back");
    B->had_bad_counts = hist[256] != 0;
    // ---- 2. layout of the block: per bin ax, ay, bx, by (stride = n rounded up to 64 elements, every plane 256-byte
    // aligned), counts (only when a bin can hold different sizes); then every bin's results back to back; then the index
    const int g_ = granularity;
    const bool counted = g_ > 1;
    std::vector<uint16_t> class_to_bin(256, 0xffff);
    std::vector<uint32_t> pair_base;
    size_t bytes = 0, pairs = 0;
    auto align_up = [](size_t v, size_t a) { return (v + a - 1) / a * a; };
    struct Plan { size_t ax, ay, bx, by, ka, kb; uint32_t stride; };
    std::vector<Plan> plan;
    for (int c = 0; c < 256; c++) {
        if (!hist[c]) continue;
        c2d_poly_bin b{};
        b.rows_a = (uint32_t)std::min(rows, (c / 16 + 1) * g_);
        b.rows_b = (uint32_t)std::min(rows, (c % 16 + 1) * g_);
        b.n = hist[c];
        const uint32_t stride = (uint32_t)align_up(b.n, 64);
        Plan p{};
        p.stride = stride;
        const size_t plane_a = (size_t)b.rows_a * stride * 4, plane_b = (size_t)b.rows_b * stride * 4;
        p.ax = bytes; bytes = align_up(bytes + plane_a, 256);
        p.ay = bytes; bytes = align_up(bytes + plane_a, 256);
        p.bx = bytes; bytes = align_up(bytes + plane_b, 256);
        p.by = bytes; bytes = align_up(bytes + plane_b, 256);
        if (counted) {
            p.ka = bytes; bytes = align_up(bytes + b.n, 256);
            p.kb = bytes; bytes = align_up(bytes + b.n, 256);
        }
        class_to_bin[c] = (uint16_t)B->bins.size();
        pair_base.push_back((uint32_t)pairs);
        pairs += b.n;
        B->bins.push_back(b);
        B->strides.push_back(stride);
        plan.push_back(p);
    }
    const size_t off_out = bytes;
    bytes = align_up(bytes + pairs, 256);
    const size_t off_index = bytes;
    bytes += n * sizeof(uint32_t);
    const size_t n_bins = B->bins.size();
    const size_t off_base = align_up(bytes, 256);
    bytes = off_base + (n_bins + 1) * sizeof(uint32_t);
    const size_t off_c2b = align_up(bytes, 256);
    bytes = off_c2b + 256 * sizeof(uint16_t);
    // the test kernel's launch table and tile list (upload_table) live in the block too: one allocation per handle
    size_t test_tiles = 0;
    for (const c2d_poly_bin& b : B->bins) test_tiles += (b.n + 63) / 64;
    const size_t off_table = align_up(bytes, 256);
    bytes = off_table + table_bytes(n_bins) + test_tiles * sizeof(uint32_t);
    lap("host layout");
    hipError_t me = hipMalloc(&B->d_block, bytes);
    lap("hipMalloc of the block");
    if (me == hipErrorOutOfMemory) { ctx->last_error = "c2d_poly_bins_from_padded: out of device memory"; return fail(C2D_ERR_NOMEM); }
    if (me != hipSuccess) return fail(c2d::fail_hip(ctx, me, "hipMalloc", __FILE__, __LINE__));
    char* base = static_cast<char*>(B->d_block);
    B->d_out_all = reinterpret_cast<uint8_t*>(base + off_out);
    B->d_index = reinterpret_cast<uint32_t*>(base + off_index);
    for (size_t b = 0; b < n_bins; b++) {
        c2d_poly_bin& d = B->bins[b];
        const Plan& p = plan[b];
        d.d_ax = reinterpret_cast<const float*>(base + p.ax);
        d.d_ay = reinterpret_cast<const float*>(base + p.ay);
        d.d_bx = reinterpret_cast<const float*>(base + p.bx);
        d.d_by = reinterpret_cast<const float*>(base + p.by);
        d.d_ka = counted ? reinterpret_cast<const uint8_t*>(base + p.ka) : nullptr;
        d.d_kb = counted ? reinterpret_cast<const uint8_t*>(base + p.kb) : nullptr;
        d.d_out = B->d_out_all + pair_base[b];
        d.stride = p.stride;
    }
    int st = upload_table(ctx, B, base + off_table, s);
    if (st != C2D_OK) return fail(st);
    lap("launch table upload");
    // the move kernel's view of the bins: polygon A = the input's polygon A
    std::vector<BinDesc> plain(n_bins);
    for (size_t b = 0; b < n_bins; b++) {
        const c2d_poly_bin& d = B->bins[b];
        BinDesc& t = plain[b];
        t.ax = d.d_ax; t.ay = d.d_ay; t.bx = d.d_bx; t.by = d.d_by; t.ka = d.d_ka; t.kb = d.d_kb; t.out = d.d_out;
        t.n = (uint32_t)d.n; t.stride = (uint32_t)d.stride; t.tile0 = 0; t.rows_a = (uint16_t)d.rows_a; t.rows_b = (uint16_t)d.rows_b;
    }
    BinDesc* d_plain = reinterpret_cast<BinDesc*>(static_cast<char*>(ctx->d_scratch) + off_plain);  // (at most 256 bins: one per class)
    if (n_bins) C2D_BIN_HIP(hipMemcpyAsync(d_plain, plain.data(), n_bins * sizeof(BinDesc), hipMemcpyHostToDevice, s));
    // ---- 3. move the vertices
    C2D_BIN_HIP(hipMemsetAsync(B->d_out_all, 0, pairs, s));
    if (n_bins) C2D_BIN_HIP(hipMemcpyAsync(base + off_base, pair_base.data(), n_bins * sizeof(uint32_t), hipMemcpyHostToDevice, s));
    C2D_BIN_HIP(hipMemcpyAsync(base + off_c2b, class_to_bin.data(), 256 * sizeof(uint16_t), hipMemcpyHostToDevice, s));
    BinMoveArgs A;
    A.vx = d_vx; A.vy = d_vy; A.k = d_k; A.n = n; A.rows = rows; A.g = g_;
    A.table = d_plain;
    A.class_to_bin = reinterpret_cast<const uint16_t*>(base + off_c2b);
    A.pair_base = reinterpret_cast<const uint32_t*>(base + off_base);
    A.tile_prefix = d_hist;
    A.totals = d_totals;
    A.n_tiles = (uint32_t)n_tiles;
    A.tiles_per_xcd = (uint32_t)((n_tiles + 7) / 8);
    A.index = B->d_index;
    lap("small copies + memset");
    A.base = reinterpret_cast<float*>(base);
#ifdef C2D_MOVE_CHECK
    A.lim_in = (unsigned long long)2 * rows * n; A.lim_k = 2ull * n; A.lim_block = bytes / 4; A.lim_prefix = n_tiles * 256 + 256; A.lim_bins = n_bins;
    unsigned long long* d_chk = nullptr;
    C2D_BIN_HIP(hipMalloc(&d_chk, 32));
    C2D_BIN_HIP(hipMemset(d_chk, 0, 32));
    A.chk = d_chk;
#endif
    if (bytes < (16ull << 30)) hipLaunchKernelGGL(poly_bin_move_kernel<uint32_t>, dim3(A.tiles_per_xcd * 8u), dim3(kBinBlock), 0, s, A);  // (n < 2^32)
    else hipLaunchKernelGGL(poly_bin_move_kernel<uint64_t>, dim3(A.tiles_per_xcd * 8u), dim3(kBinBlock), 0, s, A);
    C2D_BIN_HIP(hipStreamSynchronize(s));  // (the host vectors above must outlive their copies)
    workspace_stream_drained(ctx, s);
#ifdef C2D_MOVE_CHECK
    {
        unsigned long long chk[4];
        C2D_BIN_HIP(hipMemcpy(chk, d_chk, 32, hipMemcpyDeviceToHost));
        (void)hipFree(d_chk);
        if (chk[0]) fprintf(stderr, "[c2d move check] access %llu: index %llu, limit %llu (tile %llu, thread %llu; n %zu rows %d g %d bins %zu)\n", chk[0], chk[1], chk[2],
                            chk[3] >> 32, chk[3] & 0xffffffffull, n, rows, g_, n_bins);
    }
#endif
    lap("move kernel");
#undef C2D_BIN_HIP
    if (B->had_bad_counts) {
        release(B);
        return fail_arg(ctx, "c2d_poly_bins_from_padded: polygon vertex count outside 1..rows");
    }
    *out = B;
    return C2D_OK;
}

int c2d_poly_bins_results(c2d_ctx* ctx, const c2d_poly_bins* bins, uint8_t* d_out, c2d_stream stream)
{
    if (!ctx) return C2D_ERR_INVALID_ARG;
    if (!bins || !bins->d_index) return fail_arg(ctx, "c2d_poly_bins_results: the handle was not made by c2d_poly_bins_from_padded");
    if (bins->n_input == 0) return C2D_OK;
    if (!d_out) return fail_arg(ctx, "c2d_poly_bins_results: NULL output");
    DeviceGuard g(ctx->device);
    hipLaunchKernelGGL(poly_bins_results_kernel, dim3(grid_for(bins->n_input, 256, ctx->prop.multiProcessorCount * 16)), dim3(256), 0, (hipStream_t)stream,
                       bins->d_out_all, bins->d_index, bins->n_input, d_out);
    C2D_LAUNCH_CHECK(ctx);
    return C2D_OK;
}

}  // extern "C"
