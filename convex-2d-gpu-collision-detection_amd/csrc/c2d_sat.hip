// c2d_sat.hip — batched SAT kernels for gfx950 (MI355X).
//
// Hot path of BASELINE config 2: the arithmetic of convex_collide (reference
// utils.cu:159-184) over n independent rectangle pairs held as 16 SoA planes.
// The kernel is HBM-bound: 16 x 4 B read + 1 B written per pair (65 B/pair,
// ~300 VALU ops/pair).  Mapping: one lane owns VEC consecutive pairs, loads
// them with one global_load_dwordx4 per plane (1 KiB per wave instruction,
// 16 independent loads in flight per lane), keeps everything in VGPRs (no LDS:
// a rectangle pair has no reuse across lanes), and writes VEC result bytes
// with one store.  The count of colliding pairs is reduced per wave with
// ballot/popcount and leaves the block as a single 64-bit atomic.
#include "c2d_internal.hpp"
#include "c2d_math.hpp"

namespace c2d {

struct Planes16 { const float* p[16]; };
struct Planes10 { const float* p[10]; };
struct Planes8 { float* p[8]; };

constexpr int kBlock = 256;
constexpr int kWideBlock = 64;          // the 4-pairs-per-lane kernel runs one wave per block
constexpr int kMaxBlocks = 1 << 24;     // beyond this a launch falls back to its grid-stride loop
typedef float f32x4 __attribute__((ext_vector_type(4)));

// Colliding-pair count without a second kernel and without a hot atomic word.
// Same-address atomics from every wave serialise at the memory side (measured:
// 9766 block atomics on one word -> +45 us on a 105 us kernel).  Instead every
// wave makes ONE returning 64-bit atomic add on one of 256 words that sit on
// separate 128-byte lines of the ctx workspace; the word packs
// (arrivals << 40 | partial sum).  Every wave of the grid arrives exactly once,
// so the wave whose add completes a word's expected arrival count owns its
// sum: it clears the word (the workspace is ready for the next launch) and adds
// the sum to the caller's counter — at most 256 adds on that word per launch.
// Measured against per-block partials + a finishing kernel: 105.4 vs 107.9 us.
constexpr uint32_t kCountWords = 256;  // x 128 B = 32 KiB of ctx workspace

C2D_DEV void wave_count_arrive(uint32_t lane_count, unsigned long long* __restrict__ d_count,
                               unsigned long long* __restrict__ words)
{
    uint32_t v = lane_count;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    if ((threadIdx.x & 63) == 0) {
        const uint32_t waves_per_block = blockDim.x >> 6;
        const uint32_t wave_id = blockIdx.x * waves_per_block + (threadIdx.x >> 6);
        const uint32_t n_waves = gridDim.x * waves_per_block;
        const uint32_t slot = wave_id & (kCountWords - 1);
        const uint32_t expected = n_waves / kCountWords + (slot < (n_waves & (kCountWords - 1)) ? 1u : 0u);
        unsigned long long* w = words + (size_t)slot * 16;
        const unsigned long long old = atomicAdd(w, (1ull << 40) | (unsigned long long)v);
        if ((uint32_t)(old >> 40) + 1u == expected) {
            const unsigned long long total = (old & ((1ull << 40) - 1)) + v;
            atomicExch(w, 0ull);
            if (total) atomicAdd(d_count, total);
        }
    }
}

// ---- rectangle pairs, vertex format ------------------------------------------
// VEC == 4: planes read as float4 (16 B / lane), results written as one dword.
// VEC == 1: scalar loads; used for the tail and for unaligned buffers.
template <int VEC, int BLOCK>
__global__ __launch_bounds__(BLOCK) void sat_rect_verts_kernel(Planes16 P, size_t first, size_t n_groups,
                                                                uint8_t* __restrict__ out,
                                                                unsigned long long* __restrict__ d_count, unsigned long long* __restrict__ words)
{
    uint32_t my_count = 0;
    const size_t stride = (size_t)gridDim.x * BLOCK;
    for (size_t g = (size_t)blockIdx.x * BLOCK + threadIdx.x; g < n_groups; g += stride) {
        if constexpr (VEC == 4) {
            f32x4 v[16];
#pragma unroll
            for (int k = 0; k < 16; k++) v[k] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(P.p[k]) + g);
            uint32_t packed = 0;
#pragma unroll
            for (int e = 0; e < 4; e++) {
                float r1[8], r2[8];
#pragma unroll
                for (int k = 0; k < 8; k++) {
                    r1[k] = v[k][e];
                    r2[k] = v[8 + k][e];
                }
                uint32_t c = rect_collide(r1, r2) ? 1u : 0u;
                packed |= c << (8 * e);
            }
            my_count += (uint32_t)__popc(packed);
            __builtin_nontemporal_store(packed, reinterpret_cast<uint32_t*>(out) + g);
        } else {
            const size_t i = first + g;
            float r1[8], r2[8];
#pragma unroll
            for (int k = 0; k < 8; k++) {
                r1[k] = P.p[k][i];
                r2[k] = P.p[8 + k][i];
            }
            uint32_t c = rect_collide(r1, r2) ? 1u : 0u;
            out[i] = (uint8_t)c;
            my_count += c;
        }
    }
    if (d_count) wave_count_arrive(my_count, d_count, words);
}

// ---- rectangle pairs, array-of-rectangles format ----------------------------------------
// The reference's own argument layout, convex_collide(float* r1, float* r2) with flat
// float[8] rectangles (utils.cu:159), batched: r1, r2 are f32[n][8].  A lane reads its
// pair as 4 x 16 B; a wave covers 2 x 2 KiB of contiguous memory, so the loads coalesce
// as well as the SoA planes do (same 65 B/pair).
__global__ __launch_bounds__(kBlock) void sat_rect_aos_kernel(const float* __restrict__ r1s, const float* __restrict__ r2s,
                                                              size_t n, uint8_t* __restrict__ out,
                                                              unsigned long long* __restrict__ d_count,
                                                              unsigned long long* __restrict__ words)
{
    // A wave owns 64 consecutive pairs = 2 KiB of r1 and 2 KiB of r2.  Lane i fetches the 16-byte
    // chunks i and 64 + i of each (two fully coalesced 1-KiB loads per array), the chunks go through
    // a wave-private LDS tile, and lane p reads back its own pair (chunks 2p, 2p + 1).  Loading a
    // lane's 32 bytes directly (stride-32 16-byte loads) touches every cache line twice: 120 vs 104 us.
    __shared__ __attribute__((aligned(16))) f32x4 tile[kBlock / 64][2][128];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t my_count = 0;
    const size_t n_tiles = (n + 63) / 64;
    const size_t tiles_per_pass = (size_t)gridDim.x * (kBlock / 64);
    for (size_t t = (size_t)blockIdx.x * (kBlock / 64) + wave; t < n_tiles; t += tiles_per_pass) {
        const size_t p0 = t * 64;
        const size_t chunks = (n - p0 < 64 ? n - p0 : 64) * 2;  // 16-byte chunks of this tile per array
        const f32x4* a = reinterpret_cast<const f32x4*>(r1s) + 2 * p0;
        const f32x4* b = reinterpret_cast<const f32x4*>(r2s) + 2 * p0;
        const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
        const f32x4 a_lo = (size_t)lane < chunks ? __builtin_nontemporal_load(a + lane) : zero;
        const f32x4 a_hi = (size_t)(64 + lane) < chunks ? __builtin_nontemporal_load(a + 64 + lane) : zero;
        const f32x4 b_lo = (size_t)lane < chunks ? __builtin_nontemporal_load(b + lane) : zero;
        const f32x4 b_hi = (size_t)(64 + lane) < chunks ? __builtin_nontemporal_load(b + 64 + lane) : zero;
        tile[wave][0][lane] = a_lo;
        tile[wave][0][64 + lane] = a_hi;
        tile[wave][1][lane] = b_lo;
        tile[wave][1][64 + lane] = b_hi;
        // same wave wrote and reads: LDS operations of a wave complete in order
        const f32x4 a0 = tile[wave][0][2 * lane], a1 = tile[wave][0][2 * lane + 1];
        const f32x4 b0 = tile[wave][1][2 * lane], b1 = tile[wave][1][2 * lane + 1];
        const float r1[8] = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
        const float r2[8] = {b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]};
        const uint32_t c = rect_collide(r1, r2) ? 1u : 0u;
        if (p0 + lane < n) {
            out[p0 + lane] = (uint8_t)c;
            my_count += c;
        }
    }
    if (d_count) wave_count_arrive(my_count, d_count, words);
}

// ---- rectangle pairs, pose format (10 planes, 41 B/pair) ------------------------------
// Same lane mapping as the vertex kernel (VEC == 4: one float4 per plane, 4 pairs
// per lane); both rectangles are rebuilt per pair (2 sincos + 2 x 16 ops), which
// moves this format towards the VALU roof: ~400 VALU instructions per 41 bytes.
C2D_DEV uint32_t pose_pair_collides(const float (&v)[10])
{
    float r1[8], r2[8], s, c;
    sincos_(v[4], s, c);
    rect_from_half_extents(v[2] / 2, v[3] / 2, c, s, v[0], v[1], r1);
    sincos_(v[9], s, c);
    rect_from_half_extents(v[7] / 2, v[8] / 2, c, s, v[5], v[6], r2);
    return rect_collide(r1, r2) ? 1u : 0u;
}

template <int VEC, int BLOCK>
__global__ __launch_bounds__(BLOCK) void sat_rect_pose_kernel(Planes10 P, size_t first, size_t n_groups,
                                                              uint8_t* __restrict__ out,
                                                              unsigned long long* __restrict__ d_count,
                                                              unsigned long long* __restrict__ words)
{
    uint32_t my_count = 0;
    const size_t stride = (size_t)gridDim.x * BLOCK;
    for (size_t g = (size_t)blockIdx.x * BLOCK + threadIdx.x; g < n_groups; g += stride) {
        if constexpr (VEC == 4) {
            f32x4 q[10];
#pragma unroll
            for (int k = 0; k < 10; k++) q[k] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(P.p[k]) + g);
            uint32_t packed = 0;
#pragma unroll
            for (int e = 0; e < 4; e++) {
                float v[10];
#pragma unroll
                for (int k = 0; k < 10; k++) v[k] = q[k][e];
                packed |= pose_pair_collides(v) << (8 * e);
            }
            my_count += (uint32_t)__popc(packed);
            __builtin_nontemporal_store(packed, reinterpret_cast<uint32_t*>(out) + g);
        } else {
            const size_t i = first + g;
            float v[10];
#pragma unroll
            for (int k = 0; k < 10; k++) v[k] = P.p[k][i];
            const uint32_t hit = pose_pair_collides(v);
            out[i] = (uint8_t)hit;
            my_count += hit;
        }
    }
    if (d_count) wave_count_arrive(my_count, d_count, words);
}

// ---- create_rect + rot_trans_rectangle over SoA (reference utils.cu:119-142) ------
__global__ __launch_bounds__(kBlock) void rects_from_poses_kernel(const float* __restrict__ cx, const float* __restrict__ cy,
                                                                  const float* __restrict__ w, const float* __restrict__ h,
                                                                  const float* __restrict__ th, size_t n, Planes8 O)
{
    const size_t stride = (size_t)gridDim.x * kBlock;
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
        float r[8], s, c;
        sincos_(th[i], s, c);
        rect_from_half_extents(w[i] / 2, h[i] / 2, c, s, cx[i], cy[i], r);
#pragma unroll
        for (int k = 0; k < 8; k++) O.p[k][i] = r[k];
    }
}

// ---- convex polygons, K <= 16, true normals ----------------------------------------
// Work split: kPolyLanes (= 32 / kPolyAxes) lanes per pair, a lane owns
// kPolyAxes of the pair's ka+kb axes (axis l, l + kPolyLanes, ...), so all axes
// of a pair are evaluated side by side and "found a separating axis" is one
// ballot (the wave-level form of an early-out: nothing is evaluated after the
// decision is known because every axis runs at once).  Vertices are staged in
// LDS by the whole block with coalesced loads (pair index fastest in memory);
// every lane then walks the pair's vertex list two vertices per ds_read_b128
// (an LDS broadcast within the pair's lanes) and applies each vertex to all of
// its axes from registers.  That register blocking is what matters: with one
// axis per lane the kernel was bound by LDS instruction issue (one b128 read
// per 10 VALU instructions on every SIMD), with four axes per lane it is one
// read per 40.  The next pass's vertices are prefetched into registers while the
// current pass is evaluated, so the global-load latency hides behind compute.
constexpr int kPolyPairs = 64;                  // pairs staged per block pass
constexpr int kPolyStride = 2 * C2D_POLY_KMAX;  // vertices per pair slot (A then B) = max axes per pair
constexpr int kPolyPitch = kPolyStride + 2;     // float2 per pair slot: 272 B = 17 x 16 B keeps b128 reads
                                                // aligned and puts neighbouring pairs on disjoint banks
#ifndef C2D_POLY_AXES_PER_LANE
#define C2D_POLY_AXES_PER_LANE 4
#endif
constexpr int kPolyAxes = C2D_POLY_AXES_PER_LANE;
constexpr int kPolyLanes = kPolyStride / kPolyAxes;  // lanes per pair

C2D_DEV void minmax_update(float nx, float ny, float x, float y, float& mn, float& mx)
{
    const float p = nx * x + ny * y;  // unfused, reference utils.cu:173
    mn = __builtin_fminf(mn, p);
    mx = __builtin_fmaxf(mx, p);
}

__global__ __launch_bounds__(kBlock) void sat_poly_kernel(const float* __restrict__ vx, const float* __restrict__ vy,
                                                          const uint8_t* __restrict__ kcnt, size_t n,
                                                          uint8_t* __restrict__ out,
                                                          unsigned long long* __restrict__ d_count,
                                                          unsigned long long* __restrict__ words)
{
    __shared__ __attribute__((aligned(16))) float2 s_v[kPolyPairs][kPolyPitch];
    __shared__ uint8_t s_k[2][kPolyPairs];
    uint32_t my_count = 0;
    const int tid = threadIdx.x;
    const int group = tid / kPolyLanes;  // which pair of the current group of kBlock / kPolyLanes
    const int l = tid % kPolyLanes;      // lane within the pair
    const size_t n_pass = (n + kPolyPairs - 1) / kPolyPairs;
    // Register prefetch of the next pass: thread t stages pair j = t % 64, vertex rows
    // t/64, t/64 + 4, ... (8 rows) and, for t < 128, one vertex count.  All 32 vertex slots of the
    // padded layout are read, so the loads depend on nothing and fly during the previous pass.
    constexpr int kRows = kPolyStride / (kBlock / kPolyPairs);  // rows per thread = 8
    const int sj = tid & (kPolyPairs - 1);
    const int row0 = tid / kPolyPairs;
    float px[kRows], py[kRows];
    uint8_t pk = 0;
    auto prefetch = [&](size_t pass) {
        const size_t base = pass * kPolyPairs;
        const bool in = base + sj < n;
#pragma unroll
        for (int i = 0; i < kRows; i++) {
            const int row = row0 + i * (kBlock / kPolyPairs);
            const size_t idx = (size_t)row * n + base + sj;  // row = polygon * KMAX + vertex
            px[i] = in ? __builtin_nontemporal_load(vx + idx) : 0.0f;
            py[i] = in ? __builtin_nontemporal_load(vy + idx) : 0.0f;
        }
        if (tid < 2 * kPolyPairs) pk = in ? kcnt[(size_t)(tid / kPolyPairs) * n + base + sj] : (uint8_t)0;
    };
    size_t pass = blockIdx.x;
    if (pass < n_pass) prefetch(pass);
    for (; pass < n_pass; pass += gridDim.x) {
        const size_t base = pass * kPolyPairs;
        const int pairs_here = (int)((n - base) < (size_t)kPolyPairs ? (n - base) : (size_t)kPolyPairs);
        __syncthreads();  // previous pass finished reading LDS
#pragma unroll
        for (int i = 0; i < kRows; i++) s_v[sj][row0 + i * (kBlock / kPolyPairs)] = make_float2(px[i], py[i]);
        if (tid < 2 * kPolyPairs) s_k[tid / kPolyPairs][sj] = pk;
        __syncthreads();
        if (pass + gridDim.x < n_pass) prefetch(pass + gridDim.x);
        // evaluate: kBlock / kPolyLanes pairs at a time
        for (int j0 = 0; j0 < pairs_here; j0 += kBlock / kPolyLanes) {
            const int j = j0 + group;
            const bool live = j < pairs_here;
            const int jj = live ? j : 0;
            const int ka = live ? (int)s_k[0][jj] : 0, kb = live ? (int)s_k[1][jj] : 0;
            const float2* A = &s_v[jj][0];
            const float2* B = &s_v[jj][C2D_POLY_KMAX];
            float nx[kPolyAxes], ny[kPolyAxes], min1[kPolyAxes], max1[kPolyAxes], min2[kPolyAxes], max2[kPolyAxes];
            bool valid[kPolyAxes];
#pragma unroll
            for (int r = 0; r < kPolyAxes; r++) {
                const int a = l + r * kPolyLanes;
                valid[r] = a < ka + kb;
                const bool onA = a < ka;
                const float2* Pn = onA ? A : B;
                const int kp = onA ? ka : kb;
                const int i = valid[r] ? (onA ? a : a - ka) : 0;
                const int i1 = (i + 1 >= kp) ? 0 : i + 1;
                const float2 e0 = Pn[i], e1 = Pn[i1];
                nx[r] = -(e1.y - e0.y);   // true normal (-ey, ex)
                ny[r] = e1.x - e0.x;
                min1[r] = min2[r] = __builtin_inff();
                max1[r] = max2[r] = -__builtin_inff();
            }
            const float4* A4 = reinterpret_cast<const float4*>(A);
            const float4* B4 = reinterpret_cast<const float4*>(B);
            // two vertices per ds_read_b128; an odd count ends with one single vertex
            for (int k = 0; 2 * k + 1 < ka; k++) {
                const float4 q = A4[k];
#pragma unroll
                for (int r = 0; r < kPolyAxes; r++) {
                    minmax_update(nx[r], ny[r], q.x, q.y, min1[r], max1[r]);
                    minmax_update(nx[r], ny[r], q.z, q.w, min1[r], max1[r]);
                }
            }
            if (ka & 1) {
                const float2 q = A[ka - 1];
#pragma unroll
                for (int r = 0; r < kPolyAxes; r++) minmax_update(nx[r], ny[r], q.x, q.y, min1[r], max1[r]);
            }
            for (int k = 0; 2 * k + 1 < kb; k++) {
                const float4 q = B4[k];
#pragma unroll
                for (int r = 0; r < kPolyAxes; r++) {
                    minmax_update(nx[r], ny[r], q.x, q.y, min2[r], max2[r]);
                    minmax_update(nx[r], ny[r], q.z, q.w, min2[r], max2[r]);
                }
            }
            if (kb & 1) {
                const float2 q = B[kb - 1];
#pragma unroll
                for (int r = 0; r < kPolyAxes; r++) minmax_update(nx[r], ny[r], q.x, q.y, min2[r], max2[r]);
            }
            bool sep = false;
#pragma unroll
            for (int r = 0; r < kPolyAxes; r++) sep |= valid[r] && ((max1[r] < min2[r]) || (max2[r] < min1[r]));
            const unsigned long long ballot = __ballot(sep);
            const int shift = (tid & 63) - l;  // first lane of this pair within the wave
            const unsigned long long mine = (ballot >> shift) & ((1ull << kPolyLanes) - 1ull);
            if (l == 0 && live) {
                const uint32_t c = mine == 0 ? 1u : 0u;
                out[base + j] = (uint8_t)c;
                my_count += c;
            }
        }
    }
    if (d_count) wave_count_arrive(my_count, d_count, words);
}

__global__ void poly_validate_kernel(const uint8_t* __restrict__ kcnt, size_t n2, uint32_t* __restrict__ bad)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    uint32_t b = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += stride) {
        uint8_t k = kcnt[i];
        b |= (k < 1 || k > C2D_POLY_KMAX) ? 1u : 0u;
    }
    if (b) atomicOr(bad, 1u);
}

static inline bool aligned_to(const void* p, size_t a) { return (reinterpret_cast<uintptr_t>(p) & (a - 1)) == 0; }

}  // namespace c2d

using namespace c2d;

extern "C" {

int c2d_rects_from_poses(c2d_ctx* ctx, const float* d_cx, const float* d_cy, const float* d_w, const float* d_h,
                         const float* d_theta, size_t n, float* const d_out_planes[8], c2d_stream stream)
{
    if (!ctx) return C2D_ERR_INVALID_ARG;
    if (n == 0) return C2D_OK;
    if (!d_cx || !d_cy || !d_w || !d_h || !d_theta || !d_out_planes) return fail_arg(ctx, "c2d_rects_from_poses: NULL plane");
    Planes8 O;
    for (int k = 0; k < 8; k++) {
        if (!d_out_planes[k]) return fail_arg(ctx, "c2d_rects_from_poses: NULL output plane");
        O.p[k] = d_out_planes[k];
    }
    DeviceGuard g(ctx->device);
    const int grid = grid_for(n, kBlock, ctx->prop.multiProcessorCount * 8);
    hipLaunchKernelGGL(rects_from_poses_kernel, dim3(grid), dim3(kBlock), 0, (hipStream_t)stream, d_cx, d_cy, d_w, d_h,
                       d_theta, n, O);
    C2D_LAUNCH_CHECK(ctx);
    return C2D_OK;
}

int c2d_sat_rect_pairs_verts(c2d_ctx* ctx, const float* const d_planes[16], size_t n, uint8_t* d_out,
                             unsigned long long* d_count, c2d_stream stream)
{
    if (!ctx) return C2D_ERR_INVALID_ARG;
    if (n == 0) return C2D_OK;
    if (!d_planes || !d_out) return fail_arg(ctx, "c2d_sat_rect_pairs_verts: NULL argument");
    Planes16 P;
    bool wide = aligned_to(d_out, 4);
    for (int k = 0; k < 16; k++) {
        if (!d_planes[k]) return fail_arg(ctx, "c2d_sat_rect_pairs_verts: NULL plane");
        P.p[k] = d_planes[k];
        wide = wide && aligned_to(d_planes[k], 16);
    }
    DeviceGuard g(ctx->device);
    hipStream_t s = (hipStream_t)stream;
    // One group of 4 pairs per lane, 64-thread blocks, no grid-stride loop below kMaxBlocks:
    // short single-wave blocks retiring all through the launch stream better than a
    // resident grid-stride grid (tools/sat_tune: 103.0 vs 109.3 us per 1e7 pairs).
    const size_t n4 = wide ? n / 4 : 0;
    if (n4) {
        const int grid = grid_for(n4, kWideBlock, kMaxBlocks);
        hipLaunchKernelGGL((sat_rect_verts_kernel<4, kWideBlock>), dim3(grid), dim3(kWideBlock), 0, s, P, (size_t)0, n4, d_out,
                           d_count, ctx->d_count_words);
        C2D_LAUNCH_CHECK(ctx);
    }
    const size_t rest = n - 4 * n4;
    if (rest) {
        const int grid = grid_for(rest, kBlock, kMaxBlocks);
        hipLaunchKernelGGL((sat_rect_verts_kernel<1, kBlock>), dim3(grid), dim3(kBlock), 0, s, P, 4 * n4, rest, d_out, d_count,
                           ctx->d_count_words);
        C2D_LAUNCH_CHECK(ctx);
    }
    return C2D_OK;
}

int c2d_sat_rect_pairs_aos(c2d_ctx* ctx, const float* d_r1, const float* d_r2, size_t n, uint8_t* d_out,
                           unsigned long long* d_count, c2d_stream stream)
{
    if (!ctx) return C2D_ERR_INVALID_ARG;
    if (n == 0) return C2D_OK;
    if (!d_r1 || !d_r2 || !d_out) return fail_arg(ctx, "c2d_sat_rect_pairs_aos: NULL argument");
    if (!aligned_to(d_r1, 16) || !aligned_to(d_r2, 16)) return fail_arg(ctx, "c2d_sat_rect_pairs_aos: rectangle arrays must be 16-byte aligned");
    DeviceGuard g(ctx->device);
    const int grid = grid_for(n, kBlock, kMaxBlocks);  // one 64-pair tile per wave
    hipLaunchKernelGGL(sat_rect_aos_kernel, dim3(grid), dim3(kBlock), 0, (hipStream_t)stream, d_r1, d_r2, n, d_out, d_count,
                       ctx->d_count_words);
    C2D_LAUNCH_CHECK(ctx);
    return C2D_OK;
}

int c2d_sat_rect_pairs_pose(c2d_ctx* ctx, const float* const d_pose_planes[10], size_t n, uint8_t* d_out,
                            unsigned long long* d_count, c2d_stream stream)
{
    if (!ctx) return C2D_ERR_INVALID_ARG;
    if (n == 0) return C2D_OK;
    if (!d_pose_planes || !d_out) return fail_arg(ctx, "c2d_sat_rect_pairs_pose: NULL argument");
    Planes10 P;
    bool wide = aligned_to(d_out, 4);
    for (int k = 0; k < 10; k++) {
        if (!d_pose_planes[k]) return fail_arg(ctx, "c2d_sat_rect_pairs_pose: NULL plane");
        P.p[k] = d_pose_planes[k];
        wide = wide && aligned_to(d_pose_planes[k], 16);
    }
    DeviceGuard g(ctx->device);
    hipStream_t s = (hipStream_t)stream;
    const size_t n4 = wide ? n / 4 : 0;
    if (n4) {
        const int grid = grid_for(n4, kWideBlock, kMaxBlocks);
        hipLaunchKernelGGL((sat_rect_pose_kernel<4, kWideBlock>), dim3(grid), dim3(kWideBlock), 0, s, P, (size_t)0, n4, d_out,
                           d_count, ctx->d_count_words);
        C2D_LAUNCH_CHECK(ctx);
    }
    const size_t rest = n - 4 * n4;
    if (rest) {
        const int grid = grid_for(rest, kBlock, kMaxBlocks);
        hipLaunchKernelGGL((sat_rect_pose_kernel<1, kBlock>), dim3(grid), dim3(kBlock), 0, s, P, 4 * n4, rest, d_out, d_count,
                           ctx->d_count_words);
        C2D_LAUNCH_CHECK(ctx);
    }
    return C2D_OK;
}

int c2d_sat_poly_pairs(c2d_ctx* ctx, const float* d_vx, const float* d_vy, const uint8_t* d_k, size_t n,
                       uint8_t* d_out, unsigned long long* d_count, c2d_stream stream)
{
    if (!ctx) return C2D_ERR_INVALID_ARG;
    if (n == 0) return C2D_OK;
    if (!d_vx || !d_vy || !d_k || !d_out) return fail_arg(ctx, "c2d_sat_poly_pairs: NULL argument");
    DeviceGuard g(ctx->device);
    hipStream_t s = (hipStream_t)stream;
    // vertex counts outside 1..KMAX would index past a pair's LDS slot: reject them up front
    C2D_HIP(ctx, hipMemsetAsync(ctx->d_counters + 8, 0, sizeof(uint32_t), s));
    hipLaunchKernelGGL(poly_validate_kernel, dim3(grid_for(2 * n, 256, ctx->prop.multiProcessorCount * 4)), dim3(256), 0, s,
                       d_k, 2 * n, ctx->d_counters + 8);
    C2D_LAUNCH_CHECK(ctx);
    C2D_HIP(ctx, hipMemcpyAsync(ctx->h_pinned + 8, ctx->d_counters + 8, sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    C2D_HIP(ctx, hipStreamSynchronize(s));
    if (ctx->h_pinned[8]) return fail_arg(ctx, "c2d_sat_poly_pairs: vertex count outside 1..C2D_POLY_KMAX");
    const size_t n_pass = (n + kPolyPairs - 1) / kPolyPairs;
    const int grid = (int)(n_pass < (size_t)ctx->prop.multiProcessorCount * 8 ? n_pass : (size_t)ctx->prop.multiProcessorCount * 8);
    hipLaunchKernelGGL(sat_poly_kernel, dim3(grid), dim3(kBlock), 0, s, d_vx, d_vy, d_k, n, d_out, d_count, ctx->d_count_words);
    C2D_LAUNCH_CHECK(ctx);
    return C2D_OK;
}

}  // extern "C"
