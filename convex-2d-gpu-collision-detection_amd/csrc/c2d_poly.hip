// c2d_poly.hip — SAT for arbitrary convex polygons, K <= 16 vertices, true edge normals
// (BASELINE config 5; SURVEY.md F5: the reference's edge-as-axis shortcut, utils.cu:170-171,
// is only valid for rectangles).  Projection and the strict-< interval test are
// utils.cu:172-180 unchanged.
//
// The result is an OR over axes — "some axis separates" — so ANY axis that separates, evaluated
// with exactly the canonical arithmetic, decides a pair; the order in which axes are tried is
// free.  The kernel therefore runs in two phases per wave of 64 pairs:
//
//   phase 1, one pair per lane, everything in registers (no LDS, no cross-lane traffic):
//     the 2 x 16 padded vertex rows arrive as coalesced 256-byte row segments (pair index is
//     the fastest dimension), slots >= k are overwritten with vertex 0 (a repeated vertex adds
//     a zero-length edge, whose axis (0, 0) never separates, and repeats a projection, which
//     changes no min/max: padding is exactly neutral), and ONE axis of polygon A is tested: the
//     edge whose normal is best aligned with the direction between the vertex means (a
//     heuristic in fast arithmetic — it only chooses WHICH canonical axis is evaluated).
//     On the bench workload (94 % of the pairs separated) this single axis decides 91 % of all
//     pairs; rows above the wave's largest vertex count are neither loaded nor evaluated.
//
//   phase 2, the whole wave per undecided pair (colliding pairs need all (ka+kb)^2 products):
//     the owner lane parks its 32 vertices in a 256-byte LDS slot; lane (axis a, half h)
//     builds the normal of edge a (A's edges 0..15, B's 16..31) and projects polygon h's
//     vertices onto it with two-vertex ds_read_b128 broadcasts; the two halves swap their
//     intervals (lane ^ 32) and one ballot gives "some axis separates".
//
// The wave-wide early-out of the north star is this split: nothing beyond one axis is evaluated
// for a pair that the first axis separates, and no lane idles on another pair's full evaluation.
#include "c2d_internal.hpp"
#include "c2d_math.hpp"
#include "c2d_count.hpp"

namespace c2d {

constexpr int kSlots = 8;          // undecided pairs parked in LDS per group (16 KM bytes each)

C2D_DEV void minmax_update(float nx, float ny, float x, float y, float& mn, float& mx)
{
    const float p = nx * x + ny * y;  // unfused (translation unit is -ffp-contract=off), utils.cu:173
    mn = __builtin_fminf(mn, p);
    mx = __builtin_fmaxf(mx, p);
}

C2D_DEV uint32_t wave_max_u32(uint32_t v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const uint32_t o = (uint32_t)__shfl_xor((int)v, off, 64);
        v = o > v ? o : v;
    }
    return (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
}

// One wave = one tile of 64 pairs.  KM (4, 8 or 16) is the number of vertex slots a lane holds per polygon: the
// smallest that covers the layout's `rows` (vertex rows per polygon in memory, f32[2][rows][n]).  Small polygons
// then cost a quarter or half of the registers (8 waves per SIMD instead of 5), a shorter instruction stream, and
// in phase 2 four or eight pairs share a wave instead of two.
template <int KM, int MIN_WAVES, bool FULL>
__global__ __launch_bounds__(64, MIN_WAVES) void sat_poly_kernel(const float* __restrict__ vx, const float* __restrict__ vy,
                                                                 const uint8_t* __restrict__ kcnt, size_t n, int rows_arg,
                                                                 uint8_t* __restrict__ out,
                                                                 unsigned long long* __restrict__ d_count,
                                                                 CountWs words,
                                                                 uint32_t* __restrict__ async_err)
{
    const int rows = FULL ? KM : rows_arg;  // FULL: the layout has exactly KM rows (a compile-time constant)
    constexpr int LP = 2 * KM;      // lanes per pair in phase 2 = axis slots of a pair
    constexpr int PP = 64 / LP;     // pairs evaluated side by side in phase 2
    // phase-2 slots: A's 16 vertices then B's 16 vertices, (x, y) interleaved
    __shared__ __attribute__((aligned(16))) float2 s_slot[kSlots][2 * KM];
    const uint32_t lane = threadIdx.x;
    uint32_t n_collide = 0;
    const size_t n_tiles = (n + 63) / 64;
    for (size_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const size_t p0 = tile * 64;
        const uint32_t here = (uint32_t)((n - p0) < (size_t)64 ? (n - p0) : (size_t)64);  // wave-uniform
        const bool in = lane < here;
        const uint32_t cl = in ? lane : here - 1;  // lanes past the end re-read the last pair (never stored)
        // ---- vertex counts; out-of-range counts are clamped (memory safety) and reported -------
        int ka = kcnt[p0 + cl], kb = kcnt[n + p0 + cl];
        const bool bad = ka < 1 || ka > rows || kb < 1 || kb > rows;
        ka = ka < 1 ? 1 : (ka > rows ? rows : ka);
        kb = kb < 1 ? 1 : (kb > rows ? rows : kb);
        if (__ballot(bad) != 0 && lane == 0) __hip_atomic_fetch_or(async_err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        const int kmaxA = (int)wave_max_u32((uint32_t)ka), kmaxB = (int)wave_max_u32((uint32_t)kb);
        // ---- rows: scalar base + 32-bit lane offset, rows >= the wave's maximum are skipped ------
        float ax[KM], ay[KM], bx[KM], by[KM];
        const float* rx = vx + p0;
        const float* ry = vy + p0;
        // byte offset of the lane: global_load_dword v, v_off, s[base] (scalar row base + zero-extended 32-bit
        // lane offset) costs no address arithmetic.  The empty asm keeps the zero-extension next to each load:
        // hoisted out of the row's block, instruction selection no longer sees it and falls back to a 64-bit add.
        uint32_t off = cl * 4u;
        auto ld = [&off](const float* row) {
            asm volatile("" : "+v"(off));
            return __builtin_nontemporal_load(reinterpret_cast<const float*>(reinterpret_cast<const char*>(row) + off));
        };
#pragma unroll
        for (int r = 0; r < KM; r++) {
            if (r < kmaxA) {
                ax[r] = ld(rx);
                ay[r] = ld(ry);
            } else {
                ax[r] = 0.0f;
                ay[r] = 0.0f;
            }
            rx += n;
            ry += n;
        }
        if constexpr (!FULL) {  // polygon B starts `rows` rows in (with rows == KM the pointers are already there)
            rx = vx + (size_t)rows * n + p0;
            ry = vy + (size_t)rows * n + p0;
        }
#pragma unroll
        for (int r = 0; r < KM; r++) {
            if (r < kmaxB) {
                bx[r] = ld(rx);
                by[r] = ld(ry);
            } else {
                bx[r] = 0.0f;
                by[r] = 0.0f;
            }
            rx += n;
            ry += n;
        }
        // ---- neutral padding: slots >= k repeat vertex 0; vertex sums for the heuristic -------------
        float sax = ax[0], say = ay[0], sbx = bx[0], sby = by[0];
#pragma unroll
        for (int r = 1; r < KM; r++) {
            const bool ua = r < ka, ub = r < kb;
            ax[r] = ua ? ax[r] : ax[0];
            ay[r] = ua ? ay[r] : ay[0];
            bx[r] = ub ? bx[r] : bx[0];
            by[r] = ub ? by[r] : by[0];
            if (r < kmaxA) { sax += ax[r]; say += ay[r]; }
            if (r < kmaxB) { sbx += bx[r]; sby += by[r]; }
        }
        // mean of the real vertices: the sums above hold (kmax - k) extra copies of vertex 0
        const float fka = (float)ka, fkb = (float)kb;
        const float ia = __builtin_amdgcn_rcpf(fka), ib = __builtin_amdgcn_rcpf(fkb);
        float dX = (sbx - (float)(kmaxB - kb) * bx[0]) * ib - (sax - (float)(kmaxA - ka) * ax[0]) * ia;
        float dY = (sby - (float)(kmaxB - kb) * by[0]) * ib - (say - (float)(kmaxA - ka) * ay[0]) * ia;
        // orientation of A (sign of the first corner's cross product): clockwise polygons have
        // inward-pointing (-ey, ex), so the preferred direction flips
        {
            const float c = (ax[1] - ax[0]) * (ay[2] - ay[0]) - (ay[1] - ay[0]) * (ax[2] - ax[0]);
            const uint32_t sgn = __float_as_uint(c) & 0x80000000u;
            dX = __uint_as_float(__float_as_uint(dX) ^ sgn);
            dY = __uint_as_float(__float_as_uint(dY) ^ sgn);
        }
        // ---- phase 1: pick A's edge whose normal points best towards B, test that one axis -----------
        float best = -__builtin_inff(), nx1 = 0.0f, ny1 = 0.0f;
#pragma unroll
        for (int r = 0; r < KM; r++) {
            if (r < kmaxA) {
                const int r1 = (r + 1) & (KM - 1);
                const float ey = ay[r1] - ay[r];
                const float nx = -ey;               // true normal (-ey, ex), exactly as phase 2 and the oracle
                const float ny = ax[r1] - ax[r];
                const float nd = fma_(nx, dX, ny * dY);
                const float l2 = fma_(nx, nx, ny * ny);
                const float s = nd * __builtin_amdgcn_rsqf(l2);  // zero edge: 0 * inf = NaN, never "better"
                const bool better = s > best;
                best = better ? s : best;
                nx1 = better ? nx : nx1;
                ny1 = better ? ny : ny1;
            }
        }
        float mnA = __builtin_inff(), mxA = -__builtin_inff(), mnB = __builtin_inff(), mxB = -__builtin_inff();
#pragma unroll
        for (int r = 0; r < KM; r++)
            if (r < kmaxA) minmax_update(nx1, ny1, ax[r], ay[r], mnA, mxA);
#pragma unroll
        for (int r = 0; r < KM; r++)
            if (r < kmaxB) minmax_update(nx1, ny1, bx[r], by[r], mnB, mxB);
        // (a NaN first projection keeps an axis from separating, as the comparison-based extremes of utils.cu:176-178
        // would: see first_projections_ordered in c2d_math.hpp)
        bool sep = ((mxA < mnB) || (mxB < mnA)) && first_projections_ordered(nx1 * ax[0] + ny1 * ay[0], nx1 * bx[0] + ny1 * by[0]);
        sep = sep || bad;  // out-of-range vertex count: reported, result 0
        // ---- phase 2: full evaluation of the pairs that are still undecided -------------------------------
        // Up to kSlots undecided lanes park their vertices at once (the ds_write_b128 are issued once per group, not
        // once per pair); then PP pairs are evaluated side by side, LP = 2 KM lanes each: lane a of a pair's lanes owns
        // axis slot a (A's edges 0..KM-1, B's edges KM..2KM-1) and projects all vertices of both polygons, two per
        // broadcast ds_read_b128, so "some axis separates" is one slice of a ballot and no lane exchange is needed.
        unsigned long long todo = __ballot(in && !sep);
        while (todo) {
            // lane roles of phase 2, derived here (from an opaque copy of the lane id, so that they are not hoisted
            // into phase 1 where every register holds a vertex)
            uint32_t lane2 = lane;
            asm volatile("" : "+v"(lane2));
            const int a = (int)(lane2 % LP);
            const int sub = (int)(lane2 / LP);
            const int i0 = a, i1 = (a & KM) | ((a + 1) & (KM - 1));
            const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(todo >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)todo, 0u));
            const bool park = ((todo >> lane) & 1ull) && rank < (uint32_t)kSlots;
            __syncthreads();  // single-wave block: a wave-level fence (no s_barrier is emitted); earlier reads are done
            if (park) {
                float4* S4 = reinterpret_cast<float4*>(&s_slot[rank][0]);
#pragma unroll
                for (int r = 0; r < KM / 2; r++) {
                    S4[r] = make_float4(ax[2 * r], ay[2 * r], ax[2 * r + 1], ay[2 * r + 1]);
                    S4[KM / 2 + r] = make_float4(bx[2 * r], by[2 * r], bx[2 * r + 1], by[2 * r + 1]);
                }
            }
            __syncthreads();
            const int left = __popcll(todo);
            const int g = left < kSlots ? left : kSlots;
            for (int i = 0; i < g; i += PP) {
                const int cnt = (g - i) < PP ? (g - i) : PP;  // pairs of this trip (wave-uniform)
                int j[PP];
                int kA = 0, kB = 0;  // wave-uniform loop bounds: the largest counts among the trip's pairs
#pragma unroll
                for (int q = 0; q < PP; q++) {
                    j[q] = -1;
                    if (q < cnt) {
                        j[q] = __ffsll((long long)todo) - 1;
                        todo &= todo - 1;
                        const int kaq = __builtin_amdgcn_readlane(ka, j[q]), kbq = __builtin_amdgcn_readlane(kb, j[q]);
                        kA = kaq > kA ? kaq : kA;
                        kB = kbq > kB ? kbq : kB;
                    }
                }
                const float2* S = &s_slot[i + (sub < cnt ? sub : 0)][0];  // lanes without a pair of their own repeat the first
                const float4* S4 = reinterpret_cast<const float4*>(S);
                const float2 e0 = S[i0], e1 = S[i1];
                const float nx = -(e1.y - e0.y), ny = e1.x - e0.x;
                float mn1 = __builtin_inff(), mx1 = -__builtin_inff(), mn2 = __builtin_inff(), mx2 = -__builtin_inff();
                // slots past a polygon's count repeat its vertex 0, so running to the largest count of the trip's
                // pairs (rounded up to a vertex pair) needs no masking; one read ahead hides the LDS latency
                float4 q4 = S4[0];
                for (int r2 = 0; 2 * r2 < kA; r2++) {
                    const float4 qn = S4[r2 + 1];  // r2 + 1 <= KM / 2: at worst B's first pair, always inside the slot
                    minmax_update(nx, ny, q4.x, q4.y, mn1, mx1);
                    minmax_update(nx, ny, q4.z, q4.w, mn1, mx1);
                    q4 = qn;
                }
                q4 = S4[KM / 2];
                for (int r2 = 0; 2 * r2 < kB; r2++) {
                    const float4 qn = S4[KM / 2 + ((r2 + 1) & (KM / 2 - 1))];
                    minmax_update(nx, ny, q4.x, q4.y, mn2, mx2);
                    minmax_update(nx, ny, q4.z, q4.w, mn2, mx2);
                    q4 = qn;
                }
                const float pa0 = nx * S[0].x + ny * S[0].y, pb0 = nx * S[KM].x + ny * S[KM].y;  // first projections (vertex 0 of A, of B)
                const unsigned long long bal = __ballot(((mx1 < mn2) || (mx2 < mn1)) && first_projections_ordered(pa0, pb0));
                constexpr unsigned long long kPairMask = LP == 64 ? ~0ull : ((1ull << LP) - 1ull);
#pragma unroll
                for (int q = 0; q < PP; q++) {
                    const bool any = ((bal >> (q * LP)) & kPairMask) != 0ull;
                    sep = ((int)lane == j[q]) ? any : sep;  // j[q] == -1 matches no lane
                }
            }
        }
        const bool collide = in && !sep;
        if (in) out[p0 + lane] = collide ? (uint8_t)1 : (uint8_t)0;
        n_collide += (uint32_t)__popcll(__ballot(collide));
    }
    if (d_count) wave_count_arrive_total2(n_collide, d_count, words);  // one wave per 64 pairs: the two-level count
}

// ---- triangles and quadrilaterals in a 4-row layout: the rectangle kernel's shape ---------------------------------
// With at most 4 + 4 vertices the full evaluation — 8 true-normal axes x 8 vertices — costs what the rectangle SAT
// costs, so there is nothing to gain from a cheap first axis: a lane takes 4 consecutive pairs with 16-byte loads (16
// loads in flight, as sat_rect_verts_kernel), pads each polygon by repeating vertex 0 (exactly neutral, see above) and
// evaluates everything in registers.  No LDS, no second phase: dense and sparse scenes run at the same, HBM-bound rate.
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(64) void sat_poly4_kernel(const float* __restrict__ vx, const float* __restrict__ vy,
                                                       const uint8_t* __restrict__ kcnt, size_t n, size_t n_groups,
                                                       uint8_t* __restrict__ out, unsigned long long* __restrict__ d_count,
                                                       CountWs words, uint32_t* __restrict__ async_err)
{
    uint32_t my_count = 0;
    bool any_bad = false;
    const size_t stride = (size_t)gridDim.x * 64;
    for (size_t g = (size_t)blockIdx.x * 64 + threadIdx.x; g < n_groups; g += stride) {
        f32x4 X[8], Y[8];  // rows 0..3 = polygon A, 4..7 = polygon B; element e = pair 4 g + e
#pragma unroll
        for (int r = 0; r < 8; r++) {
            X[r] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(vx + (size_t)r * n) + g);
            Y[r] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(vy + (size_t)r * n) + g);
        }
        const uint32_t kA4 = __builtin_nontemporal_load(reinterpret_cast<const uint32_t*>(kcnt) + g);
        const uint32_t kB4 = __builtin_nontemporal_load(reinterpret_cast<const uint32_t*>(kcnt + n) + g);
        uint32_t packed = 0;
#pragma unroll
        for (int e = 0; e < 4; e++) {
            int ka = (int)((kA4 >> (8 * e)) & 0xffu), kb = (int)((kB4 >> (8 * e)) & 0xffu);
            const bool bad = ka < 1 || ka > 4 || kb < 1 || kb > 4;
            any_bad |= bad;
            ka = ka < 1 ? 1 : (ka > 4 ? 4 : ka);
            kb = kb < 1 ? 1 : (kb > 4 ? 4 : kb);
            float px[8], py[8];  // A's 4 slots then B's 4 slots
#pragma unroll
            for (int r = 0; r < 4; r++) {
                px[r] = r < ka ? X[r][e] : X[0][e];
                py[r] = r < ka ? Y[r][e] : Y[0][e];
                px[4 + r] = r < kb ? X[4 + r][e] : X[4][e];
                py[4 + r] = r < kb ? Y[4 + r][e] : Y[4][e];
            }
            bool sep = false;
#pragma unroll
            for (int a = 0; a < 8; a++) {
                const int i0 = a, i1 = (a & 4) | ((a + 1) & 3);
                const float nx = -(py[i1] - py[i0]), ny = px[i1] - px[i0];
                float mn1 = __builtin_inff(), mx1 = -__builtin_inff(), mn2 = __builtin_inff(), mx2 = -__builtin_inff();
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    minmax_update(nx, ny, px[r], py[r], mn1, mx1);
                    minmax_update(nx, ny, px[4 + r], py[4 + r], mn2, mx2);
                }
                sep |= ((mx1 < mn2) || (mx2 < mn1)) && first_projections_ordered(nx * px[0] + ny * py[0], nx * px[4] + ny * py[4]);
            }
            packed |= ((sep || bad) ? 0u : 1u) << (8 * e);
        }
        my_count += (uint32_t)__popc(packed);
        __builtin_nontemporal_store(packed, reinterpret_cast<uint32_t*>(out) + g);
    }
    if (__ballot(any_bad) != 0 && threadIdx.x == 0) __hip_atomic_fetch_or(async_err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if (d_count) wave_count_arrive(my_count, d_count, words);
}

// c2d_poly_binned.hip: a padded layout run as ONE bin of the binned kernel (its 12- and 16-row instances)
int launch_poly_onebin(c2d_ctx* ctx, hipStream_t s, const float* d_vx, const float* d_vy, const uint8_t* d_k, size_t n, int rows, uint8_t* d_out,
                       unsigned long long* d_count, uint32_t* async_err);

template <int KM, int MIN_WAVES, bool FULL>
static void launch_poly(c2d_ctx* ctx, hipStream_t s, const float* d_vx, const float* d_vy, const uint8_t* d_k, size_t n, int rows, uint8_t* d_out,
                        unsigned long long* d_count, uint32_t* async_err)
{
    // tiles per wave: 2 pay in the binned kernel (c2d_poly_binned.hip); here, at the HBM ceiling of the padded bytes, 2 and 4 change nothing
    // (config 5: 0.410 / 0.414 / 0.422 ms, K <= 8 in 8 rows: 0.264 / 0.265 / 0.270 ms)
#ifndef C2D_POLY_PADDED_TILES_PER_WAVE
#define C2D_POLY_PADDED_TILES_PER_WAVE 1
#endif
    const size_t n_tiles = (n + 63) / 64;
    const size_t want = (n_tiles + C2D_POLY_PADDED_TILES_PER_WAVE - 1) / C2D_POLY_PADDED_TILES_PER_WAVE;  // the kernel strides by the grid
    const int grid = (int)(want < (size_t)kMaxGrid ? want : (size_t)kMaxGrid);
    hipLaunchKernelGGL((sat_poly_kernel<KM, MIN_WAVES, FULL>), dim3(grid), dim3(64), 0, s, d_vx, d_vy, d_k, n, rows, d_out, d_count,
                       workspace_count_ticket2(ctx, s, (size_t)grid, d_count != nullptr), async_err);  // a wave per 64 pairs: the two-level count
}

}  // namespace c2d

using namespace c2d;

extern "C" {

int c2d_sat_poly_pairs_rows(c2d_ctx* ctx, const float* d_vx, const float* d_vy, const uint8_t* d_k, size_t n, int rows,
                            uint8_t* d_out, unsigned long long* d_count, c2d_stream stream)
{
    if (!ctx) return C2D_ERR_INVALID_ARG;
    if (rows < 1 || rows > C2D_POLY_KMAX) return fail_arg(ctx, "c2d_sat_poly_pairs_rows: rows must be 1..C2D_POLY_KMAX");
    if (n == 0) return C2D_OK;
    if (!d_vx || !d_vy || !d_k || !d_out) return fail_arg(ctx, "c2d_sat_poly_pairs: NULL argument");
    DeviceGuard g(ctx->device);
    hipStream_t s = (hipStream_t)stream;
    if (int rc = workspace_acquire(ctx, s, d_count != nullptr)) return rc;
    uint32_t* err = ctx->d_async_err;
    auto aligned = [](const void* p, size_t a) { return (reinterpret_cast<uintptr_t>(p) & (a - 1)) == 0; };
    if (rows == 4 && n % 4 == 0 && aligned(d_vx, 16) && aligned(d_vy, 16) && aligned(d_k, 4) && aligned(d_out, 4)) {
        const size_t n_groups = n / 4;
        const size_t blocks = (n_groups + 63) / 64;
        const size_t grid = blocks < (size_t)kMaxGrid ? blocks : (size_t)kMaxGrid;
        hipLaunchKernelGGL(sat_poly4_kernel, dim3((unsigned)grid), dim3(64), 0, s, d_vx, d_vy, d_k, n, n_groups, d_out, d_count,
                           workspace_count_ticket(ctx, s, grid, d_count != nullptr), err);  // a wave per 256 pairs: the single-level count
    } else if (rows == 16) launch_poly<16, 5, true>(ctx, s, d_vx, d_vy, d_k, n, rows, d_out, d_count, err);
    else if (rows > 8) {
        // 9..15 rows: the binned kernel's 12- / 16-row instances with the layout as ONE bin (all rows requested at once, straight-line
        // phase 1: 12-row layouts 0.330 instead of 0.368 ms per 1e7 pairs); planes of 4 GiB and more stay with the generic instance
        if (launch_poly_onebin(ctx, s, d_vx, d_vy, d_k, n, rows, d_out, d_count, err) != C2D_OK)
            launch_poly<16, 5, false>(ctx, s, d_vx, d_vy, d_k, n, rows, d_out, d_count, err);
    }
    else if (rows > 4) launch_poly<8, 7, false>(ctx, s, d_vx, d_vy, d_k, n, rows, d_out, d_count, err);
    else launch_poly<4, 8, false>(ctx, s, d_vx, d_vy, d_k, n, rows, d_out, d_count, err);
    C2D_LAUNCH_CHECK(ctx);
    return C2D_OK;
}

int c2d_sat_poly_pairs(c2d_ctx* ctx, const float* d_vx, const float* d_vy, const uint8_t* d_k, size_t n, uint8_t* d_out,
                       unsigned long long* d_count, c2d_stream stream)
{
    return c2d_sat_poly_pairs_rows(ctx, d_vx, d_vy, d_k, n, C2D_POLY_KMAX, d_out, d_count, stream);
}

}  // extern "C"
