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
#include "c2d_count.hpp"

namespace c2d {

struct Planes16 { const float* p[16]; };
struct Planes10 { const float* p[10]; };
struct Planes8 { float* p[8]; };

constexpr int kBlock = 256;
constexpr int kWideBlock = 64;          // the 4-pairs-per-lane kernel runs one wave per block
constexpr int kMaxBlocks = kMaxGrid;     // beyond this a launch falls back to its grid-stride loop
typedef float f32x4 __attribute__((ext_vector_type(4)));

// ---- the pairs of one lane: certified fast path, wave-wide fall-back ----------------------------------------------------
// rect_collide_certified (c2d_math.hpp) evaluates four of the eight axes and certifies the other four from the overlaps it
// saw; a pair whose certificate fails is "thin".  Thin pairs are rare (config 2: a wave in a few thousand), so the wave votes:
// if any lane holds one, every lane evaluates its pairs again with all eight axes (rect_collide) — straight-line code, no
// divergence.  `build(e, launder, r1, r2)` fills pair e of the lane; in the fall-back it is asked to launder its inputs through
// an empty asm, otherwise the compiler keeps every projection of the fast path alive for reuse there (172 spilled dwords).
// Returns bit e = pair e collides.  Headline kernel 104.5 -> 103.3 us, pose format 85.9 -> 74.9 us per 1e7 pairs.
struct Plain {};
struct Laundered {};
C2D_DEV float launder(float x, Plain) { return x; }
C2D_DEV float launder(float x, Laundered)
{
    asm volatile("" : "+v"(x));
    return x;
}

template <int PAIRS, class Build>
C2D_DEV uint32_t collide_pairs(Build build)
{
    uint32_t bits = 0;
    bool any_thin = false;
#pragma unroll
    for (int e = 0; e < PAIRS; e++) {
        float r1[8], r2[8];
        build(e, Plain{}, r1, r2);
        bool thin;
        bits |= (rect_collide_certified(r1, r2, thin) ? 1u : 0u) << e;
        any_thin |= thin;
    }
    if (__ballot(any_thin) != 0ull) {
        bits = 0;
#pragma unroll
        for (int e = 0; e < PAIRS; e++) {
            float r1[8], r2[8];
            build(e, Laundered{}, r1, r2);
            bits |= (rect_collide(r1, r2) ? 1u : 0u) << e;
        }
    }
    return bits;
}

C2D_DEV uint32_t bits_to_bytes4(uint32_t b) { return (b & 1u) | ((b & 2u) << 7) | ((b & 4u) << 14) | ((b & 8u) << 21); }

// ---- rectangle pairs, vertex format ------------------------------------------
// VEC == 4: planes read as float4 (16 B / lane), results written as one dword.
// VEC == 1: scalar loads; used for the tail and for unaligned buffers.
#ifndef C2D_SAT_PRIO
#define C2D_SAT_PRIO 0
#endif
template <int VEC, int BLOCK>
__global__ __launch_bounds__(BLOCK, (VEC == 4 ? 5 : 1)) void sat_rect_verts_kernel(Planes16 P, size_t first, size_t n_groups,
                                                                uint8_t* __restrict__ out,
                                                                unsigned long long* __restrict__ d_count, CountWs words)
{
    uint32_t my_count = 0;
    const size_t stride = (size_t)gridDim.x * BLOCK;
    for (size_t g = (size_t)blockIdx.x * BLOCK + threadIdx.x; g < n_groups; g += stride) {
        if constexpr (VEC == 4) {
            f32x4 v[16];
#if C2D_SAT_PRIO == 1  // (experiment: the wave's address arithmetic and loads ahead of older waves' arithmetic)
            __builtin_amdgcn_s_setprio(3);
#endif
#pragma unroll
            for (int k = 0; k < 16; k++) v[k] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(P.p[k]) + g);
#if C2D_SAT_PRIO == 1
            __builtin_amdgcn_s_setprio(0);
#elif C2D_SAT_PRIO == 2  // (experiment: a wave whose data has arrived ahead of younger waves, so that it frees its slot sooner)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_setprio(3);
#endif
            const uint32_t packed = bits_to_bytes4(collide_pairs<4>([&v](int e, auto how, float (&r1)[8], float (&r2)[8]) {
#pragma unroll
                for (int k = 0; k < 8; k++) {
                    r1[k] = launder(v[k][e], how);
                    r2[k] = launder(v[8 + k][e], how);
                }
            }));
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

// ---- rectangle pairs, vertex format, bit-mask output -------------------------------------------
// For callers that only need the mask: one bit per pair (bit i & 63 of word i >> 6) instead of one byte, 64.125
// instead of 65 bytes per pair.  Wide form: the same loads and lane mapping as above (lane = 4 consecutive pairs), a
// wave covers 256 pairs = four 64-bit words; lane l holds bits 4(l & 15) .. +3 of word l >> 4, and an OR over each
// 16-lane row assembles the word.  The plain form (one pair per lane, one ballot per wave) takes what is left.
__global__ __launch_bounds__(64) void sat_rect_verts_mask4_kernel(Planes16 P, size_t n_groups, unsigned long long* __restrict__ mask,
                                                                  unsigned long long* __restrict__ d_count,
                                                                  CountWs words)
{
    const uint32_t lane = threadIdx.x;
    const size_t g = (size_t)blockIdx.x * 64 + lane;  // n_groups is a multiple of 16: a 16-lane row (one word) is all in or all out
    uint32_t nib = 0;
    if (g < n_groups) {
        f32x4 v[16];
#pragma unroll
        for (int k = 0; k < 16; k++) v[k] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(P.p[k]) + g);
        nib = collide_pairs<4>([&v](int e, auto how, float (&r1)[8], float (&r2)[8]) {
#pragma unroll
            for (int k = 0; k < 8; k++) {
                r1[k] = launder(v[k][e], how);
                r2[k] = launder(v[8 + k][e], how);
            }
        });
    }
    const uint32_t sh = 4u * (lane & 15u);
    uint32_t lo = sh < 32u ? nib << sh : 0u, hi = sh >= 32u ? nib << (sh - 32u) : 0u;
#pragma unroll
    for (int off = 1; off < 16; off <<= 1) {
        lo |= (uint32_t)__shfl_xor((int)lo, off, 64);
        hi |= (uint32_t)__shfl_xor((int)hi, off, 64);
    }
    if ((lane & 15u) == 0 && g < n_groups) __builtin_nontemporal_store(((unsigned long long)hi << 32) | lo, mask + ((size_t)blockIdx.x * 4 + (lane >> 4)));
    if (d_count) wave_count_arrive((uint32_t)__popc(nib), d_count, words);
}

__global__ __launch_bounds__(kBlock) void sat_rect_verts_mask1_kernel(Planes16 P, size_t first, size_t n, unsigned long long* __restrict__ mask,
                                                                      unsigned long long* __restrict__ d_count,
                                                                      CountWs words)
{
    // `first` is a multiple of 64 and every wave owns the 64 pairs of one word
    uint32_t my_count = 0;
    const size_t stride = (size_t)gridDim.x * kBlock;
    const size_t n_round = (n + 63) / 64 * 64;
    for (size_t i = first + (size_t)blockIdx.x * kBlock + threadIdx.x; i < n_round; i += stride) {
        uint32_t c = 0;
        if (i < n) {
            float r1[8], r2[8];
#pragma unroll
            for (int k = 0; k < 8; k++) {
                r1[k] = P.p[k][i];
                r2[k] = P.p[8 + k][i];
            }
            c = rect_collide(r1, r2) ? 1u : 0u;
        }
        const unsigned long long b = __ballot(c != 0);
        if ((threadIdx.x & 63) == 0) mask[i >> 6] = b;
        my_count += c;
    }
    if (d_count) wave_count_arrive(my_count, d_count, words);
}

// ---- rectangle pairs, array-of-rectangles format ----------------------------------------
// The reference's own argument layout, convex_collide(float* r1, float* r2) with flat
// float[8] rectangles (utils.cu:159), batched: r1, r2 are f32[n][8].  A lane reads its
// pair as 4 x 16 B; a wave covers 2 x 2 KiB of contiguous memory, so the loads coalesce
// as well as the SoA planes do (same 65 B/pair).
template <bool OUT16>
__global__ __launch_bounds__(64) void sat_rect_aos_kernel(const float* __restrict__ r1s, const float* __restrict__ r2s,
                                                         size_t n, uint8_t* __restrict__ out,
                                                         unsigned long long* __restrict__ d_count,
                                                         CountWs words)
{
    // A wave owns 128 consecutive pairs = 4 KiB of r1 and 4 KiB of r2.  Lane i fetches the 16-byte chunks i, 64 + i,
    // 128 + i and 192 + i of each array (eight fully coalesced 1-KiB loads in flight per lane), the chunks go through a
    // wave-private LDS tile, and lane p reads back its own two pairs 2p, 2p + 1 (chunks 4p .. 4p + 3) and stores both
    // result bytes at once.  Loading a lane's bytes directly (strided 16-byte loads) touches every cache line twice:
    // 120 vs 104 us.  One wave per block, no grid-stride loop, as for the plane-format kernel.
    constexpr int kChunks = 256;  // 16-byte chunks per array and tile
    __shared__ __attribute__((aligned(16))) f32x4 tile[2][kChunks];
    const int lane = threadIdx.x;
    uint32_t my_count = 0;
    const size_t n_tiles = (n + 127) / 128;
    for (size_t t = blockIdx.x; t < n_tiles; t += gridDim.x) {
        const size_t p0 = t * 128;
        const size_t chunks = (n - p0 < 128 ? n - p0 : 128) * 2;  // 16-byte chunks of this tile per array
        const f32x4* a = reinterpret_cast<const f32x4*>(r1s) + 2 * p0;
        const f32x4* b = reinterpret_cast<const f32x4*>(r2s) + 2 * p0;
        const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
        f32x4 va[4], vb[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const size_t c = (size_t)(64 * k + lane);
            va[k] = c < chunks ? __builtin_nontemporal_load(a + c) : zero;
            vb[k] = c < chunks ? __builtin_nontemporal_load(b + c) : zero;
        }
#pragma unroll
        for (int k = 0; k < 4; k++) {
            tile[0][64 * k + lane] = va[k];
            tile[1][64 * k + lane] = vb[k];
        }
        // same wave wrote and reads: LDS operations of a wave complete in order; the fence pair keeps the
        // compiler from moving a lane's reads above another lane's writes (it emits no instruction)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        f32x4 pa[4], pb[4];  // the lane's two pairs: chunks 4 lane .. 4 lane + 3 of each array
#pragma unroll
        for (int c = 0; c < 4; c++) {
            pa[c] = tile[0][4 * lane + c];
            pb[c] = tile[1][4 * lane + c];
        }
        const uint32_t bits = collide_pairs<2>([&pa, &pb](int e, auto how, float (&r1)[8], float (&r2)[8]) {
#pragma unroll
            for (int k = 0; k < 8; k++) {
                r1[k] = launder(pa[2 * e + k / 4][k % 4], how);
                r2[k] = launder(pb[2 * e + k / 4][k % 4], how);
            }
        });
        const uint32_t packed = (bits & 1u) | ((bits & 2u) << 7);
        const size_t i = p0 + 2 * (size_t)lane;
        if (i + 1 < n) {
            if constexpr (OUT16) {
                *reinterpret_cast<uint16_t*>(out + i) = (uint16_t)packed;  // i is even and `out` 2-byte aligned
            } else {
                out[i] = (uint8_t)(packed & 1u);
                out[i + 1] = (uint8_t)(packed >> 8);
            }
            my_count += (uint32_t)__popc(packed);
        } else if (i < n) {
            out[i] = (uint8_t)(packed & 1u);
            my_count += packed & 1u;
        }
        // the next tile's stores must not overtake this tile's loads
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    if (d_count) wave_count_arrive(my_count, d_count, words);
}

// ---- rectangle pairs, pose format (10 planes, 41 B/pair) ------------------------------
// Same lane mapping as the vertex kernel (VEC == 4: one float4 per plane, 4 pairs per lane).  Rebuilding both rectangles and
// running the vertex arithmetic costs 294 VALU instructions per 41 bytes (394 before the parallel-axis certificates), which
// made this format VALU-bound at the clock the chip holds under it (profiles/r02_pose_probe.txt, tools/pose_probe.hip).  A pose
// pair, however, is two rectangles BUILT from centre, rotation and extents, so the closed-form gap of the Monte-Carlo kernels
// applies (c2d_mc.hip, model_gap — the derivation and the error budget are there): per frame direction e
//     G_e = |e . (centre_2 - centre_1)| - sum of the four half axes' |e . U|,
// and |G_e| h_e > (64 h_e + 20 C) u C (h_e: half extent of the edge along e, C >= |centre coordinates| + half extents, u = 2^-24)
// fixes the reference's comparisons on the two edges along e.  One separating direction decides "no collision", four
// overlapping ones decide "collision"; anything else — about one pair in 10^4 on the bench workload, and every pair with a
// non-finite or out-of-range parameter — is thin, and the wave evaluates that pair slot with the vertex arithmetic
// (collide_pairs).  Per-direction margins (not one margin from the smallest extent) keep a sliver's large margin on its own axis.
C2D_DEV bool pose_pair_closed_form(const float (&v)[10], bool& thin)
{
    float s1, c1, s2, c2;
    sincos_(v[4], s1, c1);
    sincos_(v[9], s2, c2);
    const float hw = __builtin_fabsf(v[2] / 2), hh = __builtin_fabsf(v[3] / 2), hx = __builtin_fabsf(v[7] / 2), hy = __builtin_fabsf(v[8] / 2);
    const float q1 = fma_(c1, c2, s1 * s2), q2 = fma_(c1, s2, -(s1 * c2));
    const float ex = v[5] - v[0], ey = v[6] - v[1];
    const float t1 = fma_(c1, ex, s1 * ey), t2 = fma_(c1, ey, -(s1 * ex));
    const float t3 = fma_(c2, ex, s2 * ey), t4 = fma_(c2, ey, -(s2 * ex));
    const float a1 = __builtin_fabsf(q1), a2 = __builtin_fabsf(q2);
    const float g1 = __builtin_fabsf(t1) - fma_(hx, a1, fma_(hy, a2, hw));
    const float g2 = __builtin_fabsf(t2) - fma_(hx, a2, fma_(hy, a1, hh));
    const float g3 = __builtin_fabsf(t3) - fma_(hw, a1, fma_(hh, a2, hx));
    const float g4 = __builtin_fabsf(t4) - fma_(hw, a2, fma_(hh, a1, hy));
    const float C = __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(v[0]), __builtin_fabsf(v[1])) + (hw + hh),
                                    __builtin_fmaxf(__builtin_fabsf(v[5]), __builtin_fabsf(v[6])) + (hx + hy)) * (1.0f + 0x1p-10f);
    const float uc = (0x1p-24f * (1.0f + 0x1p-10f)) * C, k64 = 64.0f * uc, k20 = (20.0f * uc) * C;
    const float z1 = g1 * hw, z2 = g2 * hh, z3 = g3 * hx, z4 = g4 * hy;
    const float m1 = fma_(k64, hw, k20), m2 = fma_(k64, hh, k20), m3 = fma_(k64, hx, k20), m4 = fma_(k64, hy, k20);
    const float over = __builtin_fmaxf(__builtin_fmaxf(z1 - m1, z2 - m2), __builtin_fmaxf(z3 - m3, z4 - m4));   // > 0: some direction separates
    const float under = __builtin_fmaxf(__builtin_fmaxf(z1 + m1, z2 + m2), __builtin_fmaxf(z3 + m3, z4 + m4));  // < 0: every direction overlaps
    // the max / min instructions drop a NaN operand, the sum does not: any non-finite parameter makes it non-finite
    const float all = (g1 + g2) + (g3 + g4);
    const float hmin = __builtin_fminf(__builtin_fminf(hw, hh), __builtin_fminf(hx, hy));
    const bool regular = (C < 1e15f) & (hmin >= 1e-12f) & (__builtin_fabsf(all) < 1e37f);
    const bool sep = over > 0.0f, col = under < 0.0f;
    thin = !(regular & (sep | col));
    return col;
}

C2D_DEV void pose_pair_rects(const float (&v)[10], float (&r1)[8], float (&r2)[8])
{
    float s, c;
    sincos_(v[4], s, c);
    rect_from_half_extents(v[2] / 2, v[3] / 2, c, s, v[0], v[1], r1);
    sincos_(v[9], s, c);
    rect_from_half_extents(v[7] / 2, v[8] / 2, c, s, v[5], v[6], r2);
}

C2D_DEV uint32_t pose_pair_collides(const float (&v)[10])
{
    float r1[8], r2[8];
    pose_pair_rects(v, r1, r2);
    return rect_collide(r1, r2) ? 1u : 0u;
}

template <int VEC, int BLOCK>
__global__ __launch_bounds__(BLOCK, (VEC == 4 ? 5 : 1)) void sat_rect_pose_kernel(Planes10 P, size_t first, size_t n_groups,
                                                              uint8_t* __restrict__ out,
                                                              unsigned long long* __restrict__ d_count,
                                                              CountWs words)
{
    uint32_t my_count = 0;
    const size_t stride = (size_t)gridDim.x * BLOCK;
    for (size_t g = (size_t)blockIdx.x * BLOCK + threadIdx.x; g < n_groups; g += stride) {
        if constexpr (VEC == 4) {
            f32x4 q[10];
#pragma unroll
            for (int k = 0; k < 10; k++) q[k] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(P.p[k]) + g);
            uint32_t bits = 0;
#pragma unroll
            for (int e = 0; e < 4; e++) {
                float v[10];
#pragma unroll
                for (int k = 0; k < 10; k++) v[k] = q[k][e];
                bool thin;
                uint32_t hit = pose_pair_closed_form(v, thin) ? 1u : 0u;
                if (__ballot(thin) != 0ull) {  // this pair slot with the vertex arithmetic, for the whole wave
                    hit = collide_pairs<1>([&q, e](int, auto, float (&r1)[8], float (&r2)[8]) {
                        float w[10];
#pragma unroll
                        for (int k = 0; k < 10; k++) w[k] = launder(q[k][e], Laundered{});
                        pose_pair_rects(w, r1, r2);
                    });
                }
                bits |= hit << e;
            }
            const uint32_t packed = bits_to_bytes4(bits);
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

// wide form: a lane builds 4 consecutive rectangles, 5 x 16-byte loads and 8 x 16-byte stores (520 B per lane);
// single-wave blocks without a grid-stride loop, as for the SAT kernels
__global__ __launch_bounds__(kWideBlock) void rects_from_poses4_kernel(const float* __restrict__ cx, const float* __restrict__ cy,
                                                                       const float* __restrict__ w, const float* __restrict__ h,
                                                                       const float* __restrict__ th, size_t n_groups, Planes8 O)
{
    const size_t stride = (size_t)gridDim.x * kWideBlock;
    for (size_t g = (size_t)blockIdx.x * kWideBlock + threadIdx.x; g < n_groups; g += stride) {
        const f32x4 vcx = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(cx) + g);
        const f32x4 vcy = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(cy) + g);
        const f32x4 vw = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(w) + g);
        const f32x4 vh = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(h) + g);
        const f32x4 vt = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(th) + g);
        f32x4 o[8];
#pragma unroll
        for (int e = 0; e < 4; e++) {
            float r[8], s, c;
            sincos_(vt[e], s, c);
            rect_from_half_extents(vw[e] / 2, vh[e] / 2, c, s, vcx[e], vcy[e], r);
#pragma unroll
            for (int k = 0; k < 8; k++) o[k][e] = r[k];
        }
#pragma unroll
        for (int k = 0; k < 8; k++) __builtin_nontemporal_store(o[k], reinterpret_cast<f32x4*>(O.p[k]) + g);
    }
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
    bool wide = aligned_to(d_cx, 16) && aligned_to(d_cy, 16) && aligned_to(d_w, 16) && aligned_to(d_h, 16) && aligned_to(d_theta, 16);
    for (int k = 0; k < 8; k++) wide = wide && aligned_to(O.p[k], 16);
    const size_t n4 = wide ? n / 4 : 0;
    if (n4) {
        hipLaunchKernelGGL(rects_from_poses4_kernel, dim3(grid_for(n4, kWideBlock, kMaxBlocks)), dim3(kWideBlock), 0, (hipStream_t)stream, d_cx, d_cy,
                           d_w, d_h, d_theta, n4, O);
        C2D_LAUNCH_CHECK(ctx);
    }
    const size_t rest = n - 4 * n4;
    if (rest) {
        Planes8 T;
        for (int k = 0; k < 8; k++) T.p[k] = O.p[k] + 4 * n4;
        hipLaunchKernelGGL(rects_from_poses_kernel, dim3(grid_for(rest, kBlock, ctx->prop.multiProcessorCount * 8)), dim3(kBlock), 0,
                           (hipStream_t)stream, d_cx + 4 * n4, d_cy + 4 * n4, d_w + 4 * n4, d_h + 4 * n4, d_theta + 4 * n4, rest, T);
        C2D_LAUNCH_CHECK(ctx);
    }
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
    if (int rc = workspace_acquire(ctx, s, d_count != nullptr)) return rc;
    // One group of 4 pairs per lane, 64-thread blocks, no grid-stride loop below kMaxBlocks:
    // short single-wave blocks retiring all through the launch stream better than a
    // resident grid-stride grid (tools/sat_tune: 103.0 vs 109.3 us per 1e7 pairs).
    const size_t n4 = wide ? n / 4 : 0;
    if (n4) {
        const int grid = grid_for(n4, kWideBlock, kMaxBlocks);
        hipLaunchKernelGGL((sat_rect_verts_kernel<4, kWideBlock>), dim3(grid), dim3(kWideBlock), 0, s, P, (size_t)0, n4, d_out,
                           d_count, workspace_count_ticket(ctx, s, (size_t)grid * (kWideBlock / 64), d_count != nullptr));
        C2D_LAUNCH_CHECK(ctx);
    }
    const size_t rest = n - 4 * n4;
    if (rest) {
        const int grid = grid_for(rest, kBlock, kMaxBlocks);
        hipLaunchKernelGGL((sat_rect_verts_kernel<1, kBlock>), dim3(grid), dim3(kBlock), 0, s, P, 4 * n4, rest, d_out, d_count,
                           workspace_count_ticket(ctx, s, (size_t)grid * (kBlock / 64), d_count != nullptr));
        C2D_LAUNCH_CHECK(ctx);
    }
    return C2D_OK;
}

int c2d_sat_rect_pairs_verts_mask(c2d_ctx* ctx, const float* const d_planes[16], size_t n, unsigned long long* d_mask,
                                  unsigned long long* d_count, c2d_stream stream)
{
    if (!ctx) return C2D_ERR_INVALID_ARG;
    if (n == 0) return C2D_OK;
    if (!d_planes || !d_mask) return fail_arg(ctx, "c2d_sat_rect_pairs_verts_mask: NULL argument");
    if (!aligned_to(d_mask, 8)) return fail_arg(ctx, "c2d_sat_rect_pairs_verts_mask: the mask must be 8-byte aligned");
    Planes16 P;
    bool wide = true;
    for (int k = 0; k < 16; k++) {
        if (!d_planes[k]) return fail_arg(ctx, "c2d_sat_rect_pairs_verts_mask: NULL plane");
        P.p[k] = d_planes[k];
        wide = wide && aligned_to(d_planes[k], 16);
    }
    DeviceGuard g(ctx->device);
    hipStream_t s = (hipStream_t)stream;
    if (int rc = workspace_acquire(ctx, s, d_count != nullptr)) return rc;
    size_t done = 0;
    while (wide && n - done >= 64) {  // whole 64-bit words; more than kMaxBlocks waves go in several launches
        size_t words_left = (n - done) / 64;
        if (words_left > (size_t)kMaxBlocks * 4) words_left = (size_t)kMaxBlocks * 4;
        Planes16 Q;
        for (int k = 0; k < 16; k++) Q.p[k] = P.p[k] + done;
        const size_t groups = words_left * 16;
        hipLaunchKernelGGL(sat_rect_verts_mask4_kernel, dim3((unsigned)((groups + 63) / 64)), dim3(64), 0, s, Q, groups, d_mask + done / 64, d_count,
                           workspace_count_ticket(ctx, s, (groups + 63) / 64, d_count != nullptr));
        C2D_LAUNCH_CHECK(ctx);
        done += words_left * 64;
    }
    if (done < n) {
        const int grid = grid_for(n - done, kBlock, kMaxBlocks);
        hipLaunchKernelGGL(sat_rect_verts_mask1_kernel, dim3(grid), dim3(kBlock), 0, s, P, done, n, d_mask, d_count,
                           workspace_count_ticket(ctx, s, (size_t)grid * (kBlock / 64), d_count != nullptr));
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
    if (int rc = workspace_acquire(ctx, (hipStream_t)stream, d_count != nullptr)) return rc;
    const int grid = grid_for(n, 128, kMaxBlocks);  // one 128-pair tile per single-wave block
    const CountWs ws = workspace_count_ticket(ctx, (hipStream_t)stream, (size_t)grid, d_count != nullptr);
    if (aligned_to(d_out, 2))
        hipLaunchKernelGGL(sat_rect_aos_kernel<true>, dim3(grid), dim3(64), 0, (hipStream_t)stream, d_r1, d_r2, n, d_out, d_count, ws);
    else
        hipLaunchKernelGGL(sat_rect_aos_kernel<false>, dim3(grid), dim3(64), 0, (hipStream_t)stream, d_r1, d_r2, n, d_out, d_count, ws);
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
    if (int rc = workspace_acquire(ctx, s, d_count != nullptr)) return rc;
    const size_t n4 = wide ? n / 4 : 0;
    if (n4) {
        const int grid = grid_for(n4, kWideBlock, kMaxBlocks);
        hipLaunchKernelGGL((sat_rect_pose_kernel<4, kWideBlock>), dim3(grid), dim3(kWideBlock), 0, s, P, (size_t)0, n4, d_out,
                           d_count, workspace_count_ticket(ctx, s, (size_t)grid * (kWideBlock / 64), d_count != nullptr));
        C2D_LAUNCH_CHECK(ctx);
    }
    const size_t rest = n - 4 * n4;
    if (rest) {
        const int grid = grid_for(rest, kBlock, kMaxBlocks);
        hipLaunchKernelGGL((sat_rect_pose_kernel<1, kBlock>), dim3(grid), dim3(kBlock), 0, s, P, 4 * n4, rest, d_out, d_count,
                           workspace_count_ticket(ctx, s, (size_t)grid * (kBlock / 64), d_count != nullptr));
        C2D_LAUNCH_CHECK(ctx);
    }
    return C2D_OK;
}

}  // extern "C"
