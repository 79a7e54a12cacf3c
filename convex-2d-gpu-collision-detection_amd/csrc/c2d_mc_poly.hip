// c2d_mc_poly.hip — Monte-Carlo collision probability for convex polygons on gfx950 (MI355X).
//
// The reference's README (README.md:3) says the code "can easily be extended to handle arbitrary convex 2D shapes"; its
// own functions stop at rectangles (sample_rectangle utils.cu:144-157, convex_collide utils.cu:159-184, the loop of
// compute_collision_probability.cu:119-139).  This file is that extension — c2d_mc_poly_pair / c2d_mc_poly_scenes of
// include/c2d.h — built on the sample loops of the rectangle kernels (c2d_mc_core.hpp: draw layout, NEAR / FAR paths, queues,
// adaptive schedule), with its own scene, certain-miss pretest and evaluation:
//
//   scene      the robot polygon, placed (utils.cu:132-142 per vertex), lives in the wave's LDS block as one float4 per edge
//              (true normal, the robot's own projection interval on it) plus its vertices; the obstacle's base vertices beside it;
//   pretest    every vertex of a sampled obstacle lies within rho of the sampled centre, so a centre outside the robot's interval
//              on some edge normal, widened by |n| rho and a rounding allowance, proves the miss (the polygon form of the
//              rectangle kernels' bounding-disk pretest), and so does a small enough Box-Muller radius word (x0);
//   evaluation one sample per lane: the lane transforms the obstacle's vertices into registers (scale, rotate, move — the
//              arithmetic of the oracle's sample_polygon, float for float) and runs the interval test of utils.cu:172-180 on
//              the true normals of all ka + kb edges; robot data arrive as LDS broadcasts (every lane reads the same address).
#include <type_traits>

#include "c2d_internal.hpp"
#include "c2d_math.hpp"

// ---- census build (-DC2D_MC_STATS, `make lib-mcstats`; never the product), as in c2d_mc.hip: words 0-4 and 7 are the sample loops'
// (samples, far path, near path, radius candidates, centres evaluated, hits); here [5] samples that reach the evaluation, [6] of
// those the survivors of the robot's normals, [8] samples whose obstacle normals ran in place, [9] ... out of the survivor queue.
#ifdef C2D_MC_STATS
namespace c2d {
__device__ unsigned long long c2d_mc_poly_stats_words[12];
}
#define C2D_MC_STAT(i, v)                                                                                     \
    do {                                                                                                      \
        const unsigned long long v__ = (unsigned long long)(v);                                               \
        if ((threadIdx.x & 63) == 0 && v__) atomicAdd(&c2d::c2d_mc_poly_stats_words[i], v__);                 \
    } while (0)
#endif
#include "c2d_mc_core.hpp"

namespace c2d {

#ifndef C2D_MC_POLY_PLAIN_PAIR_LOOP
#define C2D_MC_POLY_PLAIN_PAIR_LOOP 0   // (A/B switch, round 5: the loop over the robot's vertex pairs without the one-ahead read)
#endif
constexpr int KM = C2D_POLY_KMAX;
#ifndef C2D_MC_POLY_IN_PLACE
#define C2D_MC_POLY_IN_PLACE 48  // survivors of a pass from which stage B runs in place: the queue (two LDS round trips and the vertices
#endif                           // built a second time) costs more than the idle lanes it would fill (32 / 40 / 48: profiles/notes_r04_mc_poly.md)
// Survivor queue: at most 63 left over + at most C2D_MC_POLY_IN_PLACE - 1 pushed by one evaluation pass (a pass with more
// survivors goes on in place).  112 slots instead of round 4's 128 bring the wave's LDS block from 8 208 to 7 824 bytes, so that
// the 20 single-wave blocks per CU that __launch_bounds__(64, 5) asks for fit the 160 KB (round 4: 19).
#ifndef C2D_MC_POLY_SURV_SLOTS
#define C2D_MC_POLY_SURV_SLOTS ((64 + C2D_MC_POLY_IN_PLACE + 15) / 16 * 16)
#endif
constexpr int kSurvSlots = C2D_MC_POLY_SURV_SLOTS;
static_assert(63 + C2D_MC_POLY_IN_PLACE - 1 < kSurvSlots, "survivor queue too small for a full push");

// wave-uniform description of a polygon scene; the bulk lives in PolyQueue
struct PolyScene {
    float sx, sy, st, sw, sh;  // standard deviations (StdDev; width / height relative, include/c2d.h)
    uint32_t x0;               // radius-word form of the pretest (c2d_mc_core.hpp: radius_word_threshold)
    bool use_x0;
    int ka, kb;                // vertex counts, 1..KM
    C2D_DEV bool tame() const { return use_x0 || x0 != 0u; }
};

// the wave's LDS block.  Slots at and above a polygon's count repeat its vertex 0, which is exactly neutral: a zero-length
// edge never separates (every projection on its normal is +-0) and a repeated projection changes no extreme.
struct PolyQueue {
    float4 axis[KM];    // robot edge i: true normal (nx, ny), the robot's own interval [rmin, rmax] on it (utils.cu:176)
    float4 pre[KM];     // centre pretest on that normal: certain miss if T < plo or T > phi; .z = the robot's FIRST projection
    float4 ctr[KM];     // what the centre pretest reads, in one piece: (nx, ny, plo, phi)
    float2 rvert[KM];   // robot vertices, placed
    float2 overt[KM];   // obstacle vertices in the obstacle frame
    // survivors of the first stage of an evaluation (no robot normal separates them), waiting for the second on 64 busy lanes
    float4 surv_a[kSurvSlots];   // dx, dy, cos, sin of the sampled pose
    float2 surv_b[kSurvSlots];   // the two scale factors
    uint32_t n_surv;             // wave-uniform; 0 whenever a sample range starts or ends
    SampleQueues sq;
};

C2D_DEV float wave_max(float v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = __builtin_fmaxf(v, __shfl_xor(v, off, 64));
    return v;
}

// value of a[lane & 15] for an array that lives in scalar registers (kernel arguments): sixteen selects, once per scene
C2D_DEV float lane_pick(const float (&a)[KM], uint32_t l)
{
    float v = a[0];
#pragma unroll
    for (int k = 1; k < KM; k++) v = l == (uint32_t)k ? a[k] : v;
    return v;
}

// ---- scene set-up (ccp.cu:119-133 for polygons).  Lane l & 15 holds vertex l & 15 of the robot (robot frame: rvx, rvy) and of
// the obstacle (ovx, ovy); values of lanes at and above a polygon's count are ignored.  Every lane of the wave calls this.
//
// Certain-miss pretest (the derivation of c2d_mc.hip make_scene_values, for Ka axes and a general shape).  A sampled vertex is
// centre + w with centre = (dx, dy) and w = R(dtheta) (fx x_k, fy y_k), |fx| <= 1 + kNormalMax |sw| =: fxm, likewise fym, so
// |w| <= rho = max_k sqrt((fxm x_k)^2 + (fym y_k)^2) whatever the rotation.  On the robot's edge normal a the SAT compares the
// obstacle's computed projections a.o_k with the robot's own interval [rmin, rmax] (the floats computed below), and with
// T = a.centre
//     a.o_k  >=  T - |a|_2 rho - err,     a.o_k  <=  T + |a|_2 rho + err,
// err = the roundings on the way: vertex construction (scale, two products, their sum, the translation) <= 5u (rho + D) per
// coordinate, the projection's two products and sum <= 3u |a|_1 (rho + D), the evaluation of T <= 3u |a|_1 D; u = 2^-24,
// D >= |dx| + |dy|.  The margin M grants 2^-10 |a|_2 rho + 2^-12 |a|_1 (rho + D) — over 300 times that — and the thresholds
// move outwards by another 2^-20 relative.  Hence T > phi or T < plo on ANY robot edge normal proves that the reference's test
// finds that axis separating.  The radius-word form: |T_i| <= rad G_i (1 + 2^-10), G_i = sqrt((a_x sx)^2 + (a_y sy)^2); when the
// widened interval of edge i does not contain 0, rad < L_i / G_i already proves the miss, and R0 = max_i L_i / G_i.
C2D_DEV PolyScene build_poly_scene(float rvx, float rvy, int ka, float px, float py, float theta, float ovx, float ovy, int kb, const StdDev& sd,
                                   PolyQueue& q)
{
    const uint32_t lane = threadIdx.x & 63, l = lane & 15;
    PolyScene sc;
    sc.sx = sd.x; sc.sy = sd.y; sc.st = sd.theta; sc.sw = sd.width; sc.sh = sd.height;
    sc.ka = ka; sc.kb = kb;
    // padding: slots >= count repeat vertex 0
    const float rx0 = __shfl(rvx, 0, 64), ry0 = __shfl(rvy, 0, 64), ox0 = __shfl(ovx, 0, 64), oy0 = __shfl(ovy, 0, 64);
    const float rxl = (int)l < ka ? rvx : rx0, ryl = (int)l < ka ? rvy : ry0;
    const float oxl = (int)l < kb ? ovx : ox0, oyl = (int)l < kb ? ovy : oy0;
    // robot placement: rot_trans_rectangle per vertex (utils.cu:132-142; ccp.cu:132-133)
    float s, c;
    sincos_(theta, s, c);
    const float wx = dot2(c, rxl, -s, ryl) + px, wy = dot2(s, rxl, c, ryl) + py;
    wave_lds_sync();  // (earlier readers of this block are done)
    if (lane < KM) {
        q.rvert[l] = make_float2(wx, wy);
        q.overt[l] = make_float2(oxl, oyl);
    }
    if (lane == 0) q.n_surv = 0;
    wave_lds_sync();
    // edge l: from slot l to slot l + 1 (mod 16); with the padding above that is the polygon's closing edge for l = ka - 1 and a
    // zero-length edge beyond it
    const float2 v1 = q.rvert[(l + 1) & 15];
    const float nx = -(v1.y - wy), ny = v1.x - wx;  // true normal (-ey, ex), as the oracle's poly_collide
    float rmin = 0.0f, rmax = 0.0f, p_first = 0.0f;
#pragma unroll
    for (int k = 0; k < KM; k++) {
        const float2 v = q.rvert[k];
        const float p = nx * v.x + ny * v.y;  // unfused (-ffp-contract=off), utils.cu:173
        if (k == 0) { rmin = rmax = p_first = p; }
        else { rmin = __builtin_fminf(rmin, p); rmax = __builtin_fmaxf(rmax, p); }
    }
    // ---- pretest constants
    const float fxm = 1.0f + kNormalMax * __builtin_fabsf(sc.sw), fym = 1.0f + kNormalMax * __builtin_fabsf(sc.sh);
    const float ex = fxm * oxl, ey = fym * oyl;
    const float rho = wave_max(__builtin_sqrtf(ex * ex + ey * ey)) * (1.0f + 0x1p-10f);
    const float D = kNormalMax * (__builtin_fabsf(sc.sx) + __builtin_fabsf(sc.sy));
    const float n2 = __builtin_sqrtf(nx * nx + ny * ny), n1 = __builtin_fabsf(nx) + __builtin_fabsf(ny);
    const float M = (n2 * rho) * (1.0f + 0x1p-10f) + 0x1p-12f * (n1 * (rho + D));
    const float hi = rmax + M, lo = rmin - M;
    float phi = hi + 0x1p-20f * __builtin_fabsf(hi), plo = lo - 0x1p-20f * __builtin_fabsf(lo);
    const bool real_edge = (int)l < ka && n1 > 0.0f;
    if (!real_edge) { phi = __builtin_inff(); plo = -__builtin_inff(); }
    float r0 = 0.0f;
    {
        const float gx = nx * sc.sx, gy = ny * sc.sy;
        const float G = __builtin_sqrtf(gx * gx + gy * gy) * (1.0f + 0x1p-10f);
        const float L = plo > 0.0f ? plo : (phi < 0.0f ? -phi : 0.0f);
        if (real_edge && L > 0.0f && G > 0.0f && L < 1e30f) r0 = L / G;  // (G == 0: T_i = 0 and the centre pretest itself decides)
    }
    const float R0 = wave_max(r0);
    if (lane < KM) {
        q.axis[l] = make_float4(nx, ny, rmin, rmax);
        q.pre[l] = make_float4(plo, phi, p_first, 0.0f);
        q.ctr[l] = make_float4(nx, ny, plo, phi);
    }
    wave_lds_sync();
    // ---- tame: every parameter finite (a NaN compares false) and small enough for NO intermediate of an evaluation to overflow —
    // the tame path's extremes are v_min / v_max, which agree with the comparison loop of minmax_element only where no NaN arises,
    // and an inf - inf makes one.  Lengths (vertices, position, sigma_x, sigma_y) below 1e8 and relative shape deviations below 1e4:
    // a scale factor stays below 7e4, a sampled coordinate below 1.4e13, a normal below 2.8e13, a projection below 1e27.
    // Angles only pass through the sine and cosine.  Everything else takes one sample per lane with the all-bit-patterns test.
    // Nor may a nonzero length be so small that a product of two lengths leaves the normal range (below 1e-15): the margins of the
    // pretest are relative rounding bounds, which a denormal product does not obey (scenes scaled by 1e-22 differed from the oracle).
    auto below = [](float v, float bound) { return __builtin_fabsf(v) < bound; };
    auto length = [](float v) { const float a = __builtin_fabsf(v); return a < 1e8f && (a == 0.0f || a >= 1e-15f); };
    const float rel = 1e4f, ang = 1e15f;
    const bool mine = ((int)l >= ka || (length(rvx) && length(rvy))) && ((int)l >= kb || (length(ovx) && length(ovy)));
    const bool tame = __builtin_amdgcn_ballot_w64(!mine) == 0ull && length(px) && length(py) && below(theta, ang) && length(sd.x) && length(sd.y) &&
                      below(sd.theta, ang) && below(sd.width, rel) && below(sd.height, rel);
    sc.use_x0 = false;
    sc.x0 = 0xffffffffu;
    if (!tame) {
        sc.x0 = 0u;  // the mark (PolyScene::tame)
        return sc;
    }
    radius_word_threshold(R0, sc.x0, sc.use_x0);
    return sc;
}

// min / max of a running extreme and two new values.  The extremes of a rolled loop arrive through a phi, of which the compiler
// cannot know that it is no signalling NaN, so that it canonicalises the value (v_max_f32 x, x) before every v_min / v_max: two
// more instructions per projection in the loop over the robot's vertices, which is a fifth of a polygon evaluation.  A tame
// scene has no NaN anywhere (and arithmetic never produces a signalling one), so there the instruction is written out; the
// all-bit-patterns path keeps the builtins, whose result for quiet NaNs is the one the oracle's comparison loop gives.
template <bool NANS>
C2D_DEV float min3_(float a, float b, float c)
{
    if constexpr (NANS) return __builtin_fminf(__builtin_fminf(a, b), c);
    float r;
    asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
template <bool NANS>
C2D_DEV float max3_(float a, float b, float c)
{
    if constexpr (NANS) return __builtin_fmaxf(__builtin_fmaxf(a, b), c);
    float r;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

// ---- evaluation of one sample per lane, in two stages.  Both build the sampled obstacle in registers (the oracle's
// sample_polygon: scale, rotate, move — the same floats both times) and run the interval test of utils.cu:172-180 on true normals:
//   stage A  the robot's ka normals (its own interval is wave-uniform): 4 kb instructions per normal;
//   stage B  the obstacle's kb normals, four at a time (own interval from the registers, the robot's vertices as LDS broadcasts):
//            4 (ka + kb) per normal — two thirds to three quarters of an evaluation.
// The result is an OR over normals, so a sample that stage A separates is decided: only its survivors need stage B, and they are
// queued until 64 have gathered (PolyPolicy::evaluate) so that stage B always runs on busy lanes.  CB = the obstacle's vertex slots
// held in registers (kb rounded up to an even number; a padding slot is neutral).  NANS: the scene is not tame, so a projection may
// be a NaN and the comparison-based extremes of thrust::minmax_element (utils.cu:176-177) must be followed: one unordered compare
// of the two first projections per axis (first_projections_ordered, c2d_math.hpp).
template <int CB>
C2D_DEV void poly_sampled_vertices(const PolyQueue& q, float dx, float dy, float c, float s, float fx, float fy, float (&ox)[CB], float (&oy)[CB])
{
#pragma unroll
    for (int k = 0; k < CB; k++) {
        const float2 b = q.overt[k];
        const float x = fx * b.x, y = fy * b.y;        // utils.cu:152-155: the shape changes first
        ox[k] = dot2(c, x, -s, y) + dx;                // utils.cu:139
        oy[k] = dot2(s, x, c, y) + dy;                 // utils.cu:140
    }
}

// stage A: the lanes of `lanes` that no robot normal separates
template <int CB, bool NANS>
C2D_DEV unsigned long long poly_stage_robot(const PolyScene& sc, const PolyQueue& q, const float (&ox)[CB], const float (&oy)[CB], unsigned long long lanes)
{
    unsigned long long sep = 0ull;
    const int ka = sc.ka;
#pragma nounroll
    for (int i = 0; i < ka; i++) {
        const float4 A = q.axis[i];
        float p0 = A.x * ox[0] + A.y * oy[0];
        float omin = p0, omax = p0;
#pragma unroll
        for (int k = 1; k < CB; k++) {
            const float p = A.x * ox[k] + A.y * oy[k];
            omin = __builtin_fminf(omin, p);
            omax = __builtin_fmaxf(omax, p);
        }
        unsigned long long m = __builtin_amdgcn_ballot_w64(A.w < omin) | __builtin_amdgcn_ballot_w64(omax < A.z);  // max1 < min2 || max2 < min1
        if constexpr (NANS) m &= __builtin_amdgcn_ballot_w64(first_projections_ordered(q.pre[i].z, p0));
        sep |= m;
    }
    return lanes & ~sep;
}

// stage B: the lanes of `lanes` that no obstacle normal separates
template <int CB, bool NANS>
C2D_DEV unsigned long long poly_stage_obstacle(const PolyScene& sc, const PolyQueue& q, const float (&ox)[CB], const float (&oy)[CB], unsigned long long lanes)
{
    unsigned long long sep = 0ull;
    const int ka = sc.ka;
    // four normals at a time (the last group of a class that is not a multiple of four: two)
    auto group = [&](auto g_const, auto j0_const) {
        constexpr int G = decltype(g_const)::value, j0 = decltype(j0_const)::value;
        float nx[G], ny[G], mn1[G], mx1[G], mn2[G], mx2[G], q0[G], r0[G];
#pragma unroll
        for (int a = 0; a < G; a++) {
            const int j = j0 + a, j1 = (j + 1) % CB;
            nx[a] = -(oy[j1] - oy[j]);
            ny[a] = ox[j1] - ox[j];
            q0[a] = nx[a] * ox[0] + ny[a] * oy[0];
            mn2[a] = mx2[a] = q0[a];
        }
#pragma unroll
        for (int k = 1; k < CB; k++) {
#pragma unroll
            for (int a = 0; a < G; a++) {
                const float p = nx[a] * ox[k] + ny[a] * oy[k];
                mn2[a] = __builtin_fminf(mn2[a], p);
                mx2[a] = __builtin_fmaxf(mx2[a], p);
            }
        }
        // the robot's vertices, two per step from one 16-byte LDS read, the next pair in flight while this one is used (slots at and
        // above ka repeat vertex 0, which is neutral, so an odd count reads one slot more)
        const float4* rv2 = reinterpret_cast<const float4*>(q.rvert);  // (x0, y0, x1, y1)
        const int pairs = (ka + 1) >> 1;
#if C2D_MC_POLY_PLAIN_PAIR_LOOP
        float4 cur = rv2[0];
#else
        float4 cur = rv2[0];
        float4 nxt = rv2[pairs > 1 ? 1 : 0];
#endif
#pragma unroll
        for (int a = 0; a < G; a++) {
            r0[a] = nx[a] * cur.x + ny[a] * cur.y;
            const float p1 = nx[a] * cur.z + ny[a] * cur.w;
            mn1[a] = __builtin_fminf(r0[a], p1);
            mx1[a] = __builtin_fmaxf(r0[a], p1);
        }
#pragma nounroll
        for (int j = 1; j < pairs; j++) {
#if C2D_MC_POLY_PLAIN_PAIR_LOOP
            cur = rv2[j];
#else
            cur = nxt;
            nxt = rv2[j + 1 < pairs ? j + 1 : j];  // (the last step reads its own pair again)
#endif
#pragma unroll
            for (int a = 0; a < G; a++) {
                const float p0 = nx[a] * cur.x + ny[a] * cur.y, p1 = nx[a] * cur.z + ny[a] * cur.w;
                mn1[a] = min3_<NANS>(mn1[a], p0, p1);
                mx1[a] = max3_<NANS>(mx1[a], p0, p1);
            }
        }
#pragma unroll
        for (int a = 0; a < G; a++) {
            unsigned long long m = __builtin_amdgcn_ballot_w64(mx1[a] < mn2[a]) | __builtin_amdgcn_ballot_w64(mx2[a] < mn1[a]);
            if constexpr (NANS) m &= __builtin_amdgcn_ballot_w64(first_projections_ordered(r0[a], q0[a]));
            sep |= m;
        }
    };
    using std::integral_constant;
    if constexpr (CB >= 4) { group(integral_constant<int, 4>{}, integral_constant<int, 0>{}); if ((lanes & ~sep) == 0ull) return 0ull; }
    if constexpr (CB >= 8) { group(integral_constant<int, 4>{}, integral_constant<int, 4>{}); if ((lanes & ~sep) == 0ull) return 0ull; }
    if constexpr (CB >= 12) { group(integral_constant<int, 4>{}, integral_constant<int, 8>{}); if ((lanes & ~sep) == 0ull) return 0ull; }
    if constexpr (CB >= 16) { group(integral_constant<int, 4>{}, integral_constant<int, 12>{}); if ((lanes & ~sep) == 0ull) return 0ull; }
    if constexpr (CB % 4 != 0) group(integral_constant<int, CB % 4>{}, integral_constant<int, CB - CB % 4>{});
    return lanes & ~sep;
}

// STAGE = 0: both stages in place (the plain path); 1: stage A, and stage B in place when most of the pass survives (`final`
// says which: true = the returned lanes are the colliding ones, false = they are stage A's survivors); 2: stage B only.
template <int CB, bool NANS, int STAGE>
C2D_DEV unsigned long long poly_sample_stage(const PolyScene& sc, const PolyQueue& q, float dx, float dy, float c, float s, float fx, float fy,
                                             unsigned long long lanes, bool& final)
{
    float ox[CB], oy[CB];
    poly_sampled_vertices<CB>(q, dx, dy, c, s, fx, fy, ox, oy);
    unsigned long long alive = lanes;
    final = true;
    if constexpr (STAGE != 2) {
        alive = poly_stage_robot<CB, NANS>(sc, q, ox, oy, alive);
        if (alive == 0ull) return 0ull;  // every lane of the pass is separated: no other normal can change an answer
    }
    if constexpr (STAGE == 1) {
        if (__popcll(alive) < C2D_MC_POLY_IN_PLACE) { final = false; return alive; }
        C2D_MC_STAT(8, __popcll(alive));
    }
    return poly_stage_obstacle<CB, NANS>(sc, q, ox, oy, alive);
}

template <bool NANS, int STAGE>
C2D_DEV unsigned long long poly_sample_stage_any(const PolyScene& sc, const PolyQueue& q, float dx, float dy, float c, float s, float fx, float fy,
                                                 unsigned long long lanes, bool& final)
{
    // wave-uniform dispatch on the obstacle's vertex count.  The tame path holds EXACTLY kb vertex slots in registers (round 5: 16
    // instances; rounded up to an even number before, a 5-gon paid for a sixth vertex in every projection loop and for a sixth,
    // zero-length edge normal in stage B — 13 % of the bench scene's instructions).  The all-bit-patterns path keeps the eight
    // even classes: its padding slot repeats vertex 0, which is exactly neutral, and it is there to be right, not fast.
#ifndef C2D_MC_POLY_EVEN_SLOTS
#define C2D_MC_POLY_EVEN_SLOTS 0   // 1: round 4's dispatch everywhere (A/B builds only)
#endif
#define C2D_POLY_STAGE(CB) return poly_sample_stage<CB, NANS, STAGE>(sc, q, dx, dy, c, s, fx, fy, lanes, final)
    if constexpr (NANS || C2D_MC_POLY_EVEN_SLOTS) {
        switch ((sc.kb + 1) >> 1) {
        case 1: C2D_POLY_STAGE(2);
        case 2: C2D_POLY_STAGE(4);
        case 3: C2D_POLY_STAGE(6);
        case 4: C2D_POLY_STAGE(8);
        case 5: C2D_POLY_STAGE(10);
        case 6: C2D_POLY_STAGE(12);
        case 7: C2D_POLY_STAGE(14);
        default: C2D_POLY_STAGE(16);
        }
    } else {
        switch (sc.kb) {
        case 1: C2D_POLY_STAGE(1);
        case 2: C2D_POLY_STAGE(2);
        case 3: C2D_POLY_STAGE(3);
        case 4: C2D_POLY_STAGE(4);
        case 5: C2D_POLY_STAGE(5);
        case 6: C2D_POLY_STAGE(6);
        case 7: C2D_POLY_STAGE(7);
        case 8: C2D_POLY_STAGE(8);
        case 9: C2D_POLY_STAGE(9);
        case 10: C2D_POLY_STAGE(10);
        case 11: C2D_POLY_STAGE(11);
        case 12: C2D_POLY_STAGE(12);
        case 13: C2D_POLY_STAGE(13);
        case 14: C2D_POLY_STAGE(14);
        case 15: C2D_POLY_STAGE(15);
        default: C2D_POLY_STAGE(16);
        }
    }
#undef C2D_POLY_STAGE
}

// the rest of a sample after its centre: dtheta, dw, dh (utils.cu:148-150) -> rotation and the two scale factors.  The third
// Box-Muller pair only feeds dh: its block is skipped when sigma_h == 0, because 1 + n * 0 = 1 for every finite n.
C2D_DEV void poly_sample_shape(const PolyScene& sc, uint32_t radius_word, uint32_t angle_word, uint64_t seed, uint64_t scene_id, uint64_t sample, float& c,
                               float& s, float& fx, float& fy)
{
    float n2, n3;
    box_muller(radius_word, angle_word, n2, n3);
    const float dt = n2 * sc.st;
    const float dw = n3 * sc.sw;
    float dh = 0.0f;
    if (sc.sh != 0.0f) {  // wave-uniform
        const U4 b = philox_draw_block(seed, scene_id, sample >> 2, 4u + ((uint32_t)(sample >> 1) & 1u));
        const bool odd = (sample & 1) != 0;
        float n4, unused;
        box_muller(odd ? b.z : b.x, odd ? b.w : b.y, n4, unused);
        dh = n4 * sc.sh;
    }
    fx = 1.0f + dw;
    fy = 1.0f + dh;
    sincos_(dt, s, c);
}

struct PolyPolicy {
    using Scene = PolyScene;
    using Queue = PolyQueue;
    // true: the centre alone proves that the sample cannot collide (build_poly_scene)
    static C2D_DEV bool centre_pretest(const Scene& sc, const Queue& q, float dx, float dy, unsigned long long& miss_m)
    {
        bool miss = false;
#pragma nounroll
        for (int i = 0; i < sc.ka; i++) {
            const float4 A = q.ctr[i];
            const float t = fma_(A.x, dx, A.y * dy);
            miss |= (t > A.w) | (t < A.z);
        }
        miss_m = __builtin_amdgcn_ballot_w64(miss);
        return miss;
    }
    // stage B for the last `take` (<= 64) queued survivors
    static C2D_DEV uint32_t evaluate_survivors(const Scene& sc, Queue& q, uint32_t& n, uint32_t take)
    {
        const uint32_t lane = threadIdx.x & 63;
        wave_lds_sync();
        const unsigned long long live_m = take >= 64 ? ~0ull : (1ull << take) - 1;
        const uint32_t src = n - take + (lane < take ? lane : 0);
        const float4 a = q.surv_a[src];
        const float2 b = q.surv_b[src];
        wave_lds_sync();
        n -= take;
        bool final;
        C2D_MC_STAT(9, take);
        return (uint32_t)__popcll(poly_sample_stage_any<false, 2>(sc, q, a.x, a.y, a.z, a.w, b.x, b.y, live_m, final));
    }
    static C2D_DEV uint32_t evaluate(const Scene& sc, Queue& q, uint32_t w2r, uint32_t w2a, float dx, float dy, uint64_t seed, uint64_t scene_id, uint64_t sample,
                                     unsigned long long live_m)
    {
        float c, s, fx, fy;
        poly_sample_shape(sc, w2r, w2a, seed, scene_id, sample, c, s, fx, fy);
        bool final;
        C2D_MC_STAT(5, __popcll(live_m));
        const unsigned long long alive = poly_sample_stage_any<false, 1>(sc, q, dx, dy, c, s, fx, fy, live_m, final);
        if (final) return (uint32_t)__popcll(alive);  // (nothing survived stage A, or stage B ran in place)
        C2D_MC_STAT(6, __popcll(alive));
        uint32_t n = q.n_surv;  // (wave-uniform: every lane reads the same word)
        if ((alive >> (threadIdx.x & 63)) & 1ull) {
            const uint32_t slot = n + __builtin_amdgcn_mbcnt_hi((uint32_t)(alive >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)alive, 0u));
            q.surv_a[slot] = make_float4(dx, dy, c, s);
            q.surv_b[slot] = make_float2(fx, fy);
        }
        n += (uint32_t)__popcll(alive);
        uint32_t hits = 0;
        if (n >= 64) hits = evaluate_survivors(sc, q, n, 64);
        wave_lds_sync();
        if ((threadIdx.x & 63) == 0) q.n_surv = n;
        wave_lds_sync();
        return hits;
    }
    // the survivors still queued when a sample range ends
    static C2D_DEV uint32_t finish(const Scene& sc, Queue& q)
    {
        uint32_t n = q.n_surv;
        if (n == 0) return 0u;
        uint32_t hits = 0;
        while (n) hits += evaluate_survivors(sc, q, n, n < 64 ? n : 64u);
        wave_lds_sync();
        if ((threadIdx.x & 63) == 0) q.n_surv = 0;
        wave_lds_sync();
        return hits;
    }
    // a scene that is not tame: one sample per lane, no pretest, the all-bit-patterns test
    static C2D_DEV uint32_t plain(const Scene& sc, uint64_t seed, uint64_t scene_id, uint64_t begin, uint32_t count, const Queue& q)
    {
        const uint32_t lane = threadIdx.x & 63;
        uint32_t hits = 0;
#pragma nounroll
        for (uint32_t i = 0; i < count; i += 64) {
            const uint32_t left = count - i;
            const unsigned long long live_m = left >= 64 ? ~0ull : (1ull << left) - 1;
            const uint64_t sm = begin + i + lane;
            const uint32_t j = (uint32_t)sm & 3u;
            const U4 b0 = philox_draw_block(seed, scene_id, sm >> 2, 0), b1 = philox_draw_block(seed, scene_id, sm >> 2, 1);
            const U4 b2 = philox_draw_block(seed, scene_id, sm >> 2, 2u + (j >> 1));
            float dx, dy, c, s, fx, fy;
            sample_centre(sc, u4_word(b0, (int)j), u4_word(b1, (int)j), dx, dy);
            poly_sample_shape(sc, (j & 1u) ? b2.z : b2.x, (j & 1u) ? b2.w : b2.y, seed, scene_id, sm, c, s, fx, fy);
            bool final;
            hits += (uint32_t)__popcll(poly_sample_stage_any<true, 0>(sc, q, dx, dy, c, s, fx, fy, live_m, final));
        }
        return hits;
    }
};

// ---- one scene, sample-parallel --------------------------------------------------------------------------------------------
struct PolyPairArgs {
    c2d_polygon robot, obstacle;
    float px, py, theta;
    StdDev sd;
    uint64_t seed, scene_id, sample_begin, n_samples;
    uint32_t chunk;  // samples per wave, a multiple of 256
};

#ifndef C2D_MC_POLY_WAVES
#define C2D_MC_POLY_WAVES 5
#endif
#ifdef C2D_MC_CLOCK
C2D_MC_CLOCK_WORDS(c2d_mc_clock_poly_pair);  // clock build only (make lib-mcclock): read by c2d_debug_mc_poly_clock
C2D_MC_CLOCK_WORDS(c2d_mc_clock_poly);
#endif
__global__ __launch_bounds__(kMcBlock, C2D_MC_POLY_WAVES) void mc_poly_pair_kernel(PolyPairArgs A, unsigned long long* __restrict__ d_hits)
{
    __shared__ PolyQueue s_queue[kWavesPerBlock];
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t l = threadIdx.x & 15;
    const PolyScene sc = build_poly_scene(lane_pick(A.robot.x, l), lane_pick(A.robot.y, l), (int)A.robot.k, A.px, A.py, A.theta, lane_pick(A.obstacle.x, l),
                                          lane_pick(A.obstacle.y, l), (int)A.obstacle.k, A.sd, s_queue[wave]);
    const uint64_t n_chunks = (A.n_samples + A.chunk - 1) / A.chunk;
    unsigned long long total = 0;
    C2D_MC_CLOCK_START();
    for (uint64_t ch = (uint64_t)blockIdx.x * kWavesPerBlock + wave; ch < n_chunks; ch += (uint64_t)gridDim.x * kWavesPerBlock) {
        const uint64_t off = ch * A.chunk;
        const uint64_t left = A.n_samples - off;
        const uint32_t count = left < A.chunk ? (uint32_t)left : A.chunk;
        total += wave_count_hits<PolyPolicy>(sc, A.seed, A.scene_id, A.sample_begin + off, count, s_queue[wave]);
    }
    C2D_MC_CLOCK_STOP(c2d_mc_clock_poly_pair);
    if ((threadIdx.x & 63) == 0 && total) atomicAdd(d_hits, total);
}

// ---- many polygon scenes, adaptive: the schedule of c2d_mc_core.hpp around the polygon scene builder ----------------------------
struct PolyScenesArgs {
    // shape-independent part (run_adaptive fills it)
    const StdDev* std_devs;
    const PositionWithVarAndPoseIdx* scenes;
    AdaptiveState* state;
    uint32_t* lists[2];
    uint32_t num_std_devs;
    uint64_t seed, scene_id_base;
    ScheduleArgs sched;
    uint32_t* hits;
    uint32_t burst_steps;
    uint32_t n_bins;
    float bins[16], acc[16];
    uint32_t* n_used;
    PoseCPVarAndPoseIdx* rows;
    // polygons
    const c2d_poly_pose* poly_poses;
    uint32_t num_poly_poses;
    c2d_polygon robot;
    uint32_t* async_err;
};

struct PolyBuilder {
    using Args = PolyScenesArgs;
    using Policy = PolyPolicy;
#ifdef C2D_MC_CLOCK
    static C2D_DEV unsigned long long* clock_words() { return c2d_mc_clock_poly; }
#endif
    static C2D_DEV PolyScene scene(const Args& A, const PositionWithVarAndPoseIdx& row, PolyQueue& q)
    {
        // float -> int index conversion as in ccp.cu:121-122; clamped so that a malformed row cannot read outside the tables
        uint32_t pi = (uint32_t)(int)row.pose_idx, vi = (uint32_t)(int)row.var_idx;
        pi = pi < A.num_poly_poses ? pi : A.num_poly_poses - 1;
        vi = vi < A.num_std_devs ? vi : A.num_std_devs - 1;
        const c2d_poly_pose* pp = A.poly_poses + pi;
        const uint32_t l = threadIdx.x & 15;
        const float theta = pp->theta, ovx = pp->obstacle.x[l], ovy = pp->obstacle.y[l];
        uint32_t kb = pp->obstacle.k, ka = A.robot.k;
        if (kb < 1 || kb > (uint32_t)KM) {  // reported by the next c2d_stream_synchronize / c2d_ctx_check_async, clamped here
            if ((threadIdx.x & 63) == 0) __hip_atomic_fetch_or(A.async_err, C2D_ASYNC_ERR_POLY_K, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            kb = kb < 1 ? 1u : (uint32_t)KM;
        }
        ka = ka < 1 ? 1u : (ka > (uint32_t)KM ? (uint32_t)KM : ka);  // (validated on the host)
        return build_poly_scene(lane_pick(A.robot.x, l), lane_pick(A.robot.y, l), (int)ka, row.x, row.y, theta, ovx, ovy, (int)kb, A.std_devs[vi], q);
    }
};

template <bool BURST>
__global__ __launch_bounds__(kMcBlock, C2D_MC_POLY_WAVES) void mc_poly_scenes_advance_kernel(PolyScenesArgs A) { mc_scenes_advance_body<BURST, PolyBuilder>(A); }

}  // namespace c2d

using namespace c2d;

namespace {

const char* polygon_fault(const c2d_polygon* p)
{
    if (!p) return "NULL polygon";
    if (p->k < 1 || p->k > (uint32_t)C2D_POLY_KMAX) return "polygon vertex count outside 1..C2D_POLY_KMAX";
    return nullptr;
}

// unused slots are never interpreted, but they travel in the kernel arguments: give them a defined value
c2d_polygon tidy(const c2d_polygon& p)
{
    c2d_polygon o = p;
    for (uint32_t k = p.k; k < (uint32_t)C2D_POLY_KMAX; k++) o.x[k] = o.y[k] = 0.0f;
    return o;
}

}  // namespace

extern "C" {

int c2d_mc_poly_pair(c2d_ctx* ctx, const c2d_polygon* robot, const Position* pos, float robot_theta, const c2d_polygon* obstacle, const StdDev* std_dev,
                     uint64_t seed, uint64_t scene_id, uint64_t sample_begin, uint64_t n_samples, unsigned long long* d_hits, c2d_stream stream)
{
    if (!ctx) return C2D_ERR_INVALID_ARG;
    if (!robot || !pos || !obstacle || !std_dev || !d_hits) return fail_arg(ctx, "c2d_mc_poly_pair: NULL argument");
    if (const char* f = polygon_fault(robot)) return fail_arg(ctx, (std::string("c2d_mc_poly_pair: robot: ") + f).c_str());
    if (const char* f = polygon_fault(obstacle)) return fail_arg(ctx, (std::string("c2d_mc_poly_pair: obstacle: ") + f).c_str());
    if (n_samples == 0) return C2D_OK;
    if (sample_begin + n_samples < sample_begin || sample_begin + n_samples > (1ull << 62))
        return fail_arg(ctx, "c2d_mc_poly_pair: sample range overflows the 2^62-sample stream");
    PolyPairArgs A;
    A.robot = tidy(*robot); A.obstacle = tidy(*obstacle);
    A.px = pos->x; A.py = pos->y; A.theta = robot_theta; A.sd = *std_dev;
    A.seed = seed; A.scene_id = scene_id; A.sample_begin = sample_begin; A.n_samples = n_samples;
    // chunk: enough samples per wave to amortise the scene set-up, enough waves to fill the chip (as c2d_mc_pair)
    const uint64_t target_waves = (uint64_t)ctx->prop.multiProcessorCount * 32;
    uint64_t chunk = (n_samples + target_waves - 1) / target_waves;
    chunk = ((chunk + 255) / 256) * 256;
    if (chunk < 256) chunk = 256;
    if (chunk > 8192) chunk = 8192;
    A.chunk = (uint32_t)chunk;
    const uint64_t n_chunks = (n_samples + chunk - 1) / chunk;
    uint64_t blocks = (n_chunks + kWavesPerBlock - 1) / kWavesPerBlock;
    const uint64_t max_blocks = (uint64_t)ctx->prop.multiProcessorCount * 256 / kWavesPerBlock;
    if (blocks > max_blocks) blocks = max_blocks;
    DeviceGuard g(ctx->device);
    hipLaunchKernelGGL(mc_poly_pair_kernel, dim3((unsigned)blocks), dim3(kMcBlock), 0, (hipStream_t)stream, A, d_hits);
    C2D_LAUNCH_CHECK(ctx);
    return C2D_OK;
}

int c2d_mc_poly_scenes(c2d_ctx* ctx, const c2d_mc_poly_scenes_args* a, c2d_stream stream)
{
    if (!ctx) return C2D_ERR_INVALID_ARG;
    if (!a) return fail_arg(ctx, "c2d_mc_poly_scenes: NULL args");
    const c2d_mc_scenes_args* b = &a->base;
    if (b->n_scenes != 0) {
        const char* why = nullptr;
        if (!a->d_poly_poses) why = "c2d_mc_poly_scenes: NULL argument";
        else if (a->num_poly_poses == 0) why = "c2d_mc_poly_scenes: empty pose / std_dev table";
        else if (polygon_fault(a->robot)) why = "c2d_mc_poly_scenes: robot: NULL or vertex count outside 1..C2D_POLY_KMAX";
        if (why) {
            if (b->total_samples) *b->total_samples = 0;
            if (b->iterations) *b->iterations = 0;
            return fail_arg(ctx, why);
        }
    }
    PolyScenesArgs A;
    A.poly_poses = a->d_poly_poses; A.num_poly_poses = a->num_poly_poses;
    if (a->robot) A.robot = tidy(*a->robot);
    A.async_err = ctx->d_async_err;
    return run_adaptive(ctx, b, stream, A, [](bool burst, unsigned blocks, const PolyScenesArgs& args, hipStream_t s) {
        if (burst) hipLaunchKernelGGL(mc_poly_scenes_advance_kernel<true>, dim3(blocks), dim3(kMcBlock), 0, s, args);
        else hipLaunchKernelGGL(mc_poly_scenes_advance_kernel<false>, dim3(blocks), dim3(kMcBlock), 0, s, args);
    }, "c2d_mc_poly_scenes");
}

#ifdef C2D_MC_STATS
// census build only: copies the twelve counters to the host (after synchronising the device) and optionally clears them
int c2d_debug_mc_poly_stats(c2d_ctx* ctx, unsigned long long out[12], int reset)
{
    if (!ctx || !out) return C2D_ERR_INVALID_ARG;
    DeviceGuard g(ctx->device);
    C2D_HIP(ctx, hipDeviceSynchronize());
    C2D_HIP(ctx, hipMemcpyFromSymbol(out, HIP_SYMBOL(c2d_mc_poly_stats_words), 12 * sizeof(unsigned long long)));
    if (reset) {
        const unsigned long long zero[12] = {};
        C2D_HIP(ctx, hipMemcpyToSymbol(HIP_SYMBOL(c2d_mc_poly_stats_words), zero, sizeof zero));
    }
    return C2D_OK;
}
#endif

#ifdef C2D_MC_CLOCK
// clock build only: the stamps of mc_poly_pair_kernel (which = 0) or of mc_poly_scenes_advance_kernel (1): shader cycles, 100 MHz ticks, waves
int c2d_debug_mc_poly_clock(c2d_ctx* ctx, int which, unsigned long long out[4], int reset)
{
    if (!ctx || !out || which < 0 || which > 1) return C2D_ERR_INVALID_ARG;
    DeviceGuard g(ctx->device);
    C2D_HIP(ctx, hipDeviceSynchronize());
    const unsigned long long zero[4] = {};
    if (which == 0) {
        C2D_HIP(ctx, hipMemcpyFromSymbol(out, HIP_SYMBOL(c2d_mc_clock_poly_pair), 4 * sizeof(unsigned long long)));
        if (reset) C2D_HIP(ctx, hipMemcpyToSymbol(HIP_SYMBOL(c2d_mc_clock_poly_pair), zero, sizeof zero));
    } else {
        C2D_HIP(ctx, hipMemcpyFromSymbol(out, HIP_SYMBOL(c2d_mc_clock_poly), 4 * sizeof(unsigned long long)));
        if (reset) C2D_HIP(ctx, hipMemcpyToSymbol(HIP_SYMBOL(c2d_mc_clock_poly), zero, sizeof zero));
    }
    return C2D_OK;
}
#endif

}  // extern "C"
