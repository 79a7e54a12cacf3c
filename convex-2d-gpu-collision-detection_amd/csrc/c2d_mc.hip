// c2d_mc.hip — Monte-Carlo collision probability on gfx950 (MI355X).
//
// Replaces monte_carlo_sample_collision_dataset_uniform
// (reference compute_collision_probability.cu:90-150, generate_dataset.cu:175-253),
// sample_rectangle (utils.cu:144-157), setup_kernel (utils.cu:111-117), the
// thrust count/sort compaction (ccp.cu:307-319) and write_collision_probability
// (utils.cu:210-215).
//
// Mapping.  The reference gives one thread one scene and runs its samples
// serially.  Here one *wave* (64 lanes) owns one (scene, sample
// chunk): the scene constants are wave-uniform (scalar registers), lane l
// evaluates the four samples of group chunk_begin / 4 + l, + 64, ..., and hits are
// counted with ballot + popcount in a scalar register.  The random stream is counter based
// (Philox4x32-10 keyed by seed, scene, sample group: c2d_math.hpp), so there is no RNG state in
// memory, no set-up kernel, and any partition of the samples over waves, blocks
// or GPUs gives the same hit count.  The kernel is VALU bound (~0 B of HBM
// traffic per sample); see DESIGN.md for the per-sample op budget.
//
// Work avoidance that cannot change a result: a sample whose obstacle centre alone
// proves a miss (bounding-disk argument, make_scene) is dropped after its first
// Box-Muller pair, or — from the raw radius word, four samples per Philox block — before any transcendental; the
// undecided samples of a wave are compacted through an LDS queue so that the full
// evaluation always runs on 64 busy lanes; and the full evaluation itself decides a sample from the closed-form gap of
// the two rectangles (model_gap) whenever that gap exceeds a proven rounding margin, with the reference's vertex arithmetic
// (convex_collide's projections and comparisons, float for float) behind it for the thin ones.
#include "c2d_internal.hpp"
#include "c2d_math.hpp"

namespace c2d {

// ---- census build (-DC2D_MC_STATS, `make lib-mcstats`; never the product): where the samples of a workload go.  Wave-uniform
// counts are added to device words — [0] samples handed to a wave, [1] of those on the FAR path, [2] on the NEAR path,
// [3] candidates left by the radius-word test (far path), [4] first Box-Muller pairs evaluated (centre of the obstacle),
// [5] samples that reach the full evaluation, [6] second axes of a parallel pair evaluated by the vertex arithmetic, [7] hits,
// [8] evaluation passes (64 lanes each) that the closed-form test left undecided for some lane, so that the wave ran the vertex
// arithmetic — and read by c2d_debug_mc_stats (tests/tools/mc_stats.py).
#ifdef C2D_MC_STATS
__device__ unsigned long long c2d_mc_stats_words[12];
#define C2D_MC_STAT(i, v)                                                                                     \
    do {                                                                                                      \
        const unsigned long long v__ = (unsigned long long)(v);                                               \
        if ((threadIdx.x & 63) == 0 && v__) atomicAdd(&c2d_mc_stats_words[i], v__);                           \
    } while (0)
#else
#define C2D_MC_STAT(i, v) do { } while (0)
#endif

}  // namespace c2d

#include "c2d_mc_core.hpp"  // the shape-independent part: draw layout, NEAR / FAR sample loops, queues

#ifdef C2D_MC_CLOCK
namespace c2d {
C2D_MC_CLOCK_WORDS(c2d_mc_clock_pair);     // mc_pair_kernel
C2D_MC_CLOCK_WORDS(c2d_mc_clock_scenes);   // mc_scenes_advance_kernel
}  // namespace c2d
#endif

namespace c2d {

// Wave-uniform description of one scene (reference ccp.cu:119-133).
struct Scene {
    float robot[8];       // robot rectangle in the obstacle frame
    float hw, hh;         // obstacle half extents (create_rect: +-w/2, +-h/2)
    float sx, sy, st, sw, sh;  // standard deviations (StdDev)
    // Certain-separation pretest on the obstacle centre alone (see centre_pretest)
    float pax[2], pay[2], plo[2], phi[2];
    // Radius-only form of the same pretest: raw word x >= x0 (and use_x0) proves a miss
    uint32_t x0;
    bool use_x0;
    // Certificates for the second axis of each parallel pair (sample_collides_mask): a rectangle's edge axes 2, 3 are the
    // negatives of axes 0, 1 up to rounding, so when the two intervals overlap by enough on axis i, axis i + 2 cannot separate
    float skip_lo[2], skip_hi[2];  // robot axes: obstacle interval must reach [.., skip_lo] and [skip_hi, ..] on axis 0 / 1
    float skip_c2, skip_c3;        // obstacle axes: needed overlap = c2 |a + b|_1 + c3 |a|_1 (a, b: the two axes of the pair)
    // Closed-form test of a full evaluation (model_gap): the robot as centre, rotation and half extents, and the margin
    // within which the closed form does not decide (+inf: never decides, every evaluation takes the vertex arithmetic)
    float mrx, mry, mcr, msr, mhw, mhh, margin;
    // use_x0 == false && x0 == 0 marks a scene that is not "tame": some parameter is NaN, infinite, >= 1e15 or (a length) below 1e-15 in
    // magnitude, so a sampled vertex or a projection may be non-finite.  Such a scene takes wave_count_hits_plain: no
    // pretest (their proofs assume numbers) and the axis test that restores minmax_element's behaviour on a NaN first
    // projection (first_projections_ordered, c2d_math.hpp).  Below 1e15 no product on the way can overflow, so a tame
    // scene never produces a NaN and pays for none of this (not even a register: the mark lives in x0).
    C2D_DEV bool tame() const { return use_x0 || x0 != 0u; }
};

C2D_DEV Scene make_scene_values(float robot_w, float robot_h, float px, float py, const Pose& pose, const StdDev& sd)
{
    Scene sc;
    float s, c;
    sincos_(pose.theta, s, c);
    rect_from_half_extents(robot_w / 2, robot_h / 2, c, s, px, py, sc.robot);  // ccp.cu:132-133
    sc.hw = pose.width / 2;                                                    // ccp.cu:128
    sc.hh = pose.height / 2;
    sc.sx = sd.x; sc.sy = sd.y; sc.st = sd.theta; sc.sw = sd.width; sc.sh = sd.height;
    sc.mrx = px; sc.mry = py; sc.mcr = c; sc.msr = s;
    sc.mhw = __builtin_fabsf(robot_w / 2);
    sc.mhh = __builtin_fabsf(robot_h / 2);
    sc.margin = __builtin_inff();

    // ---- certain-separation pretest ------------------------------------------------
    // Every vertex of a sampled obstacle is  centre + w  with centre = (dx, dy) and
    // |w| <= rho = sqrt(hx^2 + hy^2) whatever the rotation, where |hx| <= |hw| + 3.385|sw|
    // and |hy| <= |hh| + 3.385|sh| (|draw| <= kNormalMax).  On robot axis a the SAT compares
    // the obstacle's projections a.o_k with the robot's own interval [rmin, rmax] (the very
    // floats computed below).  With T = a.centre,
    //     a.o_k  >=  T - |a|_2 rho - err,     a.o_k  <=  T + |a|_2 rho + err,
    // err = all fp32 roundings on the way (vertex construction ~6u |a|_2 rho, final adds and
    // the dot products ~8u |a|_1 (rho + D), the evaluation of T ~3u |a|_1 D; u = 2^-24,
    // D >= |dx| + |dy|).  The margin M below grants 2^-10 |a|_2 rho + 2^-12 |a|_1 (rho + D),
    // over 500 times that, and the thresholds are pushed outwards by another 2^-20 relative.
    // Hence T > phi or T < plo on either axis proves that convex_collide would find that
    // axis separating: the sample is a certain miss and nothing else of it needs evaluating.
    const float hxm = __builtin_fabsf(sc.hw) + 0.5f * kNormalMax * __builtin_fabsf(sc.sw);
    const float hym = __builtin_fabsf(sc.hh) + 0.5f * kNormalMax * __builtin_fabsf(sc.sh);
    const float rho = __builtin_sqrtf(hxm * hxm + hym * hym);
    const float D = kNormalMax * (__builtin_fabsf(sc.sx) + __builtin_fabsf(sc.sy));
    // ---- certificates for the parallel axes (derivation at sample_collides_mask).  C bounds every coordinate of the
    // robot and of any sampled obstacle: |o| <= |centre| + |c hx| + |s hy| <= 6.77 sigma + hxm + hym (|draw| <= kNormalMax)
    float cmax = kNormalMax * __builtin_fmaxf(__builtin_fabsf(sc.sx), __builtin_fabsf(sc.sy)) + (hxm + hym);
#pragma unroll
    for (int k = 0; k < 8; k++) cmax = __builtin_fmaxf(cmax, __builtin_fabsf(sc.robot[k]));
    cmax *= 1.0f + 0x1p-10f;
    sc.skip_c2 = (2.0f + 0x1p-8f) * cmax;
    sc.skip_c3 = (8.0f + 0x1p-6f) * 0x1p-24f * cmax;
#pragma unroll
    for (int i = 0; i < 2; i++) {
        const float ax = sc.robot[2 * i + 2] - sc.robot[2 * i];
        const float ay = sc.robot[2 * i + 3] - sc.robot[2 * i + 1];
        // the robot's own interval on its axis i: the very floats the SAT computes (dot2)
        const float p0 = dot2(ax, sc.robot[0], ay, sc.robot[1]), p1 = dot2(ax, sc.robot[2], ay, sc.robot[3]);
        const float p2 = dot2(ax, sc.robot[4], ay, sc.robot[5]), p3 = dot2(ax, sc.robot[6], ay, sc.robot[7]);
        const float rmin = min4(p0, p1, p2, p3), rmax = max4(p0, p1, p2, p3);
        {
            const float bx = sc.robot[(2 * i + 6) & 7] - sc.robot[2 * i + 4], by = sc.robot[(2 * i + 7) & 7] - sc.robot[2 * i + 5];  // axis i + 2
            const float need = (__builtin_fabsf(ax + bx) + __builtin_fabsf(ay + by)) * sc.skip_c2 +
                               (__builtin_fabsf(ax) + __builtin_fabsf(ay)) * sc.skip_c3 + 1e-36f;
            const float lo = rmin + need, hi = rmax - need;
            sc.skip_lo[i] = lo + 0x1p-20f * __builtin_fabsf(lo);
            sc.skip_hi[i] = hi - 0x1p-20f * __builtin_fabsf(hi);
            if (ax == -bx && ay == -by) {  // exactly opposite: every projection is the exact negative, axis i + 2 repeats axis i's two comparisons
                sc.skip_lo[i] = -__builtin_inff();
                sc.skip_hi[i] = __builtin_inff();
            }
        }
        const float n2 = __builtin_sqrtf(ax * ax + ay * ay), n1 = __builtin_fabsf(ax) + __builtin_fabsf(ay);
        const float M = (n2 * rho) * (1.0f + 0x1p-10f) + 0x1p-12f * (n1 * (rho + D));
        const float hi = rmax + M, lo = rmin - M;
        sc.pax[i] = ax;
        sc.pay[i] = ay;
        sc.phi[i] = hi + 0x1p-20f * __builtin_fabsf(hi);
        sc.plo[i] = lo - 0x1p-20f * __builtin_fabsf(lo);
    }

    // ---- the same pretest from the Box-Muller radius alone -----------------------------
    // T_i = pax_i dx + pay_i dy with dx = sin * rad * sx, dy = cos * rad * sy, so
    // |T_i| <= rad * G_i (1 + 2^-10), G_i = sqrt((pax_i sx)^2 + (pay_i sy)^2)   (sin^2 + cos^2 <=
    // 1 + 2^-21 for the canonical pair; the slack covers every rounding).  If the robot's widened
    // slab i does not contain the obstacle's mean centre (the origin) — plo_i > 0 or phi_i < 0 —
    // then rad < L_i / (G_i (1 + 2^-10)) with L_i = plo_i resp. -phi_i already proves the miss.
    // rad = sqrt(-2 log u) falls with u = (x + 1/2) 2^-32, so "rad < R0" is "x >= x0" for the
    // raw word x: one integer compare decides the sample before any transcendental is evaluated.
    // x0 is rounded up generously (radius_word_threshold, c2d_mc_core.hpp).
    float R0 = 0.0f;
#pragma unroll
    for (int i = 0; i < 2; i++) {
        const float gx = sc.pax[i] * sc.sx, gy = sc.pay[i] * sc.sy;
        const float G = __builtin_sqrtf(gx * gx + gy * gy) * (1.0f + 0x1p-10f);
        const float L = sc.plo[i] > 0.0f ? sc.plo[i] : (sc.phi[i] < 0.0f ? -sc.phi[i] : 0.0f);
        if (L > 0.0f && G > 0.0f) R0 = __builtin_fmaxf(R0, L / G);
        // G == 0 (sigma_x = sigma_y = 0) leaves T_i = 0: centre_pretest itself decides that case
    }
    // ---- margin of the closed-form test (derivation at model_gap).  C bounds every coordinate, every centre offset plus
    // extent sum of the robot and of any sampled obstacle; h is the smallest half extent any of the eight edges can have
    {
        const float C = __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(px), __builtin_fabsf(py)) + (sc.mhw + sc.mhh),
                                        kNormalMax * __builtin_fmaxf(__builtin_fabsf(sc.sx), __builtin_fabsf(sc.sy)) + (hxm + hym)) * (1.0f + 0x1p-10f);
        const float hx_lo = __builtin_fabsf(sc.hw) - 0.5f * kNormalMax * __builtin_fabsf(sc.sw) * (1.0f + 0x1p-10f);
        const float hy_lo = __builtin_fabsf(sc.hh) - 0.5f * kNormalMax * __builtin_fabsf(sc.sh) * (1.0f + 0x1p-10f);
        const float h = __builtin_fminf(__builtin_fminf(sc.mhw, sc.mhh), __builtin_fminf(hx_lo, hy_lo));
        if (h >= 1e-12f && C < 1e15f)  // (false for a NaN)
            sc.margin = (64.0f + 20.0f * (C / h)) * (0x1p-24f * C) * (1.0f + 0x1p-10f);
    }
    sc.use_x0 = false;
    sc.x0 = 0xffffffffu;
    {
        const float big = 1e15f;  // (a NaN compares false: not tame)
        auto ok = [big](float v) { return __builtin_fabsf(v) < big; };
        // ... and no length so small that a product of two lengths leaves the normal range: the margins of the shortcuts are
        // RELATIVE rounding bounds, which a denormal product does not obey (scenes scaled by 1e-22 differed from the oracle)
        auto len = [big](float v) { const float a = __builtin_fabsf(v); return a < big && (a == 0.0f || a >= 1e-15f); };
        const bool tame = len(robot_w) && len(robot_h) && len(px) && len(py) && len(pose.width) && len(pose.height) && ok(pose.theta) &&
                          len(sd.x) && len(sd.y) && ok(sd.theta) && len(sd.width) && len(sd.height);
        if (!tame) {
            sc.x0 = 0u;  // the mark (see Scene::tame)
            return sc;
        }
    }
    radius_word_threshold(R0, sc.x0, sc.use_x0);  // (c2d_mc_core.hpp; the error bound of its hardware exponential is stated there)
    return sc;
}

// The scene is wave-uniform, but gfx950's scalar unit has no floating point: computed by the vector unit, every field
// sits in a VGPR of its own (the same value in all 64 lanes) for as long as the sample loops run — three dozen of the
// kernels' 80 registers, and the reason for their dozen spilled dwords.  Moving the fields into scalar registers
// (v_readfirstlane; -DC2D_MC_SCENE_IN_SGPRS) was measured and is NOT the default: it frees the VGPRs (mc_pair_kernel 92 -> 72,
// no scratch left in the advance kernels) but the kernels already use every SGPR for lane masks and arguments, so 30 to 60
// scalars spill to VGPR lanes instead and come back through v_readlane inside the sample loops: config-3 scene 0.547 ->
// 0.585 ms, config-4 shard 382 -> 400 ms; 7 or 8 waves per SIMD on top of it are slower still (profiles/r03_mc_isa.md).
C2D_DEV float to_sgpr(float v) { return __uint_as_float((uint32_t)__builtin_amdgcn_readfirstlane((int)__float_as_uint(v))); }

C2D_DEV Scene make_scene(float robot_w, float robot_h, float px, float py, const Pose& pose, const StdDev& sd)
{
    Scene sc = make_scene_values(robot_w, robot_h, px, py, pose, sd);
#ifdef C2D_MC_SCENE_IN_SGPRS
#pragma unroll
    for (int k = 0; k < 8; k++) sc.robot[k] = to_sgpr(sc.robot[k]);
    sc.hw = to_sgpr(sc.hw); sc.hh = to_sgpr(sc.hh);
    sc.sx = to_sgpr(sc.sx); sc.sy = to_sgpr(sc.sy); sc.st = to_sgpr(sc.st); sc.sw = to_sgpr(sc.sw); sc.sh = to_sgpr(sc.sh);
#pragma unroll
    for (int i = 0; i < 2; i++) {
        sc.pax[i] = to_sgpr(sc.pax[i]); sc.pay[i] = to_sgpr(sc.pay[i]); sc.plo[i] = to_sgpr(sc.plo[i]); sc.phi[i] = to_sgpr(sc.phi[i]);
        sc.skip_lo[i] = to_sgpr(sc.skip_lo[i]); sc.skip_hi[i] = to_sgpr(sc.skip_hi[i]);
    }
    sc.skip_c2 = to_sgpr(sc.skip_c2); sc.skip_c3 = to_sgpr(sc.skip_c3);
    sc.mrx = to_sgpr(sc.mrx); sc.mry = to_sgpr(sc.mry); sc.mcr = to_sgpr(sc.mcr); sc.msr = to_sgpr(sc.msr);
    sc.mhw = to_sgpr(sc.mhw); sc.mhh = to_sgpr(sc.mhh); sc.margin = to_sgpr(sc.margin);
    sc.x0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)sc.x0);
#endif
    return sc;
}

// true: this sample certainly does not collide (proof in make_scene); false: unknown.
// Non-finite thresholds compare false, i.e. "unknown".
C2D_DEV bool centre_pretest(const Scene& sc, float dx, float dy, unsigned long long& miss_m)
{
    const float t0 = fma_(sc.pax[0], dx, sc.pay[0] * dy);
    const float t1 = fma_(sc.pax[1], dx, sc.pay[1] * dy);
    const bool a = t0 > sc.phi[0], b = t0 < sc.plo[0], c = t1 > sc.phi[1], d = t1 < sc.plo[1];
    // the same vote as a lane mask, from the bare comparisons (see sample_collides_mask)
    miss_m = __builtin_amdgcn_ballot_w64(a) | __builtin_amdgcn_ballot_w64(b) | __builtin_amdgcn_ballot_w64(c) | __builtin_amdgcn_ballot_w64(d);
    return a | b | c | d;
}
// The rest of a sampled obstacle (reference utils.cu:144-157) after its centre (sample_centre, c2d_mc_core.hpp): dtheta, dw, dh and the
// rotation; only needed when the centre pretest cannot rule the sample out.  The third Box-Muller pair only feeds dh; its block is
// skipped when sigma_h == 0 because dh = n*0 cannot change any vertex (the product is +-0 and is only ever added).

C2D_DEV void sample_shape(const Scene& sc, uint32_t radius_word, uint32_t angle_word, uint64_t seed, uint64_t scene_id, uint64_t sample,
                          float& hx, float& hy, float& s, float& c)
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
    // r_out = r_in + create_rect(dw, dh): half extents add (utils.cu:152-155)
    hx = sc.hw + dw / 2;
    hy = sc.hh + dh / 2;
    sincos_(dt, s, c);
}

// convex_collide(robot, obstacle) (utils.cu:159-184).  All eight axes are part
// of the result; the remaining axes are skipped only when every lane of the
// wave is already separated, which cannot change any lane's answer.  The
// wave-wide check (one ballot + scalar branch) sits after axes 2, 4 and 6: the
// undecided samples of far and low-p scenes are mostly misses and leave early, a scene
// with p ~ 0.5 pays three scalar branches (config-3 scene 0.620 -> 0.605 ms without them, but
// the config-4 shard 401 -> 410 ms; making the later checks depend on the first one's count
// costs what it saves on both).
#ifndef C2D_MC_EARLY_MASK
#define C2D_MC_EARLY_MASK 0x2A  // bit i set: check after axis i+1 (axes 0-3 robot, 4-7 obstacle)
#endif

// lanes whose sample this axis separates (reference utils.cu:172-180: unfused dots, strict <), as a lane mask
C2D_DEV unsigned long long axis_separates_mask(float ax, float ay, const float (&r1)[8], const float (&r2)[8])
{
    float p10 = dot2(ax, r1[0], ay, r1[1]), p11 = dot2(ax, r1[2], ay, r1[3]);
    float p12 = dot2(ax, r1[4], ay, r1[5]), p13 = dot2(ax, r1[6], ay, r1[7]);
    float p20 = dot2(ax, r2[0], ay, r2[1]), p21 = dot2(ax, r2[2], ay, r2[3]);
    float p22 = dot2(ax, r2[4], ay, r2[5]), p23 = dot2(ax, r2[6], ay, r2[7]);
    float min1 = min4(p10, p11, p12, p13), max1 = max4(p10, p11, p12, p13);
    float min2 = min4(p20, p21, p22, p23), max2 = max4(p20, p21, p22, p23);
    return __builtin_amdgcn_ballot_w64(max1 < min2) | __builtin_amdgcn_ballot_w64(max2 < min1);
}

#ifndef C2D_MC_NO_AXIS_SKIP
// The two halves of the test below (see its comment for the certificates).  `among`: the lanes whose answer is wanted;
// returns those of them that no axis of the half separates.
C2D_DEV unsigned long long robot_axes_survivors(const Scene& sc, const float (&o)[8], unsigned long long among)
{
    unsigned long long sep = 0;
    unsigned long long thin[2];
#pragma unroll
    for (int i = 0; i < 2; i++) {  // robot axes 0, 1 (the robot's own interval is wave-uniform)
        const float ax = sc.robot[2 * i + 2] - sc.robot[2 * i], ay = sc.robot[2 * i + 3] - sc.robot[2 * i + 1];
        const float r0 = dot2(ax, sc.robot[0], ay, sc.robot[1]), r1 = dot2(ax, sc.robot[2], ay, sc.robot[3]);
        const float r2 = dot2(ax, sc.robot[4], ay, sc.robot[5]), r3 = dot2(ax, sc.robot[6], ay, sc.robot[7]);
        const float q0 = dot2(ax, o[0], ay, o[1]), q1 = dot2(ax, o[2], ay, o[3]);
        const float q2 = dot2(ax, o[4], ay, o[5]), q3 = dot2(ax, o[6], ay, o[7]);
        const float rmin = min4(r0, r1, r2, r3), rmax = max4(r0, r1, r2, r3);
        const float omin = min4(q0, q1, q2, q3), omax = max4(q0, q1, q2, q3);
        sep |= __builtin_amdgcn_ballot_w64(rmax < omin) | __builtin_amdgcn_ballot_w64(omax < rmin);
        thin[i] = __builtin_amdgcn_ballot_w64(omax < sc.skip_lo[i]) | __builtin_amdgcn_ballot_w64(omin > sc.skip_hi[i]);
    }
    if ((among & ~sep) == 0ull) return 0ull;
#pragma unroll
    for (int i = 0; i < 2; i++) {
        if ((thin[i] & among & ~sep) != 0ull) {  // axis i + 2 in full
            C2D_MC_STAT(6, 1);
            const float ax = sc.robot[(2 * i + 6) & 7] - sc.robot[2 * i + 4], ay = sc.robot[(2 * i + 7) & 7] - sc.robot[2 * i + 5];
            sep |= axis_separates_mask(ax, ay, sc.robot, o);
        }
    }
    return among & ~sep;
}

C2D_DEV unsigned long long obstacle_axes_survivors(const Scene& sc, const float (&o)[8], unsigned long long among)
{
    unsigned long long sep = 0;
    unsigned long long thin2[2];
#pragma unroll
    for (int j = 0; j < 2; j++) {  // obstacle axes 4, 5, with the per-sample certificate for 6, 7
        const float ax = o[2 * j + 2] - o[2 * j], ay = o[2 * j + 3] - o[2 * j + 1];
        const float bx = o[(2 * j + 6) & 7] - o[2 * j + 4], by = o[(2 * j + 7) & 7] - o[2 * j + 5];  // obstacle edge j + 2
        const float r0 = dot2(ax, sc.robot[0], ay, sc.robot[1]), r1 = dot2(ax, sc.robot[2], ay, sc.robot[3]);
        const float r2 = dot2(ax, sc.robot[4], ay, sc.robot[5]), r3 = dot2(ax, sc.robot[6], ay, sc.robot[7]);
        const float q0 = dot2(ax, o[0], ay, o[1]), q1 = dot2(ax, o[2], ay, o[3]);
        const float q2 = dot2(ax, o[4], ay, o[5]), q3 = dot2(ax, o[6], ay, o[7]);
        const float min1 = min4(r0, r1, r2, r3), max1 = max4(r0, r1, r2, r3);
        const float min2 = min4(q0, q1, q2, q3), max2 = max4(q0, q1, q2, q3);
        sep |= __builtin_amdgcn_ballot_w64(max1 < min2) | __builtin_amdgcn_ballot_w64(max2 < min1);
        const float need = fma_(__builtin_fabsf(ax) + __builtin_fabsf(ay), sc.skip_c3,
                                (__builtin_fabsf(ax + bx) + __builtin_fabsf(ay + by)) * sc.skip_c2) + 1e-36f;
        thin2[j] = __builtin_amdgcn_ballot_w64(max2 - min1 < need) | __builtin_amdgcn_ballot_w64(max1 - min2 < need);
    }
    if ((among & ~sep) == 0ull) return 0ull;
#pragma unroll
    for (int j = 0; j < 2; j++) {
        if ((thin2[j] & among & ~sep) != 0ull) {  // axis j + 6 in full
            C2D_MC_STAT(6, 1);
            const float bx = o[(2 * j + 6) & 7] - o[2 * j + 4], by = o[(2 * j + 7) & 7] - o[2 * j + 5];
            sep |= axis_separates_mask(bx, by, sc.robot, o);
        }
    }
    return among & ~sep;
}
#endif

// those of `lanes` whose sample collides (no axis separates it): convex_collide(robot, obstacle), utils.cu:159-184.
//
// All eight axes are part of the reference's result.  Two kinds of work are left out here, neither of which can change a
// lane's answer:
//  * once every lane of the wave is separated, the remaining axes are skipped (checked after the robot's and after the
//    obstacle's first axis pair);
//  * a rectangle's edge axes 2, 3 are the negatives of its axes 0, 1 up to rounding: b = -a + d with |d| a few ulps of the
//    coordinates.  For every vertex v the computed projections satisfy |p_b(v) + p_a(v)| <= eta,
//        eta = |d|_1 C (1 + 3u) + 4u (1 + u) |a|_1 C,        u = 2^-24, C >= every |coordinate|,
//    (d.v plus the roundings of the two dot products, 2u (|a_x v_x| + |a_y v_y|) each), hence max1_b >= -min1_a - eta,
//    min2_b <= -max2_a + eta and likewise with 1, 2 exchanged: if both overlaps on axis a, max2_a - min1_a and
//    max1_a - min2_a, are at least 2 eta, NEITHER comparison of utils.cu:178 can hold on axis b.  d is taken from the
//    floats themselves (a + b, computed per pair of axes), C from the scene (make_scene), the thresholds carry a 2^-8
//    relative and a 1e-36 absolute allowance for their own rounding and for underflow.  Axis b is evaluated in full
//    whenever an undecided lane's overlap on axis a is thinner than that (a few waves in a hundred on the bench scenes).
// Validation builds (C2D_MC_NO_AXIS_SKIP, part of `make lib-nopretest`) evaluate every axis; tests/test_gpu_fullsize.py
// compares the two builds on whole workloads.
C2D_DEV unsigned long long sample_collides_mask(const Scene& sc, const float (&o)[8], unsigned long long lanes)
{
#ifdef C2D_MC_NO_AXIS_SKIP
    unsigned long long sep = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        float ax, ay;
        if (i < 4) {
            ax = sc.robot[(2 * i + 2) & 7] - sc.robot[2 * i];
            ay = sc.robot[(2 * i + 3) & 7] - sc.robot[2 * i + 1];
        } else {
            const int j = i - 4;
            ax = o[(2 * j + 2) & 7] - o[2 * j];
            ay = o[(2 * j + 3) & 7] - o[2 * j + 1];
        }
        sep |= axis_separates_mask(ax, ay, sc.robot, o);
        if (((C2D_MC_EARLY_MASK >> i) & 1) && (lanes & ~sep) == 0ull) return 0ull;
    }
    return lanes & ~sep;
#else
    const unsigned long long alive = robot_axes_survivors(sc, o, lanes);
    if (alive == 0ull) return 0ull;
    return obstacle_axes_survivors(sc, o, alive);
#endif
}

// Per-wave LDS: the scene fields parked by lane 0 (below) and the sample queues of the NEAR / FAR loops (c2d_mc_core.hpp)
struct WaveQueue {
    // Scene fields parked by lane 0 and read back as broadcast ds_read_b128 where they are used, instead of living in VGPRs
    // through every sample loop: ev[0..2] what every full evaluation reads (park_scene / load_eval; the adaptive kernels only),
    // ev[3..6] what only the vertex arithmetic behind a thin closed-form result reads (park_exact / load_exact; every kernel)
    float4 ev[7];
    SampleQueues sq;
};

// ---- parking.  The scene is wave-uniform, yet each of its fields occupies a VGPR (no scalar floating point on gfx950).
// Two groups of fields leave the registers:
//  * the EXACT group (robot vertices, parallel-axis certificates) is read only by the vertex arithmetic behind a thin
//    closed-form result (evaluate_samples), about one evaluation pass in a hundred: parked by every kernel (park_exact);
//  * the EVAL group (the other sigmas, the obstacle's extents, the robot's model) is read by every full evaluation: parked by
//    the adaptive kernels, whose 80-VGPR budget (six waves per SIMD) the scene does not fit — with it in registers they spilled
//    17-21 dwords and reloaded some inside the sample loops (profiles/r03_mc_isa.md) — and kept in registers by mc_pair_kernel,
//    where every sample of the config-3 scene is evaluated and the extra LDS reads cost 3 %.
// The functions return the scene with the parked fields cleared, so that their registers are free in between.
C2D_DEV Scene park_exact(const Scene& sc, WaveQueue& q)
{
    if ((threadIdx.x & 63) == 0) {
        q.ev[3] = make_float4(sc.robot[0], sc.robot[1], sc.robot[2], sc.robot[3]);
        q.ev[4] = make_float4(sc.robot[4], sc.robot[5], sc.robot[6], sc.robot[7]);
        q.ev[5] = make_float4(sc.skip_lo[0], sc.skip_lo[1], sc.skip_hi[0], sc.skip_hi[1]);
        q.ev[6] = make_float4(sc.skip_c2, sc.skip_c3, 0.0f, 0.0f);
    }
    wave_lds_sync();
    Scene h = sc;
#pragma unroll
    for (int k = 0; k < 8; k++) h.robot[k] = 0.0f;
    h.skip_c2 = h.skip_c3 = 0.0f;
#pragma unroll
    for (int i = 0; i < 2; i++) h.skip_lo[i] = h.skip_hi[i] = 0.0f;
    return h;
}

C2D_DEV Scene park_scene(const Scene& sc, WaveQueue& q)
{
    if ((threadIdx.x & 63) == 0) {
        q.ev[0] = make_float4(sc.st, sc.sw, sc.sh, sc.margin);
        q.ev[1] = make_float4(sc.hw, sc.hh, sc.mhw, sc.mhh);
        q.ev[2] = make_float4(sc.mrx, sc.mry, sc.mcr, sc.msr);
    }
    Scene h = park_exact(sc, q);
    h.hw = h.hh = h.st = h.sw = h.sh = 0.0f;
    h.mrx = h.mry = h.mcr = h.msr = h.mhw = h.mhh = h.margin = 0.0f;
    return h;
}

// the scene with its EVAL group read back (the compiler barrier keeps the reads inside the loop they are used in)
template <bool PARKED>
C2D_DEV Scene load_eval(const Scene& hot, const WaveQueue& q)
{
    if constexpr (!PARKED) return hot;
    asm volatile("" ::: "memory");
    Scene sc = hot;
    const float4 a = q.ev[0], b = q.ev[1], c = q.ev[2];
    sc.st = a.x; sc.sw = a.y; sc.sh = a.z; sc.margin = a.w;
    sc.hw = b.x; sc.hh = b.y; sc.mhw = b.z; sc.mhh = b.w;
    sc.mrx = c.x; sc.mry = c.y; sc.mcr = c.z; sc.msr = c.w;
    return sc;
}

// ... and with its EXACT group
C2D_DEV Scene load_exact(const Scene& ev, const WaveQueue& q)
{
    asm volatile("" ::: "memory");
    Scene sc = ev;
    const float4 a = q.ev[3], b = q.ev[4], c = q.ev[5], d = q.ev[6];
    sc.robot[0] = a.x; sc.robot[1] = a.y; sc.robot[2] = a.z; sc.robot[3] = a.w;
    sc.robot[4] = b.x; sc.robot[5] = b.y; sc.robot[6] = b.z; sc.robot[7] = b.w;
    sc.skip_lo[0] = c.x; sc.skip_lo[1] = c.y; sc.skip_hi[0] = c.z; sc.skip_hi[1] = c.w;
    sc.skip_c2 = d.x; sc.skip_c3 = d.y;
    return sc;
}

// ---- the closed-form test of a full evaluation -------------------------------------------------------------------------------
// Both rectangles are BUILT from a centre, a rotation (c, s) and half extents (rect_from_half_extents), so the reference's
// eight axes are, up to rounding, the four directions e of the two frames, each twice, and convex_collide's answer on an axis
// pair is the sign of the closed-form gap
//     G_e = |e . (centre_o - centre_r)| - |e . U_r| - |e . V_r| - |e . U_o| - |e . V_o|,      U = hx (c, s), V = hy (-s, c)
// whenever |G_e| exceeds what rounding can move.  Take the real rectangles M with exactly these centres and half axes (the
// floats c, s, hx, hy read as reals).  With u = 2^-24, C >= every |centre coordinate| + |hx| + |hy| and h = |half extent| of
// the edge in question:
//   * a computed vertex coordinate is within 3uC of M's (two products, their sum, the translation);
//   * a computed edge vector a^ = v_k+1 - v_k is within 8uC per component of M's edge n = +-2U or +-2V, so |a^ - n|_1 <= 16uC;
//   * a computed projection p^(v^) = fl(fl(a^x v^x) + fl(a^y v^y)) is within E = 5.1u |a^|_1 C + 16.1u C^2 of n . v
//     (its two roundings, the edge error times the coordinates, the vertex error times the edge);
//   * hence each of the four interval ends of utils.cu:176-177 is within E of M's, and the two overlaps of utils.cu:178 within
//     2E of M's, whose smaller one is exactly -G(n) = -2h G_e: |G_e| > E / h fixes both comparisons on that edge AND on
//     the opposite edge (same n up to sign), E / h <= 14.5uC + 16.1u C^2 / h;
//   * G_e as evaluated below (fma where the choice is ours) is within 20uC of the real G_e, and writing |e . U| = h for the
//     rectangle's own edge instead of h (c^2 + s^2) costs at most 16uC (|c^2 + s^2 - 1| <= 2^-20 for sincos_).
// So with margin = (64 + 20 C / h_min) u C (make_scene; >= 1.24 x the sum above): max_e G_e > margin proves that the reference
// finds a separating axis, max_e G_e < -margin proves that it finds none, and anything in between — a NaN too — is "thin" and is
// evaluated with the reference's vertex arithmetic.  h_min >= 1e-12 keeps every product far above the subnormals.  Checked
// against the oracle on random and on razor-thin pairs in tests/test_oracle.py (a numpy restatement of this function) and by
// every Monte-Carlo test on the GPU; `make lib-nopretest` builds the kernels without it.
C2D_DEV float model_gap(const Scene& ev, float dx, float dy, float c, float s, float hx, float hy)
{
    const float q1 = fma_(ev.mcr, c, ev.msr * s);       // cos, sin of the angle between the frames
    const float q2 = fma_(ev.mcr, s, -(ev.msr * c));
    const float ex = dx - ev.mrx, ey = dy - ev.mry;
    const float t1 = fma_(ev.mcr, ex, ev.msr * ey);     // centre offset along the robot's axes ...
    const float t2 = fma_(ev.mcr, ey, -(ev.msr * ex));
    const float t3 = fma_(c, ex, s * ey);               // ... and along the obstacle's
    const float t4 = fma_(c, ey, -(s * ex));
    const float a1 = __builtin_fabsf(q1), a2 = __builtin_fabsf(q2), ax = __builtin_fabsf(hx), ay = __builtin_fabsf(hy);
    const float g1 = __builtin_fabsf(t1) - fma_(ax, a1, fma_(ay, a2, ev.mhw));
    const float g2 = __builtin_fabsf(t2) - fma_(ax, a2, fma_(ay, a1, ev.mhh));
    const float g3 = __builtin_fabsf(t3) - fma_(ev.mhw, a1, fma_(ev.mhh, a2, ax));
    const float g4 = __builtin_fabsf(t4) - fma_(ev.mhw, a2, fma_(ev.mhh, a1, ay));
    return __builtin_fmaxf(__builtin_fmaxf(g1, g2), __builtin_fmaxf(g3, g4));
}

// hits among the (up to 64, mask live_m) samples of a pass: second Box-Muller pair, rotation, the closed-form test, and for a
// pass with a thin sample the reference's vertex arithmetic (axes 0, 1 of each rectangle with the parallel-axis certificates,
// sample_collides_mask) on the thin lanes
template <bool PARKED>
C2D_DEV uint32_t evaluate_samples(const Scene& sc, WaveQueue& wq, uint32_t w2r, uint32_t w2a, float dx, float dy, uint64_t seed,
                                  uint64_t scene_id, uint64_t sample, unsigned long long live_m)
{
    const Scene ev = load_eval<PARKED>(sc, wq);
    C2D_MC_STAT(5, __popcll(live_m));
    float hx, hy, s, c;
    sample_shape(ev, w2r, w2a, seed, scene_id, sample, hx, hy, s, c);
    uint32_t hits = 0;
    unsigned long long thin = live_m;
#ifndef C2D_MC_NO_MODEL_TEST
    {
        const float g = model_gap(ev, dx, dy, c, s, hx, hy);
        const unsigned long long col = __builtin_amdgcn_ballot_w64(g < -ev.margin), sep = __builtin_amdgcn_ballot_w64(g > ev.margin);
        hits = (uint32_t)__popcll(col & live_m);
        thin = live_m & ~(col | sep);
        if (thin == 0ull) return hits;
    }
#endif
    C2D_MC_STAT(8, 1);
    const Scene ex = load_exact(ev, wq);
    float o[8];
    rect_from_half_extents(hx, hy, c, s, dx, dy, o);  // utils.cu:156
    return hits + (uint32_t)__popcll(sample_collides_mask(ex, o, thin));
}

#ifndef C2D_MC_PARK_ADAPTIVE
#define C2D_MC_PARK_ADAPTIVE 1  // 0: the adaptive kernels keep the whole scene in registers too (the A/B of profiles/r03_mc_isa.md)
#endif
constexpr bool kParkAdaptive = C2D_MC_PARK_ADAPTIVE != 0;
C2D_DEV Scene adaptive_scene(const Scene& sc, WaveQueue& q)
{
    if constexpr (kParkAdaptive) return park_scene(sc, q);
    else return park_exact(sc, q);
}

// ---- PLAIN: a scene that is not tame (Scene::tame).  One sample per lane, every sample evaluated in full with the
// axis test that is defined for every bit pattern (rect_collide, c2d_math.hpp); each lane draws its own blocks.  Slow
// (four Philox blocks per sample) and rare by construction: tables of finite numbers never come here.
template <bool PARKED>
C2D_DEV uint32_t wave_count_hits_plain(const Scene& sc, uint64_t seed, uint64_t scene_id, uint64_t begin, uint32_t count, const WaveQueue& wq)
{
    const uint32_t lane = threadIdx.x & 63;
    uint32_t hits = 0;
#pragma nounroll
    for (uint32_t i = 0; i < count; i += 64) {
        const bool live = i + lane < count;
        const uint64_t s = begin + i + lane;
        const uint32_t j = (uint32_t)s & 3u;
        const U4 b0 = philox_draw_block(seed, scene_id, s >> 2, 0), b1 = philox_draw_block(seed, scene_id, s >> 2, 1);
        const U4 b2 = philox_draw_block(seed, scene_id, s >> 2, 2u + (j >> 1));
        float dx, dy, o[8];
        sample_centre(sc, u4_word(b0, (int)j), u4_word(b1, (int)j), dx, dy);
        const Scene ev = load_exact(load_eval<PARKED>(sc, wq), wq);
        float hx, hy, sn, cs;
        sample_shape(ev, (j & 1u) ? b2.z : b2.x, (j & 1u) ? b2.w : b2.y, seed, scene_id, s, hx, hy, sn, cs);
        rect_from_half_extents(hx, hy, cs, sn, dx, dy, o);  // utils.cu:156
        hits += (uint32_t)__popcll(__ballot(live && rect_collide(ev.robot, o)));
    }
    return hits;
}

// the rectangle case as a policy of the sample loops (c2d_mc_core.hpp)
template <bool PARKED>
struct RectPolicy {
    using Scene = c2d::Scene;
    using Queue = WaveQueue;
    static C2D_DEV bool centre_pretest(const Scene& sc, const Queue&, float dx, float dy, unsigned long long& miss_m) { return c2d::centre_pretest(sc, dx, dy, miss_m); }
    static C2D_DEV uint32_t evaluate(const Scene& sc, Queue& wq, uint32_t w2r, uint32_t w2a, float dx, float dy, uint64_t seed, uint64_t scene_id, uint64_t sample,
                                     unsigned long long live_m)
    {
        return evaluate_samples<PARKED>(sc, wq, w2r, w2a, dx, dy, seed, scene_id, sample, live_m);
    }
    static C2D_DEV uint32_t plain(const Scene& sc, uint64_t seed, uint64_t scene_id, uint64_t begin, uint32_t count, const Queue& wq)
    {
        return wave_count_hits_plain<PARKED>(sc, seed, scene_id, begin, count, wq);
    }
    static C2D_DEV uint32_t finish(const Scene&, Queue&) { return 0u; }  // (a rectangle's evaluation is one stage: nothing is held back)
};

// ---- one scene, sample-parallel (BASELINE config 3) -------------------------------
struct PairArgs {
    float robot_w, robot_h, px, py;
    Pose pose;
    StdDev sd;
    uint64_t seed, scene_id, sample_begin, n_samples;
    uint32_t chunk;  // samples per wave, multiple of 256 (whole iterations of wave_count_hits)
};

#ifndef C2D_MC_PAIR_WAVES
#define C2D_MC_PAIR_WAVES 5
#endif
__global__ __launch_bounds__(kMcBlock, C2D_MC_PAIR_WAVES) void mc_pair_kernel(PairArgs A, unsigned long long* __restrict__ d_hits)
{
    __shared__ WaveQueue s_queue[kWavesPerBlock];
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const Scene sc = park_exact(make_scene(A.robot_w, A.robot_h, A.px, A.py, A.pose, A.sd), s_queue[wave]);  // (EVAL group in registers: see park_scene)
    const uint64_t n_chunks = (A.n_samples + A.chunk - 1) / A.chunk;
    unsigned long long total = 0;
    C2D_MC_CLOCK_START();
    for (uint64_t ch = (uint64_t)blockIdx.x * kWavesPerBlock + wave; ch < n_chunks; ch += (uint64_t)gridDim.x * kWavesPerBlock) {
        const uint64_t off = ch * A.chunk;
        const uint64_t left = A.n_samples - off;
        const uint32_t count = left < A.chunk ? (uint32_t)left : A.chunk;
        total += wave_count_hits<RectPolicy<false>>(sc, A.seed, A.scene_id, A.sample_begin + off, count, s_queue[wave]);
    }
    C2D_MC_CLOCK_STOP(c2d_mc_clock_pair);
    if ((threadIdx.x & 63) == 0 && total) atomicAdd(d_hits, total);
}

// ---- many scenes, adaptive (BASELINE config 4): the schedule state, the advance kernel's body and the host loop are shared with
// the polygon kernels (c2d_mc_core.hpp); here are the rectangle scene builder, the advance kernel around it, and the kernels
// that do not depend on the shape (init, decide).
struct ScenesArgs {
    const Pose* poses;
    const StdDev* std_devs;
    const PositionWithVarAndPoseIdx* scenes;
    AdaptiveState* state;
    uint32_t* lists[2];       // active-scene index lists (ping-pong)
    uint32_t num_poses, num_std_devs;
    float robot_w, robot_h;
    uint64_t seed, scene_id_base;
    ScheduleArgs sched;
    uint32_t* hits;           // u32[n_scenes], accumulated
    // burst mode (first launch only; burst_steps <= 1: off)
    uint32_t burst_steps;
    uint32_t n_bins;
    float bins[16], acc[16];
    uint32_t* n_used;
    PoseCPVarAndPoseIdx* rows;  // may be NULL
};

// the rectangle scene of one dataset row (ccp.cu:119-133)
struct RectBuilder {
    using Args = ScenesArgs;
    using Policy = RectPolicy<kParkAdaptive>;
#ifdef C2D_MC_CLOCK
    static C2D_DEV unsigned long long* clock_words() { return c2d_mc_clock_scenes; }
#endif
    static C2D_DEV Scene scene(const Args& A, const PositionWithVarAndPoseIdx& row, WaveQueue& q)
    {
        // float -> int index conversion as in ccp.cu:121-122; clamped so that a malformed row cannot read outside the tables
        uint32_t pi = (uint32_t)(int)row.pose_idx, vi = (uint32_t)(int)row.var_idx;
        pi = pi < A.num_poses ? pi : A.num_poses - 1;
        vi = vi < A.num_std_devs ? vi : A.num_std_devs - 1;
        return adaptive_scene(make_scene(A.robot_w, A.robot_h, row.x, row.y, A.poses[pi], A.std_devs[vi]), q);
    }
};

__global__ void mc_scenes_init_kernel(AdaptiveState* state, uint32_t n_scenes)
{
    AdaptiveState init{};
    init.n_active = n_scenes;
    init.identity = 1;
    *state = init;
}

// 7 waves per SIMD (72 VGPRs) instead of the 5 the compiler would take: the certain-miss path is a chain of dependent Philox
// multiplies, and more waves hide it.  The sixth wave was worth 3-5 % on the config-4 workload (412 -> 400 ms per 4e6 data
// points) and on the reference-default batch; since the closed-form evaluation (model_gap) shortened the live ranges of the
// full evaluation the seventh is worth another 3 % (336 -> 325 ms, 33.1 -> 32.5 ms; 5 waves: 349 ms, 8 waves spill inside the
// sample loops).  At 72 registers 16-17 dwords spill: every one of their stores and reloads sits outside the sample loops, in
// the per-work-item code (profiles/r03_mc_isa.md lists them with their loop depth).
#ifndef C2D_MC_ADV_WAVES
#define C2D_MC_ADV_WAVES 7
#endif
template <bool BURST>
__global__ __launch_bounds__(kMcBlock, C2D_MC_ADV_WAVES) void mc_scenes_advance_kernel(ScenesArgs A) { mc_scenes_advance_body<BURST, RectBuilder>(A); }

__global__ __launch_bounds__(256) void mc_scenes_decide_kernel(DecideArgs A)
{
    const uint32_t n_active = A.state->n_active;
    if (n_active == 0) return;
    const uint32_t n_start = A.state->n_samples;
    if (n_start >= A.sched.max_samples) return;
    const uint32_t n_batch = batch_of(A.sched, n_start);
    const bool burst = A.burst_steps > 1;
    const uint32_t n = n_start + (burst ? A.burst_steps * n_batch : n_batch);
    const bool identity = A.state->identity != 0;
    const uint32_t sel = A.state->list_sel;
    const uint32_t* active = identity ? nullptr : A.lists[sel];
    uint32_t* next = A.lists[identity ? 0 : (sel ^ 1u)];
    unsigned long long drawn = 0;   // burst: samples this thread's scenes drew
    uint32_t max_steps = 0;
    __shared__ float s_bins[16], s_acc[16];
    if (threadIdx.x < 16) { s_bins[threadIdx.x] = A.bins[threadIdx.x]; s_acc[threadIdx.x] = A.acc[threadIdx.x]; }
    __syncthreads();

    const uint32_t rounds = (n_active + gridDim.x * blockDim.x - 1) / (gridDim.x * blockDim.x);
    for (uint32_t r = 0; r < rounds; r++) {  // every lane takes part in every round's ballot
        const uint32_t slot = (r * gridDim.x + blockIdx.x) * blockDim.x + threadIdx.x;
        bool survive = false;
        uint32_t g = 0;
        if (slot < n_active && burst) {
            g = active ? active[slot] : slot;
            const uint32_t nu = A.n_used[g];   // != 0: the scene's wave found its stop test passing after nu samples
            survive = nu == 0;
            const uint32_t n_g = survive ? n : nu;
            drawn += n_g - n_start;
            const uint32_t steps_g = (n_g - n_start) / n_batch;
            max_steps = steps_g > max_steps ? steps_g : max_steps;
        } else if (slot < n_active) {
            g = active ? active[slot] : slot;
            const uint32_t k = A.hits[g];
            const float slack = calc_slack(n, k);                         // ccp.cu:140
            const float p = (float)k / (float)n;                          // ccp.cu:142
            const bool done = slack <= s_acc[get_bin(p, s_bins, A.n_bins)];  // ccp.cu:144
            if (done || n >= A.sched.max_samples) {
                A.n_used[g] = n;
                if (A.rows) {
                    const PositionWithVarAndPoseIdx row = A.scenes[g];
                    PoseCPVarAndPoseIdx o;
                    o.x = row.x; o.y = row.y; o.cp = p; o.var_idx = row.var_idx; o.pose_idx = row.pose_idx;
                    A.rows[g] = o;
                }
            } else {
                survive = true;
            }
        }
        // wave-aggregated append: one atomic per wave
        const unsigned long long m = __ballot(survive);
        if (m) {
            const uint32_t lane = threadIdx.x & 63;
            uint32_t base = 0;
            if (lane == (uint32_t)__builtin_ctzll(m)) base = atomicAdd(&A.state->next_count, (uint32_t)__popcll(m));
            base = __shfl(base, __builtin_ctzll(m), 64);
            if (survive) next[base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = g;
        }
    }
    // The block whose ticket is the last one rolls the state.  Every append above is a
    // returning device-scope atomic that has completed, and the fence orders this block's
    // list stores before its ticket; the next kernel starts behind the kernel boundary.
    if (burst) {  // wave-reduced: samples drawn and the largest number of steps any scene needed
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            drawn += __shfl_down(drawn, off, 64);
            const uint32_t o = (uint32_t)__shfl_down((int)max_steps, off, 64);
            max_steps = o > max_steps ? o : max_steps;
        }
        if ((threadIdx.x & 63) == 0 && drawn) {
            atomicAdd(&A.state->total_samples, drawn);
            atomicMax(&A.state->burst_steps, max_steps);
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();
        const uint32_t t = atomicAdd(&A.state->ticket, 1u);
        if (t == gridDim.x - 1) {
            const uint32_t survivors = atomicExch(&A.state->next_count, 0u);
            if (!burst) A.state->total_samples += (unsigned long long)n_active * n_batch;
            A.state->n_active = survivors;
            A.state->n_samples = n;
            A.state->iter += burst ? atomicExch(&A.state->burst_steps, 0u) : 1u;
            A.state->ticket = 0;
            A.state->list_sel = identity ? 0u : (sel ^ 1u);
            A.state->identity = 0;
        }
    }
}

// ---- RNG parity/debug kernel ----------------------------------------------------------
__global__ void philox_normals_kernel(uint64_t seed, uint64_t scene_id, uint64_t sample_begin, size_t n,
                                      float* __restrict__ normals, uint32_t* __restrict__ raw)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        // the six words of the sample in draw order (draw layout: c2d_math.hpp)
        const uint64_t s = sample_begin + i, g = s >> 2;
        const int j = (int)(s & 3);
        const U4 b0 = philox_draw_block(seed, scene_id, g, 0), b1 = philox_draw_block(seed, scene_id, g, 1);
        const U4 b2 = philox_draw_block(seed, scene_id, g, 2 + (j >> 1)), b3 = philox_draw_block(seed, scene_id, g, 4 + (j >> 1));
        const uint32_t w[6] = {u4_word(b0, j), u4_word(b1, j), (j & 1) ? b2.z : b2.x, (j & 1) ? b2.w : b2.y,
                               (j & 1) ? b3.z : b3.x, (j & 1) ? b3.w : b3.y};
        float v[6];
        box_muller(w[0], w[1], v[0], v[1]);
        box_muller(w[2], w[3], v[2], v[3]);
        box_muller(w[4], w[5], v[4], v[5]);
        for (int k = 0; k < 5; k++) normals[i * 5 + k] = v[k];
        if (raw)
            for (int k = 0; k < 6; k++) raw[i * 6 + k] = w[k];
    }
}

// ---- canonical-math parity hook -------------------------------------------------------------
__global__ void math_eval_kernel(int fn, const uint32_t* __restrict__ in, size_t n, float* __restrict__ out0,
                                 float* __restrict__ out1)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const uint32_t bits = in[i];
        const float x = __uint_as_float(bits);
        float a = 0.0f, b = 0.0f;
        switch (fn) {
        case C2D_MATH_LOG: a = log_(x); break;
        case C2D_MATH_SINCOS: sincos_(x, a, b); break;
        case C2D_MATH_SINCOS_U32: sincos_u32(bits, a, b); break;
        case C2D_MATH_SQRT: a = sqrt_normal_range(x); break;
        case C2D_MATH_BOX_MULLER: box_muller(bits, ~bits * 2654435761u, a, b); break;
        default: break;
        }
        out0[i] = a;
        if (out1) out1[i] = b;
    }
}

// ---- scene sampler (reference generate_dataset.cu:207-219) ------------------------------
constexpr uint64_t kSceneDomain = 0x5ce9e5a3c0117de5ull;

__global__ void sample_scenes_kernel(const Pose* __restrict__ poses, uint32_t num_poses,
                                     const StdDev* __restrict__ std_devs, uint32_t num_std_devs, float r_offset,
                                     float spread, uint64_t seed, uint64_t scene_id_base, size_t n,
                                     PositionWithVarAndPoseIdx* __restrict__ scenes)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x; g < n; g += stride) {
        const U4 a = philox_block(seed ^ kSceneDomain, scene_id_base + g, 0, 0);
        const U4 b = philox_block(seed ^ kSceneDomain, scene_id_base + g, 0, 1);
        const uint32_t pose_idx = a.x % num_poses;      // :208
        const uint32_t sd_idx = a.y % num_std_devs;     // :209
        const Pose pose = poses[pose_idx];
        const StdDev sd = std_devs[sd_idx];
        const float u = fma_((float)a.z, 0x1p-32f, 0x1p-33f);
        const float theta = (float)((double)u * 2 * 3.14159265358979323846);  // :213
        float nrm, unused;
        box_muller(b.x, b.y, nrm, unused);
        const float shift = nrm * ((sd.y + sd.x) / 2) * spread;               // :214
        float st, ct;
        sincos_(theta, st, ct);
        const double bx = ((double)(pose.width / 2 + r_offset) + 2.35 + (double)sd.x) + (double)shift;   // :215
        const double by = ((double)(pose.height / 2 + r_offset) + 2.35 + (double)sd.y) + (double)shift;  // :216
        PositionWithVarAndPoseIdx row;
        row.x = (float)((double)ct * bx);
        row.y = (float)((double)st * by);
        row.var_idx = (float)sd_idx;
        row.pose_idx = (float)pose_idx;
        scenes[g] = row;
    }
}

void launch_mc_scenes_init(hipStream_t s, AdaptiveState* state, uint32_t n_scenes)
{
    hipLaunchKernelGGL(mc_scenes_init_kernel, dim3(1), dim3(1), 0, s, state, n_scenes);
}

void launch_mc_scenes_decide(hipStream_t s, unsigned blocks, const DecideArgs& D)
{
    hipLaunchKernelGGL(mc_scenes_decide_kernel, dim3(blocks), dim3(256), 0, s, D);
}

int ensure_scene_lists(c2d_ctx* ctx, size_t n)
{
    if (ctx->list_capacity >= n) return C2D_OK;
    for (auto& p : ctx->d_list) {
        if (p) (void)hipFree(p);
        p = nullptr;
    }
    ctx->list_capacity = 0;
    for (auto& p : ctx->d_list) {
        hipError_t e = hipMalloc(&p, n * sizeof(uint32_t));
        if (e != hipSuccess) { ctx->last_error = "c2d_mc_scenes: workspace allocation failed"; return C2D_ERR_NOMEM; }
    }
    ctx->list_capacity = n;
    return C2D_OK;
}

}  // namespace c2d

using namespace c2d;

extern "C" {

int c2d_math_eval(c2d_ctx* ctx, int fn, const uint32_t* d_in_bits, size_t n, float* d_out0, float* d_out1, c2d_stream stream)
{
    if (!ctx) return C2D_ERR_INVALID_ARG;
    if (n == 0) return C2D_OK;
    if (!d_in_bits || !d_out0 || fn < C2D_MATH_LOG || fn > C2D_MATH_BOX_MULLER) return fail_arg(ctx, "c2d_math_eval: bad argument");
    DeviceGuard g(ctx->device);
    hipLaunchKernelGGL(math_eval_kernel, dim3(grid_for(n, 256, ctx->prop.multiProcessorCount * 8)), dim3(256), 0,
                       (hipStream_t)stream, fn, d_in_bits, n, d_out0, d_out1);
    C2D_LAUNCH_CHECK(ctx);
    return C2D_OK;
}

int c2d_philox_normals(c2d_ctx* ctx, uint64_t seed, uint64_t scene_id, uint64_t sample_begin, size_t n,
                       float* d_normals, uint32_t* d_raw, c2d_stream stream)
{
    if (!ctx) return C2D_ERR_INVALID_ARG;
    if (n == 0) return C2D_OK;
    if (!d_normals) return fail_arg(ctx, "c2d_philox_normals: NULL output");
    DeviceGuard g(ctx->device);
    hipLaunchKernelGGL(philox_normals_kernel, dim3(grid_for(n, 256, ctx->prop.multiProcessorCount * 8)), dim3(256), 0,
                       (hipStream_t)stream, seed, scene_id, sample_begin, n, d_normals, d_raw);
    C2D_LAUNCH_CHECK(ctx);
    return C2D_OK;
}

int c2d_mc_pair(c2d_ctx* ctx, float robot_w, float robot_h, const Position* pos, const Pose* pose,
                const StdDev* std_dev, uint64_t seed, uint64_t scene_id, uint64_t sample_begin, uint64_t n_samples,
                unsigned long long* d_hits, c2d_stream stream)
{
    if (!ctx) return C2D_ERR_INVALID_ARG;
    if (!pos || !pose || !std_dev || !d_hits) return fail_arg(ctx, "c2d_mc_pair: NULL argument");
    if (n_samples == 0) return C2D_OK;
    if (sample_begin + n_samples < sample_begin || sample_begin + n_samples > (1ull << 62))
        return fail_arg(ctx, "c2d_mc_pair: sample range overflows the 2^62-sample stream");
    PairArgs A;
    A.robot_w = robot_w; A.robot_h = robot_h; A.px = pos->x; A.py = pos->y;
    A.pose = *pose; A.sd = *std_dev;
    A.seed = seed; A.scene_id = scene_id; A.sample_begin = sample_begin; A.n_samples = n_samples;
    // chunk: enough samples per wave to amortise scene set-up, enough waves to fill the chip
    const uint64_t target_waves = (uint64_t)ctx->prop.multiProcessorCount * 32;
    uint64_t chunk = (n_samples + target_waves - 1) / target_waves;
    chunk = ((chunk + 255) / 256) * 256;
    if (chunk < 256) chunk = 256;
    if (chunk > 8192) chunk = 8192;
    A.chunk = (uint32_t)chunk;
    const uint64_t n_chunks = (n_samples + chunk - 1) / chunk;
    uint64_t blocks = (n_chunks + kWavesPerBlock - 1) / kWavesPerBlock;
    const uint64_t max_blocks = (uint64_t)ctx->prop.multiProcessorCount * 256 / kWavesPerBlock;  // 256 waves per CU
    if (blocks > max_blocks) blocks = max_blocks;
    DeviceGuard g(ctx->device);
    hipLaunchKernelGGL(mc_pair_kernel, dim3((unsigned)blocks), dim3(kMcBlock), 0, (hipStream_t)stream, A, d_hits);
    C2D_LAUNCH_CHECK(ctx);
    return C2D_OK;
}

int c2d_mc_scenes(c2d_ctx* ctx, const c2d_mc_scenes_args* a, c2d_stream stream)
{
    if (!ctx) return C2D_ERR_INVALID_ARG;
    if (!a) return fail_arg(ctx, "c2d_mc_scenes: NULL args");
    if (a->n_scenes != 0 && (!a->d_poses || a->num_poses == 0)) {
        if (a->total_samples) *a->total_samples = 0;
        if (a->iterations) *a->iterations = 0;
        return fail_arg(ctx, a->d_poses ? "c2d_mc_scenes: empty pose / std_dev table" : "c2d_mc_scenes: NULL argument");
    }
    ScenesArgs A;
    A.poses = a->d_poses; A.num_poses = a->num_poses; A.robot_w = a->robot_w; A.robot_h = a->robot_h;
    return run_adaptive(ctx, a, stream, A, [](bool burst, unsigned blocks, const ScenesArgs& args, hipStream_t s) {
        if (burst) hipLaunchKernelGGL(mc_scenes_advance_kernel<true>, dim3(blocks), dim3(kMcBlock), 0, s, args);
        else hipLaunchKernelGGL(mc_scenes_advance_kernel<false>, dim3(blocks), dim3(kMcBlock), 0, s, args);
    }, "c2d_mc_scenes");
}

int c2d_sample_scenes(c2d_ctx* ctx, const Pose* d_poses, uint32_t num_poses, const StdDev* d_std_devs,
                      uint32_t num_std_devs, float robot_w, float robot_h, float spread, uint64_t seed,
                      uint64_t scene_id_base, size_t n_scenes, PositionWithVarAndPoseIdx* d_scenes, c2d_stream stream)
{
    if (!ctx) return C2D_ERR_INVALID_ARG;
    if (n_scenes == 0) return C2D_OK;
    if (!d_poses || !d_std_devs || !d_scenes || num_poses == 0 || num_std_devs == 0)
        return fail_arg(ctx, "c2d_sample_scenes: NULL / empty argument");
    DeviceGuard g(ctx->device);
    const float r_offset = (robot_w + robot_h) / 4;  // generate_dataset.cu:398
    hipLaunchKernelGGL(sample_scenes_kernel, dim3(grid_for(n_scenes, 256, ctx->prop.multiProcessorCount * 8)), dim3(256), 0,
                       (hipStream_t)stream, d_poses, num_poses, d_std_devs, num_std_devs, r_offset, spread, seed,
                       scene_id_base, n_scenes, d_scenes);
    C2D_LAUNCH_CHECK(ctx);
    return C2D_OK;
}

#ifdef C2D_MC_CLOCK
// clock build only: the stamps of mc_pair_kernel (which = 0) or of mc_scenes_advance_kernel (1): shader cycles, 100 MHz ticks, waves
int c2d_debug_mc_clock(c2d_ctx* ctx, int which, unsigned long long out[4], int reset)
{
    if (!ctx || !out || which < 0 || which > 1) return C2D_ERR_INVALID_ARG;
    DeviceGuard g(ctx->device);
    C2D_HIP(ctx, hipDeviceSynchronize());
    const unsigned long long zero[4] = {};
    if (which == 0) {
        C2D_HIP(ctx, hipMemcpyFromSymbol(out, HIP_SYMBOL(c2d_mc_clock_pair), 4 * sizeof(unsigned long long)));
        if (reset) C2D_HIP(ctx, hipMemcpyToSymbol(HIP_SYMBOL(c2d_mc_clock_pair), zero, sizeof zero));
    } else {
        C2D_HIP(ctx, hipMemcpyFromSymbol(out, HIP_SYMBOL(c2d_mc_clock_scenes), 4 * sizeof(unsigned long long)));
        if (reset) C2D_HIP(ctx, hipMemcpyToSymbol(HIP_SYMBOL(c2d_mc_clock_scenes), zero, sizeof zero));
    }
    return C2D_OK;
}
#endif

#ifdef C2D_MC_STATS
// census build only: copies the twelve counters to the host (after synchronising the device) and optionally clears them
int c2d_debug_mc_stats(c2d_ctx* ctx, unsigned long long out[12], int reset)
{
    if (!ctx || !out) return C2D_ERR_INVALID_ARG;
    DeviceGuard g(ctx->device);
    C2D_HIP(ctx, hipDeviceSynchronize());
    C2D_HIP(ctx, hipMemcpyFromSymbol(out, HIP_SYMBOL(c2d_mc_stats_words), 12 * sizeof(unsigned long long)));
    if (reset) {
        const unsigned long long zero[12] = {};
        C2D_HIP(ctx, hipMemcpyToSymbol(HIP_SYMBOL(c2d_mc_stats_words), zero, sizeof zero));
    }
    return C2D_OK;
}
#endif

// Host mirrors of the stopping statistics (same expressions as the device code).
float c2d_calc_slack(uint32_t n, uint32_t k)
{
    if (k == n || k == 0) return (float)(0x1.d82d33932720dp+1 / (double)n);
    const float z = 1.96f;
    const float kf = (float)k;
    const float kk = (float)((uint64_t)k * (uint64_t)k);
    return z / (float)n * __builtin_sqrtf(kf - kk / (float)n);
}

int c2d_get_bin(float p, const float* accuracy_bins, uint32_t n_accuracy_bins)
{
    int bin = 0;
    if (!accuracy_bins) return 0;
    for (uint32_t i = 0; i + 1 < n_accuracy_bins; i++)
        if (p >= accuracy_bins[i] && p <= accuracy_bins[i + 1]) bin = (int)i;
    return bin;
}

}  // extern "C"
