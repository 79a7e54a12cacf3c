// c2d_math.hpp — device-side canonical arithmetic for gfx950 (MI355X).
//
// Everything here is IEEE binary32, round-to-nearest-even.  The translation
// unit is compiled with -ffp-contract=off, so a*b+c is two roundings unless
// written as __builtin_fmaf.  sin/cos/log are polynomial forms built from
// +,*,fma and integer ops only (coefficients: oracle/tools/fit_poly.py), so the
// same bits come out of any IEEE machine; sqrt and divide are the correctly
// rounded forms (hipcc default -fhip-fp32-correctly-rounded-divide-sqrt).
// DESIGN.md §"Canonical arithmetic" is the specification; the CPU oracle
// restates it independently.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace c2d {

#define C2D_DEV __device__ __forceinline__

C2D_DEV float fma_(float a, float b, float c) { return __builtin_fmaf(a, b, c); }

// ---- natural log of a positive normal float ---------------------------------
C2D_DEV float log_(float u)
{
    uint32_t t = __float_as_uint(u) - 0x3f2aaaabu;
    int32_t e = (int32_t)t >> 23;
    float m = __uint_as_float((t & 0x007fffffu) + 0x3f2aaaabu);
    float f = m - 1.0f;
    float q = -0x1.04cba2p-3f;
    q = fma_(q, f, 0x1.19bbe2p-3f);
    q = fma_(q, f, -0x1.f483fap-4f);
    q = fma_(q, f, 0x1.1fd494p-3f);
    q = fma_(q, f, -0x1.55913ep-3f);
    q = fma_(q, f, 0x1.99bffep-3f);
    q = fma_(q, f, -0x1.ffff28p-3f);
    q = fma_(q, f, 0x1.55552cp-2f);
    q = fma_(q, f, -0x1.000000p-1f);
    float s = f * f;
    float r = fma_(s, q, f);
    return fma_((float)e, 0x1.62e430p-1f, r);
}

// ---- correctly rounded sqrt for x in {+-0} U [2^-96, 2^96] ----------------------------------
// hipcc's own correctly rounded sqrtf is this refinement wrapped in denormal scaling and an
// inf/nan fix-up (16 instructions); the Box-Muller radius argument -2*log(u) never needs
// either (it is -0 or within [1.1e-7, 46]), so only the refinement is kept (9 instructions):
// v_sqrt_f32 is within 1 ulp, the two fma residuals pick the neighbour that rounds correctly.
// Bit-equal to IEEE sqrtf on that domain (exhaustive two-binade test in tests/test_gpu_mc.py).
C2D_DEV float sqrt_normal_range(float x)
{
    const float s = __builtin_amdgcn_sqrtf(x);
    const float s_dn = __uint_as_float(__float_as_uint(s) - 1u);
    const float s_up = __uint_as_float(__float_as_uint(s) + 1u);
    const float r_dn = fma_(-s_dn, s, x);
    const float r_up = fma_(-s_up, s, x);
    float r = r_dn <= 0.0f ? s_dn : s;
    r = r_up > 0.0f ? s_up : r;
    return r;
}

// (sn, cs) of the first quadrant -> quadrant q (angle + q*pi/2)
C2D_DEV void quadrant_rotate(int q, float sn, float cs, float& s_out, float& c_out)
{
    // q&1 swaps, then signs: sin negative for q in {2,3}, cos negative for q in {1,2}
    float a = (q & 1) ? cs : sn;
    float b = (q & 1) ? sn : cs;
    uint32_t s_sign = ((uint32_t)q & 2u) << 30;
    uint32_t c_sign = (((uint32_t)q + 1u) & 2u) << 30;
    s_out = __uint_as_float(__float_as_uint(a) ^ s_sign);
    c_out = __uint_as_float(__float_as_uint(b) ^ c_sign);
}

// ---- sin, cos of a finite float angle (stands in for cosf/sinf, reference utils.cu:133-134)
C2D_DEV void sincos_(float x, float& s_out, float& c_out)
{
    float k = __builtin_rintf(x * 0x1.45f306p-1f);
    float r = fma_(k, -0x1.920000p+0f, x);
    r = fma_(k, -0x1.fb4000p-12f, r);
    r = fma_(k, -0x1.4442d2p-24f, r);
    float z = r * r;
    float sp = 0x1.6dac7ap-19f;
    sp = fma_(sp, z, -0x1.a01376p-13f);
    sp = fma_(sp, z, 0x1.11110ep-7f);
    sp = fma_(sp, z, -0x1.555556p-3f);
    float sn = fma_(z * r, sp, r);
    float cp = -0x1.2476a8p-22f;
    cp = fma_(cp, z, 0x1.a012bap-16f);
    cp = fma_(cp, z, -0x1.6c16bcp-10f);
    cp = fma_(cp, z, 0x1.555556p-5f);
    cp = fma_(cp, z, -0x1.000000p-1f);
    float cs = fma_(z, cp, 1.0f);
    float kc = __builtin_amdgcn_fmed3f(k, -1073741824.0f, 1073741824.0f);
    int q = (int)kc;
    quadrant_rotate(q, sn, cs, s_out, c_out);
}

// ---- sin, cos of the angle 2*pi*y/2^32 ---------------------------------------
C2D_DEV void sincos_u32(uint32_t y, float& s_out, float& c_out)
{
    int q = (int)(y >> 30);
    uint32_t fr = y & 0x3fffffffu;
    bool swap = fr > 0x20000000u;
    fr = swap ? 0x40000000u - fr : fr;
    float x = (float)(int32_t)fr * 0x1p-30f;
    float z = x * x;
    float p = 0x1.4bb0a6p-13f;
    p = fma_(p, z, -0x1.32ca4ap-8f);
    p = fma_(p, z, 0x1.466bbap-4f);
    p = fma_(p, z, -0x1.4abbcep-1f);
    p = fma_(p, z, 0x1.921fb6p+0f);
    float sn = p * x;
    float c = 0x1.d99986p-11f;
    c = fma_(c, z, -0x1.55c4e6p-6f);
    c = fma_(c, z, 0x1.03c1dap-2f);
    c = fma_(c, z, -0x1.3bd3ccp+0f);
    c = fma_(c, z, 1.0f);
    float sn2 = swap ? c : sn;
    float cs2 = swap ? sn : c;
    quadrant_rotate(q, sn2, cs2, s_out, c_out);
}

// ---- Philox4x32-10 (Salmon et al. SC'11; constants / word order of rocRAND) ---
struct U4 { uint32_t x, y, z, w; };

C2D_DEV U4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1)
{
#pragma unroll
    for (int round = 0; round < 10; round++) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        // three-input xor in one v_bitop3_b32 (truth table 0x96); hipcc emits two v_xor otherwise
        uint32_t n0 = __builtin_amdgcn_bitop3_b32((uint32_t)(p1 >> 32), c1, k0, 0x96);
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = __builtin_amdgcn_bitop3_b32((uint32_t)(p0 >> 32), c3, k1, 0x96);
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    return U4{c0, c1, c2, c3};
}

// block j (0 or 1) of item `sample` in stream (seed, scene): the scene sampler's layout (sample_scenes_kernel)
C2D_DEV U4 philox_block(uint64_t seed, uint64_t scene, uint64_t sample, uint32_t j)
{
    uint64_t blk = 2 * sample + j;
    return philox4x32_10((uint32_t)blk, (uint32_t)(blk >> 32), (uint32_t)scene, (uint32_t)(scene >> 32),
                         (uint32_t)seed, (uint32_t)(seed >> 32));
}

// ---- draw layout of the Monte-Carlo loop --------------------------------------------------------------------
// The samples of a stream (seed, scene) are drawn in GROUPS OF FOUR: sample s belongs to group g = s >> 2 as member
// j = s & 3, and the group owns Philox blocks 8g .. 8g+5 of subsequence `scene` (rocRAND: offset = 4 * (8g + b)):
//   block 8g+0   word j          radius word of sample j's first Box-Muller pair  (-> dx, dy)
//   block 8g+1   word j          angle word of that pair
//   block 8g+2   words (x,y)     radius, angle word of sample 0's second pair     (-> dtheta, dw);  (z,w): sample 1's
//   block 8g+3   words (x,y)     ... sample 2's;  (z,w): sample 3's
//   block 8g+4,5 like 8g+2,3     third pair (-> dh, second normal unused); only evaluated when sigma_h != 0
// Why: the radius word alone proves most samples of a far scene to be certain misses (c2d_mc.hip, make_scene), and a
// Philox block costs ~65 instructions whatever is used of it.  With one block per sample that proof cost a block per
// sample; with the four radius words of a group in ONE block it costs a quarter.  A near scene still pays one block per
// sample (four blocks per group), a shape-variance scene 1.5 instead of 2.
C2D_DEV U4 philox_draw_block(uint64_t seed, uint64_t scene, uint64_t group, uint32_t b)
{
    uint64_t blk = 8 * group + b;
    return philox4x32_10((uint32_t)blk, (uint32_t)(blk >> 32), (uint32_t)scene, (uint32_t)(scene >> 32),
                         (uint32_t)seed, (uint32_t)(seed >> 32));
}

// word j (0..3) of a block; j is a compile-time constant wherever this is used in a hot loop
C2D_DEV uint32_t u4_word(const U4& v, int j) { return j == 0 ? v.x : (j == 1 ? v.y : (j == 2 ? v.z : v.w)); }

// Box-Muller: x -> radius, y -> angle; n0 uses sin, n1 cos (rocRAND's roles)
C2D_DEV void box_muller(uint32_t x, uint32_t y, float& n0, float& n1)
{
    float u = fma_((float)x, 0x1p-32f, 0x1p-33f);
    float rad = sqrt_normal_range(-2.0f * log_(u));
    float sn, cs;
    sincos_u32(y, sn, cs);
    n0 = sn * rad;
    n1 = cs * rad;
}

// ---- geometry -------------------------------------------------------------------

// The reference's two-product sums a*x + b*y (utils.cu:139-140 rotation, :173-174 projection).  C2D_FMAD = 0 is the
// canonical arithmetic of the product: two roundings for the products, one for the sum.  C2D_FMAD = 1 / 2 exist only in
// the validation builds (`make lib-fmad`): the two forms nvcc's default -fmad=true can give a CUDA build of the reference
// (left or right product fused), used to MEASURE the distance between the canonical choice and such a build
// (oracle/tools/fmad_study.py, tests/test_gpu_fmad.py).
#ifndef C2D_FMAD
#define C2D_FMAD 0
#endif
C2D_DEV float dot2(float a, float x, float b, float y)
{
#if C2D_FMAD == 1
    return fma_(a, x, b * y);
#elif C2D_FMAD == 2
    return fma_(b, y, a * x);
#else
    return a * x + b * y;
#endif
}

// create_rect (reference utils.cu:119-130) followed by rot_trans_rectangle
// (utils.cu:132-142) with cos/sin given.  The four vertices of a box with half
// extents (hx, hy) are (-+hx, -+hy); since rounding is sign-symmetric the
// reference's per-vertex expressions c*x - s*y + dx, s*x + c*y + dy reduce
// exactly to the shared products below.
C2D_DEV void rect_from_half_extents(float hx, float hy, float c, float s, float dx, float dy, float (&r)[8])
{
#if C2D_FMAD == 1   // x' = fma(c, x, -(s*y)), y' = fma(s, x, c*y); fma is sign-symmetric too
    float b = s * hy, q = c * hy;
    float t1 = fma_(c, hx, b), t2 = fma_(c, hx, -b), t3 = fma_(s, hx, q), t4 = fma_(s, hx, -q);
#elif C2D_FMAD == 2 // x' = fma(-s, y, c*x), y' = fma(c, y, s*x)
    float a = c * hx, p = s * hx;
    float t1 = fma_(s, hy, a), t2 = fma_(-s, hy, a), t3 = fma_(c, hy, p), t4 = fma_(-c, hy, p);
#else
    float a = c * hx, b = s * hy, p = s * hx, q = c * hy;
    float t1 = a + b, t2 = a - b, t3 = p + q, t4 = p - q;
#endif
    r[0] = dx - t2;  // (-a) - (-b) + dx
    r[1] = dy - t3;  // (-p) + (-q) + dy
    r[2] = t1 + dx;  //   a  - (-b) + dx
    r[3] = t4 + dy;  //   p  + (-q) + dy
    r[4] = t2 + dx;  //   a  -   b  + dx
    r[5] = t3 + dy;  //   p  +   q  + dy
    r[6] = dx - t1;  // (-a) -   b  + dx
    r[7] = dy - t4;  // (-p) +   q  + dy
}

C2D_DEV float min4(float a, float b, float c, float d) { return __builtin_fminf(__builtin_fminf(a, b), __builtin_fminf(c, d)); }
C2D_DEV float max4(float a, float b, float c, float d) { return __builtin_fmaxf(__builtin_fmaxf(a, b), __builtin_fmaxf(c, d)); }

// One SAT axis (reference utils.cu:172-180): unfused dots, strict <.
//
// Non-finite inputs.  thrust::minmax_element (utils.cu:176-177) is comparison based: both extremes start at element 0
// and a later element replaces one only when `<` says so.  A NaN at k > 0 is therefore skipped — exactly what
// v_min_f32 / v_max_f32 do — while a NaN at k = 0 stays to the end, makes both comparisons of :178 false and the axis
// "not separating".  The min/max instructions below would drop that NaN too, so the one case is restored by one
// unordered compare of the two first projections: with it the kernels follow the reference (and the oracle) for
// every input bit pattern, infinities and NaNs included (tests/test_gpu_sat.py::test_non_finite_vertices).
C2D_DEV bool first_projections_ordered(float p1_first, float p2_first) { return !__builtin_isunordered(p1_first, p2_first); }

C2D_DEV bool axis_separates(float ax, float ay, const float (&r1)[8], const float (&r2)[8])
{
    float p10 = dot2(ax, r1[0], ay, r1[1]), p11 = dot2(ax, r1[2], ay, r1[3]);
    float p12 = dot2(ax, r1[4], ay, r1[5]), p13 = dot2(ax, r1[6], ay, r1[7]);
    float p20 = dot2(ax, r2[0], ay, r2[1]), p21 = dot2(ax, r2[2], ay, r2[3]);
    float p22 = dot2(ax, r2[4], ay, r2[5]), p23 = dot2(ax, r2[6], ay, r2[7]);
    float min1 = min4(p10, p11, p12, p13), max1 = max4(p10, p11, p12, p13);
    float min2 = min4(p20, p21, p22, p23), max2 = max4(p20, p21, p22, p23);
    return ((max1 < min2) || (max2 < min1)) && first_projections_ordered(p10, p20);
}

// convex_collide (reference utils.cu:159-184): 8 edge-vector axes, all evaluated.
C2D_DEV bool rect_collide(const float (&r1)[8], const float (&r2)[8])
{
    bool sep = false;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        float ax = r1[(2 * i + 2) & 7] - r1[2 * i];
        float ay = r1[(2 * i + 3) & 7] - r1[2 * i + 1];
        sep |= axis_separates(ax, ay, r1, r2);
    }
#pragma unroll
    for (int i = 0; i < 4; i++) {
        float ax = r2[(2 * i + 2) & 7] - r2[2 * i];
        float ay = r2[(2 * i + 3) & 7] - r2[2 * i + 1];
        sep |= axis_separates(ax, ay, r1, r2);
    }
    return !sep;
}

// ---- convex_collide with certificates for the second axis of each parallel pair -----------------------------------------
// A rectangle's edge axes 2, 3 are the negatives of its axes 0, 1 up to rounding: b = -a + d with |d| a few ulps of the
// coordinates.  For every vertex v the COMPUTED projections satisfy |p_b(v) + p_a(v)| <= eta,
//     eta = |d|_1 C (1 + 3u) + 4u (1 + u) |a|_1 C,      u = 2^-24, C >= every |coordinate| of the pair
// (d.v plus the roundings of the two dot products), so max1_b >= -min1_a - eta, min2_b <= -max2_a + eta and likewise with 1, 2
// exchanged: when both overlaps on axis a, max2_a - min1_a and max1_a - min2_a, are at least 2 eta, NEITHER comparison of
// utils.cu:178 can hold on axis b.  This evaluates axes 0, 1 of both rectangles, takes d from the floats (a + b) and C from
// the sixteen coordinates, and reports `thin` when the pair is not separated by those four axes and some overlap is below its
// certificate (or not a number: the comparisons are written so that a NaN anywhere reads "thin") — the caller then evaluates
// the pair in full (rect_collide).  The thresholds carry 2^-8 relative and 1e-36 absolute slack for their own rounding and
// for underflow.  The Monte-Carlo kernels use the same certificates with scene-level constants (c2d_mc.hip).
C2D_DEV bool rect_collide_certified(const float (&r1)[8], const float (&r2)[8], bool& thin)
{
    float cmax = __builtin_fabsf(r1[0]);
#pragma unroll
    for (int k = 1; k < 8; k++) cmax = __builtin_fmaxf(cmax, __builtin_fabsf(r1[k]));
#pragma unroll
    for (int k = 0; k < 8; k++) cmax = __builtin_fmaxf(cmax, __builtin_fabsf(r2[k]));
    const float c2 = (2.0f + 0x1p-7f) * cmax, c3 = (8.0f + 0x1p-5f) * 0x1p-24f * cmax;
    bool sep = false, uneasy = false;
#pragma unroll
    for (int which = 0; which < 2; which++) {
        const float (&r)[8] = which == 0 ? r1 : r2;
#pragma unroll
        for (int i = 0; i < 2; i++) {
            const float ax = r[2 * i + 2] - r[2 * i], ay = r[2 * i + 3] - r[2 * i + 1];
            const float bx = r[(2 * i + 6) & 7] - r[2 * i + 4], by = r[(2 * i + 7) & 7] - r[2 * i + 5];  // edge i + 2
            const float p10 = dot2(ax, r1[0], ay, r1[1]), p11 = dot2(ax, r1[2], ay, r1[3]);
            const float p12 = dot2(ax, r1[4], ay, r1[5]), p13 = dot2(ax, r1[6], ay, r1[7]);
            const float p20 = dot2(ax, r2[0], ay, r2[1]), p21 = dot2(ax, r2[2], ay, r2[3]);
            const float p22 = dot2(ax, r2[4], ay, r2[5]), p23 = dot2(ax, r2[6], ay, r2[7]);
            const float min1 = min4(p10, p11, p12, p13), max1 = max4(p10, p11, p12, p13);
            const float min2 = min4(p20, p21, p22, p23), max2 = max4(p20, p21, p22, p23);
            sep |= ((max1 < min2) || (max2 < min1)) && first_projections_ordered(p10, p20);
            const float need = fma_(__builtin_fabsf(ax) + __builtin_fabsf(ay), c3, (__builtin_fabsf(ax + bx) + __builtin_fabsf(ay + by)) * c2) + 1e-36f;
            uneasy |= !((max2 - min1 >= need) && (max1 - min2 >= need));
        }
    }
    thin = !sep && uneasy;
    return !sep;
}

}  // namespace c2d
