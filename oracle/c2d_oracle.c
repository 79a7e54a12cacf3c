/*
 * c2d_oracle.c — CPU restatement of the reference's collision hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  This file is the parity checker: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.  The
 * product (libc2d.so, the CLI drivers) never links, loads or calls it.
 *
 * PARITY UNPINNED: the reference ships no tests, fixtures or golden vectors
 * (SURVEY.md F2), its SAT.py does not exist (F1), and its CUDA sources cannot
 * be compiled in this image (F7: no nvcc, CUDA headers, Boost or npy.hpp), so
 * this restatement is pinned only by (i) following the reference line by line
 * as cited below, (ii) the analytic known-answer tests in tests/, (iii) an
 * independent numpy restatement (oracle/sat.py) that must agree bit for bit.
 *
 * Each function cites the reference lines it follows (paths relative to
 * /root/reference).  Arithmetic is IEEE binary32 with NO contraction (build
 * with -ffp-contract=off); fmaf appears only where the c2d canonical math
 * functions are defined with a fused multiply-add.
 *
 * Deliberate, documented deviations from the reference:
 *   D1  calcSlack squares the hit count in 64-bit (utils.cu:194 overflows int);
 *   D2  getBin stops at n_bins-1 (utils.cu:201-202 reads one past the end);
 *   RNG cuRAND XORWOW cannot be reproduced without CUDA; the stream is
 *       Philox4x32-10 keyed by (seed, scene, sample group) — see c2d_oracle_draw_words;
 *   sin/cos/log are the c2d polynomial forms instead of CUDA's libdevice.
 */
#define _GNU_SOURCE
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

/* the boundary types and schedule constants (declarations only: the oracle links nothing of the product) */
#include "../include/c2d.h"

/* C2D_ORACLE_FMAD selects how the reference's two-product sums a*x + b*y (utils.cu:139-140, :173-174) are rounded:
 *   0 (canonical, default)  fadd(fmul(a,x), fmul(b,y))       — what the c2d kernels compute;
 *   1                       fma(a, x, fmul(b,y))             — nvcc -fmad=true fusing the LEFT product;
 *   2                       fma(b, y, fmul(a,x))             — nvcc -fmad=true fusing the RIGHT product.
 * nvcc's default is -fmad=true and which product ptxas fuses cannot be observed here (no CUDA), so 1 and 2 exist
 * only to MEASURE how far the canonical choice can be from a CUDA build of the reference (oracle/tools/fmad_study.py,
 * DESIGN.md §2); everything else in this file is identical across the three builds. */
#ifndef C2D_ORACLE_FMAD
#define C2D_ORACLE_FMAD 0
#endif
static inline float dot2(float a, float x, float b, float y)
{
#if C2D_ORACLE_FMAD == 1
    return fmaf(a, x, b * y);
#elif C2D_ORACLE_FMAD == 2
    return fmaf(b, y, a * x);
#else
    return a * x + b * y;
#endif
}
int c2d_oracle_fmad_variant(void) { return C2D_ORACLE_FMAD; }

/* compiler and flags this library was built with (oracle/Makefile passes its flags): bench.py's cpu_baseline quotes them */
#ifndef C2D_ORACLE_CFLAGS
#define C2D_ORACLE_CFLAGS "(flags not recorded: built outside oracle/Makefile)"
#endif
const char* c2d_oracle_build_info(void)
{
#if defined(__clang__)
    return "clang " __VERSION__ " " C2D_ORACLE_CFLAGS;
#elif defined(__GNUC__)
    return "gcc " __VERSION__ " " C2D_ORACLE_CFLAGS;
#else
    return "cc " C2D_ORACLE_CFLAGS;
#endif
}

static inline uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

void c2d_oracle_set_num_threads(int n)
{
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

int c2d_oracle_num_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* ------------------------------------------------------------------------ */
/* Canonical math (coefficients: oracle/tools/fit_poly.py)                   */
/* ------------------------------------------------------------------------ */

/* natural log of a positive normal float */
float c2d_oracle_logf(float u)
{
    static const float L[9] = {-0x1.000000p-1f, 0x1.55552cp-2f, -0x1.ffff28p-3f,
                               0x1.99bffep-3f,  -0x1.55913ep-3f, 0x1.1fd494p-3f,
                               -0x1.f483fap-4f, 0x1.19bbe2p-3f,  -0x1.04cba2p-3f};
    uint32_t t = f2u(u) - 0x3f2aaaabu;
    int32_t e = (int32_t)t >> 23;
    float m = u2f((t & 0x007fffffu) + 0x3f2aaaabu);
    float f = m - 1.0f;
    float q = L[8];
    for (int k = 7; k >= 0; k--) q = fmaf(q, f, L[k]);
    float s = f * f;
    float r = fmaf(s, q, f);
    return fmaf((float)e, 0x1.62e430p-1f, r);
}

static inline void quadrant_rotate(int q, float sn, float cs, float* s_out, float* c_out)
{
    switch (q & 3) {
    case 0: *s_out = sn;  *c_out = cs;  break;
    case 1: *s_out = cs;  *c_out = -sn; break;
    case 2: *s_out = -sn; *c_out = -cs; break;
    default: *s_out = -cs; *c_out = sn; break;
    }
}

/* sin and cos of a finite float angle (radians); stands in for cosf/sinf of
 * utils.cu:133-134 */
void c2d_oracle_sincosf(float x, float* s_out, float* c_out)
{
    float k = rintf(x * 0x1.45f306p-1f);
    float r = fmaf(k, -0x1.920000p+0f, x);
    r = fmaf(k, -0x1.fb4000p-12f, r);
    r = fmaf(k, -0x1.4442d2p-24f, r);
    float z = r * r;
    float sp = 0x1.6dac7ap-19f;
    sp = fmaf(sp, z, -0x1.a01376p-13f);
    sp = fmaf(sp, z, 0x1.11110ep-7f);
    sp = fmaf(sp, z, -0x1.555556p-3f);
    float sn = fmaf(z * r, sp, r);
    float cp = -0x1.2476a8p-22f;
    cp = fmaf(cp, z, 0x1.a012bap-16f);
    cp = fmaf(cp, z, -0x1.6c16bcp-10f);
    cp = fmaf(cp, z, 0x1.555556p-5f);
    cp = fmaf(cp, z, -0x1.000000p-1f);
    float cs = fmaf(z, cp, 1.0f);
    float kc = fminf(fmaxf(k, -1073741824.0f), 1073741824.0f);
    int q = (int)kc;
    quadrant_rotate(q, sn, cs, s_out, c_out);
}

/* sin and cos of the angle 2*pi*y/2^32 */
void c2d_oracle_sincos_u32(uint32_t y, float* s_out, float* c_out)
{
    int q = (int)(y >> 30);
    uint32_t fr = y & 0x3fffffffu;
    int swap = fr > 0x20000000u;
    if (swap) fr = 0x40000000u - fr;
    float x = (float)(int32_t)fr * 0x1p-30f;
    float z = x * x;
    float p = 0x1.4bb0a6p-13f;
    p = fmaf(p, z, -0x1.32ca4ap-8f);
    p = fmaf(p, z, 0x1.466bbap-4f);
    p = fmaf(p, z, -0x1.4abbcep-1f);
    p = fmaf(p, z, 0x1.921fb6p+0f);
    float sn = p * x;
    float c = 0x1.d99986p-11f;
    c = fmaf(c, z, -0x1.55c4e6p-6f);
    c = fmaf(c, z, 0x1.03c1dap-2f);
    c = fmaf(c, z, -0x1.3bd3ccp+0f);
    c = fmaf(c, z, 1.0f);
    float cs = c;
    if (swap) { float t = sn; sn = cs; cs = t; }
    quadrant_rotate(q, sn, cs, s_out, c_out);
}

/* ------------------------------------------------------------------------ */
/* Philox4x32-10 (Salmon et al., SC'11; same constants / word order as       */
/* rocRAND's rocrand_philox4x32_10.h)                                        */
/* ------------------------------------------------------------------------ */
void c2d_oracle_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4])
{
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
    uint32_t k0 = key[0], k1 = key[1];
    for (int round = 0; round < 10; round++) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

/* The eight raw words of item `sample` of stream (seed, scene_id): blocks
 * 2*sample and 2*sample+1 of subsequence scene_id (the scene sampler's layout,
 * c2d_oracle_sample_scenes; also the rocRAND engine pin of tests/test_boundary.py). */
void c2d_oracle_raw8(uint64_t seed, uint64_t scene_id, uint64_t sample, uint32_t raw[8])
{
    uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
    uint64_t blk = 2 * sample;
    uint32_t ctr[4] = {(uint32_t)blk, (uint32_t)(blk >> 32), (uint32_t)scene_id, (uint32_t)(scene_id >> 32)};
    c2d_oracle_philox4x32_10(ctr, key, raw);
    blk += 1;
    ctr[0] = (uint32_t)blk; ctr[1] = (uint32_t)(blk >> 32);
    c2d_oracle_philox4x32_10(ctr, key, raw + 4);
}

/* Draw layout of the Monte-Carlo loop.  The samples of a stream (seed, scene_id)
 * are drawn in groups of four: sample s is member j = s & 3 of group g = s >> 2,
 * which owns Philox blocks 8g .. 8g+5 of subsequence scene_id:
 *   block 8g+0 word j        radius word of the sample's first Box-Muller pair (dx, dy)
 *   block 8g+1 word j        angle word of that pair
 *   block 8g+2 + (j >> 1)    words 2(j&1), 2(j&1)+1: radius, angle word of the second pair (dtheta, dw)
 *   block 8g+4 + (j >> 1)    words 2(j&1), 2(j&1)+1: third pair (dh, second normal unused)
 * (blocks 8g+6, 8g+7 are unused).  The layout is a choice of this build — the
 * reference's XORWOW stream is not reproducible without CUDA anyway — made so
 * that the four radius words of a group come out of ONE Philox block. */
static void draw_group_block(uint64_t seed, uint64_t scene_id, uint64_t group, uint32_t b, uint32_t out[4])
{
    uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
    uint64_t blk = 8 * group + b;
    uint32_t ctr[4] = {(uint32_t)blk, (uint32_t)(blk >> 32), (uint32_t)scene_id, (uint32_t)(scene_id >> 32)};
    c2d_oracle_philox4x32_10(ctr, key, out);
}

/* The six words of one sample in draw order: radius1, angle1, radius2, angle2, radius3, angle3. */
void c2d_oracle_draw_words(uint64_t seed, uint64_t scene_id, uint64_t sample, uint32_t w[6])
{
    uint64_t g = sample >> 2;
    uint32_t j = (uint32_t)(sample & 3), h = j >> 1, o = 2 * (j & 1);
    uint32_t b[4];
    draw_group_block(seed, scene_id, g, 0, b); w[0] = b[j];
    draw_group_block(seed, scene_id, g, 1, b); w[1] = b[j];
    draw_group_block(seed, scene_id, g, 2 + h, b); w[2] = b[o]; w[3] = b[o + 1];
    draw_group_block(seed, scene_id, g, 4 + h, b); w[4] = b[o]; w[5] = b[o + 1];
}

/* The same words for consecutive samples without recomputing a group's blocks
 * (the loops below walk samples in order; one cache per thread). */
typedef struct {
    uint64_t group;
    int valid;          /* bit b: blk[b] holds block 8*group + b */
    uint32_t blk[6][4];
} DrawCache;

static inline const uint32_t* cached_block(DrawCache* c, uint64_t seed, uint64_t scene_id, uint64_t g, uint32_t b)
{
    if (c->group != g) { c->group = g; c->valid = 0; }
    if (!(c->valid & (1 << b))) { draw_group_block(seed, scene_id, g, b, c->blk[b]); c->valid |= 1 << b; }
    return c->blk[b];
}

static void draw_words_cached(DrawCache* c, uint64_t seed, uint64_t scene_id, uint64_t sample, uint32_t w[6])
{
    uint64_t g = sample >> 2;
    uint32_t j = (uint32_t)(sample & 3), h = j >> 1, o = 2 * (j & 1);
    const uint32_t* b;
    b = cached_block(c, seed, scene_id, g, 0); w[0] = b[j];
    b = cached_block(c, seed, scene_id, g, 1); w[1] = b[j];
    b = cached_block(c, seed, scene_id, g, 2 + h); w[2] = b[o]; w[3] = b[o + 1];
    b = cached_block(c, seed, scene_id, g, 4 + h); w[4] = b[o]; w[5] = b[o + 1];
}

/* Box-Muller on two 32-bit words: x -> radius, y -> angle; first normal uses
 * sin, second cos (the word/role order of rocRAND's box_muller). */
void c2d_oracle_box_muller(uint32_t x, uint32_t y, float* n0, float* n1)
{
    float u = fmaf((float)x, 0x1p-32f, 0x1p-33f);
    float rad = sqrtf(-2.0f * c2d_oracle_logf(u));
    float sn, cs;
    c2d_oracle_sincos_u32(y, &sn, &cs);
    *n0 = sn * rad;
    *n1 = cs * rad;
}

/* The five N(0,1) draws of one sample in the reference's order
 * dx, dy, dtheta, dw, dh (utils.cu:146-150). */
void c2d_oracle_normals5(uint64_t seed, uint64_t scene_id, uint64_t sample, float n[5])
{
    uint32_t w[6];
    float unused;
    c2d_oracle_draw_words(seed, scene_id, sample, w);
    c2d_oracle_box_muller(w[0], w[1], &n[0], &n[1]);
    c2d_oracle_box_muller(w[2], w[3], &n[2], &n[3]);
    c2d_oracle_box_muller(w[4], w[5], &n[4], &unused);
}

/* ------------------------------------------------------------------------ */
/* Geometry                                                                  */
/* ------------------------------------------------------------------------ */

/* utils.cu:119-130 */
void c2d_oracle_create_rect(float* r, float w, float h)
{
    r[0] = -w / 2;
    r[1] = -h / 2;
    r[2] = w / 2;
    r[3] = -h / 2;
    r[4] = w / 2;
    r[5] = h / 2;
    r[6] = -w / 2;
    r[7] = h / 2;
}

/* utils.cu:132-142 */
void c2d_oracle_rot_trans_rectangle(float* r, float dx, float dy, float dt)
{
    float c, s;
    c2d_oracle_sincosf(dt, &s, &c);
    for (int i = 0; i < 4; i++) {
        float x = r[2 * i];
        float y = r[2 * i + 1];
        r[2 * i] = dot2(c, x, -s, y) + dx;     /* c*x - s*y + dx: (-s)*y == -(s*y) exactly */
        r[2 * i + 1] = dot2(s, x, c, y) + dy;
    }
}

/* utils.cu:144-157, the five normals passed in */
void c2d_oracle_sample_rectangle(const float* r_in, float* r_out, const StdDev* sd, const float n[5])
{
    float dx = n[0] * sd->x;
    float dy = n[1] * sd->y;
    float dt = n[2] * sd->theta;
    float dw = n[3] * sd->width;
    float dh = n[4] * sd->height;
    float dwh[8];
    memcpy(r_out, r_in, sizeof(float) * 8);
    c2d_oracle_create_rect(dwh, dw, dh);
    for (int i = 0; i < 8; i++) r_out[i] += dwh[i];
    c2d_oracle_rot_trans_rectangle(r_out, dx, dy, dt);
}

static inline void minmax4(const float* p, float* mn, float* mx)
{
    /* thrust::minmax_element over 4 floats (utils.cu:176-177) */
    float lo = p[0], hi = p[0];
    for (int k = 1; k < 4; k++) {
        if (p[k] < lo) lo = p[k];
        if (hi < p[k]) hi = p[k];
    }
    *mn = lo;
    *mx = hi;
}

/* utils.cu:159-184 */
int c2d_oracle_convex_collide(const float* r1, const float* r2)
{
    const float* rs[2] = {r1, r2};
    float norm[2];
    float p1[4];
    float p2[4];
    int collide = 1;
    for (int j = 0; j < 2; j++) {
        const float* r = rs[j];
        for (int i = 0; i < 4; i++) {
            norm[0] = r[(i + 1) * 2 % 8] - r[i * 2];
            norm[1] = r[((i + 1) * 2 + 1) % 8] - r[i * 2 + 1];
            for (int k = 0; k < 4; k++) {
                p1[k] = dot2(norm[0], r1[k * 2], norm[1], r1[k * 2 + 1]);
                p2[k] = dot2(norm[0], r2[k * 2], norm[1], r2[k * 2 + 1]);
            }
            float min1, max1, min2, max2;
            minmax4(p1, &min1, &max1);
            minmax4(p2, &min2, &max2);
            if (max1 < min2 || max2 < min1) collide = 0;
        }
    }
    return collide;
}

/* Convex polygon SAT: utils.cu:159-184 generalised to ka / kb vertices with
 * the true edge normal (-ey, ex) as axis. */
int c2d_oracle_poly_collide(const float* ax, const float* ay, int ka, const float* bx,
                            const float* by, int kb)
{
    int collide = 1;
    for (int j = 0; j < 2; j++) {
        const float* px = j == 0 ? ax : bx;
        const float* py = j == 0 ? ay : by;
        int kp = j == 0 ? ka : kb;
        for (int i = 0; i < kp; i++) {
            int i1 = (i + 1) % kp;
            float ex = px[i1] - px[i];
            float ey = py[i1] - py[i];
            float nx = -ey, ny = ex;
            float min1 = 0, max1 = 0, min2 = 0, max2 = 0;
            for (int k = 0; k < ka; k++) {
                float p = nx * ax[k] + ny * ay[k];
                if (k == 0) { min1 = max1 = p; }
                else { if (p < min1) min1 = p; if (max1 < p) max1 = p; }
            }
            for (int k = 0; k < kb; k++) {
                float p = nx * bx[k] + ny * by[k];
                if (k == 0) { min2 = max2 = p; }
                else { if (p < min2) min2 = p; if (max2 < p) max2 = p; }
            }
            if (max1 < min2 || max2 < min1) collide = 0;
        }
    }
    return collide;
}

/* ------------------------------------------------------------------------ */
/* Batched entry points (host pointers, same layouts as include/c2d.h)       */
/* ------------------------------------------------------------------------ */

int c2d_oracle_rects_from_poses(const float* cx, const float* cy, const float* w, const float* h,
                                const float* theta, size_t n, float* const out_planes[8])
{
#pragma omp parallel for schedule(static)
    for (long long i = 0; i < (long long)n; i++) {
        float r[8];
        c2d_oracle_create_rect(r, w[i], h[i]);
        c2d_oracle_rot_trans_rectangle(r, cx[i], cy[i], theta[i]);
        for (int k = 0; k < 8; k++) out_planes[k][i] = r[k];
    }
    return 0;
}

/* returns the number of colliding pairs */
unsigned long long c2d_oracle_sat_rect_pairs_verts(const float* const planes[16], size_t n, uint8_t* out)
{
    unsigned long long count = 0;
#pragma omp parallel for schedule(static) reduction(+ : count)
    for (long long i = 0; i < (long long)n; i++) {
        float r1[8], r2[8];
        for (int k = 0; k < 8; k++) {
            r1[k] = planes[k][i];
            r2[k] = planes[8 + k][i];
        }
        int c = c2d_oracle_convex_collide(r1, r2);
        out[i] = (uint8_t)c;
        count += (unsigned)c;
    }
    return count;
}

unsigned long long c2d_oracle_sat_rect_pairs_pose(const float* const pp[10], size_t n, uint8_t* out)
{
    unsigned long long count = 0;
#pragma omp parallel for schedule(static) reduction(+ : count)
    for (long long i = 0; i < (long long)n; i++) {
        float r1[8], r2[8];
        c2d_oracle_create_rect(r1, pp[2][i], pp[3][i]);
        c2d_oracle_rot_trans_rectangle(r1, pp[0][i], pp[1][i], pp[4][i]);
        c2d_oracle_create_rect(r2, pp[7][i], pp[8][i]);
        c2d_oracle_rot_trans_rectangle(r2, pp[5][i], pp[6][i], pp[9][i]);
        int c = c2d_oracle_convex_collide(r1, r2);
        out[i] = (uint8_t)c;
        count += (unsigned)c;
    }
    return count;
}

/* vx, vy: f32[2][rows][n] (rows vertex rows per polygon, 1..KMAX); k: u8[2][n].  Returns the colliding count, or
 * (unsigned long long)-1 on a vertex count outside 1..rows (those pairs read 0). */
unsigned long long c2d_oracle_sat_poly_pairs_rows(const float* vx, const float* vy, const uint8_t* k,
                                                  size_t n, int rows, uint8_t* out)
{
    unsigned long long count = 0;
    int bad = 0;
    if (rows < 1 || rows > C2D_POLY_KMAX) return ~0ull;
#pragma omp parallel for schedule(static) reduction(+ : count) reduction(| : bad)
    for (long long i = 0; i < (long long)n; i++) {
        float ax[C2D_POLY_KMAX], ay[C2D_POLY_KMAX], bx[C2D_POLY_KMAX], by[C2D_POLY_KMAX];
        int ka = k[i], kb = k[n + i];
        if (ka < 1 || ka > rows || kb < 1 || kb > rows) { bad = 1; out[i] = 0; continue; }
        for (int v = 0; v < ka; v++) {
            ax[v] = vx[(size_t)v * n + i];
            ay[v] = vy[(size_t)v * n + i];
        }
        for (int v = 0; v < kb; v++) {
            bx[v] = vx[((size_t)rows + v) * n + i];
            by[v] = vy[((size_t)rows + v) * n + i];
        }
        int c = c2d_oracle_poly_collide(ax, ay, ka, bx, by, kb);
        out[i] = (uint8_t)c;
        count += (unsigned)c;
    }
    return bad ? ~0ull : count;
}

unsigned long long c2d_oracle_sat_poly_pairs(const float* vx, const float* vy, const uint8_t* k,
                                             size_t n, uint8_t* out)
{
    return c2d_oracle_sat_poly_pairs_rows(vx, vy, k, n, C2D_POLY_KMAX, out);
}

/* ------------------------------------------------------------------------ */
/* Adaptive-stopping statistics                                              */
/* ------------------------------------------------------------------------ */

/* utils.cu:186-196 with D1 fixed (k*k in 64-bit).  Expression types follow
 * the reference: the zero/all branch is evaluated in double
 * (log(1.0 / alpha) / nsamples, alpha a float), the other branch in float. */
float c2d_oracle_calc_slack(uint32_t nsamples, uint32_t nsamples_true)
{
    float z = 1.96;
    float alpha = 0.025;
    if ((nsamples_true == nsamples) || (nsamples_true == 0)) {
        /* log(1.0 / (double)0.025f), folded so that no libm log is involved */
        (void)alpha;
        return (float)(0x1.d82d33932720dp+1 / (double)nsamples);
    } else {
        float k = (float)nsamples_true;
        float kk = (float)((uint64_t)nsamples_true * (uint64_t)nsamples_true);
        return z / (float)nsamples * sqrtf(k - kk / (float)nsamples);
    }
}

/* utils.cu:198-207 with D2 fixed; "last matching bin wins" kept */
int c2d_oracle_get_bin(float p, const float* accuracy_bins, uint32_t n_accuracy_bins)
{
    int bin = 0;
    for (uint32_t i = 0; i + 1 < n_accuracy_bins; i++) {
        if (p >= accuracy_bins[i] && p <= accuracy_bins[i + 1]) bin = (int)i;
    }
    return bin;
}

/* ------------------------------------------------------------------------ */
/* Monte-Carlo                                                               */
/* ------------------------------------------------------------------------ */

/* Scene set-up: compute_collision_probability.cu:127-133 */
static void scene_setup(float robot_w, float robot_h, const Position* pos, const Pose* pose,
                        float robot[8], float obstacle[8])
{
    c2d_oracle_create_rect(obstacle, pose->width, pose->height);
    c2d_oracle_create_rect(robot, robot_w, robot_h);
    c2d_oracle_rot_trans_rectangle(robot, pos->x, pos->y, pose->theta);
}

/* One sample: compute_collision_probability.cu:137-138 (all five normals are
 * drawn, as in utils.cu:146-150, whatever the standard deviations are). */
static inline int scene_sample(const float robot[8], const float obstacle[8], const StdDev* sd,
                               uint64_t seed, uint64_t scene_id, uint64_t sample, DrawCache* cache)
{
    float n[5];
    float sampled[8];
    uint32_t w[6];
    float unused;
    draw_words_cached(cache, seed, scene_id, sample, w);
    c2d_oracle_box_muller(w[0], w[1], &n[0], &n[1]);
    c2d_oracle_box_muller(w[2], w[3], &n[2], &n[3]);
    c2d_oracle_box_muller(w[4], w[5], &n[4], &unused);
    c2d_oracle_sample_rectangle(obstacle, sampled, sd, n);
    return c2d_oracle_convex_collide(robot, sampled);
}

/* hits among samples [sample_begin, sample_begin + n_samples) of one scene */
unsigned long long c2d_oracle_mc_pair(float robot_w, float robot_h, const Position* pos,
                                      const Pose* pose, const StdDev* sd, uint64_t seed,
                                      uint64_t scene_id, uint64_t sample_begin, uint64_t n_samples)
{
    float robot[8], obstacle[8];
    scene_setup(robot_w, robot_h, pos, pose, robot, obstacle);
    unsigned long long hits = 0;
#pragma omp parallel reduction(+ : hits)
    {
        DrawCache cache = {~0ull, 0, {{0}}};
#pragma omp for schedule(static)
        for (long long i = 0; i < (long long)n_samples; i++)
            hits += (unsigned)scene_sample(robot, obstacle, sd, seed, scene_id, sample_begin + (uint64_t)i, &cache);
    }
    return hits;
}

/* The sampled rectangle of one sample (debug / geometry parity). */
void c2d_oracle_mc_sampled_rect(const Pose* pose, const StdDev* sd, uint64_t seed,
                                uint64_t scene_id, uint64_t sample, float out[8])
{
    float obstacle[8], n[5];
    c2d_oracle_create_rect(obstacle, pose->width, pose->height);
    c2d_oracle_normals5(seed, scene_id, sample, n);
    c2d_oracle_sample_rectangle(obstacle, out, sd, n);
}

/* Adaptive loop for many scenes: the per-scene view of
 * compute_collision_probability.cu:276-332.  A scene is checked after every
 * batch of the fixed schedule and stops at the first check that passes or at
 * n_samples >= max_samples.  rows (optional) receives (x,y,cp,var_idx,pose_idx)
 * with cp = hits / n_used in float (utils.cu:214). Returns total samples. */
unsigned long long c2d_oracle_mc_scenes(const Pose* poses, uint32_t num_poses, const StdDev* std_devs,
                                        uint32_t num_std_devs,
                                        const PositionWithVarAndPoseIdx* scenes, size_t n_scenes,
                                        float robot_w, float robot_h, const float* accuracy_bins,
                                        const float* bin_accuracy, uint32_t n_accuracy_bins,
                                        uint32_t max_samples, uint64_t seed, uint64_t scene_id_base,
                                        uint32_t small_batch, uint32_t large_batch, uint32_t switch_at,
                                        uint32_t* hits_out, uint32_t* n_used_out,
                                        PoseCPVarAndPoseIdx* rows)
{
    if (!small_batch && !large_batch && !switch_at) { /* reference default schedule */
        small_batch = C2D_MC_SMALL_BATCH; large_batch = C2D_MC_LARGE_BATCH; switch_at = C2D_MC_SWITCH_AT;
    }
    unsigned long long total = 0;
    (void)num_poses;
    (void)num_std_devs;
#pragma omp parallel for schedule(dynamic, 1) reduction(+ : total)
    for (long long g = 0; g < (long long)n_scenes; g++) {
        Position pos = {scenes[g].x, scenes[g].y};
        int pose_idx = (int)scenes[g].pose_idx; /* ccp.cu:121 float -> int */
        int sd_idx = (int)scenes[g].var_idx;    /* ccp.cu:122 */
        Pose pose = poses[pose_idx];
        StdDev sd = std_devs[sd_idx];
        float robot[8], obstacle[8];
        scene_setup(robot_w, robot_h, &pos, &pose, robot, obstacle);
        uint64_t sid = scene_id_base + (uint64_t)g;
        uint32_t n = 0, k = 0;
        DrawCache cache = {~0ull, 0, {{0}}};
        while (n < max_samples) { /* ccp.cu:281 (num_left > 0 is this scene not being done) */
            uint32_t nb = n < switch_at ? small_batch : large_batch; /* ccp.cu:283-286; ztest.cu:332 constant */
            for (uint32_t i = 0; i < nb; i++) k += (uint32_t)scene_sample(robot, obstacle, &sd, seed, sid, (uint64_t)n + i, &cache);
            n += nb;
            float slack = c2d_oracle_calc_slack(n, k);               /* ccp.cu:140 */
            float p = (float)k / (float)n;                          /* ccp.cu:142 */
            if (slack <= bin_accuracy[c2d_oracle_get_bin(p, accuracy_bins, n_accuracy_bins)]) break; /* :144 */
        }
        hits_out[g] = k;
        n_used_out[g] = n;
        if (rows) {
            rows[g].x = pos.x;
            rows[g].y = pos.y;
            rows[g].cp = (float)k / (float)n; /* utils.cu:214 */
            rows[g].var_idx = scenes[g].var_idx;
            rows[g].pose_idx = scenes[g].pose_idx;
        }
        total += n;
    }
    return total;
}

/* ------------------------------------------------------------------------ */
/* Monte-Carlo over convex polygons                                          */
/* ------------------------------------------------------------------------ */
/* README.md:3 of the reference: "can easily be extended to handle arbitrary convex 2D shapes".  The reference itself
 * stops at rectangles (sample_rectangle utils.cu:144-157, convex_collide utils.cu:159-184); what follows is that
 * extension, stated once here and implemented independently by the kernels (csrc/c2d_mc_poly.hip):
 *   - the robot is a polygon in its own frame, rotated by theta and moved to pos with the arithmetic of
 *     rot_trans_rectangle (utils.cu:132-142; ccp.cu:132-133);
 *   - the obstacle is a polygon about the origin (ccp.cu:128); a sample draws the reference's five normals in the
 *     reference's order (utils.cu:146-150) and applies them as sample_rectangle does: dw, dh change the SHAPE first —
 *     for a rectangle the half extents grow by dw/2, dh/2 (utils.cu:152-155); for a polygon the obstacle frame's x / y
 *     coordinates are scaled by (1 + dw), (1 + dh), i.e. StdDev.width / .height are RELATIVE here (a w x h rectangle
 *     given as a 4-gon with sigma_w / w, sigma_h / h has the distribution of the reference's sample) — then the shape
 *     is rotated by dtheta about the origin and moved by (dx, dy) (utils.cu:156);
 *   - the collision test is the projection / strict-< interval test of utils.cu:172-180 on the TRUE normals of all
 *     ka + kb edges (c2d_oracle_poly_collide above; the edge-as-axis shortcut of utils.cu:170-171 is only valid for
 *     rectangles).
 * With sigma_w = sigma_h = 0 the scale factors are exactly 1 and a rectangle given as a 4-gon gets the very vertices
 * c2d_oracle_mc_pair gives it; the two tests then differ only in the scale of their axes (edge vector against normal). */
/* robot placement: rot_trans_rectangle (utils.cu:132-142) applied to every vertex of a polygon */
void c2d_oracle_place_polygon(const c2d_polygon* in, float dx, float dy, float dt, float* ox, float* oy)
{
    float c, s;
    c2d_oracle_sincosf(dt, &s, &c);
    for (uint32_t k = 0; k < in->k; k++) {
        float x = in->x[k], y = in->y[k];
        ox[k] = dot2(c, x, -s, y) + dx;
        oy[k] = dot2(s, x, c, y) + dy;
    }
}

/* utils.cu:144-157 for a polygon, the five normals passed in */
void c2d_oracle_sample_polygon(const c2d_polygon* in, const StdDev* sd, const float n[5], float* ox, float* oy)
{
    float dx = n[0] * sd->x;
    float dy = n[1] * sd->y;
    float dt = n[2] * sd->theta;
    float dw = n[3] * sd->width;
    float dh = n[4] * sd->height;
    float fx = 1.0f + dw, fy = 1.0f + dh; /* utils.cu:152-155: the shape changes before it is rotated and moved */
    float c, s;
    c2d_oracle_sincosf(dt, &s, &c);
    for (uint32_t k = 0; k < in->k; k++) {
        float x = fx * in->x[k], y = fy * in->y[k];
        ox[k] = dot2(c, x, -s, y) + dx; /* utils.cu:139 */
        oy[k] = dot2(s, x, c, y) + dy;  /* utils.cu:140 */
    }
}

static inline int poly_scene_sample(const float* rx, const float* ry, int ka, const c2d_polygon* obstacle, const StdDev* sd,
                                    uint64_t seed, uint64_t scene_id, uint64_t sample, DrawCache* cache)
{
    float n[5], unused, ox[C2D_POLY_KMAX], oy[C2D_POLY_KMAX];
    uint32_t w[6];
    draw_words_cached(cache, seed, scene_id, sample, w);
    c2d_oracle_box_muller(w[0], w[1], &n[0], &n[1]);
    c2d_oracle_box_muller(w[2], w[3], &n[2], &n[3]);
    c2d_oracle_box_muller(w[4], w[5], &n[4], &unused);
    c2d_oracle_sample_polygon(obstacle, sd, n, ox, oy);
    return c2d_oracle_poly_collide(rx, ry, ka, ox, oy, (int)obstacle->k); /* ccp.cu:138 */
}

static int polygon_ok(const c2d_polygon* p) { return p && p->k >= 1 && p->k <= C2D_POLY_KMAX; }

/* hits among samples [sample_begin, sample_begin + n_samples) of one polygon scene (ccp.cu:119-139); ~0 on a bad count */
unsigned long long c2d_oracle_mc_poly_pair(const c2d_polygon* robot, const Position* pos, float robot_theta,
                                           const c2d_polygon* obstacle, const StdDev* sd, uint64_t seed,
                                           uint64_t scene_id, uint64_t sample_begin, uint64_t n_samples)
{
    if (!polygon_ok(robot) || !polygon_ok(obstacle)) return ~0ull;
    float rx[C2D_POLY_KMAX], ry[C2D_POLY_KMAX];
    c2d_oracle_place_polygon(robot, pos->x, pos->y, robot_theta, rx, ry); /* ccp.cu:132-133 */
    unsigned long long hits = 0;
#pragma omp parallel reduction(+ : hits)
    {
        DrawCache cache = {~0ull, 0, {{0}}};
#pragma omp for schedule(static)
        for (long long i = 0; i < (long long)n_samples; i++)
            hits += (unsigned)poly_scene_sample(rx, ry, (int)robot->k, obstacle, sd, seed, scene_id, sample_begin + (uint64_t)i, &cache);
    }
    return hits;
}

/* the sampled obstacle of one sample (debug / geometry parity) */
void c2d_oracle_mc_poly_sampled(const c2d_polygon* obstacle, const StdDev* sd, uint64_t seed, uint64_t scene_id,
                                uint64_t sample, float* ox, float* oy)
{
    float n[5];
    c2d_oracle_normals5(seed, scene_id, sample, n);
    c2d_oracle_sample_polygon(obstacle, sd, n, ox, oy);
}

/* The adaptive loop of c2d_oracle_mc_scenes for polygon scenes: the table holds (robot rotation, obstacle shape) where
 * the reference's holds Pose {width, height, theta}; rows, schedule and stop rule are the same (ccp.cu:276-332).
 * Vertex counts outside 1..C2D_POLY_KMAX are clamped (the kernels do the same and report the error). */
unsigned long long c2d_oracle_mc_poly_scenes(const c2d_polygon* robot, const c2d_poly_pose* poses, uint32_t num_poses,
                                             const StdDev* std_devs, uint32_t num_std_devs,
                                             const PositionWithVarAndPoseIdx* scenes, size_t n_scenes,
                                             const float* accuracy_bins, const float* bin_accuracy,
                                             uint32_t n_accuracy_bins, uint32_t max_samples, uint64_t seed,
                                             uint64_t scene_id_base, uint32_t small_batch, uint32_t large_batch,
                                             uint32_t switch_at, uint32_t* hits_out, uint32_t* n_used_out,
                                             PoseCPVarAndPoseIdx* rows)
{
    if (!small_batch && !large_batch && !switch_at) {
        small_batch = C2D_MC_SMALL_BATCH; large_batch = C2D_MC_LARGE_BATCH; switch_at = C2D_MC_SWITCH_AT;
    }
    unsigned long long total = 0;
    (void)num_poses;
    (void)num_std_devs;
    c2d_polygon rob = *robot;
    rob.k = rob.k < 1 ? 1 : (rob.k > C2D_POLY_KMAX ? C2D_POLY_KMAX : rob.k);
#pragma omp parallel for schedule(dynamic, 1) reduction(+ : total)
    for (long long g = 0; g < (long long)n_scenes; g++) {
        int pose_idx = (int)scenes[g].pose_idx; /* ccp.cu:121 */
        int sd_idx = (int)scenes[g].var_idx;    /* ccp.cu:122 */
        c2d_poly_pose pp = poses[pose_idx];
        pp.obstacle.k = pp.obstacle.k < 1 ? 1 : (pp.obstacle.k > C2D_POLY_KMAX ? C2D_POLY_KMAX : pp.obstacle.k);
        StdDev sd = std_devs[sd_idx];
        float rx[C2D_POLY_KMAX], ry[C2D_POLY_KMAX];
        c2d_oracle_place_polygon(&rob, scenes[g].x, scenes[g].y, pp.theta, rx, ry);
        uint64_t sid = scene_id_base + (uint64_t)g;
        uint32_t n = 0, k = 0;
        DrawCache cache = {~0ull, 0, {{0}}};
        while (n < max_samples) {
            uint32_t nb = n < switch_at ? small_batch : large_batch;
            for (uint32_t i = 0; i < nb; i++)
                k += (uint32_t)poly_scene_sample(rx, ry, (int)rob.k, &pp.obstacle, &sd, seed, sid, (uint64_t)n + i, &cache);
            n += nb;
            float slack = c2d_oracle_calc_slack(n, k);
            float p = (float)k / (float)n;
            if (slack <= bin_accuracy[c2d_oracle_get_bin(p, accuracy_bins, n_accuracy_bins)]) break;
        }
        hits_out[g] = k;
        n_used_out[g] = n;
        if (rows) {
            rows[g].x = scenes[g].x;
            rows[g].y = scenes[g].y;
            rows[g].cp = (float)k / (float)n;
            rows[g].var_idx = scenes[g].var_idx;
            rows[g].pose_idx = scenes[g].pose_idx;
        }
        total += n;
    }
    return total;
}

/* Scene sampler: generate_dataset.cu:207-219.  Stream (seed ^ SCENE_DOMAIN,
 * scene id), sample 0: raw[0] -> pose_idx, raw[1] -> var_idx, raw[2] -> theta
 * (uniform (0,1] * 2 * M_PI, evaluated in double as in the reference),
 * raw[4], raw[5] -> one normal for the shift. */
#define C2D_SCENE_DOMAIN 0x5ce9e5a3c0117de5ull
void c2d_oracle_sample_scenes(const Pose* poses, uint32_t num_poses, const StdDev* std_devs,
                              uint32_t num_std_devs, float robot_w, float robot_h, float spread,
                              uint64_t seed, uint64_t scene_id_base, size_t n_scenes,
                              PositionWithVarAndPoseIdx* scenes)
{
    float r_offset = (robot_w + robot_h) / 4; /* generate_dataset.cu:398 */
    for (size_t g = 0; g < n_scenes; g++) {
        uint32_t raw[8];
        c2d_oracle_raw8(seed ^ C2D_SCENE_DOMAIN, scene_id_base + g, 0, raw);
        uint32_t pose_idx = raw[0] % num_poses;      /* :208 */
        uint32_t sd_idx = raw[1] % num_std_devs;     /* :209 */
        Pose pose = poses[pose_idx];
        StdDev sd = std_devs[sd_idx];
        float u = fmaf((float)raw[2], 0x1p-32f, 0x1p-33f);      /* curand_uniform: (0,1] */
        float theta = (float)((double)u * 2 * M_PI);             /* :213 float*int*double -> double -> float */
        float nrm, unused;
        c2d_oracle_box_muller(raw[4], raw[5], &nrm, &unused);
        float shift = nrm * ((sd.y + sd.x) / 2) * spread;       /* :214 */
        float ct, st;
        c2d_oracle_sincosf(theta, &st, &ct);
        /* :215-216 — 2.35 is a double literal: the bracket evaluates in double */
        double bx = ((double)(pose.width / 2 + r_offset) + 2.35 + (double)sd.x) + (double)shift;
        double by = ((double)(pose.height / 2 + r_offset) + 2.35 + (double)sd.y) + (double)shift;
        scenes[g].x = (float)((double)ct * bx);
        scenes[g].y = (float)((double)st * by);
        scenes[g].var_idx = (float)sd_idx;
        scenes[g].pose_idx = (float)pose_idx;
    }
}

/* ------------------------------------------------------------------------ */
/* Array forms of the canonical math (parity checks against c2d_math_eval)   */
/* ------------------------------------------------------------------------ */
void c2d_oracle_math_eval(int fn, const uint32_t* in_bits, size_t n, float* out0, float* out1)
{
#pragma omp parallel for schedule(static)
    for (long long i = 0; i < (long long)n; i++) {
        uint32_t bits = in_bits[i];
        float x = u2f(bits), a = 0.0f, b = 0.0f;
        switch (fn) {
        case 0: a = c2d_oracle_logf(x); break;
        case 1: c2d_oracle_sincosf(x, &a, &b); break;
        case 2: c2d_oracle_sincos_u32(bits, &a, &b); break;
        case 3: a = sqrtf(x); break;
        case 4: c2d_oracle_box_muller(bits, ~bits * 2654435761u, &a, &b); break;
        default: break;
        }
        out0[i] = a;
        if (out1) out1[i] = b;
    }
}
