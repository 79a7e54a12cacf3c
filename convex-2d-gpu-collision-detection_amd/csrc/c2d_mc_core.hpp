// c2d_mc_core.hpp — the shape-independent part of the Monte-Carlo kernels (gfx950): how one wave counts the hits among a range
// of samples of one scene.  Included by c2d_mc.hip (rectangles: the reference's own case) and c2d_mc_poly.hip (convex
// polygons).  What depends on the shape comes in through a policy class P:
//
//   P::Scene   wave-uniform description of the scene; the code here reads sx, sy (standard deviations of the centre), x0 /
//              use_x0 (radius-word form of the certain-miss pretest) and tame()
//   P::Queue   the wave's LDS block; it has a member `sq` (SampleQueues, below) and whatever the policy parks beside it
//   P::centre_pretest(sc, wq, dx, dy, miss_m)  true: the obstacle centre alone proves that the sample cannot collide
//   P::evaluate(sc, wq, w2r, w2a, dx, dy, seed, scene_id, sample, live_m)   hits among the (up to 64) samples of a pass whose
//              centres are (dx, dy): the rest of the sample (second / third Box-Muller pair, rotation, shape) and the test
//   P::plain(sc, seed, scene_id, begin, count, wq)   hits of a scene that is not tame: every sample evaluated in full, for
//              every bit pattern
//   P::finish(sc, wq)   hits among whatever P::evaluate has held back (a policy may evaluate in stages and keep the survivors of
//              the first stage in a queue of its own until 64 have gathered); called once at the end of a sample range
//
// The draw layout (groups of four samples, c2d_math.hpp), the two ways to run a scene (NEAR / FAR), the queues and the
// compaction are the same for every shape; DESIGN.md §5 has the measurements behind them.
#pragma once

#include "c2d_internal.hpp"
#include "c2d_math.hpp"

#ifndef C2D_MC_STAT
#define C2D_MC_STAT(i, v) do { } while (0)
#endif

// ---- clock build (-DC2D_MC_CLOCK, `make lib-mcclock`; never the product): which clock do the Monte-Carlo kernels HOLD?  Every
// wave reads the shader-cycle counter (s_memtime) and the constant 100 MHz counter (s_memrealtime) when its sample work starts
// and when it ends, and adds both differences to two device words; their ratio x 100 MHz is the time-weighted clock the waves
// ran at.  The VALU rooflines of bench.py are priced at that clock beside the nominal 2.4 GHz (tests/tools/mc_clock.py).
#ifdef C2D_MC_CLOCK
#define C2D_MC_CLOCK_WORDS(name) __device__ unsigned long long name[4]   /* shader cycles, 100 MHz ticks, waves, unused */
struct McClockStamp {
    unsigned long long c0, r0;
    __device__ __forceinline__ McClockStamp() : c0(__builtin_amdgcn_s_memtime()), r0(__builtin_amdgcn_s_memrealtime()) {}
    __device__ __forceinline__ void stop(unsigned long long* words) const
    {
        const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
        if ((threadIdx.x & 63) == 0) {
            atomicAdd(&words[0], c1 - c0);
            atomicAdd(&words[1], r1 - r0);
            atomicAdd(&words[2], 1ull);
        }
    }
};
#define C2D_MC_CLOCK_START() const McClockStamp clock_stamp__
#define C2D_MC_CLOCK_STOP(words) clock_stamp__.stop(words)
#else
#define C2D_MC_CLOCK_START() do { } while (0)
#define C2D_MC_CLOCK_STOP(words) do { } while (0)
#endif

namespace c2d {

// One wave per block: a multi-wave block keeps its LDS and its place until the slowest of its waves is done, and the
// work items differ by 4x in cost (config-4 shard 765 -> 751 ms, reference-default batch 58.8 -> 57.2 ms against 256 threads).
constexpr int kMcBlock = 64;
constexpr int kWavesPerBlock = kMcBlock / 64;

// |N(0,1) draw| of box_muller: the radius is at most sqrt(-2 ln 2^-33) = 6.7638 and
// |sin|, |cos| <= 1 + 2^-22.
constexpr float kNormalMax = 6.77f;


// ---- the certain-miss pretest from the Box-Muller radius alone: the raw radius word x with x >= x0 proves a miss.
// R0 > 0 is a radius below which the scene's centre pretest (T < plo or T > phi on some robot axis) is certain to fire
// whatever the angle (the scene builders derive it: R0 = max_i L_i / (G_i (1 + 2^-10))).  rad = sqrt(-2 log u) falls with
// u = (x + 1/2) 2^-32, so "rad < R0" is "x >= x0": one integer compare decides the sample before any transcendental is
// evaluated.  x0 is rounded UP generously (a larger x0 only dismisses fewer samples):
//     u0 = exp(-(R0 (1 - 2^-10))^2 / 2) (1 + 2^-10),      x0 = floor(u0 2^32 + 2) + 1.
// The exponential is the hardware's (__expf = v_exp_f32(x log2 e)), whose worst-case error must fit that slack:
//   * v_exp_f32 itself is within 1 ulp = 2^-23 relative;
//   * its argument t = x log2 e carries one rounding of the product and the rounding of the constant, 2 x 2^-24 |t| absolute,
//     and x = -r^2 / 2 two roundings more, 2^-23 |x| relative to x: together |t| (2^-23 + 2^-23) ln 2 relative in 2^t;
//   * x0 is only used when it is below 2^32 - 256 and the result is not flushed, i.e. |t| < 126 (for larger |t| the
//     hardware returns 0 or a denormal-flushed 0, x0 becomes 3: nothing to prove);
//   hence the relative error of u0's exponential is below 126 x 2^-22 x 0.6932 + 2^-23 < 2.1e-5 = 2^-15.5.
// The slack is (1 + 2^-10) on u0 and (1 - 2^-10) on R0, the latter worth another factor exp(R0^2 2^-10) >= 1: at least
// 9.7e-4 = 2^-10 relative, 46 times the bound, plus 2 absolute words.  (tests/test_oracle.py checks the bound on a dense sweep
// of R0 against a double-precision exponential.)
C2D_DEV void radius_word_threshold(float R0, uint32_t& x0, bool& use_x0)
{
    if (R0 > 0.25f && R0 < 1e30f) {  // below 0.25 fewer than 3 % of the draws would qualify anyway
        const float r = R0 * (1.0f - 0x1p-10f);
        const float u0 = __expf(-0.5f * r * r) * (1.0f + 0x1p-10f);
        const float xf = u0 * 4294967296.0f + 2.0f;
        if (xf < 4294967040.0f) {
            x0 = (uint32_t)xf + 1u;
            use_x0 = true;
        }
    }
}

// The sampled obstacle of one sample (reference utils.cu:144-157), in two steps: the first Box-Muller pair gives the
// centre (dx, dy); the rest (dtheta, dw, dh, rotation, vertices) is only needed when the centre pretest cannot rule the
// sample out.  The words come from the sample's group (draw layout: c2d_math.hpp, philox_draw_block).  The third pair
// only feeds dh; its block is skipped when sigma_h == 0 because dh = n*0 cannot change any vertex (the product is +-0
// and is only ever added).
template <class S>
C2D_DEV void sample_centre(const S& sc, uint32_t radius_word, uint32_t angle_word, float& dx, float& dy)
{
    float n0, n1;
    box_muller(radius_word, angle_word, n0, n1);
    dx = n0 * sc.sx;
    dy = n1 * sc.sy;
}

// The votes are kept as 64-bit lane masks built from ballots of the bare comparisons: a ballot of a comparison IS the
// v_cmp's result, while a ballot of a boolean expression (ballot(!sep), ballot(hit && live)) makes the compiler rebuild
// the boolean per lane first (v_cndmask 0/1 + v_cmp_ne), two VALU instructions per vote, four votes per evaluated sample.
C2D_DEV unsigned long long wave_lanes() { return __builtin_amdgcn_ballot_w64(true); }  // the active lanes (exec)

#ifndef C2D_MC_ILP
#define C2D_MC_ILP 1  // 1, 2, 3 blocks side by side: 425 / 422 / 453 ms on the config-4 shard; 1 keeps the LDS at 4.5 KB per wave
#endif
#ifndef C2D_MC_FAR_X0
#define C2D_MC_FAR_X0 0x80000000u  // half of the radius words are candidates (config-4 shard: 446 / 422 / 401 ms for 2^29 / 2^30 / 2^31)
#endif
[[maybe_unused]] constexpr uint32_t kFarX0 = C2D_MC_FAR_X0;
[[maybe_unused]] constexpr int kIlp = C2D_MC_ILP;  // radius blocks computed side by side on the far-scene path
constexpr int kQueueSlots = 128;      // < 64 left over + at most 64 pushed per step
constexpr int kCandSlots = 64 + 256 * C2D_MC_ILP;  // < 64 left over + every sample of the iterations fetched together
constexpr int kNearSlots = 64 + 256;  // < 64 left over + the four members of 64 groups

// Per-wave LDS queues.  Two ways to run a scene, chosen per scene (wave-uniform):
//  * NEAR (the radius word proves little or nothing): a lane owns a group of four samples and produces them back to back from
//    the group's two Philox blocks; samples the centre pretest cannot rule out wait in `near.c/idx` and are evaluated 64 at a time.
//  * FAR (the radius word alone proves at least every other sample to be a miss): the four radius words of a group
//    cost one Philox block, and everything after that works on COMPACTED samples, 64 busy lanes at a time:
//    candidates (radius word, offset) wait in `cand` for their angle word / Box-Muller / centre pretest, the undecided
//    ones among them in `und` for the full evaluation.
// In both, an undecided sample is a centre and an offset (12 B); the block of its second Box-Muller pair is drawn by the lane that
// evaluates it (evaluate_queued).
union SampleQueues {
    struct {
        float2 c[kNearSlots];        // dx, dy
        uint32_t idx[kNearSlots];    // sample offset within the chunk
    } near;
    struct {
        uint2 cand[kCandSlots];      // radius word, sample offset within the chunk
        float2 und_c[kQueueSlots];   // dx, dy
        uint32_t und_idx[kQueueSlots];
    } far;
};

C2D_DEV void wave_lds_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// ---- evaluation of queued undecided samples (centre, offset): the last `take` (<= 64) of `n` entries.  Each lane draws the
// block of ITS sample's second Box-Muller pair (block 2 or 3 of the sample's group).  Lanes read slots that other lanes of this wave
// wrote: LDS operations of a wave complete in order, the fences only stop the compiler from moving the reads above the writes
// and later writes above the reads (no instruction is emitted).
template <class P>
C2D_DEV uint32_t evaluate_queued(const typename P::Scene& sc, typename P::Queue& wq, const float2* qc, const uint32_t* qidx, uint32_t& n, uint32_t take,
                                 uint64_t seed, uint64_t scene_id, uint64_t begin)
{
    const uint32_t lane = threadIdx.x & 63;
    wave_lds_sync();
    const unsigned long long live_m = take >= 64 ? ~0ull : (1ull << take) - 1;
    const uint32_t src = n - take + (lane < take ? lane : 0);
    const float2 c = qc[src];
    const uint32_t sidx = qidx[src];
    wave_lds_sync();
    n -= take;
    const uint64_t s = begin + sidx;
    const U4 pb = philox_draw_block(seed, scene_id, s >> 2, 2u + ((uint32_t)(s >> 1) & 1u));
    const bool odd = (s & 1) != 0;
    return P::evaluate(sc, wq, odd ? pb.z : pb.x, odd ? pb.w : pb.y, c.x, c.y, seed, scene_id, s, live_m);
}

// ---- NEAR: hits among samples [begin, begin + count) of one scene, computed by one wave.  A lane owns one GROUP of four samples
// per iteration (draw layout: c2d_math.hpp), so an iteration covers 256 consecutive samples.  begin and count are arbitrary (a
// shard may start inside a group): positions outside [begin, begin + count) are masked, never drawn into the result.
// The lane produces its four members back to back in straight-line code — the group's two blocks stay in registers for exactly
// that long — and only the samples the centre pretest leaves undecided are queued (centre, offset); whenever 64 wait they are
// evaluated, each drawing the block of its own second pair.  The form of round 2 (members one per pass of a rolled loop, blocks
// in a lane-private LDS stash, the second pair's block drawn wave-wide for a pair of members as soon as ONE lane was undecided)
// paid that block for nearly every pair on a sparse scene; this one is 1.35x faster on scenes whose samples are mostly ruled
// out by their centre — the long tail of a dataset batch — and 5 % faster on the config-3 scene (four independent Box-Muller
// chains per lane for the scheduler, no stash traffic) although that scene now draws 0.65 instead of 0.44 second-pair blocks
// per sample; a scene in which EVERY sample collides is 16 % slower (one block per sample instead of one per two).  DESIGN.md §5.
template <class P>
C2D_DEV uint32_t wave_count_hits_near(const typename P::Scene& sc, uint64_t seed, uint64_t scene_id, uint64_t begin, uint32_t count,
                                      typename P::Queue& wq)
{
    auto& q = wq.sq.near;
    const uint32_t lane = threadIdx.x & 63;
    uint32_t hits = 0, qn = 0;  // wave-uniform
    const uint64_t g0 = begin >> 2;
    const uint32_t base = (uint32_t)(begin & 3);
    const uint32_t end_pos = base + count;
    const uint32_t n_groups = (end_pos + 3) >> 2;
#pragma nounroll
    for (uint32_t gi = 0; gi < n_groups; gi += 64) {
        const uint32_t p0 = 4 * gi;
        const uint32_t lo = p0 >= base ? 0u : base;
        const uint32_t rem = end_pos - p0;
        const uint32_t hi = rem < 256 ? rem : 256u;
        const bool inner = lo == 0 && hi == 256;  // every position of the iteration is this call's
        const uint64_t g = g0 + gi + lane;
        const uint32_t sidx0 = p0 - base + 4 * lane;
        const U4 r = philox_draw_block(seed, scene_id, g, 0), a = philox_draw_block(seed, scene_id, g, 1);
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const uint32_t pos = 4 * lane + (uint32_t)j;
            bool undecided = inner || (pos >= lo && pos < hi);
            unsigned long long m = inner ? wave_lanes() : (__builtin_amdgcn_ballot_w64(pos >= lo) & __builtin_amdgcn_ballot_w64(pos < hi));
            const uint32_t rw = u4_word(r, j), aw = u4_word(a, j);
#ifndef C2D_MC_NO_PRETEST
            if (sc.use_x0) {  // the radius word alone may prove the miss
                const bool c = rw < sc.x0;
                undecided = undecided && c;
                m &= __builtin_amdgcn_ballot_w64(c);
            }
#endif
            if (m == 0ull) continue;
            C2D_MC_STAT(4, __popcll(m));
            float dx, dy;
            sample_centre(sc, rw, aw, dx, dy);
#ifndef C2D_MC_NO_PRETEST
            unsigned long long miss_m;
            const bool miss = P::centre_pretest(sc, wq, dx, dy, miss_m);  // (every lane votes: no short-circuit around it)
            undecided = undecided && !miss;
            m &= ~miss_m;
            if (m == 0ull) continue;  // 64 certain misses
#endif
            if (undecided) {
                const uint32_t slot = qn + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
                q.c[slot] = make_float2(dx, dy);
                q.idx[slot] = sidx0 + (uint32_t)j;
            }
            qn += (uint32_t)__popcll(m);
        }
        while (qn >= 64) hits += evaluate_queued<P>(sc, wq, q.c, q.idx, qn, 64, seed, scene_id, begin);
    }
    if (qn) hits += evaluate_queued<P>(sc, wq, q.c, q.idx, qn, qn, seed, scene_id, begin);
    return hits;
}

#ifndef C2D_MC_NO_PRETEST
// ---- FAR: the same count for a scene whose radius test (make_scene: raw word >= x0 proves the miss) passes at most every
// other sample.  Three stages, each on 64 busy lanes, each present once in the code:
//   1  radius blocks: one Philox block per group of four samples, kIlp iterations (256 samples each) side by side —
//      independent multiply chains per lane — until 64 candidates wait or the input ends; an iteration without a
//      candidate costs nothing beyond its block;
//   2  64 candidates: angle word (block 1 of the candidate's group), Box-Muller, centre pretest; the undecided ones queue;
//   3  64 undecided samples: second pair's block (2 / 3 of the group), full evaluation.
template <class P>
C2D_DEV uint32_t wave_count_hits_far(const typename P::Scene& sc, uint64_t seed, uint64_t scene_id, uint64_t begin, uint32_t count,
                                     typename P::Queue& wq)
{
    auto& q = wq.sq.far;
    const uint32_t lane = threadIdx.x & 63;
    uint32_t hits = 0, cn = 0, un = 0;  // wave-uniform: hits, queued candidates, queued undecided samples
    const uint64_t g0 = begin >> 2;
    const uint32_t base = (uint32_t)(begin & 3);
    const uint32_t end_pos = base + count;
    const uint32_t n_groups = (end_pos + 3) >> 2;
    const uint32_t x0 = sc.x0;
    uint32_t gi = 0;

    // one lane-private word of an iteration: queue it if it is a candidate (m: the vote, a lane mask)
    auto push_word = [&](uint32_t word, bool c, unsigned long long m, uint32_t sidx) {
        if (m == 0ull) return;
        if (c) {
            const uint32_t slot = cn + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
            q.cand[slot] = make_uint2(word, sidx);
        }
        cn += (uint32_t)__popcll(m);
        C2D_MC_STAT(3, __popcll(m));
    };
    auto push_block = [&](const U4& r, uint32_t cgi) {
        if (__builtin_amdgcn_ballot_w64((r.x < x0) | (r.y < x0) | (r.z < x0) | (r.w < x0)) == 0ull) return;  // 256 certain misses
        const uint32_t p0 = 4 * cgi;
        const uint32_t lo = p0 >= base ? 0u : base;
        const uint32_t rem = end_pos - p0;
        const uint32_t hi = rem < 256 ? rem : 256u;
        const uint32_t pos = 4 * lane;
        const uint32_t sidx = p0 - base + pos;
        const uint32_t w[4] = {r.x, r.y, r.z, r.w};
        const bool edge = lo != 0 || hi != 256;  // the first / last iteration of a range: some positions are not this call's
#pragma unroll
        for (uint32_t k = 0; k < 4; k++) {
            bool c = w[k] < x0;
            unsigned long long m = __builtin_amdgcn_ballot_w64(c);  // votes of bare comparisons (see sample_collides_mask)
            if (edge) {
                const bool a = pos + k >= lo, b = pos + k < hi;
                c = c && a && b;
                m &= __builtin_amdgcn_ballot_w64(a) & __builtin_amdgcn_ballot_w64(b);
            }
            push_word(w[k], c, m, sidx + k);
        }
    };

    for (;;) {
        // ---- stage 1
        while (cn < 64 && gi + 64 * kIlp <= n_groups) {
            U4 rr[kIlp];
#pragma unroll
            for (int k = 0; k < kIlp; k++) rr[k] = philox_draw_block(seed, scene_id, g0 + gi + 64 * k + lane, 0);
#pragma unroll
            for (int k = 0; k < kIlp; k++) push_block(rr[k], gi + 64 * k);
            gi += 64 * kIlp;
        }
        while (cn < 64 && gi < n_groups) {  // the last, partial look-ahead
            push_block(philox_draw_block(seed, scene_id, g0 + gi + lane, 0), gi);
            gi += 64;
        }
        const bool drained = gi >= n_groups;
        // ---- stage 2: up to 64 candidates (fewer only when the input has ended).  Lanes read slots other lanes of this
        // wave wrote; the fences only constrain the compiler (see wave_count_hits_near)
        if (cn) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            const uint32_t take = cn < 64 ? cn : 64;
            const bool live = lane < take;
            const unsigned long long live_m = take >= 64 ? ~0ull : (1ull << take) - 1;
            const uint2 e = q.cand[cn - take + (live ? lane : 0)];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            cn -= take;
            const uint64_t s = begin + e.y;
            const U4 a = philox_draw_block(seed, scene_id, s >> 2, 1);
            const uint32_t jj = (uint32_t)s & 3u;
            const uint32_t aw = jj == 0 ? a.x : (jj == 1 ? a.y : (jj == 2 ? a.z : a.w));
            float dx, dy;
            C2D_MC_STAT(4, take);
            sample_centre(sc, e.x, aw, dx, dy);
            unsigned long long miss_m;
            const bool miss = P::centre_pretest(sc, wq, dx, dy, miss_m);  // (every lane votes: no short-circuit around it)
            const bool undecided = live && !miss;
            const unsigned long long m = live_m & ~miss_m;
            if (undecided) {
                const uint32_t slot = un + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
                q.und_c[slot] = make_float2(dx, dy);
                q.und_idx[slot] = e.y;
            }
            un += (uint32_t)__popcll(m);
        }
        // ---- stage 3: 64 undecided samples, or what is left once nothing more can come
        if (un >= 64 || (drained && cn == 0 && un)) hits += evaluate_queued<P>(sc, wq, q.und_c, q.und_idx, un, un < 64 ? un : 64u, seed, scene_id, begin);
        if (drained && cn == 0 && un == 0) break;
    }
    return hits;
}
#endif

// hits among samples [begin, begin + count) of one scene, computed by one wave
template <class P>
C2D_DEV uint32_t wave_count_hits(const typename P::Scene& sc, uint64_t seed, uint64_t scene_id, uint64_t begin, uint32_t count,
                                 typename P::Queue& q)
{
    C2D_MC_STAT(0, count);
    if (!sc.tame()) return P::plain(sc, seed, scene_id, begin, count, q);
#ifndef C2D_MC_NO_PRETEST  // validation builds evaluate every sample in full
    if (sc.use_x0 && sc.x0 < kFarX0) {
        C2D_MC_STAT(1, count);
        const uint32_t h = wave_count_hits_far<P>(sc, seed, scene_id, begin, count, q) + P::finish(sc, q);
        C2D_MC_STAT(7, h);
        return h;
    }
#endif
    C2D_MC_STAT(2, count);
    const uint32_t h = wave_count_hits_near<P>(sc, seed, scene_id, begin, count, q) + P::finish(sc, q);
    C2D_MC_STAT(7, h);
    return h;
}

// ---- many scenes, adaptive (BASELINE config 4) --------------------------------------
// The schedule state lives on the device, so the host enqueues the whole adaptive
// loop (advance, decide, advance, decide, ...) without a single read-back: the
// number of unfinished scenes, the samples drawn so far and the work split of the
// next step are read by the kernels themselves.  Steps enqueued after every scene
// has finished see n_active == 0 and retire at once.
struct AdaptiveState {
    uint32_t n_active;     // scenes still sampling
    uint32_t next_count;   // survivors appended by the running decide step
    uint32_t n_samples;    // samples drawn so far for every active scene
    uint32_t iter;         // schedule steps executed
    uint32_t ticket;       // blocks of the decide step that have finished
    uint32_t list_sel;     // which list holds the active indices
    uint32_t identity;     // 1: the active list is 0..n_active-1 (first step)
    uint32_t burst_steps;  // scratch of the burst's decide step: most steps any scene of the burst needed
    unsigned long long total_samples;
};

struct ScheduleArgs {
    uint32_t small_batch, large_batch, switch_at, max_samples;
    uint32_t want_waves;   // work items wanted per step (see c2d_mc_scenes)
    uint32_t min_chunk;    // smallest sample range worth a work item
};

C2D_DEV uint32_t batch_of(const ScheduleArgs& S, uint32_t n_samples)
{
    return n_samples < S.switch_at ? S.small_batch : S.large_batch;  // ccp.cu:283-286
}

// calcSlack (reference utils.cu:186-196), int overflow D1 fixed
C2D_DEV float calc_slack(uint32_t n, uint32_t k)
{
    if (k == n || k == 0) {
        // log(1.0 / (double)0.025f) / n, evaluated in double as in the reference
        return (float)(0x1.d82d33932720dp+1 / (double)n);
    }
    const float z = 1.96f;
    const float kf = (float)k;
    const float kk = (float)((uint64_t)k * (uint64_t)k);
    return z / (float)n * __builtin_sqrtf(kf - kk / (float)n);
}

// getBin (reference utils.cu:198-207), out-of-bounds read D2 fixed
C2D_DEV int get_bin(float p, const float* bins, uint32_t n_bins)
{
    int bin = 0;
    for (uint32_t i = 0; i + 1 < n_bins; i++)
        if (p >= bins[i] && p <= bins[i + 1]) bin = (int)i;
    return bin;
}

// Burst: the leading small-batch steps of the schedule (20 x 1000 samples by default) are run by ONE launch.  During
// those steps a scene is one work item anyway (its batch is smaller than the smallest chunk), so the wave that owns a scene
// simply goes on: batch, stop test of ccp.cu:140-148 on its own hit count, next batch ... until the test passes or the
// burst ends.  Results are the same numbers as step-by-step — the stop rule of a scene only ever looks at that scene —
// but the scene is set up once instead of once per step and 2 x (B - 1) kernel boundaries disappear.
// The arguments of an advance kernel are a struct with these fields whatever the shape (the kernels below and run_adaptive
// fill / read them by name) plus what the shape's own scene builder needs:
//     const StdDev* std_devs; const PositionWithVarAndPoseIdx* scenes; AdaptiveState* state; uint32_t* lists[2];
//     uint32_t num_std_devs; uint64_t seed, scene_id_base; ScheduleArgs sched; uint32_t* hits;
//     uint32_t burst_steps, n_bins; float bins[16], acc[16]; uint32_t* n_used; PoseCPVarAndPoseIdx* rows;
// A builder class B names them: B::Args, the sample-loop policy B::Policy, and
//     B::scene(A, row, queue)   the wave-uniform scene of one dataset row (tables gathered by index, ccp.cu:119-133)
template <bool BURST, class B>
C2D_DEV void mc_scenes_advance_body(const typename B::Args& A)
{
    using P = typename B::Policy;
    __shared__ typename P::Queue s_queue[kWavesPerBlock];
    const uint32_t n_active = A.state->n_active;
    if (n_active == 0) return;
    const uint32_t n_start = A.state->n_samples;
    if (n_start >= A.sched.max_samples) return;  // ccp.cu:281
    const uint32_t n_batch = batch_of(A.sched, n_start);
    const uint32_t* active = A.state->identity ? nullptr : A.lists[A.state->list_sel];
    // split a scene's batch over several waves when few scenes are left, so that the
    // tail of the adaptive loop still fills the chip
    const uint32_t max_split = (n_batch + A.sched.min_chunk - 1) / A.sched.min_chunk;
    uint32_t wps = (A.sched.want_waves + n_active - 1) / n_active;
    wps = wps < 1 ? 1 : (wps > max_split ? max_split : wps);
    uint32_t chunk = (n_batch + wps - 1) / wps;
    chunk = ((chunk + 255) / 256) * 256;  // whole iterations of wave_count_hits
    wps = (n_batch + chunk - 1) / chunk;

    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    C2D_MC_CLOCK_START();
    if constexpr (BURST) {  // first launch of a call: n_start == 0, identity list, one work item per scene
        for (uint64_t item = (uint64_t)blockIdx.x * kWavesPerBlock + wave; item < n_active; item += (uint64_t)gridDim.x * kWavesPerBlock) {
            const uint32_t g = (uint32_t)item;
            const PositionWithVarAndPoseIdx row = A.scenes[g];
            const typename P::Scene sc = B::scene(A, row, s_queue[wave]);
            uint32_t k = 0, n = 0;
            float p = 0.0f;
            bool done = false;
            for (uint32_t b = 0; b < A.burst_steps && !done; b++) {
                k += wave_count_hits<P>(sc, A.seed, A.scene_id_base + g, (uint64_t)n, n_batch, s_queue[wave]);
                n += n_batch;
                const float slack = calc_slack(n, k);                                       // ccp.cu:140
                p = (float)k / (float)n;                                                    // ccp.cu:142
                done = slack <= A.acc[get_bin(p, A.bins, A.n_bins)] || n >= A.sched.max_samples;  // ccp.cu:144, :281
            }
            if ((threadIdx.x & 63) == 0) {
                A.hits[g] = k;
                if (done) {  // finished inside the burst; the burst's decide step recognises it by n_used != 0
                    A.n_used[g] = n;
                    if (A.rows) {
                        PoseCPVarAndPoseIdx o;
                        o.x = row.x; o.y = row.y; o.cp = p; o.var_idx = row.var_idx; o.pose_idx = row.pose_idx;
                        A.rows[g] = o;
                    }
                }
            }
        }
        C2D_MC_CLOCK_STOP(B::clock_words());
        return;
    }
    const uint64_t n_items = (uint64_t)n_active * wps;
    for (uint64_t item = (uint64_t)blockIdx.x * kWavesPerBlock + wave; item < n_items;
         item += (uint64_t)gridDim.x * kWavesPerBlock) {
        const uint32_t slot = (uint32_t)(item / wps);
        const uint32_t ch = (uint32_t)(item % wps);
        const uint32_t off = ch * chunk;
        const uint32_t count = (n_batch - off) < chunk ? (n_batch - off) : chunk;
        const uint32_t g = active ? active[slot] : slot;
        const PositionWithVarAndPoseIdx row = A.scenes[g];
        const typename P::Scene sc = B::scene(A, row, s_queue[wave]);
        const uint32_t h = wave_count_hits<P>(sc, A.seed, A.scene_id_base + g, (uint64_t)n_start + off, count, s_queue[wave]);
        if ((threadIdx.x & 63) == 0 && h) atomicAdd(&A.hits[g], h);
    }
    C2D_MC_CLOCK_STOP(B::clock_words());
}

// After a batch: stop test of ccp.cu:140-148 per active scene, compaction of the
// survivors into the other list (replaces thrust::count + sort_by_key,
// ccp.cu:307-311), write_collision_probability (utils.cu:210-215) for the
// finished ones, and — by the block that finishes last — the roll of the schedule
// state for the next step.
struct DecideArgs {
    const PositionWithVarAndPoseIdx* scenes;
    AdaptiveState* state;
    uint32_t* lists[2];
    ScheduleArgs sched;
    float bins[16];         // accuracy_bins
    float acc[16];          // bin_accuracy
    uint32_t n_bins;
    const uint32_t* hits;
    uint32_t* n_used;
    PoseCPVarAndPoseIdx* rows;  // may be NULL
    uint32_t burst_steps;       // > 1: this is the decide step of a burst (the waves already applied the stop rule)
};

// kernels and workspace shared by every shape (defined in c2d_mc.hip)
void launch_mc_scenes_init(hipStream_t s, AdaptiveState* state, uint32_t n_scenes);
void launch_mc_scenes_decide(hipStream_t s, unsigned blocks, const DecideArgs& D);
int ensure_scene_lists(c2d_ctx* ctx, size_t n);

#ifndef C2D_MC_ITEMS_PER_CU
#define C2D_MC_ITEMS_PER_CU 512
#endif
#ifndef C2D_MC_MIN_CHUNK
#define C2D_MC_MIN_CHUNK 1024
#endif

// The host side of the adaptive loop (replaces the host loop of compute_collision_probability.cu:276-332): validates the
// shape-independent arguments, fills the shape-independent fields of the advance kernel's arguments `A` (the caller has set the
// shape's own) and enqueues the whole schedule; launch(burst, blocks, A, s) starts the shape's advance kernel.  `what` names the
// entry point in error texts.
template <class Args, class Launch>
int run_adaptive(c2d_ctx* ctx, const c2d_mc_scenes_args* a, c2d_stream stream, Args& A, Launch launch, const char* what)
{
    auto bad = [&](const char* msg) { return fail_arg(ctx, (std::string(what) + ": " + msg).c_str()); };
    if (a->total_samples) *a->total_samples = 0;
    if (a->iterations) *a->iterations = 0;
    if (a->n_scenes == 0) return C2D_OK;
    if (!a->d_std_devs || !a->d_scenes || !a->d_hits || !a->d_n_used || !a->accuracy_bins || !a->bin_accuracy) return bad("NULL argument");
    if (a->num_std_devs == 0) return bad("empty pose / std_dev table");
    if (a->n_accuracy_bins < 2 || a->n_accuracy_bins > 16) return bad("n_accuracy_bins must be 2..16");
    if (a->n_scenes > 0xffffffffull) return bad("more than 2^32-1 scenes in one call");
    ScheduleArgs S;
    S.small_batch = C2D_MC_SMALL_BATCH; S.large_batch = C2D_MC_LARGE_BATCH; S.switch_at = C2D_MC_SWITCH_AT;
    if (a->schedule_small_batch || a->schedule_large_batch || a->schedule_switch_at) {
        S.small_batch = a->schedule_small_batch; S.large_batch = a->schedule_large_batch; S.switch_at = a->schedule_switch_at;
        if (S.small_batch == 0 || S.large_batch == 0 || S.small_batch > (1u << 24) || S.large_batch > (1u << 24))
            return bad("schedule batches must be 1..2^24");
    }
    if (a->max_samples == 0 || a->max_samples > 0x7fffffffu - S.large_batch - S.small_batch) return bad("max_samples out of range");
    S.max_samples = a->max_samples;
    const uint32_t cus = (uint32_t)ctx->prop.multiProcessorCount;
    // Work items of a step = active scenes x chunks per scene.  The chip holds 5 waves per SIMD of this kernel (5120);
    // items cost between ~100 and ~430 instructions per sample depending on the scene, so the step only balances when
    // there are many more items than resident waves: CUs x 512 items of at least 1024 samples.  Reference-default batch
    // (1e5 scenes, 58 steps): 74.9 ms with CUs x 32 items, 67.8 / 63.6 / 61.2 ms with x 64 / x 128 / x 512.  Measured again at the
    // round-3 kernels: x 128 / x 256 / x 512 = 33.7 / 32.6 / 32.5 ms, and 2048 / 4096 samples at least = 32.6 / 32.4 ms (the scene
    // set-up per work item does not show); the config-4 shard does not move (324-327 ms) for any of them.
    S.want_waves = cus * C2D_MC_ITEMS_PER_CU;
    S.min_chunk = C2D_MC_MIN_CHUNK;
    // number of schedule steps until n_samples >= max_samples (ccp.cu:281-287)
    uint32_t steps = 0;
    for (uint64_t ns = 0; ns < a->max_samples; steps++) ns += ns < S.switch_at ? S.small_batch : S.large_batch;
    if (steps > 100000) return bad("more than 100000 schedule steps; use larger batches");

    DeviceGuard g(ctx->device);
    hipStream_t s = (hipStream_t)stream;
    if (int rc = workspace_acquire(ctx, s, true)) return rc;
    int st = ensure_scene_lists(ctx, a->n_scenes);
    if (st != C2D_OK) return st;
    AdaptiveState* d_state = reinterpret_cast<AdaptiveState*>(ctx->d_counters);
    static_assert(sizeof(AdaptiveState) <= 64, "AdaptiveState must fit the ctx counter block");

    // from here on the ctx state block and lists are in use on `s`: every way out of this function, the error returns
    // included, leaves the guard's stamp behind what was queued (c2d_internal.hpp)
    WorkspaceUse use(ctx, s);
    use.arm();
    launch_mc_scenes_init(s, d_state, (uint32_t)a->n_scenes);
    C2D_HIP(ctx, hipMemsetAsync(a->d_hits, 0, a->n_scenes * sizeof(uint32_t), s));
    // Burst: the leading steps that all use the small batch, when that batch is one work item per scene anyway
    uint32_t burst = 0;
    if (S.small_batch <= S.min_chunk)
        for (uint64_t ns = 0; ns < a->max_samples && ns < S.switch_at; ns += S.small_batch) burst++;
    if (burst > 1) C2D_HIP(ctx, hipMemsetAsync(a->d_n_used, 0, a->n_scenes * sizeof(uint32_t), s));  // n_used != 0 marks "finished in the burst"

    A.std_devs = a->d_std_devs; A.scenes = a->d_scenes;
    A.state = d_state; A.lists[0] = ctx->d_list[0]; A.lists[1] = ctx->d_list[1];
    A.num_std_devs = a->num_std_devs;
    A.seed = a->seed; A.scene_id_base = a->scene_id_base;
    A.sched = S; A.hits = a->d_hits;
    A.burst_steps = 0; A.n_bins = a->n_accuracy_bins; A.n_used = a->d_n_used; A.rows = a->d_rows;
    DecideArgs D;
    D.scenes = a->d_scenes; D.state = d_state; D.lists[0] = ctx->d_list[0]; D.lists[1] = ctx->d_list[1];
    D.sched = S; D.n_bins = a->n_accuracy_bins;
    for (uint32_t i = 0; i < 16; i++) {
        D.bins[i] = A.bins[i] = i < a->n_accuracy_bins ? a->accuracy_bins[i] : 0.0f;
        D.acc[i] = A.acc[i] = i + 1 < a->n_accuracy_bins ? a->bin_accuracy[i] : 0.0f;
    }
    D.hits = a->d_hits; D.n_used = a->d_n_used; D.rows = a->d_rows; D.burst_steps = 0;

    // grids sized for the largest step (every scene active); later steps leave blocks idle
    uint64_t adv_blocks = (a->n_scenes * (uint64_t)1 + kWavesPerBlock - 1) / kWavesPerBlock;
    const uint64_t adv_min = (uint64_t)cus * 32 / kWavesPerBlock, adv_max = (uint64_t)cus * 256 / kWavesPerBlock;  // 32..256 waves per CU
    adv_blocks = adv_blocks < adv_min ? adv_min : (adv_blocks > adv_max ? adv_max : adv_blocks);
    uint64_t dec_blocks = (a->n_scenes + 255) / 256;
    const uint64_t dec_max = (uint64_t)cus * 4;
    dec_blocks = dec_blocks > dec_max ? dec_max : dec_blocks;
    uint32_t it = 0;
    if (burst > 1) {  // steps 0 .. burst-1 in one advance / decide pair
        Args AB = A;
        DecideArgs DB = D;
        AB.burst_steps = DB.burst_steps = burst;
        launch(true, (unsigned)adv_blocks, AB, s);
        launch_mc_scenes_decide(s, (unsigned)dec_blocks, DB);
        it = burst;
    }
    // Steps enqueued after the last scene finished retire at once (~5 us per launch pair).  A call that hands results to
    // the host synchronises anyway, so on long schedules it looks at the state every kPollEvery steps and stops enqueuing
    // once no scene is left; a call without host outputs stays asynchronous (graph-capturable) and enqueues all of them.
    constexpr uint32_t kPollEvery = 64;
    const bool host_out = a->total_samples || a->iterations;
    AdaptiveState* h_state = reinterpret_cast<AdaptiveState*>(ctx->h_pinned);
    for (; it < steps; it++) {
        launch(false, (unsigned)adv_blocks, A, s);
        launch_mc_scenes_decide(s, (unsigned)dec_blocks, D);
        if (host_out && (it + 1) % kPollEvery == 0 && it + 1 < steps) {
            C2D_HIP(ctx, hipMemcpyAsync(h_state, d_state, sizeof(AdaptiveState), hipMemcpyDeviceToHost, s));
            C2D_HIP(ctx, hipStreamSynchronize(s));
            if (h_state->n_active == 0) break;
        }
    }
    C2D_LAUNCH_CHECK(ctx);
    use.done();      // the guard's stamp goes behind the schedule, in front of the read-back
    if (host_out) {  // host outputs requested: one read-back at the end
        C2D_HIP(ctx, hipMemcpyAsync(h_state, d_state, sizeof(AdaptiveState), hipMemcpyDeviceToHost, s));
        C2D_HIP(ctx, hipStreamSynchronize(s));
        workspace_stream_drained(ctx, s);
        if (a->total_samples) *a->total_samples = h_state->total_samples;
        if (a->iterations) *a->iterations = h_state->iter;
    }
    return C2D_OK;
}

}  // namespace c2d
