// c2d_count.hpp — colliding-pair count shared by the SAT kernels (rectangles and polygons).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "c2d_math.hpp"

namespace c2d {

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
constexpr uint32_t kCountWords1 = 2048;   // two-level form below: x 64 B
constexpr uint32_t kCountWords2 = 32;     // x 128 B, behind the first level
constexpr size_t kCountWordsBytes = (size_t)kCountWords * 128;
constexpr size_t kCountWords2Bytes = (size_t)kCountWords1 * 64 + (size_t)kCountWords2 * 128;

// Completion stamps of the workspace (the guard in c2d_internal.hpp): ONE block holds the single-level words, the two-level
// words and, behind them, one 64-bit stamp per word that can complete a launch — 256 for the single-level form, 32 for the
// second level of the two-level form, 1 for the calls that stamp with a kernel of their own (the adaptive Monte-Carlo
// schedule).  A launch carries a ticket; the wave that completes a word raises that word's stamp to the ticket, so the host
// can tell "everything this ctx launched on its workspace has retired" from the stamps alone — without asking the runtime
// about a stream handle whose lifetime belongs to the caller.  Ticket 0 = no stamping (graph capture).  The ticket travels as 32
// bits: a 64-bit kernel argument cost sat_poly_kernel<16, 5, true> — at its limit of scalar registers — two spilled SGPRs, through
// them two VGPRs in scratch, and 8 % of its run time (430 -> 465 us per 1e7 pairs, profiles/r05a_kernel_stats.csv); the host handles
// the wrap every 2^32 launches (c2d_internal.hpp).
constexpr uint32_t kStampOther = kCountWords + kCountWords2;   // index of the extra stamp
constexpr uint32_t kStampSlots = kStampOther + 1;
constexpr size_t kWorkspaceStampsOffset = kCountWordsBytes + kCountWords2Bytes;           // bytes from the start of the block
constexpr size_t kWorkspaceBytes = kWorkspaceStampsOffset + (size_t)kStampSlots * 8;

struct CountWs {
    unsigned long long* words;   // the single-level words (block + 0) or the two-level words (block + kCountWordsBytes)
    unsigned ticket;
};

// The stamp is ONE relaxed, non-returning atomic behind the word's clearing — no fence between the two.  A release fence there
// costs 1 % of the 104-us headline kernel (256 completing waves wait for their writes and write the L2 back at the kernel's
// tail; measured A/B in one process, profiles/notes_r05_workspace_guard.md), and it is not needed: the guard accepts a word only
// when it reads the stamp raised AND the word itself back at zero.  A raised stamp means every arrival at the word has been
// performed (the completing wave held the full count), after which the word stays non-zero until the clearing lands — so
// "raised and zero" is the final state whatever order the two atomics reach memory in.
// (C2D_WS_STAMP_MODE is a measurement switch of `make lib-ab-stamps`: 0 = no stamp at all — round 4's kernels, and a guard that
// refuses for ever —, 2 = with the release fence; the product is 1)
#ifndef C2D_WS_STAMP_MODE
#define C2D_WS_STAMP_MODE 1
#endif
C2D_DEV void stamp_raise(unsigned long long* stamp, unsigned ticket)
{
#if C2D_WS_STAMP_MODE > 0
    if (ticket) {
#if C2D_WS_STAMP_MODE > 1
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
#endif
        atomicMax(reinterpret_cast<unsigned*>(stamp), ticket);   // (the low word of the 64-bit slot; the high word stays 0)
    }
#endif
}

// `v` is the wave's total (wave-uniform); every wave of the grid must call this exactly once.
C2D_DEV void wave_count_arrive_total(uint32_t v, unsigned long long* __restrict__ d_count, CountWs ws)
{
    unsigned long long* __restrict__ words = ws.words;
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
            stamp_raise(words + kWorkspaceStampsOffset / 8 + slot, ws.ticket);
        }
    }
}

// Two-level form for launches with very many waves (the polygon kernels: one wave per 64 pairs, 156 250 waves per 1e7
// pairs).  Returning atomics on ONE address complete about every 64 ns, so 610 arrivals per word cost the single-level
// scheme ~40 us at the end of a 300-us kernel (measured on sat_poly_binned_kernel: 0.344 ms with the count, 0.296 without).
// Here a wave arrives at one of 2048 first-level words (76 arrivals each at that size); the wave that completes a
// first-level word carries its sum to one of 32 second-level words, and the wave that completes one of those adds to the
// caller's counter: at most 32 adds on that word, every chain short, all words self-clearing as above.
C2D_DEV void wave_count_arrive_total2(uint32_t v, unsigned long long* __restrict__ d_count, CountWs ws)
{
    unsigned long long* __restrict__ words2 = ws.words;
    if ((threadIdx.x & 63) == 0) {
        const uint32_t waves_per_block = blockDim.x >> 6;
        const uint32_t wave_id = blockIdx.x * waves_per_block + (threadIdx.x >> 6);
        const uint32_t n_waves = gridDim.x * waves_per_block;
        const uint32_t slot1 = wave_id & (kCountWords1 - 1);
        const uint32_t expected1 = n_waves / kCountWords1 + (slot1 < (n_waves & (kCountWords1 - 1)) ? 1u : 0u);
        unsigned long long* w1 = words2 + (size_t)slot1 * 8;
        const unsigned long long old1 = atomicAdd(w1, (1ull << 40) | (unsigned long long)v);
        if ((uint32_t)(old1 >> 40) + 1u == expected1) {
            const unsigned long long total1 = (old1 & ((1ull << 40) - 1)) + v;
            atomicExch(w1, 0ull);
            const uint32_t used1 = n_waves < kCountWords1 ? n_waves : kCountWords1;   // first-level words with arrivals
            const uint32_t slot2 = slot1 & (kCountWords2 - 1);
            const uint32_t expected2 = used1 / kCountWords2 + (slot2 < (used1 & (kCountWords2 - 1)) ? 1u : 0u);
            unsigned long long* w2 = words2 + (size_t)kCountWords1 * 8 + (size_t)slot2 * 16;
            const unsigned long long old2 = atomicAdd(w2, (1ull << 40) | total1);
            if ((uint32_t)(old2 >> 40) + 1u == expected2) {
                const unsigned long long total2 = (old2 & ((1ull << 40) - 1)) + total1;
                atomicExch(w2, 0ull);
                if (total2) atomicAdd(d_count, total2);
                stamp_raise(words2 + (kWorkspaceStampsOffset - kCountWordsBytes) / 8 + kCountWords + slot2, ws.ticket);
            }
        }
    }
}

// per-lane partial counts
C2D_DEV void wave_count_arrive(uint32_t lane_count, unsigned long long* __restrict__ d_count, CountWs ws)
{
    uint32_t v = lane_count;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    wave_count_arrive_total((uint32_t)__builtin_amdgcn_readfirstlane((int)v), d_count, ws);
}

}  // namespace c2d
