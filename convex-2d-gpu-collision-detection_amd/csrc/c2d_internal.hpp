// c2d_internal.hpp — context object and error plumbing shared by the C-ABI files.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>
#include <string>

#include "../../include/c2d.h"

struct c2d_host_pipe;                              // stream and buffers of the host-resident entry points (c2d_host.hip)
void c2d_host_pipe_free(c2d_host_pipe* p);

struct c2d_ctx {
    int device = 0;
    hipDeviceProp_t prop{};
    // small device workspace owned by the ctx (adaptive MC bookkeeping)
    uint32_t* d_list[2] = {nullptr, nullptr};  // active-scene index lists
    size_t list_capacity = 0;
    uint32_t* d_counters = nullptr;            // [0] next-active count
    unsigned long long* d_count_words = nullptr;  // 256 x 128 B arrival/sum words of the SAT count (self-clearing)
    unsigned long long* d_count_words2 = nullptr; // two-level form for the polygon kernels (c2d_count.hpp), self-clearing
    float* d_bins = nullptr;                   // accuracy_bins | bin_accuracy (<= 32 floats)
    c2d_host_pipe* host_pipe = nullptr;        // made at the first c2d_sat_rect_pairs_*_host call, kept
    void* d_scratch = nullptr;                 // grown on demand, kept: the binning pass's histograms and tables (c2d_poly_bins_from_padded)
    size_t scratch_bytes = 0;
    uint32_t* h_pinned = nullptr;              // pinned host word for count read-back
    // Deferred argument errors found by kernels (e.g. a polygon vertex count outside 1..KMAX): a pinned,
    // device-mapped word that kernels OR into with system scope; c2d_stream_synchronize /
    // c2d_ctx_check_async read and clear it, so asynchronous entry points need no validation pass.
    uint32_t* h_async_err = nullptr;
    uint32_t* d_async_err = nullptr;           // device address of the same word
    // Guard of the shared workspace (count words, adaptive state, survivor lists): the stream of the last call that
    // used it; a workspace call on another stream while that stream still has work queued is refused.
    hipStream_t ws_stream = nullptr;
    bool ws_busy = false;
    mutable std::string last_error;
};

#define C2D_ASYNC_ERR_POLY_K 1u   /* polygon vertex count outside 1..C2D_POLY_KMAX */

#define C2D_COUNT_WORDS_BYTES (256 * 128)
#define C2D_COUNT_WORDS2_BYTES (2048 * 64 + 32 * 128)

namespace c2d {

inline int fail_hip(const c2d_ctx* ctx, hipError_t e, const char* what, const char* file, int line)
{
    if (ctx) {
        char buf[512];
        std::snprintf(buf, sizeof buf, "%s failed: %s (%s:%d)", what, hipGetErrorString(e), file, line);
        ctx->last_error = buf;
    }
    return C2D_ERR_HIP;
}

inline int fail_arg(const c2d_ctx* ctx, const char* msg)
{
    if (ctx) ctx->last_error = msg;
    return C2D_ERR_INVALID_ARG;
}

#define C2D_HIP(ctx, call)                                                        \
    do {                                                                          \
        hipError_t e__ = (call);                                                  \
        if (e__ != hipSuccess) return c2d::fail_hip(ctx, e__, #call, __FILE__, __LINE__); \
    } while (0)

// Launch-error check that does not synchronise (graph-capture safe).
#define C2D_LAUNCH_CHECK(ctx)                                                     \
    do {                                                                          \
        hipError_t e__ = hipGetLastError();                                       \
        if (e__ != hipSuccess) return c2d::fail_hip(ctx, e__, "kernel launch", __FILE__, __LINE__); \
    } while (0)

struct DeviceGuard {
    int prev = -1;
    bool ok = true;
    explicit DeviceGuard(int dev)
    {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != dev) ok = hipSetDevice(dev) == hipSuccess;
    }
    ~DeviceGuard()
    {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
};

constexpr int kMaxGrid = 1 << 24;  // blocks per launch; kernels grid-stride beyond it

// Guard of the ctx workspace.  Calls on ONE stream are ordered by the stream; a workspace call on another stream is only
// accepted once everything queued on the previous workspace stream has completed (hipStreamQuery: a host-side check that
// puts nothing into either stream — an event per call costs a few microseconds of gap between back-to-back 100-us
// kernels).  Otherwise it returns C2D_ERR_UNSUPPORTED instead of silently corrupting both calls.
inline int workspace_acquire(c2d_ctx* ctx, hipStream_t s, bool uses)
{
    if (!uses || !ctx->ws_busy || ctx->ws_stream == s) return C2D_OK;
    if (hipStreamQuery(ctx->ws_stream) == hipErrorNotReady) {
        ctx->last_error = "this c2d_ctx still has a call in flight on another stream (one workspace per ctx: use one ctx per stream)";
        return C2D_ERR_UNSUPPORTED;
    }
    (void)hipGetLastError();  // a stream the caller has destroyed meanwhile reads as an invalid handle: nothing in flight
    ctx->ws_busy = false;
    return C2D_OK;
}

inline void workspace_release(c2d_ctx* ctx, hipStream_t s, bool uses)
{
    if (!uses) return;
    ctx->ws_stream = s;
    ctx->ws_busy = true;
}

inline int grid_for(size_t work_items, int block, int max_blocks)
{
    size_t b = (work_items + (size_t)block - 1) / (size_t)block;
    if (b < 1) b = 1;
    if (b > (size_t)max_blocks) b = (size_t)max_blocks;
    return (int)b;
}

}  // namespace c2d
