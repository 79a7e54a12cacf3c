// c2d_internal.hpp — context object and error plumbing shared by the C-ABI files.
#pragma once

#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>
#include <string>

#include "../../include/c2d.h"
#include "c2d_count.hpp"

struct c2d_host_pipe;                              // stream and buffers of the host-resident entry points (c2d_host.hip)
void c2d_host_pipe_free(c2d_host_pipe* p);

struct c2d_ctx {
    int device = 0;
    hipDeviceProp_t prop{};
    // small device workspace owned by the ctx (adaptive MC bookkeeping)
    uint32_t* d_list[2] = {nullptr, nullptr};  // active-scene index lists
    size_t list_capacity = 0;
    uint32_t* d_counters = nullptr;            // [0] next-active count
    // ONE block (c2d_count.hpp): 256 x 128 B arrival/sum words of the SAT count | the two-level form for the polygon kernels |
    // the completion stamps of the workspace guard.  The words are self-clearing, the stamps only ever rise.
    unsigned long long* d_count_words = nullptr;
    unsigned long long* d_count_words2 = nullptr; // = d_count_words + kCountWordsBytes / 8
    unsigned long long* d_ws_stamps = nullptr;    // = d_count_words + kWorkspaceStampsOffset / 8
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
    // Guard of the shared workspace (count words, adaptive state, survivor lists, scratch): see workspace_acquire below.
    // ws_stream is only ever COMPARED with the stream of the next call, never handed to the runtime: the caller may have
    // destroyed it — and the runtime may then hand the SAME ADDRESS to the next stream it creates (profiles/r05_stream_lifetime_probe.txt,
    // cases 2 and 4), so the pointer alone does not say "the same stream": ws_stream_id is hipStreamGetId of the stream at the moment
    // the tickets were issued (a number the runtime never gives twice), compared with the id of the LIVE stream of the next call.
    hipStream_t ws_stream = nullptr;
    unsigned long long ws_stream_id = 0;
    bool ws_ticket_pending = false;                 // a ticket was taken for a launch whose launch check has not run yet (C2D_LAUNCH_CHECK)
    bool ws_outstanding = false;                    // tickets were issued (on ws_stream) that no check has seen retired yet
    unsigned ws_ticket = 0;                         // the last ticket handed out (32 bits: workspace_next_ticket handles the wrap)
    unsigned ws_expect[c2d::kStampSlots] = {};      // per stamp: the ticket it must have reached
    hipStream_t ws_probe_stream = nullptr;          // the guard's own stream and page-locked copy of the workspace block (words and
    unsigned long long* h_ws_block = nullptr;       // stamps), made at the first check that needs them
    mutable std::string last_error;
};

#define C2D_ASYNC_ERR_POLY_K 1u   /* polygon vertex count outside 1..C2D_POLY_KMAX */


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

// Launch-error check that does not synchronise (graph-capture safe).  A launch that took a workspace ticket and then failed has
// no wave that will ever raise the stamps to it: the ticket is taken back (workspace_launch_failed below), or every later call
// on another stream would be refused until the caller synchronised a stream it may no longer have.
#define C2D_LAUNCH_CHECK(ctx)                                                     \
    do {                                                                          \
        hipError_t e__ = hipGetLastError();                                       \
        if (e__ != hipSuccess) {                                                  \
            c2d::workspace_launch_failed(ctx);                                    \
            return c2d::fail_hip(ctx, e__, "kernel launch", __FILE__, __LINE__);  \
        }                                                                         \
        c2d::workspace_launch_ok(ctx);                                            \
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

// Guard of the ctx workspace (count words, adaptive state, survivor lists, scratch).  Calls on ONE stream are ordered by
// the stream; a workspace call on ANOTHER stream is accepted only once everything the ctx launched on its workspace has
// retired, and is refused with C2D_ERR_UNSUPPORTED otherwise instead of silently corrupting both calls.
//
// "Has retired" is read from completion stamps the kernels raise themselves (c2d_count.hpp): every launch that uses the
// workspace carries a ticket, the wave that completes one of the launch's count words raises that word's stamp to the
// ticket (calls without counting kernels stamp with a one-thread kernel behind their last launch), and the host remembers
// per stamp the ticket it must reach.  The check is one read of the workspace block (167 KB: the words and their stamps) on a
// stream the ctx owns, made only when the stream changes while tickets are outstanding: a word counts as retired when its
// stamp has reached its ticket AND the word itself reads zero again (c2d_count.hpp: that pair is the final state whatever order
// the kernel's two atomics land in, so the kernels need no fence).  The hot path pays one 32-bit kernel argument and one
// non-returning atomic per completed word.  Round 4 asked the runtime instead (hipStreamQuery on the previous call's stream) — a handle whose
// lifetime belongs to the caller; profiles/notes_r05_workspace_guard.md has what that cost.
//
// Graph capture: a call on a capturing stream takes no ticket and makes no check — nothing executes during capture, and
// the order of a graph's replays against other work on the ctx is the caller's, as include/c2d.h states.
inline bool stream_is_capturing(hipStream_t s)
{
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &st) != hipSuccess) { (void)hipGetLastError(); return false; }
    return st != hipStreamCaptureStatusNone;
}

// What the runtime calls the LIVE stream `s` (the caller is about to launch on it): a number it gives once, to one stream.
// hipStreamGetId exists since HIP 7.1 and is looked up in the libamdhip64 this process already holds, not linked: a PyTorch process
// has loaded PyTorch's own libamdhip64.so.7 (HIP 7.0 in this image) before libc2d.so, and a library that NEEDS the symbol does not
// even load there ("version `hip_7.1' not found").  kUnknownStreamId when the runtime has no such call or will not say: identity
// then cannot be proven by id, and the address alone decides (workspace_same_stream below says what that leaves open).
constexpr unsigned long long kUnknownStreamId = ~0ull;
using StreamGetIdFn = hipError_t (*)(hipStream_t, unsigned long long*);
inline StreamGetIdFn stream_get_id_fn()
{
    static const StreamGetIdFn fn = [] {
        void* h = dlopen("libamdhip64.so.7", RTLD_NOLOAD | RTLD_NOW);   // the copy this process is already bound to, by soname
        void* sym = h ? dlsym(h, "hipStreamGetId") : nullptr;
        if (!sym) sym = dlsym(RTLD_DEFAULT, "hipStreamGetId");
        return reinterpret_cast<StreamGetIdFn>(sym);
    }();
    return fn;
}
inline unsigned long long stream_identity(hipStream_t s)
{
    const StreamGetIdFn fn = stream_get_id_fn();
    unsigned long long id = 0;
    if (!fn) return kUnknownStreamId;
    if (fn(s, &id) != hipSuccess) { (void)hipGetLastError(); return kUnknownStreamId; }
    return id;
}

// Is the call on live stream `s` ordered behind the outstanding tickets, i.e. is `s` THE stream they were issued on?
// The same address is necessary, not sufficient: a stream destroyed with work in flight and a new one created in its place are two
// streams.  Where the runtime numbers its streams, the number decides.  Where it does not (HIP 7.0: a PyTorch process)
// the address is all there is: c2d_stream_destroy then forgets the address of a stream it destroys with tickets outstanding
// (forgotten_stream, below), and what stays open is a stream destroyed BEHIND the ctx's back with work in flight whose address
// the runtime hands to a new stream that the caller uses at once — include/c2d.h says so.  (Asking the live stream whether it is
// idle — hipStreamQuery on the caller's own handle, an idle stream orders nothing so the stamps decide — closes all but a corner
// of that, and costs 3 % of the headline kernel: the runtime puts a marker behind the last kernel to answer.
// profiles/r06_idle_query_ab.txt; not shipped.)
inline bool workspace_same_stream(c2d_ctx* ctx, hipStream_t s)
{
    if (ctx->ws_stream != s) return false;
    if (ctx->ws_stream_id != kUnknownStreamId) return stream_identity(s) == ctx->ws_stream_id;
    return true;
}

// what ws_stream holds once c2d_stream_destroy has destroyed the stream of the outstanding tickets: no live stream's address
inline hipStream_t forgotten_stream() { return reinterpret_cast<hipStream_t>(~static_cast<uintptr_t>(0)); }

inline int workspace_acquire(c2d_ctx* ctx, hipStream_t s, bool uses)
{
    ctx->ws_ticket_pending = false;   // (a new call: whatever ticket an earlier call took has had its launch check)
    if (!uses || !ctx->ws_outstanding) return C2D_OK;
    // the same stream as the outstanding tickets': the stream orders the calls.  The same ADDRESS is not enough — a stream destroyed
    // with work in flight and a new one created in its place are two streams.
    if (workspace_same_stream(ctx, s)) return C2D_OK;
    if (stream_is_capturing(s)) return C2D_OK;
    if (!ctx->ws_probe_stream) {
        hipError_t e = hipStreamCreateWithFlags(&ctx->ws_probe_stream, hipStreamNonBlocking);
        if (e == hipSuccess) e = hipHostMalloc(reinterpret_cast<void**>(&ctx->h_ws_block), kWorkspaceBytes, hipHostMallocDefault);
        if (e != hipSuccess) {
            if (ctx->ws_probe_stream) { (void)hipStreamDestroy(ctx->ws_probe_stream); ctx->ws_probe_stream = nullptr; }
            return fail_hip(ctx, e, "workspace guard set-up", __FILE__, __LINE__);
        }
    }
    hipError_t e = hipMemcpyAsync(ctx->h_ws_block, ctx->d_count_words, kWorkspaceBytes, hipMemcpyDeviceToHost, ctx->ws_probe_stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->ws_probe_stream);
    if (e != hipSuccess) return fail_hip(ctx, e, "workspace guard read", __FILE__, __LINE__);
    const unsigned long long* words = ctx->h_ws_block;                                  // 256 words, 128 bytes apart
    const unsigned long long* words1 = ctx->h_ws_block + kCountWordsBytes / 8;          // 2048 first-level words, 64 bytes apart
    const unsigned long long* words2 = words1 + (size_t)kCountWords1 * 8;               // 32 second-level words, 128 bytes apart
    const unsigned long long* stamps = ctx->h_ws_block + kWorkspaceStampsOffset / 8;
    bool busy = false;
    for (uint32_t i = 0; i < kStampSlots && !busy; i++) busy = stamps[i] < ctx->ws_expect[i];
    for (uint32_t i = 0; i < kCountWords && !busy; i++) busy = words[(size_t)i * 16] != 0;
    for (uint32_t i = 0; i < kCountWords1 && !busy; i++) busy = words1[(size_t)i * 8] != 0;
    for (uint32_t i = 0; i < kCountWords2 && !busy; i++) busy = words2[(size_t)i * 16] != 0;
    if (busy) {
        ctx->last_error = "this c2d_ctx still has a call in flight on another stream (one workspace per ctx: use one ctx per stream)";
        return C2D_ERR_UNSUPPORTED;
    }
    ctx->ws_outstanding = false;
    return C2D_OK;
}

// The next ticket.  Tickets are 32 bits and only ever compared by magnitude, so when they run out — every 2^32 launches — the
// device is drained once, stamps and expectations go back to zero and counting starts again at 1.
inline unsigned workspace_next_ticket(c2d_ctx* ctx)
{
    if (ctx->ws_ticket == 0xffffffffu) {
        (void)hipDeviceSynchronize();
        (void)hipMemset(ctx->d_ws_stamps, 0, kStampSlots * sizeof(unsigned long long));
        std::memset(ctx->ws_expect, 0, sizeof ctx->ws_expect);
        ctx->ws_outstanding = false;
        ctx->ws_ticket = 0;
    }
    return ++ctx->ws_ticket;
}

// tickets are outstanding on the live stream `s` from now on (the id is asked for only when the stream changes: `s` == ws_stream with
// tickets outstanding has passed workspace_acquire's identity check in this very call)
inline void workspace_tickets_on(c2d_ctx* ctx, hipStream_t s)
{
    if (!ctx->ws_outstanding || ctx->ws_stream != s) ctx->ws_stream_id = stream_identity(s);
    ctx->ws_stream = s;
    ctx->ws_outstanding = true;
}

// the ticket of one launch whose waves count through the single-level words / the two-level words; call it right before the
// launch on stream `s` with the launch's number of waves
inline CountWs workspace_count_ticket(c2d_ctx* ctx, hipStream_t s, size_t n_waves, bool counted)
{
    CountWs ws{ctx->d_count_words, 0};
    if (!counted || stream_is_capturing(s)) return ws;
    ws.ticket = workspace_next_ticket(ctx);
    const size_t used = n_waves < kCountWords ? n_waves : kCountWords;
    for (size_t i = 0; i < used; i++) ctx->ws_expect[i] = ws.ticket;
    workspace_tickets_on(ctx, s);
    ctx->ws_ticket_pending = true;
    return ws;
}

inline CountWs workspace_count_ticket2(c2d_ctx* ctx, hipStream_t s, size_t n_waves, bool counted)
{
    CountWs ws{ctx->d_count_words2, 0};
    if (!counted || stream_is_capturing(s)) return ws;
    ws.ticket = workspace_next_ticket(ctx);
    size_t used = n_waves < kCountWords1 ? n_waves : kCountWords1;
    used = used < kCountWords2 ? used : kCountWords2;
    for (size_t i = 0; i < used; i++) ctx->ws_expect[kCountWords + i] = ws.ticket;
    workspace_tickets_on(ctx, s);
    ctx->ws_ticket_pending = true;
    return ws;
}

// For calls whose kernels do not count (the adaptive schedule): a one-thread kernel behind everything the call has queued on
// `s` raises the extra stamp.  If even that cannot be launched the stream is drained instead, so that nothing of the call
// stays in flight unguarded.  (c2d_api.hip)
void workspace_stamp_behind(c2d_ctx* ctx, hipStream_t s);

// Arms at the call's first enqueue and stamps on EVERY way out of the scope, error returns included.
struct WorkspaceUse {
    c2d_ctx* ctx;
    hipStream_t s;
    bool armed = false;
    WorkspaceUse(c2d_ctx* c, hipStream_t st) : ctx(c), s(st) {}
    WorkspaceUse(const WorkspaceUse&) = delete;
    WorkspaceUse& operator=(const WorkspaceUse&) = delete;
    void arm() { armed = true; }
    // everything is queued: stamp now (a caller that goes on to synchronise the stream then finds the guard's tickets retired)
    void done()
    {
        if (armed) workspace_stamp_behind(ctx, s);
        armed = false;
    }
    ~WorkspaceUse() { done(); }
};

// `s` has been synchronised by the caller: if it is the stream the outstanding tickets were issued on, they have retired.
inline void workspace_stream_drained(c2d_ctx* ctx, hipStream_t s)
{
    if (ctx->ws_outstanding && workspace_same_stream(ctx, s))
        ctx->ws_outstanding = false;
}

// C2D_LAUNCH_CHECK's two ways out.  After a failed launch that had taken a ticket nothing will raise the stamps to it.  ws_stream is
// the stream of the failing call (the ticket set it), hence live: drain it — every ticket outstanding on the ctx was issued on
// it, a change of stream is only ever accepted with nothing outstanding — and start again with no expectations.
inline void workspace_launch_ok(c2d_ctx* ctx) { ctx->ws_ticket_pending = false; }
inline void workspace_launch_failed(c2d_ctx* ctx)
{
    if (!ctx->ws_ticket_pending) return;
    ctx->ws_ticket_pending = false;
    if (hipStreamSynchronize(ctx->ws_stream) != hipSuccess) { (void)hipGetLastError(); return; }   // (cannot prove anything retired: stay guarded)
    std::memset(ctx->ws_expect, 0, sizeof ctx->ws_expect);
    ctx->ws_outstanding = false;
}
inline void workspace_launch_ok(const c2d_ctx*) {}       // (entry points that hold the ctx const take no tickets)
inline void workspace_launch_failed(const c2d_ctx*) {}

inline int grid_for(size_t work_items, int block, int max_blocks)
{
    size_t b = (work_items + (size_t)block - 1) / (size_t)block;
    if (b < 1) b = 1;
    if (b > (size_t)max_blocks) b = (size_t)max_blocks;
    return (int)b;
}

}  // namespace c2d
