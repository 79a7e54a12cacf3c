// c2d_host.hip — rectangle-pair batches that live in HOST memory (include/c2d.h, c2d_sat_rect_pairs_*_host).
//
// The reference moves every batch to the device and its results back itself, one blocking cudaMemcpy after the other
// (compute_collision_probability.cu:270-274, :314-318).  Around the SAT kernels that is: upload 640 MB, test for 0.1 ms, download
// 10 MB — 11.6 ms per 1e7 vertex-format pairs, of which 11.1 ms are the host-to-device link at its measured 57.6 GB/s.  The link
// IS the cost (95.5 % of the call), so this entry point is the reference's own order of operations behind one call, with what
// it needs to be correct and bounded: whole-plane copies straight from the caller's memory (the runtime moves pageable memory as
// fast as page-locked), chunks of 2^24 pairs so that a batch of any size needs at most 1 GB of device memory (kept by the ctx
// between calls), the count on the device.  Three pipelined forms were built and measured first — chunks alternating between two streams, a copy stream plus a
// work stream with events, page-locked staging filled by eight host threads — and every one of them was SLOWER than this
// (12.0-13.2 ms): two concurrent uploads share the link no better than one sequential stream uses it, and the 0.3 ms of kernel
// and download they hide are less than what their extra copies, events and host passes cost
// (tests/tools/host_batch_bench.py, profiles/notes_r04_host_batches.md).
#include <algorithm>

#include "c2d_internal.hpp"

// the call's stream and device buffers, kept by the ctx between calls (allocating and freeing 650 MB per call costs 3.7 ms, a
// third of the call itself) and grown on demand up to one chunk
struct c2d_host_pipe {
    hipStream_t stream = nullptr;
    float* d_in = nullptr;
    uint8_t* d_out = nullptr;
    unsigned long long* d_count = nullptr;
    size_t in_floats = 0, out_bytes = 0;
};

void c2d_host_pipe_free(c2d_host_pipe* p)
{
    if (!p) return;
    if (p->stream) (void)hipStreamDestroy(p->stream);
    if (p->d_in) (void)hipFree(p->d_in);
    if (p->d_out) (void)hipFree(p->d_out);
    if (p->d_count) (void)hipFree(p->d_count);
    delete p;
}

namespace {

using namespace c2d;

constexpr size_t kHostChunk = size_t(1) << 24;   // pairs per chunk: 64 MB per plane and copy

template <int PLANES>
int run_host_batch(c2d_ctx* ctx, const float* const* h_planes, size_t n, uint8_t* h_out, unsigned long long* h_count, const char* what)
{
    if (!ctx) return C2D_ERR_INVALID_ARG;
    if (h_count) *h_count = 0;
    if (n == 0) return C2D_OK;
    if (!h_planes || !h_out) return fail_arg(ctx, (std::string(what) + ": NULL argument").c_str());
    for (int k = 0; k < PLANES; k++)
        if (!h_planes[k]) return fail_arg(ctx, (std::string(what) + ": NULL plane").c_str());
    DeviceGuard g(ctx->device);
    const size_t C = std::min(n, kHostChunk);
    const size_t pitch = (C + 3) / 4 * 4;   // floats between the device planes: 16-byte aligned starts
    if (!ctx->host_pipe) ctx->host_pipe = new (std::nothrow) c2d_host_pipe();
    c2d_host_pipe* P = ctx->host_pipe;
    if (!P) return C2D_ERR_NOMEM;
    hipError_t e = hipSuccess;
    if (!P->stream) e = hipStreamCreateWithFlags(&P->stream, hipStreamNonBlocking);
    if (e == hipSuccess && !P->d_count) e = hipMalloc(&P->d_count, sizeof *P->d_count);
    if (e == hipSuccess && P->in_floats < (size_t)PLANES * pitch) {
        if (P->d_in) (void)hipFree(P->d_in);
        P->d_in = nullptr; P->in_floats = 0;
        e = hipMalloc(&P->d_in, (size_t)PLANES * pitch * sizeof(float));
        if (e == hipSuccess) P->in_floats = (size_t)PLANES * pitch;
    }
    if (e == hipSuccess && P->out_bytes < C) {
        if (P->d_out) (void)hipFree(P->d_out);
        P->d_out = nullptr; P->out_bytes = 0;
        e = hipMalloc(&P->d_out, C);
        if (e == hipSuccess) P->out_bytes = C;
    }
    hipStream_t s = P->stream;
    float* d_in = P->d_in;
    uint8_t* d_out = P->d_out;
    unsigned long long* d_count = h_count ? P->d_count : nullptr;
    if (e == hipSuccess && h_count) e = hipMemsetAsync(d_count, 0, sizeof *d_count, s);
    if (e != hipSuccess) {
        if (e == hipErrorOutOfMemory) { ctx->last_error = std::string(what) + ": out of device memory"; return C2D_ERR_NOMEM; }
        return fail_hip(ctx, e, "host batch set-up", __FILE__, __LINE__);
    }
    int st = C2D_OK;
    const float* d_planes[PLANES];
    for (int k = 0; k < PLANES; k++) d_planes[k] = d_in + (size_t)k * pitch;
    for (size_t off = 0; off < n && st == C2D_OK; off += C) {
        const size_t m = std::min(C, n - off);
        for (int k = 0; k < PLANES && e == hipSuccess; k++)
            e = hipMemcpyAsync(const_cast<float*>(d_planes[k]), h_planes[k] + off, m * sizeof(float), hipMemcpyHostToDevice, s);
        if (e != hipSuccess) { st = fail_hip(ctx, e, "upload", __FILE__, __LINE__); break; }
        if constexpr (PLANES == 16) st = c2d_sat_rect_pairs_verts(ctx, d_planes, m, d_out, d_count, s);
        else st = c2d_sat_rect_pairs_pose(ctx, d_planes, m, d_out, d_count, s);
        if (st != C2D_OK) break;
        e = hipMemcpyAsync(h_out + off, d_out, m, hipMemcpyDeviceToHost, s);
        if (e != hipSuccess) { st = fail_hip(ctx, e, "download", __FILE__, __LINE__); break; }
    }
    if (st == C2D_OK && h_count) {
        e = hipMemcpyAsync(h_count, d_count, sizeof *d_count, hipMemcpyDeviceToHost, s);
        if (e != hipSuccess) st = fail_hip(ctx, e, "count read-back", __FILE__, __LINE__);
    }
    e = hipStreamSynchronize(s);  // (also after an error: nothing of this call stays in flight)
    if (e != hipSuccess && st == C2D_OK) st = fail_hip(ctx, e, "hipStreamSynchronize", __FILE__, __LINE__);
    if (e == hipSuccess) workspace_stream_drained(ctx, s);   // (only tickets issued on this call's own stream count as retired)
    return st;
}

}  // namespace

extern "C" {

int c2d_sat_rect_pairs_verts_host(c2d_ctx* ctx, const float* const h_planes[16], size_t n, uint8_t* h_out, unsigned long long* h_count)
{
    return run_host_batch<16>(ctx, h_planes, n, h_out, h_count, "c2d_sat_rect_pairs_verts_host");
}

int c2d_sat_rect_pairs_pose_host(c2d_ctx* ctx, const float* const h_pose_planes[10], size_t n, uint8_t* h_out, unsigned long long* h_count)
{
    return run_host_batch<10>(ctx, h_pose_planes, n, h_out, h_count, "c2d_sat_rect_pairs_pose_host");
}

}  // extern "C"
