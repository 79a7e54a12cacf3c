// c2d_api.hip — context, memory and stream plumbing of the C-ABI (include/c2d.h).
// Replaces the cudaMalloc / cudaMemcpy / cudaFree / error-macro scaffolding of the
// reference mains (compute_collision_probability.cu:212-251, utils.cu:59-72).
#include "c2d_internal.hpp"

namespace c2d {

__global__ void workspace_stamp_kernel(unsigned long long* stamp, unsigned ticket) { atomicMax(reinterpret_cast<unsigned*>(stamp), ticket); }

void workspace_stamp_behind(c2d_ctx* ctx, hipStream_t s)
{
    if (stream_is_capturing(s)) return;
    const unsigned ticket = workspace_next_ticket(ctx);
    hipLaunchKernelGGL(workspace_stamp_kernel, dim3(1), dim3(1), 0, s, ctx->d_ws_stamps + kStampOther, ticket);
    if (hipGetLastError() != hipSuccess) {   // nothing of the call may stay in flight behind an unarmed guard
        (void)hipStreamSynchronize(s);
        (void)hipGetLastError();
        return;
    }
    ctx->ws_expect[kStampOther] = ticket;
    workspace_tickets_on(ctx, s);
}

}  // namespace c2d

extern "C" {

int c2d_version(void) { return C2D_VERSION_MAJOR * 1000 + C2D_VERSION_MINOR; }

const char* c2d_status_string(int status)
{
    switch (status) {
    case C2D_OK: return "ok";
    case C2D_ERR_INVALID_ARG: return "invalid argument";
    case C2D_ERR_HIP: return "HIP runtime error";
    case C2D_ERR_NO_DEVICE: return "no usable device";
    case C2D_ERR_NOMEM: return "out of memory";
    case C2D_ERR_UNSUPPORTED: return "unsupported argument combination";
    case C2D_ERR_DIST: return "multi-GPU (RCCL) error";
    default: return "unknown status";
    }
}

const char* c2d_last_error(const c2d_ctx* ctx) { return ctx ? ctx->last_error.c_str() : ""; }

int c2d_device_count(int* count)
{
    if (!count) return C2D_ERR_INVALID_ARG;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { *count = 0; return C2D_ERR_NO_DEVICE; }
    *count = n;
    return C2D_OK;
}

int c2d_ctx_create(int device, c2d_ctx** out)
{
    if (!out) return C2D_ERR_INVALID_ARG;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device < 0 || device >= n) return C2D_ERR_NO_DEVICE;
    c2d_ctx* ctx = new (std::nothrow) c2d_ctx();
    if (!ctx) return C2D_ERR_NOMEM;
    ctx->device = device;
    if (hipGetDeviceProperties(&ctx->prop, device) != hipSuccess) { delete ctx; return C2D_ERR_NO_DEVICE; }
    // The code object holds gfx950 kernels only: fail loudly anywhere else.
    if (std::strncmp(ctx->prop.gcnArchName, "gfx950", 6) != 0) { delete ctx; return C2D_ERR_NO_DEVICE; }
    c2d::DeviceGuard g(device);
    if (!g.ok) { delete ctx; return C2D_ERR_NO_DEVICE; }
    if (hipMalloc(&ctx->d_counters, 64) != hipSuccess || hipMalloc(&ctx->d_count_words, c2d::kWorkspaceBytes) != hipSuccess ||
        hipMemset(ctx->d_count_words, 0, c2d::kWorkspaceBytes) != hipSuccess || hipMalloc(&ctx->d_bins, 32 * sizeof(float)) != hipSuccess ||
        hipHostMalloc(reinterpret_cast<void**>(&ctx->h_pinned), 64, hipHostMallocDefault) != hipSuccess ||
        hipHostMalloc(reinterpret_cast<void**>(&ctx->h_async_err), 64, hipHostMallocMapped) != hipSuccess ||
        hipHostGetDevicePointer(reinterpret_cast<void**>(&ctx->d_async_err), ctx->h_async_err, 0) != hipSuccess) {
        if (ctx->d_counters) (void)hipFree(ctx->d_counters);
        if (ctx->d_count_words) (void)hipFree(ctx->d_count_words);
        if (ctx->d_bins) (void)hipFree(ctx->d_bins);
        if (ctx->h_pinned) (void)hipHostFree(ctx->h_pinned);
        if (ctx->h_async_err) (void)hipHostFree(ctx->h_async_err);
        delete ctx;
        return C2D_ERR_NOMEM;
    }
    *ctx->h_async_err = 0;
    ctx->d_count_words2 = ctx->d_count_words + c2d::kCountWordsBytes / 8;
    ctx->d_ws_stamps = ctx->d_count_words + c2d::kWorkspaceStampsOffset / 8;
    *out = ctx;
    return C2D_OK;
}

int c2d_ctx_destroy(c2d_ctx* ctx)
{
    if (!ctx) return C2D_OK;
    c2d::DeviceGuard g(ctx->device);
    for (auto& p : ctx->d_list)
        if (p) (void)hipFree(p);
    if (ctx->d_counters) (void)hipFree(ctx->d_counters);
    if (ctx->d_count_words) (void)hipFree(ctx->d_count_words);
    if (ctx->ws_probe_stream) (void)hipStreamDestroy(ctx->ws_probe_stream);
    if (ctx->h_ws_block) (void)hipHostFree(ctx->h_ws_block);
    if (ctx->d_bins) (void)hipFree(ctx->d_bins);
    if (ctx->d_scratch) (void)hipFree(ctx->d_scratch);
    c2d_host_pipe_free(ctx->host_pipe);
    if (ctx->h_pinned) (void)hipHostFree(ctx->h_pinned);
    if (ctx->h_async_err) (void)hipHostFree(ctx->h_async_err);
    delete ctx;
    return C2D_OK;
}

int c2d_ctx_info_sized(const c2d_ctx* ctx, c2d_device_info* out, size_t out_bytes)
{
    if (!ctx || !out || out_bytes == 0) return C2D_ERR_INVALID_ARG;
    c2d_device_info di;
    std::memset(&di, 0, sizeof di);
    // (some boxes of the pool report an empty marketing name: the architecture then stands in for it)
    if (ctx->prop.name[0]) std::snprintf(di.name, sizeof di.name, "%s", ctx->prop.name);
    else std::snprintf(di.name, sizeof di.name, "gfx950 device (the runtime reports no marketing name)");
    std::snprintf(di.arch, sizeof di.arch, "%s", ctx->prop.gcnArchName);
    di.device = ctx->device;
    di.compute_units = ctx->prop.multiProcessorCount;
    di.wavefront_size = ctx->prop.warpSize;
    di.lds_bytes_per_cu = (int)ctx->prop.maxSharedMemoryPerMultiProcessor;
    di.hbm_bytes = ctx->prop.totalGlobalMem;
    if (hipDeviceGetPCIBusId(di.pci_bus_id, (int)sizeof di.pci_bus_id, ctx->device) != hipSuccess) {
        (void)hipGetLastError();
        di.pci_bus_id[0] = 0;
    }
    std::memcpy(out, &di, out_bytes < sizeof di ? out_bytes : sizeof di);   // never past the caller's struct
    return C2D_OK;
}

// the exported symbol of the 0.4 layout (include/c2d.h): for binaries built before the struct grew
int (c2d_ctx_info)(const c2d_ctx* ctx, c2d_device_info* out) { return c2d_ctx_info_sized(ctx, out, C2D_DEVICE_INFO_BYTES_0_4); }

int c2d_malloc(c2d_ctx* ctx, void** d_ptr, size_t bytes)
{
    if (!ctx || !d_ptr) return C2D_ERR_INVALID_ARG;
    *d_ptr = nullptr;
    if (bytes == 0) return C2D_OK;
    c2d::DeviceGuard g(ctx->device);
    hipError_t e = hipMalloc(d_ptr, bytes);
    if (e == hipErrorOutOfMemory) { ctx->last_error = "hipMalloc: out of memory"; return C2D_ERR_NOMEM; }
    if (e != hipSuccess) return c2d::fail_hip(ctx, e, "hipMalloc", __FILE__, __LINE__);
    return C2D_OK;
}

int c2d_free(c2d_ctx* ctx, void* d_ptr)
{
    if (!ctx) return C2D_ERR_INVALID_ARG;
    if (!d_ptr) return C2D_OK;
    c2d::DeviceGuard g(ctx->device);
    C2D_HIP(ctx, hipFree(d_ptr));
    return C2D_OK;
}

int c2d_malloc_host(c2d_ctx* ctx, void** h_ptr, size_t bytes)
{
    if (!ctx || !h_ptr) return C2D_ERR_INVALID_ARG;
    *h_ptr = nullptr;
    if (bytes == 0) return C2D_OK;
    c2d::DeviceGuard g(ctx->device);
    hipError_t e = hipHostMalloc(h_ptr, bytes, hipHostMallocDefault);
    if (e == hipErrorOutOfMemory) { ctx->last_error = "hipHostMalloc: out of memory"; return C2D_ERR_NOMEM; }
    if (e != hipSuccess) return c2d::fail_hip(ctx, e, "hipHostMalloc", __FILE__, __LINE__);
    return C2D_OK;
}

int c2d_free_host(c2d_ctx* ctx, void* h_ptr)
{
    if (!ctx) return C2D_ERR_INVALID_ARG;
    if (!h_ptr) return C2D_OK;
    c2d::DeviceGuard g(ctx->device);
    C2D_HIP(ctx, hipHostFree(h_ptr));
    return C2D_OK;
}

int c2d_memset(c2d_ctx* ctx, void* d_ptr, int value, size_t bytes, c2d_stream stream)
{
    if (!ctx || (!d_ptr && bytes)) return C2D_ERR_INVALID_ARG;
    if (!bytes) return C2D_OK;
    c2d::DeviceGuard g(ctx->device);
    C2D_HIP(ctx, hipMemsetAsync(d_ptr, value, bytes, (hipStream_t)stream));
    return C2D_OK;
}

int c2d_memcpy_h2d(c2d_ctx* ctx, void* d_dst, const void* h_src, size_t bytes, c2d_stream stream)
{
    if (!ctx || ((!d_dst || !h_src) && bytes)) return C2D_ERR_INVALID_ARG;
    if (!bytes) return C2D_OK;
    c2d::DeviceGuard g(ctx->device);
    C2D_HIP(ctx, hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, (hipStream_t)stream));
    return C2D_OK;
}

int c2d_memcpy_d2h(c2d_ctx* ctx, void* h_dst, const void* d_src, size_t bytes, c2d_stream stream)
{
    if (!ctx || ((!h_dst || !d_src) && bytes)) return C2D_ERR_INVALID_ARG;
    if (!bytes) return C2D_OK;
    c2d::DeviceGuard g(ctx->device);
    C2D_HIP(ctx, hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, (hipStream_t)stream));
    return C2D_OK;
}

int c2d_stream_create(c2d_ctx* ctx, c2d_stream* out)
{
    if (!ctx || !out) return C2D_ERR_INVALID_ARG;
    c2d::DeviceGuard g(ctx->device);
    hipStream_t s = nullptr;
    C2D_HIP(ctx, hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    *out = (c2d_stream)s;
    return C2D_OK;
}

int c2d_stream_destroy(c2d_ctx* ctx, c2d_stream stream)
{
    if (!ctx) return C2D_ERR_INVALID_ARG;
    if (!stream) return C2D_OK;
    c2d::DeviceGuard g(ctx->device);
    // The guard never hands a remembered stream to the runtime, and destroying a stream with calls queued is legal: they finish.
    // But its ADDRESS may be given to the next stream created, and then must not pass for the stream of the outstanding tickets.
    if (ctx->ws_outstanding && ctx->ws_stream == (hipStream_t)stream) ctx->ws_stream = c2d::forgotten_stream();
    C2D_HIP(ctx, hipStreamDestroy((hipStream_t)stream));
    return C2D_OK;
}

int c2d_stream_synchronize(c2d_ctx* ctx, c2d_stream stream)
{
    if (!ctx) return C2D_ERR_INVALID_ARG;
    c2d::DeviceGuard g(ctx->device);
    C2D_HIP(ctx, hipStreamSynchronize((hipStream_t)stream));
    c2d::workspace_stream_drained(ctx, (hipStream_t)stream);
    return c2d_ctx_check_async(ctx);
}

int c2d_ctx_check_async(c2d_ctx* ctx)
{
    if (!ctx) return C2D_ERR_INVALID_ARG;
    const uint32_t e = __atomic_exchange_n(ctx->h_async_err, 0u, __ATOMIC_ACQ_REL);
    if (e == 0) return C2D_OK;
    if (e & C2D_ASYNC_ERR_POLY_K)
        ctx->last_error = "c2d_sat_poly_pairs: vertex count outside 1..C2D_POLY_KMAX (reported asynchronously; those pairs were written as 0)";
    else
        ctx->last_error = "asynchronous argument error reported by a kernel";
    return C2D_ERR_INVALID_ARG;
}

}  // extern "C"
