// c2d_dist.hip — multi-GPU aggregation of the hit counters (include/c2d.h, "multi-GPU" block).
//
// The reference is single-GPU (compute_collision_probability.cu:212-251); BASELINE's north star
// shards pairs / scenes / MC sample ranges over the 8 GPUs of a node — no exchange on the data
// path — and closes with ONE sum-reduction of a few 64-bit counters.  That reduction is RCCL
// (ncclAllReduce, uint64, sum) over xGMI; the payload is 8..64 bytes, so it is latency-bound
// and link bandwidth is irrelevant.
//
// One process per GPU.  librccl.so.1 (573 MB) is dlopen'ed at the first c2d_dist_* call, so a
// single-GPU run never loads it; inside a PyTorch process the already-loaded librccl of the same
// soname is reused.  The 128-byte ncclUniqueId travels by whatever channel the caller has
// (c2d_dist_init) or through a file (c2d_dist_init_file: rank 0 writes it atomically, the others wait).
//
// Rehearsal transport: RCCL refuses two ranks on one device ("Duplicate GPU detected"), so the N > 1 host logic of
// the drivers and of bench.py cannot be exercised with RCCL on a one-GPU box.  A file-based sum (host copies of the
// counters exchanged through small files) that lets several ranks share a device exists for the tests — in a SEPARATE
// build of this library only: `make lib-rehearsal` compiles this file with -DC2D_DIST_REHEARSAL into
// lib-rehearsal/libc2d.so, which the tests put in front of the product library (LD_LIBRARY_PATH / C2D_LIBRARY).  The
// product libc2d.so contains none of it: no environment variable can reroute its reduce away from RCCL.
//
// Watchdog: ncclCommInitRank and the first collective block until every rank has arrived.  Both run on a helper thread
// here and the caller waits with a deadline (timeout_s of c2d_dist_init_file, $C2D_DIST_TIMEOUT_S or 300 s for
// c2d_dist_init): a rank whose peers never show up gets C2D_ERR_DIST instead of hanging forever.  The helper thread is
// left behind inside RCCL in that case — the process is expected to report the error and end WITHOUT running exit handlers
// (the drivers _exit: the HIP / ROCr teardown of a normal exit would race the abandoned thread).
#include <dlfcn.h>
#include <fcntl.h>
#include <rccl/rccl.h>
#include <sys/stat.h>
#include <unistd.h>

#include <chrono>
#include <cstdlib>
#include <thread>
#include <vector>

#include "c2d_internal.hpp"
#include "c2d_watchdog.hpp"

static_assert(sizeof(ncclUniqueId) == C2D_DIST_ID_BYTES, "C2D_DIST_ID_BYTES must match ncclUniqueId");

namespace {

struct Rccl {
    void* handle = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclBroadcast) Broadcast = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    decltype(&ncclCommCount) CommCount = nullptr;
    decltype(&ncclGetVersion) GetVersion = nullptr;
    std::string error;
};

Rccl load_rccl()
{
    Rccl r;
    for (const char* name : {"librccl.so.1", "librccl.so"}) {
        r.handle = dlopen(name, RTLD_NOW | RTLD_LOCAL);
        if (r.handle) break;
    }
    if (!r.handle) {
        r.error = std::string("cannot load librccl.so.1: ") + dlerror();
        return r;
    }
    auto sym = [&](const char* n) {
        void* p = dlsym(r.handle, n);
        if (!p && r.error.empty()) r.error = std::string("librccl lacks ") + n;
        return p;
    };
    r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(sym("ncclGetUniqueId"));
    r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(sym("ncclCommInitRank"));
    r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(sym("ncclCommDestroy"));
    r.AllReduce = reinterpret_cast<decltype(r.AllReduce)>(sym("ncclAllReduce"));
    r.Broadcast = reinterpret_cast<decltype(r.Broadcast)>(sym("ncclBroadcast"));
    r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(sym("ncclGetErrorString"));
    r.CommCount = reinterpret_cast<decltype(r.CommCount)>(sym("ncclCommCount"));
    r.GetVersion = reinterpret_cast<decltype(r.GetVersion)>(sym("ncclGetVersion"));
    if (!r.error.empty()) {
        dlclose(r.handle);
        r.handle = nullptr;
    }
    return r;
}

// loaded once, at the first c2d_dist_* call that needs it (initialisation of a local static is thread-safe)
Rccl& rccl()
{
    static Rccl r = load_rccl();
    return r;
}

#ifdef C2D_DIST_REHEARSAL
const char kFileMagic[] = "c2d-file-transport:";
#endif

bool write_file_atomically(const std::string& path, const void* data, size_t bytes)
{
    const std::string tmp = path + ".tmp." + std::to_string((long long)getpid());
    FILE* f = std::fopen(tmp.c_str(), "wb");
    if (!f) return false;
    const bool ok = std::fwrite(data, 1, bytes, f) == bytes;
    if (std::fclose(f) != 0 || !ok) { std::remove(tmp.c_str()); return false; }
    if (std::rename(tmp.c_str(), path.c_str()) != 0) { std::remove(tmp.c_str()); return false; }
    return true;
}

// waits until `path` exists with exactly `bytes` bytes and reads it
bool read_file_when_complete(const std::string& path, void* data, size_t bytes, double timeout_s)
{
    const auto t0 = std::chrono::steady_clock::now();
    for (;;) {
        struct stat st;
        if (stat(path.c_str(), &st) == 0 && (size_t)st.st_size == bytes) {
            FILE* f = std::fopen(path.c_str(), "rb");
            if (f) {
                const bool ok = std::fread(data, 1, bytes, f) == bytes;
                std::fclose(f);
                if (ok) return true;
            }
        }
        if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > timeout_s) return false;
        std::this_thread::sleep_for(std::chrono::microseconds(300));
    }
}

using c2d::watchdog::default_timeout_s;   // the deadline mechanics live in c2d_watchdog.hpp (thread-sanitised on the CPU)
using c2d::watchdog::run_job;

}  // namespace

struct c2d_dist {
    c2d_ctx* ctx = nullptr;
    int rank = 0, world = 1;
    ncclComm_t comm = nullptr;
    double timeout_s = 300.0;
    bool timed_out = false; // a watched call ran out of time: a helper thread is still inside RCCL / HIP with this communicator,
                            // which must neither be used nor destroyed any more
#ifdef C2D_DIST_REHEARSAL
    std::string base;      // file transport: path prefix of the exchange files
    uint64_t seq = 0;      // file transport: collective sequence number
#endif
};

namespace {

int fail_dist(c2d_ctx* ctx, const std::string& msg, int code = C2D_ERR_DIST)
{
    if (ctx) ctx->last_error = msg;
    return code;
}

#ifdef C2D_DIST_REHEARSAL
// file transport: sum of `count` host words over all ranks, in place
int file_all_reduce(c2d_dist* d, unsigned long long* h, size_t count)
{
    const size_t bytes = count * sizeof(unsigned long long);
    auto name = [&](uint64_t seq, int rank) { return d->base + "." + std::to_string(seq) + "." + std::to_string(rank); };
    if (!write_file_atomically(name(d->seq, d->rank), h, bytes)) return fail_dist(d->ctx, "file transport: cannot write " + name(d->seq, d->rank));
    std::vector<unsigned long long> other(count);
    for (int r = 0; r < d->world; r++) {
        if (r == d->rank) continue;
        if (!read_file_when_complete(name(d->seq, r), other.data(), bytes, d->timeout_s))
            return fail_dist(d->ctx, "file transport: timed out waiting for rank " + std::to_string(r) + " (payload sizes must agree on all ranks)");
        for (size_t i = 0; i < count; i++) h[i] += other[i];
    }
    // A rank that writes sequence s has read every file of s - 1, so every rank has written s - 1 and hence
    // finished reading s - 2: this rank's file of s - 2 has no reader left.
    if (d->seq >= 2) std::remove(name(d->seq - 2, d->rank).c_str());
    d->seq++;
    return C2D_OK;
}
#endif

}  // namespace

extern "C" {

int c2d_dist_unique_id(void* id_out)
{
    if (!id_out) return C2D_ERR_INVALID_ARG;
    std::memset(id_out, 0, C2D_DIST_ID_BYTES);
#ifdef C2D_DIST_REHEARSAL
    const char* dir = std::getenv("TMPDIR");
    const auto now = std::chrono::steady_clock::now().time_since_epoch().count();
    std::snprintf(static_cast<char*>(id_out), C2D_DIST_ID_BYTES, "%s%s/c2d_dist_%lld_%lld", kFileMagic, dir && *dir ? dir : "/tmp",
                  (long long)getpid(), (long long)now);
    return C2D_OK;
#else
    Rccl& R = rccl();
    if (!R.handle) return C2D_ERR_DIST;
    ncclUniqueId id;
    if (R.GetUniqueId(&id) != ncclSuccess) return C2D_ERR_DIST;
    std::memcpy(id_out, &id, sizeof id);
    return C2D_OK;
#endif
}

// the communicator, created under the watchdog; timeout_s <= 0: $C2D_DIST_TIMEOUT_S or 300 s
static int dist_init_impl(c2d_ctx* ctx, int rank, int world_size, const void* id, double timeout_s, c2d_dist** out)
{
    if (!ctx || !out || !id) return C2D_ERR_INVALID_ARG;
    *out = nullptr;
    if (world_size < 1 || rank < 0 || rank >= world_size) return c2d::fail_arg(ctx, "c2d_dist_init: rank / world_size out of range");
    if (timeout_s <= 0) timeout_s = default_timeout_s();
    c2d_dist* d = new (std::nothrow) c2d_dist();
    if (!d) return C2D_ERR_NOMEM;
    d->ctx = ctx;
    d->rank = rank;
    d->world = world_size;
    d->timeout_s = timeout_s;
#ifdef C2D_DIST_REHEARSAL
    const char* idc = static_cast<const char*>(id);
    const size_t ml = sizeof(kFileMagic) - 1;
    if (std::strncmp(idc, kFileMagic, ml) != 0) { delete d; return fail_dist(ctx, "rehearsal build: the id was not made by this build's c2d_dist_unique_id"); }
    d->base = std::string(idc + ml, strnlen(idc + ml, C2D_DIST_ID_BYTES - ml));
    *out = d;
    return C2D_OK;
#else
    Rccl& R = rccl();
    if (!R.handle) { delete d; return fail_dist(ctx, R.error); }
    ncclUniqueId uid;
    std::memcpy(&uid, id, sizeof uid);
    const int device = ctx->device;
    // ncclCommInitRank returns once every rank has joined; a peer that never arrives would block it for ever.  The helper writes
    // the communicator into ITS job block, not into `d`: after a time-out nothing of this call is shared with it any more.
    struct InitJob : c2d::watchdog::Job { ncclComm_t comm = nullptr; };
    InitJob res;
    const bool in_time = run_job<InitJob>([uid, world_size, rank, device](InitJob& job) {
        if (hipSetDevice(device) != hipSuccess) { job.st = C2D_ERR_HIP; job.error = "hipSetDevice failed on the communicator thread"; return; }
        const ncclResult_t st = rccl().CommInitRank(&job.comm, world_size, uid, rank);
        if (st != ncclSuccess) { job.st = C2D_ERR_DIST; job.error = std::string("ncclCommInitRank failed: ") + rccl().GetErrorString(st); }
    }, timeout_s, &res);
    if (!in_time) {
        delete d;
        char msg[160];
        std::snprintf(msg, sizeof msg, "rank %d: ncclCommInitRank did not complete within %.0f s (are all %d ranks running?)", rank, timeout_s, world_size);
        return fail_dist(ctx, msg);
    }
    if (res.st != C2D_OK) {
        delete d;
        return fail_dist(ctx, res.error, res.st);
    }
    d->comm = res.comm;
    *out = d;
    return C2D_OK;
#endif
}

int c2d_dist_init(c2d_ctx* ctx, int rank, int world_size, const void* id, c2d_dist** out)
{
    return dist_init_impl(ctx, rank, world_size, id, 0.0, out);
}

int c2d_dist_init_file(c2d_ctx* ctx, int rank, int world_size, const char* path, double timeout_s, c2d_dist** out)
{
    if (!ctx || !out || !path || !*path) return C2D_ERR_INVALID_ARG;
    *out = nullptr;
    if (world_size < 1 || rank < 0 || rank >= world_size) return c2d::fail_arg(ctx, "c2d_dist_init_file: rank / world_size out of range");
    if (timeout_s <= 0) timeout_s = default_timeout_s();
    unsigned char id[C2D_DIST_ID_BYTES];
    if (rank == 0) {
        int st = c2d_dist_unique_id(id);
        if (st != C2D_OK) return fail_dist(ctx, rccl().error.empty() ? "ncclGetUniqueId failed" : rccl().error);
        if (!write_file_atomically(path, id, sizeof id)) return fail_dist(ctx, std::string("cannot write the id file ") + path);
    } else if (!read_file_when_complete(path, id, sizeof id, timeout_s)) {
        return fail_dist(ctx, std::string("timed out waiting for rank 0 to write the id file ") + path);
    }
    int st = dist_init_impl(ctx, rank, world_size, id, timeout_s, out);
    if (st != C2D_OK) return st;
    // every rank has read the id once the first collective completes: rank 0 then removes the file
    st = c2d_dist_barrier(*out, nullptr);
    if (st != C2D_OK) {
        if (!(*out)->timed_out) c2d_dist_destroy(*out);  // (after a watchdog time-out the communicator is left to its thread; any other failure frees it)
        *out = nullptr;
        return st;
    }
    if (rank == 0) std::remove(path);
    return C2D_OK;
}

int c2d_dist_rank(const c2d_dist* d) { return d ? d->rank : -1; }

int c2d_dist_world_size(const c2d_dist* d)
{
    if (!d) return -1;
#ifdef C2D_DIST_REHEARSAL
    return d->world;
#else
    int n = -1;  // what RCCL itself says about the communicator
    if (rccl().CommCount(d->comm, &n) != ncclSuccess) return -1;
    return n;
#endif
}

const char* c2d_dist_transport(const c2d_dist* d)
{
#ifdef C2D_DIST_REHEARSAL
    return !d ? "" : "file (rehearsal)";
#else
    return !d ? "" : "rccl";
#endif
}

int c2d_dist_rccl_version(int* version, char* path_out, size_t path_bytes)
{
    if (!version) return C2D_ERR_INVALID_ARG;
    *version = 0;
    if (path_out && path_bytes) path_out[0] = 0;
#ifdef C2D_DIST_REHEARSAL
    if (path_out && path_bytes) std::snprintf(path_out, path_bytes, "file (rehearsal)");
    return C2D_OK;
#else
    Rccl& R = rccl();
    if (!R.handle) return C2D_ERR_DIST;
    if (R.GetVersion(version) != ncclSuccess) return C2D_ERR_DIST;
    Dl_info info;
    if (path_out && path_bytes && dladdr(reinterpret_cast<void*>(R.GetVersion), &info) && info.dli_fname)
        std::snprintf(path_out, path_bytes, "%s", info.dli_fname);
    return C2D_OK;
#endif
}

int c2d_dist_all_reduce_sum_u64(c2d_dist* d, unsigned long long* d_buf, size_t count, c2d_stream stream)
{
    if (!d || (!d_buf && count)) return C2D_ERR_INVALID_ARG;
    if (count == 0) return C2D_OK;
    c2d::DeviceGuard g(d->ctx->device);
    hipStream_t s = (hipStream_t)stream;
#ifdef C2D_DIST_REHEARSAL
    std::vector<unsigned long long> h(count);
    C2D_HIP(d->ctx, hipMemcpyAsync(h.data(), d_buf, count * sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
    C2D_HIP(d->ctx, hipStreamSynchronize(s));
    if (int st = file_all_reduce(d, h.data(), count)) return st;
    C2D_HIP(d->ctx, hipMemcpyAsync(d_buf, h.data(), count * sizeof(unsigned long long), hipMemcpyHostToDevice, s));
    C2D_HIP(d->ctx, hipStreamSynchronize(s));
    return C2D_OK;
#else
    const ncclResult_t st = rccl().AllReduce(d_buf, d_buf, count, ncclUint64, ncclSum, d->comm, s);
    if (st != ncclSuccess) return fail_dist(d->ctx, std::string("ncclAllReduce failed: ") + rccl().GetErrorString(st));
    return C2D_OK;
#endif
}

int c2d_dist_broadcast_u64(c2d_dist* d, unsigned long long* d_buf, size_t count, int root, c2d_stream stream)
{
    if (!d || (!d_buf && count) || root < 0 || root >= d->world) return C2D_ERR_INVALID_ARG;
    if (count == 0) return C2D_OK;
    c2d::DeviceGuard g(d->ctx->device);
    hipStream_t s = (hipStream_t)stream;
#ifdef C2D_DIST_REHEARSAL  // sum with zeros from everyone but the root
    if (d->rank != root) C2D_HIP(d->ctx, hipMemsetAsync(d_buf, 0, count * sizeof(unsigned long long), s));
    return c2d_dist_all_reduce_sum_u64(d, d_buf, count, stream);
#else
    const ncclResult_t st = rccl().Broadcast(d_buf, d_buf, count, ncclUint64, root, d->comm, s);
    if (st != ncclSuccess) return fail_dist(d->ctx, std::string("ncclBroadcast failed: ") + rccl().GetErrorString(st));
    return C2D_OK;
#endif
}

// One-word all-reduce + stream synchronise, under the watchdog: a collective that a peer never joins would otherwise
// hold hipStreamSynchronize for ever.  The helper thread touches nothing of the ctx: its status and error text live in a
// block it co-owns, and the caller copies them into the ctx only when the helper finished in time (after a time-out the
// helper may still be running while the caller reports, or after the caller has destroyed the ctx).
int c2d_dist_barrier(c2d_dist* d, c2d_stream stream)
{
    if (!d) return C2D_ERR_INVALID_ARG;
    if (d->timed_out) return fail_dist(d->ctx, "this communicator was abandoned after a time-out");
    const int device = d->ctx->device;
    // The helper keeps `d` (left alone by c2d_dist_destroy after a time-out) and the communicator inside it, nothing of the ctx.
    c2d::watchdog::Job res;
#ifdef C2D_DIST_REHEARSAL
    c2d_ctx* ctx = d->ctx;
    const double limit = d->timeout_s + 10.0;  // (the file transport has this limit built in: let it report its own time-out)
#else
    const double limit = d->timeout_s;
#endif
    const bool in_time = run_job<c2d::watchdog::Job>([=](c2d::watchdog::Job& job) {
        if (hipSetDevice(device) != hipSuccess) { job.st = C2D_ERR_HIP; job.error = "hipSetDevice failed on the barrier thread"; return; }
        unsigned long long* w = nullptr;
        if (hipMalloc(&w, sizeof *w) != hipSuccess) { job.st = C2D_ERR_NOMEM; job.error = "barrier word allocation failed"; return; }
        if (hipMemsetAsync(w, 0, sizeof *w, (hipStream_t)stream) != hipSuccess) { job.st = C2D_ERR_HIP; job.error = "hipMemsetAsync failed in the barrier"; }
        if (job.st == C2D_OK) {
#ifdef C2D_DIST_REHEARSAL
            job.st = c2d_dist_all_reduce_sum_u64(d, w, 1, stream);  // (the file transport runs on the caller's side of the deadline anyway)
            if (job.st != C2D_OK) job.error = ctx->last_error;
#else
            const ncclResult_t st = rccl().AllReduce(w, w, 1, ncclUint64, ncclSum, d->comm, (hipStream_t)stream);
            if (st != ncclSuccess) { job.st = C2D_ERR_DIST; job.error = std::string("ncclAllReduce failed: ") + rccl().GetErrorString(st); }
#endif
        }
        if (job.st == C2D_OK && hipStreamSynchronize((hipStream_t)stream) != hipSuccess) { job.st = C2D_ERR_HIP; job.error = "hipStreamSynchronize failed in the barrier"; }
        (void)hipFree(w);
    }, limit, &res);
    if (!in_time) {
        d->timed_out = true;
        char msg[160];
        std::snprintf(msg, sizeof msg, "rank %d: barrier did not complete within %.0f s (a peer is missing or stuck)", d->rank, d->timeout_s);
        return fail_dist(d->ctx, msg);
    }
    if (res.st != C2D_OK && !res.error.empty()) d->ctx->last_error = res.error;
    return res.st;
}

// hipStreamSynchronize under the same watchdog, for the collectives a caller has queued itself (c2d_dist_all_reduce_sum_u64 /
// c2d_dist_broadcast_u64 are asynchronous): a peer that died after the communicator was built would otherwise hold the wait
// for ever.
int c2d_dist_stream_synchronize(c2d_dist* d, c2d_stream stream)
{
    if (!d) return C2D_ERR_INVALID_ARG;
    if (d->timed_out) return fail_dist(d->ctx, "this communicator was abandoned after a time-out");
    const int device = d->ctx->device;
    c2d::watchdog::Job res;
    const bool in_time = run_job<c2d::watchdog::Job>([=](c2d::watchdog::Job& job) {
        if (hipSetDevice(device) != hipSuccess || hipStreamSynchronize((hipStream_t)stream) != hipSuccess) job.st = C2D_ERR_HIP;
    }, d->timeout_s, &res);
    if (!in_time) {
        d->timed_out = true;
        char msg[160];
        std::snprintf(msg, sizeof msg, "rank %d: a collective did not complete within %.0f s (a peer is missing or stuck)", d->rank, d->timeout_s);
        return fail_dist(d->ctx, msg);
    }
    if (res.st != C2D_OK) return fail_dist(d->ctx, "hipStreamSynchronize failed behind a collective", C2D_ERR_HIP);
    return c2d_ctx_check_async(d->ctx);
}

int c2d_dist_timed_out(const c2d_dist* d) { return d && d->timed_out ? 1 : 0; }

#ifdef C2D_DIST_REHEARSAL
// Test hook of the rehearsal build only (not in include/c2d.h, not in the product library): sets the workspace guard's ticket
// counter, so that tests/test_gpu_workspace_guard.py can walk it through the 32-bit wrap that otherwise comes after 2^32 launches.
int c2d_test_set_workspace_ticket(c2d_ctx* ctx, unsigned value)
{
    if (!ctx) return C2D_ERR_INVALID_ARG;
    ctx->ws_ticket = value;
    return C2D_OK;
}
#endif

int c2d_dist_destroy(c2d_dist* d)
{
    if (!d) return C2D_OK;
    if (d->timed_out) return C2D_OK;  // a helper thread still holds it: left alone (the process is about to end)
#ifdef C2D_DIST_REHEARSAL
    // own exchange files of the last two sequence numbers may remain; the peers are past reading them
    // once they have entered destroy too, which a final barrier establishes
    if (d->world > 1) {
        unsigned long long z = 0;
        if (d->timeout_s > 10.0) d->timeout_s = 10.0;  // a peer that died must not hold this rank's exit for minutes
        (void)file_all_reduce(d, &z, 1);
    }
    for (uint64_t s = d->seq >= 3 ? d->seq - 3 : 0; s < d->seq; s++)
        if (s + 1 < d->seq || d->world == 1) std::remove((d->base + "." + std::to_string(s) + "." + std::to_string(d->rank)).c_str());
    // the file of the final barrier itself is left for the slowest reader; it is a few bytes in TMPDIR
#else
    if (d->comm) {
        c2d::DeviceGuard g(d->ctx->device);
        (void)rccl().CommDestroy(d->comm);
    }
#endif
    delete d;
    return C2D_OK;
}

}  // extern "C"
