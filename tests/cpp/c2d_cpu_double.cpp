// c2d_cpu_double.cpp — TEST DOUBLE of the part of include/c2d.h that the three CLI drivers call.
//
// TEST INFRASTRUCTURE ONLY.  It exists so that the drivers' HOST logic — flag handling, table generation and its `.npy` files,
// batch numbering, scene-id bases, the deal of batches over ranks, shuffles, the aggregated summary — runs end to end in the CPU
// suite (`-m "not gpu"`, tests/test_drivers.py) and under the sanitizers, on a machine without a GPU.  "Device" memory is host
// memory, streams do nothing, the Monte-Carlo work is handed to oracle/libc2d_oracle.so, the tables are drawn by the reference's own
// serial loop (std::default_random_engine, generate_dataset.cu:279-332), the multi-rank sum goes through small files.  Nothing in the
// product builds, links, loads or can select this file: the drivers under test are compiled against it in a temporary directory by
// the test itself.  A run of the double proves nothing about the HIP kernels — the `-m gpu` tests do that, through libc2d.so.
#include <sys/stat.h>
#include <unistd.h>

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <thread>
#include <vector>

#include "c2d.h"

extern "C" {
unsigned long long c2d_oracle_mc_pair(float robot_w, float robot_h, const Position* pos, const Pose* pose, const StdDev* sd, uint64_t seed,
                                      uint64_t scene_id, uint64_t sample_begin, uint64_t n_samples);
unsigned long long c2d_oracle_mc_scenes(const Pose* poses, uint32_t num_poses, const StdDev* std_devs, uint32_t num_std_devs,
                                        const PositionWithVarAndPoseIdx* scenes, size_t n_scenes, float robot_w, float robot_h,
                                        const float* accuracy_bins, const float* bin_accuracy, uint32_t n_accuracy_bins, uint32_t max_samples,
                                        uint64_t seed, uint64_t scene_id_base, uint32_t small_batch, uint32_t large_batch, uint32_t switch_at,
                                        uint32_t* hits_out, uint32_t* n_used_out, PoseCPVarAndPoseIdx* rows);
void c2d_oracle_sample_scenes(const Pose* poses, uint32_t num_poses, const StdDev* std_devs, uint32_t num_std_devs, float robot_w, float robot_h,
                              float spread, uint64_t seed, uint64_t scene_id_base, size_t n_scenes, PositionWithVarAndPoseIdx* scenes);
}

struct c2d_ctx {
    int device = 0;
    std::string last_error;
};

struct c2d_dist {
    c2d_ctx* ctx = nullptr;
    int rank = 0, world = 1;
    std::string base;
    unsigned long long seq = 0;
    double timeout_s = 60.0;
};

namespace {

int devices()
{
    const char* v = std::getenv("C2D_DOUBLE_DEVICES");   // how many "GPUs" the double shows (default 8: `--gpus N` needs N devices)
    return v ? std::atoi(v) : 8;
}

bool write_atomically(const std::string& path, const void* data, size_t bytes)
{
    const std::string tmp = path + ".tmp." + std::to_string((long long)getpid());
    FILE* f = std::fopen(tmp.c_str(), "wb");
    if (!f) return false;
    const bool ok = std::fwrite(data, 1, bytes, f) == bytes;
    if (std::fclose(f) != 0 || !ok) { std::remove(tmp.c_str()); return false; }
    return std::rename(tmp.c_str(), path.c_str()) == 0;
}

bool read_when_complete(const std::string& path, void* data, size_t bytes, double timeout_s)
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
        std::this_thread::sleep_for(std::chrono::microseconds(500));
    }
}

int sum_over_ranks(c2d_dist* d, unsigned long long* h, size_t count)
{
    auto name = [&](unsigned long long seq, int rank) { return d->base + "." + std::to_string(seq) + "." + std::to_string(rank); };
    if (!write_atomically(name(d->seq, d->rank), h, count * sizeof *h)) { d->ctx->last_error = "cpu double: cannot write " + name(d->seq, d->rank); return C2D_ERR_DIST; }
    std::vector<unsigned long long> other(count);
    for (int r = 0; r < d->world; r++) {
        if (r == d->rank) continue;
        if (!read_when_complete(name(d->seq, r), other.data(), count * sizeof *h, d->timeout_s)) { d->ctx->last_error = "cpu double: timed out waiting for rank " + std::to_string(r); return C2D_ERR_DIST; }
        for (size_t i = 0; i < count; i++) h[i] += other[i];
    }
    if (d->seq >= 2) std::remove(name(d->seq - 2, d->rank).c_str());
    d->seq++;
    return C2D_OK;
}

}  // namespace

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

int c2d_ctx_create(int device, c2d_ctx** out)
{
    if (!out) return C2D_ERR_INVALID_ARG;
    *out = nullptr;
    if (device < 0 || device >= devices()) return C2D_ERR_NO_DEVICE;
    *out = new c2d_ctx();
    (*out)->device = device;
    return C2D_OK;
}

int c2d_ctx_destroy(c2d_ctx* ctx) { delete ctx; return C2D_OK; }

int c2d_ctx_info_sized(const c2d_ctx* ctx, c2d_device_info* out, size_t out_bytes)
{
    if (!ctx || !out || out_bytes == 0) return C2D_ERR_INVALID_ARG;
    c2d_device_info di;
    std::memset(&di, 0, sizeof di);
    std::snprintf(di.name, sizeof di.name, "cpu double (tests only)");
    std::snprintf(di.arch, sizeof di.arch, "none");
    di.device = ctx->device;
    std::memcpy(out, &di, out_bytes < sizeof di ? out_bytes : sizeof di);
    return C2D_OK;
}

int (c2d_ctx_info)(const c2d_ctx* ctx, c2d_device_info* out) { return c2d_ctx_info_sized(ctx, out, C2D_DEVICE_INFO_BYTES_0_4); }

int c2d_malloc(c2d_ctx* ctx, void** p, size_t bytes)
{
    if (!ctx || !p) return C2D_ERR_INVALID_ARG;
    *p = bytes ? std::malloc(bytes) : nullptr;
    return bytes && !*p ? C2D_ERR_NOMEM : C2D_OK;
}
int c2d_free(c2d_ctx* ctx, void* p) { if (!ctx) return C2D_ERR_INVALID_ARG; std::free(p); return C2D_OK; }
int c2d_malloc_host(c2d_ctx* ctx, void** p, size_t bytes) { return c2d_malloc(ctx, p, bytes); }
int c2d_free_host(c2d_ctx* ctx, void* p) { return c2d_free(ctx, p); }
int c2d_memset(c2d_ctx* ctx, void* p, int v, size_t bytes, c2d_stream) { if (!ctx || (!p && bytes)) return C2D_ERR_INVALID_ARG; if (bytes) std::memset(p, v, bytes); return C2D_OK; }
int c2d_memcpy_h2d(c2d_ctx* ctx, void* d, const void* h, size_t bytes, c2d_stream) { if (!ctx || ((!d || !h) && bytes)) return C2D_ERR_INVALID_ARG; if (bytes) std::memcpy(d, h, bytes); return C2D_OK; }
int c2d_memcpy_d2h(c2d_ctx* ctx, void* h, const void* d, size_t bytes, c2d_stream) { if (!ctx || ((!d || !h) && bytes)) return C2D_ERR_INVALID_ARG; if (bytes) std::memcpy(h, d, bytes); return C2D_OK; }
int c2d_stream_create(c2d_ctx* ctx, c2d_stream* out) { if (!ctx || !out) return C2D_ERR_INVALID_ARG; *out = new int(0); return C2D_OK; }
int c2d_stream_destroy(c2d_ctx* ctx, c2d_stream s) { if (!ctx) return C2D_ERR_INVALID_ARG; delete static_cast<int*>(s); return C2D_OK; }
int c2d_stream_synchronize(c2d_ctx* ctx, c2d_stream) { return ctx ? C2D_OK : C2D_ERR_INVALID_ARG; }

int c2d_mc_pair(c2d_ctx* ctx, float robot_w, float robot_h, const Position* pos, const Pose* pose, const StdDev* sd, uint64_t seed, uint64_t scene_id,
                uint64_t sample_begin, uint64_t n_samples, unsigned long long* d_hits, c2d_stream)
{
    if (!ctx || !pos || !pose || !sd || !d_hits) return C2D_ERR_INVALID_ARG;
    *d_hits += c2d_oracle_mc_pair(robot_w, robot_h, pos, pose, sd, seed, scene_id, sample_begin, n_samples);
    return C2D_OK;
}

int c2d_mc_scenes(c2d_ctx* ctx, const c2d_mc_scenes_args* a, c2d_stream)
{
    if (!ctx || !a) return C2D_ERR_INVALID_ARG;
    if (a->total_samples) *a->total_samples = 0;
    if (a->iterations) *a->iterations = 0;
    if (a->n_scenes == 0) return C2D_OK;
    if (!a->d_poses || !a->d_std_devs || !a->d_scenes || !a->d_hits || !a->d_n_used || !a->accuracy_bins || !a->bin_accuracy || a->max_samples == 0) {
        ctx->last_error = "c2d_mc_scenes: NULL argument";
        return C2D_ERR_INVALID_ARG;
    }
    const unsigned long long total = c2d_oracle_mc_scenes(a->d_poses, a->num_poses, a->d_std_devs, a->num_std_devs, a->d_scenes, a->n_scenes, a->robot_w, a->robot_h,
                                                          a->accuracy_bins, a->bin_accuracy, a->n_accuracy_bins, a->max_samples, a->seed, a->scene_id_base,
                                                          a->schedule_small_batch, a->schedule_large_batch, a->schedule_switch_at, a->d_hits, a->d_n_used, a->d_rows);
    if (a->total_samples) *a->total_samples = total;
    return C2D_OK;
}

int c2d_sample_scenes(c2d_ctx* ctx, const Pose* poses, uint32_t num_poses, const StdDev* sds, uint32_t num_sds, float robot_w, float robot_h, float spread,
                      uint64_t seed, uint64_t scene_id_base, size_t n_scenes, PositionWithVarAndPoseIdx* scenes, c2d_stream)
{
    if (!ctx || !poses || !sds || (!scenes && n_scenes) || !num_poses || !num_sds) return C2D_ERR_INVALID_ARG;
    c2d_oracle_sample_scenes(poses, num_poses, sds, num_sds, robot_w, robot_h, spread, seed, scene_id_base, n_scenes, scenes);
    return C2D_OK;
}

// the reference's own loop (generate_dataset.cu:279-332): one default-seeded std::default_random_engine, one
// uniform_real_distribution<float> per dimension, row by row
int c2d_uniform_table_minstd(c2d_ctx* ctx, float* out, size_t rows, int dims, const float* lo, const float* hi, uint64_t first_draw, c2d_stream)
{
    if (!ctx || (!out && rows) || dims < 1 || dims > 8 || !lo || !hi) return C2D_ERR_INVALID_ARG;
    std::default_random_engine gen;
    gen.discard(first_draw);
    std::vector<std::uniform_real_distribution<float>> dist;
    for (int k = 0; k < dims; k++) dist.emplace_back(lo[k], hi[k]);
    for (size_t i = 0; i < rows; i++)
        for (int k = 0; k < dims; k++) out[i * dims + k] = dist[k](gen);
    return C2D_OK;
}

int c2d_sqrt_f32(c2d_ctx* ctx, const float* in, float* out, size_t n, c2d_stream)
{
    if (!ctx || ((!in || !out) && n)) return C2D_ERR_INVALID_ARG;
    for (size_t i = 0; i < n; i++) out[i] = std::sqrt(in[i]);
    return C2D_OK;
}

// ---- the multi-rank sum, through files (the shape of the rehearsal build's transport) ----------------------------------------
int c2d_dist_init_file(c2d_ctx* ctx, int rank, int world, const char* path, double timeout_s, c2d_dist** out)
{
    if (!ctx || !out || !path || !*path || world < 1 || rank < 0 || rank >= world) return C2D_ERR_INVALID_ARG;
    *out = nullptr;
    char id[C2D_DIST_ID_BYTES] = {};
    const double limit = timeout_s > 0 ? timeout_s : 60.0;
    if (rank == 0) {
        std::snprintf(id, sizeof id, "%s.exchange", path);
        if (!write_atomically(path, id, sizeof id)) { ctx->last_error = std::string("cannot write the id file ") + path; return C2D_ERR_DIST; }
    } else if (!read_when_complete(path, id, sizeof id, limit)) {
        ctx->last_error = std::string("timed out waiting for rank 0 to write the id file ") + path;
        return C2D_ERR_DIST;
    }
    c2d_dist* d = new c2d_dist();
    d->ctx = ctx; d->rank = rank; d->world = world; d->base = id; d->timeout_s = limit;
    unsigned long long one = 1;   // every rank has read the id once the first sum completes: rank 0 then removes the file
    const int st = sum_over_ranks(d, &one, 1);
    if (st != C2D_OK || one != (unsigned long long)world) { delete d; return C2D_ERR_DIST; }
    if (rank == 0) std::remove(path);
    *out = d;
    return C2D_OK;
}
int c2d_dist_world_size(const c2d_dist* d) { return d ? d->world : -1; }
const char* c2d_dist_transport(const c2d_dist* d) { return d ? "file (cpu double)" : ""; }
int c2d_dist_rccl_version(int* version, char* path_out, size_t path_bytes)
{
    if (!version) return C2D_ERR_INVALID_ARG;
    *version = 0;
    if (path_out && path_bytes) std::snprintf(path_out, path_bytes, "file (cpu double)");
    return C2D_OK;
}
int c2d_dist_all_reduce_sum_u64(c2d_dist* d, unsigned long long* buf, size_t count, c2d_stream) { return d && (buf || !count) ? (count ? sum_over_ranks(d, buf, count) : C2D_OK) : C2D_ERR_INVALID_ARG; }
int c2d_dist_broadcast_u64(c2d_dist* d, unsigned long long* buf, size_t count, int root, c2d_stream)
{
    if (!d || (!buf && count) || root < 0 || root >= d->world) return C2D_ERR_INVALID_ARG;
    if (d->rank != root) std::memset(buf, 0, count * sizeof *buf);
    return count ? sum_over_ranks(d, buf, count) : C2D_OK;
}
int c2d_dist_stream_synchronize(c2d_dist* d, c2d_stream) { return d ? C2D_OK : C2D_ERR_INVALID_ARG; }
int c2d_dist_destroy(c2d_dist* d)
{
    if (!d) return C2D_OK;
    if (d->world > 1) {   // a last sum: every peer is past reading this rank's earlier files
        unsigned long long z = 0;
        d->timeout_s = 10.0;
        (void)sum_over_ranks(d, &z, 1);
    }
    for (unsigned long long s = d->seq >= 3 ? d->seq - 3 : 0; s < d->seq; s++)
        if (s + 1 < d->seq || d->world == 1) std::remove((d->base + "." + std::to_string(s) + "." + std::to_string(d->rank)).c_str());
    delete d;
    return C2D_OK;
}

}  // extern "C"
