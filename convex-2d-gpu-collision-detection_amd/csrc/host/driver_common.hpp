// driver_common.hpp — pieces shared by the two CLI drivers (host side of the hot path).
#pragma once

#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <cctype>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <filesystem>
#include <iostream>
#include <random>
#include <string>
#include <thread>
#include <vector>

#include "../../../include/c2d.h"
#include "cli.hpp"
#include "npy.hpp"

// a string as the inside of a JSON string literal (a path from dladdr may hold quotes, backslashes or control characters)
inline std::string json_escape(const std::string& in)
{
    std::string out;
    for (const unsigned char c : in) {
        if (c == '"' || c == '\\') { out += '\\'; out += static_cast<char>(c); }
        else if (c < 0x20) { char b[8]; std::snprintf(b, sizeof b, "\\u%04x", c); out += b; }
        else out += static_cast<char>(c);
    }
    return out;
}

namespace fs = std::filesystem;

// Error convention of the drivers: print and leave main with EXIT_FAILURE (the
// reference's CUDA_CALL, utils.cu:70-72); the library itself never exits.
// C2D_ERR_DIST is the exception: it may come from the watchdog of the aggregation link, which has then left a helper
// thread inside RCCL / HIP (include/c2d.h).  A return from main would run exit(): static destructors and the HIP / ROCr
// teardown would race that thread and can hang — defeating the watchdog.  So: report, flush, and end at once.
[[noreturn]] inline void c2d_die_now()
{
    std::fflush(stdout);
    std::fflush(stderr);
    _exit(EXIT_FAILURE);
}
#define C2D_CALL(ctx, x)                                                                         \
    do {                                                                                         \
        int st__ = (x);                                                                          \
        if (st__ != C2D_OK) {                                                                    \
            std::fprintf(stderr, "Error at %s:%d: %s: %s (%s)\n", __FILE__, __LINE__, #x,        \
                         c2d_status_string(st__), (ctx) ? c2d_last_error(ctx) : "");            \
            if (st__ == C2D_ERR_DIST) c2d_die_now();                                             \
            return EXIT_FAILURE;                                                                 \
        }                                                                                        \
    } while (0)

// Number of .npy files whose file name starts with an integer (reference utils.cu:36-56:
// std::stoi on the file name, so "12.npy" counts and "poses.npy" does not).
inline int get_num_batches_in_dir(const std::string& directoryPath)
{
    int fileCount = 0;
    if (!fs::is_directory(directoryPath)) return 0;
    for (const auto& entry : fs::directory_iterator(directoryPath)) {
        if (entry.is_regular_file() && entry.path().extension() == ".npy") {
            try {
                (void)std::stoi(entry.path().filename().string());
                fileCount++;
            } catch (...) {
                continue;
            }
        }
    }
    return fileCount;
}

inline void mkdirs(const std::string& path)
{
    if (!path.empty()) fs::create_directories(path);
}

struct Shard {
    int rank = 0;
    int world = 1;
    int device = 0;
    std::string id_file;   // where the ranks exchange the RCCL id (c2d_dist_init_file); empty = no aggregation
    bool from_env = false; // rank / world size came from a launcher's environment or from --rank / --world_size
};

// One process per GPU.  Rank / world size come from --rank / --world_size or from the launcher's environment
// (RANK, WORLD_SIZE, LOCAL_RANK); `--gpus N` makes the driver its own launcher (launch_ranks below).  Batches are
// dealt round-robin, every batch file is written by exactly one rank, so the data path needs no collective; the
// run closes with one RCCL sum of the counters (DistLink) and rank 0 prints the aggregated summary.
inline Shard resolve_shard(const cli::Parser& p)
{
    Shard s;
    auto env_int = [](const char* k, int def) { const char* v = std::getenv(k); return v ? std::atoi(v) : def; };
    s.from_env = p.has("world_size") || p.has("rank") || std::getenv("WORLD_SIZE") || std::getenv("RANK");
    s.world = p.has("world_size") ? p.integer("world_size") : env_int("WORLD_SIZE", 1);
    s.rank = p.has("rank") ? p.integer("rank") : env_int("RANK", 0);
    s.device = p.has("device") ? p.integer("device") : env_int("LOCAL_RANK", s.rank);
    if (p.has("dist_id_file")) s.id_file = p.str("dist_id_file");
    else if (const char* f = std::getenv("C2D_DIST_ID_FILE")) s.id_file = f;
    if (s.world < 1 || s.rank < 0 || s.rank >= s.world) throw std::runtime_error("bad --rank / --world_size");
    return s;
}

inline void add_shard_options(cli::Parser& p)
{
    using K = cli::Option;
    p.add("gpus", 0, K::VALUE, "use N GPUs of this node: the driver starts one process per GPU itself and prints one aggregated summary "
                               "(refused under a preloaded profiler: profile ONE rank, --rank k --world_size N --dist_id_file F, never the launcher)");
    p.add("rank", 0, K::VALUE, "this process' shard index (default: $RANK or 0)");
    p.add("world_size", 0, K::VALUE, "number of shards = GPUs (default: $WORLD_SIZE or 1)");
    p.add("device", 0, K::VALUE, "GPU index (default: $LOCAL_RANK or rank)");
    p.add("dist_id_file", 0, K::VALUE, "fresh file through which externally launched ranks exchange the RCCL id (default: $C2D_DIST_ID_FILE)");
}

// `--gpus N`: this process becomes the launcher.  It has not touched a GPU (it only parsed flags and looked at
// directories); it starts N fresh copies of itself, one per GPU, with RANK / WORLD_SIZE / LOCAL_RANK and a fresh
// C2D_DIST_ID_FILE in their environment and `extra` appended to their arguments — values every rank must agree on
// (first output batch number, seed) are resolved ONCE here, so ranks cannot race on them — waits for all of them and
// returns the first non-zero exit status (remaining ranks are terminated when one fails).
inline int launch_ranks(int n, int argc, char** argv, const std::vector<std::string>& extra);

// The aggregation link of a rank: c2d_dist (RCCL) plus one small device buffer.
struct DistLink {
    c2d_ctx* ctx = nullptr;
    c2d_dist* dist = nullptr;
    unsigned long long* d_buf = nullptr;
    static constexpr size_t kWords = 16;
    int open(c2d_ctx* c, const Shard& sh)
    {
        ctx = c;
        if (sh.world <= 1 && sh.id_file.empty()) return C2D_OK;
        if (sh.id_file.empty()) return C2D_OK;  // externally launched without an id file: no aggregation (each rank reports itself)
        // 0 = the library's default limit: $C2D_DIST_TIMEOUT_S, or 300 s
        int st = c2d_dist_init_file(ctx, sh.rank, sh.world, sh.id_file.c_str(), 0.0, &dist);
        if (st != C2D_OK) return st;
        // one line per rank: which card, which RCCL — a multi-GPU record must be readable without guessing
        c2d_device_info di;
        char lib[1024] = "";
        (void)c2d_dist_rccl_version(&rccl_version, lib, sizeof lib);
        rccl_library = lib;
        if (c2d_ctx_info(ctx, &di) == C2D_OK)
            std::fprintf(stderr, "[c2d] rank %d of %d: device %d (%s, PCI %s), reduce: %s over %d ranks, RCCL %d from %s\n", sh.rank, sh.world, di.device,
                         di.name, di.pci_bus_id[0] ? di.pci_bus_id : "?", c2d_dist_transport(dist), c2d_dist_world_size(dist), rccl_version, lib);
        return c2d_malloc(ctx, reinterpret_cast<void**>(&d_buf), kWords * sizeof(unsigned long long));
    }
    int rccl_version = 0;
    std::string rccl_library;
    bool active() const { return dist != nullptr; }
    // in-place sum over ranks of up to kWords host counters
    int sum(unsigned long long* h, size_t count, c2d_stream stream)
    {
        if (!dist) return C2D_OK;
        if (count > kWords) return C2D_ERR_INVALID_ARG;
        int st = c2d_memcpy_h2d(ctx, d_buf, h, count * sizeof *h, stream);
        if (st == C2D_OK) st = c2d_dist_all_reduce_sum_u64(dist, d_buf, count, stream);
        if (st == C2D_OK) st = c2d_memcpy_d2h(ctx, h, d_buf, count * sizeof *h, stream);
        if (st == C2D_OK) st = c2d_dist_stream_synchronize(dist, stream);  // (under the watchdog: a peer may have died since the link was built)
        return st;
    }
    // rank 0's values to everyone
    int broadcast(unsigned long long* h, size_t count, c2d_stream stream)
    {
        if (!dist) return C2D_OK;
        if (count > kWords) return C2D_ERR_INVALID_ARG;
        int st = c2d_memcpy_h2d(ctx, d_buf, h, count * sizeof *h, stream);
        if (st == C2D_OK) st = c2d_dist_broadcast_u64(dist, d_buf, count, 0, stream);
        if (st == C2D_OK) st = c2d_memcpy_d2h(ctx, h, d_buf, count * sizeof *h, stream);
        if (st == C2D_OK) st = c2d_dist_stream_synchronize(dist, stream);
        return st;
    }
    void close()
    {
        if (d_buf) c2d_free(ctx, d_buf);
        if (dist) c2d_dist_destroy(dist);
        d_buf = nullptr;
        dist = nullptr;
    }
    // An error return from main() (C2D_CALL) releases the buffer; the communicator is left to the end of the process there:
    // ncclCommDestroy after a failed run may wait for peers that are already gone.  The normal path calls close().
    ~DistLink()
    {
        if (d_buf) c2d_free(ctx, d_buf);
        d_buf = nullptr;
    }
};

// What a driver's main() allocates through the C-ABI, released whichever way main() is left — C2D_CALL returns on the first
// error.  Declare AFTER the ctx owner (BatchSlot slots / CtxScope): objects are destroyed in reverse order of declaration, and
// the buffers must go before their ctx.
struct DeviceBuffers {
    c2d_ctx* ctx = nullptr;
    std::vector<void**> owned;
    explicit DeviceBuffers(c2d_ctx* c) : ctx(c) {}
    DeviceBuffers(const DeviceBuffers&) = delete;
    DeviceBuffers& operator=(const DeviceBuffers&) = delete;
    void own(std::initializer_list<void**> ptrs) { owned.insert(owned.end(), ptrs.begin(), ptrs.end()); }
    void release()
    {
        for (auto it = owned.rbegin(); it != owned.rend(); ++it)
            if (**it) { c2d_free(ctx, **it); **it = nullptr; }
    }
    ~DeviceBuffers() { release(); }
};

// ctx + stream of a driver that runs without BatchSlots (ztest, the single-pair mode)
struct CtxScope {
    c2d_ctx* ctx = nullptr;
    c2d_stream stream = nullptr;
    CtxScope() = default;
    CtxScope(const CtxScope&) = delete;
    CtxScope& operator=(const CtxScope&) = delete;
    int open(int device)
    {
        int st = c2d_ctx_create(device, &ctx);
        if (st == C2D_OK) st = c2d_stream_create(ctx, &stream);
        return st;
    }
    void close()
    {
        if (!ctx) return;
        if (stream) c2d_stream_destroy(ctx, stream);
        c2d_ctx_destroy(ctx);
        ctx = nullptr;
        stream = nullptr;
    }
    ~CtxScope() { close(); }
};

// Two batches in flight per rank.  A batch is: scenes on the device (sampled there, or uploaded), the adaptive loop,
// the download of its rows / hit counts / sample counts.  All of it is asynchronous on the slot's stream, so while the
// GPU works on batch k + 1 the host finishes batch k (statistics, shuffle, .npy file) — the reference does these
// strictly one after the other (compute_collision_probability.cu:259-358).  Each slot has its own c2d_ctx (a ctx owns
// one workspace) and its own stream; the pose / std_dev tables are shared device buffers.
struct BatchSlot {
    c2d_ctx* ctx = nullptr;
    c2d_stream stream = nullptr;
    void *d_scenes = nullptr, *d_hits = nullptr, *d_used = nullptr, *d_rows = nullptr;
    // page-locked host buffers, so that the copies do not hold the host thread
    PoseCPVarAndPoseIdx* dataset = nullptr;
    uint32_t *hits = nullptr, *used = nullptr;
    float* input = nullptr;     // compute_collision_probability: the [n,4] rows being uploaded
    size_t n = 0;
    int batch_index = -1;       // -1: idle
    int open(int device, size_t n_rows)
    {
        n = n_rows;
        int st = c2d_ctx_create(device, &ctx);
        if (st == C2D_OK) st = c2d_stream_create(ctx, &stream);
        if (st == C2D_OK) st = c2d_malloc(ctx, &d_scenes, n * sizeof(PositionWithVarAndPoseIdx));
        if (st == C2D_OK) st = c2d_malloc(ctx, &d_hits, n * sizeof(uint32_t));
        if (st == C2D_OK) st = c2d_malloc(ctx, &d_used, n * sizeof(uint32_t));
        if (st == C2D_OK) st = c2d_malloc(ctx, &d_rows, n * sizeof(PoseCPVarAndPoseIdx));
        if (st == C2D_OK) st = c2d_malloc_host(ctx, reinterpret_cast<void**>(&dataset), n * sizeof(PoseCPVarAndPoseIdx));
        if (st == C2D_OK) st = c2d_malloc_host(ctx, reinterpret_cast<void**>(&hits), n * sizeof(uint32_t));
        if (st == C2D_OK) st = c2d_malloc_host(ctx, reinterpret_cast<void**>(&used), n * sizeof(uint32_t));
        if (st == C2D_OK) st = c2d_malloc_host(ctx, reinterpret_cast<void**>(&input), n * sizeof(PositionWithVarAndPoseIdx));
        return st;
    }
    // enqueue the downloads of a finished adaptive loop
    int download()
    {
        int st = c2d_memcpy_d2h(ctx, dataset, d_rows, n * sizeof(PoseCPVarAndPoseIdx), stream);
        if (st == C2D_OK) st = c2d_memcpy_d2h(ctx, hits, d_hits, n * sizeof(uint32_t), stream);
        if (st == C2D_OK) st = c2d_memcpy_d2h(ctx, used, d_used, n * sizeof(uint32_t), stream);
        return st;
    }
    BatchSlot() = default;
    BatchSlot(const BatchSlot&) = delete;
    BatchSlot& operator=(const BatchSlot&) = delete;
    ~BatchSlot() { close(); }   // (an error return from main() leaves through here)
    void close()
    {
        if (!ctx) return;
        for (void* ptr : {d_scenes, d_hits, d_used, d_rows}) c2d_free(ctx, ptr);
        for (void* ptr : {static_cast<void*>(dataset), static_cast<void*>(hits), static_cast<void*>(used), static_cast<void*>(input)}) c2d_free_host(ctx, ptr);
        c2d_stream_destroy(ctx, stream);
        c2d_ctx_destroy(ctx);
        ctx = nullptr;
    }
};

inline std::vector<StdDev> std_devs_from_variances(const std::vector<float>& var_flat)
{
    // element-wise sqrt (generate_dataset.cu:309-317, compute_collision_probability.cu:188-194)
    std::vector<StdDev> sd(var_flat.size() / 5);
    for (size_t i = 0; i < sd.size(); i++) {
        sd[i].x = std::sqrt(var_flat[5 * i + 0]);
        sd[i].y = std::sqrt(var_flat[5 * i + 1]);
        sd[i].theta = std::sqrt(var_flat[5 * i + 2]);
        sd[i].width = std::sqrt(var_flat[5 * i + 3]);
        sd[i].height = std::sqrt(var_flat[5 * i + 4]);
    }
    return sd;
}

// wall-clock split of a run, printed in the JSON summary (rank 0's own phases)
struct PhaseClock {
    std::chrono::steady_clock::time_point t = std::chrono::steady_clock::now();
    std::vector<std::pair<std::string, double>> phases;
    void lap(const char* name)
    {
        const auto now = std::chrono::steady_clock::now();
        phases.emplace_back(name, std::chrono::duration<double>(now - t).count());
        t = now;
    }
    void add(const char* name, double seconds) { phases.emplace_back(name, seconds); }
    std::string json() const
    {
        std::string s = ", \"phases_s\": {";
        char buf[96];
        for (size_t i = 0; i < phases.size(); i++) {
            std::snprintf(buf, sizeof buf, "%s\"%s\": %.4f", i ? ", " : "", phases[i].first.c_str(), phases[i].second);
            s += buf;
        }
        return s + "}";
    }
};

struct RunStats {
    // histogram of cp over the bins of the reference's dataset tooling (balance_datasets.py:15-20, :36:
    // [0,.001) [.001,.01) [.01,.1) [.1,1]) — what its plt.hist(data[:,2], accuracy_bins) draws
    unsigned long long cp_hist[4] = {0, 0, 0, 0};
    void add_cp(float cp)
    {
        // double edges against the promoted float32 value, as numpy / matplotlib compare them
        static const double edges[5] = {0.0, 0.001, 0.01, 0.1, 1.0};
        const double v = cp;
        for (int i = 0; i < 4; i++) {
            const bool last = i == 3;
            if (v >= edges[i] && (last ? v <= edges[i + 1] : v < edges[i + 1])) { cp_hist[i]++; return; }
        }
    }
    unsigned long long samples = 0;
    unsigned long long hits = 0;
    unsigned long long scenes = 0;
    double seconds = 0;
};

// One line per run.  With an aggregation link it is printed by rank 0 only and holds the sums over all ranks
// (samples, hits, scenes, batches, the cp histogram; seconds = the slowest rank's).
inline int print_json_summary(const char* tool, const Shard& sh, RunStats st, int batches, DistLink* link, c2d_stream stream,
                              const std::string& extra_json = "")
{
    const char* reduce = "none";
    int ranks = 1;
    std::string rccl_json;
    if (link && link->active()) {
        rccl_json = ", \"rccl_version\": " + std::to_string(link->rccl_version) + ", \"rccl_library\": \"" + json_escape(link->rccl_library) + "\"";
        unsigned long long w[9] = {st.samples, st.hits, st.scenes, st.cp_hist[0], st.cp_hist[1], st.cp_hist[2], st.cp_hist[3],
                                   static_cast<unsigned long long>(batches), 1ull};
        int rc = link->sum(w, 9, stream);
        if (rc != C2D_OK) return rc;
        // wall time of the run = the slowest rank's: every rank puts its milliseconds into its own word of a second sum
        if (sh.world <= static_cast<int>(DistLink::kWords)) {
            unsigned long long ms[DistLink::kWords] = {};
            ms[sh.rank] = static_cast<unsigned long long>(st.seconds * 1e3);
            rc = link->sum(ms, static_cast<size_t>(sh.world), stream);
            if (rc != C2D_OK) return rc;
            st.seconds = static_cast<double>(*std::max_element(ms, ms + sh.world)) / 1e3;
        }
        st.samples = w[0]; st.hits = w[1]; st.scenes = w[2];
        for (int i = 0; i < 4; i++) st.cp_hist[i] = w[3 + i];
        batches = static_cast<int>(w[7]);
        ranks = static_cast<int>(w[8]);
        reduce = c2d_dist_transport(link->dist);
        if (sh.rank != 0) return C2D_OK;
    }
    std::printf("{\"tool\": \"%s\", \"rank\": %d, \"world_size\": %d, \"aggregated_over_ranks\": %d, \"reduce\": \"%s\", \"batches\": %d, "
                "\"scenes\": %llu, \"mc_samples\": %llu, \"hits\": %llu, \"pooled_probability\": %.9g, "
                "\"seconds\": %.3f, \"mc_samples_per_s\": %.4g, \"cp_hist_bins\": [0, 0.001, 0.01, 0.1, 1], \"cp_hist\": [%llu, %llu, %llu, %llu]%s%s}\n",
                tool, sh.rank, sh.world, ranks, reduce, batches, st.scenes, st.samples, st.hits,
                st.samples ? static_cast<double>(st.hits) / static_cast<double>(st.samples) : 0.0, st.seconds,
                st.seconds > 0 ? st.samples / st.seconds : 0.0, st.cp_hist[0], st.cp_hist[1], st.cp_hist[2], st.cp_hist[3], rccl_json.c_str(), extra_json.c_str());
    std::fflush(stdout);
    return C2D_OK;
}

#include <signal.h>
#include <sys/wait.h>
#include <unistd.h>

// --gpus N starts the ranks by fork + execv of this very program.  That is only safe while THIS process has not initialised the GPU
// — it has not: the launcher never calls into libc2d — but a profiler or tool library preloaded into it (rocprofv3 and friends
// put theirs into every process they start) has done so before main, and an exec from a process that holds the GPU is what this
// pool's hosts forbid: it can take the machine down.  So: which preloaded tool would make the self-launch unsafe, or nullptr.
inline const char* preloaded_gpu_tool(std::string* what)
{
    static const char* const vars[] = {"ROCP_TOOL_LIBRARIES", "HSA_TOOLS_LIB", "ROCPROFILER_REGISTER_LIBRARY", "LD_PRELOAD"};
    static const char* const names[] = {"rocprof", "roctracer", "roctx", "rocprofiler", "librocm-debug", "libhsa", "libamdhip"};
    for (const char* v : vars) {
        const char* val = std::getenv(v);
        if (!val || !*val) continue;
        if (std::string(v) != "LD_PRELOAD") { *what = std::string(v) + "=" + val; return v; }   // a tool library is loaded into the HSA runtime
        std::string low(val);
        for (auto& c : low) c = static_cast<char>(std::tolower(static_cast<unsigned char>(c)));
        for (const char* nm : names)
            if (low.find(nm) != std::string::npos) { *what = std::string(v) + "=" + val; return v; }
    }
    return nullptr;
}

inline int launch_ranks(int n, int argc, char** argv, const std::vector<std::string>& extra)
{
    std::string tool;
    if (preloaded_gpu_tool(&tool)) {
        std::fprintf(stderr, "error: --gpus %d refused: a profiler / tool library is preloaded into this process (%s).  The launcher starts its "
                             "ranks by execv, which is only safe from a process that has not initialised the GPU, and a preloaded tool has.  "
                             "Profile ONE rank instead: start the ranks by hand (--rank k --world_size %d --dist_id_file FILE on each, RANK / "
                             "WORLD_SIZE / LOCAL_RANK work too) and put the profiler in front of one of them.\n", n, tool.c_str(), n);
        return EXIT_FAILURE;
    }
    std::vector<std::string> args;
    for (int i = 0; i < argc; i++) {
        const std::string a = argv[i];
        if (a == "--gpus") { i++; continue; }            // the children are plain ranks
        if (a.rfind("--gpus=", 0) == 0) continue;
        args.push_back(a);
    }
    for (const auto& e : extra) args.push_back(e);
    char exe[4096];
    const ssize_t len = readlink("/proc/self/exe", exe, sizeof exe - 1);
    if (len <= 0) { std::fprintf(stderr, "error: cannot resolve /proc/self/exe\n"); return EXIT_FAILURE; }
    exe[len] = 0;
    const char* tmp = std::getenv("TMPDIR");
    const std::string id_file = std::string(tmp && *tmp ? tmp : "/tmp") + "/c2d_dist_id_" + std::to_string(static_cast<long long>(getpid())) + "_" +
                                std::to_string(static_cast<long long>(std::chrono::steady_clock::now().time_since_epoch().count()));
    std::remove(id_file.c_str());
    std::vector<pid_t> pids;
    for (int r = 0; r < n; r++) {
        const pid_t pid = fork();
        if (pid < 0) { std::perror("fork"); break; }
        if (pid == 0) {
            setenv("RANK", std::to_string(r).c_str(), 1);
            setenv("WORLD_SIZE", std::to_string(n).c_str(), 1);
            if (!std::getenv("C2D_SHARE_DEVICE")) setenv("LOCAL_RANK", std::to_string(r).c_str(), 1);
            else setenv("LOCAL_RANK", "0", 1);                      // rehearsal on a one-GPU box (with C2D_DIST_TRANSPORT=file)
            setenv("C2D_DIST_ID_FILE", id_file.c_str(), 1);
            setenv("HSA_ENABLE_IPC_MODE_LEGACY", "0", 0);
            std::vector<char*> cargv;
            for (auto& a : args) cargv.push_back(const_cast<char*>(a.c_str()));
            cargv.push_back(nullptr);
            execv(exe, cargv.data());
            std::perror("execv");
            _exit(127);
        }
        pids.push_back(pid);
    }
    int rc = static_cast<int>(pids.size()) == n ? 0 : EXIT_FAILURE;
    size_t left = pids.size();
    while (left > 0) {
        int status = 0;
        const pid_t done = wait(&status);
        if (done < 0) break;
        left--;
        const int code = WIFEXITED(status) ? WEXITSTATUS(status) : 128 + (WIFSIGNALED(status) ? WTERMSIG(status) : 0);
        if (code != 0 && rc == 0) {
            rc = code;
            for (pid_t p : pids) if (p != done) kill(p, SIGTERM);  // exact PIDs of our own children
        }
    }
    std::remove(id_file.c_str());
    return rc;
}
