// driver_common.hpp — pieces shared by the two CLI drivers (host side of the hot path).
#pragma once

#include <sys/stat.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <filesystem>
#include <iostream>
#include <random>
#include <string>
#include <vector>

#include "../../../include/c2d.h"
#include "cli.hpp"
#include "npy.hpp"

namespace fs = std::filesystem;

// Error convention of the drivers: print and leave main with EXIT_FAILURE (the
// reference's CUDA_CALL, utils.cu:70-72); the library itself never exits.
#define C2D_CALL(ctx, x)                                                                         \
    do {                                                                                         \
        int st__ = (x);                                                                          \
        if (st__ != C2D_OK) {                                                                    \
            std::fprintf(stderr, "Error at %s:%d: %s: %s (%s)\n", __FILE__, __LINE__, #x,        \
                         c2d_status_string(st__), (ctx) ? c2d_last_error(ctx) : "");            \
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
};

// One process per GPU: rank / world size come from --rank / --world_size or from the
// launcher's environment (RANK, WORLD_SIZE, LOCAL_RANK).  Batches are dealt round-robin,
// every batch file is written by exactly one rank, so no collective is needed on the data path.
inline Shard resolve_shard(const cli::Parser& p)
{
    Shard s;
    auto env_int = [](const char* k, int def) { const char* v = std::getenv(k); return v ? std::atoi(v) : def; };
    s.world = p.has("world_size") ? p.integer("world_size") : env_int("WORLD_SIZE", 1);
    s.rank = p.has("rank") ? p.integer("rank") : env_int("RANK", 0);
    s.device = p.has("device") ? p.integer("device") : env_int("LOCAL_RANK", s.rank);
    if (s.world < 1 || s.rank < 0 || s.rank >= s.world) throw std::runtime_error("bad --rank / --world_size");
    return s;
}

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

inline void print_json_summary(const char* tool, const Shard& sh, const RunStats& st, int batches)
{
    std::printf("{\"tool\": \"%s\", \"rank\": %d, \"world_size\": %d, \"batches\": %d, \"scenes\": %llu, \"mc_samples\": %llu, "
                "\"seconds\": %.3f, \"mc_samples_per_s\": %.4g, \"cp_hist_bins\": [0, 0.001, 0.01, 0.1, 1], \"cp_hist\": [%llu, %llu, %llu, %llu]}\n",
                tool, sh.rank, sh.world, batches, st.scenes, st.samples, st.seconds, st.seconds > 0 ? st.samples / st.seconds : 0.0,
                st.cp_hist[0], st.cp_hist[1], st.cp_hist[2], st.cp_hist[3]);
}
