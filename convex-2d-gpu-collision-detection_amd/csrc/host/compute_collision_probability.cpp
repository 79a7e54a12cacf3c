// compute_collision_probability — host driver with the reference's command-line surface and
// .npy layout (reference compute_collision_probability.cu:35-85 flags, :152-379 main), driving
// the HIP kernels through the C-ABI of include/c2d.h.
//
// Reads data_out/{poses,variances}.npy, data_out/meta/{accuracy_bins,bin_accuracy}.npy and the
// batches data_in/<k>.npy [N,4] rows (x, y, var_idx, pose_idx); writes data_out/<start+k>.npy
// [N,5] rows (x, y, cp, var_idx, pose_idx), start = number of integer-named .npy already in
// data_out (:157, :355).  Results come back in input order (scene i keeps slot i), so the
// reference's index permutation (:224-228, :337-344) is not needed; --shuffle (default true)
// then shuffles exactly as the reference does (:346-349).
// Deliberate differences: hit counters are zeroed per batch (D4).
// Additions: --seed; --start_batch_count; multi-GPU (--gpus N, or one externally launched process per GPU with
// --rank / --world_size / --device / --dist_id_file): batches are dealt round-robin and ONE RCCL sum of the counters
// gives a single aggregated summary (rank 0); --pair_samples S runs BASELINE config 3 instead — one scene
// (--pair_pos, --pair_pose, --pair_std_dev), S samples split over the ranks by sample index, hits summed, p printed.
#include "driver_common.hpp"

struct Arguments {  // defaults: compute_collision_probability.cu:35-42
    std::string data_in = "./data_in/";
    std::string data_out = "./data_out/";
    int max_samples = 4000000;
    float robot_width = 4.07;
    float robot_height = 1.74;
    bool shuffle = true;
    unsigned long long seed = 0;
    int start_batch_count = -1;          // -1: count the integer-named .npy files in data_out (:157)
    unsigned long long pair_samples = 0; // > 0: single-scene mode (config 3)
    std::vector<float> pair_pos = {3.0f, 1.0f};                       // robot position
    std::vector<float> pair_pose = {2.0f, 1.0f, 0.6f};                 // obstacle width, height; robot theta (Pose)
    std::vector<float> pair_std_dev = {0.3f, 0.3f, 0.2f, 0.0f, 0.0f};  // x, y, theta, width, height
};

// BASELINE config 3 over all ranks: the sample index space [0, S) is cut into contiguous ranges, one per rank
// (the stream is keyed by the sample index, so the union is bit-identical to a one-GPU run), one RCCL sum of the
// hit counts, p = hits / S.
static int run_single_pair(const Arguments& a, const Shard& shard)
{
    CtxScope scope;   // ctx, stream, link buffer and device words are released on every way out
    C2D_CALL(scope.ctx, scope.open(shard.device));
    c2d_ctx* ctx = scope.ctx;
    c2d_stream stream = scope.stream;
    DistLink link;
    C2D_CALL(ctx, link.open(ctx, shard));
    const unsigned long long S = a.pair_samples, W = static_cast<unsigned long long>(shard.world), r = static_cast<unsigned long long>(shard.rank);
    const unsigned long long base = S / W, rem = S % W;
    const unsigned long long begin = r * base + std::min(r, rem), count = base + (r < rem ? 1 : 0);
    const Position pos{a.pair_pos[0], a.pair_pos[1]};
    const Pose pose{a.pair_pose[0], a.pair_pose[1], a.pair_pose[2]};
    const StdDev sd{a.pair_std_dev[0], a.pair_std_dev[1], a.pair_std_dev[2], a.pair_std_dev[3], a.pair_std_dev[4]};
    void* d_hits = nullptr;
    DeviceBuffers buffers(ctx);
    buffers.own({&d_hits});
    C2D_CALL(ctx, c2d_malloc(ctx, &d_hits, sizeof(unsigned long long)));
    C2D_CALL(ctx, c2d_memset(ctx, d_hits, 0, sizeof(unsigned long long), stream));
    const auto t0 = std::chrono::steady_clock::now();
    if (count) C2D_CALL(ctx, c2d_mc_pair(ctx, a.robot_width, a.robot_height, &pos, &pose, &sd, a.seed, 0, begin, count,
                                         static_cast<unsigned long long*>(d_hits), stream));
    unsigned long long w[3] = {0, count, 1};
    C2D_CALL(ctx, c2d_memcpy_d2h(ctx, &w[0], d_hits, sizeof w[0], stream));
    C2D_CALL(ctx, c2d_stream_synchronize(ctx, stream));
    const double local_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    C2D_CALL(ctx, link.sum(w, 3, stream));  // the single RCCL reduce of the hit counts
    const double total_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (shard.rank == 0 || !link.active())
        std::printf("{\"tool\": \"compute_collision_probability\", \"mode\": \"single_pair\", \"rank\": %d, \"world_size\": %d, "
                    "\"aggregated_over_ranks\": %llu, \"reduce\": \"%s\", \"samples\": %llu, \"hits\": %llu, \"p\": %.9f, "
                    "\"seconds\": %.4f, \"rank_kernel_seconds\": %.4f, \"mc_samples_per_s\": %.4g}\n",
                    shard.rank, shard.world, w[2], link.active() ? c2d_dist_transport(link.dist) : "none", w[1], w[0],
                    w[1] ? static_cast<double>(w[0]) / static_cast<double>(w[1]) : 0.0, total_s, local_s, total_s > 0 ? w[1] / total_s : 0.0);
    buffers.release();
    link.close();
    scope.close();
    return 0;
}

int main(int argc, char* argv[])
{
    Arguments a;
    cli::Parser p;
    using K = cli::Option;
    p.add("help", 0, K::SWITCH, "produce help message");
    p.add("data_in", 0, K::VALUE, "where to read the data");
    p.add("data_out", 0, K::VALUE, "where to write the data");
    p.add("max_samples", 0, K::VALUE, "maximum number of samples for z-test");
    p.add("robot_width", 'w', K::VALUE, "robot width");
    p.add("robot_height", 'h', K::VALUE, "robot height");
    p.add("shuffle", 0, K::VALUE, "whether or not to shuffle data");
    p.add("seed", 0, K::VALUE, "seed of the Monte-Carlo stream (default 0; the reference's std::rand() seed is fixed too, :249-251)");
    p.add("start_batch_count", 0, K::VALUE, "number of the first output batch (default: the count of integer-named .npy in data_out)");
    p.add("pair_samples", 0, K::VALUE, "single-scene mode: Monte-Carlo samples of ONE robot/obstacle pair, split over the GPUs; prints p");
    p.add("pair_pos", 0, K::MULTI, "single-scene mode: robot position x y (default 3 1)");
    p.add("pair_pose", 0, K::MULTI, "single-scene mode: obstacle width, height and robot theta (default 2 1 0.6)");
    p.add("pair_std_dev", 0, K::MULTI, "single-scene mode: std dev of x y theta width height (default 0.3 0.3 0.2 0 0)");
    add_shard_options(p);
    Shard shard;
    int gpus = 1;
    try {
        p.parse(argc, argv);
        if (p.has("help")) { p.print_help(std::cout); return 1; }
        if (p.has("data_in")) a.data_in = p.str("data_in");
        if (p.has("data_out")) a.data_out = p.str("data_out");
        if (p.has("max_samples")) a.max_samples = p.integer("max_samples");
        if (p.has("robot_width")) a.robot_width = p.real("robot_width");
        if (p.has("robot_height")) a.robot_height = p.real("robot_height");
        if (p.has("shuffle")) a.shuffle = p.boolean("shuffle");
        if (p.has("seed")) a.seed = std::stoull(p.str("seed"), nullptr, 0);
        if (p.has("start_batch_count")) a.start_batch_count = p.integer("start_batch_count");
        if (p.has("pair_samples")) a.pair_samples = std::stoull(p.str("pair_samples"), nullptr, 0);
        auto take = [&](const char* name, std::vector<float>& dst, size_t nvals) {
            if (!p.has(name)) return;
            dst = p.reals(name);
            if (dst.size() != nvals) throw std::runtime_error(std::string("--") + name + " needs " + std::to_string(nvals) + " values");
        };
        take("pair_pos", a.pair_pos, 2);
        take("pair_pose", a.pair_pose, 3);
        take("pair_std_dev", a.pair_std_dev, 5);
        if (p.has("gpus")) gpus = p.integer("gpus");
        shard = resolve_shard(p);
        if (a.max_samples <= 0) throw std::runtime_error("--max_samples must be positive");
        if (gpus < 1) throw std::runtime_error("--gpus must be at least 1");
        if (gpus > 1 && shard.from_env) throw std::runtime_error("--gpus N starts the ranks itself: do not combine it with --rank / --world_size or a launcher");
    } catch (const std::exception& e) {
        std::cerr << "error: " << e.what() << "\n";
        p.print_help(std::cerr);
        return EXIT_FAILURE;
    }
    if (a.pair_samples == 0 && a.start_batch_count < 0) {
        // Ranks of a multi-GPU run must agree on the first output number.  Counting the directory per rank races with
        // the ranks that already write into it, so it is counted ONCE: here by the launcher (--gpus N) or by a single
        // process; externally launched ranks get it from rank 0 through the aggregation link (below) or must pass it.
        if (shard.world > 1 && shard.id_file.empty()) {
            std::cerr << "error: ranks launched by hand need --start_batch_count (or --dist_id_file / $C2D_DIST_ID_FILE, or use --gpus N)\n";
            return EXIT_FAILURE;
        }
        a.start_batch_count = get_num_batches_in_dir(a.data_out);  // :157
    }
    if (gpus > 1) {
        std::vector<std::string> extra;
        if (a.pair_samples == 0) extra = {"--start_batch_count", std::to_string(a.start_batch_count)};
        return launch_ranks(gpus, argc, argv, extra);
    }
    if (a.pair_samples > 0) return run_single_pair(a, shard);
    int start_batch_count = a.start_batch_count;
    const int num_batches = get_num_batches_in_dir(a.data_in);         // :158
    const bool chatty = shard.rank == 0;  // one rank narrates

    if (chatty) std::cout << "Reading data..." << std::endl;
    npy::Array poses, variances, first, accuracy_bins, bin_accuracy;
    try {
        poses = npy::load_f32(a.data_out + "/poses.npy");                    // :162-166
        variances = npy::load_f32(a.data_out + "/variances.npy");
        accuracy_bins = npy::load_f32(a.data_out + "/meta/accuracy_bins.npy");
        bin_accuracy = npy::load_f32(a.data_out + "/meta/bin_accuracy.npy");
        if (num_batches > 0) first = npy::load_f32(a.data_in + "/0.npy");
        if (poses.data.size() % 3 || variances.data.size() % 5) throw std::runtime_error("poses.npy must be [N,3] and variances.npy [N,5]");
        if (accuracy_bins.data.size() < 2 || bin_accuracy.data.size() + 1 < accuracy_bins.data.size())
            throw std::runtime_error("meta/bin_accuracy.npy needs accuracy_bins - 1 entries");
        if (first.data.size() % 4) throw std::runtime_error("data_in batches must be [N,4]");
    } catch (const std::exception& e) {
        std::cerr << "error: " << e.what() << "\n";
        return EXIT_FAILURE;
    }
    const int num_poses = static_cast<int>(poses.data.size() / 3);
    const int num_variances = static_cast<int>(variances.data.size() / 5);
    const size_t N = first.data.size() / 4;  // taken from 0.npy and assumed for all batches (:164, :171)
    if (chatty) {
        std::cout << "num poses: " << num_poses << std::endl;
        std::cout << "num variances: " << num_variances << std::endl;
        std::cout << "num data points: " << N << std::endl;
    }
    if (num_batches == 0 || N == 0) { if (chatty) std::cout << "nothing to do" << std::endl; return 0; }
    if (num_poses == 0 || num_variances == 0) { std::cerr << "error: empty pose / variance table\n"; return EXIT_FAILURE; }
    std::vector<StdDev> std_devs = std_devs_from_variances(variances.data);

    BatchSlot slots[2];
    for (auto& sl : slots) C2D_CALL(sl.ctx, sl.open(shard.device, N));
    c2d_ctx* ctx = slots[0].ctx;          // owner of the tables and of the aggregation link
    c2d_stream stream = slots[0].stream;
    DistLink link;
    C2D_CALL(ctx, link.open(ctx, shard));
    if (link.active()) {  // rank 0's view of the output directory is everyone's
        unsigned long long w[1] = {static_cast<unsigned long long>(start_batch_count)};
        C2D_CALL(ctx, link.broadcast(w, 1, stream));
        start_batch_count = static_cast<int>(w[0]);
    }
    void *d_poses = nullptr, *d_sd = nullptr;
    DeviceBuffers buffers(ctx);   // (declared after the slots that own the ctx: released before them on every way out)
    buffers.own({&d_poses, &d_sd});
    C2D_CALL(ctx, c2d_malloc(ctx, &d_poses, poses.data.size() * sizeof(float)));
    C2D_CALL(ctx, c2d_malloc(ctx, &d_sd, std_devs.size() * sizeof(StdDev)));
    C2D_CALL(ctx, c2d_memcpy_h2d(ctx, d_poses, poses.data.data(), poses.data.size() * sizeof(float), stream));
    C2D_CALL(ctx, c2d_memcpy_h2d(ctx, d_sd, std_devs.data(), std_devs.size() * sizeof(StdDev), stream));
    C2D_CALL(ctx, c2d_stream_synchronize(ctx, stream));

    const auto begin = std::chrono::steady_clock::now();
    if (chatty) {
        std::cout << "Total number of configurations: " << static_cast<long long>(N) * num_batches << std::endl;
        std::cout << "Begin computation..." << std::endl;
    }
    int counter = 0;
    RunStats stats;
    if (chatty) std::printf("batches generated: %i/%i\n", counter, num_batches);

    // read + upload + adaptive loop + downloads of one batch, enqueued without waiting for the GPU
    auto enqueue = [&](BatchSlot& sl, int batch_index) -> int {
        npy::Array batch = npy::load_f32(a.data_in + "/" + std::to_string(batch_index) + ".npy");  // :261
        if (batch.data.size() != N * 4) throw std::runtime_error(std::to_string(batch_index) + ".npy does not hold " + std::to_string(N) + " rows of 4");
        std::copy(batch.data.begin(), batch.data.end(), sl.input);  // page-locked: the upload below does not wait
        // the [N,4] rows are PositionWithVarAndPoseIdx records: upload them as they are (:262-274)
        int st = c2d_memcpy_h2d(sl.ctx, sl.d_scenes, sl.input, N * sizeof(PositionWithVarAndPoseIdx), sl.stream);
        if (st != C2D_OK) return st;
        c2d_mc_scenes_args m{};
        m.d_poses = static_cast<const Pose*>(d_poses); m.num_poses = num_poses;
        m.d_std_devs = static_cast<const StdDev*>(d_sd); m.num_std_devs = num_variances;
        m.d_scenes = static_cast<const PositionWithVarAndPoseIdx*>(sl.d_scenes); m.n_scenes = N;
        m.robot_w = a.robot_width; m.robot_h = a.robot_height;
        m.accuracy_bins = accuracy_bins.data.data(); m.bin_accuracy = bin_accuracy.data.data();
        m.n_accuracy_bins = static_cast<uint32_t>(accuracy_bins.data.size());
        m.max_samples = static_cast<uint32_t>(a.max_samples);
        m.seed = a.seed;
        m.scene_id_base = (static_cast<uint64_t>(start_batch_count) + batch_index) * N;
        m.d_hits = static_cast<uint32_t*>(sl.d_hits); m.d_n_used = static_cast<uint32_t*>(sl.d_used);
        m.d_rows = static_cast<PoseCPVarAndPoseIdx*>(sl.d_rows);
        st = c2d_mc_scenes(sl.ctx, &m, sl.stream);  // adaptive loop, :276-332 (no host output requested: asynchronous)
        if (st == C2D_OK) st = sl.download();
        sl.batch_index = batch_index;
        return st;
    };
    auto finish = [&](BatchSlot& sl) -> int {
        int st = c2d_stream_synchronize(sl.ctx, sl.stream);
        if (st != C2D_OK) return st;
        stats.scenes += N;
        for (size_t i = 0; i < N; i++) stats.samples += sl.used[i];
        for (size_t i = 0; i < N; i++) stats.hits += sl.hits[i];
        for (size_t i = 0; i < N; i++) stats.add_cp(sl.dataset[i].cp);
        if (a.shuffle) std::shuffle(sl.dataset, sl.dataset + N, std::default_random_engine(0));  // :346-349
        npy::save_f32(a.data_out + "/" + std::to_string(start_batch_count + sl.batch_index) + ".npy", {N, 5},
                      reinterpret_cast<const float*>(sl.dataset));  // :353-355
        sl.batch_index = -1;
        const auto now = std::chrono::steady_clock::now();
        ++counter;
        if (chatty) {
            std::printf("\33[2K\r");
            std::printf("batches generated: %i/%i, Time: %i [min]", counter * shard.world < num_batches ? counter * shard.world : num_batches, num_batches,
                        static_cast<int>(std::chrono::duration_cast<std::chrono::minutes>(now - begin).count()));
            std::fflush(stdout);
        }
        return C2D_OK;
    };
    try {
        int turn = 0;
        for (int batch_index = shard.rank; batch_index < num_batches; batch_index += shard.world, turn ^= 1) {
            BatchSlot& sl = slots[turn];
            if (sl.batch_index >= 0) C2D_CALL(sl.ctx, finish(sl));   // the batch enqueued two turns ago
            C2D_CALL(sl.ctx, enqueue(sl, batch_index));
        }
        for (int k = 0; k < 2; k++, turn ^= 1)                       // drain in submission order
            if (slots[turn].batch_index >= 0) C2D_CALL(slots[turn].ctx, finish(slots[turn]));
    } catch (const std::exception& e) {
        std::cerr << "error: " << e.what() << "\n";
        return EXIT_FAILURE;
    }
    if (chatty) std::cout << std::endl;
    const auto end = std::chrono::steady_clock::now();
    stats.seconds = std::chrono::duration<double>(end - begin).count();
    if (chatty) {
        std::cout << "Finished computation" << std::endl;
        std::cout << "Elapsed time: " << std::chrono::duration_cast<std::chrono::minutes>(end - begin).count() << " [min]" << std::endl;
    }
    C2D_CALL(ctx, print_json_summary("compute_collision_probability", shard, stats, counter, &link, stream));
    buffers.release();
    link.close();
    for (auto& sl : slots) sl.close();
    if (chatty) std::cout << "Done." << std::endl;
    return 0;
}
