// generate_dataset — host driver with the reference's command-line surface and .npy outputs
// (reference generate_dataset.cu:44-169 flags, :255-524 main), driving the HIP kernels through
// the C-ABI of include/c2d.h.  Plain C++17: no HIP headers, no Boost, no libnpy.
//
// Differences from the reference, all deliberate:
//   - directories are created before poses.npy / variances.npy are saved (D7);
//   - the scene sampler and the Monte-Carlo loop use the counter-based stream of c2d.h, keyed by
//     --seed (default: time based, as the reference's srand(time(0))) and the global scene index,
//     so a run is reproducible and independent of how batches are sharded over GPUs;
//   - --seed, --rank / --world_size / --device are additions; every reference flag is kept.
#include "driver_common.hpp"

struct Arguments {  // defaults: generate_dataset.cu:44-64
    std::string data_dir = "./data/";
    std::string pose_dir = "";
    std::string variance_dir = "";
    int num_batches = 100;
    int batch_size = 100000;
    int start_batch_count = 0;
    int num_poses = 64 * 64 * 64 * 64;
    int num_variances = 64 * 64 * 64 * 64;
    int max_samples = 4000000;
    std::vector<float> min_variance = {0.0, 0.0, 0.0, 0.0, 0.0};
    std::vector<float> max_variance = {0.3, 0.3, 0.3, 0.3, 0.3};
    std::vector<float> min_pose = {0.1, 0.1, 0.0};
    std::vector<float> max_pose = {5, 5, static_cast<float>(2 * M_PI)};
    std::vector<float> accuracy_bins = {0.0, 0.01, 0.1, 1.0};
    std::vector<float> bin_accuracy = {0.0001, 0.001, 0.01};
    float robot_width = 4.07;
    float robot_height = 1.74;
    float spread = 4;
    bool shape_variance = false;
    unsigned long long seed = 0;
    bool seed_given = false;
};

static void need_size(const std::vector<float>& v, size_t n, const char* name)
{
    if (v.size() != n) throw std::runtime_error(std::string("--") + name + " needs exactly " + std::to_string(n) + " values");
}

int main(int argc, char* argv[])
{
    Arguments a;
    cli::Parser p;
    using K = cli::Option;
    p.add("help", 0, K::SWITCH, "produce help message");
    p.add("data_dir", 0, K::VALUE, "where to store the data");
    p.add("num_batches", 'n', K::VALUE, "number of batches");
    p.add("batch_size", 'b', K::VALUE, "number of samples per batch");
    p.add("start_batch_count", 's', K::VALUE, "start value for batches");
    p.add("num_poses", 0, K::VALUE, "number of poses");
    p.add("num_variances", 0, K::VALUE, "number of variances");
    p.add("shape_variance", 0, K::SWITCH, "whether or not to have shape variance");
    p.add("max_samples", 0, K::VALUE, "maximum number of samples for z-test");
    p.add("accuracy_bins", 0, K::MULTI, "accuracy bins e.g. 0 0.01 0.1 1");
    p.add("bin_accuracy", 0, K::MULTI, "accuracy for each bin e.g. 0.0001 0.001 0.01");
    p.add("min_variance", 0, K::MULTI, "min variance for each dimension (x y theta width height)");
    p.add("max_variance", 0, K::MULTI, "max variance for each dimension");
    p.add("min_pose", 0, K::MULTI, "min pose for each dimension (width height theta)");
    p.add("max_pose", 0, K::MULTI, "max pose for each dimension");
    p.add("robot_width", 'w', K::VALUE, "robot width");
    p.add("robot_height", 'h', K::VALUE, "robot height");
    p.add("spread", 0, K::VALUE, "spread of poses");
    p.add("pose_dir", 0, K::VALUE, "poses .npy file to load instead of sampling");
    p.add("variance_dir", 0, K::VALUE, "variances .npy file to load instead of sampling");
    p.add("seed", 0, K::VALUE, "seed of the scene / Monte-Carlo streams (default: time based)");
    add_shard_options(p);
    Shard shard;
    int gpus = 1;
    try {
        p.parse(argc, argv);
        if (p.has("help")) { p.print_help(std::cout); return 1; }
        if (p.has("data_dir")) a.data_dir = p.str("data_dir");
        if (p.has("num_batches")) a.num_batches = p.integer("num_batches");
        if (p.has("batch_size")) a.batch_size = p.integer("batch_size");
        if (p.has("start_batch_count")) a.start_batch_count = p.integer("start_batch_count");
        if (p.has("num_poses")) a.num_poses = p.integer("num_poses");
        if (p.has("num_variances")) a.num_variances = p.integer("num_variances");
        if (p.has("max_samples")) a.max_samples = p.integer("max_samples");
        if (p.has("accuracy_bins")) a.accuracy_bins = p.reals("accuracy_bins");
        if (p.has("bin_accuracy")) a.bin_accuracy = p.reals("bin_accuracy");
        if (p.has("min_variance")) { a.min_variance = p.reals("min_variance"); need_size(a.min_variance, 5, "min_variance"); }
        if (p.has("max_variance")) { a.max_variance = p.reals("max_variance"); need_size(a.max_variance, 5, "max_variance"); }
        if (p.has("min_pose")) { a.min_pose = p.reals("min_pose"); need_size(a.min_pose, 3, "min_pose"); }
        if (p.has("max_pose")) { a.max_pose = p.reals("max_pose"); need_size(a.max_pose, 3, "max_pose"); }
        if (p.has("robot_width")) a.robot_width = p.real("robot_width");
        if (p.has("robot_height")) a.robot_height = p.real("robot_height");
        if (p.has("spread")) a.spread = p.real("spread");
        if (p.has("shape_variance")) a.shape_variance = true;
        if (p.has("pose_dir")) a.pose_dir = p.str("pose_dir");
        if (p.has("variance_dir")) a.variance_dir = p.str("variance_dir");
        if (p.has("seed")) { a.seed = std::stoull(p.str("seed"), nullptr, 0); a.seed_given = true; }
        if (p.has("gpus")) gpus = p.integer("gpus");
        shard = resolve_shard(p);
        if (gpus < 1) throw std::runtime_error("--gpus must be at least 1");
        if (gpus > 1 && shard.from_env) throw std::runtime_error("--gpus N starts the ranks itself: do not combine it with --rank / --world_size or a launcher");
        if (a.accuracy_bins.size() < 2 || a.bin_accuracy.size() + 1 != a.accuracy_bins.size())
            throw std::runtime_error("--bin_accuracy needs one value fewer than --accuracy_bins");
        if (a.batch_size <= 0 || a.num_batches < 0 || a.max_samples <= 0) throw std::runtime_error("sizes must be positive");
    } catch (const std::exception& e) {
        std::cerr << "error: " << e.what() << "\n";
        p.print_help(std::cerr);
        return EXIT_FAILURE;
    }
    // The seed is part of every scene's random stream: all ranks must use the same one.  The default (time based, as
    // the reference's srand(time(0)), :406) is therefore resolved ONCE — here by the launcher or the single process;
    // externally launched ranks take rank 0's through the aggregation link (below) or must pass --seed.
    if (!a.seed_given) {
        if (shard.world > 1 && shard.id_file.empty()) {
            std::cerr << "error: ranks launched by hand need --seed (or --dist_id_file / $C2D_DIST_ID_FILE, or use --gpus N)\n";
            return EXIT_FAILURE;
        }
        a.seed = static_cast<unsigned long long>(std::time(nullptr));
    }
    if (gpus > 1) return launch_ranks(gpus, argc, argv, {"--seed", std::to_string(a.seed)});
    const bool chatty = shard.rank == 0;  // one rank narrates
    const std::string& data_dir = a.data_dir;
    if (chatty) {
        std::cout << "data dir: " << data_dir << std::endl;
        std::cout << "num batches: " << a.num_batches << std::endl;
        std::cout << "num batch: " << a.batch_size << std::endl;
        std::cout << "start batch count: " << a.start_batch_count << std::endl;
    }

    try {
        mkdirs(data_dir);            // reference creates these only after saving (D7)
        mkdirs(data_dir + "/meta");
    } catch (const std::exception& e) {
        std::cerr << "error: cannot create " << data_dir << ": " << e.what() << "\n";
        return EXIT_FAILURE;
    }

    PhaseClock clock;
    const size_t B = static_cast<size_t>(a.batch_size);
    BatchSlot slots[2];
    for (auto& sl : slots) C2D_CALL(sl.ctx, sl.open(shard.device, B));
    c2d_ctx* ctx = slots[0].ctx;          // owner of the tables and of the aggregation link
    c2d_stream stream = slots[0].stream;
    clock.lap("device_open");
    DistLink link;
    C2D_CALL(ctx, link.open(ctx, shard));
    if (link.active()) {  // rank 0's seed is everyone's
        unsigned long long w[1] = {a.seed};
        C2D_CALL(ctx, link.broadcast(w, 1, stream));
        a.seed = w[0];
    }
    if (chatty) std::cout << "seed: " << a.seed << std::endl;

    // Tables.  Same generator, distributions and draw order as the reference (generate_dataset.cu:279-332):
    // std::default_random_engine, default seeded, all variances first, then all poses — drawn ON THE DEVICE
    // (c2d_uniform_table_minstd: the engine can jump, every lane starts at its own draw), bit-identical to a libstdc++
    // run of the reference's serial loop.  Nothing is drawn or uploaded per rank on the host; rank 0 saves the two
    // tables (:300-332) from a download that runs beside the batches.
    void *d_var = nullptr, *d_poses = nullptr, *d_sd = nullptr;
    DeviceBuffers buffers(ctx);   // (declared after the slots that own the ctx and before the saver thread that reads the tables:
    buffers.own({&d_var, &d_poses, &d_sd});   //  on every way out the thread is joined first, the tables go next, the ctx last)
    bool var_made = false, poses_made = false;
    try {
        if (a.variance_dir.empty()) {
            if (!a.shape_variance) {
                a.min_variance[3] = a.max_variance[3] = 0.0f;
                a.min_variance[4] = a.max_variance[4] = 0.0f;
            }
            var_made = true;
        } else {
            npy::Array v = npy::load_f32(a.variance_dir);
            if (v.data.size() % 5) throw std::runtime_error("variances file is not [N,5]");
            a.num_variances = static_cast<int>(v.data.size() / 5);
            if (a.num_variances > 0) {
                C2D_CALL(ctx, c2d_malloc(ctx, &d_var, v.data.size() * sizeof(float)));
                C2D_CALL(ctx, c2d_memcpy_h2d(ctx, d_var, v.data.data(), v.data.size() * sizeof(float), stream));
                C2D_CALL(ctx, c2d_stream_synchronize(ctx, stream));  // (the host array is a local)
            }
        }
        if (a.pose_dir.empty()) {
            poses_made = true;
        } else {
            npy::Array v = npy::load_f32(a.pose_dir);
            if (v.data.size() % 3) throw std::runtime_error("poses file is not [N,3]");
            a.num_poses = static_cast<int>(v.data.size() / 3);
            if (a.num_poses > 0) {
                C2D_CALL(ctx, c2d_malloc(ctx, &d_poses, v.data.size() * sizeof(float)));
                C2D_CALL(ctx, c2d_memcpy_h2d(ctx, d_poses, v.data.data(), v.data.size() * sizeof(float), stream));
                C2D_CALL(ctx, c2d_stream_synchronize(ctx, stream));
            }
        }
        if (shard.rank == 0) {
            npy::save_f32(data_dir + "/meta/accuracy_bins.npy", {a.accuracy_bins.size()}, a.accuracy_bins.data());
            npy::save_f32(data_dir + "/meta/bin_accuracy.npy", {a.bin_accuracy.size()}, a.bin_accuracy.data());
        }
    } catch (const std::exception& e) {
        std::cerr << "error: " << e.what() << "\n";
        return EXIT_FAILURE;
    }
    if (a.num_poses <= 0 || a.num_variances <= 0) { std::cerr << "error: empty pose / variance table\n"; return EXIT_FAILURE; }
    const size_t nv = static_cast<size_t>(a.num_variances), np_ = static_cast<size_t>(a.num_poses);
    if (var_made) {
        C2D_CALL(ctx, c2d_malloc(ctx, &d_var, nv * 5 * sizeof(float)));
        C2D_CALL(ctx, c2d_uniform_table_minstd(ctx, static_cast<float*>(d_var), nv, 5, a.min_variance.data(), a.max_variance.data(), 0, stream));
    }
    if (poses_made) {
        C2D_CALL(ctx, c2d_malloc(ctx, &d_poses, np_ * 3 * sizeof(float)));
        // the engine has made 5 * num_variances calls when the pose loop starts — only if the variance loop ran (:282, :321)
        C2D_CALL(ctx, c2d_uniform_table_minstd(ctx, static_cast<float*>(d_poses), np_, 3, a.min_pose.data(), a.max_pose.data(), var_made ? nv * 5 : 0, stream));
    }
    C2D_CALL(ctx, c2d_malloc(ctx, &d_sd, nv * sizeof(StdDev)));
    C2D_CALL(ctx, c2d_sqrt_f32(ctx, static_cast<const float*>(d_var), static_cast<float*>(d_sd), nv * 5, stream));  // :309-317
    C2D_CALL(ctx, c2d_stream_synchronize(ctx, stream));
    clock.lap("tables_on_device");
    // rank 0 saves the generated tables: its own thread, ctx and stream, page-locked staging, beside the batches
    double table_save_s = 0.0;
    std::string table_save_error;
    std::thread table_saver;
    struct JoinOnExit { std::thread& t; ~JoinOnExit() { if (t.joinable()) t.join(); } } join_saver{table_saver};  // (every return path)
    if (shard.rank == 0 && (var_made || poses_made)) {
        table_saver = std::thread([&, nv, np_]() {
            const auto t0 = std::chrono::steady_clock::now();
            c2d_ctx* c2 = nullptr;
            c2d_stream s2 = nullptr;
            void* h = nullptr;
            auto step = [&](int st, const char* what) {
                if (st != C2D_OK && table_save_error.empty()) table_save_error = std::string(what) + ": " + (c2 ? c2d_last_error(c2) : c2d_status_string(st));
                return st == C2D_OK;
            };
            try {
                const size_t floats = std::max(var_made ? nv * 5 : 0, poses_made ? np_ * 3 : 0);
                if (step(c2d_ctx_create(shard.device, &c2), "c2d_ctx_create") && step(c2d_stream_create(c2, &s2), "c2d_stream_create") &&
                    step(c2d_malloc_host(c2, &h, floats * sizeof(float)), "c2d_malloc_host")) {
                    if (var_made && step(c2d_memcpy_d2h(c2, h, d_var, nv * 5 * sizeof(float), s2), "download") && step(c2d_stream_synchronize(c2, s2), "download"))
                        npy::save_f32(data_dir + "/variances.npy", {nv, 5}, static_cast<const float*>(h));
                    if (poses_made && step(c2d_memcpy_d2h(c2, h, d_poses, np_ * 3 * sizeof(float), s2), "download") && step(c2d_stream_synchronize(c2, s2), "download"))
                        npy::save_f32(data_dir + "/poses.npy", {np_, 3}, static_cast<const float*>(h));
                }
            } catch (const std::exception& e) {
                if (table_save_error.empty()) table_save_error = e.what();
            }
            if (h) c2d_free_host(c2, h);
            if (s2) c2d_stream_destroy(c2, s2);
            if (c2) c2d_ctx_destroy(c2);
            table_save_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        });
    }
    if (chatty) {
        std::cout << "num poses: " << a.num_poses << std::endl;
        std::cout << "num variances: " << a.num_variances << std::endl;
    }
    double wait_gpu_s = 0.0, host_batch_s = 0.0;  // inside the batch loop: waiting for a batch's stream / statistics + shuffle + file

    const auto begin = std::chrono::steady_clock::now();
    if (chatty) {
        std::cout << "Total number of configurations: " << static_cast<long long>(a.batch_size) * a.num_batches << std::endl;
        std::cout << "Begin computation..." << std::endl;
    }
    int counter = 0;
    RunStats stats;
    if (chatty) std::printf("batches generated: %i/%i", counter, a.num_batches);

    // everything of one batch that runs on the GPU, enqueued without waiting
    auto enqueue = [&](BatchSlot& sl, int batch_index) -> int {
        const uint64_t scene_base = (static_cast<uint64_t>(a.start_batch_count) + batch_index) * B;
        // iteration == 0 branch of the reference kernel: draw the scenes (:207-219)
        int st = c2d_sample_scenes(sl.ctx, static_cast<const Pose*>(d_poses), a.num_poses, static_cast<const StdDev*>(d_sd),
                                   a.num_variances, a.robot_width, a.robot_height, a.spread, a.seed, scene_base, B,
                                   static_cast<PositionWithVarAndPoseIdx*>(sl.d_scenes), sl.stream);
        if (st != C2D_OK) return st;
        c2d_mc_scenes_args m{};
        m.d_poses = static_cast<const Pose*>(d_poses); m.num_poses = a.num_poses;
        m.d_std_devs = static_cast<const StdDev*>(d_sd); m.num_std_devs = a.num_variances;
        m.d_scenes = static_cast<const PositionWithVarAndPoseIdx*>(sl.d_scenes); m.n_scenes = B;
        m.robot_w = a.robot_width; m.robot_h = a.robot_height;
        m.accuracy_bins = a.accuracy_bins.data(); m.bin_accuracy = a.bin_accuracy.data();
        m.n_accuracy_bins = static_cast<uint32_t>(a.accuracy_bins.size());
        m.max_samples = static_cast<uint32_t>(a.max_samples);
        m.seed = a.seed; m.scene_id_base = scene_base;
        m.d_hits = static_cast<uint32_t*>(sl.d_hits); m.d_n_used = static_cast<uint32_t*>(sl.d_used);
        m.d_rows = static_cast<PoseCPVarAndPoseIdx*>(sl.d_rows);
        st = c2d_mc_scenes(sl.ctx, &m, sl.stream);  // adaptive loop, :425-479 (no host output requested: asynchronous)
        if (st == C2D_OK) st = sl.download();
        sl.batch_index = batch_index;
        return st;
    };
    // the host side of a batch: wait for its stream, statistics, shuffle, file
    auto finish = [&](BatchSlot& sl) -> int {
        const auto w0 = std::chrono::steady_clock::now();
        int st = c2d_stream_synchronize(sl.ctx, sl.stream);
        const auto w1 = std::chrono::steady_clock::now();
        wait_gpu_s += std::chrono::duration<double>(w1 - w0).count();
        if (st != C2D_OK) return st;
        stats.scenes += B;
        for (size_t i = 0; i < B; i++) stats.samples += sl.used[i];
        for (size_t i = 0; i < B; i++) stats.hits += sl.hits[i];
        for (size_t i = 0; i < B; i++) stats.add_cp(sl.dataset[i].cp);
        std::shuffle(sl.dataset, sl.dataset + B, std::default_random_engine(0));  // :496
        npy::save_f32(data_dir + "/" + std::to_string(a.start_batch_count + sl.batch_index) + ".npy", {B, 5},
                      reinterpret_cast<const float*>(sl.dataset));  // :499-500
        sl.batch_index = -1;
        const auto now = std::chrono::steady_clock::now();
        host_batch_s += std::chrono::duration<double>(now - w1).count();
        ++counter;
        if (chatty) {
            std::printf("\33[2K\r");
            std::printf("batches generated: %i/%i, Time: %i [min]", counter * shard.world < a.num_batches ? counter * shard.world : a.num_batches,
                        a.num_batches, static_cast<int>(std::chrono::duration_cast<std::chrono::minutes>(now - begin).count()));
            std::fflush(stdout);
        }
        return C2D_OK;
    };
    try {
        int turn = 0;
        for (int batch_index = shard.rank; batch_index < a.num_batches; batch_index += shard.world, turn ^= 1) {
            BatchSlot& sl = slots[turn];
            if (sl.batch_index >= 0) C2D_CALL(sl.ctx, finish(sl));   // the batch enqueued two turns ago
            C2D_CALL(sl.ctx, enqueue(sl, batch_index));
        }
        for (int k = 0; k < 2; k++, turn ^= 1)                       // drain in submission order
            if (slots[turn].batch_index >= 0) C2D_CALL(slots[turn].ctx, finish(slots[turn]));
    } catch (const std::exception& e) {
        std::cerr << "\nerror: " << e.what() << "\n";
        return EXIT_FAILURE;
    }
    if (chatty) std::cout << std::endl;
    const auto end = std::chrono::steady_clock::now();
    stats.seconds = std::chrono::duration<double>(end - begin).count();
    if (chatty) {
        std::cout << "Finished computation" << std::endl;
        std::cout << "Elapsed time: " << std::chrono::duration_cast<std::chrono::minutes>(end - begin).count() << " [min]" << std::endl;
    }
    clock.lap("batches");
    if (table_saver.joinable()) {
        table_saver.join();
        clock.lap("tables_save_npy_after_the_batches");
        clock.add("tables_save_npy_thread", table_save_s);
        if (!table_save_error.empty()) { std::cerr << "error: saving the tables: " << table_save_error << "\n"; return EXIT_FAILURE; }
    }
    clock.add("batches_waiting_for_gpu", wait_gpu_s);
    clock.add("batches_host_stats_shuffle_npy", host_batch_s);
    C2D_CALL(ctx, print_json_summary("generate_dataset", shard, stats, counter, &link, stream, clock.json()));
    buffers.release();
    link.close();
    for (auto& sl : slots) sl.close();
    if (chatty) std::cout << "Done." << std::endl;
    return 0;
}
