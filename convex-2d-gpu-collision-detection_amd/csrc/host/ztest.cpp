// ztest — single-file variant of compute_collision_probability with the reference's command
// line (reference ztest.cu:37-101 flags, :168-444 main): explicit input / output files, optional
// --meta_dir, --cps_only (1-D output of probabilities) and a CONSTANT batch of 10000 samples per
// adaptive step (ztest.cu:332-339) instead of the 1000 / 100000 schedule.
//
// Deliberate differences: the reference's --shuffle branches are inverted (D6:
// it shuffles the array it does not write); here --shuffle shuffles what is written.  --seed,
// --device are additions.
#include "driver_common.hpp"

int main(int argc, char* argv[])
{
    std::string data_dir = "./data/", data_file_in, data_file_out, meta_dir;  // ztest.cu:37-47
    int max_samples = 4000000;
    float robot_width = 4.07f, robot_height = 1.74f;
    bool shuffle = true, cps_only = false;
    unsigned long long seed = 0;
    cli::Parser p;
    using K = cli::Option;
    p.add("help", 0, K::SWITCH, "produce help message");
    p.add("data_dir", 0, K::VALUE, "where to read the data (poses.npy, variances.npy, meta/)");
    p.add("data_file_in", 0, K::VALUE, "input .npy [N,4] (default <data_dir>/tmp/0.npy)");
    p.add("data_file_out", 0, K::VALUE, "output .npy (default <data_dir>/0.npy)");
    p.add("max_samples", 0, K::VALUE, "maximum number of samples for z-test");
    p.add("robot_width", 'w', K::VALUE, "robot width");
    p.add("robot_height", 'h', K::VALUE, "robot height");
    p.add("shuffle", 0, K::VALUE, "whether or not to shuffle data");
    p.add("cps_only", 0, K::VALUE, "whether or not to only write collision probabilities");
    p.add("meta_dir", 0, K::VALUE, "folder containing accuracy_bins.npy and bin_accuracy.npy");
    p.add("seed", 0, K::VALUE, "seed of the Monte-Carlo stream (default 0)");
    p.add("device", 0, K::VALUE, "GPU index (default: $LOCAL_RANK or 0)");
    Shard shard;
    try {
        p.parse(argc, argv);
        if (p.has("help")) { p.print_help(std::cout); return 1; }
        if (p.has("data_dir")) data_dir = p.str("data_dir");
        if (p.has("data_file_in")) data_file_in = p.str("data_file_in");
        if (p.has("data_file_out")) data_file_out = p.str("data_file_out");
        if (p.has("max_samples")) max_samples = p.integer("max_samples");
        if (p.has("robot_width")) robot_width = p.real("robot_width");
        if (p.has("robot_height")) robot_height = p.real("robot_height");
        if (p.has("shuffle")) shuffle = p.boolean("shuffle");
        if (p.has("cps_only")) cps_only = p.boolean("cps_only");
        if (p.has("meta_dir")) meta_dir = p.str("meta_dir");
        if (p.has("seed")) seed = std::stoull(p.str("seed"), nullptr, 0);
        shard = resolve_shard(p);
        if (max_samples <= 0) throw std::runtime_error("--max_samples must be positive");
    } catch (const std::exception& e) {
        std::cerr << "error: " << e.what() << "\n";
        p.print_help(std::cerr);
        return EXIT_FAILURE;
    }
    const fs::path dd = data_dir;
    if (!fs::exists(dd)) { std::cout << "Error: data_dir " << dd << " does not exist." << std::endl; return 1; }  // :180-183
    npy::Array poses, variances, scenes, accuracy_bins, bin_accuracy;
    try {
        fs::path md = meta_dir.empty() ? dd / "meta" : fs::path(meta_dir);
        if (meta_dir.empty()) {  // :184-192: write the default bins
            fs::create_directories(md);
            const float bins[4] = {0.0f, 0.01f, 0.1f, 1.0f}, acc[3] = {0.0001f, 0.001f, 0.01f};
            npy::save_f32((md / "accuracy_bins.npy").string(), {4}, bins);
            npy::save_f32((md / "bin_accuracy.npy").string(), {3}, acc);
        }
        if (data_file_in.empty()) {
            data_file_in = (dd / "tmp/0.npy").string();
            std::cout << "Using default input file: " << data_file_in << std::endl;
        }
        if (data_file_out.empty()) {
            data_file_out = (dd / "0.npy").string();
            std::cout << "Using default output file: " << data_file_out << std::endl;
        }
        if (fs::exists(data_file_out)) std::cout << "Warning: " << data_file_out << " already exists, will be overwritten" << std::endl;
        std::cout << "Reading data..." << std::endl;
        poses = npy::load_f32((dd / "poses.npy").string());
        variances = npy::load_f32((dd / "variances.npy").string());
        scenes = npy::load_f32(data_file_in);
        accuracy_bins = npy::load_f32((md / "accuracy_bins.npy").string());
        bin_accuracy = npy::load_f32((md / "bin_accuracy.npy").string());
        if (poses.data.size() % 3 || variances.data.size() % 5 || scenes.data.size() % 4) throw std::runtime_error("unexpected array shapes");
        if (accuracy_bins.data.size() < 2 || bin_accuracy.data.size() + 1 < accuracy_bins.data.size())
            throw std::runtime_error("bin_accuracy.npy needs accuracy_bins - 1 entries");
    } catch (const std::exception& e) {
        std::cout << "Error while reading numpy arrays" << std::endl << e.what() << std::endl;
        return 1;
    }
    const int num_poses = static_cast<int>(poses.data.size() / 3), num_variances = static_cast<int>(variances.data.size() / 5);
    const size_t N = scenes.data.size() / 4;
    std::cout << "num poses: " << num_poses << std::endl;
    std::cout << "num variances: " << num_variances << std::endl;
    std::cout << "num data points: " << N << std::endl;
    if (N == 0 || num_poses == 0 || num_variances == 0) { std::cerr << "error: empty input\n"; return EXIT_FAILURE; }
    std::vector<StdDev> std_devs = std_devs_from_variances(variances.data);

    CtxScope scope;   // ctx, stream and (below) the device buffers are released on every way out of main()
    C2D_CALL(scope.ctx, scope.open(shard.device));
    c2d_ctx* ctx = scope.ctx;
    c2d_stream stream = scope.stream;
    void *d_poses = nullptr, *d_sd = nullptr, *d_scenes = nullptr, *d_hits = nullptr, *d_used = nullptr, *d_rows = nullptr;
    DeviceBuffers buffers(ctx);
    buffers.own({&d_poses, &d_sd, &d_scenes, &d_hits, &d_used, &d_rows});
    C2D_CALL(ctx, c2d_malloc(ctx, &d_poses, poses.data.size() * sizeof(float)));
    C2D_CALL(ctx, c2d_malloc(ctx, &d_sd, std_devs.size() * sizeof(StdDev)));
    C2D_CALL(ctx, c2d_malloc(ctx, &d_scenes, N * sizeof(PositionWithVarAndPoseIdx)));
    C2D_CALL(ctx, c2d_malloc(ctx, &d_hits, N * sizeof(uint32_t)));
    C2D_CALL(ctx, c2d_malloc(ctx, &d_used, N * sizeof(uint32_t)));
    C2D_CALL(ctx, c2d_malloc(ctx, &d_rows, N * sizeof(PoseCPVarAndPoseIdx)));
    C2D_CALL(ctx, c2d_memcpy_h2d(ctx, d_poses, poses.data.data(), poses.data.size() * sizeof(float), stream));
    C2D_CALL(ctx, c2d_memcpy_h2d(ctx, d_sd, std_devs.data(), std_devs.size() * sizeof(StdDev), stream));
    C2D_CALL(ctx, c2d_memcpy_h2d(ctx, d_scenes, scenes.data.data(), N * sizeof(PositionWithVarAndPoseIdx), stream));
    const auto begin = std::chrono::steady_clock::now();
    std::cout << "Begin computation..." << std::endl;
    c2d_mc_scenes_args m{};
    m.d_poses = static_cast<const Pose*>(d_poses); m.num_poses = num_poses;
    m.d_std_devs = static_cast<const StdDev*>(d_sd); m.num_std_devs = num_variances;
    m.d_scenes = static_cast<const PositionWithVarAndPoseIdx*>(d_scenes); m.n_scenes = N;
    m.robot_w = robot_width; m.robot_h = robot_height;
    m.accuracy_bins = accuracy_bins.data.data(); m.bin_accuracy = bin_accuracy.data.data();
    m.n_accuracy_bins = static_cast<uint32_t>(accuracy_bins.data.size());
    m.max_samples = static_cast<uint32_t>(max_samples);
    m.seed = seed; m.scene_id_base = 0;
    m.schedule_small_batch = m.schedule_large_batch = 10000; m.schedule_switch_at = 0;  // ztest.cu:332
    m.d_hits = static_cast<uint32_t*>(d_hits); m.d_n_used = static_cast<uint32_t*>(d_used);
    m.d_rows = static_cast<PoseCPVarAndPoseIdx*>(d_rows);
    uint64_t total = 0;
    m.total_samples = &total;
    C2D_CALL(ctx, c2d_mc_scenes(ctx, &m, stream));
    std::vector<PoseCPVarAndPoseIdx> dataset(N);
    C2D_CALL(ctx, c2d_memcpy_d2h(ctx, dataset.data(), d_rows, N * sizeof(PoseCPVarAndPoseIdx), stream));
    C2D_CALL(ctx, c2d_stream_synchronize(ctx, stream));
    RunStats stats;
    for (const auto& r : dataset) stats.add_cp(r.cp);
    try {
        if (cps_only) {  // ztest.cu:390-396, :416-418
            std::vector<float> cps(N);
            for (size_t i = 0; i < N; i++) cps[i] = dataset[i].cp;
            if (shuffle) std::shuffle(cps.begin(), cps.end(), std::default_random_engine(0));
            npy::save_f32(data_file_out, {N}, cps.data());
        } else {
            if (shuffle) std::shuffle(dataset.begin(), dataset.end(), std::default_random_engine(0));
            npy::save_f32(data_file_out, {N, 5}, reinterpret_cast<const float*>(dataset.data()));
        }
    } catch (const std::exception& e) {
        std::cerr << "error: " << e.what() << "\n";
        return EXIT_FAILURE;
    }
    const auto end = std::chrono::steady_clock::now();
    stats.samples = total; stats.scenes = N; stats.seconds = std::chrono::duration<double>(end - begin).count();
    std::cout << "Finished computation" << std::endl;
    std::cout << "Elapsed time: " << std::chrono::duration_cast<std::chrono::minutes>(end - begin).count() << " [min]" << std::endl;
    C2D_CALL(ctx, print_json_summary("ztest", shard, stats, 1, nullptr, stream));
    buffers.release();
    scope.close();
    std::cout << "Done." << std::endl;
    return 0;
}
