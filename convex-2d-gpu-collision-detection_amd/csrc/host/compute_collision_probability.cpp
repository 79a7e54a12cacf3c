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
// Deliberate differences: hit counters are zeroed per batch (D4); --seed,
// --rank / --world_size / --device are additions.
#include "driver_common.hpp"

struct Arguments {  // defaults: compute_collision_probability.cu:35-42
    std::string data_in = "./data_in/";
    std::string data_out = "./data_out/";
    int max_samples = 4000000;
    float robot_width = 4.07;
    float robot_height = 1.74;
    bool shuffle = true;
    unsigned long long seed = 0;
};

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
    p.add("rank", 0, K::VALUE, "this process' shard index (default: $RANK or 0)");
    p.add("world_size", 0, K::VALUE, "number of shards = GPUs (default: $WORLD_SIZE or 1)");
    p.add("device", 0, K::VALUE, "GPU index (default: $LOCAL_RANK or rank)");
    Shard shard;
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
        shard = resolve_shard(p);
        if (a.max_samples <= 0) throw std::runtime_error("--max_samples must be positive");
    } catch (const std::exception& e) {
        std::cerr << "error: " << e.what() << "\n";
        p.print_help(std::cerr);
        return EXIT_FAILURE;
    }
    const int start_batch_count = get_num_batches_in_dir(a.data_out);  // :157
    const int num_batches = get_num_batches_in_dir(a.data_in);         // :158

    std::cout << "Reading data..." << std::endl;
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
    std::cout << "num poses: " << num_poses << std::endl;
    std::cout << "num variances: " << num_variances << std::endl;
    std::cout << "num data points: " << N << std::endl;
    if (num_batches == 0 || N == 0) { std::cout << "nothing to do" << std::endl; return 0; }
    if (num_poses == 0 || num_variances == 0) { std::cerr << "error: empty pose / variance table\n"; return EXIT_FAILURE; }
    std::vector<StdDev> std_devs = std_devs_from_variances(variances.data);

    c2d_ctx* ctx = nullptr;
    C2D_CALL(ctx, c2d_ctx_create(shard.device, &ctx));
    c2d_stream stream = nullptr;
    C2D_CALL(ctx, c2d_stream_create(ctx, &stream));
    void *d_poses = nullptr, *d_sd = nullptr, *d_scenes = nullptr, *d_hits = nullptr, *d_used = nullptr, *d_rows = nullptr;
    C2D_CALL(ctx, c2d_malloc(ctx, &d_poses, poses.data.size() * sizeof(float)));
    C2D_CALL(ctx, c2d_malloc(ctx, &d_sd, std_devs.size() * sizeof(StdDev)));
    C2D_CALL(ctx, c2d_malloc(ctx, &d_scenes, N * sizeof(PositionWithVarAndPoseIdx)));
    C2D_CALL(ctx, c2d_malloc(ctx, &d_hits, N * sizeof(uint32_t)));
    C2D_CALL(ctx, c2d_malloc(ctx, &d_used, N * sizeof(uint32_t)));
    C2D_CALL(ctx, c2d_malloc(ctx, &d_rows, N * sizeof(PoseCPVarAndPoseIdx)));
    C2D_CALL(ctx, c2d_memcpy_h2d(ctx, d_poses, poses.data.data(), poses.data.size() * sizeof(float), stream));
    C2D_CALL(ctx, c2d_memcpy_h2d(ctx, d_sd, std_devs.data(), std_devs.size() * sizeof(StdDev), stream));
    C2D_CALL(ctx, c2d_stream_synchronize(ctx, stream));

    std::vector<PoseCPVarAndPoseIdx> dataset(N);
    std::vector<uint32_t> hits(N);
    const auto begin = std::chrono::steady_clock::now();
    std::cout << "Total number of configurations: " << static_cast<long long>(N) * num_batches << std::endl;
    std::cout << "Begin computation..." << std::endl;
    int counter = 0;
    RunStats stats;
    std::printf("batches generated: %i/%i\n", counter, num_batches);
    for (int batch_index = shard.rank; batch_index < num_batches; batch_index += shard.world) {
        npy::Array batch;
        try {
            batch = npy::load_f32(a.data_in + "/" + std::to_string(batch_index) + ".npy");  // :261
        } catch (const std::exception& e) {
            std::cerr << "error: " << e.what() << "\n";
            return EXIT_FAILURE;
        }
        if (batch.data.size() != N * 4) {
            std::cerr << "error: " << batch_index << ".npy does not hold " << N << " rows of 4\n";
            return EXIT_FAILURE;
        }
        // the [N,4] rows are PositionWithVarAndPoseIdx records: upload them as they are (:262-274)
        C2D_CALL(ctx, c2d_memcpy_h2d(ctx, d_scenes, batch.data.data(), N * sizeof(PositionWithVarAndPoseIdx), stream));
        c2d_mc_scenes_args m{};
        m.d_poses = static_cast<const Pose*>(d_poses); m.num_poses = num_poses;
        m.d_std_devs = static_cast<const StdDev*>(d_sd); m.num_std_devs = num_variances;
        m.d_scenes = static_cast<const PositionWithVarAndPoseIdx*>(d_scenes); m.n_scenes = N;
        m.robot_w = a.robot_width; m.robot_h = a.robot_height;
        m.accuracy_bins = accuracy_bins.data.data(); m.bin_accuracy = bin_accuracy.data.data();
        m.n_accuracy_bins = static_cast<uint32_t>(accuracy_bins.data.size());
        m.max_samples = static_cast<uint32_t>(a.max_samples);
        m.seed = a.seed;
        m.scene_id_base = (static_cast<uint64_t>(start_batch_count) + batch_index) * N;
        m.d_hits = static_cast<uint32_t*>(d_hits); m.d_n_used = static_cast<uint32_t*>(d_used);
        m.d_rows = static_cast<PoseCPVarAndPoseIdx*>(d_rows);
        uint64_t total = 0;
        m.total_samples = &total;
        C2D_CALL(ctx, c2d_mc_scenes(ctx, &m, stream));  // adaptive loop, :276-332
        C2D_CALL(ctx, c2d_memcpy_d2h(ctx, dataset.data(), d_rows, N * sizeof(PoseCPVarAndPoseIdx), stream));
        C2D_CALL(ctx, c2d_memcpy_d2h(ctx, hits.data(), d_hits, N * sizeof(uint32_t), stream));
        C2D_CALL(ctx, c2d_stream_synchronize(ctx, stream));
        stats.samples += total;
        stats.scenes += N;
        for (uint32_t h : hits) stats.hits += h;
        for (const auto& r : dataset) stats.add_cp(r.cp);
        if (a.shuffle) std::shuffle(dataset.begin(), dataset.end(), std::default_random_engine(0));  // :346-349
        try {
            npy::save_f32(a.data_out + "/" + std::to_string(start_batch_count + batch_index) + ".npy", {N, 5},
                          reinterpret_cast<const float*>(dataset.data()));  // :353-355
        } catch (const std::exception& e) {
            std::cerr << "error: " << e.what() << "\n";
            return EXIT_FAILURE;
        }
        const auto now = std::chrono::steady_clock::now();
        std::printf("\33[2K\r");
        std::printf("batches generated: %i/%i, Time: %i [min]", ++counter, num_batches,
                    static_cast<int>(std::chrono::duration_cast<std::chrono::minutes>(now - begin).count()));
        std::fflush(stdout);
    }
    std::cout << std::endl;
    const auto end = std::chrono::steady_clock::now();
    stats.seconds = std::chrono::duration<double>(end - begin).count();
    std::cout << "Finished computation" << std::endl;
    std::cout << "Elapsed time: " << std::chrono::duration_cast<std::chrono::minutes>(end - begin).count() << " [min]" << std::endl;
    print_json_summary("compute_collision_probability", shard, stats, counter);
    for (void* ptr : {d_poses, d_sd, d_scenes, d_hits, d_used, d_rows}) c2d_free(ctx, ptr);
    c2d_stream_destroy(ctx, stream);
    c2d_ctx_destroy(ctx);
    std::cout << "Done." << std::endl;
    return 0;
}
