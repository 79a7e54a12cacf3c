// CPU unit test of the drivers' host helpers (csrc/host/npy.hpp, cli.hpp, driver_common.hpp).
// Compiled and run by tests/test_host_helpers.py; exits non-zero on the first failed check.
#include <cstdio>
#include <cstdlib>
#include <fstream>

#include "driver_common.hpp"

#define CHECK(cond)                                                              \
    do {                                                                         \
        if (!(cond)) { std::fprintf(stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond); return 1; } \
    } while (0)

static int test_npy(const std::string& dir)
{
    // 1-D, 2-D and empty arrays round-trip; the header block is a multiple of 64 bytes
    for (size_t rows : {size_t(0), size_t(1), size_t(7), size_t(1000)}) {
        std::vector<float> v(rows * 5);
        for (size_t i = 0; i < v.size(); i++) v[i] = 0.25f * static_cast<float>(i) - 3.0f;
        const std::string p = dir + "/a" + std::to_string(rows) + ".npy";
        npy::save_f32(p, {rows, 5}, v.data());
        npy::Array a = npy::load_f32(p);
        CHECK(a.shape.size() == 2 && a.shape[0] == rows && a.shape[1] == 5 && a.data == v);
        std::ifstream f(p, std::ios::binary);
        unsigned char pre[10];
        f.read(reinterpret_cast<char*>(pre), 10);
        const size_t hl = pre[8] | (size_t(pre[9]) << 8);
        CHECK((10 + hl) % 64 == 0);
    }
    std::vector<float> one = {1.5f, -2.5f, 3.5f};
    npy::save_f32(dir + "/one.npy", {3}, one.data());
    npy::Array a = npy::load_f32(dir + "/one.npy");
    CHECK(a.shape.size() == 1 && a.shape[0] == 3 && a.rows() == 3 && a.cols() == 1 && a.data == one);
    // rejects what it cannot represent
    bool threw = false;
    try { npy::load_f32(dir + "/missing.npy"); } catch (const std::exception&) { threw = true; }
    CHECK(threw);
    { std::ofstream bad(dir + "/bad.npy", std::ios::binary); bad << "not an npy file at all"; }
    threw = false;
    try { npy::load_f32(dir + "/bad.npy"); } catch (const std::exception&) { threw = true; }
    CHECK(threw);
    // truncated payload
    { std::ifstream in(dir + "/a1000.npy", std::ios::binary); std::string all((std::istreambuf_iterator<char>(in)), {});
      std::ofstream out(dir + "/trunc.npy", std::ios::binary); out.write(all.data(), static_cast<std::streamsize>(all.size() - 100)); }
    threw = false;
    try { npy::load_f32(dir + "/trunc.npy"); } catch (const std::exception&) { threw = true; }
    CHECK(threw);
    return 0;
}

static int test_cli()
{
    using K = cli::Option;
    cli::Parser p;
    p.add("num_batches", 'n', K::VALUE, "");
    p.add("shape_variance", 0, K::SWITCH, "");
    p.add("min_pose", 0, K::MULTI, "");
    p.add("robot_width", 'w', K::VALUE, "");
    p.add("data_dir", 0, K::VALUE, "");
    const char* argv[] = {"prog", "-n", "7", "--min_pose", "0.5", "-0.25", "1e-3", "--shape_variance", "--robot_width=4.5",
                          "--data_dir", "/tmp/x y"};
    p.parse(11, const_cast<char**>(argv));
    CHECK(p.integer("num_batches") == 7);
    CHECK(p.has("shape_variance") && !p.has("nope"));
    std::vector<float> mp = p.reals("min_pose");
    CHECK(mp.size() == 3 && mp[0] == 0.5f && mp[1] == -0.25f && mp[2] == 1e-3f);  // negative numbers are values, not flags
    CHECK(p.real("robot_width") == 4.5f && p.str("data_dir") == "/tmp/x y");
    cli::Parser q;
    q.add("a", 0, K::VALUE, "");
    bool threw = false;
    const char* bad1[] = {"prog", "--b", "1"};
    try { q.parse(3, const_cast<char**>(bad1)); } catch (const std::exception&) { threw = true; }
    CHECK(threw);
    threw = false;
    const char* bad2[] = {"prog", "--a"};
    try { q.parse(2, const_cast<char**>(bad2)); } catch (const std::exception&) { threw = true; }
    CHECK(threw);
    return 0;
}

static int test_common(const std::string& dir)
{
    // get_num_batches_in_dir counts .npy files whose name starts with an integer (reference utils.cu:36-56)
    const std::string d = dir + "/batches";
    mkdirs(d + "/meta");
    float z = 0;
    for (const char* n : {"0.npy", "1.npy", "12.npy", "poses.npy", "variances.npy", "3abc.npy", "x7.npy", "5.txt"}) {
        if (std::string(n).find(".npy") != std::string::npos) npy::save_f32(d + "/" + n, {1}, &z);
        else { std::ofstream f(d + "/" + n); f << "x"; }
    }
    CHECK(get_num_batches_in_dir(d) == 4);  // 0, 1, 12, 3abc
    CHECK(get_num_batches_in_dir(dir + "/does_not_exist") == 0);
    std::vector<float> var = {0.04f, 0.09f, 0.25f, 0.0f, 1.0f};
    std::vector<StdDev> sd = std_devs_from_variances(var);
    CHECK(sd.size() == 1 && sd[0].x == 0.2f && sd[0].y == 0.3f && sd[0].theta == 0.5f && sd[0].width == 0.0f && sd[0].height == 1.0f);
    RunStats st;
    for (float cp : {0.0f, 0.0005f, 0.001f, 0.0099f, 0.01f, 0.05f, 0.1f, 0.5f, 1.0f}) st.add_cp(cp);
    // numpy.histogram(cp, [0,.001,.01,.1,1]) on the float32 values: 0.001f, 0.01f, 0.1f are just above/below their double edges
    CHECK(st.cp_hist[0] == 2 && st.cp_hist[1] == 3 && st.cp_hist[2] == 1 && st.cp_hist[3] == 3);
    float r[8];
    create_rect(r, 4.0f, 2.0f);
    CHECK(r[0] == -2 && r[1] == -1 && r[2] == 2 && r[3] == -1 && r[4] == 2 && r[5] == 1 && r[6] == -2 && r[7] == 1);
    return 0;
}

int main(int argc, char** argv)
{
    if (argc < 2) return 2;
    const std::string dir = argv[1];
    if (test_npy(dir)) return 1;
    if (test_cli()) return 1;
    if (test_common(dir)) return 1;
    std::puts("host helpers ok");
    return 0;
}
