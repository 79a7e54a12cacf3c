// GPU test: c2d_uniform_table_minstd / c2d_sqrt_f32 against the reference's own loop — ONE std::default_random_engine, default
// seeded, std::uniform_real_distribution<float> per dimension, variances first, then poses (generate_dataset.cu:279-332) —
// bit for bit.  usage: test_device_tables <num_variances> <num_poses> [shape_variance]
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#include "../../include/c2d.h"

#define OK(x)                                                                                   \
    do {                                                                                        \
        int st__ = (x);                                                                         \
        if (st__ != C2D_OK) { std::fprintf(stderr, "%s -> %d (%s)\n", #x, st__, ctx ? c2d_last_error(ctx) : ""); return 1; } \
    } while (0)

int main(int argc, char** argv)
{
    if (argc < 3) return 2;
    const size_t nv = std::strtoull(argv[1], nullptr, 0), np_ = std::strtoull(argv[2], nullptr, 0);
    const bool shape = argc > 3 && std::atoi(argv[3]) != 0;
    const float lo5[5] = {0.f, 0.f, 0.f, 0.f, 0.f}, hi5[5] = {0.3f, 0.3f, 0.3f, shape ? 0.3f : 0.f, shape ? 0.3f : 0.f};  // generate_dataset.cu:54-55
    const float lo3[3] = {0.1f, 0.1f, 0.f}, hi3[3] = {5.f, 5.f, 6.28318530718f};                                          // :56-57
    std::vector<float> ref_v(nv * 5), ref_p(np_ * 3), ref_sd(nv * 5);
    {
        std::default_random_engine gen;  // :280
        std::vector<std::uniform_real_distribution<float>> u, w;
        for (int d = 0; d < 5; d++) u.emplace_back(lo5[d], hi5[d]);
        for (size_t i = 0; i < nv; i++) for (int d = 0; d < 5; d++) ref_v[i * 5 + d] = u[d](gen);
        for (int d = 0; d < 3; d++) w.emplace_back(lo3[d], hi3[d]);
        for (size_t i = 0; i < np_; i++) for (int d = 0; d < 3; d++) ref_p[i * 3 + d] = w[d](gen);
        for (size_t i = 0; i < nv * 5; i++) ref_sd[i] = std::sqrt(ref_v[i]);
    }
    c2d_ctx* ctx = nullptr;
    OK(c2d_ctx_create(0, &ctx));
    void *d_v = nullptr, *d_p = nullptr, *d_s = nullptr;
    OK(c2d_malloc(ctx, &d_v, nv * 5 * sizeof(float) + 4));
    OK(c2d_malloc(ctx, &d_p, np_ * 3 * sizeof(float) + 4));
    OK(c2d_malloc(ctx, &d_s, nv * 5 * sizeof(float) + 4));
    OK(c2d_uniform_table_minstd(ctx, static_cast<float*>(d_v), nv, 5, lo5, hi5, 0, nullptr));
    OK(c2d_uniform_table_minstd(ctx, static_cast<float*>(d_p), np_, 3, lo3, hi3, nv * 5, nullptr));
    OK(c2d_sqrt_f32(ctx, static_cast<const float*>(d_v), static_cast<float*>(d_s), nv * 5, nullptr));
    std::vector<float> v(nv * 5), p(np_ * 3), sd(nv * 5);
    OK(c2d_memcpy_d2h(ctx, v.data(), d_v, v.size() * sizeof(float), nullptr));
    OK(c2d_memcpy_d2h(ctx, p.data(), d_p, p.size() * sizeof(float), nullptr));
    OK(c2d_memcpy_d2h(ctx, sd.data(), d_s, sd.size() * sizeof(float), nullptr));
    OK(c2d_stream_synchronize(ctx, nullptr));
    size_t bad = 0;
    for (size_t i = 0; i < v.size(); i++) bad += std::memcmp(&v[i], &ref_v[i], 4) != 0;
    for (size_t i = 0; i < p.size(); i++) bad += std::memcmp(&p[i], &ref_p[i], 4) != 0;
    for (size_t i = 0; i < sd.size(); i++) bad += std::memcmp(&sd[i], &ref_sd[i], 4) != 0;
    // bad arguments
    if (c2d_uniform_table_minstd(ctx, static_cast<float*>(d_v), 1, 9, lo5, hi5, 0, nullptr) == C2D_OK) bad++;
    if (c2d_uniform_table_minstd(ctx, nullptr, 1, 5, lo5, hi5, 0, nullptr) == C2D_OK) bad++;
    if (c2d_uniform_table_minstd(ctx, nullptr, 0, 5, lo5, hi5, 0, nullptr) != C2D_OK) bad++;
    c2d_free(ctx, d_v); c2d_free(ctx, d_p); c2d_free(ctx, d_s);
    c2d_ctx_destroy(ctx);
    std::printf("device tables: %zu variances x 5, %zu poses x 3, %zu differing floats\n", nv, np_, bad);
    return bad ? 1 : 0;
}
