// Prints the raw 32-bit words of rocRAND's Philox4x32-10 device-API engine run on the
// HOST (the engine is __host__ __device__): used by tests/test_boundary.py to show that
// the c2d stream (seed, scene, sample) is rocrand_state_philox4x32_10 initialised with
// (seed, subsequence = scene, offset = 8 * sample).  TEST INFRASTRUCTURE ONLY.
// With a fifth argument "draws" it prints the six words of each sample in the Monte-Carlo loop's draw layout
// (groups of four samples, include/c2d.h), fetched through the same engine; with "normals", rocRAND's own
// Box-Muller normals of those words: the c2d normals use the same words in the same roles with the canonical
// log / sqrt / sincos forms, so they agree to rounding.
#include <rocrand/rocrand_philox4x32_10.h>
#include <rocrand/rocrand_normal.h>

#include <cstring>

#include <cstdio>
#include <cstdlib>

int main(int argc, char** argv)
{
    if (argc < 5) return 2;
    unsigned long long seed = strtoull(argv[1], nullptr, 0), scene = strtoull(argv[2], nullptr, 0),
                       sample = strtoull(argv[3], nullptr, 0);
    int n = atoi(argv[4]);
    // word `w` (0..3) of Philox block `blk` of subsequence `scene`, through rocRAND's engine
    auto word = [&](unsigned long long blk, int w) {
        rocrand_state_philox4x32_10 st;
        rocrand_init(seed, scene, 4ull * blk + (unsigned)w, &st);
        return rocrand(&st);
    };
    const bool normals = argc > 5 && !strcmp(argv[5], "normals");
    if (normals || (argc > 5 && !strcmp(argv[5], "draws"))) {
        // the group-of-four draw layout of the Monte-Carlo loop (include/c2d.h, "random stream")
        for (int i = 0; i < n; i++) {
            const unsigned long long s = sample + i, g = s >> 2;
            const int j = (int)(s & 3), h = j >> 1, o = 2 * (j & 1);
            const unsigned int w[6] = {word(8 * g + 0, j), word(8 * g + 1, j), word(8 * g + 2 + h, o), word(8 * g + 2 + h, o + 1),
                                       word(8 * g + 4 + h, o), word(8 * g + 4 + h, o + 1)};
            if (!normals) {
                printf("%u %u %u %u %u %u\n", w[0], w[1], w[2], w[3], w[4], w[5]);
                continue;
            }
            // rocRAND's own Box-Muller (what rocrand_normal2 / rocrand_normal4 apply to the words of a state)
            const float2 a = rocrand_device::detail::box_muller(w[0], w[1]);  // dx, dy (reference draw order utils.cu:146-147)
            const float2 b = rocrand_device::detail::box_muller(w[2], w[3]);  // dtheta, dw (:148-149)
            const float2 c = rocrand_device::detail::box_muller(w[4], w[5]);  // dh (:150)
            printf("%.9g %.9g %.9g %.9g %.9g\n", a.x, a.y, b.x, b.y, c.x);
        }
        return 0;
    }
    for (int i = 0; i < n; i++) {
        rocrand_state_philox4x32_10 st;
        rocrand_init(seed, scene, 8ull * (sample + i), &st);
        for (int k = 0; k < 8; k++) printf("%u%c", rocrand(&st), k == 7 ? '\n' : ' ');
    }
    return 0;
}
