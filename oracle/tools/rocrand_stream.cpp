// Prints the raw 32-bit words of rocRAND's Philox4x32-10 device-API engine run on the
// HOST (the engine is __host__ __device__): used by tests/test_boundary.py to show that
// the c2d stream (seed, scene, sample) is rocrand_state_philox4x32_10 initialised with
// (seed, subsequence = scene, offset = 8 * sample).  TEST INFRASTRUCTURE ONLY.
// With a fifth argument "normals" it prints rocRAND's own five normals of each sample instead
// (rocrand_normal4 on the first Philox block, rocrand_normal on the second): the c2d normals use the
// same words in the same roles with the canonical log / sqrt / sincos forms, so they agree to rounding.
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
    if (argc > 5 && !strcmp(argv[5], "normals")) {
        for (int i = 0; i < n; i++) {
            rocrand_state_philox4x32_10 st;
            rocrand_init(seed, scene, 8ull * (sample + i), &st);
            const float4 a = rocrand_normal4(&st);   // dx, dy, dtheta, dw (reference draw order utils.cu:146-149)
            const float e = rocrand_normal(&st);     // dh (:150), first word pair of the second block
            printf("%.9g %.9g %.9g %.9g %.9g\n", a.x, a.y, a.z, a.w, e);
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
