/* Runs the CPU oracle's entry points on small inputs under AddressSanitizer + UndefinedBehaviourSanitizer
 * (tests/test_sanitizers.py compiles this file together with oracle/c2d_oracle.c).  The GPU pool has no device
 * sanitizer, so memory-safety checking happens on the CPU builds.  TEST INFRASTRUCTURE. */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "../../include/utils.h"

unsigned long long c2d_oracle_sat_rect_pairs_verts(const float* const planes[16], size_t n, uint8_t* out);
unsigned long long c2d_oracle_sat_rect_pairs_pose(const float* const pp[10], size_t n, uint8_t* out);
unsigned long long c2d_oracle_sat_poly_pairs(const float* vx, const float* vy, const uint8_t* k, size_t n, uint8_t* out);
int c2d_oracle_rects_from_poses(const float* cx, const float* cy, const float* w, const float* h, const float* theta, size_t n,
                                float* const out_planes[8]);
unsigned long long c2d_oracle_mc_pair(float robot_w, float robot_h, const Position* pos, const Pose* pose, const StdDev* sd, uint64_t seed,
                                      uint64_t scene_id, uint64_t sample_begin, uint64_t n_samples);
unsigned long long c2d_oracle_mc_scenes(const Pose* poses, uint32_t num_poses, const StdDev* std_devs, uint32_t num_std_devs,
                                        const PositionWithVarAndPoseIdx* scenes, size_t n_scenes, float robot_w, float robot_h,
                                        const float* bins, const float* acc, uint32_t n_bins, uint32_t max_samples, uint64_t seed,
                                        uint64_t scene_id_base, uint32_t small_batch, uint32_t large_batch, uint32_t switch_at,
                                        uint32_t* hits, uint32_t* n_used, PoseCPVarAndPoseIdx* rows);
void c2d_oracle_sample_scenes(const Pose* poses, uint32_t num_poses, const StdDev* std_devs, uint32_t num_std_devs, float robot_w,
                              float robot_h, float spread, uint64_t seed, uint64_t scene_id_base, size_t n_scenes,
                              PositionWithVarAndPoseIdx* scenes);

static float frand(void) { return (float)rand() / (float)RAND_MAX; }

int main(void)
{
    enum { N = 777, KM = C2D_POLY_KMAX };
    float* pose = malloc(sizeof(float) * 10 * N);
    float* planes = malloc(sizeof(float) * 16 * N);
    uint8_t* out = malloc(N);
    const float* pp[10];
    float* op[8];
    const float* cp[16];
    for (int i = 0; i < 10 * N; i++) pose[i] = frand() * 6.0f - 1.0f;
    for (int k = 0; k < 10; k++) pp[k] = pose + k * N;
    for (int r = 0; r < 2; r++) {
        for (int k = 0; k < 8; k++) op[k] = planes + (8 * r + k) * N;
        c2d_oracle_rects_from_poses(pp[5 * r], pp[5 * r + 1], pp[5 * r + 2], pp[5 * r + 3], pp[5 * r + 4], N, op);
    }
    for (int k = 0; k < 16; k++) cp[k] = planes + k * N;
    unsigned long long a = c2d_oracle_sat_rect_pairs_verts(cp, N, out);
    unsigned long long b = c2d_oracle_sat_rect_pairs_pose(pp, N, out);
    if (a != b) return 1;
    /* polygons: every slot at or above the count is left UNINITIALISED on purpose: the oracle must not read it */
    float* vx = malloc(sizeof(float) * 2 * KM * N);
    float* vy = malloc(sizeof(float) * 2 * KM * N);
    uint8_t* kk = malloc(2 * N);
    for (int p = 0; p < 2; p++)
        for (int i = 0; i < N; i++) {
            const int k = 1 + rand() % KM;
            kk[p * N + i] = (uint8_t)k;
            for (int v = 0; v < k; v++) {
                vx[((size_t)p * KM + v) * N + i] = frand() * 4.0f;
                vy[((size_t)p * KM + v) * N + i] = frand() * 4.0f;
            }
        }
    (void)c2d_oracle_sat_poly_pairs(vx, vy, kk, N, out);
    /* Monte-Carlo */
    const Position pos = {3.0f, 1.0f};
    const Pose po = {2.0f, 1.0f, 0.6f};
    const StdDev sd = {0.3f, 0.3f, 0.2f, 0.1f, 0.05f};
    unsigned long long h = c2d_oracle_mc_pair(4.07f, 1.74f, &pos, &po, &sd, 1, 2, (1ull << 32) - 50, 100);
    enum { NS = 37, NT = 5 };
    Pose poses[NT];
    StdDev sds[NT];
    for (int i = 0; i < NT; i++) {
        poses[i] = (Pose){0.5f + frand() * 3, 0.5f + frand() * 3, frand() * 6};
        sds[i] = (StdDev){frand() * 0.5f, frand() * 0.5f, frand() * 0.5f, 0, 0};
    }
    PositionWithVarAndPoseIdx scenes[NS];
    c2d_oracle_sample_scenes(poses, NT, sds, NT, 4.07f, 1.74f, 4.0f, 9, 100, NS, scenes);
    const float bins[4] = {0, 0.01f, 0.1f, 1}, acc[3] = {1e-4f, 1e-3f, 1e-2f};
    uint32_t hits[NS], used[NS];
    PoseCPVarAndPoseIdx rows[NS];
    unsigned long long tot = c2d_oracle_mc_scenes(poses, NT, sds, NT, scenes, NS, 4.07f, 1.74f, bins, acc, 4, 3000, 9, 100, 0, 0, 0, hits, used, rows);
    printf("sanitize ok %llu %llu %llu\n", a, h, tot);
    free(pose); free(planes); free(out); free(vx); free(vy); free(kk);
    return 0;
}
