// instr_probe.hip — developer tool: issue cost of single VALU instructions on gfx950, in s_memtime ticks per
// wave-instruction, with 4 waves per SIMD (enough to saturate the issue port) and 16 independent accumulators per wave.
// Used to price the Monte-Carlo kernel's instruction mix (Philox multiplies vs fp32 ops).  Usage: instr_probe
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); return 1; } } while (0)
struct Stamp { unsigned long long c0, c1, r0, r1; };
constexpr int ITER = 4000;

#define PROBE(NAME, ASM, CONSTRAINT_TYPE, INIT)                                                         \
    __global__ __launch_bounds__(64) void NAME(unsigned* sink, Stamp* st)                               \
    {                                                                                                   \
        CONSTRAINT_TYPE a[16];                                                                          \
        unsigned m = 0x9E3779B9u + (threadIdx.x >> 7);                                                  \
        _Pragma("unroll") for (int i = 0; i < 16; i++) a[i] = (CONSTRAINT_TYPE)(INIT + threadIdx.x + i); \
        Stamp s;                                                                                        \
        s.c0 = __builtin_amdgcn_s_memtime();                                                            \
        s.r0 = __builtin_amdgcn_s_memrealtime();                                                        \
        for (int it = 0; it < ITER; it++) {                                                             \
            _Pragma("unroll") for (int i = 0; i < 16; i++) asm volatile(ASM : "+v"(a[i]) : "v"(m) : "vcc", "s20", "s21"); \
        }                                                                                               \
        s.c1 = __builtin_amdgcn_s_memtime();                                                            \
        s.r1 = __builtin_amdgcn_s_memrealtime();                                                        \
        unsigned t = 0;                                                                                 \
        _Pragma("unroll") for (int i = 0; i < 16; i++) t += (unsigned)a[i];                             \
        if (t == 0x12345678u) sink[0] = t;                                                              \
        if (threadIdx.x == 0) st[blockIdx.x] = s;                                                       \
    }

PROBE(k_fma, "v_fma_f32 %0, %0, 1.0, %1", float, 1.0f)
PROBE(k_mul, "v_mul_f32 %0, 1.0, %0", float, 1.0f)
PROBE(k_min3, "v_min3_f32 %0, %0, 1.0, %1", float, 1.0f)
PROBE(k_mul_lo, "v_mul_lo_u32 %0, %0, %1", unsigned, 3u)
PROBE(k_mul_hi, "v_mul_hi_u32 %0, %0, %1", unsigned, 3u)
PROBE(k_mul_u24, "v_mul_u32_u24 %0, %0, %1", unsigned, 3u)
PROBE(k_mad64, "v_mad_u64_u32 %0, vcc, %1, %1, 0", unsigned long long, 3ull)
PROBE(k_bitop3, "v_bitop3_b32 %0, %0, %1, %1 bitop3:0x96", unsigned, 3u)
PROBE(k_rcp, "v_rcp_f32 %0, %0", float, 1.0f)
PROBE(k_sqrt, "v_sqrt_f32 %0, %0", float, 1.0f)
PROBE(k_cvt, "v_cvt_i32_f32 %0, %0", float, 1.0f)
PROBE(k_rndne, "v_rndne_f32 %0, %0", float, 1.0f)
PROBE(k_pkmul, "v_pk_mul_f32 %0, %0, %0", double, 1.0)
// round 5: the rest of the Monte-Carlo kernels' mix (static histogram of c2d_mc.hip + c2d_mc_poly.hip), to price an
// issue-weighted VALU roof (profiles/counts.py, bench.py `frac_issue_weighted`)
PROBE(k_add, "v_add_f32 %0, 1.0, %0", float, 1.0f)
PROBE(k_sub, "v_sub_f32 %0, %0, %1", float, 1.0f)
PROBE(k_max, "v_max_f32 %0, %0, %1", float, 1.0f)
PROBE(k_min, "v_min_f32 %0, %0, %1", float, 1.0f)
PROBE(k_max3, "v_max3_f32 %0, %0, 1.0, %1", float, 1.0f)
PROBE(k_med3, "v_med3_f32 %0, %0, 1.0, %1", float, 1.0f)
PROBE(k_cmp32, "v_cmp_lt_f32 vcc, %0, %1", float, 1.0f)
PROBE(k_cmp64, "v_cmp_lt_f32 s[20:21], %0, %1", float, 1.0f)
PROBE(k_cmpu, "v_cmp_ne_u32 vcc, %0, %1", unsigned, 3u)
PROBE(k_cndmask, "v_cndmask_b32 %0, %0, %1, vcc", unsigned, 3u)
PROBE(k_mov, "v_mov_b32 %0, %1", unsigned, 3u)
PROBE(k_addu, "v_add_u32 %0, %0, %1", unsigned, 3u)
PROBE(k_and, "v_and_b32 %0, %0, %1", unsigned, 3u)
PROBE(k_xor, "v_xor_b32 %0, %0, %1", unsigned, 3u)
PROBE(k_lshl, "v_lshlrev_b32 %0, 1, %0", unsigned, 3u)
PROBE(k_lshladd, "v_lshl_add_u32 %0, %0, 1, %1", unsigned, 3u)
PROBE(k_bfi, "v_bfi_b32 %0, %1, %0, %1", unsigned, 3u)
PROBE(k_alignbit, "v_alignbit_b32 %0, %0, %1, 7", unsigned, 3u)
PROBE(k_fmaak, "v_fmaak_f32 %0, %0, %1, 0x3f8ccccd", float, 1.0f)
PROBE(k_fmac, "v_fmac_f32 %0, %1, %1", float, 1.0f)
PROBE(k_cvtfu, "v_cvt_f32_u32 %0, %0", unsigned, 3u)
PROBE(k_exp, "v_exp_f32 %0, %0", float, 1.0f)
PROBE(k_log, "v_log_f32 %0, %0", float, 1.0f)
PROBE(k_sin, "v_sin_f32 %0, %0", float, 1.0f)
PROBE(k_mbcnt, "v_mbcnt_lo_u32_b32 %0, %1, %0", unsigned, 3u)
PROBE(k_bcnt, "v_bcnt_u32_b32 %0, %0, %1", unsigned, 3u)
PROBE(k_lshladd64, "v_lshl_add_u64 %0, %0, 1, %0", unsigned long long, 3ull)

int main()
{
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    unsigned* sink; Stamp* st;
    CK(hipMalloc(&sink, 4));
    const int w = 4, grid = prop.multiProcessorCount * 4 * w;
    CK(hipMalloc(&st, sizeof(Stamp) * grid));
    struct K { const char* name; void (*fn)(unsigned*, Stamp*); } ks[] = {
        {"v_fma_f32", k_fma}, {"v_mul_f32", k_mul}, {"v_min3_f32", k_min3}, {"v_mul_lo_u32", k_mul_lo}, {"v_mul_hi_u32", k_mul_hi},
        {"v_mul_u32_u24", k_mul_u24}, {"v_mad_u64_u32", k_mad64}, {"v_bitop3_b32", k_bitop3}, {"v_rcp_f32", k_rcp}, {"v_sqrt_f32", k_sqrt},
        {"v_cvt_i32_f32", k_cvt}, {"v_rndne_f32", k_rndne}, {"v_pk_mul_f32", k_pkmul},
        {"v_add_f32", k_add}, {"v_sub_f32", k_sub}, {"v_max_f32", k_max}, {"v_min_f32", k_min}, {"v_max3_f32", k_max3}, {"v_med3_f32", k_med3},
        {"v_cmp_lt_f32 (vcc)", k_cmp32}, {"v_cmp_lt_f32 (sgpr)", k_cmp64}, {"v_cmp_ne_u32", k_cmpu}, {"v_cndmask_b32", k_cndmask}, {"v_mov_b32", k_mov},
        {"v_add_u32", k_addu}, {"v_and_b32", k_and}, {"v_xor_b32", k_xor}, {"v_lshlrev_b32", k_lshl}, {"v_lshl_add_u32", k_lshladd}, {"v_bfi_b32", k_bfi},
        {"v_alignbit_b32", k_alignbit}, {"v_fmaak_f32", k_fmaak}, {"v_fmac_f32", k_fmac}, {"v_cvt_f32_u32", k_cvtfu}, {"v_exp_f32", k_exp},
        {"v_log_f32", k_log}, {"v_sin_f32", k_sin}, {"v_mbcnt_lo_u32_b32", k_mbcnt}, {"v_bcnt_u32_b32", k_bcnt}, {"v_lshl_add_u64", k_lshladd64}};
    printf("%d waves per SIMD, 16 independent accumulators; ticks per own instruction / %d = SIMD issue cost in ticks\n", w, w);
    for (auto& k : ks) {
        for (int rep = 0; rep < 2; rep++) hipLaunchKernelGGL(k.fn, dim3(grid), dim3(64), 0, 0, sink, st);
        CK(hipDeviceSynchronize());
        std::vector<Stamp> h(grid);
        CK(hipMemcpy(h.data(), st, sizeof(Stamp) * grid, hipMemcpyDeviceToHost));
        std::vector<double> tpi, ghz;
        for (auto& s : h) { tpi.push_back((double)(s.c1 - s.c0) / (ITER * 16.0)); ghz.push_back((double)(s.c1 - s.c0) / ((s.r1 - s.r0) * 10.0)); }
        std::sort(tpi.begin(), tpi.end()); std::sort(ghz.begin(), ghz.end());
        printf("%-20s ticks per own instr %.2f -> issue cost %.2f ticks;  s_memtime %.2f ticks/ns\n", k.name, tpi[grid / 2], tpi[grid / 2] / w, ghz[grid / 2]);
    }
    return 0;
}
