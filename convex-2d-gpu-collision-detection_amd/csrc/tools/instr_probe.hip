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

// One probe = one kernel whose timed region is a single asm block: the loop, its 16 independent instructions on fixed
// registers v32..v47 (v[32:63] for the 64-bit ones), the constant operand in v50 / v[50:51], vcc = alternating lanes.  Nothing
// of it is left to the compiler — round 5 found that per-instruction asm statements with a vcc clobber make hipcc put an
// s_nop between every two of them, which changes what is measured.
#define REP16(F) F(32) F(33) F(34) F(35) F(36) F(37) F(38) F(39) F(40) F(41) F(42) F(43) F(44) F(45) F(46) F(47)
#define REP16_64(F) F(32, 33) F(34, 35) F(36, 37) F(38, 39) F(40, 41) F(42, 43) F(44, 45) F(46, 47) F(52, 53) F(54, 55) F(56, 57) F(58, 59) F(60, 61) F(62, 63) F(64, 65) F(66, 67)
#define CLOBBERS "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63", "v64", "v65", "v66", "v67", "vcc", "s20", "s21", "s22", "s23", "scc"
#define INIT_F(r) "v_cvt_f32_u32 v" #r ", %2\n v_add_f32 v" #r ", 1.5, v" #r "\n"
#define INIT_U(r) "v_add_u32 v" #r ", " #r ", %2\n"
#define INIT_U64(r, q) "v_add_u32 v" #r ", " #r ", %2\n v_mov_b32 v" #q ", 1\n"

#define PROBE_BODY(NAME, INIT, BODY)                                                                    \
    __global__ __launch_bounds__(64) void NAME(unsigned* sink, Stamp* st)                               \
    {                                                                                                   \
        Stamp s;                                                                                        \
        unsigned long long c0, c1, r0, r1;                                                              \
        asm volatile(INIT                                                                               \
                     "v_mov_b32 v50, 0x3f8ccccd\n v_mov_b32 v51, 0\n"                                   \
                     "s_mov_b32 vcc_lo, 0x55555555\n s_mov_b32 vcc_hi, 0x55555555\n"                    \
                     "s_movk_i32 s22, %3\n"                                                             \
                     "s_memtime %0\n s_memrealtime %1\n s_waitcnt lgkmcnt(0)\n"                         \
                     ".p2align 8\n"                                                                    \
                     ".Lprobe_loop%=:\n" BODY                                                           \
                     "s_sub_u32 s22, s22, 1\n s_cmp_lg_u32 s22, 0\n s_cbranch_scc1 .Lprobe_loop%=\n"     \
                     : "=s"(c0), "=s"(r0) : "v"(threadIdx.x), "n"(ITER) : CLOBBERS);                      \
        asm volatile("s_memtime %0\n s_memrealtime %1\n s_waitcnt lgkmcnt(0)\n" : "=s"(c1), "=s"(r1) : : "memory"); \
        s.c0 = c0; s.c1 = c1; s.r0 = r0; s.r1 = r1;                                                     \
        unsigned t;                                                                                     \
        asm volatile("v_xor_b32 %0, v32, v47" : "=v"(t) : : );                                          \
        if (t == 0x12345678u) sink[0] = t;                                                              \
        if (threadIdx.x == 0) st[blockIdx.x] = s;                                                       \
    }
#define PROBE_F(NAME, F) PROBE_BODY(NAME, REP16(INIT_F), REP16(F))
#define PROBE_U(NAME, F) PROBE_BODY(NAME, REP16(INIT_U), REP16(F))
#define PROBE_U64(NAME, F) PROBE_BODY(NAME, REP16_64(INIT_U64), REP16_64(F))

#define I_FMA(r) "v_fma_f32 v" #r ", v" #r ", 1.0, v50\n"
#define I_MUL(r) "v_mul_f32 v" #r ", 1.0, v" #r "\n"
#define I_ADD(r) "v_add_f32 v" #r ", 1.0, v" #r "\n"
#define I_SUB(r) "v_sub_f32 v" #r ", v" #r ", v50\n"
#define I_MAX(r) "v_max_f32 v" #r ", v" #r ", v50\n"
#define I_MIN(r) "v_min_f32 v" #r ", v" #r ", v50\n"
#define I_MIN3(r) "v_min3_f32 v" #r ", v" #r ", 1.0, v50\n"
#define I_MAX3(r) "v_max3_f32 v" #r ", v" #r ", 1.0, v50\n"
#define I_MED3(r) "v_med3_f32 v" #r ", v" #r ", 1.0, v50\n"
#define I_FMAAK(r) "v_fmaak_f32 v" #r ", v" #r ", v50, 0x3f8ccccd\n"
#define I_FMAC(r) "v_fmac_f32 v" #r ", v50, v50\n"
#define I_RCP(r) "v_rcp_f32 v" #r ", v" #r "\n"
#define I_SQRT(r) "v_sqrt_f32 v" #r ", v" #r "\n"
#define I_EXP(r) "v_exp_f32 v" #r ", v" #r "\n"
#define I_LOG(r) "v_log_f32 v" #r ", v" #r "\n"
#define I_SIN(r) "v_sin_f32 v" #r ", v" #r "\n"
#define I_CVTIF(r) "v_cvt_i32_f32 v" #r ", v" #r "\n"
#define I_CVTFU(r) "v_cvt_f32_u32 v" #r ", v" #r "\n"
#define I_RNDNE(r) "v_rndne_f32 v" #r ", v" #r "\n"
#define I_CMPVCC(r) "v_cmp_lt_f32 vcc, v" #r ", v50\n"
#define I_CMPSGPR(r) "v_cmp_lt_f32 s[20:21], v" #r ", v50\n"
#define I_CMPU(r) "v_cmp_ne_u32 vcc, v" #r ", v50\n"
#define I_CNDMASK(r) "v_cndmask_b32 v" #r ", v" #r ", v50, vcc\n"
#define I_MOV(r) "v_mov_b32 v" #r ", v50\n"
#define I_ADDU(r) "v_add_u32 v" #r ", v" #r ", v50\n"
#define I_AND(r) "v_and_b32 v" #r ", v" #r ", v50\n"
#define I_XOR(r) "v_xor_b32 v" #r ", v" #r ", v50\n"
#define I_LSHL(r) "v_lshlrev_b32 v" #r ", 1, v" #r "\n"
#define I_LSHLADD(r) "v_lshl_add_u32 v" #r ", v" #r ", 1, v50\n"
#define I_BFI(r) "v_bfi_b32 v" #r ", v50, v" #r ", v50\n"
#define I_ALIGNBIT(r) "v_alignbit_b32 v" #r ", v" #r ", v50, 7\n"
#define I_MULLO(r) "v_mul_lo_u32 v" #r ", v" #r ", v50\n"
#define I_MULHI(r) "v_mul_hi_u32 v" #r ", v" #r ", v50\n"
#define I_MULU24(r) "v_mul_u32_u24 v" #r ", v" #r ", v50\n"
#define I_BITOP3(r) "v_bitop3_b32 v" #r ", v" #r ", v50, v50 bitop3:0x96\n"
#define I_MBCNT(r) "v_mbcnt_lo_u32_b32 v" #r ", v50, v" #r "\n"
#define I_BCNT(r) "v_bcnt_u32_b32 v" #r ", v" #r ", v50\n"
#define I_MAD64(r, q) "v_mad_u64_u32 v[" #r ":" #q "], vcc, v50, v50, 0\n"
#define I_LSHLADD64(r, q) "v_lshl_add_u64 v[" #r ":" #q "], v[" #r ":" #q "], 1, v[" #r ":" #q "]\n"
#define I_PKMUL(r, q) "v_pk_mul_f32 v[" #r ":" #q "], v[" #r ":" #q "], v[" #r ":" #q "]\n"
#define I_PKFMA(r, q) "v_pk_fma_f32 v[" #r ":" #q "], v[" #r ":" #q "], v[50:51], v[50:51]\n"

// pairs of instruction types in one stream, alternating (eight of each per pass): additive prices mean both occupy the same
// issue port; a pair that runs faster than the mean of its members shows two paths that overlap across waves
#define ALT16(A, B) A(32) B(33) A(34) B(35) A(36) B(37) A(38) B(39) A(40) B(41) A(42) B(43) A(44) B(45) A(46) B(47)
#define PROBE_PAIR(NAME, INIT, A, B) PROBE_BODY(NAME, REP16(INIT), ALT16(A, B))
#define ALT16_64(A, B) A(32) B(52, 53) A(34) B(54, 55) A(36) B(56, 57) A(38) B(58, 59) A(40) B(60, 61) A(42) B(62, 63) A(44) B(64, 65) A(46) B(66, 67)
#define PROBE_PAIR64(NAME, INIT, A, B) PROBE_BODY(NAME, REP16(INIT), ALT16_64(A, B))
#define I_CNDMASK_S(r) "v_cndmask_b32 v" #r ", v" #r ", v50, s[20:21]\n"
PROBE_PAIR(k_mul_max3, INIT_F, I_MUL, I_MAX3)
PROBE_PAIR(k_mul_cmp, INIT_F, I_MUL, I_CMPVCC)
PROBE_PAIR(k_mul_bitop3, INIT_U, I_MUL, I_BITOP3)
PROBE_PAIR64(k_mul_mad64, INIT_U, I_MUL, I_MAD64)
PROBE_PAIR(k_mul_cndmask, INIT_U, I_MUL, I_CNDMASK)
PROBE_PAIR(k_max3_bitop3, INIT_U, I_MAX3, I_BITOP3)
PROBE_PAIR64(k_max3_mad64, INIT_U, I_MAX3, I_MAD64)
PROBE_PAIR(k_mul_sqrt, INIT_F, I_MUL, I_SQRT)
PROBE_PAIR(k_fma_mul, INIT_F, I_FMA, I_MUL)
PROBE_U(k_cndmask_s, I_CNDMASK_S)

// the Monte-Carlo kernels' own instruction mixes as dependency-free streams (generated by profiles/valu_issue.py mixes <tag> from
// the per-type PMC counts and the static mix of each kernel's loop code; shuffled with a fixed seed): what the VALU port needs for
// THAT mix when nothing else holds it up — the roof bench.py's frac_issue_weighted is taken against
#if __has_include("instr_probe_mixes.inc")
#include "instr_probe_mixes.inc"
#endif

PROBE_F(k_fma, I_FMA)
PROBE_F(k_mul, I_MUL)
PROBE_F(k_add, I_ADD)
PROBE_F(k_sub, I_SUB)
PROBE_F(k_max, I_MAX)
PROBE_F(k_min, I_MIN)
PROBE_F(k_min3, I_MIN3)
PROBE_F(k_max3, I_MAX3)
PROBE_F(k_med3, I_MED3)
PROBE_F(k_fmaak, I_FMAAK)
PROBE_F(k_fmac, I_FMAC)
PROBE_F(k_rcp, I_RCP)
PROBE_F(k_sqrt, I_SQRT)
PROBE_F(k_exp, I_EXP)
PROBE_F(k_log, I_LOG)
PROBE_F(k_sin, I_SIN)
PROBE_F(k_cvt, I_CVTIF)
PROBE_U(k_cvtfu, I_CVTFU)
PROBE_F(k_rndne, I_RNDNE)
PROBE_F(k_cmp32, I_CMPVCC)
PROBE_F(k_cmp64, I_CMPSGPR)
PROBE_U(k_cmpu, I_CMPU)
PROBE_U(k_cndmask, I_CNDMASK)
PROBE_U(k_mov, I_MOV)
PROBE_U(k_addu, I_ADDU)
PROBE_U(k_and, I_AND)
PROBE_U(k_xor, I_XOR)
PROBE_U(k_lshl, I_LSHL)
PROBE_U(k_lshladd, I_LSHLADD)
PROBE_U(k_bfi, I_BFI)
PROBE_U(k_alignbit, I_ALIGNBIT)
PROBE_U(k_mul_lo, I_MULLO)
PROBE_U(k_mul_hi, I_MULHI)
PROBE_U(k_mul_u24, I_MULU24)
PROBE_U(k_bitop3, I_BITOP3)
PROBE_U(k_mbcnt, I_MBCNT)
PROBE_U(k_bcnt, I_BCNT)
PROBE_U64(k_mad64, I_MAD64)
PROBE_U64(k_lshladd64, I_LSHLADD64)
PROBE_U64(k_pkmul, I_PKMUL)
PROBE_U64(k_pkfma, I_PKFMA)

int main()
{
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    unsigned* sink; Stamp* st;
    CK(hipMalloc(&sink, 4));
    const int w = 4, grid = prop.multiProcessorCount * 4 * w;
    CK(hipMalloc(&st, sizeof(Stamp) * grid));
    struct K { const char* name; void (*fn)(unsigned*, Stamp*); int per_pass = 16; } ks[] = {
        {"v_fma_f32", k_fma}, {"v_mul_f32", k_mul}, {"v_add_f32", k_add}, {"v_sub_f32", k_sub}, {"v_max_f32", k_max}, {"v_min_f32", k_min},
        {"v_min3_f32", k_min3}, {"v_max3_f32", k_max3}, {"v_med3_f32", k_med3}, {"v_fmaak_f32", k_fmaak}, {"v_fmac_f32", k_fmac},
        {"v_rcp_f32", k_rcp}, {"v_sqrt_f32", k_sqrt}, {"v_exp_f32", k_exp}, {"v_log_f32", k_log}, {"v_sin_f32", k_sin},
        {"v_cvt_i32_f32", k_cvt}, {"v_cvt_f32_u32", k_cvtfu}, {"v_rndne_f32", k_rndne},
        {"v_cmp_lt_f32 (vcc)", k_cmp32}, {"v_cmp_lt_f32 (sgpr)", k_cmp64}, {"v_cmp_ne_u32", k_cmpu}, {"v_cndmask_b32", k_cndmask}, {"v_mov_b32", k_mov},
        {"v_add_u32", k_addu}, {"v_and_b32", k_and}, {"v_xor_b32", k_xor}, {"v_lshlrev_b32", k_lshl}, {"v_lshl_add_u32", k_lshladd}, {"v_bfi_b32", k_bfi},
        {"v_alignbit_b32", k_alignbit}, {"v_mul_lo_u32", k_mul_lo}, {"v_mul_hi_u32", k_mul_hi}, {"v_mul_u32_u24", k_mul_u24}, {"v_bitop3_b32", k_bitop3},
        {"v_mbcnt_lo_u32_b32", k_mbcnt}, {"v_bcnt_u32_b32", k_bcnt}, {"v_mad_u64_u32", k_mad64}, {"v_lshl_add_u64", k_lshladd64},
        {"v_pk_mul_f32", k_pkmul}, {"v_pk_fma_f32", k_pkfma}, {"v_cndmask_b32 (sgpr)", k_cndmask_s},
        {"pair mul+max3", k_mul_max3}, {"pair mul+cmp", k_mul_cmp}, {"pair mul+bitop3", k_mul_bitop3}, {"pair mul+mad64", k_mul_mad64},
        {"pair mul+cndmask", k_mul_cndmask}, {"pair max3+bitop3", k_max3_bitop3}, {"pair max3+mad64", k_max3_mad64}, {"pair mul+sqrt", k_mul_sqrt},
        {"pair fma+mul", k_fma_mul}
#ifdef MIX_KERNELS
        , MIX_KERNELS
#endif
    };
    printf("%d waves per SIMD, 16 independent accumulators; ticks per own instruction / %d = SIMD issue cost in ticks\n", w, w);
    for (auto& k : ks) {
        for (int rep = 0; rep < 2; rep++) hipLaunchKernelGGL(k.fn, dim3(grid), dim3(64), 0, 0, sink, st);
        CK(hipDeviceSynchronize());
        std::vector<Stamp> h(grid);
        CK(hipMemcpy(h.data(), st, sizeof(Stamp) * grid, hipMemcpyDeviceToHost));
        std::vector<double> tpi, ghz;
        for (auto& s : h) { tpi.push_back((double)(s.c1 - s.c0) / ((double)ITER * k.per_pass)); ghz.push_back((double)(s.c1 - s.c0) / ((s.r1 - s.r0) * 10.0)); }
        std::sort(tpi.begin(), tpi.end()); std::sort(ghz.begin(), ghz.end());
        printf("%-34s ticks per own instr %.2f -> issue cost %.2f ticks;  s_memtime %.2f ticks/ns\n", k.name, tpi[grid / 2], tpi[grid / 2] / w, ghz[grid / 2]);
    }
    return 0;
}
