// Issue cost of single gfx950 vector instructions, by name: each kernel repeats one instruction (inline
// assembly, eight independent destination registers, so neither the compiler nor dependences get in the way)
// and the table gives cycles per wave64 instruction per SIMD with 4 and with 8 waves on every SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 valu_issue.hip -o valu_issue
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

constexpr int ITER = 2048, UNROLL = 64;

#define REP8(S0, S1, S2, S3, S4, S5, S6, S7) S0 "\n" S1 "\n" S2 "\n" S3 "\n" S4 "\n" S5 "\n" S6 "\n" S7 "\n"
// one instruction with eight different destinations, D = destination register pair / register
#define BODY64(INS, SRC) REP8(INS " v[20:21], " SRC, INS " v[22:23], " SRC, INS " v[24:25], " SRC, INS " v[26:27], " SRC, \
                              INS " v[28:29], " SRC, INS " v[30:31], " SRC, INS " v[32:33], " SRC, INS " v[34:35], " SRC)
#define BODY32(INS, SRC) REP8(INS " v20, " SRC, INS " v21, " SRC, INS " v22, " SRC, INS " v23, " SRC, \
                              INS " v24, " SRC, INS " v25, " SRC, INS " v26, " SRC, INS " v27, " SRC)
#define BODYCMP(INS, SRC) REP8(INS " s[20:21], " SRC, INS " s[22:23], " SRC, INS " s[24:25], " SRC, INS " s[26:27], " SRC, \
                               INS " s[28:29], " SRC, INS " s[30:31], " SRC, INS " s[32:33], " SRC, INS " s[34:35], " SRC)
#define X8(B) B B B B B B B B

#define KERNEL(NAME, BODY)                                                                                         \
    __global__ void NAME(double* out, double seed) {                                                               \
        double a = seed + threadIdx.x, b = seed * 0.5 + 1.0;                                                       \
        for (int i = 0; i < ITER; ++i) {                                                                           \
            asm volatile("v_mov_b64 v[10:11], %0\nv_mov_b64 v[12:13], %1\nv_mov_b64 v[14:15], %0\n" X8(BODY)       \
                         :                                                                                         \
                         : "v"(a), "v"(b)                                                                          \
                         : "v10", "v11", "v12", "v13", "v14", "v15", "v20", "v21", "v22", "v23", "v24", "v25",     \
                           "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "s20", "s21",     \
                           "s22", "s23", "s24", "s25", "s26", "s27", "s28", "s29", "s30", "s31", "s32", "s33",     \
                           "s34", "s35", "vcc");                                                                   \
        }                                                                                                          \
        if (a == 12345.678) out[0] = a;                                                                            \
    }

KERNEL(k_fma64, BODY64("v_fma_f64", "v[10:11], v[12:13], v[14:15]"))
KERNEL(k_add64, BODY64("v_add_f64", "v[10:11], v[12:13]"))
KERNEL(k_mul64, BODY64("v_mul_f64", "v[10:11], v[12:13]"))
KERNEL(k_min64, BODY64("v_min_f64", "v[10:11], v[12:13]"))
KERNEL(k_max64, BODY64("v_max_f64", "v[10:11], v[12:13]"))
KERNEL(k_mov64, BODY64("v_mov_b64", "v[10:11]"))
KERNEL(k_rcp64, BODY64("v_rcp_f64", "v[10:11]"))
KERNEL(k_rsq64, BODY64("v_rsq_f64", "v[10:11]"))
KERNEL(k_ldexp64, BODY64("v_ldexp_f64", "v[10:11], v12"))
KERNEL(k_cvt_f64_u32, BODY64("v_cvt_f64_u32", "v10"))
KERNEL(k_lshr64, BODY64("v_lshrrev_b64", "7, v[10:11]"))
KERNEL(k_mad_u64_u32, BODY64("v_mad_u64_u32", "vcc, v10, v12, v[14:15]"))
KERNEL(k_fma32, BODY32("v_fma_f32", "v10, v12, v14"))
KERNEL(k_add32, BODY32("v_add_u32", "v10, v12"))
KERNEL(k_mov32, BODY32("v_mov_b32", "v10"))
KERNEL(k_xor32, BODY32("v_xor_b32", "v10, v12"))
KERNEL(k_cndmask_e64, BODY32("v_cndmask_b32", "v10, v12, s[36:37]"))
KERNEL(k_mul_lo, BODY32("v_mul_lo_u32", "v10, v12"))
KERNEL(k_mul_hi, BODY32("v_mul_hi_u32", "v10, v12"))
KERNEL(k_add3, BODY32("v_add3_u32", "v10, v12, v14"))
KERNEL(k_cmp_lt64, BODYCMP("v_cmp_lt_f64", "v[10:11], v[12:13]"))
KERNEL(k_cmp_class64, BODYCMP("v_cmp_class_f64", "v[10:11], v12"))
KERNEL(k_cmp_lt32, BODYCMP("v_cmp_lt_i32", "v10, v12"))
KERNEL(k_readlane, REP8("v_readlane_b32 s20, v10, 3", "v_readlane_b32 s21, v11, 3", "v_readlane_b32 s22, v12, 3", "v_readlane_b32 s23, v13, 3",
                        "v_readlane_b32 s24, v10, 5", "v_readlane_b32 s25, v11, 5", "v_readlane_b32 s26, v12, 5", "v_readlane_b32 s27, v13, 5"))
// dependent chains: every instruction reads the result of the one before it (one chain, or two interleaved)
#define CHAIN1(INS, D, REST) REP8(INS " " D ", " D REST, INS " " D ", " D REST, INS " " D ", " D REST, INS " " D ", " D REST, \
                                  INS " " D ", " D REST, INS " " D ", " D REST, INS " " D ", " D REST, INS " " D ", " D REST)
#define CHAIN2(INS, D, E, REST) REP8(INS " " D ", " D REST, INS " " E ", " E REST, INS " " D ", " D REST, INS " " E ", " E REST, \
                                     INS " " D ", " D REST, INS " " E ", " E REST, INS " " D ", " D REST, INS " " E ", " E REST)
KERNEL(k_dep_fma64, CHAIN1("v_fma_f64", "v[20:21]", ", v[12:13], v[14:15]"))
KERNEL(k_dep2_fma64, CHAIN2("v_fma_f64", "v[20:21]", "v[22:23]", ", v[12:13], v[14:15]"))
KERNEL(k_dep_add64, CHAIN1("v_add_f64", "v[20:21]", ", v[12:13]"))
KERNEL(k_dep_max64, CHAIN1("v_max_f64", "v[20:21]", ", v[20:21]"))
KERNEL(k_dep_fma32, CHAIN1("v_fma_f32", "v20", ", v12, v14"))
KERNEL(k_dep_add32, CHAIN1("v_add_u32", "v20", ", v12"))
KERNEL(k_dep_rcp64, CHAIN1("v_rcp_f64", "v[20:21]", ""))

template <class K>
void run_waves(K kern, const char* name, double* out, double clock_ghz) {
    std::printf("%-28s", name);
    for (int waves_per_simd = 1; waves_per_simd <= 4; waves_per_simd *= 2) {
        const int blocks = 256 * 4 * waves_per_simd;
        hipEvent_t e0, e1;
        hipEventCreate(&e0), hipEventCreate(&e1);
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(64), 0, 0, out, 1.5);
        hipEventRecord(e0);
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(64), 0, 0, out, 2.5);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        // cycles a WAVE spends per instruction of its chain
        std::printf("  %d wave%s/SIMD: %6.2f cycles per instruction of a wave", waves_per_simd, waves_per_simd > 1 ? "s" : " ", ms * 1e6 * clock_ghz / ((double)ITER * 64));
        hipEventDestroy(e0), hipEventDestroy(e1);
    }
    std::printf("\n");
}

template <class K>
void run(K kern, const char* name, double* out, double ins_per_body, double clock_ghz) {
    double cyc[2];
    for (int v = 0; v < 2; ++v) {
        const int waves_per_simd = v == 0 ? 4 : 8;
        const int blocks = 256 * 4 * waves_per_simd;  // single-wave workgroups
        hipEvent_t e0, e1;
        hipEventCreate(&e0), hipEventCreate(&e1);
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(64), 0, 0, out, 1.5);
        hipEventRecord(e0);
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(64), 0, 0, out, 2.5);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        const double ins_per_simd = (double)waves_per_simd * ITER * 8 * ins_per_body;
        cyc[v] = ms * 1e6 * clock_ghz / ins_per_simd;
        hipEventDestroy(e0), hipEventDestroy(e1);
    }
    std::printf("%-28s %6.2f cycles per wave instruction per SIMD at 4 waves, %6.2f at 8\n", name, cyc[0], cyc[1]);
}

int main() {
    double* out;
    hipMalloc(&out, 64);
    int khz = 0;
    hipDeviceGetAttribute(&khz, hipDeviceAttributeClockRate, 0);
    const double ghz = khz * 1e-6;
    std::printf("clock %.3f GHz (attribute; cycles below assume it)\n", ghz);
#define RUN(K, N) run(K, #K, out, N, ghz)
    RUN(k_fma64, 8), RUN(k_add64, 8), RUN(k_mul64, 8), RUN(k_min64, 8), RUN(k_max64, 8), RUN(k_mov64, 8), RUN(k_rcp64, 8), RUN(k_rsq64, 8);
    RUN(k_ldexp64, 8), RUN(k_cvt_f64_u32, 8), RUN(k_lshr64, 8), RUN(k_mad_u64_u32, 8);
    RUN(k_fma32, 8), RUN(k_add32, 8), RUN(k_mov32, 8), RUN(k_xor32, 8), RUN(k_cndmask_e64, 8), RUN(k_mul_lo, 8), RUN(k_mul_hi, 8), RUN(k_add3, 8);
    RUN(k_cmp_lt64, 8), RUN(k_cmp_class64, 8), RUN(k_cmp_lt32, 8), RUN(k_readlane, 8);
#define RUNW(K) run_waves(K, #K, out, ghz)
    RUNW(k_dep_fma64), RUNW(k_dep2_fma64), RUNW(k_dep_add64), RUNW(k_dep_max64), RUNW(k_dep_fma32), RUNW(k_dep_add32), RUNW(k_dep_rcp64);
    RUNW(k_fma64), RUNW(k_cndmask_e64);
    hipFree(out);
    return 0;
}
