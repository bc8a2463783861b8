// Issue rates of the VALU operations the path tracer leans on (per wave64 instruction, in cycles at
// the measured clock): 32-bit integer multiplies, the 64-bit multiply of the counter-based generator,
// f64 fma, f32 fma, 32-bit add. Build: hipcc --offload-arch=gfx950 -O3 valu_rates.hip -o valu_rates
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

constexpr int ITER = 4096, CHAINS = 8;
template <int OP>
__global__ void k(uint64_t* out, uint64_t seed) {
    uint64_t a[CHAINS];
    for (int c = 0; c < CHAINS; ++c) a[c] = seed + threadIdx.x * 977 + c * 7919;
    for (int i = 0; i < ITER; ++i) {
#pragma unroll
        for (int c = 0; c < CHAINS; ++c) {
            if (OP == 0) a[c] = (uint32_t)a[c] * 0x9E3779B9u + 1u;                                  // v_mul_lo_u32 (+ add)
            if (OP == 1) a[c] = a[c] * 0xBF58476D1CE4E5B9ULL + 1u;                                   // 64 x 64 -> 64
            if (OP == 2) a[c] = (uint32_t)a[c] + 0x9E3779B9u + (uint32_t)i;                            // adds
            if (OP == 3) a[c] = __double_as_longlong(fma(__longlong_as_double(a[c]), 1.0000001, 0.5)); // v_fma_f64
            if (OP == 4) a[c] = __float_as_uint(fmaf(__uint_as_float((uint32_t)a[c]), 1.0001f, 0.5f)); // v_fma_f32
            if (OP == 5) a[c] = __umulhi((uint32_t)a[c], 0x9E3779B9u) + (uint32_t)i;                   // v_mul_hi_u32
            if (OP == 6) { uint64_t z = a[c]; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL; z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL; a[c] = z ^ (z >> 31); }  // mix64
        }
    }
    uint64_t s = 0;
    for (int c = 0; c < CHAINS; ++c) s ^= a[c];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int OP>
double run(uint64_t* out, const char* name, double ops_per_iter) {
    const int blocks = 256 * 8, threads = 256;
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(threads), 0, 0, out, 1ull);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(threads), 0, 0, out, 2ull);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    // wave-instructions per SIMD: blocks * 4 waves * ITER * CHAINS * ops / (256 CUs * 4 SIMDs)
    double wave_ops = (double)blocks * 4 * ITER * CHAINS * ops_per_iter / (256.0 * 4.0);
    double ns_per_op = ms * 1e6 / wave_ops;
    std::printf("%-28s %8.3f ms  %6.2f ns per wave-op per SIMD (x clock GHz = cycles)\n", name, ms, ns_per_op);
    return ns_per_op;
}
int main() {
    uint64_t* out;
    hipMalloc(&out, sizeof(uint64_t) * 256 * 8 * 256);
    run<2>(out, "u32 add (2 per step)", 2);
    run<4>(out, "f32 fma", 1);
    run<3>(out, "f64 fma", 1);
    run<0>(out, "u32 mul_lo (+1 add)", 1);
    run<5>(out, "u32 mul_hi (+1 add)", 1);
    run<1>(out, "u64 mul (+add)", 1);
    run<6>(out, "mix64 (2 u64 mul, 3 xorshift)", 1);
    hipFree(out);
    return 0;
}
