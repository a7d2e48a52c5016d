// microbench3.hip — issue cost of the 64-bit multiply-add forms the Poseidon2 kernel is built from, at the kernel's own
// occupancy (3 waves per SIMD) and at 8: v_mad_u64_u32 vs v_mad_i64_i32, carry-out to vcc vs a scratch SGPR pair, with and
// without the s_nop the assembler inserts between inline-asm multiply-adds, and the dependent triple (product, low
// multiply, correction) of one Montgomery reduction at ILP 1 / 4.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

#include "../boundless_amd/csrc/poseidon2_arith.hpp"

#define REP16(x) x x x x x x x x x x x x x x x x
#define KERNEL(name, INS, NINS)                                                                      \
    __global__ __launch_bounds__(256) void name(uint32_t* out, int iters, uint32_t seed) {          \
        uint32_t a0 = threadIdx.x + seed, a1 = a0 * 3u, a2 = a0 * 5u, a3 = a0 * 7u, b = seed | 1u, c = seed + 12345u; \
        uint64_t w0 = a0, w1 = a1, w2 = a2, w3 = a3;                                                 \
        for (int it = 0; it < iters; ++it) {                                                         \
            asm volatile(REP16(INS) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(w0), "+v"(w1), "+v"(w2), "+v"(w3) : "v"(b), "v"(c) : "vcc", "s40", "s41"); \
        }                                                                                            \
        if ((a0 ^ a1 ^ a2 ^ a3 ^ (uint32_t)w0 ^ (uint32_t)w1 ^ (uint32_t)w2 ^ (uint32_t)w3) == 0x12345u) out[0] = a0; \
    }                                                                                                \
    static const int name##_n = NINS;

// operands: %0-%3 a0-a3 (32-bit), %4-%7 w0-w3 (64-bit), %8 b, %9 c
KERNEL(k_umad_vcc, "v_mad_u64_u32 %4, vcc, %0, %8, %4\n v_mad_u64_u32 %5, vcc, %1, %8, %5\n v_mad_u64_u32 %6, vcc, %2, %8, %6\n v_mad_u64_u32 %7, vcc, %3, %8, %7\n", 4)
KERNEL(k_smad_vcc, "v_mad_i64_i32 %4, vcc, %0, %8, %4\n v_mad_i64_i32 %5, vcc, %1, %8, %5\n v_mad_i64_i32 %6, vcc, %2, %8, %6\n v_mad_i64_i32 %7, vcc, %3, %8, %7\n", 4)
KERNEL(k_smad_sgpr, "v_mad_i64_i32 %4, s[40:41], %0, %8, %4\n v_mad_i64_i32 %5, s[40:41], %1, %8, %5\n v_mad_i64_i32 %6, s[40:41], %2, %8, %6\n v_mad_i64_i32 %7, s[40:41], %3, %8, %7\n", 4)
KERNEL(k_smad_nop, "v_mad_i64_i32 %4, s[40:41], %0, %8, %4\n s_nop 0\n v_mad_i64_i32 %5, s[40:41], %1, %8, %5\n s_nop 0\n v_mad_i64_i32 %6, s[40:41], %2, %8, %6\n s_nop 0\n v_mad_i64_i32 %7, s[40:41], %3, %8, %7\n s_nop 0\n", 4)
KERNEL(k_mullo, "v_mul_lo_u32 %0, %0, %8\n v_mul_lo_u32 %1, %1, %8\n v_mul_lo_u32 %2, %2, %8\n v_mul_lo_u32 %3, %3, %8\n", 4)
KERNEL(k_add, "v_add_u32 %0, %0, %8\n v_add_u32 %1, %1, %8\n v_add_u32 %2, %2, %8\n v_add_u32 %3, %3, %8\n", 4)

// The kernel's own building block: ILP independent chains x <- sredc(x*x), issued stage-wise with the pinned primitives
// (signed) or left to the compiler (unsigned lazy form).  3 instructions per element and iteration (+2 with the reduce).
template <int ILP>
__global__ __launch_bounds__(256) void k_chain_signed(uint32_t* out, int iters, uint32_t seed) {
    bx::i32 x[ILP];
    for (int i = 0; i < ILP; ++i) x[i] = (bx::i32)((threadIdx.x * 2654435761u + seed + i * 977u) % bx::P) - (bx::i32)(bx::P / 2);
    for (int it = 0; it < iters * 4; ++it) {
        bx::i64 t[ILP];
#pragma unroll
        for (int i = 0; i < ILP; ++i) t[i] = bx::smul(x[i], x[i]);
        bx::sredc_n<ILP>(t, x);
    }
    uint32_t acc = 0;
    for (int i = 0; i < ILP; ++i) acc ^= (uint32_t)x[i];
    if (acc == 0x12345u) out[0] = acc;
}
template <int ILP>
__global__ __launch_bounds__(256) void k_chain_unsigned(uint32_t* out, int iters, uint32_t seed) {
    uint32_t x[ILP];
    for (int i = 0; i < ILP; ++i) x[i] = (threadIdx.x * 2654435761u + seed + i * 977u) % bx::P;
    for (int it = 0; it < iters * 4; ++it) {
#pragma unroll
        for (int i = 0; i < ILP; ++i) x[i] = bx::fp_reduce(bx::fp_mul_lazy(x[i], x[i]));
    }
    uint32_t acc = 0;
    for (int i = 0; i < ILP; ++i) acc ^= x[i];
    if (acc == 0x12345u) out[0] = acc;
}
template <class K>
static void run(const char* name, K kernel, double n_per_rep, int waves_per_simd, int cus) {
    uint32_t* d;
    hipMalloc(&d, 64);
    const int iters = 256;
    const int blocks = cus * waves_per_simd;  // 4 waves per block, 4 SIMDs per CU
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(kernel, dim3(blocks), dim3(256), 0, 0, d, iters, 7u);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(kernel, dim3(blocks), dim3(256), 0, 0, d, iters, 7u);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    double insts_per_simd = 3.0 * (double)blocks * 4 * iters * 16.0 * n_per_rep / (cus * 4.0);
    printf("{\"seq\":\"%s\",\"waves_per_simd\":%d,\"ns_per_wave_instr_per_simd\":%.4f}\n", name, waves_per_simd, ms * 1e6 / insts_per_simd);
    hipFree(d);
}
#define RUN(k, w) run(#k, k, k##_n, w, prop.multiProcessorCount)

int main() {
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    for (int w : {8, 3, 1}) {
        RUN(k_add, w);
        RUN(k_mullo, w);
        RUN(k_umad_vcc, w);
        RUN(k_smad_vcc, w);
        RUN(k_smad_sgpr, w);
        RUN(k_smad_nop, w);
        // chains: iters*4 iterations of ILP elements x 3 (signed) or 5 (unsigned, with the conditional subtraction) instructions;
        // n_per_rep is chosen so that 16 * n_per_rep * iters = ILP * instr * 4 * iters
        run("chain_signed_ilp1 (3 instr/elem)", k_chain_signed<1>, 1 * 3 * 4 / 16.0, w, prop.multiProcessorCount);
        run("chain_signed_ilp4", k_chain_signed<4>, 4 * 3 * 4 / 16.0, w, prop.multiProcessorCount);
        run("chain_signed_ilp24", k_chain_signed<24>, 24 * 3 * 4 / 16.0, w, prop.multiProcessorCount);
        run("chain_unsigned_ilp1 (5 instr/elem)", k_chain_unsigned<1>, 1 * 5 * 4 / 16.0, w, prop.multiProcessorCount);
        run("chain_unsigned_ilp4", k_chain_unsigned<4>, 4 * 5 * 4 / 16.0, w, prop.multiProcessorCount);
        run("chain_unsigned_ilp24", k_chain_unsigned<24>, 24 * 5 * 4 / 16.0, w, prop.multiProcessorCount);
    }
    return 0;
}
