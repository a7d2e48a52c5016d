// microbench2.hip — issue cost of individual gfx950 VALU instructions (inline asm so nothing is folded away).
// Each kernel runs REP back-to-back copies of one instruction on 4 independent register sets (ILP 4), 8 waves per SIMD.
// Output: cycles per wave-instruction per SIMD, assuming the measured effective clock printed first (s_memtime based).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

#define REP16(x) x x x x x x x x x x x x x x x x
#define BODY(INS)                                                                                    \
    for (int it = 0; it < iters; ++it) {                                                             \
        asm volatile(REP16(INS) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(w0), "+v"(w1) : "v"(b), "v"(c) : "vcc"); \
    }

#define KERNEL(name, INS)                                                                            \
    __global__ __launch_bounds__(256) void name(uint32_t* out, int iters, uint32_t seed) {          \
        uint32_t a0 = threadIdx.x + seed, a1 = a0 * 3u, a2 = a0 * 5u, a3 = a0 * 7u, b = seed | 1u, c = seed + 12345u; \
        uint64_t w0 = a0, w1 = a1;                                                                   \
        BODY(INS)                                                                                    \
        if ((a0 ^ a1 ^ a2 ^ a3 ^ (uint32_t)w0 ^ (uint32_t)w1) == 0x12345u) out[0] = a0;             \
    }

KERNEL(k_add, "v_add_u32 %0, %0, %6\n v_add_u32 %1, %1, %6\n v_add_u32 %2, %2, %6\n v_add_u32 %3, %3, %6\n")
KERNEL(k_min, "v_min_u32 %0, %0, %6\n v_min_u32 %1, %1, %6\n v_min_u32 %2, %2, %6\n v_min_u32 %3, %3, %6\n")
KERNEL(k_mov, "v_mov_b32 %0, %1\n v_mov_b32 %1, %2\n v_mov_b32 %2, %3\n v_mov_b32 %3, %0\n")
KERNEL(k_lshladd, "v_lshl_add_u32 %0, %0, 3, %6\n v_lshl_add_u32 %1, %1, 3, %6\n v_lshl_add_u32 %2, %2, 3, %6\n v_lshl_add_u32 %3, %3, 3, %6\n")
KERNEL(k_add3, "v_add3_u32 %0, %0, %6, %7\n v_add3_u32 %1, %1, %6, %7\n v_add3_u32 %2, %2, %6, %7\n v_add3_u32 %3, %3, %6, %7\n")
KERNEL(k_mullo, "v_mul_lo_u32 %0, %0, %6\n v_mul_lo_u32 %1, %1, %6\n v_mul_lo_u32 %2, %2, %6\n v_mul_lo_u32 %3, %3, %6\n")
KERNEL(k_mulhi, "v_mul_hi_u32 %0, %0, %6\n v_mul_hi_u32 %1, %1, %6\n v_mul_hi_u32 %2, %2, %6\n v_mul_hi_u32 %3, %3, %6\n")
KERNEL(k_mad64, "v_mad_u64_u32 %4, vcc, %0, %6, %4\n v_mad_u64_u32 %5, vcc, %1, %6, %5\n v_mad_u64_u32 %4, vcc, %2, %6, %4\n v_mad_u64_u32 %5, vcc, %3, %6, %5\n")
KERNEL(k_add64, "v_lshl_add_u64 %4, %4, 0, %5\n v_lshl_add_u64 %5, %5, 1, %4\n v_lshl_add_u64 %4, %4, 0, %5\n v_lshl_add_u64 %5, %5, 1, %4\n")
KERNEL(k_mul24, "v_mul_u32_u24 %0, %0, %6\n v_mul_u32_u24 %1, %1, %6\n v_mul_u32_u24 %2, %2, %6\n v_mul_u32_u24 %3, %3, %6\n")
KERNEL(k_mad24, "v_mad_u32_u24 %0, %0, %6, %7\n v_mad_u32_u24 %1, %1, %6, %7\n v_mad_u32_u24 %2, %2, %6, %7\n v_mad_u32_u24 %3, %3, %6, %7\n")
KERNEL(k_sub, "v_sub_u32 %0, %0, %6\n v_sub_u32 %1, %1, %6\n v_sub_u32 %2, %2, %6\n v_sub_u32 %3, %3, %6\n")
KERNEL(k_xor, "v_xor_b32 %0, %0, %6\n v_xor_b32 %1, %1, %6\n v_xor_b32 %2, %2, %6\n v_xor_b32 %3, %3, %6\n")

KERNEL(k_subco, "v_sub_co_u32 %0, vcc, %0, %6\n v_sub_co_u32 %1, vcc, %1, %6\n v_sub_co_u32 %2, vcc, %2, %6\n v_sub_co_u32 %3, vcc, %3, %6\n")
KERNEL(k_cndmask, "v_cndmask_b32 %0, %0, %6, vcc\n v_cndmask_b32 %1, %1, %6, vcc\n v_cndmask_b32 %2, %2, %6, vcc\n v_cndmask_b32 %3, %3, %6, vcc\n")
KERNEL(k_cmp, "v_cmp_lt_u32 vcc, %0, %6\n v_cmp_lt_u32 vcc, %1, %6\n v_cmp_lt_u32 vcc, %2, %6\n v_cmp_lt_u32 vcc, %3, %6\n")
KERNEL(k_ashr, "v_ashrrev_i32 %0, 3, %0\n v_ashrrev_i32 %1, 3, %1\n v_ashrrev_i32 %2, 3, %2\n v_ashrrev_i32 %3, 3, %3\n")
KERNEL(k_lshl, "v_lshlrev_b32 %0, 3, %0\n v_lshlrev_b32 %1, 3, %1\n v_lshlrev_b32 %2, 3, %2\n v_lshlrev_b32 %3, 3, %3\n")
KERNEL(k_and, "v_and_b32 %0, %0, %6\n v_and_b32 %1, %1, %6\n v_and_b32 %2, %2, %6\n v_and_b32 %3, %3, %6\n")
KERNEL(k_addco, "v_add_co_u32 %0, vcc, %0, %6\n v_addc_co_u32 %1, vcc, %1, %7, vcc\n v_add_co_u32 %2, vcc, %2, %6\n v_addc_co_u32 %3, vcc, %3, %7, vcc\n")
KERNEL(k_max, "v_max_u32 %0, %0, %6\n v_max_u32 %1, %1, %6\n v_max_u32 %2, %2, %6\n v_max_u32 %3, %3, %6\n")
KERNEL(k_min3, "v_min3_u32 %0, %0, %6, %7\n v_min3_u32 %1, %1, %6, %7\n v_min3_u32 %2, %2, %6, %7\n v_min3_u32 %3, %3, %6, %7\n")
KERNEL(k_mini32, "v_min_i32 %0, %0, %6\n v_min_i32 %1, %1, %6\n v_min_i32 %2, %2, %6\n v_min_i32 %3, %3, %6\n")
KERNEL(k_subrev, "v_subrev_u32 %0, %6, %0\n v_subrev_u32 %1, %6, %1\n v_subrev_u32 %2, %6, %2\n v_subrev_u32 %3, %6, %3\n")
KERNEL(k_fma32, "v_fma_f32 %0, %0, %6, %7\n v_fma_f32 %1, %1, %6, %7\n v_fma_f32 %2, %2, %6, %7\n v_fma_f32 %3, %3, %6, %7\n")
KERNEL(k_mulf32, "v_mul_f32 %0, %0, %6\n v_mul_f32 %1, %1, %6\n v_mul_f32 %2, %2, %6\n v_mul_f32 %3, %3, %6\n")
KERNEL(k_mad64_0, "v_mad_u64_u32 %4, vcc, %0, %6, 0\n v_mad_u64_u32 %5, vcc, %1, %6, 0\n v_mad_u64_u32 %4, vcc, %2, %6, 0\n v_mad_u64_u32 %5, vcc, %3, %6, 0\n")

__global__ void k_clock(unsigned long long* out) {
    unsigned long long t0 = __builtin_readcyclecounter();
    unsigned long long w0 = wall_clock64();
    for (volatile int i = 0; i < 200000; ++i) {}
    unsigned long long t1 = __builtin_readcyclecounter();
    unsigned long long w1 = wall_clock64();
    out[0] = t1 - t0;
    out[1] = w1 - w0;
}

template <class K>
static void run(const char* name, K kernel, int blocks, double* base) {
    uint32_t* d;
    hipMalloc(&d, 64);
    const int iters = 256;
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
    double insts_per_simd = 3.0 * (double)blocks * 4 /*waves per block*/ * iters * 64 /*16 x 4 instr per iter*/ / 1024.0;
    double ns_per_inst = ms * 1e6 / insts_per_simd;
    if (*base == 0) *base = ns_per_inst;
    printf("{\"instr\":\"%s\",\"ns_per_wave_instr_per_simd\":%.4f,\"rel_to_v_add\":%.2f}\n", name, ns_per_inst, ns_per_inst / *base);
    hipFree(d);
}

int main() {
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    int blocks = prop.multiProcessorCount * 8;
    double base = 0;
    run("v_add_u32", k_add, blocks, &base);
    run("v_sub_u32", k_sub, blocks, &base);
    run("v_min_u32", k_min, blocks, &base);
    run("v_xor_b32", k_xor, blocks, &base);
    run("v_mov_b32", k_mov, blocks, &base);
    run("v_lshl_add_u32", k_lshladd, blocks, &base);
    run("v_add3_u32", k_add3, blocks, &base);
    run("v_mul_u32_u24", k_mul24, blocks, &base);
    run("v_mad_u32_u24", k_mad24, blocks, &base);
    run("v_mul_lo_u32", k_mullo, blocks, &base);
    run("v_mul_hi_u32", k_mulhi, blocks, &base);
    run("v_mad_u64_u32", k_mad64, blocks, &base);
    run("v_lshl_add_u64", k_add64, blocks, &base);
    run("v_sub_co_u32", k_subco, blocks, &base);
    run("v_cndmask_b32", k_cndmask, blocks, &base);
    run("v_cmp_lt_u32", k_cmp, blocks, &base);
    run("v_ashrrev_i32", k_ashr, blocks, &base);
    run("v_lshlrev_b32", k_lshl, blocks, &base);
    run("v_and_b32", k_and, blocks, &base);
    run("v_add_co+v_addc_co", k_addco, blocks, &base);
    run("v_max_u32", k_max, blocks, &base);
    run("v_min_i32", k_mini32, blocks, &base);
    run("v_min3_u32", k_min3, blocks, &base);
    run("v_subrev_u32", k_subrev, blocks, &base);
    run("v_fma_f32", k_fma32, blocks, &base);
    run("v_mul_f32", k_mulf32, blocks, &base);
    run("v_mad_u64_u32(c=0)", k_mad64_0, blocks, &base);
    return 0;
}
