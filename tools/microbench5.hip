// microbench5.hip — how do cheap (v_add_u32) and multiply-class (v_mad_u64_u32 / v_min_u32) instructions share a gfx950 SIMD?
// microbench4 found that a strictly alternating add/mad stream costs 1.72 ns per instruction, not the 1.44 ns the two classes
// average to.  Here: runs of R adds followed by R multiply-class instructions (R = 1, 2, 4, 8, 16, 32), unequal mixes, and waves
// that run only adds next to waves that run only mads on the same SIMD.  8 and 4 waves per SIMD.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#define A4 "v_add_u32 %0, %0, %6\n v_add_u32 %1, %1, %6\n v_add_u32 %2, %2, %6\n v_add_u32 %3, %3, %6\n"
#define M4 "v_mad_u64_u32 %4, vcc, %0, %6, %4\n v_mad_u64_u32 %5, vcc, %1, %6, %5\n v_mad_u64_u32 %4, vcc, %2, %6, %4\n v_mad_u64_u32 %5, vcc, %3, %6, %5\n"
#define N4 "v_min_u32 %0, %0, %6\n v_min_u32 %1, %1, %6\n v_min_u32 %2, %2, %6\n v_min_u32 %3, %3, %6\n"
#define A1 "v_add_u32 %0, %0, %6\n"
#define A1b "v_add_u32 %1, %1, %6\n"
#define M1 "v_mad_u64_u32 %4, vcc, %2, %6, %4\n"
#define M1b "v_mad_u64_u32 %5, vcc, %3, %6, %5\n"
#define KERNEL(name, INS)                                                                                                \
    __global__ __launch_bounds__(256) void name(uint32_t* out, int iters, uint32_t seed) {                               \
        uint32_t a0 = threadIdx.x + seed, a1 = a0 * 3u, a2 = a0 * 5u, a3 = a0 * 7u, b = seed | 1u, c = seed + 12345u;    \
        uint64_t w0 = a0, w1 = a1;                                                                                       \
        for (int it = 0; it < iters; ++it) {                                                                             \
            asm volatile(INS : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(w0), "+v"(w1) : "v"(b), "v"(c) : "vcc");     \
        }                                                                                                                \
        if ((a0 ^ a1 ^ a2 ^ a3 ^ (uint32_t)w0 ^ (uint32_t)w1) == 0x12345u) out[0] = a0;                                  \
    }
// every body is 64 instructions: 32 adds + 32 multiply-class unless noted
KERNEL(k_r1, A1 M1 A1b M1b A1 M1 A1b M1b A1 M1 A1b M1b A1 M1 A1b M1b A1 M1 A1b M1b A1 M1 A1b M1b A1 M1 A1b M1b A1 M1 A1b M1b
             A1 M1 A1b M1b A1 M1 A1b M1b A1 M1 A1b M1b A1 M1 A1b M1b A1 M1 A1b M1b A1 M1 A1b M1b A1 M1 A1b M1b A1 M1 A1b M1b)
KERNEL(k_r2, A1 A1b M1 M1b A1 A1b M1 M1b A1 A1b M1 M1b A1 A1b M1 M1b A1 A1b M1 M1b A1 A1b M1 M1b A1 A1b M1 M1b A1 A1b M1 M1b
             A1 A1b M1 M1b A1 A1b M1 M1b A1 A1b M1 M1b A1 A1b M1 M1b A1 A1b M1 M1b A1 A1b M1 M1b A1 A1b M1 M1b A1 A1b M1 M1b)
KERNEL(k_r4, A4 M4 A4 M4 A4 M4 A4 M4 A4 M4 A4 M4 A4 M4 A4 M4)
KERNEL(k_r8, A4 A4 M4 M4 A4 A4 M4 M4 A4 A4 M4 M4 A4 A4 M4 M4)
KERNEL(k_r16, A4 A4 A4 A4 M4 M4 M4 M4 A4 A4 A4 A4 M4 M4 M4 M4)
KERNEL(k_r32, A4 A4 A4 A4 A4 A4 A4 A4 M4 M4 M4 M4 M4 M4 M4 M4)
KERNEL(k_min_r4, A4 N4 A4 N4 A4 N4 A4 N4 A4 N4 A4 N4 A4 N4 A4 N4)
KERNEL(k_min_r16, A4 A4 A4 A4 N4 N4 N4 N4 A4 A4 A4 A4 N4 N4 N4 N4)
KERNEL(k_a48_m16, A4 A4 A4 M4 A4 A4 A4 M4 A4 A4 A4 M4 A4 A4 A4 M4)
KERNEL(k_a16_m48, A4 M4 M4 M4 A4 M4 M4 M4 A4 M4 M4 M4 A4 M4 M4 M4)
KERNEL(k_all_a, A4 A4 A4 A4 A4 A4 A4 A4 A4 A4 A4 A4 A4 A4 A4 A4)
KERNEL(k_all_m, M4 M4 M4 M4 M4 M4 M4 M4 M4 M4 M4 M4 M4 M4 M4 M4)

// odd waves of a workgroup run only adds, even waves only mads (the four waves of a 256-thread workgroup sit on four different
// SIMDs, so with 8 workgroups per CU every SIMD holds waves of ONE kind when `by_block` is 0 ... use the block index instead)
__global__ __launch_bounds__(256) void k_split(uint32_t* out, int iters, uint32_t seed) {
    uint32_t a0 = threadIdx.x + seed, a1 = a0 * 3u, a2 = a0 * 5u, a3 = a0 * 7u, b = seed | 1u, c = seed + 12345u;
    uint64_t w0 = a0, w1 = a1;
    if (blockIdx.x & 1) {
        for (int it = 0; it < iters; ++it)
            asm volatile(A4 A4 A4 A4 A4 A4 A4 A4 A4 A4 A4 A4 A4 A4 A4 A4 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(w0), "+v"(w1) : "v"(b), "v"(c) : "vcc");
    } else {
        for (int it = 0; it < iters; ++it)
            asm volatile(M4 M4 M4 M4 M4 M4 M4 M4 M4 M4 M4 M4 M4 M4 M4 M4 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(w0), "+v"(w1) : "v"(b), "v"(c) : "vcc");
    }
    if ((a0 ^ a1 ^ a2 ^ a3 ^ (uint32_t)w0 ^ (uint32_t)w1) == 0x12345u) out[0] = a0;
}

template <class K>
static void run(const char* name, K kernel, int blocks) {
    const int iters = 512;
    uint32_t* d;
    (void)hipMalloc(&d, 64);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(kernel, dim3(blocks), dim3(256), 0, 0, d, iters, 7u);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(kernel, dim3(blocks), dim3(256), 0, 0, d, iters, 7u);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipFree(d);
    const double insts_per_simd = (double)blocks * 4 * iters * 64 / 1024.0;
    printf("{\"pattern\":\"%s\",\"ns_per_wave_instr_per_simd\":%.4f}\n", name, ms / 5.0 * 1e6 / insts_per_simd);
}

int main() {
    hipDeviceProp_t prop;
    (void)hipGetDeviceProperties(&prop, 0);
    for (int wps : {8, 4, 2}) {
        const int blocks = prop.multiProcessorCount * wps;
        printf("{\"waves_per_simd\":%d}\n", wps);
        run("all add", k_all_a, blocks);
        run("all mad64", k_all_m, blocks);
        run("add/mad runs of 1", k_r1, blocks);
        run("add/mad runs of 2", k_r2, blocks);
        run("add/mad runs of 4", k_r4, blocks);
        run("add/mad runs of 8", k_r8, blocks);
        run("add/mad runs of 16", k_r16, blocks);
        run("add/mad runs of 32", k_r32, blocks);
        run("add/min runs of 4", k_min_r4, blocks);
        run("add/min runs of 16", k_min_r16, blocks);
        run("48 add : 16 mad (runs 12/4)", k_a48_m16, blocks);
        run("16 add : 48 mad (runs 4/12)", k_a16_m48, blocks);
        run("half the workgroups all-add, half all-mad", k_split, blocks);
    }
    return 0;
}
