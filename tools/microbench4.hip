// microbench4.hip — does the OPERAND KIND change the issue cost of a gfx950 VALU instruction?  (round 5)
// The NTT butterflies hipcc emits carry P / -P as 32-bit literals (`v_add_u32 v, 0x78000001, v`: an 8-byte encoding) and the
// Montgomery constants as SGPRs.  microbench2 measured VGPR operands only.  Same harness: REP back-to-back copies on 4 independent
// registers, 8 waves per SIMD; plus the whole butterfly as the compiler emits it (constants as literals vs passed in as arguments).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#define REP16(x) x x x x x x x x x x x x x x x x
#define KERNEL(name, INS, CB, CC)                                                                                       \
    __global__ __launch_bounds__(256) void name(uint32_t* out, int iters, uint32_t seed) {                               \
        uint32_t a0 = threadIdx.x + seed, a1 = a0 * 3u, a2 = a0 * 5u, a3 = a0 * 7u, b = seed | 1u, c = seed + 12345u;    \
        uint64_t w0 = a0, w1 = a1;                                                                                       \
        for (int it = 0; it < iters; ++it) {                                                                             \
            asm volatile(REP16(INS) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(w0), "+v"(w1) : CB(b), CC(c) : "vcc"); \
        }                                                                                                                \
        if ((a0 ^ a1 ^ a2 ^ a3 ^ (uint32_t)w0 ^ (uint32_t)w1) == 0x12345u) out[0] = a0;                                  \
    }
#define V "v"
#define S "s"

KERNEL(k_add_v, "v_add_u32 %0, %0, %6\n v_add_u32 %1, %1, %6\n v_add_u32 %2, %2, %6\n v_add_u32 %3, %3, %6\n", V, V)
KERNEL(k_add_s, "v_add_u32 %0, %6, %0\n v_add_u32 %1, %6, %1\n v_add_u32 %2, %6, %2\n v_add_u32 %3, %6, %3\n", S, V)
KERNEL(k_add_lit, "v_add_u32 %0, 0x78000001, %0\n v_add_u32 %1, 0x78000001, %1\n v_add_u32 %2, 0x78000001, %2\n v_add_u32 %3, 0x78000001, %3\n", V, V)
KERNEL(k_add_inl, "v_add_u32 %0, 17, %0\n v_add_u32 %1, 17, %1\n v_add_u32 %2, 17, %2\n v_add_u32 %3, 17, %3\n", V, V)
KERNEL(k_min_v, "v_min_u32 %0, %0, %6\n v_min_u32 %1, %1, %6\n v_min_u32 %2, %2, %6\n v_min_u32 %3, %3, %6\n", V, V)
KERNEL(k_min_s, "v_min_u32 %0, %6, %0\n v_min_u32 %1, %6, %1\n v_min_u32 %2, %6, %2\n v_min_u32 %3, %6, %3\n", S, V)
KERNEL(k_mullo_v, "v_mul_lo_u32 %0, %0, %6\n v_mul_lo_u32 %1, %1, %6\n v_mul_lo_u32 %2, %2, %6\n v_mul_lo_u32 %3, %3, %6\n", V, V)
KERNEL(k_mullo_s, "v_mul_lo_u32 %0, %0, %6\n v_mul_lo_u32 %1, %1, %6\n v_mul_lo_u32 %2, %2, %6\n v_mul_lo_u32 %3, %3, %6\n", S, V)
KERNEL(k_mad64_v, "v_mad_u64_u32 %4, vcc, %0, %6, %4\n v_mad_u64_u32 %5, vcc, %1, %6, %5\n v_mad_u64_u32 %4, vcc, %2, %6, %4\n v_mad_u64_u32 %5, vcc, %3, %6, %5\n", V, V)
KERNEL(k_mad64_s, "v_mad_u64_u32 %4, vcc, %0, %6, %4\n v_mad_u64_u32 %5, vcc, %1, %6, %5\n v_mad_u64_u32 %4, vcc, %2, %6, %4\n v_mad_u64_u32 %5, vcc, %3, %6, %5\n", S, V)
// alternating cheap / multiply-class: does a cheap instruction hide behind a multiply-class one?
KERNEL(k_mix_add_mad, "v_add_u32 %0, %0, %6\n v_mad_u64_u32 %4, vcc, %1, %6, %4\n v_add_u32 %2, %2, %6\n v_mad_u64_u32 %5, vcc, %3, %6, %5\n", V, V)
KERNEL(k_mix_add_min, "v_add_u32 %0, %0, %6\n v_min_u32 %1, %1, %6\n v_add_u32 %2, %2, %6\n v_min_u32 %3, %3, %6\n", V, V)

constexpr uint32_t P = 2013265921u, NEG_P_INV = 0x77FFFFFFu;
__device__ __forceinline__ uint32_t umin_(uint32_t a, uint32_t b) { return a < b ? a : b; }
template <bool ARGS>
__global__ __launch_bounds__(256) void k_bfly(uint32_t* out, int iters, uint32_t seed, uint32_t p_arg, uint32_t npi_arg) {
    const uint32_t p = ARGS ? p_arg : P, npi = ARGS ? npi_arg : NEG_P_INV;
    uint32_t a[8], b[8], w[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { a[j] = (threadIdx.x * 7u + j * 11u + seed) % P; b[j] = (threadIdx.x * 13u + j * 5u + seed) % P; w[j] = (seed * (j + 3u) + 99u) % P; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const uint64_t ab = (uint64_t)b[j] * w[j];
            const uint32_t m = (uint32_t)ab * npi;
            uint32_t t = (uint32_t)((ab + (uint64_t)m * p) >> 32);
            t = umin_(t, t - p);
            const uint32_t d = a[j] - t, u = a[j] + t;
            b[j] = umin_(d, d + p);
            a[j] = umin_(u, u - p);
        }
    }
    uint32_t acc = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) acc ^= a[j] ^ b[j];
    if (acc == 0x12345u) out[0] = acc;
}

// the fused-reduction butterfly of ntt_r16.hpp: u = REDC(a R + b w), d = REDC(a R + b (P - w)) — 7 multiply-class instructions
template <int ILP>
__global__ __launch_bounds__(256) void k_bfly_fused(uint32_t* out, int iters, uint32_t seed, uint32_t p_arg, uint32_t npi_arg) {
    const uint32_t R = 268435454u;
    uint32_t a[ILP], b[ILP], w[ILP], wn[ILP];
#pragma unroll
    for (int j = 0; j < ILP; ++j) { a[j] = (threadIdx.x * 7u + j * 11u + seed) % P; b[j] = (threadIdx.x * 13u + j * 5u + seed) % P; w[j] = (seed * (j + 3u) + 99u) % P; wn[j] = P - w[j]; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < ILP; ++j) {
            const uint64_t t0 = (uint64_t)a[j] * R;
            const uint64_t tu = t0 + (uint64_t)b[j] * w[j], td = t0 + (uint64_t)b[j] * wn[j];
            const uint32_t mu = (uint32_t)tu * NEG_P_INV, md = (uint32_t)td * NEG_P_INV;
            a[j] = (uint32_t)((tu + (uint64_t)mu * P) >> 32);
            b[j] = (uint32_t)((td + (uint64_t)md * P) >> 32);
        }
    }
    uint32_t acc = 0;
#pragma unroll
    for (int j = 0; j < ILP; ++j) acc ^= a[j] ^ b[j];
    if (acc == 0x12345u) out[0] = acc;
}

template <class K, class... A>
static double time_kernel(K kernel, int blocks, int iters, A... extra) {
    uint32_t* d;
    hipMalloc(&d, 64);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(kernel, dim3(blocks), dim3(256), 0, 0, d, iters, 7u, extra...);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(kernel, dim3(blocks), dim3(256), 0, 0, d, iters, 7u, extra...);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    hipFree(d);
    return ms / 5.0;
}
template <class K>
static void run(const char* name, K kernel, int blocks, double* base) {
    const int iters = 512;
    const double ms = time_kernel(kernel, blocks, iters);
    const double insts_per_simd = (double)blocks * 4 * iters * 64 / 1024.0;
    const double ns = ms * 1e6 / insts_per_simd;
    if (*base == 0) *base = ns;
    printf("{\"instr\":\"%s\",\"ns_per_wave_instr_per_simd\":%.4f,\"rel_to_v_add_vgpr\":%.2f}\n", name, ns, ns / *base);
}

int main() {
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    for (int wps : {8, 4, 2}) {
        const int blocks = prop.multiProcessorCount * wps;
        printf("{\"waves_per_simd\":%d}\n", wps);
        double base = 0;
        run("v_add_u32 vgpr", k_add_v, blocks, &base);
        run("v_add_u32 sgpr", k_add_s, blocks, &base);
        run("v_add_u32 literal", k_add_lit, blocks, &base);
        run("v_add_u32 inline-const", k_add_inl, blocks, &base);
        run("v_min_u32 vgpr", k_min_v, blocks, &base);
        run("v_min_u32 sgpr", k_min_s, blocks, &base);
        run("v_mul_lo_u32 vgpr", k_mullo_v, blocks, &base);
        run("v_mul_lo_u32 sgpr", k_mullo_s, blocks, &base);
        run("v_mad_u64_u32 vgpr", k_mad64_v, blocks, &base);
        run("v_mad_u64_u32 sgpr", k_mad64_s, blocks, &base);
        run("alternating add/mad64", k_mix_add_mad, blocks, &base);
        run("alternating add/min", k_mix_add_min, blocks, &base);
        const int iters = 512;
        for (int args = 0; args < 2; ++args) {
            const double ms = args ? time_kernel(k_bfly<true>, blocks, iters, P, NEG_P_INV) : time_kernel(k_bfly<false>, blocks, iters, P, NEG_P_INV);
            const double bf_per_simd = (double)blocks * 4 * iters * 8 / 1024.0;
            printf("{\"seq\":\"non-lazy DIT butterfly (6 mul-class + 5 cheap), constants as %s\",\"ns_per_wave_butterfly_per_simd\":%.3f,\"model_ns_at_1.9_1.05\":16.65}\n",
                   args ? "kernel arguments (SGPR)" : "literals", ms * 1e6 / bf_per_simd);
        }
        {
            const double bf8 = (double)blocks * 4 * iters * 8 / 1024.0, bf16 = (double)blocks * 4 * iters * 16 / 1024.0;
            printf("{\"seq\":\"fused-reduction butterfly (7 mul-class), 8 independent per wave\",\"ns_per_wave_butterfly_per_simd\":%.3f}\n",
                   time_kernel(k_bfly_fused<8>, blocks, iters, P, NEG_P_INV) * 1e6 / bf8);
            printf("{\"seq\":\"fused-reduction butterfly (7 mul-class), 16 independent per wave\",\"ns_per_wave_butterfly_per_simd\":%.3f}\n",
                   time_kernel(k_bfly_fused<16>, blocks, iters, P, NEG_P_INV) * 1e6 / bf16);
        }
    }
    return 0;
}
