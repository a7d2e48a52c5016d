// microbench.hip — VALU throughput of the candidate BabyBear modmul sequences on gfx950.
// Build: hipcc -O3 --offload-arch=gfx950 -I boundless_amd/csrc tools/microbench.hip -o tools/microbench
// Run on the GPU box; prints one JSON line per variant (G-ops/s over the whole chip).
#include <hip/hip_runtime.h>
#include <stdio.h>

#include "fp.hpp"
using namespace bx;

constexpr int ILP = 8;
constexpr int ITERS = 4096;

struct OpAdd { __device__ static uint32_t f(uint32_t x, uint32_t y) { return x + y; } };
struct OpFpAdd { __device__ static uint32_t f(uint32_t x, uint32_t y) { return fp_add(x, y); } };
struct OpMulLo { __device__ static uint32_t f(uint32_t x, uint32_t y) { return x * y; } };
struct OpMulHi { __device__ static uint32_t f(uint32_t x, uint32_t y) { return __umulhi(x, y) + y; } };
struct OpMul24 { __device__ static uint32_t f(uint32_t x, uint32_t y) { return __umul24(x, y) + 1u; } };
struct OpMad64 {
    __device__ static uint32_t f(uint32_t x, uint32_t y) {
        uint64_t r = (uint64_t)x * y + x;
        return (uint32_t)r ^ (uint32_t)(r >> 32);
    }
};
struct OpLshlAdd { __device__ static uint32_t f(uint32_t x, uint32_t y) { return (x << 27) + y; } };
struct OpFpMul { __device__ static uint32_t f(uint32_t x, uint32_t y) { return fp_mul(x, y); } };
struct OpFpMulAsm {
    __device__ static uint32_t f(uint32_t a, uint32_t b) {
        uint64_t ab = (uint64_t)a * b;
        uint32_t lo = (uint32_t)ab, hi = (uint32_t)(ab >> 32), t;
        asm("v_lshl_add_u32 %0, %1, 27, %1\n\tv_lshl_add_u32 %0, %1, 31, %0" : "=&v"(t) : "v"(lo));
        uint32_t u = __umulhi(t, P);
        uint32_t d = hi - u;
        return umin(d, d + P);
    }
};
struct OpFpMulRisc0 {
    __device__ static uint32_t f(uint32_t a, uint32_t b) {
        uint64_t o = (uint64_t)a * b;
        uint32_t low = 0u - (uint32_t)o;
        uint32_t red = 0x88000001u * low;
        o += (uint64_t)red * P;
        uint32_t r = (uint32_t)(o >> 32);
        return r >= P ? r - P : r;
    }
};
struct OpFpMulSplit {  // separate mul_lo / mul_hi instead of mad_u64
    __device__ static uint32_t f(uint32_t a, uint32_t b) {
        uint32_t lo = a * b, hi = __umulhi(a, b);
        uint32_t t = lo * 0x88000001u;
        uint32_t u = __umulhi(t, P);
        uint32_t d = hi - u;
        return umin(d, d + P);
    }
};

template <class Op>
__global__ __launch_bounds__(256) void k_u32(uint32_t* out, uint32_t seed) {
    uint32_t x[ILP], y = seed | 1u;
    for (int i = 0; i < ILP; ++i) x[i] = (threadIdx.x * 2654435761u + i * 40503u + seed) % P;
    y = y % P;
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int i = 0; i < ILP; ++i) x[i] = Op::f(x[i], y);
    }
    uint32_t acc = 0;
    for (int i = 0; i < ILP; ++i) acc ^= x[i];
    if (acc == 0x12345678u) out[0] = acc;
}
__global__ __launch_bounds__(256) void k_f64(double* out, double seed) {
    double x[ILP], y = seed;
    for (int i = 0; i < ILP; ++i) x[i] = threadIdx.x * 1e-3 + i;
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int i = 0; i < ILP; ++i) x[i] = __builtin_fma(x[i], y, 0.5);
    }
    double acc = 0;
    for (int i = 0; i < ILP; ++i) acc += x[i];
    if (acc == 1.2345) out[0] = acc;
}
// exact a*b mod P in doubles (canonical representation), see DESIGN.md "modmul candidates"
__device__ __forceinline__ double fpmul_f64(double a, double b) {
    const double p = 2013265921.0, pinv = 1.0 / 2013265921.0;
    double h = a * b;
    double l = __builtin_fma(a, b, -h);
    double q = __builtin_rint(h * pinv);
    double r = __builtin_fma(-q, p, h) + l;
    r = r < 0 ? r + p : r;
    r = r >= p ? r - p : r;
    return r;
}
__global__ __launch_bounds__(256) void k_fpmul_f64(double* out, double seed) {
    double x[ILP], y = seed;
    for (int i = 0; i < ILP; ++i) x[i] = (double)((threadIdx.x * 2654435761u + i * 40503u) % P);
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int i = 0; i < ILP; ++i) x[i] = fpmul_f64(x[i], y);
    }
    double acc = 0;
    for (int i = 0; i < ILP; ++i) acc += x[i];
    if (acc == 1.2345) out[0] = acc;
}

template <class F>
static void run(const char* name, F launch, int blocks) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    launch();
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) launch();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    double ops = 5.0 * blocks * 256.0 * ILP * ITERS;
    printf("{\"variant\":\"%s\",\"gops\":%.1f,\"ms\":%.3f}\n", name, ops / (ms * 1e6), ms / 5);
}

int main() {
    uint32_t* d;
    hipMalloc(&d, 1024);
    double* dd = (double*)d;
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    int blocks = prop.multiProcessorCount * 8;
    printf("{\"device\":\"%s\",\"cus\":%d,\"clock_mhz\":%d}\n", prop.name, prop.multiProcessorCount, prop.clockRate / 1000);
#define RUN(name, Op) run(name, [&] { hipLaunchKernelGGL(k_u32<Op>, dim3(blocks), dim3(256), 0, 0, d, 12345u); }, blocks)
    RUN("v_add_u32", OpAdd);
    RUN("fp_add", OpFpAdd);
    RUN("v_lshl_add_u32", OpLshlAdd);
    RUN("v_mul_lo_u32", OpMulLo);
    RUN("v_mul_hi_u32(+add)", OpMulHi);
    RUN("v_mul_u32_u24(+add)", OpMul24);
    RUN("v_mad_u64_u32(+xor)", OpMad64);
    RUN("fp_mul(product)", OpFpMul);
    RUN("fp_mul(asm shift-add)", OpFpMulAsm);
    RUN("fp_mul(risc0 form)", OpFpMulRisc0);
    RUN("fp_mul(split lo/hi)", OpFpMulSplit);
    run("v_fma_f64", [&] { hipLaunchKernelGGL(k_f64, dim3(blocks), dim3(256), 0, 0, dd, 1.0000001); }, blocks);
    run("fp_mul(f64)", [&] { hipLaunchKernelGGL(k_fpmul_f64, dim3(blocks), dim3(256), 0, 0, dd, 123456789.0); }, blocks);
    return 0;
}
