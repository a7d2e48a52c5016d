// boundary_bench.hip — the gap between two dependent kernels on one stream as a function of how long the first one ran.
// `spin` occupies the whole chip for a fixed WALL-CLOCK time (wall_clock64, 100 MHz), so chain time - sum of the spins = the gaps.
//   hipcc -O3 --offload-arch=gfx950 tools/boundary_bench.hip -o /tmp/bb && /tmp/bb
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(256) void spin(unsigned long long ticks, uint32_t* sink) {
    const unsigned long long t0 = wall_clock64();
    uint32_t x = threadIdx.x;
    while (wall_clock64() - t0 < ticks) x = x * 1664525u + 1013904223u;
    if (x == 0xdeadbeefu) *sink = x;
}
__global__ void tiny(uint32_t* sink) { if (threadIdx.x == 0) *sink += 1; }
int main() {
    hipStream_t st;
    (void)hipStreamCreate(&st);
    uint32_t* sink;
    (void)hipMalloc(&sink, 4);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    int rate_khz = 0;
    (void)hipDeviceGetAttribute(&rate_khz, hipDeviceAttributeWallClockRate, 0);
    for (int us : {20, 100, 500, 2000, 5000}) {
        for (int with_tiny = 0; with_tiny < 2; ++with_tiny) {
            const int reps = us >= 2000 ? 20 : 60;
            const unsigned long long ticks = (unsigned long long)us * (unsigned long long)rate_khz / 1000ull;
            float best = 1e9f;
            for (int trial = 0; trial < 3; ++trial) {
                (void)hipEventRecord(e0, st);
                for (int r = 0; r < reps; ++r) {
                    hipLaunchKernelGGL(spin, dim3(2048), dim3(256), 0, st, ticks, sink);
                    if (with_tiny) hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, st, sink);
                }
                (void)hipEventRecord(e1, st);
                (void)hipEventSynchronize(e1);
                float ms;
                (void)hipEventElapsedTime(&ms, e0, e1);
                if (ms < best) best = ms;
            }
            printf("{\"spin_us\": %d, \"tiny_after\": %d, \"us_per_iteration\": %.2f, \"overhead_us_per_iteration\": %.2f}\n", us, with_tiny, 1e3 * best / reps,
                   1e3 * best / reps - us);
        }
    }
    return 0;
}
