// ntstore_bench.hip — what does a kernel boundary cost after a kernel that left the L2s full of dirty lines, and does writing the
// output with nontemporal stores (which do not linger in L2) shorten it?  Chain of dependent launches on one stream: a "producer"
// that writes `mb` MB (plain or nontemporal 16-byte stores) followed by a one-wave "consumer" that reads one word of it.
//   hipcc -O3 --offload-arch=gfx950 tools/ntstore_bench.hip -o /tmp/ntb && /tmp/ntb
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <bool NT>
__global__ __launch_bounds__(256) void producer(uint4* out, size_t n, uint32_t seed) {
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        uint4 v = make_uint4(seed + (uint32_t)i, seed, seed ^ (uint32_t)i, 7u);
        typedef uint32_t v4u __attribute__((ext_vector_type(4)));
        if (NT) __builtin_nontemporal_store((v4u){v.x, v.y, v.z, v.w}, reinterpret_cast<v4u*>(out + i));
        else out[i] = v;
    }
}
__global__ void consumer(const uint4* in, size_t n, uint32_t* sink) { if (threadIdx.x == 0) *sink += in[n - 1].x; }
int main() {
    hipStream_t st;
    hipStreamCreate(&st);
    uint32_t* sink;
    hipMalloc(&sink, 4);
    hipMemset(sink, 0, 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (size_t mb : {16, 64, 256, 1024}) {
        const size_t n = mb * (1u << 20) / 16;
        uint4* buf;
        hipMalloc(&buf, n * 16);
        for (int nt = 0; nt < 2; ++nt) {
            for (int with_consumer = 0; with_consumer < 2; ++with_consumer) {
                const int reps = 40;
                float best = 1e9f;
                for (int trial = 0; trial < 3; ++trial) {
                    hipEventRecord(e0, st);
                    for (int r = 0; r < reps; ++r) {
                        if (nt) hipLaunchKernelGGL(producer<true>, dim3(4096), dim3(256), 0, st, buf, n, (uint32_t)r);
                        else hipLaunchKernelGGL(producer<false>, dim3(4096), dim3(256), 0, st, buf, n, (uint32_t)r);
                        if (with_consumer) hipLaunchKernelGGL(consumer, dim3(1), dim3(64), 0, st, buf, n, sink);
                    }
                    hipEventRecord(e1, st);
                    hipEventSynchronize(e1);
                    float ms;
                    hipEventElapsedTime(&ms, e0, e1);
                    if (ms < best) best = ms;
                }
                printf("{\"mb\": %zu, \"nontemporal\": %d, \"consumer\": %d, \"us_per_iteration\": %.2f, \"GBps\": %.0f}\n", mb, nt, with_consumer, 1e3 * best / reps,
                       mb * 1.048576 / (best / reps));
            }
        }
        hipFree(buf);
    }
    return 0;
}
