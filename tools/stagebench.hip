// stagebench.hip — what it costs the host to get an 80 MB segment from pageable memory into HBM (MI355X, ROCm 7.2):
//   (a) memcpy into a pinned staging buffer + hipMemcpyAsync (what bx_prover_submit_segment does)
//   (b) hipHostRegister the caller's buffer in place + hipMemcpyAsync + hipHostUnregister
//   (c) hipMemcpy straight from pageable memory (the runtime stages it itself)
// build: hipcc -O2 --offload-arch=gfx950 tools/stagebench.hip -o tools/stagebench ; prints one JSON line per variant.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("{\"error\": \"%s at %s\"}\n", hipGetErrorString(e_), #x); return 1; } } while (0)
int main(int argc, char** argv) {
    const size_t n = argc > 1 ? strtoull(argv[1], nullptr, 10) : 80000000;
    const int reps = 10;
    char* src[4];
    for (auto& p : src) { p = (char*)malloc(n); memset(p, 1, n); }  // several source buffers: a fresh blob per proof, not a cache-hot one
    char *pinned = nullptr, *dev = nullptr;
    hipStream_t s;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    CK(hipHostMalloc((void**)&pinned, n, hipHostMallocDefault));
    CK(hipMalloc((void**)&dev, n));
    for (int variant = 0; variant < 3; ++variant) {
        double host = 0, total = 0;
        for (int r = 0; r < reps + 2; ++r) {
            char* p = src[r & 3];
            p[r] = (char)r;
            double t0 = now(), t1;
            if (variant == 0) {
                memcpy(pinned, p, n);
                CK(hipMemcpyAsync(dev, pinned, n, hipMemcpyHostToDevice, s));
                t1 = now();
                CK(hipStreamSynchronize(s));
            } else if (variant == 1) {
                CK(hipHostRegister(p, n, hipHostRegisterDefault));
                CK(hipMemcpyAsync(dev, p, n, hipMemcpyHostToDevice, s));
                t1 = now();
                CK(hipStreamSynchronize(s));
                double t2 = now();
                CK(hipHostUnregister(p));
                t1 += now() - t2;  // unregistering is host time as well
            } else {
                CK(hipMemcpy(dev, p, n, hipMemcpyHostToDevice));
                t1 = now();
            }
            double t3 = now();
            if (r >= 2) { host += t1 - t0; total += t3 - t0; }
        }
        const char* names[] = {"memcpy_to_pinned_then_async", "host_register_in_place_then_async", "hipMemcpy_from_pageable"};
        printf("{\"variant\": \"%s\", \"bytes\": %zu, \"host_ms\": %.3f, \"total_ms\": %.3f}\n", names[variant], n, 1e3 * host / reps, 1e3 * total / reps);
    }
    return 0;
}
