// waitbench.hip — which host wait primitive actually lets the thread sleep on this ROCm?  
// A ~20 ms kernel is launched and waited for in five ways; for each the waiting thread's own CPU time
// (CLOCK_THREAD_CPUTIME_ID) and the wall time are printed as JSON lines.
//   hipcc --offload-arch=gfx950 -O2 -o tools/waitbench tools/waitbench.hip && tools/waitbench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <time.h>
#include <unistd.h>

__global__ void spin_kernel(unsigned long long cycles, unsigned* out) {
    unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < cycles) {}
    if (out) *out = 1;
}
static double now(clockid_t c) {
    timespec t;
    clock_gettime(c, &t);
    return t.tv_sec + 1e-9 * t.tv_nsec;
}
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("{\"error\": \"%s: %s\"}\n", #x, hipGetErrorString(e)); return 1; } } while (0)

int main(int argc, char** argv) {
    const bool sched_blocking = argc > 1 && argv[1][0] == 'b';
    if (sched_blocking) CK(hipSetDeviceFlags(hipDeviceScheduleBlockingSync));
    hipStream_t s;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    hipEvent_t eb, ed;
    CK(hipEventCreateWithFlags(&eb, hipEventBlockingSync | hipEventDisableTiming));
    CK(hipEventCreateWithFlags(&ed, hipEventDisableTiming));
    unsigned* flag;
    CK(hipHostMalloc((void**)&flag, 4, hipHostMallocDefault));
    const unsigned long long cyc = 2000000ull;  // wall_clock64 ticks at 100 MHz: 20 ms
    hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, s, 1000ull, nullptr);
    CK(hipStreamSynchronize(s));
    for (int mode = 0; mode < 5; ++mode) {
        double cpu = 0, wall = 0;
        for (int rep = 0; rep < 5; ++rep) {
            *flag = 0;
            hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, s, cyc, flag);
            double c0 = now(CLOCK_THREAD_CPUTIME_ID), w0 = now(CLOCK_MONOTONIC);
            switch (mode) {
                case 0: CK(hipStreamSynchronize(s)); break;
                case 1: CK(hipEventRecord(eb, s)); CK(hipEventSynchronize(eb)); break;
                case 2: CK(hipEventRecord(ed, s)); CK(hipEventSynchronize(ed)); break;
                case 3: CK(hipEventRecord(ed, s)); while (hipEventQuery(ed) == hipErrorNotReady) usleep(50); break;
                case 4: while (*(volatile unsigned*)flag == 0) usleep(50); CK(hipStreamSynchronize(s)); break;
            }
            cpu += now(CLOCK_THREAD_CPUTIME_ID) - c0;
            wall += now(CLOCK_MONOTONIC) - w0;
        }
        static const char* names[] = {"hipStreamSynchronize", "blocking event: record + hipEventSynchronize", "plain event: record + hipEventSynchronize",
                                      "hipEventQuery + usleep(50)", "pinned flag written by the kernel + usleep(50)"};
        printf("{\"device_flags\": \"%s\", \"wait\": \"%s\", \"thread_cpu_ms_per_wait\": %.3f, \"wall_ms_per_wait\": %.3f}\n",
               sched_blocking ? "hipDeviceScheduleBlockingSync" : "default", names[mode], cpu / 5 * 1e3, wall / 5 * 1e3);
    }
    return 0;
}
