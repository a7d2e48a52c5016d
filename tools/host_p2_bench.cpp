// host_p2_bench.cpp — the host Poseidon2 permutation (boundless_amd/csrc/transcript.hpp): scalar form vs the run-time selected form, words
// compared on 20 000 states, then timed.  g++ -O3 -std=c++17 -Iboundless_amd/csrc tools/host_p2_bench.cpp -o /tmp/hp2 && /tmp/hp2
#include <chrono>
#include <cstdio>
#include "transcript.hpp"
#include "poseidon2_params.hpp"
using namespace bx;
int main() {
    HostPoseidon2 h;
    h.load(POSEIDON2_RC, POSEIDON2_DIAG);
    uint32_t s[24], t[24];
    uint64_t z = 88172645463325252ull;
    int bad = 0;
    for (int it = 0; it < 20000; ++it) {
        for (int i = 0; i < 24; ++i) { z ^= z << 13; z ^= z >> 7; z ^= z << 17; s[i] = it % 7 == 0 ? (i % 3 ? P - 1 : 0) : (uint32_t)(z % P); t[i] = s[i]; }
        h.mix_scalar(s);
        h.mix(t);
        if (memcmp(s, t, sizeof s)) ++bad;
    }
    printf("vec=%d mismatches=%d\n", (int)h.vec, bad);
    for (int i = 0; i < 24; ++i) s[i] = fp_encode(i + 1);
    for (int rep = 0; rep < 2; ++rep) {
        auto t0 = std::chrono::steady_clock::now();
        const int n = 50000;
        for (int i = 0; i < n; ++i) rep ? h.mix(s) : h.mix_scalar(s);
        double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
        printf("%s %.3f us per permutation (%u)\n", rep ? "mix" : "scalar", us / n, s[0]);
    }
}
