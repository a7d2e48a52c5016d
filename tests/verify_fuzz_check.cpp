// verify_fuzz_check.cpp — the host verifier (boundless_amd/csrc/verify.cpp) under AddressSanitizer / UBSan against mutated seals.
//
// Built and run by tests/test_verifier_sanitizers_cpu.py:  verify_fuzz_check <seal.bin> <iterations>
// seal.bin = an honest seal (u32 words, little endian) written by the test from the CPU oracle's prover.  The honest seal must
// verify; every mutation (truncation, padding, bit flips, random words, non-canonical words, block swaps, header edits, pure
// garbage) must be rejected with an error string — and none may make the verifier read or write out of bounds, overflow a
// signed integer, shift out of range or leak (the sanitizers abort the process if one does).
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <vector>

#include "../boundless_amd/csrc/circuit.hpp"
#include "../include/bx_circuit.h"

namespace bx {
const char* synthetic_constraints_at(void*, const bx_segment_params* shape, const bx_tap_reader* taps, const uint32_t poly_mix[4],
                                     const uint32_t mix[4], const uint32_t* globals, uint32_t out[4]);
}
// the library's table lives in circuit.hip next to the device stages; the verifier only needs the host entries
extern "C" const bx_circuit_ops* bx_synthetic_circuit(void) {
    static const bx_circuit_ops ops = {nullptr, "synthetic (host entries only)", bx::synth_normalize, bx::synth_taps, bx::synth_n_globals,
                                       nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, bx::synthetic_constraints_at, nullptr,
                                       bx::synth_check_code};
    return &ops;
}

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static uint64_t rnd() {  // splitmix64
    uint64_t z = (rng_state += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

int main(int argc, char** argv) {
    if (argc < 3) return 2;
    FILE* f = fopen(argv[1], "rb");
    if (!f) return 2;
    std::vector<uint32_t> seal;
    uint32_t w;
    while (fread(&w, 4, 1, f) == 1) seal.push_back(w);
    fclose(f);
    const long iters = atol(argv[2]);
    if (const char* e = bx_verify_segment(seal.data(), seal.size())) {
        printf("honest seal rejected: %s\n", e);
        return 1;
    }
    if (bx_verify_segment(nullptr, 0) == nullptr || bx_verify_segment(seal.data(), 0) == nullptr) {
        printf("empty seal accepted\n");
        return 1;
    }
    // the explicit VerifierContext path: a context that holds the circuit's control ID for this seal's shape accepts it, one that
    // holds another ID (or the right ID under another po2) does not
    bx_verifier_ctx *good = nullptr, *bad = nullptr;
    uint32_t id[8];
    if (bx_verifier_ctx_create(&good) || bx_verifier_ctx_create(&bad) || bx_synthetic_control_id_host(seal[0], seal[1], id)) return 2;
    if (bx_verifier_ctx_add_control_id(good, seal[0], id) || bx_verifier_ctx_add_control_id(bad, seal[0] == 9 ? 10 : 9, id)) return 2;
    id[3] ^= 1u;
    if (bx_verifier_ctx_add_control_id(bad, seal[0], id)) return 2;
    if (const char* e = bx_verify_segment_with_context(seal.data(), seal.size(), nullptr, good)) {
        printf("honest seal rejected against its own control ID: %s\n", e);
        return 1;
    }
    if (bx_verify_segment_with_context(seal.data(), seal.size(), nullptr, bad) == nullptr || bx_verifier_ctx_size(bad) != 2 ||
        bx_verifier_ctx_count(bad, seal[0]) != 1) {
        printf("a context without the seal's control ID accepted it\n");
        return 1;
    }
    const uint32_t P = 2013265921u;
    long accepted = 0;
    for (long it = 0; it < iters; ++it) {
        std::vector<uint32_t> m = seal;
        const size_t n = seal.size();
        switch (rnd() % 9) {
            case 0: m.resize(rnd() % n); break;                                        // truncation
            case 1: m.resize(n + 1 + rnd() % 64, (uint32_t)rnd()); break;              // padding
            case 2: m[rnd() % n] ^= 1u << (rnd() % 32); break;                         // one bit
            case 3: m[rnd() % n] = (uint32_t)rnd(); break;                             // one random word
            case 4: m[rnd() % n] += P; break;                                          // non-canonical representative
            case 5: {                                                                  // swap two blocks
                size_t len = 1 + rnd() % 64, a = rnd() % (n - len), b = rnd() % (n - len);
                if (a == b) b = (a + len) % (n - len);
                for (size_t i = 0; i < len; ++i) std::swap(m[a + i], m[b + i]);
                bool same = true;
                for (size_t i = 0; i < n && same; ++i) same = m[i] == seal[i];
                if (same) continue;
                break;
            }
            case 6: m[rnd() % 8] = (uint32_t)(rnd() % 70000); break;                   // header / first public words
            case 7: {                                                                  // garbage behind a plausible header
                for (size_t i = 6; i < n; ++i) m[i] = (uint32_t)rnd() % P;
                break;
            }
            default: {                                                                 // a run of random canonical words
                size_t len = 1 + rnd() % 256, a = rnd() % (n - len);
                for (size_t i = 0; i < len; ++i) m[a + i] = (uint32_t)(rnd() % P);
                break;
            }
        }
        if (m.size() == n && memcmp(m.data(), seal.data(), 4 * n) == 0) continue;
        // exact-size heap copy so that a read one word past the end is an ASan report
        uint32_t* heap = (uint32_t*)malloc(m.size() * 4 + (m.empty() ? 1 : 0));
        memcpy(heap, m.data(), m.size() * 4);
        // the verdict — and its text — must not depend on how many threads share the queries: one thread reads the seal front to
        // back, several check the queries independently and report the first failing one in seal order
        std::string verdict[2];
        const int threads[2] = {1, 1 + (int)(rnd() % 7)};
        for (int v = 0; v < 2; ++v) {
            if (bx_verify_set_threads(threads[v])) return 2;
            const char* e = (it & 1) ? bx_verify_segment(heap, m.size()) : bx_verify_segment_with_context(heap, m.size(), nullptr, good);
            verdict[v] = e ? e : "";
        }
        free(heap);
        if (verdict[0] != verdict[1]) {
            printf("mutation %ld: '%s' on one thread, '%s' on %d\n", it, verdict[0].c_str(), verdict[1].c_str(), threads[1]);
            return 1;
        }
        if (verdict[0].empty()) {
            ++accepted;
            printf("mutation %ld accepted\n", it);
        }
    }
    bx_verifier_ctx_destroy(good);
    bx_verifier_ctx_destroy(bad);
    if (accepted) return 1;
    printf("verify_fuzz_check ok (%ld mutations rejected)\n", iters);
    return 0;
}
