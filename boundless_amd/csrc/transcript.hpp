// transcript.hpp — host-side Fiat–Shamir transcript of the segment prover (product code, not the oracle).
//
// Restates risc0_zkp::prove::write_iop::WriteIOP and core::hash::poseidon2::{Poseidon2Rng, poseidon2_mix,
// unpadded_hash} (risc0-zkp 3.0.3, reference Cargo.lock:9155).  Upstream also runs the transcript on the CPU:
// it touches a few hundred words per proof, so there is nothing to gain from the GPU here.
#pragma once
#include <stdint.h>
#include <string.h>

#include <vector>

#include "fp.hpp"

namespace bx {

struct HostPoseidon2 {
    uint32_t rc[213];   // Montgomery
    uint32_t diag[24];  // Montgomery
    void load(const uint32_t* rc_canon, const uint32_t* diag_canon) {
        for (int i = 0; i < 213; ++i) rc[i] = fp_encode(rc_canon[i]);
        for (int i = 0; i < 24; ++i) diag[i] = fp_encode(diag_canon[i]);
    }
    static uint32_t sbox(uint32_t x) {
        uint32_t x2 = fp_mul(x, x), x3 = fp_mul(x2, x), x4 = fp_mul(x2, x2);
        return fp_mul(x3, x4);
    }
    static void m_ext(uint32_t* s) {
        // y_k = M4 x_k + T,  T = sum_k M4 x_k  (circ(2*M4, M4, ..., M4)); M4 = [[5,7,1,3],[4,6,1,1],[1,3,5,7],[1,1,4,6]] by the
        // Poseidon2 addition chain, unreduced in 64 bits (rows of the whole layer sum to <= 112: < 2^38), one reduction per cell
        uint64_t y[24], t[4] = {0, 0, 0, 0};
        for (int k = 0; k < 24; k += 4) {
            const uint64_t a = s[k], b = s[k + 1], c = s[k + 2], d = s[k + 3];
            const uint64_t t0 = a + b, t1 = c + d, t2 = 2 * b + t1, t3 = 2 * d + t0, t4 = 4 * t1 + t3, t5 = 4 * t0 + t2;
            y[k] = t3 + t5;
            y[k + 1] = t5;
            y[k + 2] = t2 + t4;
            y[k + 3] = t4;
            for (int j = 0; j < 4; ++j) t[j] += y[k + j];
        }
        for (int i = 0; i < 24; ++i) s[i] = (uint32_t)((y[i] + t[i & 3]) % P);
    }
    void m_int(uint32_t* s) const {
        uint64_t acc = 0;
        for (int i = 0; i < 24; ++i) acc += s[i];
        const uint32_t sum = (uint32_t)(acc % P);
        for (int i = 0; i < 24; ++i) s[i] = fp_add(sum, fp_mul(diag[i], s[i]));
    }
    void mix(uint32_t* s) const {
        m_ext(s);
        const uint32_t* c = rc;
        for (int r = 0; r < 4; ++r, c += 24) {
            for (int i = 0; i < 24; ++i) s[i] = sbox(fp_add(s[i], c[i]));
            m_ext(s);
        }
        for (int r = 0; r < 21; ++r) {
            s[0] = sbox(fp_add(s[0], *c++));
            m_int(s);
        }
        for (int r = 0; r < 4; ++r, c += 24) {
            for (int i = 0; i < 24; ++i) s[i] = sbox(fp_add(s[i], c[i]));
            m_ext(s);
        }
    }
    // unpadded_hash over Montgomery words
    void hash_elems(uint32_t out[8], const uint32_t* elems, size_t n) const {
        uint32_t s[24];
        memset(s, 0, sizeof s);
        size_t unmixed = 0;
        for (size_t i = 0; i < n; ++i) {
            s[unmixed++] = elems[i];
            if (unmixed == 16) {
                mix(s);
                unmixed = 0;
            }
        }
        if (unmixed != 0 || n == 0) {
            for (size_t i = unmixed; i < 16; ++i) s[i] = 0;
            mix(s);
        }
        memcpy(out, s, 32);
    }
};

// WriteIOP: the seal is the concatenation of everything written; `commit` feeds the Poseidon2 RNG.
struct Transcript {
    const HostPoseidon2* h;
    std::vector<uint32_t> seal;
    uint32_t cells[24];
    unsigned pool_used;
    explicit Transcript(const HostPoseidon2* hp) : h(hp) { reset(); }
    void reset() {
        seal.clear();
        memset(cells, 0, sizeof cells);
        pool_used = 0;
    }
    void write(const uint32_t* w, size_t n) { seal.insert(seal.end(), w, w + n); }
    void commit(const uint32_t digest[8]) {
        if (pool_used != 0) {
            h->mix(cells);
            pool_used = 0;
        }
        for (int i = 0; i < 8; ++i) cells[i] = fp_add(cells[i], digest[i]);
        h->mix(cells);
    }
    uint32_t random_elem() {
        if (pool_used == 16) {
            h->mix(cells);
            pool_used = 0;
        }
        return cells[pool_used++];
    }
    Fp4 random_ext() {
        Fp4 r;
        for (int k = 0; k < 4; ++k) r.c[k] = random_elem();
        return r;
    }
    uint32_t random_bits(unsigned bits) {
        uint32_t val = fp_decode(random_elem());
        for (int i = 0; i < 3; ++i) {
            uint32_t nv = fp_decode(random_elem());
            if (val == 0) val = nv;
        }
        return bits >= 32 ? val : (val & ((1u << bits) - 1u));
    }
};

}  // namespace bx
