// transcript.hpp — host-side Fiat–Shamir transcript of the segment prover (product code, not the oracle).
//
// Restates risc0_zkp::prove::write_iop::WriteIOP and core::hash::poseidon2::{Poseidon2Rng, poseidon2_mix,
// unpadded_hash} (risc0-zkp 3.0.3, reference Cargo.lock:9155).  Upstream also runs the transcript on the CPU:
// it touches a few hundred words per proof, so there is nothing to gain from the GPU here.
//
// The permutation itself is on a lone proof's critical path, though: ~220 sequential permutations per proof (the coeff_u sponge,
// the final coefficients, the query draws) sit between GPU stages, and the verifier runs ~8 000 per seal.  On x86-64 hosts with
// AVX2 (checked at run time) `mix` therefore runs an 8-lane form — the 24 cells as four vectors, lane k of vector j = cell
// 4k + j, so the M4 blocks and the S-boxes are purely vertical — 0.70 us instead of 1.05 us on the GPU box's EPYC 9575F, 1.3 instead of 2.5 on the build container's Xeon (the 21
// internal rounds are a chain of dependent S-boxes either way); same words (tests/host_arith_check.cpp runs both forms against each
// other and against the published known answer).
#pragma once
#include <stdint.h>
#include <string.h>

#include <vector>

#include "fp.hpp"

#if defined(__x86_64__) && !defined(__HIP_DEVICE_COMPILE__) && !defined(BX_NO_HOST_AVX2)
#define BX_HOST_AVX2 1
#include <immintrin.h>
#endif

namespace bx {

struct HostPoseidon2 {
    uint32_t rc[213];   // Montgomery
    uint32_t diag[24];  // Montgomery
    alignas(32) uint32_t rc_t[8][4][8];  // external rounds' constants in the vector layout: [round][j][k] = rc of cell 4k + j
    alignas(32) uint32_t diag_t[4][8];
    bool vec = false;  // use the AVX2 form
    void load(const uint32_t* rc_canon, const uint32_t* diag_canon) {
        for (int i = 0; i < 213; ++i) rc[i] = fp_encode(rc_canon[i]);
        for (int i = 0; i < 24; ++i) diag[i] = fp_encode(diag_canon[i]);
        memset(rc_t, 0, sizeof rc_t);
        memset(diag_t, 0, sizeof diag_t);
        for (int r = 0; r < 8; ++r)
            for (int i = 0; i < 24; ++i) rc_t[r][i & 3][i >> 2] = rc[(r < 4 ? 24 * r : 96 + 21 + 24 * (r - 4)) + i];
        for (int i = 0; i < 24; ++i) diag_t[i & 3][i >> 2] = diag[i];
#if defined(BX_HOST_AVX2)
        vec = __builtin_cpu_supports("avx2") != 0;
#endif
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
#if defined(BX_HOST_AVX2)
        if (vec) return mix_avx2(s);
#endif
        mix_scalar(s);
    }
    void mix_scalar(uint32_t* s) const {
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
#if defined(BX_HOST_AVX2)
#define BX_AVX2 __attribute__((target("avx2"))) static inline
    // vpmuludq as an opaque instruction: clang (the host compiler of the .hip files) otherwise folds the products by the constants P
    // and P^-1 and the masking they imply into emulated 64-bit multiplications, three vpmuludq each (0.93 us per permutation
    // instead of 0.70 on the GPU box's EPYC 9575F; gcc emits the same code either way)
    BX_AVX2 __m256i v_mulu(__m256i a, __m256i b) {
        __m256i r;
        asm("vpmuludq %2, %1, %0" : "=x"(r) : "x"(a), "x"(b));
        return r;
    }
    // eight Montgomery products: even and odd lanes through vpmuludq, t - (t P^-1 mod 2^32) P has a zero low word and its high
    // word in (-P, P); + P where negative (as unsigned words the smaller of r and r + P)
    BX_AVX2 __m256i v_mul(__m256i a, __m256i b) {
        const __m256i vp = _mm256_set1_epi32((int)P), vmu = _mm256_set1_epi32((int)P_INV);
        const __m256i te = v_mulu(a, b), to = v_mulu(_mm256_srli_epi64(a, 32), _mm256_srli_epi64(b, 32));
        const __m256i qe = v_mulu(te, vmu), qo = v_mulu(to, vmu);
        const __m256i de = _mm256_sub_epi64(te, v_mulu(qe, vp)), dx = _mm256_sub_epi64(to, v_mulu(qo, vp));
        const __m256i r = _mm256_blend_epi32(_mm256_srli_epi64(de, 32), dx, 0xAA);
        return _mm256_min_epu32(r, _mm256_add_epi32(r, vp));
    }
    BX_AVX2 __m256i v_add(__m256i a, __m256i b) {
        const __m256i s = _mm256_add_epi32(a, b);
        return _mm256_min_epu32(s, _mm256_sub_epi32(s, _mm256_set1_epi32((int)P)));
    }
    BX_AVX2 __m256i v_sbox(__m256i x) {
        const __m256i x2 = v_mul(x, x), x3 = v_mul(x2, x), x4 = v_mul(x2, x2);
        return v_mul(x3, x4);
    }
    // sum of the six cells of a vector (lanes 6, 7 are padding), in every lane
    BX_AVX2 __m256i v_hsum6(__m256i v) {
        v = _mm256_and_si256(v, _mm256_setr_epi32(-1, -1, -1, -1, -1, -1, 0, 0));
        v = v_add(v, _mm256_permute2x128_si256(v, v, 1));
        v = v_add(v, _mm256_shuffle_epi32(v, 0x4E));
        return v_add(v, _mm256_shuffle_epi32(v, 0xB1));
    }
    BX_AVX2 void v_m_ext(__m256i* v) {
        const __m256i a = v[0], b = v[1], c = v[2], d = v[3];
        const __m256i t0 = v_add(a, b), t1 = v_add(c, d), t2 = v_add(v_add(b, b), t1), t3 = v_add(v_add(d, d), t0);
        const __m256i t1x2 = v_add(t1, t1), t0x2 = v_add(t0, t0);
        const __m256i t4 = v_add(v_add(t1x2, t1x2), t3), t5 = v_add(v_add(t0x2, t0x2), t2);
        const __m256i y[4] = {v_add(t3, t5), t5, v_add(t2, t4), t4};
        for (int j = 0; j < 4; ++j) v[j] = v_add(y[j], v_hsum6(y[j]));
    }
    BX_AVX2 void v_full_round(__m256i* v, const uint32_t (*c)[8]) {
        for (int j = 0; j < 4; ++j) v[j] = v_sbox(v_add(v[j], _mm256_load_si256((const __m256i*)c[j])));
        v_m_ext(v);
    }
    __attribute__((target("avx2"))) void mix_avx2(uint32_t* s) const {
        alignas(32) uint32_t buf[4][8];
        memset(buf, 0, sizeof buf);
        for (int i = 0; i < 24; ++i) buf[i & 3][i >> 2] = s[i];
        __m256i v[4];
        for (int j = 0; j < 4; ++j) v[j] = _mm256_load_si256((const __m256i*)buf[j]);
        v_m_ext(v);
        for (int r = 0; r < 4; ++r) v_full_round(v, rc_t[r]);
        const __m256i d0 = _mm256_load_si256((const __m256i*)diag_t[0]), d1 = _mm256_load_si256((const __m256i*)diag_t[1]);
        const __m256i d2 = _mm256_load_si256((const __m256i*)diag_t[2]), d3 = _mm256_load_si256((const __m256i*)diag_t[3]);
        const __m256i not0 = _mm256_setr_epi32(0, -1, -1, -1, -1, -1, 0, 0);
        for (int r = 0; r < 21; ++r) {
            // cell 0 alone goes through the S-box (scalar: a chain of three dependent products); the other 23 cells are summed
            // meanwhile
            const uint32_t s0 = sbox(fp_add((uint32_t)_mm_cvtsi128_si32(_mm256_castsi256_si128(v[0])), rc[96 + r]));
            __m256i rest = v_add(v_add(_mm256_and_si256(v[0], not0), v[1]), v_add(v[2], v[3]));
            rest = v_hsum6(rest);
            const __m256i sum = v_add(rest, _mm256_set1_epi32((int)s0));
            v[0] = _mm256_blend_epi32(v[0], _mm256_set1_epi32((int)s0), 1);
            v[0] = v_add(v_mul(v[0], d0), sum);
            v[1] = v_add(v_mul(v[1], d1), sum);
            v[2] = v_add(v_mul(v[2], d2), sum);
            v[3] = v_add(v_mul(v[3], d3), sum);
        }
        for (int r = 4; r < 8; ++r) v_full_round(v, rc_t[r]);
        for (int j = 0; j < 4; ++j) _mm256_store_si256((__m256i*)buf[j], v[j]);
        for (int i = 0; i < 24; ++i) s[i] = buf[i & 3][i >> 2];
    }
#undef BX_AVX2
#endif
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
