// host_arith_check.cpp — CPU check of the exact arithmetic source the gfx950 Poseidon2/NTT kernels are built from
// (boundless_amd/csrc/fp.hpp, poseidon2_arith.hpp), compiled with -DBX_CHECK_BOUNDS so every documented bound and every
// 64-bit accumulation is asserted while extreme and random operands are pushed through.  Exact results come from
// unsigned __int128 arithmetic mod P.  Built and run by tests/test_host_arith_cpu.py.
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "fp.hpp"
#include "poseidon2_arith.hpp"
#include "poseidon2_params.hpp"
#include "transcript.hpp"

using namespace bx;
typedef unsigned __int128 u128;

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static uint64_t rnd64() {
    uint64_t z = (rng_state += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
static const uint64_t RINV = 943718400ull;  // 2^-32 mod P
static uint32_t redc_exact(u128 v) { return (uint32_t)((v % P) * RINV % P); }
#define REQUIRE(cond)                                                     \
    do {                                                                  \
        if (!(cond)) {                                                    \
            fprintf(stderr, "FAILED %s (line %d)\n", #cond, __LINE__);    \
            return 1;                                                     \
        }                                                                 \
    } while (0)

int main() {
    REQUIRE((u128)RINV * ((u128)1 << 32) % P == 1);
    // ---- canonical field ops at the edges and at random ----
    const uint32_t edge[] = {0, 1, 2, P - 1, P - 2, (P - 1) / 2, MONT_ONE, R2, R3, 0x7fffffffu % P};
    for (uint32_t a : edge)
        for (uint32_t b : edge) {
            REQUIRE(fp_mul(a, b) == redc_exact((u128)a * b));
            REQUIRE(fp_add(a, b) == (uint32_t)(((uint64_t)a + b) % P));
            REQUIRE(fp_sub(a, b) == (uint32_t)(((uint64_t)a + P - b) % P));
        }
    for (int i = 0; i < 2000000; ++i) {
        uint32_t a = (uint32_t)(rnd64() % P), b = (uint32_t)(rnd64() % P);
        REQUIRE(fp_mul(a, b) == redc_exact((u128)a * b));
    }
    // ---- lazy multiply-add across its whole contract: a*b + c < 2^64 - (2^32-1)P ----
    for (int i = 0; i < 2000000; ++i) {
        uint32_t a = (uint32_t)rnd64(), b = (uint32_t)rnd64(), c = (uint32_t)rnd64();
        u128 v = (u128)a * b + c;
        if (v + (u128)0xffffffffu * P >= ((u128)1 << 64)) continue;
        uint32_t r = fp_mad_lazy(a, b, c);
        REQUIRE(r % P == redc_exact(v));
        REQUIRE((u128)r <= v / ((u128)1 << 32) + P);
    }
    // ---- S-box on the whole admissible input range incl. its upper edge ----
    const uint64_t sb_in[] = {0, 1, P - 1, P, P + 1, B_RED64 - 1, B_RED64};
    for (uint64_t x : sb_in) {
        uint32_t r = sbox7_bounded((uint32_t)x);
        u128 xr = x % P, e = xr;
        // Montgomery: sbox7(x) == x^7 * 2^(-6*32)
        for (int k = 0; k < 6; ++k) e = (u128)redc_exact(e * xr);
        REQUIRE(r % P == (uint32_t)e && r <= B_SBOX_OUT);
    }
    for (int i = 0; i < 500000; ++i) {
        uint32_t x = (uint32_t)(rnd64() % (B_RED64 + 1));
        uint32_t r = sbox7_bounded(x);
        u128 xr = x % P, e = xr;
        for (int k = 0; k < 6; ++k) e = (u128)redc_exact(e * xr);
        REQUIRE(r % P == (uint32_t)e);
    }
    // ---- the S-box without its final subtraction (external rounds): same residue, documented wide bound ----
    for (uint64_t x : sb_in) {
        uint32_t r = sbox7_wide((uint32_t)x);
        REQUIRE(r % P == sbox7_bounded((uint32_t)x) % P && r <= B_SBOX_WIDE);
    }
    for (int i = 0; i < 500000; ++i) {
        uint32_t x = (uint32_t)(rnd64() % (B_RED64 + 1));
        uint32_t r = sbox7_wide(x);
        REQUIRE(r % P == sbox7_bounded(x) % P && r <= B_SBOX_WIDE);
    }
    // ---- red64 over y < 2^39 incl. the edges ----
    const uint64_t ys[] = {0, 1, P, 0xffffffffull, 0x100000000ull, ((uint64_t)1 << 38) - 1, 112ull * B_SBOX_OUT,
                           ((uint64_t)1 << 39) - 1, 112ull * B_SBOX_WIDE};
    for (uint64_t y : ys)
        for (uint32_t add : edge) {
            // REDC(y_lo*2^32 + y_hi*2^64 + add_rr) = y + add_rr * 2^-32: the table stores (Montgomery rc) * 2^32
            uint32_t a_plain = redc_exact(add);
            uint32_t r = red64_lazy(y, add);
            REQUIRE(r % P == (uint32_t)(((u128)(y % P) * ((u128)1) + a_plain) % P));
            REQUIRE(red64(y, add) == r % P);
        }
    for (int i = 0; i < 500000; ++i) {
        uint64_t y = rnd64() >> 25;
        uint32_t add = (uint32_t)(rnd64() % P);
        REQUIRE(red64_lazy(y, add) % P == (uint32_t)((y % P + redc_exact(add)) % P));
    }
    // ---- the external layer (cells up to 2.05423 P) at its extremes ----
    {
        uint32_t s[24];
        uint64_t y[24];
        const int M4[4][4] = {{5, 7, 1, 3}, {4, 6, 1, 1}, {1, 3, 5, 7}, {1, 1, 4, 6}};
        for (int t = 0; t < 20001; ++t) {
            for (int i = 0; i < 24; ++i) s[i] = t == 0 ? (uint32_t)B_SBOX_WIDE : (uint32_t)(rnd64() % (B_SBOX_WIDE + 1));
            m_ext64w(s, y);
            for (int i = 0; i < 24; ++i) {
                u128 want = 0;
                for (int j = 0; j < 24; ++j) want += (u128)M4[i & 3][j & 3] * (i / 4 == j / 4 ? 2 : 1) * s[j];
                REQUIRE((u128)y[i] == want && (y[i] >> 39) == 0);
            }
        }
    }
    // ---- internal-round sum_r at its edges ----
    for (uint64_t sum : {(uint64_t)0, (uint64_t)P, (uint64_t)0xffffffffull, (uint64_t)B_SBOX_OUT + 23 * B_INT_CELL, ((uint64_t)1 << 37) - 1})
        REQUIRE(internal_sum_r(sum) == (uint32_t)((u128)(sum % P) * (((u128)1 << 32) % P) % P));
    // ---- the whole permutation, device order and arithmetic, vs the plain canonical implementation ----
    {
        HostPoseidon2 ref;
        ref.load(POSEIDON2_RC, POSEIDON2_DIAG);
        uint32_t prm[240];
        memset(prm, 0, sizeof prm);
        for (int i = 0; i < 213; ++i) prm[i] = fp_encode(fp_encode(POSEIDON2_RC[i]));
        for (int i = 0; i < 24; ++i) prm[216 + i] = fp_encode(POSEIDON2_DIAG[i]);
        const uint32_t pool[] = {0, 1, 2, P - 1, P - 2, (P - 1) / 2, MONT_ONE, R2};
        for (int t = 0; t < 6000; ++t) {
            uint32_t a[24], b[24];
            for (int i = 0; i < 24; ++i) {
                uint64_t r = rnd64();
                a[i] = t < 2000 ? pool[r % 8] : (t == 2000 ? P - 1 : (uint32_t)(r % P));
            }
            memcpy(b, a, sizeof a);
            ref.mix(a);
            poseidon2_mix_bounded<216>(b, prm);
            REQUIRE(memcmp(a, b, sizeof a) == 0);
        }
        // published KAT through the bounded form
        uint32_t k[24];
        for (int i = 0; i < 24; ++i) k[i] = fp_encode((uint32_t)i);
        poseidon2_mix_bounded<216>(k, prm);
        REQUIRE(fp_decode(k[0]) == 0x2ed3e23du && fp_decode(k[1]) == 0x12921fb0u && fp_decode(k[23]) == 0x57a99864u);
    }
    printf("host_arith_check ok\n");
    return 0;
}
