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
    // ---- the returns to 32 bits over y < 2^39 incl. the edges ----
    const uint64_t ys[] = {0, 1, P, 0xffffffffull, 0x100000000ull, ((uint64_t)1 << 38) - 1, 112ull * B_SBOX_OUT,
                           ((uint64_t)1 << 39) - 1, 112ull * B_SBOX_WIDE};
    const uint32_t rinv = cx_rpow(-1);
    const auto red_mid = [](uint64_t y, uint32_t a) { return red64k_lazy<K1_MID, K2_MID>(y, a); };
    const auto red_end = [](uint64_t y, uint32_t a) { return red64k_lazy<K1_END, K2_END>(y, a); };
    for (uint64_t y : ys)
        for (uint32_t add : edge) {
            const uint32_t a = add % P;
            // bare REDC: (y + rc) * R^-1
            uint32_t r = redc64(y, a);
            REQUIRE(r % P == (uint32_t)((u128)((y + a) % P) * rinv % P) && r <= B_REDC);
            // representation-changing REDC: (y * K1 + add) * R^-1
            uint32_t m = red_mid(y, a), e = red_end(y, a);
            REQUIRE(m % P == (uint32_t)(((u128)(y % P) * K1_MID + a) % P * rinv % P) && m <= B_RED64);
            REQUIRE(e % P == (uint32_t)(((u128)(y % P) * K1_END + a) % P * rinv % P) && e <= B_END);
        }
    for (int i = 0; i < 500000; ++i) {
        uint64_t y = rnd64() >> 25;
        uint32_t a = (uint32_t)(rnd64() % P);
        REQUIRE(redc64(y, a) % P == (uint32_t)((u128)((y + a) % P) * rinv % P));
        REQUIRE(red_mid(y, a) % P == (uint32_t)(((u128)(y % P) * K1_MID + a) % P * rinv % P));
    }
    // the representation constants are what the exponent bookkeeping says (R = 2^32 mod P)
    {
        const auto rp = [](int64_t e) { return cx_rpow(e); };
        REQUIRE(cx_mul(rp(1), rp(-1)) == 1 && rp(1) == MONT_ONE && rp(2) == R2 && rp(3) == R3);
        int64_t e = 1;                      // canonical Montgomery input
        e -= 1;                             // initial layer, bare REDC
        for (int r = 0; r < 3; ++r) e = 7 * e - 6 - 1;  // S-box, layer, bare REDC
        REQUIRE(e == -399);
        REQUIRE(cx_mul(rp(7 * e - 6), K1_MID) == rp(2));  // (y * K1_MID) * R^-1 is back at R^1
        e = 1;                              // after the internal rounds
        e = 7 * e - 6 - 1;                  // round 4
        for (int r = 0; r < 2; ++r) e = 7 * e - 6 - 1;  // rounds 5, 6
        REQUIRE(e == -56 && cx_mul(rp(7 * e - 6), K1_END) == rp(2));
        REQUIRE(p2_rc_scale(0) == rp(1) && p2_rc_scale(24) == rp(-6) && p2_rc_scale(48) == rp(-55) && p2_rc_scale(72) == rp(-398));
        REQUIRE(p2_rc_scale(96) == R2 && p2_rc_scale(117) == R2 && p2_rc_scale(141) == rp(1) && p2_rc_scale(165) == rp(-6) &&
                p2_rc_scale(189) == rp(-55));
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
        for (int i = 0; i < 213; ++i) prm[i] = (uint32_t)((uint64_t)POSEIDON2_RC[i] * p2_rc_scale(i) % P);
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
