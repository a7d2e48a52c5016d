// poseidon2_arith.hpp — the bounded-cell arithmetic of the Poseidon2 kernels, usable from host and device.
//
// The same source is compiled into the gfx950 kernels (poseidon2.hip) and into a host checker
// (tests/host_arith_check.cpp, built with g++ by tests/test_host_arith_cpu.py) that drives every helper with extreme and
// random operands, compares against exact 128-bit arithmetic mod P, and asserts the documented bounds and the absence
// of 64-bit overflow (BX_CHECK_BOUNDS).  See the bounds table in poseidon2.hip.
#pragma once
#include "fp.hpp"

#if defined(BX_CHECK_BOUNDS) && !defined(__HIP_DEVICE_COMPILE__)
#include <stdio.h>
#include <stdlib.h>
#define BX_ASSERT_BOUND(cond, what)                                              \
    do {                                                                         \
        if (!(cond)) {                                                           \
            fprintf(stderr, "bound violated: %s (%s:%d)\n", what, __FILE__, __LINE__); \
            abort();                                                             \
        }                                                                        \
    } while (0)
#else
#define BX_ASSERT_BOUND(cond, what) ((void)0)
#endif

namespace bx {

constexpr int P2_CELLS = 24;

// ---- representation tracking ------------------------------------------------------------------------------------------
// A cell holds v * c mod P for a per-round constant factor c = R^e (R = 2^32 mod P), not always the Montgomery factor R:
// the external linear layers are homogeneous, so their return to 32 bits can be a bare REDC (factor * R^-1, 2 multiply-class
// instructions) instead of a factor-preserving one (4), and the S-box maps R^e to R^(7e - 6).  With canonical Montgomery
// input (e = 1) the exponents entering the S-boxes of external rounds 0..3 are 0, -7, -56, -399; the layer after round 3
// multiplies by K1_MID = R^2801 to hand the internal rounds (which need one common factor for all cells, i.e. e = 1) the
// standard form back; rounds 4..7 see 1, 0, -7, -56 and the last layer restores e = 1 with K1_END = R^400.  Round
// constants are stored pre-scaled to the representation of the point where they are added (p2_rc_scale).
constexpr uint32_t cx_mul(uint32_t a, uint32_t b) { return (uint32_t)((uint64_t)a * b % P); }
constexpr uint32_t cx_pow(uint32_t b, uint64_t e) {
    uint32_t r = 1;
    while (e) {
        if (e & 1) r = cx_mul(r, b);
        b = cx_mul(b, b);
        e >>= 1;
    }
    return r;
}
constexpr uint32_t cx_rpow(int64_t e) { return cx_pow(MONT_ONE, e >= 0 ? (uint64_t)e : (uint64_t)((int64_t)(P - 1) + e)); }
constexpr uint32_t K1_MID = cx_rpow(2801), K2_MID = cx_mul(K1_MID, MONT_ONE);
constexpr uint32_t K1_END = cx_rpow(400), K2_END = cx_mul(K1_END, MONT_ONE);
// factor (power of R) a canonical round constant is multiplied by in the device table; index as in the 213-word layout
// 4x24 external | 21 internal | 4x24 external.  Constants added inside a REDC accumulator carry one extra R.
constexpr uint32_t p2_rc_scale(int i) {
    return i < 24 ? cx_rpow(1) : i < 48 ? cx_rpow(-6) : i < 72 ? cx_rpow(-55) : i < 96 ? cx_rpow(-398)
         : i < 141 ? cx_rpow(2) /* internal rounds and external round 4: added to Montgomery-form cells inside a REDC */
         : i < 165 ? cx_rpow(1) : i < 189 ? cx_rpow(-6) : cx_rpow(-55);
}

// bounds as integers (floor of the real bound, see poseidon2.hip)
constexpr uint64_t B_REDC = (uint64_t)P + 129;  // redc64 output: (2^39 + P) / 2^32 + P
constexpr uint64_t B_RED64 = (((uint64_t)0xffffffffu * K1_MID + 127ull * K2_MID + 2ull * P) >> 32) + P;  // red64k<MID> output (1.126 P)
constexpr uint64_t B_END = (((uint64_t)0xffffffffu * K1_END + 127ull * K2_END + 2ull * P) >> 32) + P;    // red64k<END> output (1.055 P)
constexpr uint64_t B_SBOX_OUT = 2122444806ull;  // sbox7_bounded output (1.05423 P), the internal rounds' S-box cell
constexpr uint64_t B_SBOX_WIDE = 4135710731ull; // sbox7_wide output bound for inputs < 1.13334 P (2.05423 P < 2^32); 112 * this < 2^39
constexpr uint64_t B_INT_CELL = 3789677028ull;  // internal-round cells: fixed point of B -> ((P-1) B + 2P)/2^32 + P  (1.88235 P)
static_assert(B_RED64 < 2281701410ull && B_END < 2ull * P && B_REDC < B_RED64, "representation constants moved the bounds");

BX_HD uint64_t mad64(uint32_t a, uint32_t b, uint64_t c) { return (uint64_t)a * (uint64_t)b + c; }

// x^7 for x < 1.13334 P; returns a value < 1.05423 P congruent to x^7 * 2^(-6*32) (Montgomery).  16 instructions.
BX_HD uint32_t sbox7_bounded(uint32_t x) {
    BX_ASSERT_BOUND(x <= 2281701410ull, "sbox input");
    uint32_t x2 = fp_reduce(fp_mul_lazy(x, x));
    uint32_t x3 = fp_mul_lazy(x2, x);
    uint32_t x4 = fp_mul_lazy(x2, x2);
    const uint32_t x7 = fp_mul_lazy(x3, x4);
    const uint32_t r = fp_reduce(x7);
    BX_ASSERT_BOUND(r <= B_SBOX_OUT, "sbox output");
    return r;
}

// x^7 without the final subtraction: x < 1.13334 P -> value < 2.05423 P (still a u32) congruent to x^7 * 2^(-6*32).
// The external layer that follows (m_ext64w) forms its pair sums in 64 bits instead; 14 instructions.
BX_HD uint32_t sbox7_wide(uint32_t x) {
    BX_ASSERT_BOUND(x <= 2281701410ull, "sbox input");
    uint32_t x2 = fp_reduce(fp_mul_lazy(x, x));
    uint32_t x3 = fp_mul_lazy(x2, x);
    uint32_t x4 = fp_mul_lazy(x2, x2);
    const uint32_t x7 = fp_mul_lazy(x3, x4);
    BX_ASSERT_BOUND(x7 <= B_SBOX_WIDE, "wide sbox output");
    return x7;
}

// Bare REDC of an unreduced linear-layer output plus a pre-scaled round constant:  r == (y + rc) * 2^-32 (mod P),
// r < (2^39 + P)/2^32 + P = P + 129.  3 instructions (64-bit add, v_mul_lo, v_mad_u64_u32).
BX_HD uint32_t redc64(uint64_t y, uint32_t rc) {
    BX_ASSERT_BOUND((y >> 39) == 0, "redc64 input < 2^39");
    const uint64_t acc = y + rc;
    const uint32_t m = (uint32_t)acc * NEG_P_INV;
    const uint32_t r = (uint32_t)((acc + (uint64_t)m * (uint64_t)P) >> 32);
    BX_ASSERT_BOUND(r <= B_REDC, "redc64 output");
    return r;
}
// REDC with a change of representation:  r == (y * K1 + add) * 2^-32 (mod P), K2 = K1 * 2^32 mod P.
//   acc = y_lo*K1 + y_hi*K2 + add < 2^32 * K1 + 128 * P + 2P,  acc + m*P < 2^64,  r < K1 + 61 + P.  4 instructions.
// On the device the two products are pinned to v_mad_u64_u32: left alone, hipcc reassociates y_lo*K1 + y_hi*K2 into
// y*K1 + y_hi*(K2 - K1*2^32) as a 64x64-bit multiply (4 multiply-adds and 4 moves per cell instead of 2).
BX_HD uint64_t mad64_pinned(uint32_t a, uint32_t k, uint64_t c) {
#if defined(__HIP_DEVICE_COMPILE__)
    uint64_t r;
    asm("v_mad_u64_u32 %0, vcc, %1, %2, %3" : "=v"(r) : "v"(a), "s"(k), "v"(c) : "vcc");
    return r;
#else
    return mad64(a, k, c);
#endif
}
BX_HD uint64_t mad64_pinned0(uint32_t a, uint32_t k) {
#if defined(__HIP_DEVICE_COMPILE__)
    uint64_t r;
    asm("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(r) : "v"(a), "s"(k) : "vcc");
    return r;
#else
    return mad64(a, k, 0ull);
#endif
}
template <uint32_t K1, uint32_t K2, bool HAS_ADD = true>
BX_HD uint32_t red64k_lazy(uint64_t y, uint32_t add) {
    BX_ASSERT_BOUND((y >> 39) == 0, "red64k input < 2^39");
    uint64_t acc = HAS_ADD ? mad64((uint32_t)y, K1, add) : mad64_pinned0((uint32_t)y, K1);
    acc = mad64_pinned((uint32_t)(y >> 32), K2, acc);
    const uint32_t m = (uint32_t)acc * NEG_P_INV;
    return (uint32_t)((acc + (uint64_t)m * (uint64_t)P) >> 32);
}

// external layer circ(2*M4, M4, ..., M4) on cells < 2.05423 P (sbox7_wide outputs or canonical inputs), unreduced 64-bit
// outputs:  w_k = M4 * x_k,  T = sum_k w_k,  y_k = w_k + T.   M4 by the Poseidon2 addition chain (appendix B):
//   t0 = a+b, t1 = c+d, t2 = 2b + t1, t3 = 2d + t0, t4 = 4 t1 + t3, t5 = 4 t0 + t2, w = [t3+t5, t5, t2+t4, t4].
// The pair sums do not fit 32 bits and are formed in 64; 4*t + u is one shift-add; y < 230.1 P < 2^39.
BX_HD void m_ext64w(const uint32_t* s, uint64_t* y) {
#pragma unroll
    for (int k = 0; k < P2_CELLS; k += 4) {
        const uint32_t a = s[k], b = s[k + 1], c = s[k + 2], d = s[k + 3];
        BX_ASSERT_BOUND(a <= B_SBOX_WIDE && b <= B_SBOX_WIDE && c <= B_SBOX_WIDE && d <= B_SBOX_WIDE, "m_ext64w input");
        const uint64_t t0 = (uint64_t)a + b, t1 = (uint64_t)c + d;
        const uint64_t t2 = mad64(2u, b, t1), t3 = mad64(2u, d, t0);
        const uint64_t t4 = (t1 << 2) + t3, t5 = (t0 << 2) + t2;
        y[k] = t3 + t5;
        y[k + 1] = t5;
        y[k + 2] = t2 + t4;
        y[k + 3] = t4;
    }
    uint64_t t[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        t[j] = y[j];
#pragma unroll
        for (int k = 4; k < P2_CELLS; k += 4) t[j] += y[k + j];
    }
#pragma unroll
    for (int i = 0; i < P2_CELLS; ++i) y[i] += t[i & 3];
}

// internal round helpers (host + device): sum (< 2^37) -> sum * 2^32 mod P, canonical
BX_HD uint32_t internal_sum_r(uint64_t sum) {
    BX_ASSERT_BOUND((sum >> 37) == 0, "internal sum < 2^37");
    uint64_t acc = mad64((uint32_t)sum, R2, 0ull);
    acc = mad64((uint32_t)(sum >> 32), R3, acc);
    const uint32_t m = (uint32_t)acc * NEG_P_INV;
    return fp_reduce((uint32_t)((acc + (uint64_t)m * (uint64_t)P) >> 32));
}

// The whole permutation in the exact order and arithmetic of the device kernel (poseidon2.hip: poseidon2_mix); the
// device version differs only in pinning the internal-round sum to v_mad_u64_u32 and keeping the diagonal in VGPRs.
// prm: [0,96) | [96,117) | [117,213) round constants * p2_rc_scale(i), [DIAG..DIAG+24) diagonal (Montgomery).
template <int DIAG>
BX_HD void poseidon2_mix_bounded(uint32_t* s, const uint32_t* prm) {
    uint64_t y[P2_CELLS];
    m_ext64w(s, y);
    for (int i = 0; i < P2_CELLS; ++i) s[i] = redc64(y[i], prm[i]);
    for (int r = 0; r < 4; ++r) {
        for (int i = 0; i < P2_CELLS; ++i) s[i] = sbox7_wide(s[i]);
        m_ext64w(s, y);
        if (r < 3) {
            for (int i = 0; i < P2_CELLS; ++i) s[i] = redc64(y[i], prm[(r + 1) * P2_CELLS + i]);
        } else {
            s[0] = red64k_lazy<K1_MID, K2_MID>(y[0], prm[96]);
            for (int i = 1; i < P2_CELLS; ++i) s[i] = red64k_lazy<K1_MID, K2_MID, false>(y[i], 0u);
            for (int i = 0; i < P2_CELLS; ++i) BX_ASSERT_BOUND(s[i] <= B_RED64, "mid transition output");
        }
    }
    const uint32_t* diag = prm + DIAG;
    for (int r = 0; r < 21; ++r) {
        s[0] = sbox7_bounded(s[0]);
        uint64_t sum = 0;
        for (int i = 0; i < P2_CELLS; ++i) sum += s[i];
        const uint32_t sum_r = internal_sum_r(sum);
        if (r < 20) {
            s[0] = fp_reduce(fp_mad_lazy(diag[0], s[0], sum_r + prm[97 + r]));
            for (int i = 1; i < P2_CELLS; ++i) {
                s[i] = fp_mad_lazy(diag[i], s[i], sum_r);
                BX_ASSERT_BOUND(s[i] <= B_INT_CELL, "internal cell");
            }
        } else {
            for (int i = 0; i < P2_CELLS; ++i) s[i] = fp_reduce(fp_mad_lazy(diag[i], s[i], sum_r + prm[117 + i]));
        }
    }
    for (int r = 0; r < 4; ++r) {
        for (int i = 0; i < P2_CELLS; ++i) s[i] = sbox7_wide(s[i]);
        m_ext64w(s, y);
        if (r < 3) {
            for (int i = 0; i < P2_CELLS; ++i) s[i] = redc64(y[i], prm[117 + (r + 1) * P2_CELLS + i]);
        } else {
            for (int i = 0; i < P2_CELLS; ++i) {
                const uint32_t v = red64k_lazy<K1_END, K2_END, false>(y[i], 0u);
                BX_ASSERT_BOUND(v <= B_END, "end transition output");
                s[i] = fp_reduce(v);
            }
        }
    }
}

}  // namespace bx
