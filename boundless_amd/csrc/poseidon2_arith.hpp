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
// bounds as integers (floor of the real bound, see poseidon2.hip)
constexpr uint64_t B_RED64 = 2281701410ull;    // red64_lazy output: ((2^32-1) M1 + 127 R2 + P-1)/2^32 + P  (1.13334 P)
constexpr uint64_t B_SBOX_OUT = 2122444806ull;  // sbox7_bounded output (1.05423 P), the internal rounds' S-box cell
constexpr uint64_t B_SBOX_WIDE = 4135710731ull; // sbox7_wide output (2.05423 P < 2^32); 112 * this < 2^39
constexpr uint64_t B_INT_CELL = 3789677028ull;  // internal-round cells: fixed point of B -> ((P-1) B + 2P)/2^32 + P  (1.88235 P)

BX_HD uint64_t mad64(uint32_t a, uint32_t b, uint64_t c) { return (uint64_t)a * (uint64_t)b + c; }

// x^7 for x < 1.13334 P; returns a value < 1.05423 P congruent to x^7 * 2^(-6*32) (Montgomery).  16 instructions.
BX_HD uint32_t sbox7_bounded(uint32_t x) {
    BX_ASSERT_BOUND(x <= B_RED64, "sbox input");
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
    BX_ASSERT_BOUND(x <= B_RED64, "sbox input");
    uint32_t x2 = fp_reduce(fp_mul_lazy(x, x));
    uint32_t x3 = fp_mul_lazy(x2, x);
    uint32_t x4 = fp_mul_lazy(x2, x2);
    const uint32_t x7 = fp_mul_lazy(x3, x4);
    BX_ASSERT_BOUND(x7 <= B_SBOX_WIDE, "wide sbox output");
    return x7;
}

// y (< 2^39, unreduced linear-layer output) plus a round constant -> 32 bits:  r == y + a (mod P), r < 1.13334 P, where
// `add_rr` = a * 2^64 mod P (a in the cells' Montgomery representation).
//   acc = y_lo*(2^32 mod P) + y_hi*(2^64 mod P) + add_rr < 2^32 * 268435454 + 128 * 1172168163 + P < 1.16e18,
// so acc + m*P < 2^64 and r < 268435490 + P.  4 instructions (+2 for the canonical form).
BX_HD uint32_t red64_lazy(uint64_t y, uint32_t add_rr) {
    uint64_t acc = mad64((uint32_t)y, MONT_ONE, add_rr);
    acc = mad64((uint32_t)(y >> 32), R2, acc);
    uint32_t m = (uint32_t)acc * NEG_P_INV;
    BX_ASSERT_BOUND((y >> 39) == 0, "red64 input < 2^39");
    const uint32_t r = (uint32_t)((acc + (uint64_t)m * (uint64_t)P) >> 32);
    BX_ASSERT_BOUND(r <= B_RED64, "red64 output");
    return r;
}
BX_HD uint32_t red64(uint64_t y, uint32_t add_rr) { return fp_reduce(red64_lazy(y, add_rr)); }
// the same without a constant (addend literal 0)
BX_HD uint32_t red64_lazy0(uint64_t y) {
    uint64_t acc = mad64((uint32_t)(y >> 32), R2, mad64((uint32_t)y, MONT_ONE, 0ull));
    uint32_t m = (uint32_t)acc * NEG_P_INV;
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
// prm: [0,96) | [96,117) | [117,213) round constants * 2^64 mod P, [DIAG..DIAG+24) diagonal (Montgomery).
template <int DIAG>
BX_HD void poseidon2_mix_bounded(uint32_t* s, const uint32_t* prm) {
    uint64_t y[P2_CELLS];
    m_ext64w(s, y);
    for (int i = 0; i < P2_CELLS; ++i) s[i] = red64_lazy(y[i], prm[i]);
    for (int r = 0; r < 4; ++r) {
        for (int i = 0; i < P2_CELLS; ++i) s[i] = sbox7_wide(s[i]);
        m_ext64w(s, y);
        if (r < 3) {
            for (int i = 0; i < P2_CELLS; ++i) s[i] = red64_lazy(y[i], prm[(r + 1) * P2_CELLS + i]);
        } else {
            s[0] = red64_lazy(y[0], prm[96]);
            for (int i = 1; i < P2_CELLS; ++i) s[i] = red64_lazy(y[i], 0u);
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
            for (int i = 0; i < P2_CELLS; ++i) s[i] = red64_lazy(y[i], prm[117 + (r + 1) * P2_CELLS + i]);
        } else {
            for (int i = 0; i < P2_CELLS; ++i) s[i] = red64(y[i], 0u);
        }
    }
}

}  // namespace bx
