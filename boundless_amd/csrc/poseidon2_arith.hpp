// poseidon2_arith.hpp — the signed, bounded-cell arithmetic of the Poseidon2 kernels, usable from host and device.
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

// The multiply-add helpers below have two bodies (pinned instructions / plain C, chosen per translation unit by BX_PLAIN_MAD).
// Each flavour lives in its own inline namespace, so the two never share a mangled name: one definition per entity even if the
// build is ever linked with -fgpu-rdc.
#if defined(BX_PLAIN_MAD)
#define BX_MAD_FLAVOUR mad_plain
#else
#define BX_MAD_FLAVOUR mad_pinned
#endif
namespace bx {
inline namespace BX_MAD_FLAVOUR {

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

// ---- signed Montgomery arithmetic ---------------------------------------------------------------------------------------
// Inside the permutation a cell is a SIGNED 32-bit integer congruent to its value, only bounded in magnitude.  The signed
// reduction  sredc(t) = (t + m*P) / 2^32  with  m = -t * P^-1 mod 2^32 taken in [-2^31, 2^31)  returns
//   r == t * 2^-32 (mod P),   t/2^32 - P/2 <= r < t/2^32 + P/2,
// so a product of two cells of magnitude < P comes back with magnitude < 0.97 P: the set is closed under multiplication
// and NO conditional subtraction is needed anywhere inside the permutation (the unsigned form needed one per S-box plus
// one per reduced cell).  Three instructions per product as before: v_mad_i64_i32, v_mul_lo_u32, v_mad_i64_i32.
// Only the 24 words that leave the permutation are made canonical (v_add + v_min_u32 each).
// Validity: |t| <= SREDC_MAX keeps the result inside an int32 and t + m*P inside an int64.
using i32 = int32_t;
using i64 = int64_t;
constexpr i64 SREDC_MAX = ((i64)0x7fffffff - (i64)(P / 2) - 2) * ((i64)1 << 32);  // = 1.209 P^2

// The multiply-add primitives.  On the device each one is a single pinned instruction: written as plain C, hipcc widens the
// loop-carried cells to i64, forgets that they are sign-extended 32-bit values and emulates 64x64-bit products
// (v_mad_u64_u32 + 2 v_mul_lo + v_add3 per product), or inserts v_ashrrev/v_mov pairs to build sign-extended register
// pairs for its shift-add forms.  The statements are volatile so that the stage-wise source order below (independent chains
// interleaved) survives the scheduler, which otherwise re-serialises each cell's chain to save registers.
// `alt` (mod 4) picks one of four scratch SGPR pairs for the unused carry-out: hipcc separates adjacent inline-asm
// statements whose register operands overlap with an s_nop (an assumed forwarding hazard), and an s_nop after every
// multiply-add costs ~18 % at the kernel's 3 waves per SIMD (profiles/r01_microbench3_mad_forms.jsonl).  Rotating the pair
// and keeping dependent statements a stage apart removes most of them (467 -> 198 in hash_fold); the rest are spread thinly
// enough that removing them is below measurement noise.
// The host versions are the definitions; a translation unit may define BX_PLAIN_MAD to get them on the device too (circuit.hip).
#if defined(__HIP_DEVICE_COMPILE__) && !defined(BX_PLAIN_MAD)
#define BX_MAD_ASM(op, args, ...)                                                                  \
    do {                                                                                           \
        switch (alt & 3) {                                                                         \
            case 0: asm volatile(op " %0, s[40:41], " args : "=v"(r) : __VA_ARGS__ : "s40", "s41"); break; \
            case 1: asm volatile(op " %0, s[42:43], " args : "=v"(r) : __VA_ARGS__ : "s42", "s43"); break; \
            case 2: asm volatile(op " %0, s[44:45], " args : "=v"(r) : __VA_ARGS__ : "s44", "s45"); break; \
            default: asm volatile(op " %0, s[46:47], " args : "=v"(r) : __VA_ARGS__ : "s46", "s47"); break; \
        }                                                                                          \
    } while (0)
#endif
BX_HD i64 smad(i32 a, i32 b, i64 c, int alt = 0) {  // a*b + c, all signed
#if defined(__HIP_DEVICE_COMPILE__) && !defined(BX_PLAIN_MAD)
    i64 r;
    BX_MAD_ASM("v_mad_i64_i32", "%1, %2, %3", "v"(a), "v"(b), "v"(c));
    return r;
#else
    (void)alt;
    return (i64)a * (i64)b + c;
#endif
}
BX_HD i64 smul(i32 a, i32 b, int alt = 0) {  // a*b
#if defined(__HIP_DEVICE_COMPILE__) && !defined(BX_PLAIN_MAD)
    i64 r;
    BX_MAD_ASM("v_mad_i64_i32", "%1, %2, 0", "v"(a), "v"(b));
    return r;
#else
    (void)alt;
    return (i64)a * (i64)b;
#endif
}
BX_HD i64 smad_k(i32 a, uint32_t k, i64 c, int alt = 0) {  // a*k + c, k a wave-uniform constant < 2^31 (scalar operand)
#if defined(__HIP_DEVICE_COMPILE__) && !defined(BX_PLAIN_MAD)
    i64 r;
    BX_MAD_ASM("v_mad_i64_i32", "%1, %2, %3", "v"(a), "s"(k), "v"(c));
    return r;
#else
    (void)alt;
    return (i64)a * (i64)k + c;
#endif
}
BX_HD i64 smul_k(i32 a, uint32_t k, int alt = 0) {  // a*k
#if defined(__HIP_DEVICE_COMPILE__) && !defined(BX_PLAIN_MAD)
    i64 r;
    BX_MAD_ASM("v_mad_i64_i32", "%1, %2, 0", "v"(a), "s"(k));
    return r;
#else
    (void)alt;
    return (i64)a * (i64)k;
#endif
}
template <int K>
BX_HD i64 smadc(i32 a, i64 c, int alt = 0) {  // a*K + c, K an inline constant
#if defined(__HIP_DEVICE_COMPILE__) && !defined(BX_PLAIN_MAD)
    i64 r;
    BX_MAD_ASM("v_mad_i64_i32", "%1, %3, %2", "v"(a), "v"(c), "n"(K));
    return r;
#else
    (void)alt;
    return (i64)a * K + c;
#endif
}
template <int K>
BX_HD i64 smulc(i32 a, int alt = 0) {  // a*K
#if defined(__HIP_DEVICE_COMPILE__) && !defined(BX_PLAIN_MAD)
    i64 r;
    BX_MAD_ASM("v_mad_i64_i32", "%1, %2, 0", "v"(a), "n"(K));
    return r;
#else
    (void)alt;
    return (i64)a * K;
#endif
}
BX_HD i64 add_u32(i64 c, uint32_t k, int alt = 0) {  // c + k, k an unsigned wave-uniform word (round constant)
#if defined(__HIP_DEVICE_COMPILE__) && !defined(BX_PLAIN_MAD)
    i64 r;
    BX_MAD_ASM("v_mad_u64_u32", "%1, 1, %2", "s"(k), "v"(c));
    return r;
#else
    (void)alt;
    return c + (i64)k;
#endif
}
BX_HD i64 umul_k(uint32_t a, uint32_t k, int alt = 0) {  // a*k, both unsigned
#if defined(__HIP_DEVICE_COMPILE__) && !defined(BX_PLAIN_MAD)
    i64 r;
    BX_MAD_ASM("v_mad_u64_u32", "%1, %2, 0", "v"(a), "s"(k));
    return r;
#else
    (void)alt;
    return (i64)((uint64_t)a * (uint64_t)k);
#endif
}
BX_HD i32 mont_m(i64 t) {  // m = -t * P^-1 mod 2^32 as a signed word (pinned so that a stage's 24 low products stay together)
#if defined(__HIP_DEVICE_COMPILE__) && !defined(BX_PLAIN_MAD)
    i32 m;
    asm volatile("v_mul_lo_u32 %0, %1, %2" : "=v"(m) : "v"((uint32_t)t), "s"(NEG_P_INV));
    return m;
#else
    return (i32)((uint32_t)t * NEG_P_INV);
#endif
}
BX_HD i32 sredc(i64 t, int alt = 0) {
    BX_ASSERT_BOUND(t <= SREDC_MAX && t >= -SREDC_MAX, "sredc operand");
    const i32 m = mont_m(t);
    return (i32)(smad_k(m, P, t, alt) >> 32);  // exact: the low word of the sum is zero
}
BX_HD i64 iabs64(i64 v) { return v < 0 ? -v : v; }

// magnitude bounds (integers); ub(T) = T/2^32 + P/2 + 1 is the magnitude bound of sredc(t) for |t| <= T
constexpr i64 ub(i64 T) { return (T >> 32) + (i64)(P / 2) + 1; }
constexpr i64 B_IN = (i64)P - 1;                      // cells entering the permutation (canonical)
constexpr i64 B_Y = 112 * ((i64)1 << 31);             // any external-layer output: 112 * max cell magnitude (< 2^31)
constexpr i64 B_EXT = ub(B_Y + (i64)P);               // redc64s output = S-box input of the external rounds: P/2 + 58
constexpr i64 B_MIDOUT = ub(((i64)1 << 32) * K1_MID + ((B_Y >> 32) + 1) * (i64)K2_MID + 2 * (i64)P);  // cells entering the internal rounds
constexpr i64 B_SUMR = ub(((i64)1 << 31) * R2 + 33 * (i64)R3);  // sum_r: balanced low word times R2 (0.791 P)
// internal-round cells: B -> ub(P * B + B_SUMR + P) has its fixed point at 0.9412 P; 0.945 P is an invariant upper bound
constexpr i64 B_INT = (i64)((double)P * 0.945);
static_assert(ub((i64)(P - 1) * B_INT + B_SUMR + (i64)P) <= B_INT && B_MIDOUT <= B_INT, "internal-round bound is not invariant");
static_assert((i64)(P - 1) * B_INT + B_SUMR + (i64)P <= SREDC_MAX && B_INT * B_INT <= SREDC_MAX, "internal-round products overflow sredc");
static_assert(24 * B_INT < ((i64)1 << 37), "internal sum");

BX_HD uint64_t mad64(uint32_t a, uint32_t b, uint64_t c) { return (uint64_t)a * (uint64_t)b + c; }

// N independent reductions, issued stage by stage (all low products, then all corrections): the pinned primitives are
// opaque to the compiler's scheduler, so independent chains are interleaved here, in the source, to keep dependent
// multiply-adds of one cell apart.
template <int N>
BX_HD void sredc_n(const i64* t, i32* r) {
    i32 m[N];
#pragma unroll
    for (int i = 0; i < N; ++i) {
        BX_ASSERT_BOUND(t[i] <= SREDC_MAX && t[i] >= -SREDC_MAX, "sredc operand");
        m[i] = mont_m(t[i]);
    }
#pragma unroll
    for (int i = 0; i < N; ++i) r[i] = (i32)(smad_k(m[i], P, t[i], i & 3) >> 32);
}

// x^7 * 2^(-6*32) for |x| <= B_INT: four signed products, no subtraction.  |x2| < 0.92 P, |x3|, |x4| < 0.91 P, |x7| < 0.88 P.
BX_HD i32 sbox7s(i32 x) {
    BX_ASSERT_BOUND(iabs64(x) <= B_INT, "sbox input");
    const i32 x2 = sredc(smul(x, x));
    const i32 x3 = sredc(smul(x2, x));
    const i32 x4 = sredc(smul(x2, x2));
    return sredc(smul(x3, x4));
}
// the same on N cells at once, stage-wise
template <int N>
BX_HD void sbox7s_n(i32* x) {
    i64 t[N], u[N];
    i32 x2[N], x3[N], x4[N];
#pragma unroll
    for (int i = 0; i < N; ++i) {
        BX_ASSERT_BOUND(iabs64(x[i]) <= B_INT, "sbox input");
        t[i] = smul(x[i], x[i], i & 3);
    }
    sredc_n<N>(t, x2);
#pragma unroll
    for (int i = 0; i < N; ++i) {
        t[i] = smul(x2[i], x[i], 2 * i);
        u[i] = smul(x2[i], x2[i], 2 * i + 1);
    }
    sredc_n<N>(t, x3);
    sredc_n<N>(u, x4);
#pragma unroll
    for (int i = 0; i < N; ++i) t[i] = smul(x3[i], x4[i], i & 3);
    sredc_n<N>(t, x);
}

// Bare REDC of an external-layer output plus a pre-scaled round constant (canonical, < P):  r == (y + rc) * 2^-32 (mod P),
// |r| <= B_EXT.  3 instructions (64-bit add, v_mul_lo, v_mad_i64_i32).
BX_HD i32 redc64s(i64 y, uint32_t rc) {
    BX_ASSERT_BOUND(iabs64(y) <= B_Y, "external layer output");
    const i32 r = sredc(add_u32(y, rc));
    BX_ASSERT_BOUND(iabs64(r) <= B_EXT, "redc64s output");
    return r;
}

// the 24 bare reductions of a layer, stage-wise; rc = 24 pre-scaled constants
BX_HD void redc64s_all(const i64* y, const uint32_t* rc, i32* s) {
    constexpr int H = P2_CELLS / 2;  // two halves: 12 chains are enough ILP and halve the live temporaries
#pragma unroll
    for (int h = 0; h < P2_CELLS; h += H) {
        i64 acc[H];
#pragma unroll
        for (int i = 0; i < H; ++i) {
            BX_ASSERT_BOUND(iabs64(y[h + i]) <= B_Y, "external layer output");
            acc[i] = add_u32(y[h + i], rc[h + i], i & 3);
        }
        sredc_n<H>(acc, s + h);
    }
#pragma unroll
    for (int i = 0; i < P2_CELLS; ++i) BX_ASSERT_BOUND(iabs64(s[i]) <= B_EXT, "redc64s output");
}

// REDC with a change of representation:  r == (y * K1 + add) * 2^-32 (mod P),  K2 = K1 * 2^32 mod P,  y = y_hi * 2^32 + y_lo
// with y_lo unsigned, y_hi signed:  acc = y_lo*K1 + y_hi*K2 + add  in (-(|y_hi|)*P, 2^32 * K1 + ...),  |r| < K1 + P/2 + 60.
template <uint32_t K1, uint32_t K2, bool HAS_ADD = true>
BX_HD i32 red64ks(i64 y, uint32_t add) {
    BX_ASSERT_BOUND(iabs64(y) <= B_Y, "external layer output");
    static_assert(K1 < (1u << 29) && K2 < P, "correction constant too large for the accumulator bound");
    i64 acc = umul_k((uint32_t)y, K1);
    if (HAS_ADD) acc = add_u32(acc, add);
    return sredc(smad_k((i32)(y >> 32), K2, acc));
}
// the 24 representation-changing reductions of a layer, stage-wise; only cell 0 may carry an addend
template <uint32_t K1, uint32_t K2, bool HAS_ADD0>
BX_HD void red64ks_all(const i64* y, uint32_t add0, i32* s) {
    static_assert(K1 < (1u << 29) && K2 < P, "correction constant too large for the accumulator bound");
    constexpr int H = P2_CELLS / 2;
#pragma unroll
    for (int h = 0; h < P2_CELLS; h += H) {
        i64 acc[H];
#pragma unroll
        for (int i = 0; i < H; ++i) {
            BX_ASSERT_BOUND(iabs64(y[h + i]) <= B_Y, "external layer output");
            acc[i] = umul_k((uint32_t)y[h + i], K1, i & 3);
        }
        if (HAS_ADD0 && h == 0) acc[0] = add_u32(acc[0], add0, 0);
#pragma unroll
        for (int i = 0; i < H; ++i) acc[i] = smad_k((i32)(y[h + i] >> 32), K2, acc[i], (i + 1) & 3);
        sredc_n<H>(acc, s + h);
    }
}

// external layer circ(2*M4, M4, ..., M4) on signed cells, unreduced 64-bit outputs:
//   w_k = M4 * x_k,  T = sum_k w_k,  y_k = w_k + T.   M4 by the Poseidon2 addition chain (appendix B):
//   t0 = a+b, t1 = c+d, t2 = 2b + t1, t3 = 2d + t0, t4 = 4 t1 + t3, t5 = 4 t0 + t2, w = [t3+t5, t5, t2+t4, t4].
// Rows of the whole layer sum to <= 112, so |y| <= 112 * 2^31 = B_Y for any int32 cells.
BX_HD void m_ext64s(const i32* s, i64* y) {
    // The 4-cell groups advance three at a time (see sredc_n): enough independent chains to keep dependent multiply-adds
    // apart, and half the live 64-bit temporaries of advancing all six together (131 -> <= 128 VGPRs = a fourth wave per SIMD).
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        i64 t0[3], t1[3], t2[3], t3[3];
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            const int g = 3 * h + q;
            t0[q] = smulc<1>(s[4 * g], 2 * q);
            t1[q] = smulc<1>(s[4 * g + 2], 2 * q + 1);
        }
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            const int g = 3 * h + q;
            t0[q] = smadc<1>(s[4 * g + 1], t0[q], 2 * q + 2);
            t1[q] = smadc<1>(s[4 * g + 3], t1[q], 2 * q + 3);
        }
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            const int g = 3 * h + q;
            t2[q] = smadc<2>(s[4 * g + 1], t1[q], 2 * q);
            t3[q] = smadc<2>(s[4 * g + 3], t0[q], 2 * q + 1);
        }
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            const int g = 3 * h + q;
            const i64 t4 = t1[q] * 4 + t3[q], t5 = t0[q] * 4 + t2[q];
            y[4 * g] = t3[q] + t5;
            y[4 * g + 1] = t5;
            y[4 * g + 2] = t2[q] + t4;
            y[4 * g + 3] = t4;
        }
    }
    i64 t[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        t[j] = y[j];
#pragma unroll
        for (int k = 4; k < P2_CELLS; k += 4) t[j] += y[k + j];
    }
#pragma unroll
    for (int i = 0; i < P2_CELLS; ++i) y[i] += t[i & 3];
}

// internal rounds: sum of the 24 cells (|sum| < 2^37) -> sum * 2^32 mod P with |result| <= B_SUMR.  The low word is taken
// BALANCED (in [-2^31, 2^31)) so that its product with R2 stays within +-2^31 R2 and the result within 0.791 P.
BX_HD i32 internal_sum_rs(i64 sum) {
    BX_ASSERT_BOUND(iabs64(sum) < ((i64)1 << 37), "internal sum < 2^37");
    const i32 lo = (i32)(uint32_t)sum;
    const i32 hi = (i32)(sum >> 32) - (lo >> 31);  // sum = hi * 2^32 + lo with lo signed
    const i32 r = sredc(smad_k(hi, R3, smul_k(lo, R2)));
    BX_ASSERT_BOUND(iabs64(r) <= B_SUMR, "sum_r");
    return r;
}
// one internal round on all cells: s[0] <- S-box, sum, s[i] <- sredc(diag[i] * s[i] + sum_r [+ rc]).
// RC_ALL = false: only cell 0 gets the constant rc[0]; true: every cell i gets rc[i] (the last internal round).
template <bool RC_ALL>
BX_HD void internal_round(i32* s, const uint32_t* diag, const uint32_t* rc) {
    // The S-box of cell 0 is one dependent chain of 12 instructions; the sum of the other 23 cells does not depend on it and
    // is issued in between (two partial sums), so that no two adjacent statements depend on each other.  The trailing
    // argument of each primitive is its position in the instruction stream (carry-out pair rotation).
    BX_ASSERT_BOUND(iabs64(s[0]) <= B_INT, "sbox input");
    i64 pa = smulc<1>(s[1], 0), pb = smulc<1>(s[2], 1);
    i64 t = smul(s[0], s[0], 2);                                    // x^2
    pa = smadc<1>(s[3], pa, 3);
    i32 m = mont_m(t);
    pb = smadc<1>(s[4], pb, 0);
    const i32 x2 = (i32)(smad_k(m, P, t, 1) >> 32);
    pa = smadc<1>(s[5], pa, 2);
    t = smul(x2, s[0], 3);                                          // x^3
    pb = smadc<1>(s[6], pb, 0);
    i64 u = smul(x2, x2, 1);                                        // x^4
    pa = smadc<1>(s[7], pa, 2);
    m = mont_m(t);
    pb = smadc<1>(s[8], pb, 3);
    i32 m2 = mont_m(u);
    pa = smadc<1>(s[9], pa, 0);
    const i32 x3 = (i32)(smad_k(m, P, t, 1) >> 32);
    pb = smadc<1>(s[10], pb, 2);
    const i32 x4 = (i32)(smad_k(m2, P, u, 3) >> 32);
    pa = smadc<1>(s[11], pa, 0);
    t = smul(x3, x4, 1);                                            // x^7
    pb = smadc<1>(s[12], pb, 2);
    pa = smadc<1>(s[13], pa, 3);
    m = mont_m(t);
    pb = smadc<1>(s[14], pb, 0);
    pa = smadc<1>(s[15], pa, 1);
    s[0] = (i32)(smad_k(m, P, t, 2) >> 32);
#pragma unroll
    for (int i = 16; i < P2_CELLS; ++i) {
        if (i & 1) pa = smadc<1>(s[i], pa, i + 3);
        else pb = smadc<1>(s[i], pb, i + 3);
    }
    pa = smadc<1>(s[0], pa, 3);
    const i64 sum = pa + pb;
    const i64 c = smulc<1>(internal_sum_rs(sum), 1);
    constexpr int H = P2_CELLS / 2;
#pragma unroll
    for (int h = 0; h < P2_CELLS; h += H) {
        i64 tt[H];
#pragma unroll
        for (int i = 0; i < H; ++i) {
            if (RC_ALL || h + i == 0)
                tt[i] = smad_k(s[h + i], diag[h + i], add_u32(c, rc[h + i], 2 * i + 2), 2 * i + 3);  // diag < P < 2^31: a scalar operand
            else
                tt[i] = smad_k(s[h + i], diag[h + i], c, i + 2);
        }
        sredc_n<H>(tt, s + h);
    }
#pragma unroll
    for (int i = 0; i < P2_CELLS; ++i) BX_ASSERT_BOUND(iabs64(s[i]) <= B_INT, "internal cell");
}

// signed cell -> canonical word (|v| < P)
BX_HD uint32_t canon(i32 v) {
    BX_ASSERT_BOUND(iabs64(v) < (i64)P, "canonicalisation input");
    const uint32_t u = (uint32_t)v;
    return umin(u, u + P);
}

// The whole permutation in the exact order and arithmetic of the device kernel (poseidon2.hip: poseidon2_mix); the device
// version differs only in pinning the internal-round sum to v_mad_i64_i32 and keeping the diagonal in VGPRs.
// prm: [0,96) | [96,117) | [117,213) round constants * p2_rc_scale(i) (canonical), [DIAG..DIAG+24) diagonal (Montgomery).
// Input and output: canonical Montgomery words.
template <int DIAG>
BX_HD void poseidon2_mix_bounded(uint32_t* io, const uint32_t* prm) {
    i32 s[P2_CELLS];
    i64 y[P2_CELLS];
    const uint32_t* diag = prm + DIAG;
    for (int i = 0; i < P2_CELLS; ++i) {
        BX_ASSERT_BOUND(io[i] <= B_IN, "canonical input");
        s[i] = (i32)io[i];
    }
    m_ext64s(s, y);
    redc64s_all(y, prm, s);
    for (int r = 0; r < 4; ++r) {
        sbox7s_n<P2_CELLS>(s);
        m_ext64s(s, y);
        if (r < 3) {
            redc64s_all(y, prm + (r + 1) * P2_CELLS, s);
        } else {
            red64ks_all<K1_MID, K2_MID, true>(y, prm[96], s);
            for (int i = 0; i < P2_CELLS; ++i) BX_ASSERT_BOUND(iabs64(s[i]) <= B_MIDOUT, "mid transition output");
        }
    }
    for (int r = 0; r < 20; ++r) internal_round<false>(s, diag, prm + 97 + r);
    internal_round<true>(s, diag, prm + 117);
    for (int r = 0; r < 4; ++r) {
        sbox7s_n<P2_CELLS>(s);
        m_ext64s(s, y);
        if (r < 3) {
            redc64s_all(y, prm + 117 + (r + 1) * P2_CELLS, s);
        } else {
            red64ks_all<K1_END, K2_END, false>(y, 0u, s);
            for (int i = 0; i < P2_CELLS; ++i) io[i] = canon(s[i]);
        }
    }
}

}  // inline namespace BX_MAD_FLAVOUR
}  // namespace bx
