// fp.hpp — BabyBear field (Montgomery u32) and its quartic extension, usable from host and gfx950 device code.
//
// Restates risc0_core::field::baby_bear::{Elem, ExtElem} (risc0-core 3.0.0, pinned by the reference's
// Cargo.lock:9012; reached from bento/crates/workflow/src/tasks/prove.rs:41-49).  Every value that crosses the
// HAL boundary is the Montgomery word itself, so results are bit-identical as long as each op returns the
// canonical representative in [0, P).
#pragma once
#include <stdint.h>
#if defined(BX_CHECK_BOUNDS)
#include <stdio.h>
#include <stdlib.h>
#endif

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define BX_HD __host__ __device__ __forceinline__
#else
#define BX_HD inline
#endif

namespace bx {

constexpr uint32_t P = 2013265921u;      // 15 * 2^27 + 1
constexpr uint32_t P_INV = 0x88000001u;  // P^-1 mod 2^32  (= 2^31 + 2^27 + 1)
constexpr uint32_t R2 = 1172168163u;     // 2^64 mod P
constexpr uint32_t MONT_ONE = 268435454u;  // 2^32 mod P
constexpr uint32_t R3 = 317946875u;      // 2^96 mod P

BX_HD uint32_t umin(uint32_t a, uint32_t b) { return a < b ? a : b; }

// a + b mod P for canonical inputs: s < 2P < 2^32; s - P wraps above s exactly when s < P.
BX_HD uint32_t fp_add(uint32_t a, uint32_t b) {
    uint32_t s = a + b;
    return umin(s, s - P);
}
// a - b mod P: d wraps above P exactly when a < b, and then d + P wraps back into [0, P).
BX_HD uint32_t fp_sub(uint32_t a, uint32_t b) {
    uint32_t d = a - b;
    return umin(d, d + P);
}
BX_HD uint32_t fp_neg(uint32_t a) { return fp_sub(0u, a); }
BX_HD uint32_t fp_dbl(uint32_t a) { return fp_add(a, a); }

constexpr uint32_t NEG_P_INV = 0x77FFFFFFu;  // -P^-1 mod 2^32

// Lazy Montgomery multiply-add: r == (a*b + c) * 2^-32 (mod P) with r < (a*b + c)/2^32 + P, NOT reduced.
// With m = -(a*b + c) * P^-1 mod 2^32 the sum a*b + c + m*P is divisible by 2^32.  Three VALU instructions on gfx950
// (v_mad_u64_u32, v_mul_lo_u32, v_mad_u64_u32).  Contract: a*b + c + (2^32 - 1)*P < 2^64, i.e. a*b + c < 2.42 * P^2.
BX_HD uint32_t fp_mad_lazy(uint32_t a, uint32_t b, uint32_t c) {
    uint64_t ab = (uint64_t)a * (uint64_t)b + c;
    uint32_t m = (uint32_t)ab * NEG_P_INV;
#if defined(BX_CHECK_BOUNDS) && !defined(__HIP_DEVICE_COMPILE__)
    if ((((unsigned __int128)a * b + c + (unsigned __int128)m * P) >> 64) != 0) {
        fprintf(stderr, "fp_mad_lazy: 64-bit overflow (a=%u b=%u c=%u)\n", a, b, c);
        abort();
    }
#endif
    return (uint32_t)((ab + (uint64_t)m * (uint64_t)P) >> 32);
}
BX_HD uint32_t fp_mul_lazy(uint32_t a, uint32_t b) { return fp_mad_lazy(a, b, 0u); }
// x in [0, 2P) -> canonical
BX_HD uint32_t fp_reduce(uint32_t x) { return umin(x, x - P); }

// Fused-reduction radix-2 butterfly (the round-5 experiment now kept as tools/experiments/r05_ntt_fused_reduction.patch; the host
// checker tests/host_arith_check.cpp still exercises the arithmetic): u = REDC(a R + b w), d = REDC(a R + b wn) with R = MONT_ONE =
// 2^32 mod P and wn = P - w, i.e. u == a + b w 2^-32 and d == a - b w 2^-32 (mod P) — the Montgomery butterfly with its addition and
// subtraction inside the reductions.  a, b may be ANY u32 and w, wn <= P - 1: a R + b w + m P <= (2^32 - 1)(R + 2P - 1) =
// (2^32 - 1)^2 < 2^64 because R + 2P = 2^32 exactly; the results are < 2^32 (not < 2P): see reduce_any().
BX_HD void bfly_fused(uint32_t& a, uint32_t& b, uint32_t w, uint32_t wn) {
    const uint64_t t0 = (uint64_t)a * MONT_ONE;
    const uint64_t tu = t0 + (uint64_t)b * w, td = t0 + (uint64_t)b * wn;
    const uint32_t mu = (uint32_t)tu * NEG_P_INV, md = (uint32_t)td * NEG_P_INV;
#if defined(BX_CHECK_BOUNDS) && !defined(__HIP_DEVICE_COMPILE__)
    if ((((unsigned __int128)a * MONT_ONE + (unsigned __int128)b * w + (unsigned __int128)mu * P) >> 64) != 0 ||
        (((unsigned __int128)a * MONT_ONE + (unsigned __int128)b * wn + (unsigned __int128)md * P) >> 64) != 0) {
        fprintf(stderr, "bfly_fused: 64-bit overflow (a=%u b=%u w=%u wn=%u)\n", a, b, w, wn);
        abort();
    }
#endif
    a = (uint32_t)((tu + (uint64_t)mu * P) >> 32);
    b = (uint32_t)((td + (uint64_t)md * P) >> 32);
}
// any u32 -> canonical: x < 2^32 = 2P + R with R < P, so at most one subtraction of 2P and one of P
BX_HD uint32_t reduce_any(uint32_t x) {
    x = umin(x, x - 2u * P);
    return umin(x, x - P);
}

// Montgomery product a*b*2^-32 mod P, canonical, for canonical inputs: the lazy result is < P^2/2^32 + P < 1.47 P,
// so one conditional subtraction finishes it (5 instructions; every 32-bit integer VALU op issues at the same rate on
// gfx950, see profiles/r01_microbench_valu.jsonl, so instruction count is what matters).
BX_HD uint32_t fp_mul(uint32_t a, uint32_t b) { return fp_reduce(fp_mul_lazy(a, b)); }
BX_HD uint32_t fp_sqr(uint32_t a) { return fp_mul(a, a); }
BX_HD uint32_t fp_encode(uint32_t canonical) { return fp_mul(R2, canonical % P); }
BX_HD uint32_t fp_decode(uint32_t mont) { return fp_mul(1u, mont); }
BX_HD uint32_t fp_pow(uint32_t a, uint64_t e) {
    uint32_t r = MONT_ONE;
    while (e) {
        if (e & 1) r = fp_mul(r, a);
        a = fp_mul(a, a);
        e >>= 1;
    }
    return r;
}
BX_HD uint32_t fp_inv(uint32_t a) { return fp_pow(a, (uint64_t)P - 2); }

// Montgomery forms of a few small constants (checked against fp_encode at ctx init, hal.hip).
constexpr uint32_t MONT_NBETA = 1073741848u;  // encode(P - 11)
constexpr uint32_t MONT_BETA = 939524073u;    // encode(11)
constexpr uint32_t MONT_THREE = 805306362u;   // encode(3)

// Fp4 = Fp[X]/(X^4 + 11), AoS.
struct Fp4 {
    uint32_t c[4];
};
BX_HD Fp4 f4_zero() { return Fp4{{0u, 0u, 0u, 0u}}; }
BX_HD Fp4 f4_one() { return Fp4{{MONT_ONE, 0u, 0u, 0u}}; }
BX_HD Fp4 f4_add(const Fp4& a, const Fp4& b) {
    return Fp4{{fp_add(a.c[0], b.c[0]), fp_add(a.c[1], b.c[1]), fp_add(a.c[2], b.c[2]), fp_add(a.c[3], b.c[3])}};
}
BX_HD Fp4 f4_sub(const Fp4& a, const Fp4& b) {
    return Fp4{{fp_sub(a.c[0], b.c[0]), fp_sub(a.c[1], b.c[1]), fp_sub(a.c[2], b.c[2]), fp_sub(a.c[3], b.c[3])}};
}
BX_HD Fp4 f4_scale(const Fp4& a, uint32_t s) {
    return Fp4{{fp_mul(a.c[0], s), fp_mul(a.c[1], s), fp_mul(a.c[2], s), fp_mul(a.c[3], s)}};
}
// [EXT] ExtElem::mul: X^4 = -11 (NBETA = P - 11 in Montgomery form passed as a constant).
BX_HD Fp4 f4_mul(const Fp4& a, const Fp4& b) {
    const uint32_t nb = MONT_NBETA;
    Fp4 r;
    r.c[0] = fp_add(fp_mul(a.c[0], b.c[0]),
                    fp_mul(nb, fp_add(fp_add(fp_mul(a.c[1], b.c[3]), fp_mul(a.c[2], b.c[2])), fp_mul(a.c[3], b.c[1]))));
    r.c[1] = fp_add(fp_add(fp_mul(a.c[0], b.c[1]), fp_mul(a.c[1], b.c[0])),
                    fp_mul(nb, fp_add(fp_mul(a.c[2], b.c[3]), fp_mul(a.c[3], b.c[2]))));
    r.c[2] = fp_add(fp_add(fp_add(fp_mul(a.c[0], b.c[2]), fp_mul(a.c[1], b.c[1])), fp_mul(a.c[2], b.c[0])),
                    fp_mul(nb, fp_mul(a.c[3], b.c[3])));
    r.c[3] = fp_add(fp_add(fp_add(fp_mul(a.c[0], b.c[3]), fp_mul(a.c[1], b.c[2])), fp_mul(a.c[2], b.c[1])),
                    fp_mul(a.c[3], b.c[0]));
    return r;
}
BX_HD Fp4 f4_pow(Fp4 a, uint64_t e) {
    Fp4 r = f4_one();
    while (e) {
        if (e & 1) r = f4_mul(r, a);
        a = f4_mul(a, a);
        e >>= 1;
    }
    return r;
}
BX_HD bool f4_is_zero(const Fp4& a) { return (a.c[0] | a.c[1] | a.c[2] | a.c[3]) == 0u; }
// inverse through the norm to Fp[X^2]/(X^4+11): a(X)a(-X) = b0 + b2 X^2, (b0 + b2 Y)(b0 - b2 Y) = b0^2 + 11 b2^2.
BX_HD Fp4 f4_inv(const Fp4& a) {
    const uint32_t beta = MONT_BETA;
    uint32_t b0 = fp_add(fp_sqr(a.c[0]), fp_mul(beta, fp_sub(fp_mul(fp_dbl(a.c[1]), a.c[3]), fp_sqr(a.c[2]))));
    uint32_t b2 = fp_add(fp_sub(fp_mul(fp_dbl(a.c[0]), a.c[2]), fp_sqr(a.c[1])), fp_mul(beta, fp_sqr(a.c[3])));
    uint32_t ic = fp_inv(fp_add(fp_sqr(b0), fp_mul(beta, fp_sqr(b2))));
    Fp4 an{{a.c[0], fp_neg(a.c[1]), a.c[2], fp_neg(a.c[3])}};
    Fp4 d{{fp_mul(b0, ic), 0u, fp_neg(fp_mul(b2, ic)), 0u}};
    return f4_mul(an, d);
}

BX_HD uint32_t bit_reverse32(uint32_t v) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __brev(v);
#else
    v = ((v >> 1) & 0x55555555u) | ((v & 0x55555555u) << 1);
    v = ((v >> 2) & 0x33333333u) | ((v & 0x33333333u) << 2);
    v = ((v >> 4) & 0x0F0F0F0Fu) | ((v & 0x0F0F0F0Fu) << 4);
    v = ((v >> 8) & 0x00FF00FFu) | ((v & 0x00FF00FFu) << 8);
    return (v >> 16) | (v << 16);
#endif
}
BX_HD uint32_t bit_reverse(uint32_t v, unsigned bits) { return bits ? bit_reverse32(v) >> (32u - bits) : 0u; }

}  // namespace bx
