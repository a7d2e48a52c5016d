// lazy_ext.hpp — sum_k w_k * x_k with w_k in Fp4 (a wave-uniform weight: a mix power, a power of the evaluation point) and
// x_k in Fp (a coefficient, a constraint value), without a reduction per product.
//
// The canonical form (f4_scale + f4_add) costs 4 x (5 + 3) = 32 VALU instructions per term; mix_poly_coeffs,
// batch_evaluate_any and the mixing of eval_check are nothing else, which made these "streaming" entry points VALU-bound
// (mix_poly_coeffs 3.0 TB/s, batch_evaluate_any 2.5 TB/s in round 1).  Here, on the signed arithmetic of poseidon2_arith.hpp:
//   weights are centred once where their table is built (|w| <= P/2), x stays canonical (0 <= x < P);
//   two raw products per component accumulate in 64 bits:      |a1| <= 2 * (P/2) * P = P^2      <= SREDC_MAX (1.209 P^2)
//   first reduction  r = sredc(a1):                            |r|  <= 0.47 P + 0.5 P = 0.97 P
//   eight of them accumulate as r * R (R = 2^32 mod P = 0.1334 P, so the second reduction returns their sum):
//                                                              |a2| <= 8 * 0.97 * 0.1334 P^2 = 1.04 P^2 <= SREDC_MAX
//   second reduction r2 = sredc(a2):  |r2| <= 0.99 P < P  ->  one conditional add makes it canonical, one modular add per
//   16 terms folds it into the running canonical sum.
// Every step is a congruence mod P with the same power of 2^-32 as fp_mul, so the result equals the canonical form bit for
// bit.  4 multiply-adds + 12/2 + 16/16 ... = 10.3 instructions per term instead of 32.
#pragma once
#include "fp.hpp"
#include "poseidon2_arith.hpp"

namespace bx {
inline namespace BX_MAD_FLAVOUR {

BX_HD i32 fp_centre_w(uint32_t v) { return (i32)v - (v > P / 2 ? (i32)P : 0); }  // canonical -> [-P/2, P/2]

// ---- ext x ext product on the signed lazy arithmetic: 40 multiply-class instructions instead of the 69 (+62 cheap ones) of
// fp.hpp's f4_mul, which reduces and conditionally subtracts after every one of its 19 base products.  Operands are CENTRED
// (|.| <= P/2), so four raw products fit one 64-bit accumulator (4 (P/2)^2 = P^2 <= SREDC_MAX = 1.2 P^2): every output
// component is one accumulation of its <= 4 products plus NBETA times the (once reduced) wrapped-around part, and one
// reduction.  Bounds (rho = P/2^32 = 0.469): u = sredc(<= 3 products) has |u| <= 0.75 rho P + P/2 = 0.85 P, |NB| = 0.467 P, so the
// largest accumulator is component 2's: 3 (P/2)^2 + 0.467 P * 0.617 P = 1.04 P^2.  Result canonical; congruent to f4_mul.
struct C4 {
    i32 c[4];
};
BX_HD C4 f4_centre(const Fp4& a) { return C4{{fp_centre_w(a.c[0]), fp_centre_w(a.c[1]), fp_centre_w(a.c[2]), fp_centre_w(a.c[3])}}; }
BX_HD Fp4 f4_mul_cc(const C4& a, const C4& b) {
    constexpr i32 NB = (i32)MONT_NBETA - (i32)P;  // Montgomery form of -11, centred
    const i64 t3 = smad(a.c[0], b.c[3], smad(a.c[1], b.c[2], smad(a.c[2], b.c[1], smul(a.c[3], b.c[0]))));
    const i32 u2 = sredc(smul(a.c[3], b.c[3]));
    const i64 t2 = smad(NB, u2, smad(a.c[0], b.c[2], smad(a.c[1], b.c[1], smul(a.c[2], b.c[0]))));
    const i32 u1 = sredc(smad(a.c[2], b.c[3], smul(a.c[3], b.c[2])));
    const i64 t1 = smad(NB, u1, smad(a.c[0], b.c[1], smul(a.c[1], b.c[0])));
    const i32 u0 = sredc(smad(a.c[1], b.c[3], smad(a.c[2], b.c[2], smul(a.c[3], b.c[1]))));
    const i64 t0 = smad(NB, u0, smul(a.c[0], b.c[0]));
    return Fp4{{canon(sredc(t0)), canon(sredc(t1)), canon(sredc(t2)), canon(sredc(t3))}};
}
BX_HD Fp4 f4_mul_lz(const Fp4& a, const Fp4& b) { return f4_mul_cc(f4_centre(a), f4_centre(b)); }  // canonical in, canonical out

struct LazyExtAcc {
    i64 a1[4], a2[4];
    Fp4 sum;
    int n1, n2;
    BX_HD void reset() {
        for (int c = 0; c < 4; ++c) a1[c] = a2[c] = 0;
        sum = f4_zero();
        n1 = n2 = 0;
    }
    BX_HD void fold2() {
        for (int c = 0; c < 4; ++c) {
            sum.c[c] = fp_add(sum.c[c], canon(sredc(a2[c])));
            a2[c] = 0;
        }
        n2 = 0;
    }
    BX_HD void fold1() {
        for (int c = 0; c < 4; ++c) {
            a2[c] = smad_k(sredc(a1[c]), MONT_ONE, a2[c]);
            a1[c] = 0;
        }
        n1 = 0;
        if (++n2 == 8) fold2();
    }
    // w and x both centred (|.| <= P/2): four raw products fit one first-level accumulator (4 (P/2)^2 = P^2 <= SREDC_MAX = 1.2 P^2),
    // so the first-level reduction runs once per four terms instead of two; do not mix with add() between folds
    BX_HD void add_centred(const i32 w[4], i32 xc) {
        for (int c = 0; c < 4; ++c) a1[c] = smad(w[c], xc, a1[c]);
        if (++n1 == 4) fold1();
    }
    // w: centred weight (|w[c]| <= P/2), x: canonical
    BX_HD void add(const i32 w[4], uint32_t x) {
        for (int c = 0; c < 4; ++c) a1[c] = smad(w[c], (i32)x, a1[c]);
        if (++n1 == 2) fold1();
    }
    BX_HD Fp4 finish() {
        if (n1) fold1();
        if (n2) fold2();
        return sum;
    }
};

}  // inline namespace BX_MAD_FLAVOUR
}  // namespace bx
