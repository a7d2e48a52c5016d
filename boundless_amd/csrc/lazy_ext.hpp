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
