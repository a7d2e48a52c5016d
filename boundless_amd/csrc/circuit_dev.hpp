// circuit_dev.hpp — the arithmetic core shared by witness generation and eval_check: the value of one derived cell,
//     sum_{t<T} prod_{f<G} pool[idx(t,f)]            (include/bx_prover.h, "The synthetic circuit")
// Both knobs are compile-time for the shapes the prover is tuned for (every pool reference is then a fixed register);
// <0, 0> is the run-time fallback for any other (T, G) and uses plain canonical arithmetic.
//
// The compile-time path runs on the signed, bounded Montgomery arithmetic of poseidon2_arith.hpp (sredc(t) = (t + mP)/2^32,
// |result| <= |t|/2^32 + P/2, valid for |t| <= SREDC_MAX = 1.209 P^2) so that no product is followed by a conditional
// subtraction and the sums stay unreduced in 64 bits:
//   pool entries are centred once per cell:                 |c| <= P/2
//   inner products of a term  x <- sredc(x * c):            |x| <= 0.617 P after the first, <= 0.65 P for any chain length
//   a group of 3 terms        acc = sum x * c_last:         |acc| <= 3 * 0.65 P * 0.5 P = 0.975 P^2 <= SREDC_MAX
//   its reduction             r = sredc(acc):               |r| <= 0.457 P + 0.5 P = 0.957 P
//   second level, 8 groups    acc2 = sum r * R  (R = 2^32 mod P = 0.1334 P, so sredc(acc2) == sum r):
//                                                           |acc2| <= 8 * 0.957 * 0.1334 P^2 = 1.02 P^2 <= SREDC_MAX,  |r2| <= 0.98 P
//   third level, <= 7 second-level results, same form:      |acc3| <= 7 * 0.98 * 0.1334 P^2 = 0.92 P^2,               |r3| <= 0.93 P
// The last reduction is < P in magnitude, so one conditional add of P makes it canonical.  Every step is a congruence mod P
// (each sredc divides by 2^32 exactly as the canonical Montgomery product does), so the result is bit-identical to the
// plain form; tests compare both paths with the oracle.  5 VALU instructions per term for G = 3 instead of 13.
#pragma once
#include "circuit.hpp"
#include "fp.hpp"
#include "poseidon2_arith.hpp"

namespace bx {
inline namespace BX_MAD_FLAVOUR {

// canonical [0, P) -> the representative in [-P/2, P/2]
BX_HD i32 fp_centre(uint32_t v) { return (i32)v - (v > P / 2 ? (i32)P : 0); }

template <int TT, int GG>
BX_HD uint32_t cons_sum(const uint32_t (&pool_u)[Circuit::POOL], uint32_t T, uint32_t G) {
    if constexpr (TT > 0) {
        static_assert(TT <= 3 * 8 * 7 && GG >= 1, "cons_sum: term count beyond the three reduction levels");
        i32 pool[Circuit::POOL];
#pragma unroll
        for (unsigned i = 0; i < Circuit::POOL; ++i) pool[i] = fp_centre(pool_u[i]);
        constexpr int GRP = 3, NG = (TT + GRP - 1) / GRP;
        i64 acc2 = 0, acc3 = 0;
        i32 result = 0;
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            const int nk = (g + 1) * GRP <= TT ? GRP : TT - g * GRP;
            // inner products of the group's terms, stage by stage (the pinned multiply-adds are opaque to the scheduler,
            // so the independent chains are interleaved here)
            i32 x[GRP];
#pragma unroll
            for (int k = 0; k < GRP; ++k) x[k] = pool[Circuit::pool_idx((unsigned)(g * GRP + (k < nk ? k : 0)), 0u)];
#pragma unroll
            for (int f = 1; f + 1 < GG; ++f) {
                i64 t[GRP];
#pragma unroll
                for (int k = 0; k < GRP; ++k) t[k] = smul(x[k], pool[Circuit::pool_idx((unsigned)(g * GRP + (k < nk ? k : 0)), (unsigned)f)], k);
                sredc_n<GRP>(t, x);
            }
            i64 acc = 0;
#pragma unroll
            for (int k = 0; k < GRP; ++k) {
                if (k < nk) {
                    if constexpr (GG == 1) acc = smad_k(x[k], MONT_ONE, acc, k);
                    else acc = smad(x[k], pool[Circuit::pool_idx((unsigned)(g * GRP + k), (unsigned)(GG - 1))], acc, k);
                }
            }
            const i32 r = sredc(acc, 3);
            acc2 = smad_k(r, MONT_ONE, acc2, g);
            if ((g & 7) == 7 || g == NG - 1) {
                const i32 r2 = sredc(acc2, 1);
                acc2 = 0;
                if constexpr (NG <= 8) result = r2;
                else acc3 = smad_k(r2, MONT_ONE, acc3, 2);
            }
        }
        if constexpr (NG > 8) result = sredc(acc3, 0);
        return (uint32_t)(result + (result < 0 ? (i32)P : 0));
    } else {
        uint32_t sum = 0;
        for (uint32_t t = 0; t < T; ++t) {
            uint32_t prod = pool_u[Circuit::pool_idx(t, 0u)];
            for (uint32_t f = 1; f < G; ++f) prod = fp_mul(prod, pool_u[Circuit::pool_idx(t, f)]);
            sum = fp_add(sum, prod);
        }
        return sum;
    }
}

}  // inline namespace BX_MAD_FLAVOUR
}  // namespace bx
