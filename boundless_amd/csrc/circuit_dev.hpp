// circuit_dev.hpp — the arithmetic core shared by witness generation and eval_check: the value of one derived cell,
//     sum_{t<T} prod_{f<G} pool[idx(t,f)]            (include/bx_prover.h, "The synthetic circuit")
// Both knobs are compile-time for the shapes the prover is tuned for (every pool reference is then a fixed register);
// <0, 0> is the run-time fallback for any other (T, G).
#pragma once
#include "circuit.hpp"
#include "fp.hpp"

namespace bx {

template <int TT, int GG>
__device__ __forceinline__ uint32_t cons_sum(const uint32_t (&pool)[Circuit::POOL], uint32_t T, uint32_t G) {
    uint32_t sum = 0;
    if constexpr (TT > 0) {
#pragma unroll
        for (int t = 0; t < TT; ++t) {
            uint32_t prod = pool[Circuit::pool_idx((unsigned)t, 0u)];
#pragma unroll
            for (int f = 1; f < GG; ++f) prod = fp_mul(prod, pool[Circuit::pool_idx((unsigned)t, (unsigned)f)]);
            sum = fp_add(sum, prod);
        }
    } else {
        for (uint32_t t = 0; t < T; ++t) {
            uint32_t prod = pool[Circuit::pool_idx(t, 0u)];
            for (uint32_t f = 1; f < G; ++f) prod = fp_mul(prod, pool[Circuit::pool_idx(t, f)]);
            sum = fp_add(sum, prod);
        }
    }
    return sum;
}

}  // namespace bx
