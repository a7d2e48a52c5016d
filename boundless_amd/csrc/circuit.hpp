// circuit.hpp — shape of the synthetic circuit that stands in for risc0-circuit-rv32im (include/bx_prover.h, "The
// synthetic circuit", is the normative text): which columns are free / derived / accumulators, which have a tap one row
// back, which pool entry is factor f of term t.  Shared by the prover (prover.hip, circuit.hip) and the host verifier
// (verify.cpp).  The test oracle under oracle/ restates the same rules independently.
#pragma once
#include <stddef.h>
#include <stdint.h>

#include "../../include/bx_prover.h"

#if defined(__HIPCC__)
#define BX_CIRC_HD __host__ __device__
#else
#define BX_CIRC_HD
#endif

namespace bx {

// pseudo-random words of the synthetic witness (bx_prover.h, "seeds")
BX_CIRC_HD inline uint64_t splitmix64(uint64_t x) {
    uint64_t z = x + 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
BX_CIRC_HD inline uint32_t synth_word(uint64_t seed, uint32_t col, uint32_t row) {
    const uint32_t v = (uint32_t)(splitmix64(seed ^ (((uint64_t)col << 32) | row)) >> 33);
    return v >= 2013265921u ? v - 2013265921u : v;
}
// the code group's generator: a constant, so that the code group — and with it the control ID — depends on the shape alone
constexpr uint64_t SYNTH_CODE_SEED = 0x434F4E54524F4C21ull;  // "CONTROL!"

struct Circuit {
    // pool of a derived column: [0] free column j, [1] the same one or two rows back (slot1_back), [2] [3] free columns
    // j+1, j+2 (mod F), [4..11] the eight previous derived columns, [12..15] code columns csel(j..j+3)
    static constexpr unsigned POOL = 16;
    struct Src {
        int group;      // 0 code, 1 data, -1 = the constant one
        uint32_t col;
        int back;       // rows back (0 = this row)
    };
    uint32_t po2, wc, wd, wa, T, G;
    uint32_t F, J, E, pairs;
    Circuit() = default;
    Circuit(uint32_t po2_, uint32_t w_code, uint32_t w_data, uint32_t w_accum, uint32_t terms, uint32_t degree)
        : po2(po2_), wc(w_code), wd(w_data), wa(w_accum), T(terms ? terms : BX_CIRCUIT_DEFAULT_TERMS),
          G(degree ? degree : BX_CIRCUIT_DEFAULT_DEGREE) {
        F = (wd + 1) / 2;  // free data columns [0, F); derived [F, wd)
        J = wd - F;
        E = wa / 4;  // ext accumulators; accum columns [4E, wa) are noise
        // the largest count with 2(pairs-1)+1 < E and 4(pairs-1)+3 < F, in closed form (this constructor runs once per column in
        // the verifier's tap loop: a loop here let a hostile header buy seconds of CPU before the seal length was looked at)
        pairs = (E >= 2 && F >= 4) ? (((E - 2) / 2 + 1 < (F - 4) / 4 + 1) ? (E - 2) / 2 + 1 : (F - 4) / 4 + 1) : 0;
        if (wc < 2) pairs = 0;  // the closing constraint needs the `last` selector (code column 1)
    }
    // public words of the statement: the first cell of data column 0 ("where the segment starts") and, with a `last` selector,
    // the last cell of the last data column ("what it computed"); each is tied to the trace by a boundary constraint
    BX_CIRC_HD uint32_t globals() const { return wc >= 2 ? 2u : 1u; }
    size_t constraints() const { return (size_t)J + E + pairs + globals(); }
    BX_CIRC_HD static constexpr unsigned pool_idx(unsigned t, unsigned f) { return (7 * t + 3 * f + (t >> 2) * f + (t >> 4)) & 15u; }
    // where pool entry `slot` of derived column j comes from
    BX_CIRC_HD Src pool_src(uint32_t j, unsigned slot) const {
        if (slot == 0) return Src{1, j, 0};
        if (slot == 1) return Src{1, j, slot1_back(j)};
        if (slot <= 3) return Src{1, (j + slot - 1) % F, 0};
        if (slot <= 11) {
            const uint32_t s = slot - 3;  // 1..8
            if (j >= s) return Src{1, F + j - s, 0};
            const int c = csel_col(s - j - 1);
            return c < 0 ? Src{-1, 0, 0} : Src{0, (uint32_t)c, 0};
        }
        const int c = csel_col(j + slot - 12);
        return c < 0 ? Src{-1, 0, 0} : Src{0, (uint32_t)c, 0};
    }
    // code column behind csel(i), or -1 for the constant one (fewer than three code columns)
    BX_CIRC_HD int csel_col(unsigned i) const { return wc >= 3 ? (int)(2 + i % (wc - 2)) : -1; }
    // data column accumulator e runs over
    BX_CIRC_HD uint32_t acc_src(uint32_t e) const {
        const uint32_t p = e / 2;
        if (p < pairs) return (e & 1) ? 4 * p + 3 : 4 * p + 2;
        return e % F;
    }
    // ZK noise rows: the last zk_rows() rows of the trace hold noise in every free data column (drawn from the segment's
    // noise seed), the derived columns and the accumulators simply continue over them (their constraints hold on every row), and
    // `last`, the public word g_1 and the permuted copies refer to the active rows [0, active_rows()) only.
    // risc0_zkp::ZK_CYCLES = 1994 [EXT]; capped at N/4 for the small sizes the tests use.
    BX_CIRC_HD uint32_t zk_rows() const { return ((1u << po2) >> 2) < 1994u ? ((1u << po2) >> 2) : 1994u; }
    BX_CIRC_HD uint32_t active_rows() const { return (1u << po2) - zk_rows(); }
    // row permutation of pair p over the active rows: data[4p+3][perm(r)] = data[4p+2][r], r < active_rows()
    // (a bijection: the multiplier is a prime larger than any row count)
    BX_CIRC_HD uint32_t perm_row(uint32_t p, uint32_t r) const {
        return (uint32_t)(((uint64_t)r * 2654435761ull + 12345u + p) % active_rows());
    }
    // tap set of column c of group g (0 code, 1 data, 2 accum): the rows back it is opened at; returns their count.
    // data: c % 8 == 0 -> {0,1}, c % 8 == 4 -> {0,1,2}; accumulator columns {0,1}; everything else {0}
    BX_CIRC_HD uint32_t backs_of(int g, uint32_t c, uint32_t* out) const {
        out[0] = 0;
        if (g == 1 && c % 8 == 0) { out[1] = 1; return 2; }
        if (g == 1 && c % 8 == 4) { out[1] = 1; out[2] = 2; return 3; }
        if (g == 2 && c < 4 * E) { out[1] = 1; return 2; }
        return 1;
    }
    // the row offset pool slot 1 of derived column j reads free column j at (0 = this row: the slot repeats slot 0)
    BX_CIRC_HD static constexpr int slot1_back(uint32_t j) { return j % 8 == 0 ? 1 : (j % 8 == 4 ? 2 : 0); }
};

// The host-only entries of the synthetic circuit's bx_circuit_ops table (shape handling; no device work): shared by the
// library's table (circuit.hip) and by host-only builds of the verifier (tests/verify_fuzz_check.cpp).
inline Circuit circuit_of(const bx_segment_params* s) { return Circuit(s->po2, s->w_code, s->w_data, s->w_accum, s->cons_terms, s->cons_degree); }
inline const char* synth_normalize(void*, bx_segment_params* s) {
    if (!s) return "circuit: null shape";
    if (s->cons_terms > BX_CIRCUIT_MAX_TERMS || s->cons_degree > BX_CIRCUIT_MAX_DEGREE)
        return "synthetic circuit: cons_terms must be <= 64 and cons_degree <= 5 (0 = default)";
    if (!s->cons_terms) s->cons_terms = BX_CIRCUIT_DEFAULT_TERMS;
    if (!s->cons_degree) s->cons_degree = BX_CIRCUIT_DEFAULT_DEGREE;
    return nullptr;
}
inline uint32_t synth_taps(void*, const bx_segment_params* s, int group, uint32_t col, uint32_t* backs_out /* BX_MAX_TAPS */) {
    return circuit_of(s).backs_of(group, col, backs_out);
}
inline uint32_t synth_n_globals(void*, const bx_segment_params* s) { return circuit_of(s).globals(); }
// verifier side of the code-group binding (control_id.cpp): the built-in table, else the cached host computation
const char* synth_check_code(void*, const bx_segment_params* s, const uint32_t root[8]);
// cell (col, row) of the synthetic circuit's code group (bx_prover.h, "code")
BX_CIRC_HD inline uint32_t synth_code_cell(const Circuit& cc, uint32_t col, uint32_t row) {
    return col == 0 ? (row == 0 ? 268435454u : 0u) : col == 1 ? (row == cc.active_rows() - 1 ? 268435454u : 0u) : synth_word(SYNTH_CODE_SEED, col, row);
}

}  // namespace bx
