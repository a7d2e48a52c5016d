// poseidon2_params.hpp — default parameters of the Poseidon2 BabyBear t=24 instance used by the reference's
// `poseidon2` hash suite (risc0-zkp 3.0.3 core/hash/poseidon2/consts.rs: ROUND_CONSTANTS, M_INT_DIAG_HZN; not
// vendored in the reference tree, Cargo.lock:9155).
//
// The 213 round constants are not typed in: they are regenerated at ctx init with the generator the Poseidon /
// Poseidon2 papers specify (Grain LFSR, self-shrinking mode; parameters field=1, sbox=0 (x^alpha), n=31, t=24,
// R_F=8, R_P=21; Poseidon2 draws t*R_F + R_P values).  The internal-layer diagonal is the published 24-word table
// of that instance (it was chosen at random by its authors and cannot be derived).  Both are validated end to
// end by the published known-answer test: permutation(0..23) — see tests/golden/poseidon2_kat.json.
#pragma once
#include <stdint.h>
#include <string.h>

namespace bx {

inline void poseidon2_grain_constants(uint32_t* out, int count) {
    unsigned char bits[80];
    int pos = 0;
    const unsigned vals[6] = {1, 0, 31, 24, 8, 21}, widths[6] = {2, 4, 12, 12, 10, 10};
    for (int f = 0; f < 6; ++f)
        for (int b = (int)widths[f] - 1; b >= 0; --b) bits[pos++] = (unsigned char)((vals[f] >> b) & 1u);
    while (pos < 80) bits[pos++] = 1;
    auto step = [&]() -> unsigned {
        unsigned nb = bits[62] ^ bits[51] ^ bits[38] ^ bits[23] ^ bits[13] ^ bits[0];
        memmove(bits, bits + 1, 79);
        bits[79] = (unsigned char)nb;
        return nb;
    };
    for (int i = 0; i < 160; ++i) step();
    auto next_bit = [&]() -> unsigned {
        for (;;) {
            if (step()) return step();
            step();
        }
    };
    for (int k = 0; k < count;) {
        uint32_t v = 0;
        for (int i = 0; i < 31; ++i) v = (v << 1) | next_bit();
        if (v < 2013265921u) out[k++] = v;
    }
}

struct Poseidon2Defaults {
    uint32_t rc[213];
    Poseidon2Defaults() { poseidon2_grain_constants(rc, 213); }
};
inline const uint32_t* poseidon2_default_rc() {
    static Poseidon2Defaults d;
    return d.rc;
}
#define POSEIDON2_RC (bx::poseidon2_default_rc())

static const uint32_t POSEIDON2_DIAG[24] = {
    0x409133f0u, 0x1667a8a1u, 0x06a6c7b6u, 0x6f53160eu, 0x273b11d1u, 0x03176c5du, 0x72f9bbf9u, 0x73ceba91u,
    0x5cdef81du, 0x01393285u, 0x46daee06u, 0x065d7ba6u, 0x52d72d6fu, 0x05dd05e0u, 0x3bab4b63u, 0x6ada3842u,
    0x2fc5fbecu, 0x770d61b0u, 0x5715aae9u, 0x03ef0e90u, 0x75b6c770u, 0x242adf5fu, 0x00d0ca4cu, 0x36c0e388u};

}  // namespace bx
