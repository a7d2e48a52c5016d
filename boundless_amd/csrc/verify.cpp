// verify.cpp — host-side verifier of the segment seal (include/bx_prover.h: bx_verify_segment).
//
// Counterpart of `SegmentReceipt::verify_integrity_with_context` (bento/crates/workflow/src/tasks/prove.rs:53-55) for the
// circuit-independent pipeline of bx_prove_segment.  Structure follows risc0_zkp::verify::{verify, fri::fri_verify,
// merkle::MerkleTreeVerifier} (risc0-zkp 3.0.3, reference Cargo.lock:9155): replay the transcript, check the constraint
// identity at the random point Z (here: the stand-in check polynomial, see bx_prover.h), then for each query verify the
// Merkle openings, recompute the DEEP quotient from the opened trace rows and follow the FRI folds to the final polynomial.
// Pure CPU code (the reference verifies on the CPU as well); it shares only fp.hpp/transcript.hpp with the prover.
#include <stdio.h>
#include <string.h>

#include <string>
#include <vector>

#include "../../include/bx_prover.h"
#include "fp.hpp"
#include "poseidon2_params.hpp"
#include "transcript.hpp"

namespace {
using namespace bx;

struct Fail {
    std::string msg;
};
#define VCHECK(cond, text)           \
    do {                             \
        if (!(cond)) throw Fail{text}; \
    } while (0)

struct Reader {
    const uint32_t* p;
    size_t n, pos = 0;
    const uint32_t* take(size_t k) {
        if (pos + k > n) throw Fail{"seal truncated"};
        const uint32_t* r = p + pos;
        pos += k;
        return r;
    }
};

unsigned ilog2u(size_t n) {
    unsigned k = 0;
    while (((size_t)1 << k) < n) k++;
    return k;
}
unsigned top_layer_of(unsigned layers) {
    unsigned top = 0;
    for (unsigned i = 1; i < layers; ++i) {
        if ((1u << i) > BX_QUERIES) break;
        top = i;
    }
    return top;
}
Fp4 ld(const uint32_t* w) { return Fp4{{w[0], w[1], w[2], w[3]}}; }
bool eq(const Fp4& a, const Fp4& b) { return a.c[0] == b.c[0] && a.c[1] == b.c[1] && a.c[2] == b.c[2] && a.c[3] == b.c[3]; }
Fp4 from_base(uint32_t v) { return Fp4{{v, 0, 0, 0}}; }
uint32_t rou(unsigned k) { return fp_pow(fp_encode(137u), (uint64_t)1 << (27 - k)); }  // w_{2^k}

// MerkleTreeVerifier: top layer read from the seal, root recomputed and committed
struct TreeV {
    size_t rows, cols;
    unsigned layers, top_layer;
    std::vector<uint32_t> top;  // nodes [top_size, 2*top_size) as in the prover, indexable by node id - top_size
    size_t top_size() const { return (size_t)1 << top_layer; }
    void read_and_commit(Reader& rd, Transcript& T, const HostPoseidon2& h, size_t rows_, size_t cols_) {
        rows = rows_;
        cols = cols_;
        layers = ilog2u(rows);
        top_layer = top_layer_of(layers);
        size_t ts = top_size();
        const uint32_t* t = rd.take(8 * ts);
        top.assign(t, t + 8 * ts);
        // fold the top layer to the root
        std::vector<uint32_t> layer(top);
        for (size_t sz = ts; sz > 1; sz >>= 1) {
            std::vector<uint32_t> next(8 * (sz / 2));
            for (size_t i = 0; i < sz / 2; ++i) h.hash_elems(&next[8 * i], &layer[16 * i], 16);  // hash_pair == sponge of 16 words
            layer.swap(next);
        }
        T.commit(layer.data());
    }
    // reads `cols` values + the path from the seal, checks them against the top layer; returns the column values
    const uint32_t* verify_open(Reader& rd, const HostPoseidon2& h, size_t idx) const {
        const uint32_t* vals = rd.take(cols);
        uint32_t cur[8];
        h.hash_elems(cur, vals, cols);
        size_t node = idx + rows;
        while (node >= 2 * top_size()) {
            const uint32_t* sib = rd.take(8);
            uint32_t pair[16];
            if (node & 1) {
                memcpy(pair, sib, 32);
                memcpy(pair + 8, cur, 32);
            } else {
                memcpy(pair, cur, 32);
                memcpy(pair + 8, sib, 32);
            }
            h.hash_elems(cur, pair, 16);
            node >>= 1;
        }
        VCHECK(memcmp(cur, &top[8 * (node - top_size())], 32) == 0, "Merkle opening does not match the committed top layer");
        return vals;
    }
};

void verify(const uint32_t* seal, size_t words) {
    HostPoseidon2 h;
    h.load(POSEIDON2_RC, POSEIDON2_DIAG);
    Transcript T(&h);
    Reader rd{seal, words};

    // ---- header ----
    const uint32_t* hdr = rd.take(4);
    const uint32_t po2 = hdr[0];
    const uint32_t widths[4] = {hdr[1], hdr[2], hdr[3], BX_CHECK_SIZE};
    VCHECK(po2 >= 9 && po2 <= 24, "header: po2 out of range");
    VCHECK(widths[0] >= 1 && widths[1] >= 1 && widths[2] >= 1 && widths[0] < 65536 && widths[1] < 65536 && widths[2] < 65536,
           "header: bad group widths");
    {
        uint32_t enc[4], dg[8];
        for (int i = 0; i < 4; ++i) enc[i] = fp_encode(hdr[i]);
        h.hash_elems(dg, enc, 4);
        T.commit(dg);
    }
    const size_t N = (size_t)1 << po2, D = 4 * N;
    // ---- trace commitments ----
    TreeV trees[4];
    trees[0].read_and_commit(rd, T, h, D, widths[0]);
    trees[1].read_and_commit(rd, T, h, D, widths[1]);
    (void)T.random_ext();  // accum mix
    trees[2].read_and_commit(rd, T, h, D, widths[2]);
    const Fp4 poly_mix = T.random_ext();
    trees[3].read_and_commit(rd, T, h, D, widths[3]);
    const Fp4 Z = T.random_ext();
    // ---- taps ----
    std::vector<std::vector<uint32_t>> taps(4);
    size_t total_taps = 0;
    for (int g = 0; g < 4; ++g) {
        taps[g].assign(widths[g], 1);
        if (g == 1 || g == 2)
            for (uint32_t c = 0; c < widths[g]; c += 4) taps[g][c] = 2;
        for (uint32_t t : taps[g]) total_taps += t;
    }
    const uint32_t* coeff_u = rd.take(4 * total_taps);
    {
        uint32_t dg[8];
        h.hash_elems(dg, coeff_u, 4 * total_taps);
        T.commit(dg);
    }
    const uint32_t back_one = fp_inv(rou(po2));
    const Fp4 Zb = f4_scale(Z, back_one);
    const Fp4 Z4 = f4_scale(f4_pow(Z, 4), fp_inv(MONT_THREE));
    // ---- the check identity at Z:  sum_k X^k sum_q Z^(rev2 q) g_{4k+q}(Z^4/3)  ==  sum_c poly_mix^c (v_c^3 + v_c) ----
    {
        Fp4 rhs = f4_zero(), cur = f4_one();
        size_t u = 0;
        for (int g = 0; g < 3; ++g)
            for (uint32_t c = 0; c < widths[g]; ++c) {
                Fp4 v = ld(coeff_u + u);
                if (taps[g][c] == 2) v = f4_add(v, f4_mul(ld(coeff_u + u + 4), Z));  // u(Z) = c0 + c1 Z
                u += 4 * taps[g][c];
                Fp4 t = f4_add(f4_mul(f4_mul(v, v), v), v);
                rhs = f4_add(rhs, f4_mul(cur, t));
                cur = f4_mul(cur, poly_mix);
            }
        Fp4 lhs = f4_zero();
        const Fp4 zp[4] = {f4_one(), Z, f4_mul(Z, Z), f4_mul(f4_mul(Z, Z), Z)};
        for (int k = 0; k < 4; ++k) {
            Fp4 plane = f4_zero();
            for (int q = 0; q < 4; ++q) plane = f4_add(plane, f4_mul(zp[bit_reverse((uint32_t)q, 2)], ld(coeff_u + u + 4 * (4 * k + q))));
            Fp4 xk = f4_zero();
            xk.c[k] = MONT_ONE;  // the basis element X^k
            lhs = f4_add(lhs, f4_mul(xk, plane));
        }
        VCHECK(eq(lhs, rhs), "check polynomial identity fails at Z");
    }
    const Fp4 mix = T.random_ext();
    // mixed u polynomials per combo (as the prover subtracts them) and per-column mix powers
    Fp4 combo_u[3][2] = {{f4_zero(), f4_zero()}, {f4_zero(), f4_zero()}, {f4_zero(), f4_zero()}};
    std::vector<Fp4> mixpow;
    {
        Fp4 cur = f4_one();
        size_t u = 0;
        for (int g = 0; g < 4; ++g)
            for (uint32_t c = 0; c < widths[g]; ++c) {
                int id = g == 3 ? 2 : (taps[g][c] == 2 ? 1 : 0);
                for (uint32_t t = 0; t < taps[g][c]; ++t, u += 4) combo_u[id][t] = f4_add(combo_u[id][t], f4_mul(cur, ld(coeff_u + u)));
                mixpow.push_back(cur);
                cur = f4_mul(cur, mix);
            }
    }
    // ---- FRI commitments ----
    struct Round {
        size_t size;
        TreeV tree;
        Fp4 fold_mix;
    };
    std::vector<Round> rounds;
    size_t size = N;
    while (size > BX_FRI_MIN_DEGREE) {
        rounds.emplace_back();
        Round& r = rounds.back();
        r.size = size;
        r.tree.read_and_commit(rd, T, h, 4 * size / BX_FRI_FOLD, 4 * BX_FRI_FOLD);
        r.fold_mix = T.random_ext();
        size /= BX_FRI_FOLD;
    }
    const size_t final_size = size;
    const uint32_t* fin = rd.take(4 * final_size);  // SoA planes, natural coefficient order
    {
        uint32_t dg[8];
        h.hash_elems(dg, fin, 4 * final_size);
        T.commit(dg);
    }
    // ---- queries ----
    const uint32_t w16 = rou(4), w16_inv = fp_inv(w16), inv16 = fp_inv(fp_encode(16u));
    for (int q = 0; q < BX_QUERIES; ++q) {
        size_t pos = T.random_bits(ilog2u(D)) % D;
        const Fp4 y = from_base(fp_pow(rou(po2 + 2), pos));  // evaluation point of row `pos` (coset shift lives in the coefficients)
        // DEEP quotient from the opened rows
        Fp4 num[3] = {f4_zero(), f4_zero(), f4_zero()};
        size_t col_global = 0;
        for (int g = 0; g < 4; ++g) {
            const uint32_t* vals = trees[g].verify_open(rd, h, pos);
            for (uint32_t c = 0; c < widths[g]; ++c, ++col_global) {
                int id = g == 3 ? 2 : (taps[g][c] == 2 ? 1 : 0);
                num[id] = f4_add(num[id], f4_scale(mixpow[col_global], vals[c]));
            }
        }
        Fp4 goal = f4_zero();
        {
            Fp4 n0 = f4_sub(num[0], combo_u[0][0]);
            goal = f4_add(goal, f4_mul(n0, f4_inv(f4_sub(y, Z))));
            Fp4 n1 = f4_sub(num[1], f4_add(combo_u[1][0], f4_mul(combo_u[1][1], y)));
            goal = f4_add(goal, f4_mul(n1, f4_inv(f4_mul(f4_sub(y, Z), f4_sub(y, Zb)))));
            Fp4 n2 = f4_sub(num[2], combo_u[2][0]);
            goal = f4_add(goal, f4_mul(n2, f4_inv(f4_sub(y, Z4))));
        }
        // FRI chain
        size_t domain = D;
        for (Round& r : rounds) {
            const size_t rows = domain / BX_FRI_FOLD;
            const size_t group = pos % rows, quot = pos / rows;
            const uint32_t* vals = r.tree.verify_open(rd, h, group);
            Fp4 ev[BX_FRI_FOLD];
            for (int j = 0; j < BX_FRI_FOLD; ++j) ev[j] = Fp4{{vals[j], vals[16 + j], vals[32 + j], vals[48 + j]}};
            VCHECK(eq(ev[quot], goal), "FRI layer value does not match the value carried from the previous layer");
            // interpolate the 16 evaluations on {y0 * w16^t}: c_i = 1/16 sum_t ev[t] w16^(-i t); fold = sum_i c_i (mix / y0)^i
            const uint32_t y0_inv = fp_inv(fp_pow(rou(ilog2u(domain)), group));
            Fp4 folded = f4_zero(), mp = f4_one();
            uint32_t yi = MONT_ONE;
            for (int i = 0; i < BX_FRI_FOLD; ++i) {
                Fp4 ci = f4_zero();
                for (int t = 0; t < BX_FRI_FOLD; ++t) ci = f4_add(ci, f4_scale(ev[t], fp_pow(w16_inv, (uint64_t)((i * t) % 16))));
                ci = f4_scale(ci, inv16);
                folded = f4_add(folded, f4_mul(f4_scale(ci, yi), mp));
                yi = fp_mul(yi, y0_inv);
                mp = f4_mul(mp, r.fold_mix);
            }
            goal = folded;
            pos = group;
            domain = rows;
        }
        // final polynomial (degree < final_size) at the last point
        {
            const Fp4 x = from_base(fp_pow(rou(ilog2u(domain)), pos));
            Fp4 acc = f4_zero();
            for (size_t i = final_size; i-- > 0;) {
                Fp4 ci{{fin[i], fin[final_size + i], fin[2 * final_size + i], fin[3 * final_size + i]}};
                acc = f4_add(f4_mul(acc, x), ci);
            }
            VCHECK(eq(acc, goal), "final polynomial does not match the last FRI fold");
        }
    }
    VCHECK(rd.pos == rd.n, "trailing words after the last query");
}

}  // namespace

extern "C" const char* bx_verify_segment(const uint32_t* seal, size_t seal_words) {
    static thread_local char err[256];
    if (!seal) return "bx_verify_segment: null seal";
    try {
        verify(seal, seal_words);
    } catch (const Fail& f) {
        snprintf(err, sizeof err, "bx_verify_segment: %s", f.msg.c_str());
        return err;
    } catch (const std::exception& e) {
        snprintf(err, sizeof err, "bx_verify_segment: %s", e.what());
        return err;
    }
    return nullptr;
}
