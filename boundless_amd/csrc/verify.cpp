// verify.cpp — host-side verifier of the segment seal (include/bx_prover.h: bx_verify_segment).
//
// Counterpart of `SegmentReceipt::verify_integrity_with_context` (bento/crates/workflow/src/tasks/prove.rs:53-55) for the
// circuit-independent pipeline of bx_prove_segment.  Structure follows risc0_zkp::verify::{verify, fri::fri_verify,
// merkle::MerkleTreeVerifier} (risc0-zkp 3.0.3, reference Cargo.lock:9155): replay the transcript, check the constraint
// identity at the random point Z (here: of the synthetic circuit specified in bx_prover.h), then for each query verify the
// Merkle openings, recompute the DEEP quotient from the opened trace rows and follow the FRI folds to the final polynomial.
// Pure CPU code (the reference verifies on the CPU as well); it shares only fp.hpp/transcript.hpp with the prover.
#include <stdio.h>
#include <string.h>

#include <stdlib.h>

#include <atomic>
#include <string>
#include <thread>
#include <vector>

#include "../../include/bx_circuit.h"
#include "../../include/bx_prover.h"
#include "circuit.hpp"
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
    // field elements and digest words: only the canonical representative is a valid encoding.  (x + P hashes, adds and
    // multiplies like x in the lazily reduced arithmetic, so without this check seals would be malleable and the
    // 64-bit contracts of fp_mad_lazy would not hold for words taken from the seal.)
    const uint32_t* take_elems(size_t k) {
        const uint32_t* r = take(k);
        for (size_t i = 0; i < k; ++i)
            if (r[i] >= P) throw Fail{"non-canonical field element in the seal"};
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
    uint32_t root[8];           // recomputed from the top layer
    size_t top_size() const { return (size_t)1 << top_layer; }
    void read_and_commit(Reader& rd, Transcript& T, const HostPoseidon2& h, size_t rows_, size_t cols_) {
        rows = rows_;
        cols = cols_;
        layers = ilog2u(rows);
        top_layer = top_layer_of(layers);
        size_t ts = top_size();
        const uint32_t* t = rd.take_elems(8 * ts);
        top.assign(t, t + 8 * ts);
        // fold the top layer to the root
        std::vector<uint32_t> layer(top);
        for (size_t sz = ts; sz > 1; sz >>= 1) {
            std::vector<uint32_t> next(8 * (sz / 2));
            for (size_t i = 0; i < sz / 2; ++i) h.hash_elems(&next[8 * i], &layer[16 * i], 16);  // hash_pair == sponge of 16 words
            layer.swap(next);
        }
        memcpy(root, layer.data(), 32);
        T.commit(root);
    }
    // reads `cols` values + the path from the seal, checks them against the top layer; returns the column values
    const uint32_t* verify_open(Reader& rd, const HostPoseidon2& h, size_t idx) const {
        const uint32_t* vals = rd.take_elems(cols);
        uint32_t cur[8];
        h.hash_elems(cur, vals, cols);
        size_t node = idx + rows;
        while (node >= 2 * top_size()) {
            const uint32_t* sib = rd.take_elems(8);
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

}  // namespace
namespace bx {
bool verifier_ctx_contains(const bx_verifier_ctx* v, uint32_t po2, const uint32_t root[8]);  // control_id.cpp
}
namespace {

// threads that share the 50 queries of one verification: bx_verify_set_threads, else BX_VERIFY_THREADS, else min(4, cores)
std::atomic<int> g_verify_threads{0};
int verify_threads() {
    int n = g_verify_threads.load(std::memory_order_relaxed);
    if (n > 0) return n;
    static const int from_env = [] {  // read once: getenv is not safe against a concurrent setenv (e.g. from Python)
        const char* e = getenv("BX_VERIFY_THREADS");
        const int v = e ? atoi(e) : 0;
        return v >= 1 && v <= 64 ? v : 0;
    }();
    if (from_env) return from_env;
    unsigned hw = std::thread::hardware_concurrency();
    return hw >= 4 ? 4 : hw >= 1 ? (int)hw : 1;
}

void verify(const uint32_t* seal, size_t words, const bx_circuit_ops* circ, const bx_verifier_ctx* vctx) {
    HostPoseidon2 h;
    h.load(POSEIDON2_RC, POSEIDON2_DIAG);
    Transcript T(&h);
    Reader rd{seal, words};

    // ---- header ----
    const uint32_t* hdr = rd.take(6);
    const uint32_t po2 = hdr[0];
    const uint32_t widths[4] = {hdr[1], hdr[2], hdr[3], BX_CHECK_SIZE};
    VCHECK(po2 >= 9 && po2 <= 24, "header: po2 out of range");
    VCHECK(widths[0] >= 1 && widths[1] >= 1 && widths[2] >= 1 && widths[0] < 65536 && widths[1] < 65536 && widths[2] < 65536,
           "header: bad group widths");
    // the circuit's knobs are part of the statement: they must be a fixed point of the circuit's own normalisation
    bx_segment_params shape{po2, widths[0], widths[1], widths[2], hdr[4], hdr[5]};
    {
        bx_segment_params norm = shape;
        // a zero knob is "use the default" on the proving side and never reaches a seal of the built-in circuit; a plug-in
        // circuit may leave the two words unused (bx_circuit.h: they are the circuit's to interpret), so only its own
        // normalisation decides
        if (circ == bx_synthetic_circuit()) VCHECK(hdr[4] != 0 && hdr[5] != 0, "header: bad circuit knobs");
        if (circ->normalize) VCHECK(circ->normalize(circ->user, &norm) == nullptr, "header: bad circuit knobs");
        VCHECK(norm.cons_terms == shape.cons_terms && norm.cons_degree == shape.cons_degree, "header: bad circuit knobs");
    }
    {
        uint32_t enc[6], dg[8];
        for (int i = 0; i < 6; ++i) enc[i] = fp_encode(hdr[i]);
        h.hash_elems(dg, enc, 6);
        T.commit(dg);
    }
    const size_t N = (size_t)1 << po2, D = 4 * N;
    // ---- the statement's public words ----
    const uint32_t n_globals = circ->n_globals ? circ->n_globals(circ->user, &shape) : 0;
    VCHECK(n_globals <= BX_MAX_GLOBALS, "circuit: too many public words");
    const uint32_t* globals = rd.take_elems(n_globals);
    if (n_globals) {
        uint32_t dg[8];
        h.hash_elems(dg, globals, n_globals);
        T.commit(dg);
    }
    // ---- trace commitments ----
    TreeV trees[4];
    trees[0].read_and_commit(rd, T, h, D, widths[0]);
    // ---- the code group is the circuit's, not the prover's: its root must be a control ID (upstream: check_code(po2, root)).
    //      An explicit context is a lookup and is asked at once; the circuit's own check may have to compute the ID (seconds for
    //      a shape outside its table), so it runs last: only a seal that is otherwise a valid proof can make the verifier pay ----
    if (vctx) VCHECK(verifier_ctx_contains(vctx, po2, trees[0].root), "the code group's root is not one of the verifier context's control IDs for this po2");
    else VCHECK(circ->check_code != nullptr, "no control IDs to check the code root against (the circuit table has no check_code: pass a verifier context)");
    trees[1].read_and_commit(rd, T, h, D, widths[1]);
    const Fp4 beta = T.random_ext();  // the accumulators' mix
    trees[2].read_and_commit(rd, T, h, D, widths[2]);
    const Fp4 poly_mix = T.random_ext();
    trees[3].read_and_commit(rd, T, h, D, widths[3]);
    const Fp4 Z = T.random_ext();
    // ---- taps ----
    // every column contributes at least one ext tap value to coeff_u: refuse a header whose widths the seal cannot back before
    // spending any per-column work on it (an 8 KB seal claiming three 65535-column groups used to cost seconds here)
    VCHECK(rd.n - rd.pos >= 4 * ((size_t)widths[0] + widths[1] + widths[2] + widths[3]), "seal truncated");
    // tap set of every column (the rows back it is opened at) and the combo it belongs to: columns with the same set share
    // a DEEP combination polynomial, combos in order of first appearance, the check group's last (as in prover.hip)
    std::vector<std::vector<std::vector<uint32_t>>> backs(4);
    std::vector<std::vector<uint32_t>> combo(4);
    std::vector<std::vector<uint32_t>> combo_backs;
    size_t total_taps = 0;
    for (int g = 0; g < 4; ++g) {
        backs[g].resize(widths[g]);
        combo[g].resize(widths[g]);
        for (uint32_t c = 0; c < widths[g]; ++c) {
            if (g == 3) {
                backs[g][c] = {0};
            } else {
                uint32_t bk[BX_MAX_TAPS];
                const uint32_t k = circ->taps(circ->user, &shape, g, c, bk);
                VCHECK(k >= 1 && k <= BX_MAX_TAPS && bk[0] == 0, "circuit: a tap set has 1..8 entries and starts with 0");
                for (uint32_t t = 1; t < k; ++t) VCHECK(bk[t] > bk[t - 1] && bk[t] < N, "circuit: tap sets are strictly increasing");
                backs[g][c].assign(bk, bk + k);
                size_t id = 0;
                while (id < combo_backs.size() && combo_backs[id] != backs[g][c]) ++id;
                if (id == combo_backs.size()) combo_backs.push_back(backs[g][c]);
                combo[g][c] = (uint32_t)id;
            }
            total_taps += backs[g][c].size();
        }
    }
    VCHECK(combo_backs.size() + 1 <= BX_MAX_COMBOS, "circuit: too many distinct tap sets");
    const size_t n_trace_combos = combo_backs.size(), n_combos = n_trace_combos + 1;
    for (uint32_t c = 0; c < widths[3]; ++c) combo[3][c] = (uint32_t)n_trace_combos;
    const uint32_t* coeff_u = rd.take_elems(4 * total_taps);
    {
        uint32_t dg[8];
        h.hash_elems(dg, coeff_u, 4 * total_taps);
        T.commit(dg);
    }
    const uint32_t back_one = fp_inv(rou(po2));
    const Fp4 Z4 = f4_scale(f4_pow(Z, 4), fp_inv(MONT_THREE));
    auto zback = [&](uint32_t b) { return f4_scale(Z, fp_pow(back_one, b)); };  // Z * w_N^-b
    // ---- the constraint identity at Z:  check(Z) * ((3Z)^N - 1)  ==  sum_i poly_mix^i C_i(taps)  with
    //      check(Z) = sum_k X^k sum_q Z^(rev2 q) g_{4k+q}(Z^4/3) from the check group's taps ----
    {
        // tap values: where[g][c] = offset of column c's first ext element in coeff_u
        std::vector<std::vector<size_t>> where(4);
        size_t u = 0;
        for (int g = 0; g < 4; ++g) {
            where[g].resize(widths[g]);
            for (uint32_t c = 0; c < widths[g]; ++c) {
                where[g][c] = u;
                u += 4 * backs[g][c].size();
            }
        }
        auto at = [&](int g, uint32_t c, int back) -> Fp4 {  // the column's polynomial at Z (back 0) or Z * w_N^-1 (back 1)
            VCHECK(g >= 0 && g < 4 && c < widths[g] && back >= 0, "internal: tap out of range");
            const auto& B = backs[g][c];
            bool member = false;
            for (uint32_t b : B) member |= b == (uint32_t)back;
            VCHECK(member, "circuit: reads a tap that is not in the column's tap set");
            const uint32_t* w = coeff_u + where[g][c];
            if (B.size() == 1) return ld(w);
            const Fp4 x = g == 3 ? Z4 : zback((uint32_t)back);  // u(x) = sum_t c_t x^t (Horner)
            Fp4 v = f4_zero();
            for (size_t t = B.size(); t-- > 0;) v = f4_add(f4_mul(v, x), ld(w + 4 * t));
            return v;
        };
        // the circuit evaluates sum_i poly_mix^i C_i from the taps (upstream: the circuit's poly_ext)
        struct TapCtx {
            decltype(at)* fn;
        } tctx{&at};
        bx_tap_reader reader{&tctx, [](const void* ctx, int g, uint32_t c, int back, uint32_t out[4]) -> const char* {
                                 try {
                                     const Fp4 v = (*((const TapCtx*)ctx)->fn)(g, c, back);
                                     memcpy(out, v.c, 16);
                                     return nullptr;
                                 } catch (const Fail&) {
                                     return "tap not available";
                                 }
                             }};
        Fp4 rhs;
        const char* ce = circ->constraints_at(circ->user, &shape, &reader, poly_mix.c, beta.c, globals, rhs.c);
        VCHECK(ce == nullptr, std::string("circuit: ") + (ce ? ce : ""));
        Fp4 lhs = f4_zero();
        const Fp4 zp[4] = {f4_one(), Z, f4_mul(Z, Z), f4_mul(f4_mul(Z, Z), Z)};
        for (int k = 0; k < 4; ++k) {
            Fp4 plane = f4_zero();
            for (int q = 0; q < 4; ++q) plane = f4_add(plane, f4_mul(zp[bit_reverse((uint32_t)q, 2)], at(3, (uint32_t)(4 * k + q), 0)));
            Fp4 xk = f4_zero();
            xk.c[k] = MONT_ONE;  // the basis element X^k
            lhs = f4_add(lhs, f4_mul(xk, plane));
        }
        const Fp4 vanish = f4_sub(f4_pow(f4_scale(Z, MONT_THREE), (uint64_t)N), f4_one());
        VCHECK(eq(f4_mul(lhs, vanish), rhs), "constraint identity fails at Z");
    }
    const Fp4 mix = T.random_ext();
    // mixed u polynomials per combo (as the prover subtracts them) and per-column mix powers
    std::vector<Fp4> combo_u(n_combos * BX_MAX_TAPS, f4_zero());
    std::vector<Fp4> mixpow;
    {
        Fp4 cur = f4_one();
        size_t u = 0;
        for (int g = 0; g < 4; ++g)
            for (uint32_t c = 0; c < widths[g]; ++c) {
                const uint32_t id = combo[g][c];
                for (size_t t = 0; t < backs[g][c].size(); ++t, u += 4)
                    combo_u[id * BX_MAX_TAPS + t] = f4_add(combo_u[id * BX_MAX_TAPS + t], f4_mul(cur, ld(coeff_u + u)));
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
    const uint32_t* fin = rd.take_elems(4 * final_size);  // SoA planes, natural coefficient order
    {
        uint32_t dg[8];
        h.hash_elems(dg, fin, 4 * final_size);
        T.commit(dg);
    }
    // ---- queries ----
    const uint32_t w16 = rou(4), w16_inv = fp_inv(w16), inv16 = fp_inv(fp_encode(16u));
    // The queries commit nothing to the transcript, so their positions do not depend on what they open: all are drawn first, and
    // — every opening having a length fixed by the shape — each query's words are at a known offset.  The queries are then
    // checked independently, on a few threads (bx_verify_set_threads); the verdict is the one of the first failing query in seal
    // order, exactly what reading the seal front to back reports.
    size_t positions[BX_QUERIES];
    for (int q = 0; q < BX_QUERIES; ++q) positions[q] = T.random_bits(ilog2u(D)) % D;
    size_t per_query = 0;
    for (int g = 0; g < 4; ++g) per_query += trees[g].cols + 8 * (size_t)(trees[g].layers - trees[g].top_layer);
    for (const Round& r : rounds) per_query += r.tree.cols + 8 * (size_t)(r.tree.layers - r.tree.top_layer);
    const size_t queries_at = rd.pos;
    auto check_query = [&](int q) {
        Reader rd{seal, words};  // this query's own cursor (shadows the stream's)
        rd.pos = queries_at + (size_t)q * per_query;
        if (rd.pos > rd.n) throw Fail{"seal truncated"};
        size_t pos = positions[q];
        const Fp4 y = from_base(fp_pow(rou(po2 + 2), pos));  // evaluation point of row `pos` (coset shift lives in the coefficients)
        // DEEP quotient from the opened rows
        std::vector<Fp4> num(n_combos, f4_zero());
        size_t col_global = 0;
        for (int g = 0; g < 4; ++g) {
            const uint32_t* vals = trees[g].verify_open(rd, h, pos);
            for (uint32_t c = 0; c < widths[g]; ++c, ++col_global) {
                const uint32_t id = combo[g][c];
                num[id] = f4_add(num[id], f4_scale(mixpow[col_global], vals[c]));
            }
        }
        // goal = sum over combos of (combo(y) - u_combo(y)) / prod_{b in its tap set} (y - Z w_N^-b); the check combo's point is Z^4/3
        Fp4 goal = f4_zero();
        for (size_t id = 0; id < n_combos; ++id) {
            const size_t k = id < n_trace_combos ? combo_backs[id].size() : 1;
            Fp4 uy = f4_zero();
            for (size_t t = k; t-- > 0;) uy = f4_add(f4_mul(uy, y), combo_u[id * BX_MAX_TAPS + t]);
            Fp4 den = f4_one();
            for (size_t t = 0; t < k; ++t) den = f4_mul(den, f4_sub(y, id < n_trace_combos ? zback(combo_backs[id][t]) : Z4));
            goal = f4_add(goal, f4_mul(f4_sub(num[id], uy), f4_inv(den)));
        }
        // FRI chain
        size_t domain = D;
        for (const Round& r : rounds) {
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
    };
    {
        int threads = verify_threads();
        if (threads > BX_QUERIES) threads = BX_QUERIES;
        std::string errs[BX_QUERIES];
        bool failed[BX_QUERIES] = {false};
        std::atomic<int> first_bad{BX_QUERIES};
        auto worker = [&](int t) {
            for (int q = t; q < BX_QUERIES; q += threads) {
                if (q > first_bad.load(std::memory_order_relaxed)) break;  // an earlier query already decides the verdict
                try {
                    check_query(q);
                } catch (const Fail& f) {
                    failed[q] = true;
                    try {
                        errs[q] = f.msg;
                    } catch (...) {
                    }
                } catch (const std::exception& e) {
                    failed[q] = true;
                    try {
                        errs[q] = e.what();
                    } catch (...) {
                    }
                }
                if (failed[q]) {
                    int cur = first_bad.load();
                    while (q < cur && !first_bad.compare_exchange_weak(cur, q)) {
                    }
                }
            }
        };
        std::vector<std::thread> pool;
        try {
            for (int t = 1; t < threads; ++t) pool.emplace_back(worker, t);
        } catch (...) {  // no more threads to be had: the calling thread checks what nobody was started for
            const int started = (int)pool.size() + 1;
            for (auto& th : pool) th.join();
            pool.clear();
            for (int t = started; t < threads; ++t) worker(t);
        }
        worker(0);
        for (auto& th : pool) th.join();
        for (int q = 0; q < BX_QUERIES; ++q)
            if (failed[q]) throw Fail{errs[q].empty() ? "query check failed" : errs[q]};
    }
    rd.pos = queries_at + (size_t)BX_QUERIES * per_query;
    if (rd.pos > rd.n) throw Fail{"seal truncated"};
    VCHECK(rd.pos == rd.n, "trailing words after the last query");
    if (!vctx) {
        const char* ce = circ->check_code(circ->user, &shape, trees[0].root);
        VCHECK(ce == nullptr, std::string("control ID: ") + (ce ? ce : ""));
    }
}

}  // namespace

// sum_i poly_mix^i C_i of the synthetic circuit (include/bx_prover.h) from the tap values: the verifier-side half of the
// bx_circuit_ops table built in circuit.hip.  Host arithmetic only.
namespace bx {
const char* synthetic_constraints_at(void*, const bx_segment_params* shape, const bx_tap_reader* taps, const uint32_t poly_mix_w[4],
                                     const uint32_t mix_w[4], const uint32_t* globals, uint32_t out[4]) {
    const Circuit cc(shape->po2, shape->w_code, shape->w_data, shape->w_accum, shape->cons_terms, shape->cons_degree);
    const Fp4 poly_mix = ld(poly_mix_w), beta = ld(mix_w);
    const char* err = nullptr;
    auto at = [&](int g, uint32_t c, int back) -> Fp4 {
        Fp4 v = f4_zero();
        if (const char* e = taps->at(taps->ctx, g, c, back, v.c)) err = e;
        return v;
    };
    Fp4 rhs = f4_zero(), cur = f4_one();
    for (uint32_t j = 0; j < cc.J; ++j) {
        Fp4 pool[Circuit::POOL];
        for (unsigned slot = 0; slot < Circuit::POOL; ++slot) {
            const Circuit::Src src = cc.pool_src(j, slot);
            pool[slot] = src.group < 0 ? f4_one() : at(src.group, src.col, src.back);
        }
        Fp4 sum = f4_zero();
        for (uint32_t t = 0; t < cc.T; ++t) {
            Fp4 prod = pool[Circuit::pool_idx(t, 0)];
            for (uint32_t f = 1; f < cc.G; ++f) prod = f4_mul(prod, pool[Circuit::pool_idx(t, f)]);
            sum = f4_add(sum, prod);
        }
        rhs = f4_add(rhs, f4_mul(cur, f4_sub(at(1, cc.F + j, 0), sum)));
        cur = f4_mul(cur, poly_mix);
    }
    auto acc_at = [&](uint32_t e, int back) -> Fp4 {  // the ext-valued accumulator: sum_k X^k * column(4e+k)
        Fp4 r = f4_zero();
        for (int k = 0; k < 4; ++k) {
            Fp4 xk = f4_zero();
            xk.c[k] = MONT_ONE;
            r = f4_add(r, f4_mul(xk, at(2, 4 * e + k, back)));
        }
        return r;
    };
    const Fp4 first = at(0, 0, 0);
    Fp4 be = beta;
    for (uint32_t e = 0; e < cc.E; ++e) {
        Fp4 inner = f4_add(first, f4_mul(f4_sub(f4_one(), first), acc_at(e, 1)));
        Fp4 cons = f4_sub(acc_at(e, 0), f4_mul(inner, f4_add(be, at(1, cc.acc_src(e), 0))));
        rhs = f4_add(rhs, f4_mul(cur, cons));
        cur = f4_mul(cur, poly_mix);
        if (e & 1) be = f4_mul(be, beta);  // beta^(floor(e/2)+1)
    }
    for (uint32_t p = 0; p < cc.pairs; ++p) {
        Fp4 cons = f4_mul(at(0, 1, 0), f4_sub(acc_at(2 * p + 1, 0), acc_at(2 * p, 0)));
        rhs = f4_add(rhs, f4_mul(cur, cons));
        cur = f4_mul(cur, poly_mix);
    }
    // boundary constraints tying the public words to the trace
    rhs = f4_add(rhs, f4_mul(cur, f4_mul(first, f4_sub(at(1, 0, 0), from_base(globals[0])))));
    cur = f4_mul(cur, poly_mix);
    if (cc.globals() > 1) {
        rhs = f4_add(rhs, f4_mul(cur, f4_mul(at(0, 1, 0), f4_sub(at(1, cc.wd - 1, 0), from_base(globals[1])))));
        cur = f4_mul(cur, poly_mix);
    }
    memcpy(out, rhs.c, 16);
    return err;
}
}  // namespace bx

// The compiled-in Poseidon2 table (canonical integers): what bx_init loads into every new ctx and what the verifier uses.
// Needs no ctx and no GPU, so that the fixture manifest (tests/golden/MANIFEST.json) can pin its SHA-256 on any host.
extern "C" const char* bx_poseidon2_default_params(uint32_t* rc213, uint32_t* diag24) {
    if (!rc213 || !diag24) return "bx_poseidon2_default_params: null table";
    const uint32_t* rc = POSEIDON2_RC;
    for (int i = 0; i < 213; ++i) rc213[i] = rc[i];
    for (int i = 0; i < 24; ++i) diag24[i] = POSEIDON2_DIAG[i];
    return nullptr;
}

extern "C" const char* bx_verify_set_threads(int threads) {
    if (threads < 0 || threads > 64) return "bx_verify_set_threads: 0 (default) .. 64";
    g_verify_threads.store(threads);
    return nullptr;
}
extern "C" const char* bx_verify_segment(const uint32_t* seal, size_t seal_words) {
    return bx_verify_segment_with_circuit(seal, seal_words, nullptr);
}
extern "C" const char* bx_verify_segment_with_circuit(const uint32_t* seal, size_t seal_words, const bx_circuit_ops* circuit) {
    return bx_verify_segment_with_context(seal, seal_words, circuit, nullptr);
}
extern "C" const char* bx_verify_segment_with_context(const uint32_t* seal, size_t seal_words, const bx_circuit_ops* circuit,
                                                      const bx_verifier_ctx* vctx) {
    static thread_local char err[384];
    if (!seal) return "bx_verify_segment: null seal";
    if (!circuit) circuit = bx_synthetic_circuit();
    if (!circuit->taps || !circuit->constraints_at) return "bx_verify_segment: circuit table incomplete";
    try {
        verify(seal, seal_words, circuit, vctx);
    } catch (const Fail& f) {
        snprintf(err, sizeof err, "bx_verify_segment: %s", f.msg.c_str());
        return err;
    } catch (const std::exception& e) {
        snprintf(err, sizeof err, "bx_verify_segment: %s", e.what());
        return err;
    }
    return nullptr;
}
