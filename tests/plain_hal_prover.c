/*
 * plain_hal_prover.c — TEST-SIDE driver: one whole segment proof sequenced from OUTSIDE the library through nothing but the
 * plain `Hal`-trait entry points of SURVEY.md section 8(b2) (include/bx_hal.h) and the public circuit table (include/bx_circuit.h).
 *
 * What it is for: the boundary north_star names is `risc0_zkp::hal::Hal`.  A maintainer who binds
 * libbx_hip_hal.so behind `impl Hal for HipHal` gets risc0-zkp's own prover calling the trait
 * methods one by one — `bento/crates/workflow/src/tasks/prove.rs:41-49` -> `bento/crates/workflow/src/lib.rs:246-249` ->
 * [EXT] risc0_zkp::prove::{Prover::commit_group, Prover::finalize, fri::fri_prove, merkle::MerkleTreeProver::{new, prove}}
 * (risc0-zkp 3.0.3, reference Cargo.lock:9155).  This file restates THAT call order in plain C:
 *
 *     commit_group      bx_batch_interpolate_ntt, bx_zk_shift, bx_batch_expand_into_evaluate_ntt, bx_batch_bit_reverse
 *     MerkleTreeProver  bx_hash_rows, one bx_hash_fold per layer, root + top layer read with bx_d2h (WriteIOP needs them)
 *     DEEP              bx_batch_evaluate_any per group, `combos` from bx_alloc_zeroed per proof (alloc_extelem_zeroed), bx_mix_poly_coeffs per group,
 *                       Buffer::view_mut of the low coefficients (bx_d2h / bx_h2d), bx_poly_divide per combo and point,
 *                       bx_eltwise_sum_extelem, bx_batch_bit_reverse
 *     fri_prove         bx_batch_expand_into_evaluate_ntt, Merkle as above, bx_fri_fold with the HOST's mix, bx_eltwise_copy_elem
 *     queries           bx_gather_sample per opened row and per path digest, one bx_d2h per tree
 *     circuit stages    bx_synthetic_circuit()'s code_group / witgen / accumulate / eval_check through the public table
 *
 * with buffers sliced by pointer arithmetic on bx_buf as `Buffer::slice` would.  The host half (Poseidon2 transcript, Fp4
 * arithmetic, Lagrange matrices) is written here a third time, from the parameters bx_poseidon2_default_params() publishes;
 * it uses neither the library's prover (csrc/prover.hip) nor oracle/.  tests/test_plain_hal_gpu.py then requires
 *     seal(this driver) == seal(bx_prove_segment) == seal(oracle)      word for word,
 * and bench.py times it beside bx_prove_segment (`single_proof_ms.plain_hal`, untimed extra).
 *
 * `flags` swap ONE plain call sequence each for the library's extension entry point that the in-library prover uses, so that
 * INTEGRATION.md can quote what each extension is worth instead of a guess:
 */
#define _POSIX_C_SOURCE 199309L /* clock_gettime under -std=c99 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "bx_circuit.h"
#include "bx_hal.h"
#include "bx_prover.h"

#define PH_EXT_INTERPOLATE_ZK 1u /* bx_batch_interpolate_zk            for interpolate_ntt + zk_shift                     */
#define PH_EXT_MERKLE_BUILD 2u   /* bx_merkle_build                    for hash_rows + one hash_fold per layer            */
#define PH_EXT_COEFFS_BITREV 4u  /* trace coefficients stay bit-reversed: bx_batch_evaluate_any_bitrev +
                                    bx_batch_bit_reverse_ext of the combos for batch_bit_reverse of every group (N >= 2^15) */
#define PH_EXT_DIVIDE_BATCH 8u   /* bx_poly_divide_batch_indexed       for one bx_poly_divide per combo and point         */
#define PH_EXT_QUERY_GATHER 16u  /* bx_merkle_query_gather             for gather_sample per row / per digest             */
#define PH_EXT_EVAL_PTRS 32u     /* bx_batch_evaluate_ptrs (one call)  for batch_evaluate_any per group (N >= 2^15)       */
#define PH_EXT_ALL 63u
#define PH_ALLOC_PER_PROOF 64u   /* not an extension: the big buffers (coefficients, evaluations, Merkle nodes, FRI rounds, query buffer:
                                    ~9 GB at 2^20 / 16-256-64) are allocated at the start of every proof and released at its end, as
                                    risc0-zkp's prover does (hal.alloc_* inside commit_group / finalize / fri_prove) — what bx_alloc /
                                    bx_release cost a trait-level caller.  Default: everything is allocated once in ph_create. */

#define P BX_P
#define MONT_ONE 268435454u
#define MONT_NBETA 1073741848u /* encode(P - 11) */
#define MONT_BETA 939524073u
#define R2 1172168163u
#define MAX_ROUNDS 8
#define MAX_TREES (4 + MAX_ROUNDS)

/* ---------------------------------------------------------------- BabyBear, Montgomery words ------------------------------- */
static uint32_t fadd(uint32_t a, uint32_t b) { uint32_t s = a + b; return s >= P ? s - P : s; }
static uint32_t fsub(uint32_t a, uint32_t b) { return a >= b ? a - b : a + P - b; }
static uint32_t fmul(uint32_t a, uint32_t b) {
    uint64_t t = (uint64_t)a * b;
    uint32_t m = (uint32_t)t * 0x77FFFFFFu; /* -P^-1 mod 2^32 */
    uint32_t r = (uint32_t)((t + (uint64_t)m * P) >> 32);
    return r >= P ? r - P : r;
}
static uint32_t fenc(uint32_t canonical) { return fmul(R2, canonical % P); }
static uint32_t fdec(uint32_t mont) { return fmul(1u, mont); }
static uint32_t fpow(uint32_t a, uint64_t e) {
    uint32_t r = MONT_ONE;
    for (; e; e >>= 1, a = fmul(a, a))
        if (e & 1) r = fmul(r, a);
    return r;
}
static uint32_t finv(uint32_t a) { return fpow(a, (uint64_t)P - 2); }
typedef struct { uint32_t c[4]; } ext;
static ext xzero(void) { ext r = {{0, 0, 0, 0}}; return r; }
static ext xone(void) { ext r = {{MONT_ONE, 0, 0, 0}}; return r; }
static ext xadd(ext a, ext b) { ext r; for (int k = 0; k < 4; ++k) r.c[k] = fadd(a.c[k], b.c[k]); return r; }
static ext xsub(ext a, ext b) { ext r; for (int k = 0; k < 4; ++k) r.c[k] = fsub(a.c[k], b.c[k]); return r; }
static ext xscale(ext a, uint32_t s) { ext r; for (int k = 0; k < 4; ++k) r.c[k] = fmul(a.c[k], s); return r; }
/* Fp[X] / (X^4 + 11): schoolbook product, the high half folded with X^4 = -11 */
static ext xmul(ext a, ext b) {
    uint32_t t[7] = {0, 0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) t[i + j] = fadd(t[i + j], fmul(a.c[i], b.c[j]));
    ext r;
    for (int k = 0; k < 4; ++k) r.c[k] = k < 3 ? fadd(t[k], fmul(MONT_NBETA, t[k + 4])) : t[3];
    return r;
}
static ext xpow(ext a, uint64_t e) {
    ext r = xone();
    for (; e; e >>= 1, a = xmul(a, a))
        if (e & 1) r = xmul(r, a);
    return r;
}
/* a^-1 = a^(P^4 - 2) would need big exponents; use the norm chain instead: a(X) a(-X) lies in Fp[X^2] */
static ext xinv(ext a) {
    ext an = {{a.c[0], fsub(0, a.c[1]), a.c[2], fsub(0, a.c[3])}};
    ext n = xmul(a, an); /* n.c[1] == n.c[3] == 0: n = b0 + b2 X^2 */
    uint32_t b0 = n.c[0], b2 = n.c[2];
    uint32_t d = finv(fadd(fmul(b0, b0), fmul(MONT_BETA, fmul(b2, b2)))); /* (b0 + b2 Y)(b0 - b2 Y) = b0^2 + 11 b2^2, Y^2 = -11 */
    ext c = {{fmul(b0, d), 0, fsub(0, fmul(b2, d)), 0}};
    return xmul(an, c);
}

/* ---------------------------------------------------------------- Poseidon2 t = 24 and the transcript ----------------------- */
typedef struct { uint32_t rc[213], diag[24]; } p2params;
static uint32_t sbox(uint32_t x) { uint32_t x2 = fmul(x, x), x4 = fmul(x2, x2); return fmul(fmul(x2, x), x4); }
static void m4(uint32_t* x) { /* [[5,7,1,3],[4,6,1,1],[1,3,5,7],[1,1,4,6]] */
    uint64_t a = x[0], b = x[1], c = x[2], d = x[3];
    x[0] = (uint32_t)((5 * a + 7 * b + c + 3 * d) % P);
    x[1] = (uint32_t)((4 * a + 6 * b + c + d) % P);
    x[2] = (uint32_t)((a + 3 * b + 5 * c + 7 * d) % P);
    x[3] = (uint32_t)((a + b + 4 * c + 6 * d) % P);
}
static void m_ext(uint32_t* s) {
    uint32_t t[4] = {0, 0, 0, 0};
    for (int k = 0; k < 24; k += 4) m4(s + k);
    for (int i = 0; i < 24; ++i) t[i & 3] = fadd(t[i & 3], s[i]);
    for (int i = 0; i < 24; ++i) s[i] = fadd(s[i], t[i & 3]);
}
static void p2_mix(const p2params* h, uint32_t* s) {
    const uint32_t* c = h->rc;
    m_ext(s);
    for (int r = 0; r < 4; ++r, c += 24) {
        for (int i = 0; i < 24; ++i) s[i] = sbox(fadd(s[i], c[i]));
        m_ext(s);
    }
    for (int r = 0; r < 21; ++r) {
        uint32_t sum = 0;
        s[0] = sbox(fadd(s[0], *c++));
        for (int i = 0; i < 24; ++i) sum = fadd(sum, s[i]);
        for (int i = 0; i < 24; ++i) s[i] = fadd(sum, fmul(h->diag[i], s[i]));
    }
    for (int r = 0; r < 4; ++r, c += 24) {
        for (int i = 0; i < 24; ++i) s[i] = sbox(fadd(s[i], c[i]));
        m_ext(s);
    }
}
/* unpadded_hash: overwrite sponge, rate 16, a partial (or empty) last block zero-filled */
static void p2_hash(const p2params* h, uint32_t out[8], const uint32_t* e, size_t n) {
    uint32_t s[24];
    size_t used = 0;
    memset(s, 0, sizeof s);
    for (size_t i = 0; i < n; ++i) {
        s[used++] = e[i];
        if (used == 16) p2_mix(h, s), used = 0;
    }
    if (used != 0 || n == 0) {
        for (size_t i = used; i < 16; ++i) s[i] = 0;
        p2_mix(h, s);
    }
    memcpy(out, s, 32);
}
/* WriteIOP + Poseidon2Rng */
typedef struct {
    const p2params* h;
    uint32_t cells[24];
    unsigned pool_used;
    uint32_t* seal;
    size_t words, cap;
    int overflow;
} iop;
static void iop_write(iop* t, const uint32_t* w, size_t n) {
    if (t->words + n > t->cap) { t->overflow = 1; return; }
    memcpy(t->seal + t->words, w, 4 * n);
    t->words += n;
}
static void iop_commit(iop* t, const uint32_t dg[8]) {
    if (t->pool_used) p2_mix(t->h, t->cells), t->pool_used = 0;
    for (int i = 0; i < 8; ++i) t->cells[i] = fadd(t->cells[i], dg[i]);
    p2_mix(t->h, t->cells);
}
static uint32_t iop_elem(iop* t) {
    if (t->pool_used == 16) p2_mix(t->h, t->cells), t->pool_used = 0;
    return t->cells[t->pool_used++];
}
static ext iop_ext(iop* t) { ext r; for (int k = 0; k < 4; ++k) r.c[k] = iop_elem(t); return r; }
static uint32_t iop_bits(iop* t, unsigned bits) {
    uint32_t v = fdec(iop_elem(t));
    for (int i = 0; i < 3; ++i) {
        uint32_t nv = fdec(iop_elem(t));
        if (v == 0) v = nv;
    }
    return bits >= 32 ? v : v & ((1u << bits) - 1u);
}

/* ---------------------------------------------------------------- the driver's state ---------------------------------------- */
typedef struct {
    size_t rows, cols, top; /* top = nodes of the layer written to the seal */
    unsigned layers, depth;
    bx_buf nodes; /* 2 * rows digests; node i at [8i, 8i + 8), layer of s nodes at [s, 2s) */
    uint32_t root[8];
} tree;
typedef struct {
    uint32_t width;
    bx_buf coeffs, evaluated, combo_ids;
    tree tr;
    uint32_t* n_backs;                 /* per column */
    uint32_t (*backs)[BX_MAX_TAPS];    /* per column */
    uint32_t* combo;                   /* per column */
} group;
typedef struct {
    size_t size;
    bx_buf evaluated, out_coeffs;
    tree tr;
} fri_round;
typedef struct ph_prover {
    bx_ctx* c;
    int own_ctx;
    const bx_circuit_ops* circ;
    void* circ_state;
    bx_segment_params shape;
    size_t N;
    unsigned flags;
    p2params h;
    group g[4];
    size_t n_trace_combos;
    uint32_t combo_n[BX_MAX_COMBOS];
    uint32_t combo_backs[BX_MAX_COMBOS][BX_MAX_TAPS];
    size_t total_taps, tap_first[5], n_div;
    uint32_t n_globals;
    bx_buf code_w, combos, final_poly, which, xs, evals, rems, final_coeffs, seg_dev, qout, positions, tap_ptrs, tap_flags;
    fri_round rounds[MAX_ROUNDS];
    size_t n_rounds, final_size, seal_cap;
    size_t qwords;
    int big;      /* the big buffers are allocated */
    size_t calls; /* entry-point calls of the last proof */
    char err[512];
} ph_prover;

static const char* fail(ph_prover* p, const char* m) {
    if (m != p->err) snprintf(p->err, sizeof p->err, "%s", m);
    return p->err;
}
#define PH(expr)                              \
    do {                                      \
        const char* m_ = (expr);              \
        p->calls += 1;                        \
        if (m_) return fail(p, m_);           \
    } while (0)
static bx_buf slice(bx_buf b, size_t off, size_t len) { /* Buffer::slice */
    bx_buf r;
    r.dptr = (uint32_t*)b.dptr + off;
    r.len = len;
    return r;
}
static unsigned ilog2(size_t v) { unsigned r = 0; while (v > 1) v >>= 1, ++r; return r; }
static const char* tree_init(ph_prover* p, tree* t, size_t rows, size_t cols) {
    unsigned top = 0;
    t->rows = rows, t->cols = cols, t->layers = ilog2(rows);
    for (unsigned i = 1; i < t->layers; ++i) { /* the largest layer of at most QUERIES nodes, below the leaves */
        if ((1u << i) > BX_QUERIES) break;
        top = i;
    }
    t->top = (size_t)1 << top;
    t->depth = t->layers - top;
    (void)p;
    return NULL;
}
/* the big per-proof buffers (what risc0-zkp's prover allocates inside a proof) */
static const char* big_alloc(ph_prover* p) {
    bx_ctx* c = p->c;
    const size_t N = p->N, D = 4 * N;
    for (int g = 0; g < 4; ++g) {
        group* G = &p->g[g];
        PH(bx_alloc(c, (size_t)G->width * N, &G->coeffs));
        PH(bx_alloc(c, (size_t)G->width * D, &G->evaluated));
        PH(bx_alloc(c, 16 * G->tr.rows, &G->tr.nodes));
    }
    PH(bx_alloc(c, (size_t)p->shape.w_code * N, &p->code_w));
    PH(bx_alloc(c, 4 * N, &p->final_poly));
    for (size_t r = 0; r < p->n_rounds; ++r) {
        fri_round* R = &p->rounds[r];
        PH(bx_alloc(c, 16 * R->size, &R->evaluated));
        PH(bx_alloc(c, 4 * R->size / BX_FRI_FOLD, &R->out_coeffs));
        PH(bx_alloc(c, 16 * R->tr.rows, &R->tr.nodes));
    }
    PH(bx_alloc(c, 4 * p->final_size, &p->final_coeffs));
    PH(bx_alloc(c, p->qwords * BX_QUERIES, &p->qout));
    p->big = 1;
    return NULL;
}
static void big_release(ph_prover* p) {
    bx_buf* bufs[4 * 3 + 4 + 3 * MAX_ROUNDS];
    size_t n = 0;
    for (int g = 0; g < 4; ++g) bufs[n++] = &p->g[g].coeffs, bufs[n++] = &p->g[g].evaluated, bufs[n++] = &p->g[g].tr.nodes;
    bufs[n++] = &p->code_w, bufs[n++] = &p->final_poly, bufs[n++] = &p->final_coeffs, bufs[n++] = &p->qout;
    for (size_t r = 0; r < p->n_rounds; ++r) bufs[n++] = &p->rounds[r].evaluated, bufs[n++] = &p->rounds[r].out_coeffs, bufs[n++] = &p->rounds[r].tr.nodes;
    for (size_t i = 0; i < n; ++i) {
        if (bufs[i]->dptr) (void)bx_release(p->c, *bufs[i]), p->calls += 1;
        bufs[i]->dptr = NULL;
    }
    if (p->combos.dptr) (void)bx_release(p->c, p->combos), p->calls += 1;
    p->combos.dptr = NULL;
    p->big = 0;
}

const char* ph_error(const ph_prover* p) { return p ? p->err : "null"; }
size_t ph_seal_words(const ph_prover* p) { return p ? p->seal_cap : 0; }
size_t ph_last_calls(const ph_prover* p) { return p ? p->calls : 0; }

const char* ph_destroy(ph_prover* p) {
    if (!p) return NULL;
    if (p->c) (void)bx_sync(p->c);
    if (p->circ && p->circ_state && p->circ->destroy) p->circ->destroy(p->circ->user, p->circ_state);
    if (p->c) big_release(p);
    bx_buf* bufs[] = {&p->which, &p->xs, &p->evals, &p->rems, &p->seg_dev, &p->positions, &p->tap_ptrs, &p->tap_flags};
    for (size_t i = 0; i < sizeof bufs / sizeof bufs[0]; ++i)
        if (bufs[i]->dptr) (void)bx_release(p->c, *bufs[i]);
    for (int g = 0; g < 4; ++g) {
        group* G = &p->g[g];
        if (G->combo_ids.dptr) (void)bx_release(p->c, G->combo_ids);
        free(G->n_backs), free(G->backs), free(G->combo);
    }
    if (p->own_ctx && p->c) (void)bx_free(p->c);
    free(p);
    return NULL;
}

/* Everything a proof needs is allocated here (like bx_prover_create: nothing per proof), so ph_prove times the calls alone.
 * ctx NULL: the driver opens its own on `device`. */
const char* ph_create(bx_ctx* ctx, int device, const bx_segment_params* shape, unsigned flags, ph_prover** out) {
    static char msg[512];
    ph_prover* p = (ph_prover*)calloc(1, sizeof *p);
    if (!p) return "ph_create: out of memory";
    if (!ctx) {
        const char* e = bx_init(device, &ctx);
        if (e) { snprintf(msg, sizeof msg, "%s", e); free(p); return msg; }
        p->own_ctx = 1;
    }
    p->c = ctx;
    p->shape = *shape;
    p->flags = flags;
    p->circ = bx_synthetic_circuit();
    const char* r = NULL;
#define CR(expr)                                       \
    do {                                               \
        const char* m_ = (expr);                       \
        if (m_) { r = m_; goto bad; }                  \
    } while (0)
    {
        uint32_t rc[213], dg[24];
        CR(bx_poseidon2_default_params(rc, dg));
        for (int i = 0; i < 213; ++i) p->h.rc[i] = fenc(rc[i]);
        for (int i = 0; i < 24; ++i) p->h.diag[i] = fenc(dg[i]);
    }
    if (p->circ->normalize) CR(p->circ->normalize(p->circ->user, &p->shape));
    if (shape->po2 < 9 || shape->po2 > 24) CR("ph_create: po2 must be in [9, 24]");
    p->N = (size_t)1 << shape->po2;
    if (p->N < ((size_t)1 << 15)) p->flags &= ~(PH_EXT_COEFFS_BITREV | PH_EXT_EVAL_PTRS); /* those entry points start at 2^15 */
    const size_t N = p->N, D = 4 * N;
    const uint32_t widths[4] = {p->shape.w_code, p->shape.w_data, p->shape.w_accum, BX_CHECK_SIZE};
    for (int g = 0; g < 4; ++g) {
        group* G = &p->g[g];
        G->width = widths[g];
        CR(tree_init(p, &G->tr, D, G->width));
        G->n_backs = (uint32_t*)calloc(G->width, 4);
        G->backs = (uint32_t(*)[BX_MAX_TAPS])calloc(G->width, sizeof *G->backs);
        G->combo = (uint32_t*)calloc(G->width, 4);
        if (!G->n_backs || !G->backs || !G->combo) CR("ph_create: out of memory");
        for (uint32_t col = 0; col < G->width; ++col) {
            if (g == 3) { G->n_backs[col] = 1; continue; }
            const uint32_t k = p->circ->taps(p->circ->user, &p->shape, g, col, G->backs[col]);
            if (k < 1 || k > BX_MAX_TAPS || G->backs[col][0] != 0) CR("ph_create: bad tap set");
            G->n_backs[col] = k;
            size_t id = 0; /* TapSet: columns with the same set share a combo, numbered in order of first appearance */
            for (; id < p->n_trace_combos; ++id)
                if (p->combo_n[id] == k && memcmp(p->combo_backs[id], G->backs[col], 4 * k) == 0) break;
            if (id == p->n_trace_combos) {
                if (id + 2 > BX_MAX_COMBOS) CR("ph_create: too many tap sets");
                p->combo_n[id] = k;
                memcpy(p->combo_backs[id], G->backs[col], 4 * k);
                p->n_trace_combos += 1;
            }
            G->combo[col] = (uint32_t)id;
        }
    }
    const size_t n_combos = p->n_trace_combos + 1;
    for (int g = 0; g < 4; ++g) {
        group* G = &p->g[g];
        if (g == 3)
            for (uint32_t col = 0; col < G->width; ++col) G->combo[col] = (uint32_t)(n_combos - 1);
        CR(bx_alloc(ctx, G->width, &G->combo_ids));
        CR(bx_h2d(ctx, G->combo_ids, G->combo, G->width));
        for (uint32_t col = 0; col < G->width; ++col) p->total_taps += G->n_backs[col];
        p->tap_first[g + 1] = p->total_taps;
    }
    p->n_div = 1;
    for (size_t id = 0; id < p->n_trace_combos; ++id) p->n_div += p->combo_n[id];
    if (p->circ->create) CR(p->circ->create(p->circ->user, ctx, &p->shape, &p->circ_state));
    p->n_globals = p->circ->n_globals ? p->circ->n_globals(p->circ->user, &p->shape) : 0;
    /* `combos` is allocated per proof (alloc_extelem_zeroed in Prover::finalize): mix_poly_coeffs accumulates into it */
    CR(bx_alloc(ctx, p->total_taps, &p->which));
    CR(bx_alloc(ctx, 4 * p->total_taps, &p->xs));
    CR(bx_alloc(ctx, 4 * p->total_taps, &p->evals));
    CR(bx_alloc(ctx, 4 * p->n_div, &p->rems));
    CR(bx_alloc(ctx, (BX_SEGMENT_WIRE_BYTES + 3) / 4, &p->seg_dev));
    {   /* `which` of batch_evaluate_any: the polynomial (column) of every tap evaluation, group by group */
        uint32_t* w = (uint32_t*)malloc(4 * p->total_taps);
        size_t e = 0;
        if (!w) CR("ph_create: out of memory");
        for (int g = 0; g < 4; ++g)
            for (uint32_t col = 0; col < p->g[g].width; ++col)
                for (uint32_t t = 0; t < p->g[g].n_backs[col]; ++t, ++e) w[e] = col;
        const char* m = bx_h2d(ctx, p->which, w, p->total_taps);
        free(w);
        CR(m);
    }
    size_t size = N, qwords = 0;
    while (size > BX_FRI_MIN_DEGREE) {
        fri_round* R = &p->rounds[p->n_rounds++];
        R->size = size;
        CR(tree_init(p, &R->tr, 4 * size / BX_FRI_FOLD, 4 * BX_FRI_FOLD));
        qwords += R->tr.cols + 8 * R->tr.depth;
        size /= BX_FRI_FOLD;
    }
    p->final_size = size;
    for (int g = 0; g < 4; ++g) qwords += p->g[g].tr.cols + 8 * p->g[g].tr.depth;
    p->qwords = qwords;
    CR(bx_alloc(ctx, BX_QUERIES * MAX_TREES, &p->positions));
    p->seal_cap = BX_SEAL_HEADER_WORDS + p->n_globals + 4 * p->total_taps + 4 * size + BX_QUERIES * qwords;
    for (int g = 0; g < 4; ++g) p->seal_cap += 8 * p->g[g].tr.top;
    for (size_t r = 0; r < p->n_rounds; ++r) p->seal_cap += 8 * p->rounds[r].tr.top;
    if (p->flags & PH_ALLOC_PER_PROOF) p->flags &= ~PH_EXT_EVAL_PTRS; /* that entry point takes column ADDRESSES, fixed per shape */
    if (!(p->flags & PH_ALLOC_PER_PROOF)) CR(big_alloc(p));
    if (p->flags & PH_EXT_EVAL_PTRS) { /* bx_batch_evaluate_ptrs: the device address and storage order of every tap's column */
        uint32_t* ptrs = (uint32_t*)malloc(8 * p->total_taps);
        uint32_t* fl = (uint32_t*)malloc(4 * p->total_taps);
        size_t e = 0;
        const char* m = (!ptrs || !fl) ? "ph_create: out of memory" : NULL;
        for (int g = 0; g < 4 && !m; ++g)
            for (uint32_t col = 0; col < p->g[g].width; ++col)
                for (uint32_t t = 0; t < p->g[g].n_backs[col]; ++t, ++e) {
                    const unsigned long long a = (unsigned long long)(uintptr_t)((uint32_t*)p->g[g].coeffs.dptr + (size_t)col * N);
                    ptrs[2 * e] = (uint32_t)a, ptrs[2 * e + 1] = (uint32_t)(a >> 32);
                    fl[e] = ((p->flags & PH_EXT_COEFFS_BITREV) && g < 3) ? 1u : 0u;
                }
        if (!m) m = bx_alloc(ctx, 2 * p->total_taps, &p->tap_ptrs);
        if (!m) m = bx_alloc(ctx, p->total_taps, &p->tap_flags);
        if (!m) m = bx_h2d(ctx, p->tap_ptrs, ptrs, 2 * p->total_taps);
        if (!m) m = bx_h2d(ctx, p->tap_flags, fl, p->total_taps);
        free(ptrs), free(fl);
        CR(m);
    }
    CR(bx_sync(ctx));
    *out = p;
    return NULL;
bad:
    snprintf(msg, sizeof msg, "%s", r);
    ph_destroy(p);
    return msg;
#undef CR
}

/* MerkleTreeProver::new + commit: leaves, every layer with its own hash_fold, then the root and the top layer come to the host */
static const char* tree_commit(ph_prover* p, tree* t, bx_buf matrix, iop* T) {
    bx_ctx* c = p->c;
    if (p->flags & PH_EXT_MERKLE_BUILD) {
        PH(bx_merkle_build(c, t->nodes, matrix, t->rows));
    } else {
        PH(bx_hash_rows(c, slice(t->nodes, 8 * t->rows, 8 * t->rows), matrix));
        for (size_t size = t->rows; size > 1; size /= 2) PH(bx_hash_fold(c, t->nodes, size, size / 2));
    }
    uint32_t top[8 * 64];
    PH(bx_d2h(c, top, slice(t->nodes, 8 * t->top, 8 * t->top), 8 * t->top));
    PH(bx_d2h(c, t->root, slice(t->nodes, 8, 8), 8));
    iop_write(T, top, 8 * t->top);
    iop_commit(T, t->root);
    if (getenv("PH_DEBUG")) fprintf(stderr, "tree rows %zu cols %zu root %08x %08x\n", t->rows, t->cols, t->root[0], t->root[1]);
    return NULL;
}
/* Prover::commit_group: interpolate, zk_shift, PolyGroup::new (expand + evaluate, coefficients to natural order, Merkle) */
static const char* commit_group(ph_prover* p, group* G, int is_trace, iop* T) {
    bx_ctx* c = p->c;
    if (p->flags & PH_EXT_INTERPOLATE_ZK) {
        PH(bx_batch_interpolate_zk(c, G->coeffs, G->width));
    } else {
        PH(bx_batch_interpolate_ntt(c, G->coeffs, G->width));
        PH(bx_zk_shift(c, G->coeffs, G->width));
    }
    PH(bx_batch_expand_into_evaluate_ntt(c, G->evaluated, G->coeffs, G->width, 2));
    if (!(is_trace && (p->flags & PH_EXT_COEFFS_BITREV))) PH(bx_batch_bit_reverse(c, G->coeffs, G->width));
    return tree_commit(p, &G->tr, G->evaluated, T);
}

/* One proof of the stand-in segment (index 0, the prover's po2, `seed`, no payload): the seal bx_prove_segment(prover, seed) writes. */
const char* ph_prove(ph_prover* p, uint64_t seed, uint32_t* seal_out, size_t seal_cap, size_t* seal_words, double* wall_ms) {
    if (!p) return "ph_prove: null";
    bx_ctx* c = p->c;
    const bx_circuit_ops* circ = p->circ;
    const size_t N = p->N, D = 4 * N;
    const uint32_t po2 = p->shape.po2;
    const size_t n_trace = p->n_trace_combos, n_combos = n_trace + 1;
    struct timespec t0, t1;
    iop T;
    memset(&T, 0, sizeof T);
    T.h = &p->h, T.seal = seal_out, T.cap = seal_cap;
    p->calls = 0;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    if (p->flags & PH_ALLOC_PER_PROOF) {
        if (p->big) big_release(p); /* an earlier proof failed half-way */
        const char* e = big_alloc(p);
        if (e) { big_release(p); return e; }
    }

    /* header */
    {
        uint32_t hdr[BX_SEAL_HEADER_WORDS] = {po2, p->shape.w_code, p->shape.w_data, p->shape.w_accum, p->shape.cons_terms, p->shape.cons_degree};
        uint32_t enc[BX_SEAL_HEADER_WORDS], dg[8];
        for (int i = 0; i < BX_SEAL_HEADER_WORDS; ++i) enc[i] = fenc(hdr[i]);
        iop_write(&T, hdr, BX_SEAL_HEADER_WORDS);
        p2_hash(&p->h, dg, enc, BX_SEAL_HEADER_WORDS);
        iop_commit(&T, dg);
    }
    /* the segment's bytes -> witness (code group: a function of the shape; data group: from the segment) */
    uint32_t globals[BX_MAX_GLOBALS];
    memset(globals, 0, sizeof globals);
    {
        uint32_t wire[(BX_SEGMENT_WIRE_BYTES + 3) / 4];
        memset(wire, 0, sizeof wire);
        bx_segment_encode(0, po2, seed, (uint8_t*)wire);
        PH(bx_h2d(c, p->seg_dev, wire, p->seg_dev.len));
        PH(circ->code_group(circ->user, p->circ_state, c, p->code_w));
        PH(bx_eltwise_copy_elem(c, p->g[0].coeffs, p->code_w));
        PH(circ->witgen(circ->user, p->circ_state, c, p->code_w, p->g[1].coeffs, (const uint8_t*)wire, BX_SEGMENT_WIRE_BYTES, p->seg_dev, globals));
    }
    if (p->n_globals) {
        uint32_t dg[8];
        for (uint32_t i = 0; i < p->n_globals; ++i)
            if (globals[i] >= P) return fail(p, "ph_prove: non-canonical public word");
        iop_write(&T, globals, p->n_globals);
        p2_hash(&p->h, dg, globals, p->n_globals);
        iop_commit(&T, dg);
    }
    { const char* e; if ((e = commit_group(p, &p->g[0], 1, &T)) != NULL) return e; }
    { const char* e; if ((e = commit_group(p, &p->g[1], 1, &T)) != NULL) return e; }
    const ext beta = iop_ext(&T);
    PH(circ->accumulate(circ->user, p->circ_state, c, p->g[2].coeffs, beta.c));
    { const char* e; if ((e = commit_group(p, &p->g[2], 1, &T)) != NULL) return e; }
    /* eval_check and the check group: 4 ext planes over the 4N domain -> 16 columns of size N */
    group* CK = &p->g[3];
    {
        const ext poly_mix = iop_ext(&T);
        PH(circ->eval_check(circ->user, p->circ_state, c, CK->coeffs, p->g[0].evaluated, p->g[1].evaluated, p->g[2].evaluated, poly_mix.c, beta.c,
                            globals));
        PH(bx_batch_interpolate_ntt(c, CK->coeffs, 4));
        PH(bx_zk_shift(c, CK->coeffs, BX_CHECK_SIZE));
        PH(bx_batch_expand_into_evaluate_ntt(c, CK->evaluated, CK->coeffs, BX_CHECK_SIZE, 2));
        PH(bx_batch_bit_reverse(c, CK->coeffs, BX_CHECK_SIZE));
        const char* e = tree_commit(p, &CK->tr, CK->evaluated, &T);
        if (e) return e;
    }
    /* DEEP: every tap evaluated at Z * w_N^-back (check columns at Z^4 / 3) */
    const ext Z = iop_ext(&T);
    const uint32_t back_one = finv(fpow(fenc(137u), (uint64_t)1 << (27 - po2)));
    const ext Z4 = xscale(xpow(Z, 4), finv(fenc(3u)));
    ext pts[BX_MAX_COMBOS][BX_MAX_TAPS], interp[BX_MAX_COMBOS][BX_MAX_TAPS * BX_MAX_TAPS];
    for (size_t id = 0; id < n_trace; ++id) {
        const size_t k = p->combo_n[id];
        for (size_t t = 0; t < k; ++t) pts[id][t] = xscale(Z, fpow(back_one, p->combo_backs[id][t]));
        for (size_t i = 0; i < k; ++i) { /* coefficient t of the Lagrange basis polynomial L_i over the combo's points */
            ext poly[BX_MAX_TAPS + 1], denom = xone();
            size_t deg = 0;
            poly[0] = xone();
            for (size_t j = 0; j < k; ++j) {
                if (j == i) continue;
                poly[deg + 1] = xzero();
                for (size_t d = deg + 1; d-- > 0;) {
                    poly[d + 1] = xadd(poly[d + 1], poly[d]);
                    poly[d] = xsub(xzero(), xmul(poly[d], pts[id][j]));
                }
                /* poly(x) *= (x - x_j), in place from the high end (coefficients are stored low first) */
                deg += 1;
                denom = xmul(denom, xsub(pts[id][i], pts[id][j]));
            }
            const ext inv = xinv(denom);
            for (size_t t = 0; t < k; ++t) interp[id][t * k + i] = xmul(poly[t], inv);
        }
    }
    const size_t ne_all = p->total_taps;
    uint32_t* coeff_u = (uint32_t*)malloc(16 * ne_all);
    uint32_t* hostbuf = (uint32_t*)malloc(16 * ne_all);
    if (!coeff_u || !hostbuf) { free(coeff_u), free(hostbuf); return fail(p, "ph_prove: out of memory"); }
    const char* err = NULL;
#define PF(expr)                                        \
    do {                                                \
        const char* m_ = (expr);                        \
        p->calls += 1;                                  \
        if (m_) { err = fail(p, m_); goto done; }       \
    } while (0)
    {
        size_t e = 0;
        for (int g = 0; g < 4; ++g)
            for (uint32_t col = 0; col < p->g[g].width; ++col)
                for (uint32_t t = 0; t < p->g[g].n_backs[col]; ++t, ++e)
                    memcpy(hostbuf + 4 * e, g == 3 ? Z4.c : pts[p->g[g].combo[col]][t].c, 16);
        PF(bx_h2d(c, p->xs, hostbuf, 4 * ne_all));
        if (p->flags & PH_EXT_EVAL_PTRS) {
            PF(bx_batch_evaluate_ptrs(c, p->tap_ptrs, p->tap_flags, N, p->xs, p->evals));
        } else {
            for (int g = 0; g < 4; ++g) {
                const size_t o = p->tap_first[g], ne = p->tap_first[g + 1] - o;
                if ((p->flags & PH_EXT_COEFFS_BITREV) && g < 3)
                    PF(bx_batch_evaluate_any_bitrev(c, p->g[g].coeffs, p->g[g].width, slice(p->which, o, ne), slice(p->xs, 4 * o, 4 * ne),
                                                    slice(p->evals, 4 * o, 4 * ne)));
                else
                    PF(bx_batch_evaluate_any(c, p->g[g].coeffs, p->g[g].width, slice(p->which, o, ne), slice(p->xs, 4 * o, 4 * ne),
                                             slice(p->evals, 4 * o, 4 * ne)));
            }
        }
        PF(bx_d2h(c, hostbuf, p->evals, 4 * ne_all));
        /* coeff_u: per column the coefficients of the polynomial through its tap values (a single tap: the value itself) */
        e = 0;
        for (int g = 0; g < 4; ++g)
            for (uint32_t col = 0; col < p->g[g].width; ++col) {
                const size_t k = p->g[g].n_backs[col];
                if (k == 1) {
                    memcpy(coeff_u + 4 * e, hostbuf + 4 * e, 16);
                } else {
                    const ext* M = interp[p->g[g].combo[col]];
                    for (size_t t = 0; t < k; ++t) {
                        ext ct = xzero();
                        for (size_t i = 0; i < k; ++i) {
                            ext y;
                            memcpy(y.c, hostbuf + 4 * (e + i), 16);
                            ct = xadd(ct, xmul(M[t * k + i], y));
                        }
                        memcpy(coeff_u + 4 * (e + t), ct.c, 16);
                    }
                }
                e += k;
            }
        uint32_t dg[8];
        iop_write(&T, coeff_u, 4 * ne_all);
        p2_hash(&p->h, dg, coeff_u, 4 * ne_all);
        iop_commit(&T, dg);
    }
    /* DEEP: mix every column into its combo, subtract the mixed u polynomials, divide by every tap point */
    {
        const ext mix = iop_ext(&T);
        ext cur = xone(), combo_u[BX_MAX_COMBOS][BX_MAX_TAPS];
        size_t u = 0;
        for (size_t id = 0; id < n_combos; ++id)
            for (size_t t = 0; t < BX_MAX_TAPS; ++t) combo_u[id][t] = xzero();
        if (p->combos.dptr) PF(bx_release(c, p->combos));
        p->combos.dptr = NULL;
        PF(bx_alloc_zeroed(c, n_combos * 4 * N, &p->combos));
        for (int g = 0; g < 4; ++g) {
            group* G = &p->g[g];
            PF(bx_mix_poly_coeffs(c, p->combos, cur.c, mix.c, G->coeffs, G->combo_ids, G->width, N));
            for (uint32_t col = 0; col < G->width; ++col) {
                for (uint32_t t = 0; t < G->n_backs[col]; ++t, u += 4) {
                    ext cu;
                    memcpy(cu.c, coeff_u + u, 16);
                    combo_u[G->combo[col]][t] = xadd(combo_u[G->combo[col]][t], xmul(cur, cu));
                }
                cur = xmul(cur, mix);
            }
        }
        if (p->flags & PH_EXT_COEFFS_BITREV) PF(bx_batch_bit_reverse_ext(c, slice(p->combos, 0, 4 * N * n_trace), n_trace));
        for (size_t id = 0; id < n_combos; ++id) { /* Buffer::view_mut over the low coefficients of combo id */
            const size_t k = id < n_trace ? p->combo_n[id] : 1;
            uint32_t low[4 * BX_MAX_TAPS];
            PF(bx_d2h(c, low, slice(p->combos, id * 4 * N, 4 * k), 4 * k));
            for (size_t t = 0; t < k; ++t)
                for (int w = 0; w < 4; ++w) low[4 * t + w] = fsub(low[4 * t + w], combo_u[id][t].c[w]);
            PF(bx_h2d(c, slice(p->combos, id * 4 * N, 4 * k), low, 4 * k));
        }
        if (p->flags & PH_EXT_DIVIDE_BATCH) {
            size_t d = 0;
            for (size_t r = 0;; ++r) {
                uint32_t which[BX_MAX_COMBOS], zs[4 * BX_MAX_COMBOS];
                size_t cnt = 0;
                for (size_t id = 0; id < n_trace; ++id)
                    if (p->combo_n[id] > r) which[cnt] = (uint32_t)id, memcpy(zs + 4 * cnt++, pts[id][r].c, 16);
                if (r == 0) which[cnt] = (uint32_t)n_trace, memcpy(zs + 4 * cnt++, Z4.c, 16);
                if (!cnt) break;
                PF(bx_poly_divide_batch_indexed(c, p->combos, n_combos, cnt, which, zs, slice(p->rems, 4 * d, 4 * cnt)));
                d += cnt;
            }
        } else {
            size_t d = 0;
            for (size_t id = 0; id < n_trace; ++id)
                for (size_t t = 0; t < p->combo_n[id]; ++t, ++d)
                    PF(bx_poly_divide(c, slice(p->combos, id * 4 * N, 4 * N), pts[id][t].c, slice(p->rems, 4 * d, 4)));
            PF(bx_poly_divide(c, slice(p->combos, n_trace * 4 * N, 4 * N), Z4.c, slice(p->rems, 4 * d, 4)));
        }
        PF(bx_d2h(c, hostbuf, p->rems, 4 * p->n_div));
        for (size_t i = 0; i < 4 * p->n_div; ++i)
            if (hostbuf[i] != 0) {
                if (getenv("PH_DEBUG")) {
                    for (size_t d2 = 0; d2 < p->n_div; ++d2)
                        fprintf(stderr, "rem[%zu] = %u %u %u %u\n", d2, hostbuf[4 * d2], hostbuf[4 * d2 + 1], hostbuf[4 * d2 + 2], hostbuf[4 * d2 + 3]);
                    for (size_t id = 0; id < n_trace; ++id) fprintf(stderr, "combo %zu: %u taps\n", id, p->combo_n[id]);
                }
                err = fail(p, "ph_prove: DEEP quotient has a non-zero remainder");
                goto done;
            }
        PF(bx_eltwise_sum_extelem(c, p->final_poly, p->combos));
        PF(bx_batch_bit_reverse(c, p->final_poly, 4));
    }
    /* fri_prove */
    {
        bx_buf coeffs = p->final_poly;
        for (size_t r = 0; r < p->n_rounds; ++r) {
            fri_round* R = &p->rounds[r];
            PF(bx_batch_expand_into_evaluate_ntt(c, R->evaluated, coeffs, 4, 2));
            if ((err = tree_commit(p, &R->tr, R->evaluated, &T)) != NULL) goto done;
            const ext fold_mix = iop_ext(&T);
            PF(bx_fri_fold(c, R->out_coeffs, coeffs, fold_mix.c));
            coeffs = R->out_coeffs;
        }
        uint32_t fc[4 * BX_FRI_MIN_DEGREE], dg[8];
        PF(bx_eltwise_copy_elem(c, p->final_coeffs, coeffs));
        PF(bx_batch_bit_reverse(c, p->final_coeffs, 4));
        PF(bx_d2h(c, fc, p->final_coeffs, 4 * p->final_size));
        iop_write(&T, fc, 4 * p->final_size);
        p2_hash(&p->h, dg, fc, 4 * p->final_size);
        iop_commit(&T, dg);
    }
    /* queries: MerkleTreeProver::prove per tree and position — the opened row, then the path up to the top layer */
    {
        const unsigned bits = ilog2(D);
        const size_t n_trees = 4 + p->n_rounds;
        uint32_t pos[MAX_TREES][BX_QUERIES];
        size_t off[MAX_TREES + 1], qw[MAX_TREES];
        off[0] = 0;
        for (int q = 0; q < BX_QUERIES; ++q) pos[0][q] = iop_bits(&T, bits) % (uint32_t)D;
        for (size_t t = 0; t < n_trees; ++t) {
            tree* tr = t < 4 ? &p->g[t].tr : &p->rounds[t - 4].tr;
            if (t) for (int q = 0; q < BX_QUERIES; ++q) pos[t][q] = pos[t - 1][q] % (uint32_t)tr->rows;
            qw[t] = tr->cols + 8 * tr->depth;
            off[t + 1] = off[t] + qw[t] * BX_QUERIES;
        }
        uint32_t* qhost = (uint32_t*)malloc(4 * off[n_trees]);
        if (!qhost) { err = fail(p, "ph_prove: out of memory"); goto done; }
        if (p->flags & PH_EXT_QUERY_GATHER) PF(bx_h2d(c, p->positions, &pos[0][0], n_trees * BX_QUERIES));
        for (size_t t = 0; t < n_trees && !err; ++t) {
            tree* tr = t < 4 ? &p->g[t].tr : &p->rounds[t - 4].tr;
            bx_buf matrix = t < 4 ? p->g[t].evaluated : p->rounds[t - 4].evaluated;
            bx_buf dst = slice(p->qout, off[t], qw[t] * BX_QUERIES);
            const char* m = NULL;
            if (p->flags & PH_EXT_QUERY_GATHER) {
                m = bx_merkle_query_gather(c, dst, matrix, tr->nodes, tr->rows, tr->cols, slice(p->positions, t * BX_QUERIES, BX_QUERIES), BX_QUERIES,
                                           tr->top);
                p->calls += 1;
            } else {
                for (int q = 0; q < BX_QUERIES && !m; ++q) {
                    bx_buf o = slice(dst, (size_t)q * qw[t], qw[t]);
                    m = bx_gather_sample(c, o, matrix, pos[t][q], tr->cols, tr->rows); /* row pos of the column-major matrix */
                    p->calls += 1;
                    for (unsigned lv = 0; lv < tr->depth && !m; ++lv) { /* the sibling on every layer below the top one */
                        const size_t sib = ((pos[t][q] + tr->rows) >> lv) ^ 1u;
                        m = bx_gather_sample(c, slice(o, tr->cols + 8 * lv, 8), tr->nodes, 8 * sib, 8, 1);
                        p->calls += 1;
                    }
                }
            }
            if (!m) m = bx_d2h(c, qhost + off[t], dst, qw[t] * BX_QUERIES), p->calls += 1;
            if (m) err = fail(p, m);
        }
        if (!err)
            for (int q = 0; q < BX_QUERIES; ++q)
                for (size_t t = 0; t < n_trees; ++t) iop_write(&T, qhost + off[t] + (size_t)q * qw[t], qw[t]);
        free(qhost);
    }
done:
    free(coeff_u), free(hostbuf);
    if (p->flags & PH_ALLOC_PER_PROOF) big_release(p);
    if (err) return err;
    clock_gettime(CLOCK_MONOTONIC, &t1);
    if (wall_ms) *wall_ms = (t1.tv_sec - t0.tv_sec) * 1e3 + (t1.tv_nsec - t0.tv_nsec) * 1e-6;
    if (seal_words) *seal_words = T.words;
    if (T.overflow) return fail(p, "ph_prove: seal buffer too small");
    return NULL;
#undef PF
}
