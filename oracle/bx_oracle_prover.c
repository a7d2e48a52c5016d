/*
 * bx_oracle_prover.c — CPU ORACLE of the segment-prover pipeline (test infrastructure only).
 *
 * Sequential restatement, on host arrays and with the oracle's own HAL functions (bx_oracle.c), of
 *   risc0_zkp::prove::Prover::{commit_group, finalize}, prove::fri::fri_prove, prove::merkle::MerkleTreeProver,
 *   prove::write_iop::WriteIOP and core::hash::poseidon2::Poseidon2Rng            [EXT: risc0-zkp 3.0.3]
 * as driven by ProverServer::prove_segment (bento/crates/workflow/src/tasks/prove.rs:41-49).  The circuit is the
 * synthetic AIR specified in include/bx_prover.h ("The synthetic circuit": witness generation with derived columns and
 * scatter-placed permuted copies, grand-product accumulators, a constraint polynomial divided by the vanishing
 * polynomial); everything else keeps upstream's order and constants.  Written straight-line (one query at a time, one layer at a time) so that it
 * shares no structure with the product's batched device code.  Seal parity vs the Rust prover: UNPINNED (no
 * circuit, no vectors in the reference); this oracle pins the product against an independent implementation.
 */
#include <stdlib.h>
#include <string.h>

#include "bx_oracle.h"

#define QUERIES 50
#define INV_RATE 4
#define FRI_FOLD 16
#define FRI_MIN_DEGREE 256
#define CHECK_SIZE 16
#define GOLDEN 0x9E3779B97F4A7C15ull
#define CODE_SEED 0x434F4E54524F4C21ull /* "CONTROL!": the code group is public and depends on the shape alone (include/bx_prover.h) */

typedef struct { uint32_t c[4]; } e4;

/* ---- transcript: WriteIOP + Poseidon2Rng ---- */
typedef struct {
    uint32_t* seal;
    size_t len, cap;
    uint32_t cells[24];
    unsigned pool_used;
} iop_t;
static void iop_write(iop_t* io, const uint32_t* w, size_t n) {
    if (io->len + n > io->cap) {
        io->cap = (io->len + n) * 2;
        io->seal = (uint32_t*)realloc(io->seal, io->cap * 4);
    }
    memcpy(io->seal + io->len, w, n * 4);
    io->len += n;
}
/* [EXT] Poseidon2Rng::mix */
static void iop_commit(iop_t* io, const uint32_t d[8]) {
    if (io->pool_used != 0) {
        bxo_poseidon2_mix(io->cells);
        io->pool_used = 0;
    }
    for (int i = 0; i < 8; i++) io->cells[i] = bxo_fp_add(io->cells[i], d[i]);
    bxo_poseidon2_mix(io->cells);
}
/* [EXT] Poseidon2Rng::random_elem */
static uint32_t iop_random_elem(iop_t* io) {
    if (io->pool_used == BXO_RATE) {
        bxo_poseidon2_mix(io->cells);
        io->pool_used = 0;
    }
    return io->cells[io->pool_used++];
}
static e4 iop_random_ext(iop_t* io) {
    e4 r;
    for (int k = 0; k < 4; k++) r.c[k] = iop_random_elem(io);
    return r;
}
/* [EXT] Poseidon2Rng::random_bits */
static uint32_t iop_random_bits(iop_t* io, unsigned bits) {
    uint32_t val = bxo_fp_decode(iop_random_elem(io));
    for (int i = 0; i < 3; i++) {
        uint32_t nv = bxo_fp_decode(iop_random_elem(io));
        if (val == 0) val = nv;
    }
    return val & (uint32_t)(((uint64_t)1 << bits) - 1);
}

void bxo_transcript_step(uint32_t state[25], const uint32_t* digests, size_t n_commit, uint32_t* out, size_t n_elems) {
    iop_t io;
    memset(&io, 0, sizeof io);
    memcpy(io.cells, state, sizeof io.cells);
    io.pool_used = state[24];
    for (size_t i = 0; i < n_commit; i++) iop_commit(&io, digests + 8 * i);
    for (size_t e = 0; e < n_elems; e++) out[e] = iop_random_elem(&io);
    memcpy(state, io.cells, sizeof io.cells);
    state[24] = io.pool_used;
}

/* [EXT] Poseidon2Rng::random_bits on an exported state (24 cells + pool counter) */
uint32_t bxo_rng_random_bits(uint32_t state[25], unsigned bits) {
    iop_t io;
    memset(&io, 0, sizeof io);
    memcpy(io.cells, state, sizeof io.cells);
    io.pool_used = state[24];
    uint32_t v = iop_random_bits(&io, bits);
    memcpy(state, io.cells, sizeof io.cells);
    state[24] = io.pool_used;
    return v;
}

/* ---- ext helpers on top of the oracle field ---- */
static e4 e4mul(e4 a, e4 b) { e4 r; bxo_fp4_mul(r.c, a.c, b.c); return r; }
static e4 e4inv(e4 a) { e4 r; bxo_fp4_inv(r.c, a.c); return r; }

static e4 e4sub(e4 a, e4 b) { e4 r; for (int k = 0; k < 4; k++) r.c[k] = bxo_fp_sub(a.c[k], b.c[k]); return r; }
static e4 e4scale(e4 a, uint32_t s) { e4 r; for (int k = 0; k < 4; k++) r.c[k] = bxo_fp_mul(a.c[k], s); return r; }
static e4 e4one(void) { e4 r = {{bxo_fp_encode(1), 0, 0, 0}}; return r; }


/* ---- pseudo-random words of the synthetic witness (definition: include/bx_prover.h, "seeds") ---- */
static uint64_t splitmix64(uint64_t x) {
    uint64_t z = x + GOLDEN;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
static uint32_t synth_word(uint64_t seed, uint32_t col, uint32_t row) {
    uint32_t v = (uint32_t)(splitmix64(seed ^ (((uint64_t)col << 32) | row)) >> 33);
    return v >= BXO_P ? v - BXO_P : v;
}

/* ---- the synthetic circuit (include/bx_prover.h, "The synthetic circuit") ---- */
#define DEFAULT_TERMS 64
#define DEFAULT_DEGREE 4
#define POOL 16
typedef struct {
    uint32_t po2, wc, wd, wa, T, G;
    size_t n, zk, act; /* rows; ZK noise rows at the end of the trace; active rows = n - zk */
    uint32_t F, J, E, pairs;
} circ_t;
static void circ_init(circ_t* c, uint32_t po2, uint32_t wc, uint32_t wd, uint32_t wa, uint32_t terms, uint32_t degree) {
    c->po2 = po2; c->wc = wc; c->wd = wd; c->wa = wa;
    c->T = terms ? terms : DEFAULT_TERMS;
    c->G = degree ? degree : DEFAULT_DEGREE;
    c->n = (size_t)1 << po2;
    c->zk = c->n / 4 < 1994 ? c->n / 4 : 1994; /* risc0_zkp::ZK_CYCLES = 1994 [EXT], capped for the small sizes the tests use */
    c->act = c->n - c->zk;
    c->F = (wd + 1) / 2;
    c->J = wd - c->F;
    c->E = wa / 4;
    c->pairs = 0;
    while (2 * c->pairs + 1 < c->E && 4 * c->pairs + 3 < c->F) c->pairs++;
    if (wc < 2) c->pairs = 0; /* the closing constraint needs the `last` selector */
}
/* which pool entry is factor f of term t */
static unsigned pool_idx(unsigned t, unsigned f) { return (7 * t + 3 * f + (t >> 2) * f + (t >> 4)) & 15u; }
/* code column standing behind csel(i); -1 = the constant one */
static int csel_col(const circ_t* c, unsigned i) { return c->wc >= 3 ? (int)(2 + i % (c->wc - 2)) : -1; }
/* data column an accumulator runs over */
static uint32_t acc_src(const circ_t* c, uint32_t e) {
    uint32_t p = e / 2;
    if (p < c->pairs) return (e & 1) ? 4 * p + 3 : 4 * p + 2;
    return e % c->F;
}
/* the row permutation of pair p: data[4p+3][perm(r)] = data[4p+2][r] */
static size_t perm_row(const circ_t* c, uint32_t p, size_t r) { return (r * 2654435761ull + 12345u + p) % c->act; } /* a bijection of the active rows: 2654435761 is prime and > act */
/* tap set of a column: the rows back it is opened at (first entry 0); returns their number */
#define MAX_TAPS 8
#define MAX_COMBOS 16
static uint32_t backs_of(const circ_t* c, int g, uint32_t col, uint32_t* out) {
    out[0] = 0;
    if (g == 1 && col % 8 == 0) { out[1] = 1; return 2; }
    if (g == 1 && col % 8 == 4) { out[1] = 1; out[2] = 2; return 3; }
    if (g == 2 && col < 4 * c->E) { out[1] = 1; return 2; }
    return 1;
}
/* rows back at which pool slot 1 of derived column j reads free column j (0: the slot repeats slot 0) */
static unsigned slot1_back(uint32_t j) { return j % 8 == 0 ? 1u : (j % 8 == 4 ? 2u : 0u); }

/* ---- MerkleTreeProver ---- */
typedef struct {
    size_t rows, cols;
    unsigned layers, top_layer;
    uint32_t* nodes;         /* 2*rows digests */
    const uint32_t* matrix;  /* rows x cols column-major */
} tree_t;
static unsigned ilog2sz(size_t n) { unsigned k = 0; while (((size_t)1 << k) < n) k++; return k; }
/* [EXT] MerkleTreeParams::new + MerkleTreeProver::new */
static void tree_build(tree_t* t, const uint32_t* matrix, size_t rows, size_t cols) {
    t->rows = rows;
    t->cols = cols;
    t->matrix = matrix;
    t->layers = ilog2sz(rows);
    t->top_layer = 0;
    for (unsigned i = 1; i < t->layers; i++) {
        if (((size_t)1 << i) > QUERIES) break;
        t->top_layer = i;
    }
    t->nodes = (uint32_t*)calloc(16 * rows, 4);
    bxo_hash_rows(t->nodes + 8 * rows, matrix, rows, cols);
    for (unsigned i = t->layers; i-- > 0;) bxo_hash_fold(t->nodes, (size_t)2 << i, (size_t)1 << i);
}
/* [EXT] MerkleTreeProver::commit */
static void tree_commit(tree_t* t, iop_t* io) {
    size_t top = (size_t)1 << t->top_layer;
    iop_write(io, t->nodes + 8 * top, 8 * top);
    iop_commit(io, t->nodes + 8);
}
/* [EXT] MerkleTreeProver::prove */
static void tree_prove(const tree_t* t, iop_t* io, size_t idx) {
    size_t top = (size_t)1 << t->top_layer;
    uint32_t* col = (uint32_t*)malloc(t->cols * 4);
    bxo_gather_sample(col, t->matrix, idx, t->cols, t->rows);
    iop_write(io, col, t->cols);
    free(col);
    idx += t->rows;
    while (idx >= 2 * top) {
        size_t low = idx & 1;
        idx >>= 1;
        iop_write(io, t->nodes + 8 * (2 * idx + (1 - low)), 8);
    }
}

/* ---- PolyGroup ---- */
typedef struct {
    uint32_t width;
    uint32_t* coeffs;     /* width x N, natural-order coefficients after the bit reverse */
    uint32_t* evaluated;  /* width x 4N */
    tree_t tree;
    uint32_t* ntaps;      /* taps per column */
    uint32_t (*backs)[MAX_TAPS]; /* their row offsets */
    uint32_t* combo;      /* combo of the column (columns with the same tap set share one) */
} group_t;
/* [EXT] Prover::commit_group + PolyGroup::new; `coeffs` holds the witness evaluations on entry */
static void commit_group(group_t* g, size_t n, iop_t* io) {
    bxo_batch_interpolate_ntt(g->coeffs, g->width, n);
    bxo_zk_shift(g->coeffs, g->width, n);
    g->evaluated = (uint32_t*)malloc((size_t)g->width * 4 * n * 4);
    bxo_batch_expand_into_evaluate_ntt(g->evaluated, g->coeffs, g->width, n, 2);
    bxo_batch_bit_reverse(g->coeffs, g->width, n);
    tree_build(&g->tree, g->evaluated, 4 * n, g->width);
    tree_commit(&g->tree, io);
}
static void group_free(group_t* g) {
    free(g->coeffs);
    free(g->evaluated);
    free(g->tree.nodes);
    free(g->ntaps);
    free(g->backs);
    free(g->combo);
}

/* Test hook: corrupt one witness cell (group, column, row) after witness generation, so that the statement being proved
 * is false; group < 0 switches the fault off.  The verifier must then refuse the seal at the constraint identity. */
static int fault_group = -1;
static uint32_t fault_col, fault_row;
void bxo_set_witness_fault(int group, uint32_t col, uint32_t row) {
    fault_group = group;
    fault_col = col;
    fault_row = row;
}

/* Test hook: a DISHONEST prover that commits a code group of its own choosing (mode 0 = honest).
 *   mode 1: the `last` selector (code column 1) is committed as the zero column and the public word g_1 is reported as g_1 + 1.
 *           Every constraint gated by `last` (the pair closings and the g_1 boundary constraint) is then multiplied by zero, so
 *           the constraint identity holds for ANY claimed g_1: only a check of the code root against the circuit's control ID
 *           (risc0's check_code) can refuse the seal.
 *   mode 2: the `first` selector (code column 0) is the zero column, every accumulator is the zero column (the recurrence
 *           acc(r) = acc(r-1) * (beta + x) then holds on every row, cyclically) and g_0 is reported as g_0 + 1. */
static int cheat_mode = 0;
void bxo_set_cheat(int mode) { cheat_mode = mode; }

/* Returns a malloc'ed seal (caller frees with bxo_free) or NULL on an internal consistency failure. */
uint32_t* bxo_prove_segment_ex(uint32_t po2, uint32_t w_code, uint32_t w_data, uint32_t w_accum, uint32_t terms, uint32_t degree,
                               uint64_t seed, size_t* seal_words, uint32_t roots_out[32]) {
    /* default noise seed of include/bx_prover.h: splitmix64(seed ^ "ZKNOISE!") */
    return bxo_prove_segment_zk(po2, w_code, w_data, w_accum, terms, degree, seed, splitmix64(seed ^ 0x5A4B4E4F49534521ull), seal_words, roots_out);
}
uint32_t* bxo_prove_segment_zk(uint32_t po2, uint32_t w_code, uint32_t w_data, uint32_t w_accum, uint32_t terms, uint32_t degree,
                               uint64_t seed, uint64_t noise_seed, size_t* seal_words, uint32_t roots_out[32]) {
    bxo_init();
    const size_t n = (size_t)1 << po2, dom = 4 * n;
    const uint32_t widths[4] = {w_code, w_data, w_accum, CHECK_SIZE};
    circ_t cc;
    circ_init(&cc, po2, w_code, w_data, w_accum, terms, degree);
    iop_t io;
    memset(&io, 0, sizeof io);
    group_t grp[4];
    memset(grp, 0, sizeof grp);
    uint32_t *code_w = NULL, *data_w = NULL; /* witness copies of the code and data groups (commit_group works in place) */
    uint32_t combo_backs[MAX_COMBOS][MAX_TAPS], combo_len[MAX_COMBOS], n_trace_combos = 0; /* distinct tap sets of the trace groups */
    uint32_t globals[2] = {0, 0}, n_globals = 0; /* the statement's public words */

    /* header */
    {
        uint32_t hdr[6] = {po2, w_code, w_data, w_accum, cc.T, cc.G}, enc[6], dg[8];
        for (int i = 0; i < 6; i++) enc[i] = bxo_fp_encode(hdr[i]);
        iop_write(&io, hdr, 6);
        bxo_hash_elem_slice(dg, enc, 6, 1);
        iop_commit(&io, dg);
    }
    /* ---- witness generation + trace commitments (code, data, then accum which depends on the transcript) ---- */
    e4 beta = {{0, 0, 0, 0}};
    for (int g = 0; g < 3; g++) {
        group_t* G = &grp[g];
        G->width = widths[g];
        uint64_t gseed = seed + GOLDEN * (uint64_t)(g + 1);
        const uint64_t nseed = noise_seed + GOLDEN * (uint64_t)(g + 1); /* the ZK rows' own generator (upstream: a thread RNG) */
        if (g == 2) {
            beta = iop_random_ext(&io); /* the accumulators' mix */
            gseed ^= ((uint64_t)beta.c[0] << 32) | beta.c[1];
        }
        uint32_t* w = G->coeffs = (uint32_t*)malloc((size_t)G->width * n * 4); /* the witness, column-major */
        if (g == 0) {
            /* code: selectors first / last, then public pseudo-random control words */
            for (uint32_t c = 0; c < G->width; c++)
                for (size_t r = 0; r < n; r++)
                    w[(size_t)c * n + r] = c == 0 ? (r == 0 ? bxo_fp_encode(1) : 0)
                                           : c == 1 ? (r == cc.act - 1 ? bxo_fp_encode(1) : 0) /* last ACTIVE row */
                                                    : synth_word(CODE_SEED, c, (uint32_t)r);
        } else if (g == 1) {
            /* free columns; the last zk rows of every one of them (the permuted copies included) are ZK noise */
            for (uint32_t c = 0; c < cc.F; c++)
                for (size_t r = 0; r < n; r++) w[(size_t)c * n + r] = synth_word(r < cc.act ? gseed : nseed, c, (uint32_t)r);
            /* permuted copies, placed with the oracle's scatter (one entry per cycle) */
            for (uint32_t p = 0; p < cc.pairs; p++) {
                uint32_t* index = (uint32_t*)malloc((n + 1) * 4);
                uint32_t* offsets = (uint32_t*)malloc(n * 4);
                for (size_t r = 0; r <= n; r++) index[r] = (uint32_t)(r < cc.act ? r : cc.act); /* noise cycles scatter nothing */
                for (size_t r = 0; r < n; r++) offsets[r] = r < cc.act ? (uint32_t)((size_t)(4 * p + 3) * n + perm_row(&cc, p, r)) : 0;
                bxo_scatter(w, index, offsets, w + (size_t)(4 * p + 2) * n, n);
                free(index);
                free(offsets);
            }
            /* derived columns: data[F+j][r] = sum_t prod_f pool_j[idx(t,f)] */
            for (uint32_t j = 0; j < cc.J; j++) {
                uint32_t* out = w + (size_t)(cc.F + j) * n;
                _Pragma("omp parallel for schedule(static) num_threads(bxo_get_threads())")
                for (size_t r = 0; r < n; r++) {
                    /* pool: free column j, the same one row back, free columns j+1 and j+2, the eight previous derived
                     * columns (control columns before the first), control columns csel(j..j+3) */
                    uint32_t pool[POOL];
                    pool[0] = w[(size_t)j * n + r];
                    pool[1] = slot1_back(j) ? w[(size_t)j * n + (r + n - slot1_back(j)) % n] : pool[0];
                    pool[2] = w[(size_t)((j + 1) % cc.F) * n + r];
                    pool[3] = w[(size_t)((j + 2) % cc.F) * n + r];
                    for (uint32_t s_ = 1; s_ <= 8; s_++) {
                        if (j >= s_) pool[3 + s_] = w[(size_t)(cc.F + j - s_) * n + r];
                        else {
                            int cs = csel_col(&cc, s_ - j - 1);
                            pool[3 + s_] = cs < 0 ? bxo_fp_encode(1) : code_w[(size_t)cs * n + r];
                        }
                    }
                    for (uint32_t q = 0; q < 4; q++) {
                        int ck = csel_col(&cc, j + q);
                        pool[12 + q] = ck < 0 ? bxo_fp_encode(1) : code_w[(size_t)ck * n + r];
                    }
                    uint32_t sum = 0;
                    for (uint32_t t = 0; t < cc.T; t++) {
                        uint32_t prod = pool[pool_idx(t, 0)];
                        for (uint32_t f = 1; f < cc.G; f++) prod = bxo_fp_mul(prod, pool[pool_idx(t, f)]);
                        sum = bxo_fp_add(sum, prod);
                    }
                    out[r] = sum;
                }
            }
        } else {
            /* accumulators: acc_e(r) = prod_{i<=r} (beta^(e+1) + data[src(e)][i]), one ext element per row, through the
             * oracle's prefix_products; component k goes to column 4e+k.  Leftover columns are noise. */
            e4 be = beta; /* beta^(floor(e/2)+1): the two accumulators of a pair share their challenge */
            uint32_t* run = (uint32_t*)malloc(n * 16);
            for (uint32_t e = 0; e < cc.E; e++) {
                const uint32_t* x = data_w + (size_t)acc_src(&cc, e) * n;
                for (size_t r = 0; r < n; r++) {
                    run[4 * r + 0] = bxo_fp_add(be.c[0], x[r]);
                    run[4 * r + 1] = be.c[1];
                    run[4 * r + 2] = be.c[2];
                    run[4 * r + 3] = be.c[3];
                }
                bxo_prefix_products(run, n);
                for (int k = 0; k < 4; k++)
                    for (size_t r = 0; r < n; r++) w[(size_t)(4 * e + k) * n + r] = run[4 * r + k];
                if (e & 1) be = e4mul(be, beta);
            }
            free(run);
            for (uint32_t c = 4 * cc.E; c < G->width; c++)
                for (size_t r = 0; r < n; r++) w[(size_t)c * n + r] = synth_word(gseed, c, (uint32_t)r);
        }
        if (cheat_mode == 1 && g == 0 && G->width >= 2) memset(w + n, 0, n * 4);
        if (cheat_mode == 2 && g == 0) memset(w, 0, n * 4);
        if (cheat_mode == 2 && g == 2) memset(w, 0, (size_t)4 * cc.E * n * 4);
        if (fault_group == g && fault_col < G->width && fault_row < n) {
            uint32_t* cell = &w[(size_t)fault_col * n + fault_row];
            *cell = bxo_fp_add(*cell, bxo_fp_encode(1));
        }
        /* commit_group interpolates in place, so keep what later stages of witness generation read */
        if (g == 0) {
            code_w = (uint32_t*)malloc((size_t)G->width * n * 4);
            memcpy(code_w, w, (size_t)G->width * n * 4);
        } else if (g == 1) {
            data_w = (uint32_t*)malloc((size_t)G->width * n * 4);
            memcpy(data_w, w, (size_t)G->width * n * 4);
        }
        G->ntaps = (uint32_t*)malloc(G->width * 4);
        G->backs = malloc(G->width * sizeof *G->backs);
        G->combo = (uint32_t*)malloc(G->width * 4);
        for (uint32_t c = 0; c < G->width; c++) {
            G->ntaps[c] = backs_of(&cc, g, c, G->backs[c]);
            /* combos in order of first appearance over code, data, accum */
            uint32_t id = 0;
            for (; id < n_trace_combos; id++)
                if (combo_len[id] == G->ntaps[c] && memcmp(combo_backs[id], G->backs[c], 4 * G->ntaps[c]) == 0) break;
            if (id == n_trace_combos) {
                combo_len[id] = G->ntaps[c];
                memcpy(combo_backs[id], G->backs[c], 4 * G->ntaps[c]);
                n_trace_combos++;
            }
            G->combo[c] = id;
        }
        /* The statement's public words come out of the data witness and are bound into the transcript before any
         * commitment: the code group is committed only once they are known. */
        if (g == 0) continue;
        if (g == 1) {
            n_globals = cc.wc >= 2 ? 2 : 1;
            globals[0] = data_w[0];                                        /* data[0][0] */
            globals[1] = data_w[(size_t)(cc.wd - 1) * n + (cc.act - 1)];     /* data[wd-1][last active row] */
            if (cheat_mode == 1) globals[1] = bxo_fp_add(globals[1], bxo_fp_encode(1)); /* a false claim */
            if (cheat_mode == 2) globals[0] = bxo_fp_add(globals[0], bxo_fp_encode(1));
            uint32_t dg[8];
            iop_write(&io, globals, n_globals);
            bxo_hash_elem_slice(dg, globals, n_globals, 1);
            iop_commit(&io, dg);
            commit_group(&grp[0], n, &io);
            if (roots_out) memcpy(roots_out, grp[0].tree.nodes + 8, 32);
        }
        commit_group(G, n, &io);
        if (roots_out) memcpy(roots_out + 8 * g, G->tree.nodes + 8, 32);
    }
    free(code_w);
    free(data_w);
    /* ---- eval_check: check(x) = sum_i poly_mix^i C_i(x) / ((3x)^N - 1) on the 4N domain x = w_4N^row ---- */
    group_t* CK = &grp[3];
    {
        CK->width = CHECK_SIZE;
        e4 poly_mix = iop_random_ext(&io);
        uint32_t* check = (uint32_t*)calloc(16 * n, 4); /* 4 planes x 4N */
        const size_t n_cons = (size_t)cc.J + cc.E + cc.pairs + n_globals;
        e4* mixpow = (e4*)malloc((n_cons + 1) * sizeof(e4));
        mixpow[0] = e4one();
        for (size_t i = 1; i <= n_cons; i++) mixpow[i] = e4mul(mixpow[i - 1], poly_mix);
        e4* betas = (e4*)malloc((cc.E + 1) * sizeof(e4));
        betas[0] = beta;
        for (uint32_t e = 1; e < cc.E; e++) betas[e] = (e & 1) ? betas[e - 1] : e4mul(betas[e - 1], beta);
        /* 1 / ((3x)^N - 1) takes four values on the domain: (3 w_4N^row)^N = 3^N w_4^(row mod 4) */
        uint32_t zinv[4];
        {
            uint32_t t3n = bxo_fp_pow(bxo_fp_encode(3), n), w4 = bxo_rou_fwd(2), cur = bxo_fp_encode(1);
            for (int m = 0; m < 4; m++) {
                zinv[m] = bxo_fp_inv(bxo_fp_sub(bxo_fp_mul(t3n, cur), bxo_fp_encode(1)));
                cur = bxo_fp_mul(cur, w4);
            }
        }
        const uint32_t* ecode = grp[0].evaluated;
        const uint32_t* edata = grp[1].evaluated;
        const uint32_t* eacc = grp[2].evaluated;
        const uint32_t one = bxo_fp_encode(1);
        _Pragma("omp parallel for schedule(static) num_threads(bxo_get_threads())")
        for (size_t i = 0; i < dom; i++) {
            const size_t ib = (i + dom - 4) % dom; /* one row back: x * w_N^-1 = w_4N^(row - 4) */
            e4 tot = {{0, 0, 0, 0}};
            for (uint32_t j = 0; j < cc.J; j++) {
                uint32_t pool[POOL];
                pool[0] = edata[(size_t)j * dom + i];
                pool[1] = slot1_back(j) ? edata[(size_t)j * dom + (i + dom - 4 * slot1_back(j)) % dom] : pool[0];
                pool[2] = edata[(size_t)((j + 1) % cc.F) * dom + i];
                pool[3] = edata[(size_t)((j + 2) % cc.F) * dom + i];
                for (uint32_t s_ = 1; s_ <= 8; s_++) {
                    if (j >= s_) pool[3 + s_] = edata[(size_t)(cc.F + j - s_) * dom + i];
                    else {
                        int cs = csel_col(&cc, s_ - j - 1);
                        pool[3 + s_] = cs < 0 ? one : ecode[(size_t)cs * dom + i];
                    }
                }
                for (uint32_t q = 0; q < 4; q++) {
                    int ck = csel_col(&cc, j + q);
                    pool[12 + q] = ck < 0 ? one : ecode[(size_t)ck * dom + i];
                }
                uint32_t sum = 0;
                for (uint32_t t = 0; t < cc.T; t++) {
                    uint32_t prod = pool[pool_idx(t, 0)];
                    for (uint32_t f = 1; f < cc.G; f++) prod = bxo_fp_mul(prod, pool[pool_idx(t, f)]);
                    sum = bxo_fp_add(sum, prod);
                }
                uint32_t cons = bxo_fp_sub(edata[(size_t)(cc.F + j) * dom + i], sum);
                for (int k = 0; k < 4; k++) tot.c[k] = bxo_fp_add(tot.c[k], bxo_fp_mul(mixpow[j].c[k], cons));
            }
            const uint32_t first = ecode[i], last = cc.wc >= 2 ? ecode[dom + i] : 0;
            for (uint32_t e = 0; e < cc.E; e++) {
                e4 a, ab, inner, fac, cons;
                for (int k = 0; k < 4; k++) {
                    a.c[k] = eacc[(size_t)(4 * e + k) * dom + i];
                    ab.c[k] = eacc[(size_t)(4 * e + k) * dom + ib];
                }
                inner = e4scale(ab, bxo_fp_sub(one, first));
                inner.c[0] = bxo_fp_add(inner.c[0], first);
                fac = betas[e];
                fac.c[0] = bxo_fp_add(fac.c[0], edata[(size_t)acc_src(&cc, e) * dom + i]);
                cons = e4sub(a, e4mul(inner, fac));
                cons = e4mul(mixpow[cc.J + e], cons);
                for (int k = 0; k < 4; k++) tot.c[k] = bxo_fp_add(tot.c[k], cons.c[k]);
            }
            for (uint32_t p = 0; p < cc.pairs; p++) {
                e4 d;
                for (int k = 0; k < 4; k++)
                    d.c[k] = bxo_fp_sub(eacc[(size_t)(4 * (2 * p + 1) + k) * dom + i], eacc[(size_t)(4 * (2 * p) + k) * dom + i]);
                d = e4mul(mixpow[cc.J + cc.E + p], e4scale(d, last));
                for (int k = 0; k < 4; k++) tot.c[k] = bxo_fp_add(tot.c[k], d.c[k]);
            }
            {   /* boundary constraints: first * (data[0] - g0), last * (data[wd-1] - g1) */
                const size_t b0 = (size_t)cc.J + cc.E + cc.pairs;
                uint32_t v = bxo_fp_mul(first, bxo_fp_sub(edata[i], globals[0]));
                for (int k = 0; k < 4; k++) tot.c[k] = bxo_fp_add(tot.c[k], bxo_fp_mul(mixpow[b0].c[k], v));
                if (n_globals > 1) {
                    v = bxo_fp_mul(last, bxo_fp_sub(edata[(size_t)(cc.wd - 1) * dom + i], globals[1]));
                    for (int k = 0; k < 4; k++) tot.c[k] = bxo_fp_add(tot.c[k], bxo_fp_mul(mixpow[b0 + 1].c[k], v));
                }
            }
            tot = e4scale(tot, zinv[i & 3]);
            for (int k = 0; k < 4; k++) check[(size_t)k * dom + i] = tot.c[k];
        }
        free(mixpow);
        free(betas);
        bxo_batch_interpolate_ntt(check, 4, dom);
        CK->coeffs = check; /* now viewed as 16 polynomials of size n */
        CK->ntaps = (uint32_t*)malloc(CHECK_SIZE * 4);
        CK->backs = malloc(CHECK_SIZE * sizeof *CK->backs);
        CK->combo = (uint32_t*)malloc(CHECK_SIZE * 4);
        for (int c = 0; c < CHECK_SIZE; c++) {
            CK->ntaps[c] = 1;
            CK->backs[c][0] = 0;
            CK->combo[c] = n_trace_combos; /* the check group's combo comes last */
        }
        bxo_zk_shift(CK->coeffs, CHECK_SIZE, n);
        CK->evaluated = (uint32_t*)malloc((size_t)CHECK_SIZE * dom * 4);
        bxo_batch_expand_into_evaluate_ntt(CK->evaluated, CK->coeffs, CHECK_SIZE, n, 2);
        bxo_batch_bit_reverse(CK->coeffs, CHECK_SIZE, n);
        tree_build(&CK->tree, CK->evaluated, dom, CHECK_SIZE);
        tree_commit(&CK->tree, &io);
        if (roots_out) memcpy(roots_out + 24, CK->tree.nodes + 8, 32);
    }
    /* DEEP: taps.  Column c of a trace group is opened at Z * w_N^-b for every b of its tap set; the check columns at Z^4 / 3 */
    e4 Z = iop_random_ext(&io);
    e4 Z4 = e4scale(e4mul(e4mul(Z, Z), e4mul(Z, Z)), bxo_fp_inv(bxo_fp_encode(3)));
    const uint32_t n_combos = n_trace_combos + 1;
    size_t total_taps = 0;
    for (int g = 0; g < 4; g++)
        for (uint32_t c = 0; c < grp[g].width; c++) total_taps += grp[g].ntaps[c];
    uint32_t* coeff_u = (uint32_t*)malloc(total_taps * 16);
    {
        size_t u = 0;
        for (int g = 0; g < 4; g++) {
            /* one batch_evaluate_any per group, like upstream (which = column, xs = Z * back_one^back) */
            size_t ne = 0;
            for (uint32_t c = 0; c < grp[g].width; c++) ne += grp[g].ntaps[c];
            uint32_t* which = (uint32_t*)malloc(ne * 4);
            uint32_t* xs = (uint32_t*)malloc(ne * 16);
            uint32_t* ev = (uint32_t*)malloc(ne * 16);
            size_t e = 0;
            for (uint32_t c = 0; c < grp[g].width; c++)
                for (uint32_t t = 0; t < grp[g].ntaps[c]; t++, e++) {
                    which[e] = c;
                    e4 x = g == 3 ? Z4 : e4scale(Z, bxo_fp_pow(bxo_rou_rev(po2), grp[g].backs[c][t]));
                    memcpy(xs + 4 * e, x.c, 16);
                }
            bxo_batch_evaluate_any(grp[g].coeffs, n, which, xs, ev, ne);
            e = 0;
            for (uint32_t c = 0; c < grp[g].width; c++) {
                /* [EXT] poly_interpolate: the polynomial of degree < k through the k tap points, by Newton's divided
                 * differences, then expanded into monomial coefficients */
                const uint32_t k = grp[g].ntaps[c];
                e4 px[MAX_TAPS], dd[MAX_TAPS], co[MAX_TAPS];
                for (uint32_t t = 0; t < k; t++) {
                    memcpy(px[t].c, xs + 4 * (e + t), 16);
                    memcpy(dd[t].c, ev + 4 * (e + t), 16);
                }
                for (uint32_t lvl = 1; lvl < k; lvl++)
                    for (uint32_t t = k - 1; t >= lvl; t--)
                        dd[t] = e4mul(e4sub(dd[t], dd[t - 1]), e4inv(e4sub(px[t], px[t - lvl])));
                /* Horner from the highest divided difference: co <- co * (x - px[t]) + dd[t], polynomial arithmetic on co */
                {
                    e4 tmp[MAX_TAPS];
                    for (uint32_t t = 0; t < k; t++) memset(co[t].c, 0, 16);
                    co[0] = dd[k - 1];
                    uint32_t deg = 0;
                    for (uint32_t t = k - 1; t-- > 0;) {
                        for (uint32_t d = 0; d <= deg + 1; d++) memset(tmp[d].c, 0, 16);
                        for (uint32_t d = 0; d <= deg; d++) {
                            for (int q = 0; q < 4; q++) tmp[d + 1].c[q] = bxo_fp_add(tmp[d + 1].c[q], co[d].c[q]);
                            tmp[d] = e4sub(tmp[d], e4mul(co[d], px[t]));
                        }
                        for (int q = 0; q < 4; q++) tmp[0].c[q] = bxo_fp_add(tmp[0].c[q], dd[t].c[q]);
                        deg++;
                        for (uint32_t d = 0; d <= deg; d++) co[d] = tmp[d];
                    }
                }
                for (uint32_t t = 0; t < k; t++) memcpy(coeff_u + u + 4 * t, co[t].c, 16);
                u += 4 * k;
                e += k;
            }
            free(which);
            free(xs);
            free(ev);
        }
        uint32_t dg[8];
        iop_write(&io, coeff_u, 4 * total_taps);
        bxo_hash_elem_slice(dg, coeff_u, 4 * total_taps, 1);
        iop_commit(&io, dg);
    }
    /* DEEP: mix, subtract u, divide by every tap point of the combo */
    e4 mix = iop_random_ext(&io);
    uint32_t* combos = (uint32_t*)calloc((size_t)n_combos * 4 * n, 4);
    int ok = 1;
    {
        e4 cur = e4one();
        size_t u = 0;
        for (int g = 0; g < 4; g++) {
            bxo_mix_poly_coeffs(combos, cur.c, mix.c, grp[g].coeffs, grp[g].combo, grp[g].width, n);
            for (uint32_t c = 0; c < grp[g].width; c++) {
                uint32_t* target = combos + (size_t)grp[g].combo[c] * 4 * n;
                for (uint32_t t = 0; t < grp[g].ntaps[c]; t++, u += 4) {
                    e4 cu;
                    memcpy(cu.c, coeff_u + u, 16);
                    e4 m = e4mul(cur, cu);
                    for (int k = 0; k < 4; k++) target[4 * t + k] = bxo_fp_sub(target[4 * t + k], m.c[k]);
                }
                cur = e4mul(cur, mix);
            }
        }
        uint32_t rem[4];
        for (uint32_t id = 0; id < n_trace_combos; id++)
            for (uint32_t t = 0; t < combo_len[id]; t++) {
                e4 x = e4scale(Z, bxo_fp_pow(bxo_rou_rev(po2), combo_backs[id][t]));
                ok &= bxo_poly_divide(combos + (size_t)id * 4 * n, n, x.c, rem);
            }
        ok &= bxo_poly_divide(combos + (size_t)n_trace_combos * 4 * n, n, Z4.c, rem);
    }
    uint32_t* fin = (uint32_t*)malloc(4 * n * 4);
    bxo_eltwise_sum_extelem(fin, combos, n, n_combos);
    bxo_batch_bit_reverse(fin, 4, n);
    free(combos);
    free(coeff_u);

    /* fri_prove */
    size_t n_rounds = 0;
    for (size_t s = n; s > FRI_MIN_DEGREE; s /= FRI_FOLD) n_rounds++;
    tree_t* rt = (tree_t*)calloc(n_rounds ? n_rounds : 1, sizeof(tree_t));
    uint32_t** revals = (uint32_t**)calloc(n_rounds ? n_rounds : 1, sizeof(uint32_t*));
    uint32_t* coeffs = fin;
    size_t size = n;
    for (size_t r = 0; r < n_rounds; r++) {
        size_t domain = size * INV_RATE;
        revals[r] = (uint32_t*)malloc(4 * domain * 4);
        bxo_batch_expand_into_evaluate_ntt(revals[r], coeffs, 4, size, 2);
        tree_build(&rt[r], revals[r], domain / FRI_FOLD, FRI_FOLD * 4);
        tree_commit(&rt[r], &io);
        e4 fold_mix = iop_random_ext(&io);
        uint32_t* out = (uint32_t*)malloc(4 * (size / FRI_FOLD) * 4);
        bxo_fri_fold(out, coeffs, fold_mix.c, size / FRI_FOLD);
        free(coeffs);
        coeffs = out;
        size /= FRI_FOLD;
    }
    {
        bxo_batch_bit_reverse(coeffs, 4, size);
        uint32_t dg[8];
        iop_write(&io, coeffs, 4 * size);
        bxo_hash_elem_slice(dg, coeffs, 4 * size, 1);
        iop_commit(&io, dg);
        free(coeffs);
    }
    for (int q = 0; q < QUERIES; q++) {
        uint32_t rng = iop_random_bits(&io, ilog2sz(dom));
        size_t pos = rng % dom;
        for (int g = 0; g < 4; g++) tree_prove(&grp[g].tree, &io, pos);
        for (size_t r = 0; r < n_rounds; r++) {
            size_t group = pos % rt[r].rows;
            tree_prove(&rt[r], &io, group);
            pos = group;
        }
    }
    for (size_t r = 0; r < n_rounds; r++) {
        free(rt[r].nodes);
        free(revals[r]);
    }
    free(rt);
    free(revals);
    for (int g = 0; g < 4; g++) group_free(&grp[g]);
    if (!ok) {
        free(io.seal);
        return NULL;
    }
    *seal_words = io.len;
    return io.seal;
}
/* The synthetic circuit's control ID for (po2, w_code): the Merkle root of the committed code group, with the oracle's own
 * commit_group ([EXT] risc0's control IDs are the same thing for its circuits: the table check_code compares against). */
void bxo_control_id(uint32_t po2, uint32_t w_code, uint32_t id_out[8]) {
    bxo_init();
    const size_t n = (size_t)1 << po2;
    const size_t zk = n / 4 < 1994 ? n / 4 : 1994, act = n - zk;
    group_t G;
    iop_t io;
    memset(&G, 0, sizeof G);
    memset(&io, 0, sizeof io);
    G.width = w_code;
    G.coeffs = (uint32_t*)malloc((size_t)w_code * n * 4);
    for (uint32_t c = 0; c < w_code; c++)
        for (size_t r = 0; r < n; r++)
            G.coeffs[(size_t)c * n + r] = c == 0 ? (r == 0 ? bxo_fp_encode(1) : 0) : c == 1 ? (r == act - 1 ? bxo_fp_encode(1) : 0) : synth_word(CODE_SEED, c, (uint32_t)r);
    commit_group(&G, n, &io);
    memcpy(id_out, G.tree.nodes + 8, 32);
    group_free(&G);
    free(io.seal);
}

uint32_t* bxo_prove_segment(uint32_t po2, uint32_t w_code, uint32_t w_data, uint32_t w_accum, uint64_t seed,
                            size_t* seal_words, uint32_t roots_out[32]) {
    return bxo_prove_segment_ex(po2, w_code, w_data, w_accum, 0, 0, seed, seal_words, roots_out);
}
void bxo_free(void* p) { free(p); }
