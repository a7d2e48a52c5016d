/*
 * bx_oracle.c — CPU ORACLE (test infrastructure only; see bx_oracle.h for scope and parity status).
 *
 * Every function names the upstream item it restates.  "[EXT]" = the item lives in a crate that the
 * reference pins but does not vendor (risc0-zkp 3.0.3 / risc0-core 3.0.0, reference Cargo.lock:9155,9012);
 * the in-tree call sites that reach it are bento/crates/workflow/src/tasks/prove.rs:41-49,92-100.
 */
#include "bx_oracle.h"

#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define P BXO_P
/* [EXT] risc0-core field/baby_bear.rs: M = P^-1 mod 2^32 (used negated), R2 = 2^64 mod P. */
#define MONT_M 0x88000001u
#define MONT_R2 1172168163u

static int g_threads = 0;
void bxo_set_threads(int n) { g_threads = n; }
int bxo_get_threads(void) {
#ifdef _OPENMP
    return g_threads > 0 ? g_threads : omp_get_max_threads();
#else
    return 1;
#endif
}
#ifdef _OPENMP
#define PAR_FOR _Pragma("omp parallel for schedule(static) num_threads(bxo_get_threads())")
#else
#define PAR_FOR
#endif

/* ------------------------------------------------------------------ field */
/* [EXT] baby_bear.rs `const fn mul(lhs, rhs)`: Montgomery product, result canonical in [0,P). */
static inline uint32_t fmul(uint32_t a, uint32_t b) {
    uint64_t o = (uint64_t)a * (uint64_t)b;
    uint32_t low = 0u - (uint32_t)o;
    uint32_t red = MONT_M * low;
    o += (uint64_t)red * (uint64_t)P;
    uint32_t r = (uint32_t)(o >> 32);
    return r >= P ? r - P : r;
}
/* [EXT] baby_bear.rs `add`: wrapping add then conditional subtract. */
static inline uint32_t fadd(uint32_t a, uint32_t b) {
    uint32_t r = a + b;
    return r >= P ? r - P : r;
}
/* [EXT] baby_bear.rs `sub`: wrapping sub, add P on borrow. */
static inline uint32_t fsub(uint32_t a, uint32_t b) {
    uint32_t r = a - b;
    return r > P ? r + P : r;
}
static uint32_t fpow(uint32_t a, uint64_t e) {
    uint32_t r = fmul(MONT_R2, 1u); /* encode(1) */
    while (e) {
        if (e & 1) r = fmul(r, a);
        a = fmul(a, a);
        e >>= 1;
    }
    return r;
}
uint32_t bxo_fp_encode(uint32_t c) { return fmul(MONT_R2, c % P); }
uint32_t bxo_fp_decode(uint32_t m) { return fmul(1u, m); }
uint32_t bxo_fp_add(uint32_t a, uint32_t b) { return fadd(a, b); }
uint32_t bxo_fp_sub(uint32_t a, uint32_t b) { return fsub(a, b); }
uint32_t bxo_fp_mul(uint32_t a, uint32_t b) { return fmul(a, b); }
uint32_t bxo_fp_pow(uint32_t a, uint64_t e) { return fpow(a, e); }
uint32_t bxo_fp_inv(uint32_t a) { return fpow(a, (uint64_t)P - 2); }

/* Fp4 = Fp[X]/(X^4 + 11); [EXT] baby_bear.rs ExtElem::mul (NBETA = P - 11). */
typedef struct { uint32_t c[4]; } fp4;
static uint32_t g_nbeta, g_one;
static inline fp4 f4mul(fp4 a, fp4 b) {
    fp4 r;
    r.c[0] = fadd(fmul(a.c[0], b.c[0]),
                  fmul(g_nbeta, fadd(fadd(fmul(a.c[1], b.c[3]), fmul(a.c[2], b.c[2])), fmul(a.c[3], b.c[1]))));
    r.c[1] = fadd(fadd(fmul(a.c[0], b.c[1]), fmul(a.c[1], b.c[0])),
                  fmul(g_nbeta, fadd(fmul(a.c[2], b.c[3]), fmul(a.c[3], b.c[2]))));
    r.c[2] = fadd(fadd(fadd(fmul(a.c[0], b.c[2]), fmul(a.c[1], b.c[1])), fmul(a.c[2], b.c[0])),
                  fmul(g_nbeta, fmul(a.c[3], b.c[3])));
    r.c[3] = fadd(fadd(fadd(fmul(a.c[0], b.c[3]), fmul(a.c[1], b.c[2])), fmul(a.c[2], b.c[1])), fmul(a.c[3], b.c[0]));
    return r;
}
static inline fp4 f4add(fp4 a, fp4 b) {
    fp4 r;
    for (int k = 0; k < 4; k++) r.c[k] = fadd(a.c[k], b.c[k]);
    return r;
}
static inline fp4 f4sub(fp4 a, fp4 b) {
    fp4 r;
    for (int k = 0; k < 4; k++) r.c[k] = fsub(a.c[k], b.c[k]);
    return r;
}
static inline fp4 f4scale(fp4 a, uint32_t s) {
    fp4 r;
    for (int k = 0; k < 4; k++) r.c[k] = fmul(a.c[k], s);
    return r;
}
static inline fp4 f4one(void) { fp4 r = {{g_one, 0, 0, 0}}; return r; }
static inline fp4 f4zero(void) { fp4 r = {{0, 0, 0, 0}}; return r; }
static inline fp4 f4load(const uint32_t* p) { fp4 r; memcpy(r.c, p, 16); return r; }
void bxo_fp4_mul(uint32_t out[4], const uint32_t a[4], const uint32_t b[4]) {
    bxo_init();
    fp4 r = f4mul(f4load(a), f4load(b));
    memcpy(out, r.c, 16);
}
/* [EXT] ExtElem::inv — via the norm to the quadratic subfield (a(X)·a(-X) has only even powers). */
void bxo_fp4_inv(uint32_t out[4], const uint32_t a[4]) {
    bxo_init();
    /* b0 + b2 X^2 = a(X) * a(-X), with X^4 = -11 */
    uint32_t a0 = a[0], a1 = a[1], a2 = a[2], a3 = a[3];
    uint32_t beta = fsub(0, g_nbeta); /* 11 */
    /* a(X)a(-X) = (a0 + a2 X^2)^2 - X^2 (a1 + a3 X^2)^2 */
    uint32_t b0 = fadd(fmul(a0, a0), fmul(beta, fsub(fmul(fadd(a1, a1), a3), fmul(a2, a2))));
    uint32_t b2 = fadd(fsub(fmul(fadd(a0, a0), a2), fmul(a1, a1)), fmul(beta, fmul(a3, a3)));
    /* norm c = b0^2 + 11 b2^2 in Fp;  inv = a(-X) * (b0 - b2 X^2) / c */
    uint32_t c = fadd(fmul(b0, b0), fmul(beta, fmul(b2, b2)));
    uint32_t ic = bxo_fp_inv(c);
    uint32_t d0 = fmul(b0, ic), d2 = fsub(0, fmul(b2, ic));
    fp4 an = {{a0, fsub(0, a1), a2, fsub(0, a3)}};
    fp4 d = {{d0, 0, d2, 0}};
    fp4 r = f4mul(an, d);
    memcpy(out, r.c, 16);
}

/* ------------------------------------------------------------------ tables */
static uint32_t g_rou_fwd[28], g_rou_rev[28];
static int g_inited = 0;
static void poseidon2_default_params(void);

void bxo_init(void) {
    if (g_inited) return;
    g_one = bxo_fp_encode(1);
    g_nbeta = bxo_fp_encode(P - 11);
    /* [EXT] baby_bear.rs ROU_FWD/ROU_REV: 137 is a primitive 2^27-th root of unity. */
    g_rou_fwd[27] = bxo_fp_encode(137);
    for (int k = 26; k >= 0; k--) g_rou_fwd[k] = fmul(g_rou_fwd[k + 1], g_rou_fwd[k + 1]);
    for (int k = 0; k < 28; k++) g_rou_rev[k] = bxo_fp_inv(g_rou_fwd[k]);
    poseidon2_default_params();
    g_inited = 1;
}
uint32_t bxo_rou_fwd(unsigned k) { bxo_init(); return g_rou_fwd[k]; }
uint32_t bxo_rou_rev(unsigned k) { bxo_init(); return g_rou_rev[k]; }

static unsigned log2_exact(size_t n) {
    unsigned k = 0;
    while (((size_t)1 << k) < n) k++;
    return k;
}
static inline uint32_t bitrev(uint32_t v, unsigned bits) {
    uint32_t r = 0;
    for (unsigned i = 0; i < bits; i++) r |= ((v >> i) & 1u) << (bits - 1 - i);
    return r;
}

/* ------------------------------------------------------------------ NTT */
/* [EXT] risc0-zkp core/ntt.rs interpolate_ntt: recursive DIF, natural evals -> bit-reversed coeffs
 * (the final 1/N scale is applied by the caller exactly as upstream does after the recursion). */
static void dif_rev(uint32_t* io, unsigned n) {
    if (n == 0) return;
    size_t half = (size_t)1 << (n - 1);
    uint32_t step = g_rou_rev[n], cur = g_one;
    for (size_t i = 0; i < half; i++) {
        uint32_t a = io[i], b = io[i + half];
        io[i] = fadd(a, b);
        io[i + half] = fmul(fsub(a, b), cur);
        cur = fmul(cur, step);
    }
    dif_rev(io, n - 1);
    dif_rev(io + half, n - 1);
}
/* [EXT] core/ntt.rs evaluate_ntt: recursive DIT, bit-reversed coeffs -> natural evals; recursion stops
 * at n == expand_bits (the duplicated input already equals those trivial stages). */
static void dit_fwd(uint32_t* io, unsigned n, unsigned expand_bits) {
    if (n == expand_bits) return;
    size_t half = (size_t)1 << (n - 1);
    dit_fwd(io, n - 1, expand_bits);
    dit_fwd(io + half, n - 1, expand_bits);
    uint32_t step = g_rou_fwd[n], cur = g_one;
    for (size_t i = 0; i < half; i++) {
        uint32_t a = io[i], b = fmul(io[i + half], cur);
        io[i] = fadd(a, b);
        io[i + half] = fsub(a, b);
        cur = fmul(cur, step);
    }
}
/* [EXT] hal/cpu.rs batch_interpolate_ntt */
void bxo_batch_interpolate_ntt(uint32_t* io, size_t count, size_t size) {
    bxo_init();
    unsigned n = log2_exact(size);
    uint32_t norm = bxo_fp_inv(bxo_fp_encode((uint32_t)size));
    PAR_FOR
    for (size_t c = 0; c < count; c++) {
        uint32_t* col = io + c * size;
        dif_rev(col, n);
        for (size_t i = 0; i < size; i++) col[i] = fmul(col[i], norm);
    }
}
void bxo_batch_evaluate_ntt(uint32_t* io, size_t count, size_t size, unsigned expand_bits) {
    bxo_init();
    unsigned n = log2_exact(size);
    PAR_FOR
    for (size_t c = 0; c < count; c++) dit_fwd(io + c * size, n, expand_bits);
}
/* [EXT] hal/cpu.rs batch_expand_into_evaluate_ntt = batch_expand (out[i] = in[i >> bits]) + evaluate_ntt */
void bxo_batch_expand_into_evaluate_ntt(uint32_t* out, const uint32_t* in, size_t count, size_t in_size,
                                        unsigned expand_bits) {
    bxo_init();
    size_t out_size = in_size << expand_bits;
    unsigned n = log2_exact(out_size);
    PAR_FOR
    for (size_t c = 0; c < count; c++) {
        uint32_t* o = out + c * out_size;
        const uint32_t* s = in + c * in_size;
        for (size_t i = 0; i < out_size; i++) o[i] = s[i >> expand_bits];
        dit_fwd(o, n, expand_bits);
    }
}
/* [EXT] hal/cpu.rs batch_bit_reverse */
void bxo_batch_bit_reverse(uint32_t* io, size_t count, size_t size) {
    unsigned n = log2_exact(size);
    PAR_FOR
    for (size_t c = 0; c < count; c++) {
        uint32_t* col = io + c * size;
        for (size_t i = 0; i < size; i++) {
            size_t r = bitrev((uint32_t)i, n);
            if (i < r) { uint32_t t = col[i]; col[i] = col[r]; col[r] = t; }
        }
    }
}
/* [EXT] hal/cpu.rs zk_shift: io[i] *= 3^bitrev(i mod size) */
void bxo_zk_shift(uint32_t* io, size_t count, size_t size) {
    bxo_init();
    unsigned n = log2_exact(size);
    uint32_t three = bxo_fp_encode(3);
    PAR_FOR
    for (size_t c = 0; c < count; c++) {
        uint32_t* col = io + c * size;
        for (size_t i = 0; i < size; i++) col[i] = fmul(col[i], fpow(three, bitrev((uint32_t)i, n)));
    }
}

/* ------------------------------------------------------------------ Poseidon2 */
#define ROUNDS_HALF_FULL 4
#define ROUNDS_PARTIAL 21
#define N_RC (BXO_CELLS * 2 * ROUNDS_HALF_FULL + ROUNDS_PARTIAL) /* 213 */
static uint32_t g_rc[N_RC];          /* Montgomery */
static uint32_t g_diag[BXO_CELLS];   /* Montgomery */
static uint32_t g_rc_canon[N_RC], g_diag_canon[BXO_CELLS];

/* Round constants: the Poseidon/Poseidon2 reference generator (Grain LFSR, "generate_params_poseidon.sage" /
 * HorizenLabs poseidon2 "poseidon2_rust_params.sage"): 80-bit state = field(2)=1 | sbox(4)=0 | n(12)=31 |
 * t(12)=24 | R_F(10)=8 | R_P(10)=21 | 30 ones; taps 62,51,38,23,13,0; discard 160; self-shrinking output;
 * rejection-sample 31-bit integers < P; Poseidon2 draws t*R_F + R_P constants.  Upstream stores the
 * resulting table in risc0-zkp core/hash/poseidon2/consts.rs (ROUND_CONSTANTS) [EXT]. */
static void grain_round_constants(uint32_t* out) {
    unsigned char bits[80];
    int pos = 0;
    unsigned vals[6] = {1, 0, 31, 24, 8, 21}, widths[6] = {2, 4, 12, 12, 10, 10};
    for (int f = 0; f < 6; f++)
        for (int b = (int)widths[f] - 1; b >= 0; b--) bits[pos++] = (vals[f] >> b) & 1u;
    while (pos < 80) bits[pos++] = 1;
#define GRAIN_STEP(nb)                                                                   \
    do {                                                                                 \
        nb = bits[62] ^ bits[51] ^ bits[38] ^ bits[23] ^ bits[13] ^ bits[0];             \
        memmove(bits, bits + 1, 79);                                                     \
        bits[79] = (unsigned char)nb;                                                    \
    } while (0)
    unsigned nb;
    for (int i = 0; i < 160; i++) GRAIN_STEP(nb);
    for (int k = 0; k < N_RC; k++) {
        uint32_t v;
        do {
            v = 0;
            for (int i = 0; i < 31; i++) {
                for (;;) {
                    GRAIN_STEP(nb);
                    if (nb) { GRAIN_STEP(nb); break; }
                    GRAIN_STEP(nb);
                }
                v = (v << 1) | nb;
            }
        } while (v >= P);
        out[k] = v;
    }
#undef GRAIN_STEP
}
/* Internal-layer diagonal (matrix = 1·1^T + diag(d)): the published HorizenLabs BabyBear t=24 instance
 * (MAT_DIAG24_M_1), which upstream names M_INT_DIAG_HZN [EXT].  These 24 words are not derivable; their
 * correctness is established by the KAT (tests/golden/poseidon2_kat.json). */
static const uint32_t DIAG_HZN[BXO_CELLS] = {
    0x409133f0, 0x1667a8a1, 0x06a6c7b6, 0x6f53160e, 0x273b11d1, 0x03176c5d, 0x72f9bbf9, 0x73ceba91,
    0x5cdef81d, 0x01393285, 0x46daee06, 0x065d7ba6, 0x52d72d6f, 0x05dd05e0, 0x3bab4b63, 0x6ada3842,
    0x2fc5fbec, 0x770d61b0, 0x5715aae9, 0x03ef0e90, 0x75b6c770, 0x242adf5f, 0x00d0ca4c, 0x36c0e388};

void bxo_poseidon2_set_params(const uint32_t rc[213], const uint32_t diag[24]) {
    for (int i = 0; i < N_RC; i++) { g_rc_canon[i] = rc[i]; g_rc[i] = bxo_fp_encode(rc[i]); }
    for (int i = 0; i < BXO_CELLS; i++) { g_diag_canon[i] = diag[i]; g_diag[i] = bxo_fp_encode(diag[i]); }
}
static void poseidon2_default_params(void) {
    uint32_t rc[N_RC];
    grain_round_constants(rc);
    bxo_poseidon2_set_params(rc, DIAG_HZN);
}
void bxo_poseidon2_get_params(uint32_t rc[213], uint32_t diag[24]) {
    bxo_init();
    memcpy(rc, g_rc_canon, sizeof g_rc_canon);
    memcpy(diag, g_diag_canon, sizeof g_diag_canon);
}

/* [EXT] poseidon2/mod.rs sbox2: x^7 */
static inline uint32_t sbox(uint32_t x) {
    uint32_t x2 = fmul(x, x), x4 = fmul(x2, x2), x6 = fmul(x4, x2);
    return fmul(x6, x);
}
/* [EXT] poseidon2/mod.rs multiply_by_4x4_circulant: M4 = [[5,7,1,3],[4,6,1,1],[1,3,5,7],[1,1,4,6]] */
static inline void m4(uint32_t* x) {
    uint32_t t0 = fadd(x[0], x[1]), t1 = fadd(x[2], x[3]);
    uint32_t t2 = fadd(fadd(x[1], x[1]), t1), t3 = fadd(fadd(x[3], x[3]), t0);
    uint32_t t1_2 = fadd(t1, t1), t0_2 = fadd(t0, t0);
    uint32_t t4 = fadd(fadd(t1_2, t1_2), t3), t5 = fadd(fadd(t0_2, t0_2), t2);
    uint32_t t6 = fadd(t3, t5), t7 = fadd(t2, t4);
    x[0] = t6; x[1] = t5; x[2] = t7; x[3] = t4;
}
/* [EXT] multiply_by_m_ext: circ(2·M4, M4, …, M4) */
static void m_ext(uint32_t* s) {
    uint32_t sums[4] = {0, 0, 0, 0};
    for (int i = 0; i < BXO_CELLS; i += 4) {
        m4(s + i);
        for (int j = 0; j < 4; j++) sums[j] = fadd(sums[j], s[i + j]);
    }
    for (int i = 0; i < BXO_CELLS; i++) s[i] = fadd(s[i], sums[i & 3]);
}
/* [EXT] multiply_by_m_int: cells[i] = sum + diag[i]*cells[i] */
static void m_int(uint32_t* s) {
    uint32_t sum = 0;
    for (int i = 0; i < BXO_CELLS; i++) sum = fadd(sum, s[i]);
    for (int i = 0; i < BXO_CELLS; i++) s[i] = fadd(sum, fmul(g_diag[i], s[i]));
}
/* [EXT] poseidon2/mod.rs poseidon2_mix: initial M_E, 4 full, 21 partial, 4 full. */
static void p2_mix(uint32_t* s) {
    const uint32_t* rc = g_rc;
    m_ext(s);
    for (int r = 0; r < ROUNDS_HALF_FULL; r++) {
        for (int i = 0; i < BXO_CELLS; i++) s[i] = sbox(fadd(s[i], rc[i]));
        rc += BXO_CELLS;
        m_ext(s);
    }
    for (int r = 0; r < ROUNDS_PARTIAL; r++) {
        s[0] = sbox(fadd(s[0], *rc++));
        m_int(s);
    }
    for (int r = 0; r < ROUNDS_HALF_FULL; r++) {
        for (int i = 0; i < BXO_CELLS; i++) s[i] = sbox(fadd(s[i], rc[i]));
        rc += BXO_CELLS;
        m_ext(s);
    }
}
void bxo_poseidon2_mix(uint32_t cells[24]) { bxo_init(); p2_mix(cells); }

/* [EXT] poseidon2/mod.rs unpadded_hash: overwrite-mode sponge, rate 16, zero pad, out = cells[0..8)
 * (digest words are the Montgomery u32 of each element). `stride` lets a matrix row be hashed in place. */
void bxo_hash_elem_slice(uint32_t digest[8], const uint32_t* elems, size_t n, size_t stride) {
    bxo_init();
    uint32_t s[BXO_CELLS];
    memset(s, 0, sizeof s);
    size_t unmixed = 0;
    for (size_t i = 0; i < n; i++) {
        s[unmixed++] = elems[i * stride];
        if (unmixed == BXO_RATE) { p2_mix(s); unmixed = 0; }
    }
    if (unmixed != 0 || n == 0) {
        for (size_t i = unmixed; i < BXO_RATE; i++) s[i] = 0;
        p2_mix(s);
    }
    memcpy(digest, s, 32);
}
/* [EXT] Poseidon2HashFn::hash_pair */
void bxo_hash_pair(uint32_t out[8], const uint32_t a[8], const uint32_t b[8]) {
    bxo_init();
    uint32_t s[BXO_CELLS];
    memcpy(s, a, 32);
    memcpy(s + 8, b, 32);
    memset(s + 16, 0, 32);
    p2_mix(s);
    memcpy(out, s, 32);
}
/* [EXT] hal/cpu.rs hash_rows: out[r] = hash of row r of a column-major rows x cols matrix */
void bxo_hash_rows(uint32_t* out, const uint32_t* matrix, size_t rows, size_t cols) {
    bxo_init();
    PAR_FOR
    for (size_t r = 0; r < rows; r++) bxo_hash_elem_slice(out + 8 * r, matrix + r, cols, rows);
}
/* [EXT] hal/cpu.rs hash_fold: io[out+i] = H(io[in+2i], io[in+2i+1]) */
void bxo_hash_fold(uint32_t* io, size_t input_size, size_t output_size) {
    bxo_init();
    PAR_FOR
    for (size_t i = 0; i < output_size; i++)
        bxo_hash_pair(io + 8 * (output_size + i), io + 8 * (input_size + 2 * i), io + 8 * (input_size + 2 * i + 1));
}

/* ------------------------------------------------------------------ FRI / DEEP */
/* [EXT] hal/cpu.rs fri_fold: SoA ext planes in (16*count each), SoA out (count each). */
void bxo_fri_fold(uint32_t* out, const uint32_t* in, const uint32_t mixw[4], size_t count) {
    bxo_init();
    fp4 mix = f4load(mixw);
    PAR_FOR
    for (size_t idx = 0; idx < count; idx++) {
        fp4 tot = f4zero(), cur = f4one();
        for (uint32_t i = 0; i < BXO_FRI_FOLD; i++) {
            size_t r = (size_t)bitrev(i, 4) * count + idx;
            fp4 f = {{in[r], in[BXO_FRI_FOLD * count + r], in[2 * BXO_FRI_FOLD * count + r],
                      in[3 * BXO_FRI_FOLD * count + r]}};
            tot = f4add(tot, f4mul(cur, f));
            cur = f4mul(cur, mix);
        }
        for (int k = 0; k < 4; k++) out[k * count + idx] = tot.c[k];
    }
}
/* [EXT] hal/cpu.rs mix_poly_coeffs: out (AoS ext) [combos[i]*count + idx] += mix_start*mix^i * in[i*count+idx] */
void bxo_mix_poly_coeffs(uint32_t* out, const uint32_t mix_start[4], const uint32_t mixw[4], const uint32_t* in,
                         const uint32_t* combos, size_t input_size, size_t count) {
    bxo_init();
    fp4 mix = f4load(mixw), start = f4load(mix_start);
    PAR_FOR
    for (size_t idx = 0; idx < count; idx++) {
        fp4 cur = start;
        for (size_t i = 0; i < input_size; i++) {
            uint32_t* o = out + 4 * ((size_t)combos[i] * count + idx);
            fp4 acc = f4add(f4load(o), f4scale(cur, in[i * count + idx]));
            memcpy(o, acc.c, 16);
            cur = f4mul(cur, mix);
        }
    }
}
/* [EXT] hal/cpu.rs batch_evaluate_any + core/poly.rs poly_eval (natural-order coefficients) */
void bxo_batch_evaluate_any(const uint32_t* coeffs, size_t poly_size, const uint32_t* which, const uint32_t* xs,
                            uint32_t* out, size_t eval_count) {
    bxo_init();
    PAR_FOR
    for (size_t e = 0; e < eval_count; e++) {
        const uint32_t* c = coeffs + (size_t)which[e] * poly_size;
        fp4 x = f4load(xs + 4 * e), mul_x = f4one(), tot = f4zero();
        for (size_t i = 0; i < poly_size; i++) {
            tot = f4add(tot, f4scale(mul_x, c[i]));
            mul_x = f4mul(mul_x, x);
        }
        memcpy(out + 4 * e, tot.c, 16);
    }
}
void bxo_eltwise_add(uint32_t* out, const uint32_t* a, const uint32_t* b, size_t n) {
    for (size_t i = 0; i < n; i++) out[i] = fadd(a[i], b[i]);
}
/* [EXT] hal/cpu.rs eltwise_sum_extelem: out (SoA planes, count each) = sum over to_add AoS ext blocks */
void bxo_eltwise_sum_extelem(uint32_t* out, const uint32_t* in, size_t count, size_t to_add) {
    PAR_FOR
    for (size_t idx = 0; idx < count; idx++) {
        fp4 tot = f4zero();
        for (size_t j = 0; j < to_add; j++) tot = f4add(tot, f4load(in + 4 * (j * count + idx)));
        for (int k = 0; k < 4; k++) out[k * count + idx] = tot.c[k];
    }
}
/* [EXT] eltwise_zeroize_elem: INVALID (0xffffffff) marker cells -> 0, valid cells unchanged */
void bxo_eltwise_zeroize(uint32_t* io, size_t n) {
    for (size_t i = 0; i < n; i++)
        if (io[i] == 0xffffffffu) io[i] = 0;
}
/* [EXT] hal/cpu.rs gather_sample */
void bxo_gather_sample(uint32_t* dst, const uint32_t* src, size_t idx, size_t size, size_t stride) {
    for (size_t i = 0; i < size; i++) dst[i] = src[idx + i * stride];
}
/* [EXT] core/poly.rs poly_divide: in-place synthetic division by (x - z); returns remainder. */
int bxo_poly_divide(uint32_t* poly, size_t size, const uint32_t zw[4], uint32_t rem_out[4]) {
    bxo_init();
    fp4 z = f4load(zw), cur = f4zero();
    for (size_t i = size; i-- > 0;) {
        fp4 next = f4add(f4mul(z, cur), f4load(poly + 4 * i));
        memcpy(poly + 4 * i, cur.c, 16);
        cur = next;
    }
    if (rem_out) memcpy(rem_out, cur.c, 16);
    return (cur.c[0] | cur.c[1] | cur.c[2] | cur.c[3]) == 0;
}

/* [EXT] hal/cpu.rs prefix_products: io[i] = io[i] * io[i-1] (inclusive running product of ext elements, AoS) */
void bxo_prefix_products(uint32_t* io, size_t n) {
    bxo_init();
    for (size_t i = 1; i < n; i++) {
        fp4 r = f4mul(f4load(io + 4 * i), f4load(io + 4 * (i - 1)));
        memcpy(io + 4 * i, r.c, 16);
    }
}
/* [EXT] hal/cpu.rs scatter: for cycle c, entries index[c] .. index[c+1] write into[offsets[e]] = values[e] */
void bxo_scatter(uint32_t* into, const uint32_t* index, const uint32_t* offsets, const uint32_t* values, size_t cycles) {
    for (size_t c = 0; c < cycles; c++)
        for (uint32_t e = index[c]; e < index[c + 1]; e++) into[offsets[e]] = values[e];
}
