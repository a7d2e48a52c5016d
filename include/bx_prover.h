/*
 * bx_prover.h — C ABI of the segment prover that sits directly under the agent's prove task.
 *
 * Reference boundary this replaces
 * --------------------------------
 *   bento/crates/workflow/src/tasks/prove.rs:41-49   let segment_receipt = prover.prove_segment(&ctx, &segment)
 *   bento/crates/workflow/src/lib.rs:192,246-249     Agent.prover: Option<Rc<dyn ProverServer>> built once per process
 *   prover/crates/workflow/src/tasks/prove.rs:44-52   (next-gen copy, same call)
 * `ProverServer::prove_segment` (risc0-zkvm 3.0.4, Cargo.lock:9187) drives risc0-zkp's `Prover`:
 * commit_group x3 -> eval_check -> check commit -> DEEP taps/mix/divide -> fri_prove -> seal (Vec<u32>).
 *
 * What is and is not reproduced
 * ---------------------------------------------
 * The rv32im circuit (witness generation, the eval_check constraint polynomial, the accumulate step, the tap set) is
 * machine-generated code in crates the reference does not vendor, so a real zkVM segment cannot be proved offline.
 * bx_prove_segment runs upstream's prover pipeline — every HAL kernel, the Poseidon2 Fiat-Shamir transcript, Merkle
 * commits, DEEP quotients, FRI rounds and the 50 queries, in upstream's order and with upstream's constants — over a
 * SYNTHETIC CIRCUIT of the same anatomy: a witness stage (free cells, scatter-placed permuted copies, derived columns),
 * an accumulate stage (grand products through Hal::prefix_products, keyed by a transcript challenge), and an eval_check
 * stage whose constraints really vanish on the trace domain and are divided by the vanishing polynomial.  The seal is a
 * STARK proof of that circuit: bx_verify_segment accepts it only if the constraint identity holds at the random point Z.
 * It is a deterministic function of (params, seed) and bit-identical to the CPU oracle's seal (oracle/bx_oracle_prover.c).
 * It is NOT a risc0 receipt: no image id and no claim (the ZK blinding rows are there, seeded).  Like a risc0 seal it is bound
 * to its circuit: the code group is a public function of the shape, its Merkle root is the circuit's CONTROL ID for that shape,
 * and bx_verify_segment refuses a seal whose code root is anything else (upstream's `check_code`).
 *
 * The synthetic circuit (normative; N = 2^po2 rows, all row indices cyclic mod N)
 * -------------------------------------------------------------------------------
 *   knobs      T = cons_terms (product terms per derived-column constraint), G = cons_degree (factors per term, <= 5)
 *   seeds      gseed_g = seed + (g+1) * 0x9E3779B97F4A7C15 (g = 1 data, 2 accum);  word(s,c,r) = splitmix64(s ^ (c << 32 | r)) >> 33, minus P if >= P
 *              cseed = 0x434F4E54524F4C21 ("CONTROL!"), a constant: the code group does not depend on the segment
 *   zk rows    Z = min(1994, N/4) (risc0_zkp::ZK_CYCLES = 1994 [EXT]); A = N - Z active rows.  Rows >= A of every free data column
 *              (the permuted copies included) are noise: word(nseed_1, c, r), nseed_g = noise_seed + (g+1) * 0x9E3779B97F4A7C15.
 *              Derived columns and accumulators are computed on them like on any row (their constraints hold on every row),
 *              so they are blinded through the free cells; the code group is public and carries no noise.
 *   code       column 0 = first (1 at row 0, else 0); column 1 = last (1 at row A-1, the last active row); column c >= 2 = word(cseed, c, r).
 *              A function of (po2, w_code) only.  control ID(po2, w_code) = Merkle root of its commitment (bx_circuit.h).
 *              csel(i) = code column 2 + i mod (w_code - 2) when w_code >= 3, else the constant 1.
 *   data       F = ceil(w_data / 2) free columns, J = w_data - F derived columns.
 *              free column c: word(gseed_1, c, r) on the active rows, except the permuted copies: for pair p < pairs and r < A,
 *                  data[4p+3][perm_p(r)] = data[4p+2][r],  perm_p(r) = (r * 2654435761 + 12345 + p) mod A
 *              (placed with Hal::scatter).  pairs = the number of p with 2p+1 < E and 4p+3 < F (0 when w_code < 2).
 *              derived column F+j: with the 16-entry pool
 *                  pool_j = [ u = data[j][r],  ub = data[j][r-1] if j % 8 == 0, data[j][r-2] if j % 8 == 4, else u,  data[(j+1) mod F][r],  data[(j+2) mod F][r],
 *                             p_1 .. p_8 with p_s = data[F+j-s][r] (csel(s-j-1)[r] when j < s),  csel(j)[r] .. csel(j+3)[r] ]
 *                  data[F+j][r] = sum_{t<T} prod_{f<G} pool_j[idx(t,f)],  idx(t,f) = (7t + 3f + floor(t/4) f + floor(t/16)) mod 16
 *   accum      drawn after the data commit: beta (ext).  E = floor(w_accum / 4) ext accumulators, accumulator e in columns
 *              4e..4e+3 (component k in column 4e+k):  acc_e(r) = prod_{i<=r} (beta_e + data[src(e)][i]),  beta_e = beta^(floor(e/2)+1)   (Hal::prefix_products)
 *              src(e) = 4p+2 / 4p+3 for e = 2p / 2p+1 when p < pairs, else e mod F.  Columns >= 4E: word(gseed_2 ^ (beta.c0<<32|beta.c1), c, r).
 *   taps       every column at Z; data columns c % 8 == 0 and the accumulator columns (c < 4E) also one row back (Z * w_N^-1); data
 *              columns c % 8 == 4 one and two rows back.  Columns with the same tap set share a DEEP combination polynomial.
 *   constraints, in mixing order (constraint i is weighted poly_mix^i):
 *              j < J :  data[F+j][r] - sum_t prod_f pool_j[idx(t,f)]                                          = 0
 *              e < E :  acc_e(r) - (first(r) + (1 - first(r)) * acc_e(r-1)) * (beta_e + data[src(e)][r])        = 0
 *              p < pairs :  last(r) * (acc_{2p+1}(r) - acc_{2p}(r))                                             = 0
 *              first(r) * (data[0][r] - g_0) = 0   and, when w_code >= 2,   last(r) * (data[w_data-1][r] - g_1) = 0
 *   globals    the statement's public words g_0 = data[0][0], g_1 = data[w_data-1][A-1] (Montgomery words): written to the seal
 *              right after the header and bound into the transcript before the first commitment; the two boundary constraints
 *              above tie them to the trace, so the seal proves "a trace of this circuit that starts at g_0 ends at g_1".
 *   check      check(x) = sum_i poly_mix^i C_i(x) / ((3x)^N - 1), evaluated on the 4N domain x = w_4N^row from the
 *              committed evaluations (which are F(3x): the coset shift lives in the coefficients), then split into the
 *              16 check columns exactly as upstream splits its check polynomial.
 *
 * Threading/ownership follow the reference: one prover per ctx (= per GPU, compose.yml:113), one call at a time,
 * blocking, errors returned as strings (never abort: bento/crates/workflow/src/lib.rs:381-436 retries on Err).
 */
#ifndef BX_PROVER_H
#define BX_PROVER_H
#include "bx_hal.h"
#ifdef __cplusplus
extern "C" {
#endif
/* the library is built with -fvisibility=hidden: only what these headers declare is exported */
#if defined(__GNUC__)
#pragma GCC visibility push(default)
#endif

typedef struct bx_prover bx_prover;

/* Shape of a synthetic segment: 2^po2 cycles (rows) and the widths of the three trace groups.
 * BASELINE.json config[1]: po2 = 20, w_code = 16, w_data = 256, w_accum = 64. */
typedef struct bx_segment_params {
    uint32_t po2;
    uint32_t w_code;
    uint32_t w_data;
    uint32_t w_accum;
    uint32_t cons_terms;  /* T: product terms per derived-column constraint; 0 = BX_CIRCUIT_DEFAULT_TERMS */
    uint32_t cons_degree; /* G: factors per term (multiplicative degree), 1..5; 0 = BX_CIRCUIT_DEFAULT_DEGREE */
} bx_segment_params;
#define BX_CIRCUIT_DEFAULT_TERMS 64
#define BX_CIRCUIT_DEFAULT_DEGREE 4
#define BX_CIRCUIT_MAX_TERMS 64
#define BX_CIRCUIT_MAX_DEGREE 5
#define BX_SEAL_HEADER_WORDS 6 /* po2, w_code, w_data, w_accum, T, G; the circuit's public words (globals) follow */

/* Allocates every device buffer the pipeline needs for this shape (nothing is allocated per proof). */
const char* bx_prover_create(bx_ctx* ctx, const bx_segment_params* shape, bx_prover** out);
const char* bx_prover_destroy(bx_prover* prover);
/* Upper bound of the seal length in u32 words for this shape. */
size_t bx_prover_seal_words(const bx_prover* prover);
/* ---- the segment on the wire ----
 * The reference moves bincode(risc0_zkvm::Segment) (tasks/mod.rs:40-47; ~80 MB for a 2^20-cycle segment, executor.rs:45); that
 * layout needs risc0's types.  The synthetic circuit's segment is the tagged stand-in
 *   "BXSYNSEG" | index u64 | po2 u32 | seed u64 | payload bytes ...            (28 bytes + payload, little endian)
 * whose payload (any length, may be empty) stands for the preflight trace: it crosses PCIe like one and is handed to the
 * circuit's witgen on the host and in HBM, but the synthetic witness is a function of `seed` alone. */
#define BX_SEGMENT_WIRE_BYTES 28
#define BX_SEGMENT_MAGIC "BXSYNSEG"
void bx_segment_encode(uint64_t index, uint32_t po2, uint64_t seed, uint8_t out[BX_SEGMENT_WIRE_BYTES]);
/* error: "Failed to deserialize segment data from redis ..." (prove.rs:36-37); len >= 28, the payload is not looked at */
const char* bx_segment_decode(const uint8_t* blob, size_t len, uint64_t* index, uint32_t* po2, uint64_t* seed);

/* `ProverServer::prove_segment(&ctx, &segment)`: prove the segment whose serialized bytes are given (for the built-in circuit:
 * the stand-in above; a plug-in circuit defines its own).  The proof starts with what does not need the bytes — the code group and
 * its whole commitment are enqueued first — and meanwhile the bytes are copied to pinned staging memory and uploaded on the prover's
 * copy stream in 8 MB pieces (2.3 ms of host copy and 1.4 ms of DMA for 80 MB, pipelined, hidden behind the 2.2 ms code commitment);
 * witgen, which receives them on the host and in HBM, waits for the upload's event on the stream.  The seal (u32 words) and its
 * length are written.  Blocks.  (Page-locking the caller's buffer in place instead of copying it — hipHostRegister — is cheaper
 * in isolation, 0.4 ms, and was tried: with fresh buffers and three lanes it stalls every lane of the process for the duration of
 * the driver call, 24.5 -> 18.6 proofs/s through the agent) */
const char* bx_prove_segment_bytes(bx_prover* prover, const uint8_t* segment, size_t segment_len, uint32_t* seal_out, size_t seal_cap,
                                   size_t* seal_words);
/* The same in two steps, two deep ("H2D of next segment overlapped with compute"): submit copies the bytes
 * into one of the prover's two pinned staging slots and enqueues the upload on the copy stream (returns once the host copy is
 * made; the caller's buffer is free again); prove_submitted proves the oldest submitted segment — its compute stream waits on the
 * upload's event, not the host.  At most two segments may be outstanding ("staging slots busy").  submit may be called from
 * another thread while prove_submitted runs: segment k+1 goes up while segment k is proved. */
const char* bx_prover_submit_segment(bx_prover* prover, const uint8_t* segment, size_t segment_len);
const char* bx_prove_submitted(bx_prover* prover, uint32_t* seal_out, size_t seal_cap, size_t* seal_words);
/* Device time (HIP events on the copy stream) and size of the upload of the segment proved last. */
const char* bx_prover_last_upload(const bx_prover* prover, double* ms, size_t* bytes);
/* Convenience: the built-in stand-in segment (index 0, the prover's po2, `seed`, no payload) through bx_prove_segment_bytes. */
const char* bx_prove_segment(bx_prover* prover, uint64_t seed, uint32_t* seal_out, size_t seal_cap,
                             size_t* seal_words);
/* The same with an explicit generator for the ZK noise cells (the last min(1994, N/4) rows of the free data columns of the
 * built-in circuit).  Upstream draws them from a thread RNG, so its seals are not reproducible; here (seed, noise_seed) fixes the
 * seal.  bx_prove_segment uses noise_seed = splitmix64(seed ^ 0x5A4B4E4F49534521).  Two noise seeds give two different seals of
 * the same statement (same header and public words), both accepted by bx_verify_segment. */
const char* bx_prove_segment_zk(bx_prover* prover, uint64_t seed, uint64_t noise_seed, uint32_t* seal_out, size_t seal_cap,
                                size_t* seal_words);
/* bx_circuit_ops::set_noise_seed for the next proof, for callers of the bytes entry points. */
const char* bx_prover_set_noise_seed(bx_prover* prover, uint64_t noise_seed);
/* Merkle root (8 words each) of the code, data, accum and check groups of the last proof. */
const char* bx_prover_last_roots(const bx_prover* prover, uint32_t roots_out[32]);

/* MerkleTreeProver::prove for a batch of queries: for query q, out[q*(cols + 8*depth) ...] receives the `cols`
 * column values of row positions[q] followed by the `depth` sibling digests from the leaf layer up to (excluding)
 * the layer of `top_size` nodes. positions is a device u32 buffer. */
const char* bx_merkle_query_gather(bx_ctx* ctx, bx_buf out, bx_buf matrix, bx_buf nodes_digests, size_t rows,
                                   size_t cols, bx_buf positions_u32, size_t n_queries, size_t top_size);

/* CPU verifier of a seal produced by bx_prove_segment (the reference verifies every receipt right after proving it:
 * bento/crates/workflow/src/tasks/prove.rs:53-55 `segment_receipt.verify_integrity_with_context`).  Pure host code, no
 * ctx and no GPU needed.  Replays the Poseidon2 transcript, checks the check-polynomial identity at Z, every Merkle
 * opening, the DEEP quotient at each of the 50 query points and the FRI folding chain down to the final polynomial, and
 * compares the code group's root with the circuit's control ID for the seal's shape (upstream: `check_code`; the built-in
 * circuit knows its IDs — a table for w_code = 16, a cached host computation otherwise; bx_circuit.h has the explicit
 * VerifierContext form).  Returns NULL when the seal is accepted, otherwise a message (thread-local storage) naming the first failed check. */
const char* bx_verify_segment(const uint32_t* seal, size_t seal_words);
/* Host threads that share the 50 queries of ONE verification (each query is independent once the transcript has been replayed):
 * 0 = the default — BX_VERIFY_THREADS from the environment, else min(4, cores); 1 = the calling thread alone.  The verdict and its
 * text do not depend on it (the first failing query in seal order decides).  A 2^18 seal: 9 ms on one thread, 3 ms on four — the
 * verification of a join's result is on the critical path of a job's join tail. */
const char* bx_verify_set_threads(int threads);

#if defined(__GNUC__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif
