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
 * --------------------------------------------
 * The rv32im circuit (witness generation, eval_check constraint polynomial, tap set) is machine-generated code in
 * crates the reference does not vendor, so a real zkVM segment cannot be built offline.  bx_prove_segment runs the
 * *circuit-independent* prover pipeline — every HAL kernel, the Poseidon2 Fiat–Shamir transcript, Merkle commits,
 * DEEP quotients, FRI rounds and the 50 queries, in upstream's order and with upstream's constants — over a
 * synthetic segment: witness columns are filled from a seed by a device kernel (stand-in for witgen) and the
 * check polynomial is a fixed cubic mix of the committed columns (stand-in for eval_check).  The seal is a
 * deterministic function of (params, seed) and is bit-identical to the CPU oracle's seal (oracle/bx_oracle_prover.c).
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

typedef struct bx_prover bx_prover;

/* Shape of a synthetic segment: 2^po2 cycles (rows) and the widths of the three trace groups.
 * BASELINE.json config[1]: po2 = 20, w_code = 16, w_data = 256, w_accum = 64. */
typedef struct bx_segment_params {
    uint32_t po2;
    uint32_t w_code;
    uint32_t w_data;
    uint32_t w_accum;
} bx_segment_params;

/* Allocates every device buffer the pipeline needs for this shape (nothing is allocated per proof). */
const char* bx_prover_create(bx_ctx* ctx, const bx_segment_params* shape, bx_prover** out);
const char* bx_prover_destroy(bx_prover* prover);
/* Upper bound of the seal length in u32 words for this shape. */
size_t bx_prover_seal_words(const bx_prover* prover);
/* Prove one synthetic segment identified by `seed`; writes the seal (u32 words) and its length. Blocks. */
const char* bx_prove_segment(bx_prover* prover, uint64_t seed, uint32_t* seal_out, size_t seal_cap,
                             size_t* seal_words);
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
 * opening, the DEEP quotient at each of the 50 query points and the FRI folding chain down to the final polynomial.
 * Returns NULL when the seal is accepted, otherwise a message (thread-local storage) naming the first failed check. */
const char* bx_verify_segment(const uint32_t* seal, size_t seal_words);

#ifdef __cplusplus
}
#endif
#endif
