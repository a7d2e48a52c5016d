/*
 * bx_circuit.h — the circuit half of the segment prover as a plug-in table: the C ABI counterpart of
 * `risc0_zkp::hal::CircuitHal` plus the witness-generation entry of the circuit's prover.
 *
 * Reference boundary
 * ------------------
 *   bento/crates/workflow/src/tasks/prove.rs:41-49   prover.prove_segment(&ctx, &segment)
 * reaches, below `ProverServer`, two trait objects [EXT, SURVEY.md App. A.3]:
 *   risc0_zkp::hal::Hal          circuit-independent kernels            -> include/bx_hal.h
 *   risc0_zkp::hal::CircuitHal   eval_check(check, groups, globals, poly_mix, po2, steps), accumulate(...)
 * and the circuit crate's witness generation (risc0-circuit-rv32im 4.0.3 `witgen` / `step_exec`, -sys 4.0.1 kernels,
 * reference Cargo.lock:8962,8996).  Those are machine-generated and not in the reference tree; this table is where they —
 * recompiled for gfx950 against fp.hpp — plug into bx_prove_segment, and where this repository's SYNTHETIC circuit
 * (bx_synthetic_circuit(), specified in bx_prover.h) plugs in today.  Everything else of the proof (commits, transcript,
 * DEEP, FRI, queries, seal layout) is circuit-independent and stays in the prover.
 *
 * Conventions as in bx_hal.h: device stages enqueue on the ctx's stream and return NULL or an error string; matrices are
 * column-major N x width (witness) or 4N x width (evaluations) of Montgomery u32 words; nothing aborts.
 */
#ifndef BX_CIRCUIT_H
#define BX_CIRCUIT_H
#include "bx_prover.h"
#ifdef __cplusplus
extern "C" {
#endif
/* the library is built with -fvisibility=hidden: only what these headers declare is exported */
#if defined(__GNUC__)
#pragma GCC visibility push(default)
#endif

#define BX_MAX_TAPS 8   /* back offsets per column */
#define BX_MAX_COMBOS 16 /* distinct tap sets ("combos", as in upstream's TapSet) over the three trace groups + the check group */
#define BX_MAX_GLOBALS 64 /* public words ("globals" / io of upstream's circuits) a segment's statement may carry */

/* Tap values handed to the verifier-side constraint evaluation: the value of column `col` of group `group` (0 code, 1 data,
 * 2 accum) at Z * w_N^-back, for a `back` that belongs to the column's tap set.  out = 4 words. */
typedef struct bx_tap_reader {
    const void* ctx;
    const char* (*at)(const void* ctx, int group, uint32_t col, int back, uint32_t out[4]);
} bx_tap_reader;

typedef struct bx_circuit_ops {
    void* user;
    const char* name;
    /* Fills in defaults and validates the circuit's knobs in *shape (cons_terms / cons_degree are the circuit's to interpret;
     * they travel in the seal header).  NULL = ok. */
    const char* (*normalize)(void* user, bx_segment_params* shape);
    /* Tap set of column `col` of trace group `group` (the tap set is circuit data: upstream reads it from the circuit's
     * TapSet): writes the row offsets the column is opened at — strictly increasing, backs_out[0] == 0 (every column is
     * opened at Z), back b meaning the point Z * w_N^-b — and returns their count, 1..BX_MAX_TAPS.  Columns with the same set
     * share a DEEP combination polynomial ("combo"); combos are numbered in order of first appearance over code, data,
     * accum, and the check group's combo comes last. */
    uint32_t (*taps)(void* user, const bx_segment_params* shape, int group, uint32_t col, uint32_t backs_out[BX_MAX_TAPS]);
    /* Number of public words of a segment's statement ("globals" in CircuitHal::eval_check: what the receipt's claim is made
     * of).  They are produced by witgen, written to the seal right after the header, bound into the transcript before any
     * commitment, and handed to eval_check / constraints_at, which tie them to the trace (boundary constraints). */
    uint32_t (*n_globals)(void* user, const bx_segment_params* shape);
    /* Per-prover device state of the circuit (tables, scratch).  Called once from bx_prover_create. */
    const char* (*create)(void* user, bx_ctx* ctx, const bx_segment_params* shape, void** state);
    void (*destroy)(void* user, void* state);
    /* The code ("control") group: the circuit's public columns (cycle selectors, control words).  A function of the shape
     * ALONE — never of the segment — because its Merkle root is the circuit's CONTROL ID for that shape: what upstream's
     * verifier compares the seal's code root with (`check_code`, risc0-zkp 3.0.3 verify; the table one level up is
     * contracts/src/blake3-groth16/ControlID.sol:13).  bx_prove_segment calls it before witgen; bx_prover_control_id runs it
     * through the same commit to produce the ID. */
    const char* (*code_group)(void* user, void* state, bx_ctx* ctx, bx_buf code);
    /* Witness generation for the data group from the SEGMENT'S BYTES — what the reference hands to
     * `prover.prove_segment(&ctx, &segment)` (bento/crates/workflow/src/tasks/prove.rs:41-49) after
     * `bincode::deserialize(&segment_vec)` (prove.rs:36-37; upstream: preflight trace -> witgen kernels).  `segment` is the host
     * copy (pinned memory owned by the prover, valid for the duration of the call), `segment_dev` the same bytes in HBM
     * (len = ceil(segment_len / 4) words, uploaded by the prover on its copy stream; the stream wait is already enqueued).
     * `code` was filled by code_group and must not be written.  The prover interpolates both buffers in place right afterwards,
     * so whatever `accumulate` needs of them — or of the segment — is kept by the circuit in its state. */
    const char* (*witgen)(void* user, void* state, bx_ctx* ctx, bx_buf code, bx_buf data, const uint8_t* segment, size_t segment_len,
                          bx_buf segment_dev, uint32_t* globals_out /* host, n_globals Montgomery words */);
    /* CircuitHal::accumulate: fills the accum group's witness; `mix` is the ext challenge drawn after the data commit. */
    const char* (*accumulate)(void* user, void* state, bx_ctx* ctx, bx_buf accum, const uint32_t mix[4]);
    /* CircuitHal::eval_check: the four ext planes (check.len = 16N words) of  sum_i poly_mix^i C_i(x) / ((3x)^N - 1)  over the
     * domain x = w_4N^row, from the committed 4N evaluations of the three trace groups.  `mix` as given to accumulate. */
    const char* (*eval_check)(void* user, void* state, bx_ctx* ctx, bx_buf check, bx_buf code_eval, bx_buf data_eval, bx_buf accum_eval,
                              const uint32_t poly_mix[4], const uint32_t mix[4], const uint32_t* globals /* host */);
    /* Verifier side (host, no GPU): sum_i poly_mix^i C_i evaluated from the tap values.  Upstream: the circuit's
     * `poly_ext` called by risc0_zkp::verify. */
    const char* (*constraints_at)(void* user, const bx_segment_params* shape, const bx_tap_reader* taps, const uint32_t poly_mix[4],
                                  const uint32_t mix[4], const uint32_t* globals, uint32_t out[4]);
    /* ZK blinding: the generator of the noise cells of the NEXT witgen/accumulate (upstream fills them from a thread RNG, which
     * is why its seals differ from run to run; here every random fill is seeded through the ABI — SURVEY.md section 7, hard part
     * 3).  Called by bx_prove_segment_zk before witgen.  May be NULL (a circuit without noise cells, or one that derives them
     * from the seed); tables that predate this member must zero-initialise it. */
    void (*set_noise_seed)(void* user, void* state, uint64_t noise_seed);
    /* Verifier side (host, no GPU), upstream's `check_code(po2, root)`: NULL when `root` — the Merkle root of the code group a
     * seal of this shape committed (the 8 digest words as they stand in the seal) — is this circuit's control ID for
     * the shape, else a message.  Without it the selectors and control words a seal's constraints are evaluated with would be
     * the prover's own choice.  Used when bx_verify_segment_with_context is given no explicit context; a table without it can
     * only be verified against an explicit bx_verifier_ctx. */
    const char* (*check_code)(void* user, const bx_segment_params* shape, const uint32_t root[8]);
} bx_circuit_ops;

/* The synthetic circuit of bx_prover.h ("The synthetic circuit"); what bx_prover_create / bx_verify_segment use. */
const bx_circuit_ops* bx_synthetic_circuit(void);

/* bx_prover_create with an explicit circuit (NULL = the synthetic one).  The table must outlive the prover. */
const char* bx_prover_create_with_circuit(bx_ctx* ctx, const bx_segment_params* shape, const bx_circuit_ops* circuit, bx_prover** out);
/* bx_verify_segment against an explicit circuit (NULL = the synthetic one); the code root is checked by circuit->check_code. */
const char* bx_verify_segment_with_circuit(const uint32_t* seal, size_t seal_words, const bx_circuit_ops* circuit);

/* The circuit's control ID for the prover's shape: the Poseidon2 Merkle root of the committed code group, computed on the
 * device with the kernels of a proof (code_group -> interpolate/zk_shift -> 4x LDE -> hash_rows -> tree).  Deterministic per
 * (circuit, shape).  Upstream ships these as a generated table (risc0-circuit-rv32im `control_id.rs`); an agent builds its
 * verifier context from this call at start-up (bento/crates/workflow/src/lib.rs:241 `verifier_ctx`).  Blocks; must not run
 * concurrently with a proof on the same prover. */
const char* bx_prover_control_id(bx_prover* prover, uint32_t id_out[8]);

/* VerifierContext (risc0-zkvm; built once per agent at bento/crates/workflow/src/lib.rs:241 and passed to
 * `segment_receipt.verify_integrity_with_context`, tasks/prove.rs:53-55): the set of control IDs a seal's code root may be.
 * Thread-safe for concurrent verification once filled; add is not concurrent with verify. */
typedef struct bx_verifier_ctx bx_verifier_ctx;
const char* bx_verifier_ctx_create(bx_verifier_ctx** out);
void bx_verifier_ctx_destroy(bx_verifier_ctx* v);
const char* bx_verifier_ctx_add_control_id(bx_verifier_ctx* v, uint32_t po2, const uint32_t id[8]);
size_t bx_verifier_ctx_size(const bx_verifier_ctx* v);
/* how many IDs the context holds for segments of 2^po2 cycles */
size_t bx_verifier_ctx_count(const bx_verifier_ctx* v, uint32_t po2);
/* bx_verify_segment with everything explicit.  circuit NULL = the synthetic one.  vctx non-NULL: the code root must be one of
 * the context's IDs for the seal's po2 (circuit->check_code is not consulted); vctx NULL: circuit->check_code decides, and a
 * circuit without one is refused ("no control IDs to check the code root against"). */
const char* bx_verify_segment_with_context(const uint32_t* seal, size_t seal_words, const bx_circuit_ops* circuit, const bx_verifier_ctx* vctx);
/* The synthetic circuit's control ID computed on the HOST (no GPU): the definition-level path (code columns -> interpolation ->
 * evaluation on the coset 3<w_4N> -> Poseidon2 rows -> tree) behind bx_synthetic_circuit()->check_code for shapes outside its
 * built-in table (w_code = 16, po2 9..24).  Seconds at po2 >= 18; results are cached per (po2, w_code). */
const char* bx_synthetic_control_id_host(uint32_t po2, uint32_t w_code, uint32_t id_out[8]);

#if defined(__GNUC__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif
