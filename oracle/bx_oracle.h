/*
 * bx_oracle.h — CPU ORACLE (TEST INFRASTRUCTURE ONLY) for the segment-prove hot path.
 *
 * This is a plain-C restatement of the algorithms that the reference reaches through
 *   agent.prover.prove_segment(...)            bento/crates/workflow/src/tasks/prove.rs:41-49
 *   get_prover_server(&ProverOpts::default())  bento/crates/workflow/src/lib.rs:246-249
 * i.e. the `risc0_zkp::hal::Hal` CPU implementation.  That arithmetic is NOT in /root/reference:
 * it lives in the un-vendored crates pinned by the reference's lockfile
 *   risc0-zkp 3.0.3 (Cargo.lock:9155), risc0-core 3.0.0 (Cargo.lock:9012).
 * The restatement therefore follows their *published* algorithms (BabyBear Montgomery field,
 * recursive DIF/DIT NTT, Poseidon2 t=24 sponge, FRI fold-by-16), as summarised in SURVEY.md
 * Appendix A.
 *
 * PARITY STATUS
 *   - Poseidon2 permutation, rate-16 overwrite sponge and pair hash: PINNED to a vector the reference holds —
 *     compute_image_id(crates/povw/elfs/boundless-povw-log-updater.bin) == ...log-updater.iid, the reference's own test
 *     (crates/povw/src/log_updater.rs:383-388), reproduced by bx_oracle_image.c (tests/test_image_id_cpu.py).  Also the
 *     published BabyBear t=24 known-answer test (tests/golden/poseidon2_kat.json).  Round constants are *derived*
 *     (Poseidon Grain-LFSR generator), not typed in.
 *   - Field constants / roots of unity: pinned numerically (tests/golden/babybear_consts.json).
 *   - NTT ordering / FRI-fold indexing / Merkle top-layer rule / transcript random_bits / seal bytes vs. the Rust CPU prover:
 *     PARITY UNPINNED — the reference holds no vectors for them and cannot be built here.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use this library.
 * The product (boundless_amd/) never links, imports or calls it.
 *
 * All field elements cross this API as u32 words in Montgomery form (R = 2^32), exactly the
 * in-memory representation of risc0_core::field::baby_bear::Elem.
 */
#ifndef BX_ORACLE_H
#define BX_ORACLE_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define BXO_P 2013265921u
#define BXO_CELLS 24
#define BXO_RATE 16
#define BXO_DIGEST_WORDS 8
#define BXO_EXT 4
#define BXO_FRI_FOLD 16

void bxo_init(void); /* idempotent; builds root tables + Poseidon2 constants */
void bxo_set_threads(int n); /* OpenMP threads for batch ops (0 = all cores) */
int bxo_get_threads(void);

/* ---- field ---- */
uint32_t bxo_fp_encode(uint32_t canonical);
uint32_t bxo_fp_decode(uint32_t mont);
uint32_t bxo_fp_add(uint32_t a, uint32_t b);
uint32_t bxo_fp_sub(uint32_t a, uint32_t b);
uint32_t bxo_fp_mul(uint32_t a, uint32_t b);
uint32_t bxo_fp_pow(uint32_t a, uint64_t e);
uint32_t bxo_fp_inv(uint32_t a);
void bxo_fp4_mul(uint32_t out[4], const uint32_t a[4], const uint32_t b[4]);
void bxo_fp4_inv(uint32_t out[4], const uint32_t a[4]);
uint32_t bxo_rou_fwd(unsigned k); /* Montgomery form of ROU_FWD[k], order 2^k */
uint32_t bxo_rou_rev(unsigned k);

/* ---- NTT family (column-major: poly c occupies [c*size,(c+1)*size)) ---- */
void bxo_batch_interpolate_ntt(uint32_t* io, size_t count, size_t size);
void bxo_batch_evaluate_ntt(uint32_t* io, size_t count, size_t size, unsigned expand_bits);
void bxo_batch_expand_into_evaluate_ntt(uint32_t* out, const uint32_t* in, size_t count,
                                        size_t in_size, unsigned expand_bits);
void bxo_batch_bit_reverse(uint32_t* io, size_t count, size_t size);
void bxo_zk_shift(uint32_t* io, size_t count, size_t size);

/* ---- Poseidon2 (BabyBear, t=24, rate 16, 8 full + 21 partial rounds, x^7) ---- */
void bxo_poseidon2_get_params(uint32_t rc_canonical[213], uint32_t diag_canonical[24]);
void bxo_poseidon2_set_params(const uint32_t rc_canonical[213], const uint32_t diag_canonical[24]);
void bxo_poseidon2_mix(uint32_t cells[24]);
void bxo_hash_elem_slice(uint32_t digest[8], const uint32_t* elems, size_t n, size_t stride);
void bxo_hash_pair(uint32_t out[8], const uint32_t a[8], const uint32_t b[8]);
void bxo_hash_rows(uint32_t* out_digests, const uint32_t* matrix, size_t rows, size_t cols);
void bxo_hash_fold(uint32_t* io_digests, size_t input_size, size_t output_size);

/* ---- FRI / DEEP helpers ---- */
void bxo_fri_fold(uint32_t* out, const uint32_t* in, const uint32_t mix[4], size_t out_count);
void bxo_mix_poly_coeffs(uint32_t* out_ext, const uint32_t mix_start[4], const uint32_t mix[4],
                         const uint32_t* in, const uint32_t* combos, size_t input_size,
                         size_t count);
void bxo_batch_evaluate_any(const uint32_t* coeffs, size_t poly_size, const uint32_t* which,
                            const uint32_t* xs_ext, uint32_t* out_ext, size_t eval_count);
void bxo_eltwise_add(uint32_t* out, const uint32_t* a, const uint32_t* b, size_t n);
void bxo_eltwise_sum_extelem(uint32_t* out, const uint32_t* in_ext, size_t count, size_t to_add);
void bxo_eltwise_zeroize(uint32_t* io, size_t n);
void bxo_gather_sample(uint32_t* dst, const uint32_t* src, size_t idx, size_t size,
                       size_t stride);
/* DEEP quotient: combo (SoA ext planes? no: AoS ext, `size` elems) <- (combo - value)/(x - z),
 * synthetic division, coefficients in natural order. Returns the remainder-is-zero flag. */
int bxo_poly_divide(uint32_t* poly_ext, size_t size, const uint32_t z[4], uint32_t rem_out[4]);

void bxo_prefix_products(uint32_t* io_ext, size_t n);
void bxo_scatter(uint32_t* into, const uint32_t* index, const uint32_t* offsets, const uint32_t* values, size_t cycles);

/* ---- segment-prover pipeline (bx_oracle_prover.c): returns a malloc'ed seal (free with bxo_free) or NULL ---- */
uint32_t* bxo_prove_segment(uint32_t po2, uint32_t w_code, uint32_t w_data, uint32_t w_accum, uint64_t seed,
                            size_t* seal_words, uint32_t roots_out[32]);
/* same with the circuit's knobs: product terms per derived-column constraint and factors per term (0 = defaults) */
uint32_t* bxo_prove_segment_ex(uint32_t po2, uint32_t w_code, uint32_t w_data, uint32_t w_accum, uint32_t terms, uint32_t degree,
                               uint64_t seed, size_t* seal_words, uint32_t roots_out[32]);
/* same with an explicit seed for the ZK noise rows (the last min(1994, N/4) rows of the free data columns; upstream draws them
 * from a thread RNG, so its seals differ run to run — SURVEY.md section 7, hard part 3).  bxo_prove_segment_ex derives
 * noise_seed = splitmix64(seed ^ 0x5A4B4E4F49534521). */
uint32_t* bxo_prove_segment_zk(uint32_t po2, uint32_t w_code, uint32_t w_data, uint32_t w_accum, uint32_t terms, uint32_t degree,
                               uint64_t seed, uint64_t noise_seed, size_t* seal_words, uint32_t roots_out[32]);
/* control ID of the synthetic circuit for (po2, w_code): Merkle root of the committed code group (include/bx_circuit.h) */
void bxo_control_id(uint32_t po2, uint32_t w_code, uint32_t id_out[8]);
/* the transcript's RNG alone ([EXT] Poseidon2Rng): state = 24 cells + the number of rate cells already handed out; `mix(digest)`
 * n_commit times, then `random_elem` n_elems times.  What the prover's iop_commit / iop_random_elem do, exported so that the
 * device-side step (bx_transcript_step) can be checked on its own. */
void bxo_transcript_step(uint32_t state[25], const uint32_t* digests, size_t n_commit, uint32_t* out, size_t n_elems);
uint32_t bxo_rng_random_bits(uint32_t state[25], unsigned bits);
/* test hook: add 1 to witness cell (group, col, row) before it is committed, making the proved statement false (group < 0: off) */
void bxo_set_witness_fault(int group, uint32_t col, uint32_t row);
/* test hook: a dishonest prover that commits its own code group (1: `last` == 0 and a false g_1; 2: `first` == 0, zero
 * accumulators and a false g_0; 0: honest).  Only a control-ID check of the code root can refuse such a seal. */
void bxo_set_cheat(int mode);
void bxo_free(void* p);

/* ---- program image (bx_oracle_image.c): risc0_zkvm::compute_image_id of an "R0BF" program binary.  PINNED by the
 * reference's own vector (crates/povw/src/log_updater.rs:383-388).  Returns 0 or a negative error code;
 * root_canonical (may be NULL) receives the Poseidon2 Merkle root of the memory image as canonical words. ---- */
int bxo_compute_image_id(const uint8_t* blob, size_t len, uint8_t id_out[32], uint32_t root_canonical[8]);
void bxo_sha256(uint8_t out[32], const uint8_t* msg, size_t len);

#ifdef __cplusplus
}
#endif
#endif
