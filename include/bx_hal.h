/*
 * bx_hal.h — C ABI of libbx_hip_hal.so: the MI355X (gfx950) implementation of the kernel-level
 * interface that sits under the Bento prove agent's segment-prove hot path.
 *
 * Reference boundary this replaces
 * --------------------------------
 *   bento/crates/workflow/src/tasks/prove.rs:41-49   prover.prove_segment(&verifier_ctx, &segment)
 *   bento/crates/workflow/src/tasks/prove.rs:92-100  prover.lift(&segment_receipt)
 *   bento/crates/workflow/src/lib.rs:246-249         get_prover_server(&ProverOpts::default())
 *   blake3_groth16/src/prove/cuda.rs:59              risc0_zkp::hal::cuda::singleton()  (the only in-tree
 *                                                    touch of the HAL module)
 * Below `ProverServer` the reference reaches `risc0_zkp::hal::Hal` (risc0-zkp 3.0.3, Cargo.lock:9155) whose
 * CUDA implementation binds `extern "C"` kernels from risc0-sys 1.5.0 (Cargo.lock:9131) that return a
 * `const char*` error string (NULL = ok).  This header is the HIP counterpart of that `extern "C"` surface:
 * one entry point per `Hal` trait method the segment prover calls.  INTEGRATION.md shows the Rust
 * `impl Hal for HipHal` shim that binds it.
 *
 * Conventions
 * -----------
 *   - Every call returns NULL on success or a NUL-terminated message owned by the library (valid until the
 *     next call on the same ctx; for bx_init failures a static string).  No call aborts the process and no
 *     C++ exception crosses the ABI — the agent's retry machinery (bento/crates/workflow/src/lib.rs:381-436)
 *     relies on errors surfacing as values.
 *   - A ctx owns one HIP device + one stream.  Calls on one ctx are stream-ordered; only bx_d2h, bx_sync,
 *     bx_timer_stop, bx_profile_report and the prover entry points block the host.  A ctx is not
 *     thread-safe; distinct ctxs may be used from distinct threads.
 *   - All field elements are u32 words in Montgomery form (R = 2^32) over BabyBear P = 15*2^27 + 1 — the
 *     in-memory representation of risc0_core::field::baby_bear::Elem.  `bx_buf.len` counts u32 words.
 *     Buffer<ExtElem> is AoS (4 consecutive words); ext data held in a Buffer<Elem> is SoA (plane k at
 *     k*size).  Digests are 8 words.  Matrices are column-major (column c = [c*rows, (c+1)*rows)).
 *   - bx_buf is a plain (device pointer, length) pair, so memory owned by another allocator on the same
 *     device (e.g. a torch tensor's data_ptr) may be passed in.
 */
#ifndef BX_HAL_H
#define BX_HAL_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif
/* the library is built with -fvisibility=hidden: only what these headers declare is exported */
#if defined(__GNUC__)
#pragma GCC visibility push(default)
#endif

#define BX_P 2013265921u
#define BX_DIGEST_WORDS 8
#define BX_EXT_SIZE 4
#define BX_FRI_FOLD 16
#define BX_INV_RATE 4
#define BX_QUERIES 50
#define BX_FRI_MIN_DEGREE 256
#define BX_CHECK_SIZE 16
#define BX_POSEIDON2_RC_COUNT 213 /* 24*8 external + 21 internal */

typedef struct bx_ctx bx_ctx;
typedef struct bx_buf {
    void* dptr;
    size_t len; /* u32 words */
} bx_buf;

/* ---- context / memory (Hal::alloc_*, copy_from_*, Buffer::view) ---- */
const char* bx_init(int device, bx_ctx** out);
const char* bx_free(bx_ctx* ctx);
const char* bx_device_name(bx_ctx* ctx, char* out, size_t cap);
const char* bx_set_stream(bx_ctx* ctx, void* hip_stream); /* adopt an external hipStream_t (NULL = own stream) */
void* bx_get_stream(bx_ctx* ctx);
/* Hal::alloc_*: contents undefined.  bx_alloc / bx_release go through a per-ctx POOL: a released block is kept (up to the tunable
 * "alloc_cache_mb", default 16384; 0 = off) and handed to the next request of about its size — no driver call and no wait, every use
 * of the memory being ordered on the ctx's one stream.  risc0-zkp's prover allocates every buffer inside a proof and drops it at
 * the end; with raw hipMalloc / hipFree that costs 6 ms per lone 2^20 proof and, because hipFree drains the WHOLE device, takes
 * three provers in flight from 23.7 to 2.5-3.1 proofs/s in four runs of seven (20-21 in the other three: profiles/
 * r06_alloc_per_proof_bimodal.json).  Memory obtained here must be
 * returned with bx_release (not hipFree); bx_free returns the pool to the driver. */
const char* bx_alloc(bx_ctx* ctx, size_t words, bx_buf* out);
/* Hal::alloc_extelem_zeroed / alloc_elem_init(.., 0): an allocation whose words are 0, cleared on the ctx's stream (enqueued, like
 * every other call).  risc0-zkp's prover relies on it where it accumulates into a fresh buffer — `combos` in Prover::finalize is
 * filled by mix_poly_coeffs' `+=` — and Hal::eltwise_zeroize_elem is NOT a clear (it maps the INVALID marker 0xffffffff to 0 and
 * leaves every other word alone), so a trait-level caller has no other way to get zeros (tests/plain_hal_prover.c found this). */
const char* bx_alloc_zeroed(bx_ctx* ctx, size_t words, bx_buf* out);
/* Hal::alloc_elem_init(name, size, value): an allocation whose words all hold `value` (any 32-bit pattern: rv32im's witness generator
 * fills its buffers with the INVALID marker 0xffffffff and lets eltwise_zeroize_elem clear what was never written).  Filled on the
 * ctx's stream (hipMemsetD32Async), enqueued like every other call. */
const char* bx_alloc_init(bx_ctx* ctx, size_t words, uint32_t value, bx_buf* out);
/* Enqueue-only for pooled blocks (the block is reused behind everything already on THIS ctx's stream — work of another ctx or stream
 * that still uses the memory must have been waited for by the caller; hipFree's device-wide wait no longer hides that); blocks the
 * pool does not keep, and pointers it never handed out, are freed after a stream wait as before.  Releasing a block that already
 * idles in the pool (a double release) is an error, not a hipFree. */
const char* bx_release(bx_ctx* ctx, bx_buf buf);
const char* bx_h2d(bx_ctx* ctx, bx_buf dst, const uint32_t* src, size_t words);
const char* bx_d2h(bx_ctx* ctx, uint32_t* dst, bx_buf src, size_t words); /* blocks */
const char* bx_d2d(bx_ctx* ctx, bx_buf dst, bx_buf src, size_t words);
/* Hal::eltwise_copy_elem_slice(into, from: &[Elem], from_rows, from_cols, from_offset, from_stride, into_offset, into_stride)
 * ([EXT] risc0-zkp; the CUDA HAL uploads `from` and runs its `eltwise_copy_fp_region` kernel): a strided 2-D copy of a HOST slice
 * into a device buffer,   into[into_offset + r * into_stride + c] = from[from_offset + r * from_stride + c],   r < from_rows,
 * c < from_cols — how the prover places a witness of `steps` rows per column into buffers of 2^po2 rows per column.  One
 * hipMemcpy2DAsync on the ctx's stream; blocks like bx_h2d (`from` may be pageable and freed by the caller right after).
 * `from_len` = the slice's length in words.  Rows that would overlap in `into` (into_stride < from_cols) are refused. */
const char* bx_eltwise_copy_elem_slice(bx_ctx* ctx, bx_buf into, const uint32_t* from, size_t from_len, size_t from_rows, size_t from_cols,
                                       size_t from_offset, size_t from_stride, size_t into_offset, size_t into_stride);
const char* bx_sync(bx_ctx* ctx);
/* Hal::get_hash_suite (its name: "poseidon2", the reference's default hashfn) and Hal::has_unified_memory (0). */
const char* bx_hash_suite_name(void);
int bx_has_unified_memory(bx_ctx* ctx);

/* ---- Hal NTT family ---- */
/* Hal::batch_interpolate_ntt(io, count): `count` polys of size io.len/count, natural-order evaluations ->
 * bit-reversed coefficients, in place. */
const char* bx_batch_interpolate_ntt(bx_ctx* ctx, bx_buf io, size_t count);
/* Hal::batch_evaluate_ntt(io, count, expand_bits): bit-reversed coefficients -> natural-order evaluations,
 * in place, skipping the first expand_bits stages. */
const char* bx_batch_evaluate_ntt(bx_ctx* ctx, bx_buf io, size_t count, size_t expand_bits);
/* Hal::batch_expand_into_evaluate_ntt(out, in, count, expand_bits): out[i] = in[i >> bits] then evaluate. */
const char* bx_batch_expand_into_evaluate_ntt(bx_ctx* ctx, bx_buf out, bx_buf in, size_t count,
                                              size_t expand_bits);
/* Hal::batch_bit_reverse(io, count) */
const char* bx_batch_bit_reverse(bx_ctx* ctx, bx_buf io, size_t count);
/* Hal::zk_shift(io, count): io[i] *= 3^bitrev(i mod size) */
const char* bx_zk_shift(bx_ctx* ctx, bx_buf io, size_t count);

/* ---- Hal hash family (Poseidon2 suite: BabyBear t=24, rate 16, 8+21 rounds, x^7) ---- */
/* Replace the permutation parameters (canonical, non-Montgomery integers): 213 round constants laid out as
 * 4x24 external | 21 internal | 4x24 external, and the 24-entry internal diagonal (matrix = 1*1^T + diag).
 * The library default is the published BabyBear t=24 instance used by the reference's `poseidon2` hashfn.
 * Refused while a bx_prover exists on the ctx (its host transcript holds the table it was created with), and seals made
 * under a non-default table are not accepted by bx_verify_segment, which uses the default. */
const char* bx_poseidon2_set_params(bx_ctx* ctx, const uint32_t* rc213, const uint32_t* diag24);
const char* bx_poseidon2_get_params(bx_ctx* ctx, uint32_t* rc213, uint32_t* diag24);
/* The library's compiled-in default table, same layout; host only (no ctx, no GPU).  Its SHA-256 is pinned in
 * tests/golden/MANIFEST.json. */
const char* bx_poseidon2_default_params(uint32_t* rc213, uint32_t* diag24);
/* Hal::hash_rows(output, matrix): out_digests.len/8 rows; cols = matrix.len/rows. */
const char* bx_hash_rows(bx_ctx* ctx, bx_buf out_digests, bx_buf matrix);
/* Hal::hash_fold(io, input_size, output_size): io[out+i] = H(io[in+2i] || io[in+2i+1]), digest indices. */
const char* bx_hash_fold(bx_ctx* ctx, bx_buf io_digests, size_t input_size, size_t output_size);
/* MerkleTreeProver::new's loop in one call: nodes has 2*rows digests; hashes the rows of `matrix` into
 * nodes[rows..2rows) and folds every layer down to nodes[1]. */
const char* bx_merkle_build(bx_ctx* ctx, bx_buf nodes_digests, bx_buf matrix, size_t rows);
/* Extension: the fold half alone — the leaf digests are already in nodes[rows .. 2 rows). */
const char* bx_merkle_fold(bx_ctx* ctx, bx_buf nodes_digests, size_t rows);

/* ---- Hal FRI / DEEP family ---- */
/* Hal::fri_fold(output, input, mix): SoA planes; input.len = 16*output.len. `mix` = 4 host words. */
const char* bx_fri_fold(bx_ctx* ctx, bx_buf out, bx_buf in, const uint32_t mix[4]);
/* Extension: the same with `mix` in device memory (4 words), so that a challenge drawn on the device (bx_transcript_step) feeds
 * the fold without a host round trip. */
const char* bx_fri_fold_dev(bx_ctx* ctx, bx_buf out, bx_buf in, bx_buf mix_ext);
/* Extension: one step of the Fiat-Shamir transcript on the device (risc0_zkp Poseidon2Rng: `mix(digest)` n_commit times, then
 * `random_elem` 4 * n_ext times).  state = 25 words: the 24 sponge cells (Montgomery) and the number of rate cells already handed
 * out; digests = n_commit x 8 words (a Merkle root as the tree holds it: nodes[8..16)); out_ext receives the n_ext challenges.
 * Enqueued like everything else: with the tunable `dev_draws` = 1 the prover uses it where a challenge depends on nothing but a root
 * (the FRI rounds; measured no faster than the host round trips on MI355X, hence off by default), reads roots,
 * top layers and the drawn challenges back later in ONE copy, and replays the same steps on the host transcript (which also
 * writes the seal), checking that both sides drew the same words. */
const char* bx_transcript_step(bx_ctx* ctx, bx_buf state25, bx_buf digests, size_t n_commit, bx_buf out_ext, size_t n_ext);
/* Hal::mix_poly_coeffs(output, mix_start, mix, input, combos, input_size, count):
 * out_ext[combos[i]*count + idx] += mix_start*mix^i * in[i*count + idx], i < input_size, idx < count. */
const char* bx_mix_poly_coeffs(bx_ctx* ctx, bx_buf out_ext, const uint32_t mix_start[4],
                               const uint32_t mix[4], bx_buf in, bx_buf combos_u32, size_t input_size,
                               size_t count);
/* Hal::batch_evaluate_any(coeffs, poly_count, which, xs, out): out[i] = sum_j coeffs[which[i]*size+j]*xs[i]^j */
const char* bx_batch_evaluate_any(bx_ctx* ctx, bx_buf coeffs, size_t poly_count, bx_buf which_u32,
                                  bx_buf xs_ext, bx_buf out_ext);
/* Extensions for provers that keep coefficient polynomials in bit-reversed order (what batch_interpolate_ntt leaves
 * and batch_expand_into_evaluate_ntt reads), so that only the few DEEP combination polynomials are ever bit-reversed:
 * batch_evaluate_any over bit-reversed coefficient storage (polynomial size a power of two >= 2^15; same result as
 * batch_evaluate_any on the natural-order array), and batch_bit_reverse for AoS Buffer<ExtElem> (16-byte elements). */
const char* bx_batch_evaluate_any_bitrev(bx_ctx* ctx, bx_buf coeffs_bitrev, size_t poly_count, bx_buf which_u32,
                                         bx_buf xs_ext, bx_buf out_ext);
const char* bx_batch_bit_reverse_ext(bx_ctx* ctx, bx_buf io_ext, size_t count);
/* Extension: evaluations over several coefficient buffers in one launch set: evaluation i evaluates, at xs[i], the polynomial of
 * poly_size coefficients (a power of two in [2^15, 2^24]) at device address poly_ptrs[i] (a little-endian u64 in two words), stored
 * bit-reversed when flags[i] & 1.  The prover's DEEP step evaluates the taps of all four groups with one call. */
const char* bx_batch_evaluate_ptrs(bx_ctx* ctx, bx_buf poly_ptrs_u64, bx_buf flags_u32, size_t poly_size, bx_buf xs_ext, bx_buf out_ext);
/* Extension: batch_interpolate_ntt(io, count) followed by zk_shift(io, count) as one call; on the register-radix path the
 * shift rides on the final store of the inverse transform.  Same result as the two calls. */
const char* bx_batch_interpolate_zk(bx_ctx* ctx, bx_buf io, size_t count);
/* Hal::eltwise_add_elem / eltwise_copy_elem / eltwise_zeroize_elem / eltwise_sum_extelem */
const char* bx_eltwise_add_elem(bx_ctx* ctx, bx_buf out, bx_buf a, bx_buf b);
const char* bx_eltwise_copy_elem(bx_ctx* ctx, bx_buf out, bx_buf in);
const char* bx_eltwise_zeroize_elem(bx_ctx* ctx, bx_buf io);
const char* bx_eltwise_sum_extelem(bx_ctx* ctx, bx_buf out, bx_buf in_ext);
/* the CUDA HAL's eltwise_mul_factor_fp kernel (risc0-sys): io[i] *= factor, factor a Montgomery word */
const char* bx_eltwise_mul_factor(bx_ctx* ctx, bx_buf io, uint32_t factor_mont);
/* Hal::gather_sample(dst, src, idx, size, stride): dst[i] = src[idx + i*stride].  MerkleTreeProver::prove issues thousands of these
 * per proof (one per opened row, one per path digest), so a small gather is QUEUED and the queue is launched as one kernel by the
 * next call of any kind on the ctx (or bx_get_stream): stream order as the caller observes it is unchanged, the ~3.7 us of launch
 * cost per gather is paid once per batch.  Tunable "gather_defer" = 0 launches each gather at once; so does a ctx that runs on
 * an adopted stream (bx_set_stream), whose owner enqueues work there that this library does not see.  Work a caller enqueues
 * directly on the ctx's OWN stream must fetch it with bx_get_stream each time (which flushes), not from an earlier call. */
const char* bx_gather_sample(bx_ctx* ctx, bx_buf dst, bx_buf src, size_t idx, size_t size, size_t stride);
/* Hal::prefix_products(io): io[i] = io[i] * io[i-1] over AoS ext elements (inclusive running product; the circuit's
 * accumulate step uses it for its grand products). */
const char* bx_prefix_products(bx_ctx* ctx, bx_buf io_ext);
/* Extension: `count` independent sequences of io_ext.len/4/count ext elements each, laid out back to back, scanned in one
 * set of launches (the accumulate step runs one sequence per accumulator). */
const char* bx_batch_prefix_products(bx_ctx* ctx, bx_buf io_ext, size_t count);
/* Hal::scatter(into, index, offsets, values): for cycle c < index.len - 1, every entry e in [index[c], index[c+1])
 * writes into[offsets[e]] = values[e].  All four are device buffers.  Asynchronous: an offset outside `into` or an index
 * range outside offsets/values is detected on the device and reported by the next blocking call on the ctx (bx_d2h,
 * bx_sync), which returns the error string. */
const char* bx_scatter(bx_ctx* ctx, bx_buf into, bx_buf index_u32, bx_buf offsets_u32, bx_buf values);
/* DEEP quotient (upstream: core/poly.rs poly_divide, run per combo): in-place synthetic division of the
 * natural-order AoS ext polynomial by (x - z); the remainder (4 words) is written to rem_out_dev. */
const char* bx_poly_divide(bx_ctx* ctx, bx_buf poly_ext, const uint32_t z[4], bx_buf rem_out_dev);

/* Extension: `count` AoS ext polynomials back to back, polynomial q divided in place by (x - zs[q]) (zs: 4 host words per
 * polynomial), all in one launch; rem_out_dev receives 4 words per polynomial.  The prover divides every DEEP combination
 * polynomial by its next tap point with one call per round instead of one call per division. */
const char* bx_poly_divide_batch(bx_ctx* ctx, bx_buf polys_ext, size_t count, const uint32_t* zs, bx_buf rem_out_dev);
/* The same over a subset: the buffer holds n_polys polynomials, division q < count divides polynomial which[q] (host array, no
 * repeats) by (x - zs[q]) and writes its remainder to rem_out_dev[4q..]. */
const char* bx_poly_divide_batch_indexed(bx_ctx* ctx, bx_buf polys_ext, size_t n_polys, size_t count, const uint32_t* which, const uint32_t* zs,
                                         bx_buf rem_out_dev);

/* ---- measurement (HIP events on the ctx's stream) ---- */
const char* bx_timer_start(bx_ctx* ctx);
const char* bx_timer_stop(bx_ctx* ctx, float* ms_out); /* blocks */
/* Per-entry-point profiling: when enabled every HAL call is bracketed by hipEvents on the ctx stream and
 * accumulated by op name together with its algorithmic bytes. */
const char* bx_profile_enable(bx_ctx* ctx, int on);
const char* bx_profile_reset(bx_ctx* ctx);
const char* bx_profile_report(bx_ctx* ctx, char* json_out, size_t cap); /* blocks */
/* Tracing: roctx ranges around every HAL entry point (named as in bx_profile_report) and around the stages of
 * bx_prove_segment ("bx:witgen", "bx:commit_code", ... "bx:queries"), for `rocprofv3 --marker-trace --kernel-trace`.
 * [EXT] risc0's prover brackets its stages the same way (nvtx `scope!` ranges under its CUDA HAL); the reference's own
 * crates only log task-level `tracing` events (bento/crates/workflow/src/tasks/prove.rs:28,50-51,115).  Process-wide.  level 0 = off (default),
 * 1 = ranges, 2 = ranges + the ctx's stream drained at the end of every prover stage, so that a stage's host-side range is
 * its device time (costs the overlap between stages: a measurement mode).  The roctx library is dlopen'ed on first use;
 * an error is returned if none is installed.  The environment variable BX_TRACE=1|2 switches it on at the first bx_init. */
const char* bx_trace_enable(int level);
int bx_trace_level(void);
/* Tunables (NTT pass split, tile sizes, Merkle layer launches, wait policy, allocation pool ...): name/value pairs in DESIGN.md
 * section 9.  Unknown names and out-of-range values are errors; every combination gives the same words. */
const char* bx_set_tunable(bx_ctx* ctx, const char* name, long value);

#if defined(__GNUC__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif
