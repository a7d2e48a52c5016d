/*
 * bx_image.h — the zkVM memory image and its Poseidon2 Merkle root / image ID on the MI355X.
 *
 * Reference interface this replaces
 * ---------------------------------
 *   risc0_zkvm::compute_image_id(blob) -> Digest        crates/risc0-backend/src/lib.rs:590,627,655,718
 *                                                       crates/executor/src/api.rs:166-178 (upload admission check)
 *                                                       crates/indexer/src/market/service/execution.rs:699
 *   the reference's own known-answer test               crates/povw/src/log_updater.rs:383-388 on
 *                                                       crates/povw/elfs/boundless-povw-log-updater.{bin,iid}
 *   `Segment.partial_image` (the MemoryImage half of    bento/crates/workflow/src/tasks/prove.rs:36-49 — what
 *   the wire type handed to prove_segment)              prove_segment page-ins and re-hashes under its Merkle root
 * The algorithm lives in risc0-binfmt 3.0.3 (Cargo.lock:8806-8809, not vendored): ProgramBinary::decode,
 * Program::load_elf, MemoryImage::with_kernel, Page::digest, DigestPair::digest, SystemState::digest.  What is restated:
 *
 *   image  : sparse 4 GiB address space of 1 KiB pages (2^22 leaves); both ELFs' PT_LOAD segments, the user program's
 *            words winning where both map an address; user entry at 0x0001_0000, kernel entry at 0xffff_0210, mode 1 at
 *            0xffff_0214
 *   leaf   : Poseidon2 rate-16 overwrite sponge over the page's 512 half-word cells (lo, hi per word)  -> bx_hash_rows'
 *            kernel on a (pages x 512) column-major matrix built on the device
 *   node i : Poseidon2 of (digest[2i+1] | digest[2i]) — right child first; absent subtrees use the level's all-zero
 *            digest — one indexed fold launch per level (22 levels)
 *   id     : SHA-256(SHA-256("risc0.SystemState") | root as 8 canonical LE words | u32 pc = 0 | u16 1)   (host)
 *
 * This is the one place the reference pins a Poseidon2 output byte for byte: tests/test_image_id_gpu.py computes the
 * reference's .iid through these entry points (and through plain bx_hash_rows + bx_hash_fold).
 *
 * Conventions as in bx_hal.h (NULL = ok, message owned by the ctx; no exception crosses the ABI).  A bx_image is
 * host-side state (the page table); hashing runs on the ctx's stream.
 */
#ifndef BX_IMAGE_H
#define BX_IMAGE_H
#include "bx_hal.h"
#ifdef __cplusplus
extern "C" {
#endif
/* the library is built with -fvisibility=hidden: only what these headers declare is exported */
#if defined(__GNUC__)
#pragma GCC visibility push(default)
#endif

#define BX_PAGE_BYTES 1024u
#define BX_PAGE_WORDS 256u
#define BX_MERKLE_DEPTH 22

typedef struct bx_image bx_image;

/* ProgramBinary::decode + to_image: parse an "R0BF" blob (user ELF + kernel ELF) into a memory image.  Host only (ctx may
 * be NULL; then errors are returned as static strings). */
const char* bx_image_from_program(bx_ctx* ctx, const uint8_t* blob, size_t len, bx_image** out);
/* An empty image, and direct page access (MemoryImage::set_page / get_page): `words` = 256 little-endian u32. */
const char* bx_image_new(bx_image** out);
const char* bx_image_set_page(bx_image* im, uint32_t page_idx, const uint32_t* words);
const char* bx_image_get_page(const bx_image* im, uint32_t page_idx, uint32_t* words_out); /* zero page if absent */
/* Partial images (the `partial_image` of a Segment: MemoryImage with `pages` for what the segment touches and `digests` for the
 * subtrees it does not): the digest of node `node_idx` (root = 1, children of i = 2i and 2i+1, page p = node 2^22 + p), 8
 * canonical words as they appear in risc0's Digest.  The subtree below that node must hold no page and no other digest;
 * bx_image_root then folds pages and given digests together and fails if they overlap. */
const char* bx_image_set_digest(bx_image* im, uint32_t node_idx, const uint32_t digest_canonical[8]);
size_t bx_image_digest_count(const bx_image* im);
/* Digest of any node of the tree (8 canonical words): how a full image is pruned into a partial one. Blocks. */
const char* bx_image_node_digest(bx_ctx* ctx, const bx_image* im, uint32_t node_idx, uint32_t digest_canonical[8]);
size_t bx_image_page_count(const bx_image* im);
/* page indices in ascending order (cap entries at most); returns the count */
size_t bx_image_page_indices(const bx_image* im, uint32_t* out, size_t cap);
void bx_image_free(bx_image* im);

/* MemoryImage::image_id's Merkle half on the GPU: the root of the 2^22-leaf Poseidon2 tree, as 8 canonical words. Blocks. */
const char* bx_image_root(bx_ctx* ctx, const bx_image* im, uint32_t root_canonical[8]);
/* SystemState { pc, merkle_root }.digest() — host SHA-256 (tagged struct "risc0.SystemState"). */
void bx_system_state_digest(const uint32_t root_canonical[8], uint32_t pc, uint8_t out[32]);
/* risc0_zkvm::compute_image_id: the three steps above. Blocks. */
const char* bx_compute_image_id(bx_ctx* ctx, const uint8_t* blob, size_t len, uint8_t id_out[32]);

/* The indexed fold the tree levels use, exposed because a sparse Merkle update is also what MemoryImage::update_digests
 * does after a segment dirties pages: out[j] = H(in[sel[2j]] | in[sel[2j+1]]), j < count; sel indexes digests of `in`.
 * `out` may be a range of the same pool as `in` (the tree levels are built that way), but no sel entry may name a digest inside
 * `out`: a level reads only what earlier launches wrote.  Every sel entry must be < in.len / 8; the entry point checks the
 * host-visible sizes, the kernel clamps an entry beyond the pool to its last digest instead of reading out of bounds. */
const char* bx_hash_fold_indexed(bx_ctx* ctx, bx_buf out_digests, bx_buf in_digests, bx_buf sel_u32, size_t count);
/* pages (n x 256 raw u32 words, row-major as in memory) -> the (n x 512) column-major matrix of Montgomery cells that
 * bx_hash_rows hashes into page digests. */
const char* bx_image_page_cells(bx_ctx* ctx, bx_buf out_matrix, bx_buf pages_raw, size_t n_pages);

#if defined(__GNUC__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif
