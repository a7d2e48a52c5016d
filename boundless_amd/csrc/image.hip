// image.hip — the zkVM memory image, its Poseidon2 Merkle root and the image ID (include/bx_image.h).
//
// Restates risc0_zkvm::compute_image_id (risc0-binfmt 3.0.3, reference Cargo.lock:8806-8809, not vendored:
// ProgramBinary::decode, Program::load_elf, MemoryImage::with_kernel, Page::digest, DigestPair::digest, SystemState::digest),
// called by the reference at crates/risc0-backend/src/lib.rs:590-718 and crates/executor/src/api.rs:178 and tested by it on
// its own data at crates/povw/src/log_updater.rs:383-388 — the vector tests/test_image_id_gpu.py reproduces through this file.
//
// Split: the page table is host state (a sorted map of 1 KiB pages; a program touches a few hundred of the 2^22).  All
// hashing runs on the device: one kernel turns the raw pages into the column-major matrix of Montgomery cells, hash_rows'
// kernel hashes every page (plus the zero page) in one launch, and each of the 22 tree levels is one indexed-fold launch
// whose (right, left) source list the host derived from the page indices alone — so the whole job is 24 launches queued
// back to back, one 32-byte read-back at the end and no host round trip in between.
#include <string.h>

#include <vector>

#include "ctx.hpp"
#include "image.hpp"

namespace bx {

// pages_raw: n x 256 words.  out: (n x 512) column-major, out[c*n + p] = encode(half c of page p), half 2w = low 16 bits of
// word w, half 2w+1 = high 16 bits.  A workgroup transposes a 64-page x 64-word tile through LDS so that both sides are
// coalesced (reads: 64 consecutive words of a page; writes: 64 consecutive pages of a column).
__global__ __launch_bounds__(256) void page_cells_kernel(uint32_t* __restrict__ out, const uint32_t* __restrict__ raw, uint32_t n) {
    __shared__ uint32_t tile[64][65];
    const uint32_t p0 = blockIdx.x * 64, w0 = blockIdx.y * 64;
    for (uint32_t k = threadIdx.x; k < 64 * 64; k += 256) {
        uint32_t p = k >> 6, w = k & 63;
        tile[p][w] = (p0 + p < n) ? raw[(size_t)(p0 + p) * BX_PAGE_WORDS + w0 + w] : 0u;
    }
    __syncthreads();
    for (uint32_t k = threadIdx.x; k < 64 * 64; k += 256) {
        uint32_t w = k >> 6, p = k & 63;
        if (p0 + p >= n) continue;
        uint32_t v = tile[p][w];
        size_t c = 2 * (size_t)(w0 + w);
        out[c * n + p0 + p] = fp_mul(R2, v & 0xffffu);  // canonical -> Montgomery
        out[(c + 1) * n + p0 + p] = fp_mul(R2, v >> 16);
    }
}

}  // namespace bx

using namespace bx;

extern "C" const char* bx_image_page_cells(bx_ctx* c, bx_buf out, bx_buf raw, size_t n) {
    if (!c) return "bx_image_page_cells: null ctx";
    BX_REQUIRE(c, raw.len >= n * BX_PAGE_WORDS && out.len >= n * 2 * BX_PAGE_WORDS, "image_page_cells: buffer too small");
    BX_REQUIRE(c, n <= (1u << BX_MERKLE_DEPTH) + 1u, "image_page_cells: more pages than the address space holds");
    if (n == 0) return nullptr;
    BX_HIP(c, hipSetDevice(c->device));
    OpScope op(c, "image_page_cells", 12.0 * BX_PAGE_WORDS * (double)n);
    hipLaunchKernelGGL(page_cells_kernel, dim3((unsigned)((n + 63) / 64), BX_PAGE_WORDS / 64), dim3(256), 0, c->stream, (uint32_t*)out.dptr,
                       (const uint32_t*)raw.dptr, (uint32_t)n);
    BX_LAUNCH_CHECK(c);
    return nullptr;
}

static const char* image_root(bx_ctx* c, const bx_image* im, uint32_t root_out[8]) {
    const size_t n = im->pages.size(), n1 = n + 1;  // + the zero page, always the last entry of every level
    // per level: which two entries of the level below feed each surviving parent — (right, left) as DigestPair::digest
    // puts the right child first — and the all-zero parent last
    std::vector<uint32_t> idx;
    idx.reserve(n);
    for (auto& kv : im->pages) idx.push_back(kv.first);
    std::vector<uint32_t> sel;
    std::vector<size_t> level_off, level_cnt;
    {
        std::vector<uint32_t> cur = idx, nxt;
        for (int d = 0; d < BX_MERKLE_DEPTH; ++d) {
            const uint32_t zero = (uint32_t)cur.size();
            level_off.push_back(sel.size());
            nxt.clear();
            for (size_t i = 0; i < cur.size();) {
                uint32_t lhs = zero, rhs = zero;
                const uint32_t parent = cur[i] >> 1;
                if ((cur[i] & 1u) == 0) {
                    lhs = (uint32_t)i;
                    if (i + 1 < cur.size() && cur[i + 1] == cur[i] + 1) rhs = (uint32_t)++i;
                } else {
                    rhs = (uint32_t)i;
                }
                ++i;
                sel.push_back(rhs), sel.push_back(lhs);
                nxt.push_back(parent);
            }
            sel.push_back(zero), sel.push_back(zero);
            level_cnt.push_back(nxt.size() + 1);
            cur.swap(nxt);
        }
    }
    // device layout, one allocation: raw pages | cell matrix | digests A | digests B | sel
    const size_t raw_w = n1 * BX_PAGE_WORDS, mat_w = n1 * 2 * BX_PAGE_WORDS, dig_w = n1 * 8;
    bx_buf all{nullptr, 0};
    BX_TRY(bx_alloc(c, raw_w + mat_w + 2 * dig_w + sel.size(), &all));
    uint32_t* base = (uint32_t*)all.dptr;
    bx_buf raw{base, raw_w}, mat{base + raw_w, mat_w}, da{base + raw_w + mat_w, dig_w}, db{base + raw_w + mat_w + dig_w, dig_w},
        dsel{base + raw_w + mat_w + 2 * dig_w, sel.size()};
    const char* m = nullptr;
    do {
        std::vector<uint32_t> host(raw_w, 0u);
        size_t k = 0;
        for (auto& kv : im->pages) memcpy(host.data() + (k++) * BX_PAGE_WORDS, kv.second.data(), BX_PAGE_BYTES);
        if ((m = bx_h2d(c, raw, host.data(), raw_w))) break;
        if ((m = bx_h2d(c, dsel, sel.data(), sel.size()))) break;
        if ((m = bx_image_page_cells(c, mat, raw, n1))) break;
        if ((m = bx_hash_rows(c, da, mat))) break;
        bx_buf src = da, dst = db;
        for (int d = 0; d < BX_MERKLE_DEPTH && !m; ++d) {
            bx_buf s{(uint32_t*)dsel.dptr + level_off[d], 2 * level_cnt[d]};
            m = bx_hash_fold_indexed(c, dst, src, s, level_cnt[d]);
            bx_buf t = src;
            src = dst, dst = t;
        }
        if (m) break;
        uint32_t mont[8];
        if ((m = bx_d2h(c, mont, src, 8))) break;  // entry 0: the root (the zero root when the image is empty)
        for (int i = 0; i < 8; ++i) root_out[i] = fp_decode(mont[i]);
    } while (0);
    const char* r = bx_release(c, all);
    return m ? m : r;
}

extern "C" const char* bx_image_root(bx_ctx* c, const bx_image* im, uint32_t root_canonical[8]) {
    if (!c) return "bx_image_root: null ctx";
    BX_REQUIRE(c, im && root_canonical, "image_root: null argument");
    try {
        return image_root(c, im, root_canonical);
    } catch (...) {
        return set_msg(c, "image_root: out of memory");
    }
}

extern "C" const char* bx_compute_image_id(bx_ctx* c, const uint8_t* blob, size_t len, uint8_t id_out[32]) {
    if (!c) return "bx_compute_image_id: null ctx";
    BX_REQUIRE(c, id_out, "compute_image_id: null out");
    bx_image* im = nullptr;
    BX_TRY(bx_image_from_program(c, blob, len, &im));
    uint32_t root[8];
    const char* m = bx_image_root(c, im, root);
    bx_image_free(im);
    if (m) return m;
    bx_system_state_digest(root, 0u, id_out);
    return nullptr;
}
