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

#include <algorithm>

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

extern "C" const char* bx_image_page_cells(bx_ctx* c, bx_buf out, bx_buf raw, size_t n) try {
    if (!c) return "bx_image_page_cells: null ctx";
    BX_REQUIRE(c, raw.len >= n * BX_PAGE_WORDS && out.len >= n * 2 * BX_PAGE_WORDS, "image_page_cells: buffer too small");
    BX_REQUIRE(c, n <= (1u << BX_MERKLE_DEPTH) + 1u, "image_page_cells: more pages than the address space holds");
    if (n == 0) return nullptr;
    BX_ENTER(c);
    OpScope op(c, "image_page_cells", 12.0 * BX_PAGE_WORDS * (double)n);
    hipLaunchKernelGGL(page_cells_kernel, dim3((unsigned)((n + 63) / 64), BX_PAGE_WORDS / 64), dim3(256), 0, c->stream, (uint32_t*)out.dptr,
                       (const uint32_t*)raw.dptr, (uint32_t)n);
    BX_LAUNCH_CHECK(c);
    return nullptr;
} BX_ABI_CATCH(c, "bx_image_page_cells")

// Digest of node `top` (canonical words) from the pages and the given digests below it.  Every digest lives in one device pool:
// [0, n) page digests | [n] the zero page | per level its parents then its zero subtree | the given digests; a level's fold reads
// any earlier entry of the pool through absolute indices, so given digests enter at whatever level they belong to.
static const char* image_node(bx_ctx* c, const bx_image* im, uint32_t top, uint32_t out[8]) {
    // node 0 does not exist (0 << k never reaches the leaf layer: the loop below would not end) and nothing lies beyond the leaves
    BX_REQUIRE(c, top >= 1 && top < (2u << BX_MERKLE_DEPTH), "image: node index outside the tree");
    int top_level = 0;  // levels above the leaves
    while (top_level < BX_MERKLE_DEPTH && (top << top_level) < (1u << BX_MERKLE_DEPTH)) ++top_level;
    const uint32_t leaf_lo = top << top_level, leaf_hi = leaf_lo + (1u << top_level);  // node indices of the leaves below `top`
    // the given digests at or below `top`, by level (0 = leaves); one AT an ancestor of `top` would hide it
    std::vector<std::vector<std::pair<uint32_t, const uint32_t*>>> given(top_level + 1);
    size_t n_given = 0;
    for (auto& kv : im->digests) {
        int lvl = 0;
        while ((kv.first << lvl) < (1u << BX_MERKLE_DEPTH)) ++lvl;
        const uint32_t lo = kv.first << lvl;
        if (lo >= leaf_lo && lo < leaf_hi && lvl <= top_level) {
            given[lvl].push_back({kv.first, kv.second.data()});
            ++n_given;
        } else if (lvl > top_level && (kv.first << (lvl - top_level)) <= top && top < ((kv.first + 1) << (lvl - top_level))) {
            return set_msg(c, "image: the node lies inside a subtree that is only given by its digest");
        }
    }
    std::vector<uint32_t> page_idx;
    for (auto it = im->pages.lower_bound(leaf_lo - (1u << BX_MERKLE_DEPTH)); it != im->pages.end() && it->first + (1u << BX_MERKLE_DEPTH) < leaf_hi; ++it)
        page_idx.push_back(it->first);
    const size_t n = page_idx.size();
    // pool positions: pages [0,n), zero page n; level outputs appended; given digests at the end
    struct Node { uint32_t idx, pos; };
    std::vector<Node> cur;
    for (size_t i = 0; i < n; ++i) cur.push_back({page_idx[i] + (1u << BX_MERKLE_DEPTH), (uint32_t)i});
    uint32_t zero_pos = (uint32_t)n, pool = (uint32_t)n + 1;
    // given digests get their pool positions after all computed entries; count computed entries first (each level: parents + zero)
    std::vector<uint32_t> sel;
    std::vector<size_t> level_off, level_cnt, level_out;
    std::vector<std::pair<uint32_t, const uint32_t*>> given_order;  // pool order of the given digests
    // two passes are avoided by reserving the given digests' positions relative to an offset fixed afterwards
    const uint32_t GIVEN = 0x80000000u;  // position = GIVEN | ordinal, rewritten below
    auto merge_given = [&](int lvl, std::vector<Node>& nodes) -> const char* {
        for (auto& g : given[lvl]) {
            nodes.push_back({g.first, GIVEN | (uint32_t)given_order.size()});
            given_order.push_back(g);
        }
        std::sort(nodes.begin(), nodes.end(), [](const Node& a, const Node& b) { return a.idx < b.idx; });
        for (size_t i = 1; i < nodes.size(); ++i)
            if (nodes[i].idx == nodes[i - 1].idx) return "image: a subtree is given both by its digest and by pages or digests inside it";
        return nullptr;
    };
    if (const char* e = merge_given(0, cur)) return set_msg(c, e);
    for (int d = 0; d < top_level; ++d) {
        level_off.push_back(sel.size());
        level_out.push_back(pool);
        std::vector<Node> nxt;
        for (size_t i = 0; i < cur.size();) {
            uint32_t lhs = zero_pos, rhs = zero_pos;
            const uint32_t parent = cur[i].idx >> 1;
            if ((cur[i].idx & 1u) == 0) {
                lhs = cur[i].pos;
                if (i + 1 < cur.size() && cur[i + 1].idx == cur[i].idx + 1) rhs = cur[++i].pos;
            } else {
                rhs = cur[i].pos;
            }
            ++i;
            sel.push_back(rhs), sel.push_back(lhs);  // DigestPair::digest puts the right child first
            nxt.push_back({parent, pool++});
        }
        sel.push_back(zero_pos), sel.push_back(zero_pos);  // the level's all-zero subtree
        zero_pos = pool++;
        level_cnt.push_back(nxt.size() + 1);
        if (const char* e = merge_given(d + 1, nxt)) return set_msg(c, e);
        cur.swap(nxt);
    }
    BX_REQUIRE(c, given_order.size() == n_given, "image: internal error (given digests)");
    const uint32_t given_base = pool;
    for (auto& v : sel)
        if (v & GIVEN) v = given_base + (v & ~GIVEN);
    uint32_t result_pos = cur.empty() ? zero_pos : cur[0].pos;
    if (result_pos & GIVEN) result_pos = given_base + (result_pos & ~GIVEN);
    const size_t pool_digests = (size_t)given_base + n_given;
    // device layout, one allocation: raw pages | cell matrix | digest pool | sel
    const size_t n1 = n + 1, raw_w = n1 * BX_PAGE_WORDS, mat_w = n1 * 2 * BX_PAGE_WORDS, pool_w = pool_digests * 8;
    bx_buf all{nullptr, 0};
    BX_TRY(bx_alloc(c, raw_w + mat_w + pool_w + sel.size() + 4, &all));
    uint32_t* base = (uint32_t*)all.dptr;
    bx_buf raw{base, raw_w}, mat{base + raw_w, mat_w}, dpool{base + raw_w + mat_w, pool_w}, dsel{base + raw_w + mat_w + pool_w, sel.size()};
    const char* m = nullptr;
    do {
        std::vector<uint32_t> host(raw_w, 0u);
        for (size_t k = 0; k < n; ++k) memcpy(host.data() + k * BX_PAGE_WORDS, im->pages.at(page_idx[k]).data(), BX_PAGE_BYTES);
        if ((m = bx_h2d(c, raw, host.data(), raw_w))) break;
        if (!sel.empty() && (m = bx_h2d(c, dsel, sel.data(), sel.size()))) break;
        if (n_given) {
            std::vector<uint32_t> g(8 * n_given);
            for (size_t k = 0; k < n_given; ++k)
                for (int w = 0; w < 8; ++w) g[8 * k + w] = fp_encode(given_order[k].second[w]);  // canonical -> Montgomery, as BabyBearElem::new
            if ((m = bx_h2d(c, bx_buf{(uint32_t*)dpool.dptr + 8 * (size_t)given_base, 8 * n_given}, g.data(), g.size()))) break;
        }
        if ((m = bx_image_page_cells(c, mat, raw, n1))) break;
        if ((m = bx_hash_rows(c, bx_buf{dpool.dptr, 8 * n1}, mat))) break;
        for (int d = 0; d < top_level && !m; ++d)
            m = bx_hash_fold_indexed(c, bx_buf{(uint32_t*)dpool.dptr + 8 * level_out[d], 8 * level_cnt[d]}, dpool,
                                     bx_buf{(uint32_t*)dsel.dptr + level_off[d], 2 * level_cnt[d]}, level_cnt[d]);
        if (m) break;
        uint32_t mont[8];
        if ((m = bx_d2h(c, mont, bx_buf{(uint32_t*)dpool.dptr + 8 * (size_t)result_pos, 8}, 8))) break;
        for (int i = 0; i < 8; ++i) out[i] = fp_decode(mont[i]);
    } while (0);
    const char* r = bx_release(c, all);
    return m ? m : r;
}
static const char* image_root(bx_ctx* c, const bx_image* im, uint32_t root_out[8]) { return image_node(c, im, 1u, root_out); }

extern "C" const char* bx_image_node_digest(bx_ctx* c, const bx_image* im, uint32_t node_idx, uint32_t digest_canonical[8]) try {
    if (!c) return "bx_image_node_digest: null ctx";
    BX_REQUIRE(c, im && digest_canonical, "image_node_digest: null argument");
    try {
        return image_node(c, im, node_idx, digest_canonical);
    } catch (...) {
        return set_msg(c, "image_node_digest: out of memory");
    }
} BX_ABI_CATCH(c, "bx_image_node_digest")

extern "C" const char* bx_image_root(bx_ctx* c, const bx_image* im, uint32_t root_canonical[8]) try {
    if (!c) return "bx_image_root: null ctx";
    BX_REQUIRE(c, im && root_canonical, "image_root: null argument");
    try {
        return image_root(c, im, root_canonical);
    } catch (...) {
        return set_msg(c, "image_root: out of memory");
    }
} BX_ABI_CATCH(c, "bx_image_root")

extern "C" const char* bx_compute_image_id(bx_ctx* c, const uint8_t* blob, size_t len, uint8_t id_out[32]) try {
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
} BX_ABI_CATCH(c, "bx_compute_image_id")
