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

#include <map>
#include <new>
#include <vector>

#include "../../include/bx_image.h"
#include "ctx.hpp"

struct bx_image {
    std::map<uint32_t, std::vector<uint32_t>> pages;  // page index -> 256 words
};

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

static uint32_t rd32(const uint8_t* p) { return (uint32_t)p[0] | (uint32_t)p[1] << 8 | (uint32_t)p[2] << 16 | (uint32_t)p[3] << 24; }
static uint32_t rd16(const uint8_t* p) { return (uint32_t)p[0] | (uint32_t)p[1] << 8; }

static void store_word(bx_image* im, uint32_t addr, uint32_t word) {
    auto& pg = im->pages[addr >> 10];
    if (pg.empty()) pg.assign(BX_PAGE_WORDS, 0u);
    pg[(addr & 1023u) >> 2] = word;
}

// Program::load_elf: every PT_LOAD segment word by word (p_memsz words, zero past p_filesz); later loads overwrite earlier ones.
static const char* load_elf(bx_image* im, const uint8_t* e, size_t len, uint32_t max_mem, uint32_t* entry) {
    if (len < 52 || memcmp(e, "\x7f" "ELF", 4) != 0) return "image: not an ELF file";
    if (e[4] != 1) return "image: not a 32-bit ELF";
    if (e[5] != 1) return "image: not a little-endian ELF";
    if (rd16(e + 18) != 243) return "image: invalid machine type, must be RISC-V";
    if (rd16(e + 16) != 2) return "image: invalid ELF type, must be executable";
    *entry = rd32(e + 24);
    if (*entry >= max_mem || (*entry & 3)) return "image: invalid entrypoint";
    const uint32_t phoff = rd32(e + 28), phentsize = rd16(e + 42), phnum = rd16(e + 44);
    if (phnum > 256) return "image: too many program headers";
    if (phentsize != 32 || (uint64_t)phoff + (uint64_t)phnum * 32 > len) return "image: program headers outside the file";
    for (uint32_t i = 0; i < phnum; ++i) {
        const uint8_t* ph = e + phoff + 32 * i;
        if (rd32(ph) != 1) continue;  // PT_LOAD
        const uint32_t off = rd32(ph + 4), vaddr = rd32(ph + 8), filesz = rd32(ph + 16), memsz = rd32(ph + 20);
        if (filesz >= max_mem) return "image: invalid segment file_size";
        if (memsz >= max_mem) return "image: invalid segment mem_size";
        if (vaddr & 3) return "image: unaligned segment vaddr";
        for (uint32_t k = 0; k < memsz; k += 4) {
            const uint32_t addr = vaddr + k;
            if (addr < vaddr) return "image: invalid segment vaddr";
            if (addr >= max_mem) return "image: address outside guest memory";
            uint32_t word = 0;
            if (k < filesz) {
                const uint32_t nb = filesz - k < 4 ? filesz - k : 4;
                for (uint32_t j = 0; j < nb; ++j) {
                    if ((uint64_t)off + k + j >= len) return "image: invalid segment offset";
                    word |= (uint32_t)e[(size_t)off + k + j] << (8 * j);
                }
            }
            store_word(im, addr, word);
        }
    }
    return nullptr;
}

// ---- SHA-256 (FIPS 180-4) for the SystemState digest: 70 bytes per image, host only ----
static const uint32_t K256[64] = {
    0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5, 0xd807aa98, 0x12835b01, 0x243185be,
    0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174, 0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc, 0x2de92c6f, 0x4a7484aa,
    0x5cb0a9dc, 0x76f988da, 0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147, 0x06ca6351, 0x14292967, 0x27b70a85,
    0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85, 0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3,
    0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070, 0x19a4c116, 0x1e376c08, 0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f,
    0x682e6ff3, 0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208, 0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};
struct Sha256 {
    uint32_t h[8] = {0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19};
    uint8_t buf[64];
    size_t fill = 0;
    uint64_t total = 0;
    static uint32_t rr(uint32_t x, int n) { return (x >> n) | (x << (32 - n)); }
    void block(const uint8_t* p) {
        uint32_t w[64];
        for (int i = 0; i < 16; ++i) w[i] = (uint32_t)p[4 * i] << 24 | (uint32_t)p[4 * i + 1] << 16 | (uint32_t)p[4 * i + 2] << 8 | p[4 * i + 3];
        for (int i = 16; i < 64; ++i)
            w[i] = w[i - 16] + (rr(w[i - 15], 7) ^ rr(w[i - 15], 18) ^ (w[i - 15] >> 3)) + w[i - 7] + (rr(w[i - 2], 17) ^ rr(w[i - 2], 19) ^ (w[i - 2] >> 10));
        uint32_t v[8];
        memcpy(v, h, sizeof v);
        for (int i = 0; i < 64; ++i) {
            uint32_t t1 = v[7] + (rr(v[4], 6) ^ rr(v[4], 11) ^ rr(v[4], 25)) + ((v[4] & v[5]) ^ (~v[4] & v[6])) + K256[i] + w[i];
            uint32_t t2 = (rr(v[0], 2) ^ rr(v[0], 13) ^ rr(v[0], 22)) + ((v[0] & v[1]) ^ (v[0] & v[2]) ^ (v[1] & v[2]));
            for (int k = 7; k > 0; --k) v[k] = v[k - 1];
            v[4] += t1;
            v[0] = t1 + t2;
        }
        for (int k = 0; k < 8; ++k) h[k] += v[k];
    }
    void update(const uint8_t* p, size_t n) {
        total += n;
        while (n) {
            size_t take = 64 - fill < n ? 64 - fill : n;
            memcpy(buf + fill, p, take);
            fill += take, p += take, n -= take;
            if (fill == 64) block(buf), fill = 0;
        }
    }
    void finish(uint8_t out[32]) {
        const uint64_t bits = total * 8;
        const uint8_t one = 0x80, zero = 0;
        update(&one, 1);
        while (fill != 56) update(&zero, 1);
        uint8_t lenb[8];
        for (int k = 0; k < 8; ++k) lenb[k] = (uint8_t)(bits >> (8 * (7 - k)));
        update(lenb, 8);
        for (int k = 0; k < 8; ++k) out[4 * k] = h[k] >> 24, out[4 * k + 1] = h[k] >> 16, out[4 * k + 2] = h[k] >> 8, out[4 * k + 3] = h[k];
    }
};

}  // namespace bx

using namespace bx;

extern "C" const char* bx_image_new(bx_image** out) {
    if (!out) return "bx_image_new: null out";
    *out = new (std::nothrow) bx_image();
    return *out ? nullptr : "bx_image_new: out of memory";
}
extern "C" void bx_image_free(bx_image* im) { delete im; }
extern "C" size_t bx_image_page_count(const bx_image* im) { return im ? im->pages.size() : 0; }
extern "C" size_t bx_image_page_indices(const bx_image* im, uint32_t* out, size_t cap) {
    if (!im) return 0;
    size_t k = 0;
    for (auto& kv : im->pages) {
        if (k < cap && out) out[k] = kv.first;
        ++k;
    }
    return k;
}
extern "C" const char* bx_image_set_page(bx_image* im, uint32_t page_idx, const uint32_t* words) {
    if (!im || !words) return "bx_image_set_page: null argument";
    if (page_idx >= (1u << BX_MERKLE_DEPTH)) return "bx_image_set_page: page index outside the 4 GiB address space";
    try {
        im->pages[page_idx].assign(words, words + BX_PAGE_WORDS);
    } catch (...) {
        return "bx_image_set_page: out of memory";
    }
    return nullptr;
}
extern "C" const char* bx_image_get_page(const bx_image* im, uint32_t page_idx, uint32_t* words_out) {
    if (!im || !words_out) return "bx_image_get_page: null argument";
    auto it = im->pages.find(page_idx);
    if (it == im->pages.end()) memset(words_out, 0, BX_PAGE_BYTES);
    else memcpy(words_out, it->second.data(), BX_PAGE_BYTES);
    return nullptr;
}

static const char* image_from_program(const uint8_t* blob, size_t len, bx_image* im) {
    if (!blob || len < 16 || memcmp(blob, "R0BF", 4) != 0) return "image: malformed ProgramBinary (magic)";
    if (rd32(blob + 4) != 1) return "image: ProgramBinary binary format version mismatch";
    const uint32_t hlen = rd32(blob + 8);
    if ((uint64_t)12 + hlen + 4 > len) return "image: malformed ProgramBinary (header)";
    const uint32_t ulen = rd32(blob + 12 + hlen);
    if ((uint64_t)16 + hlen + ulen > len) return "image: malformed ProgramBinary (user ELF length)";
    const uint8_t* user = blob + 16 + hlen;
    const uint8_t* kern = user + ulen;
    const size_t klen = len - 16 - hlen - ulen;
    uint32_t uentry = 0, kentry = 0;
    // kernel first, user on top: where both ELFs map an address (each maps its own headers at 0x0001_0000) the user
    // program's word is the one the image keeps — settled by the reference's vector, see bx_image.h
    if (const char* m = load_elf(im, kern, klen, 0xffffffffu, &kentry)) return m;
    if (const char* m = load_elf(im, user, ulen, 0xc0000000u, &uentry)) return m;
    store_word(im, 0x00010000u, uentry);  // USER_START_ADDR (the kernel's _start reads the user entry here)
    store_word(im, 0xffff0210u, kentry);  // SUSPEND_PC_ADDR
    store_word(im, 0xffff0214u, 1u);      // SUSPEND_MODE_ADDR = machine mode
    return nullptr;
}

extern "C" const char* bx_image_from_program(bx_ctx* c, const uint8_t* blob, size_t len, bx_image** out) {
    if (!out) return "bx_image_from_program: null out";
    *out = nullptr;
    bx_image* im = new (std::nothrow) bx_image();
    if (!im) return "bx_image_from_program: out of memory";
    const char* m = nullptr;
    try {
        m = image_from_program(blob, len, im);
    } catch (...) {
        m = "image: out of memory";
    }
    if (m) {
        delete im;
        return c ? set_msg(c, m) : m;
    }
    *out = im;
    return nullptr;
}

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

extern "C" void bx_system_state_digest(const uint32_t root[8], uint32_t pc, uint8_t out[32]) {
    // tagged_struct("risc0.SystemState", down = [merkle_root], data = [pc]): SHA-256(tag digest | down | data LE | u16 LE count)
    static const char tag[] = "risc0.SystemState";
    uint8_t t[32], body[32 + 4 + 2];
    Sha256 a;
    a.update((const uint8_t*)tag, sizeof tag - 1);
    a.finish(t);
    for (int k = 0; k < 8; ++k)
        for (int j = 0; j < 4; ++j) body[4 * k + j] = (uint8_t)(root[k] >> (8 * j));
    for (int j = 0; j < 4; ++j) body[32 + j] = (uint8_t)(pc >> (8 * j));
    body[36] = 1, body[37] = 0;
    Sha256 b;
    b.update(t, 32);
    b.update(body, sizeof body);
    b.finish(out);
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
