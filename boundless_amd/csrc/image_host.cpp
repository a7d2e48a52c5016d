// image_host.cpp — the host half of include/bx_image.h: ProgramBinary / ELF decoding into the sparse page table, page access,
// and the SHA-256 of the SystemState.  No device code: this file also builds with plain g++ (tests/image_fuzz_check.cpp runs it
// under AddressSanitizer / UBSan against mutated program binaries — program binaries arrive from outside: the reference's
// executor API recomputes the image ID of whatever is uploaded, crates/executor/src/api.rs:166-178).
//
// Restates risc0-binfmt 3.0.3 (reference Cargo.lock:8806-8809, not vendored): ProgramBinary::decode, Program::load_elf,
// MemoryImage::with_kernel, SystemState::digest.  Pinned by the reference's own vector (tests/test_image_id_{cpu,gpu}.py).
#include <string.h>

#include <new>

#include "image.hpp"

namespace {

static uint32_t rd32(const uint8_t* p) { return (uint32_t)p[0] | (uint32_t)p[1] << 8 | (uint32_t)p[2] << 16 | (uint32_t)p[3] << 24; }
static uint32_t rd16(const uint8_t* p) { return (uint32_t)p[0] | (uint32_t)p[1] << 8; }

// An absent page IS the zero page (its digest is the level's zero-subtree digest either way), so a zero word never creates
// one: a segment whose p_memsz claims gigabytes of .bss costs nothing here (upstream inserts every word of it into a BTreeMap).
static void store_word(bx_image* im, uint32_t addr, uint32_t word) {
    auto it = im->pages.find(addr >> 10);
    if (it == im->pages.end()) {
        if (word == 0) return;
        it = im->pages.emplace(addr >> 10, std::vector<uint32_t>(BX_PAGE_WORDS, 0u)).first;
    }
    it->second[(addr & 1023u) >> 2] = word;
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
        if (memsz && (vaddr + (memsz - 1) < vaddr || vaddr + (memsz - 1) >= max_mem)) return "image: address outside guest memory";
        if ((uint64_t)off + filesz > len) return "image: invalid segment offset";
        const uint32_t file_part = filesz < memsz ? filesz : memsz;  // words that come from the file (the last one may be partial)
        for (uint32_t k = 0; k < file_part; k += 4) {
            uint32_t word = 0;
            const uint32_t nb = file_part - k < 4 ? file_part - k : 4;
            for (uint32_t j = 0; j < nb; ++j) word |= (uint32_t)e[(size_t)off + k + j] << (8 * j);
            store_word(im, vaddr + k, word);
        }
        // the zero tail [file_part rounded up, memsz): zero words only matter where a page already exists (store_word skips the
        // rest), so walk the existing pages of the range instead of its words — a p_memsz of gigabytes is a map lookup
        const uint64_t z0 = (uint64_t)vaddr + ((file_part + 3u) & ~3u), z1 = (uint64_t)vaddr + memsz;
        if (z0 < z1) {
            for (auto it = im->pages.lower_bound((uint32_t)(z0 >> 10)); it != im->pages.end() && it->first <= (uint32_t)((z1 - 1) >> 10); ++it) {
                const uint64_t p0 = (uint64_t)it->first << 10;
                const uint64_t a = z0 > p0 ? z0 : p0, b = z1 < p0 + 1024 ? z1 : p0 + 1024;
                for (uint64_t addr = a; addr < b; addr += 4) it->second[(addr & 1023u) >> 2] = 0u;
            }
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


}  // namespace


extern "C" const char* bx_image_new(bx_image** out) {
    if (!out) return "bx_image_new: null out";
    *out = new (std::nothrow) bx_image();
    return *out ? nullptr : "bx_image_new: out of memory";
}
extern "C" void bx_image_free(bx_image* im) { delete im; }
extern "C" size_t bx_image_page_count(const bx_image* im) { return im ? im->pages.size() : 0; }
extern "C" size_t bx_image_digest_count(const bx_image* im) { return im ? im->digests.size() : 0; }
extern "C" const char* bx_image_set_digest(bx_image* im, uint32_t node_idx, const uint32_t d[8]) {
    if (!im || !d) return "bx_image_set_digest: null argument";
    if (node_idx < 1 || node_idx >= (2u << BX_MERKLE_DEPTH)) return "bx_image_set_digest: node index outside the tree";
    for (int k = 0; k < 8; ++k)
        if (d[k] >= BX_P) return "bx_image_set_digest: digest words are canonical field elements (< P)";
    try {
        std::array<uint32_t, 8> a;
        for (int k = 0; k < 8; ++k) a[k] = d[k];
        im->digests[node_idx] = a;
    } catch (...) {
        return "bx_image_set_digest: out of memory";
    }
    return nullptr;
}
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
    (void)c;  // messages are static strings
    if (m) {
        delete im;
        return m;
    }
    *out = im;
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

