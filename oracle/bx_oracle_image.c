/*
 * bx_oracle_image.c — CPU ORACLE (TEST INFRASTRUCTURE ONLY): compute_image_id of a RISC Zero program binary.
 *
 * What it restates.  The reference calls `risc0_zkvm::compute_image_id(blob)` wherever it admits a guest program
 *   crates/risc0-backend/src/lib.rs:590,627,655,718      crates/executor/src/api.rs:166-178
 * and *tests* it on data it ships:
 *   crates/povw/src/log_updater.rs:383-388   compute_image_id(BOUNDLESS_POVW_LOG_UPDATER_ELF) == BOUNDLESS_POVW_LOG_UPDATER_ID
 *   crates/povw/src/lib.rs:18-22             the two files: elfs/boundless-povw-log-updater.{bin,iid}
 * The function itself lives in risc0-binfmt 3.0.3 (Cargo.lock:8806-8809, not vendored): `ProgramBinary::decode`,
 * `Program::load_elf`, `MemoryImage::with_kernel`, `Page::digest`, `DigestPair::digest`, `SystemState::digest`.  Its
 * published algorithm, restated here and PINNED by the reference's own vector above (tests/test_image_id_cpu.py):
 *
 *   blob   = "R0BF" | u32 version | u32 header_len | header | u32 user_len | user ELF | kernel ELF          (little endian)
 *   image  = sparse map word address -> u32 of every PT_LOAD segment of both ELFs (p_memsz words, zero past p_filesz);
 *            where both map an address the USER program's word stays; then
 *            image[0x0001_0000] = user entry, image[0xffff_0210] = kernel entry, image[0xffff_0214] = 1
 *   page   = 1 KiB = 256 words; digest = Poseidon2 sponge (overwrite mode, rate 16, t = 24) over its 512 cells
 *            (word & 0xffff, word >> 16 per word, as field elements), i.e. 32 permutations, cells[0..8) out
 *   node i = Poseidon2 permutation of (digest[2i+1] | digest[2i] | 0^8) — the RIGHT child first — cells[0..8) out;
 *            4 GiB / 1 KiB = 2^22 leaves, depth 22, all-zero subtrees from a per-level cache; digest words are the
 *            CANONICAL values of the cells
 *   id     = SHA-256( SHA-256("risc0.SystemState") | root (8 LE words) | pc = 0 (u32 LE) | 1 (u16 LE) )
 *
 * The three conventions the reference vector settles — which ELF wins an address both map, child order, canonical vs
 * Montgomery digest words — were found by tools/image_id_search.py over the variants listed there; exactly one
 * combination reproduces the .iid.  Because the 32 bytes depend on every one of the 237 Poseidon2 parameters, on the
 * sponge and on the pair hash, this vector pins bxo_poseidon2_mix / bxo_hash_elem_slice / bxo_hash_pair to the reference.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use this file.
 */
#include <stdlib.h>
#include <string.h>

#include "bx_oracle.h"

/* ---- SHA-256 (FIPS 180-4), byte-oriented ---- */
static const uint32_t K256[64] = {
    0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5, 0xd807aa98, 0x12835b01, 0x243185be,
    0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174, 0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc, 0x2de92c6f, 0x4a7484aa,
    0x5cb0a9dc, 0x76f988da, 0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147, 0x06ca6351, 0x14292967, 0x27b70a85,
    0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85, 0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3,
    0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070, 0x19a4c116, 0x1e376c08, 0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f,
    0x682e6ff3, 0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208, 0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};
static uint32_t ror(uint32_t x, int n) { return (x >> n) | (x << (32 - n)); }
static void sha256_block(uint32_t h[8], const uint8_t* p) {
    uint32_t w[64];
    for (int i = 0; i < 16; i++) w[i] = (uint32_t)p[4 * i] << 24 | (uint32_t)p[4 * i + 1] << 16 | (uint32_t)p[4 * i + 2] << 8 | p[4 * i + 3];
    for (int i = 16; i < 64; i++) {
        uint32_t s0 = ror(w[i - 15], 7) ^ ror(w[i - 15], 18) ^ (w[i - 15] >> 3), s1 = ror(w[i - 2], 17) ^ ror(w[i - 2], 19) ^ (w[i - 2] >> 10);
        w[i] = w[i - 16] + s0 + w[i - 7] + s1;
    }
    uint32_t a = h[0], b = h[1], c = h[2], d = h[3], e = h[4], f = h[5], g = h[6], hh = h[7];
    for (int i = 0; i < 64; i++) {
        uint32_t t1 = hh + (ror(e, 6) ^ ror(e, 11) ^ ror(e, 25)) + ((e & f) ^ (~e & g)) + K256[i] + w[i];
        uint32_t t2 = (ror(a, 2) ^ ror(a, 13) ^ ror(a, 22)) + ((a & b) ^ (a & c) ^ (b & c));
        hh = g, g = f, f = e, e = d + t1, d = c, c = b, b = a, a = t1 + t2;
    }
    h[0] += a, h[1] += b, h[2] += c, h[3] += d, h[4] += e, h[5] += f, h[6] += g, h[7] += hh;
}
void bxo_sha256(uint8_t out[32], const uint8_t* msg, size_t len) {
    uint32_t h[8] = {0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19};
    size_t i = 0;
    for (; i + 64 <= len; i += 64) sha256_block(h, msg + i);
    uint8_t tail[128] = {0};
    size_t r = len - i;
    memcpy(tail, msg + i, r);
    tail[r] = 0x80;
    size_t tl = r + 9 <= 64 ? 64 : 128;
    uint64_t bits = (uint64_t)len * 8;
    for (int k = 0; k < 8; k++) tail[tl - 1 - k] = (uint8_t)(bits >> (8 * k));
    sha256_block(h, tail);
    if (tl == 128) sha256_block(h, tail + 64);
    for (int k = 0; k < 8; k++) out[4 * k] = h[k] >> 24, out[4 * k + 1] = h[k] >> 16, out[4 * k + 2] = h[k] >> 8, out[4 * k + 3] = h[k];
}

/* ---- sparse memory image: pages kept sorted by index ---- */
#define PAGE_WORDS 256u
#define TREE_DEPTH 22
typedef struct {
    uint32_t idx;
    uint32_t w[PAGE_WORDS];
} page_t;
typedef struct {
    page_t* p;
    size_t n, cap;
} image_t;

static page_t* image_page(image_t* im, uint32_t idx) {
    size_t lo = 0, hi = im->n;
    while (lo < hi) {
        size_t mid = (lo + hi) / 2;
        if (im->p[mid].idx < idx) lo = mid + 1;
        else hi = mid;
    }
    if (lo < im->n && im->p[lo].idx == idx) return &im->p[lo];
    if (im->n == im->cap) {
        im->cap = im->cap ? 2 * im->cap : 64;
        im->p = realloc(im->p, im->cap * sizeof(page_t));
    }
    memmove(&im->p[lo + 1], &im->p[lo], (im->n - lo) * sizeof(page_t));
    im->n++;
    im->p[lo].idx = idx;
    memset(im->p[lo].w, 0, sizeof im->p[lo].w);
    return &im->p[lo];
}
static void image_store(image_t* im, uint32_t addr, uint32_t word) { image_page(im, addr >> 10)->w[(addr & 1023u) >> 2] = word; }

static uint32_t rd32(const uint8_t* p) { return (uint32_t)p[0] | (uint32_t)p[1] << 8 | (uint32_t)p[2] << 16 | (uint32_t)p[3] << 24; }
static uint16_t rd16(const uint8_t* p) { return (uint16_t)(p[0] | p[1] << 8); }

/* Program::load_elf: every PT_LOAD segment, word by word; returns 0 and the entry point, or a negative error */
static int load_elf(image_t* im, const uint8_t* e, size_t len, uint32_t max_mem, uint32_t* entry) {
    if (len < 52 || memcmp(e, "\x7f" "ELF", 4) != 0) return -10;
    if (e[4] != 1 || e[5] != 1) return -11;                /* ELF32, little endian */
    if (rd16(e + 16) != 2 || rd16(e + 18) != 243) return -12; /* ET_EXEC, EM_RISCV */
    *entry = rd32(e + 24);
    if (*entry >= max_mem || (*entry & 3)) return -13;
    uint32_t phoff = rd32(e + 28);
    uint16_t phentsize = rd16(e + 42), phnum = rd16(e + 44);
    if (phentsize != 32 || phnum > 256 || (uint64_t)phoff + (uint64_t)phnum * 32 > len) return -14;
    for (unsigned i = 0; i < phnum; i++) {
        const uint8_t* ph = e + phoff + 32 * i;
        if (rd32(ph) != 1) continue; /* PT_LOAD */
        uint32_t off = rd32(ph + 4), vaddr = rd32(ph + 8), filesz = rd32(ph + 16), memsz = rd32(ph + 20);
        if (filesz >= max_mem || memsz >= max_mem || (vaddr & 3)) return -15;
        for (uint32_t k = 0; k < memsz; k += 4) {
            uint32_t addr = vaddr + k;
            if (addr < vaddr || addr >= max_mem) return -16;
            uint32_t word = 0;
            if (k < filesz) {
                uint32_t n = filesz - k < 4 ? filesz - k : 4;
                for (uint32_t j = 0; j < n; j++) {
                    if ((uint64_t)off + k + j >= len) return -17;
                    word |= (uint32_t)e[off + k + j] << (8 * j);
                }
            }
            image_store(im, addr, word);
        }
    }
    return 0;
}

/* Page::digest: 32 absorptions of 16 half-word cells = the rate-16 overwrite sponge over 512 elements */
static void page_digest(uint32_t out_mont[8], const uint32_t w[PAGE_WORDS]) {
    uint32_t cells[2 * PAGE_WORDS];
    for (unsigned i = 0; i < PAGE_WORDS; i++) cells[2 * i] = bxo_fp_encode(w[i] & 0xffffu), cells[2 * i + 1] = bxo_fp_encode(w[i] >> 16);
    bxo_hash_elem_slice(out_mont, cells, 2 * PAGE_WORDS, 1);
}

/* Digests are carried as Montgomery words inside (BabyBearElem::new(canonical word) re-encodes exactly that), decoded once
 * at the root.  root_canonical may be NULL.  Returns 0 or a negative error code. */
int bxo_compute_image_id(const uint8_t* blob, size_t len, uint8_t id_out[32], uint32_t root_canonical[8]) {
    bxo_init();
    if (len < 16 || memcmp(blob, "R0BF", 4) != 0) return -1;
    if (rd32(blob + 4) != 1) return -2; /* binary format version */
    uint32_t hlen = rd32(blob + 8);
    if ((uint64_t)12 + hlen + 4 > len) return -3;
    uint32_t ulen = rd32(blob + 12 + hlen);
    const uint8_t* user = blob + 16 + hlen;
    if ((uint64_t)16 + hlen + ulen > len) return -4;
    const uint8_t* kern = user + ulen;
    size_t klen = len - 16 - hlen - ulen;

    image_t im = {0};
    uint32_t uentry = 0, kentry = 0;
    /* kernel first, then the user program on top: the user's word stays where both ELFs map an address */
    int rc = load_elf(&im, kern, klen, 0xffffffffu, &kentry);
    if (rc == 0) rc = load_elf(&im, user, ulen, 0xc0000000u, &uentry);
    if (rc != 0) {
        free(im.p);
        return rc;
    }
    image_store(&im, 0x00010000u, uentry);  /* USER_START_ADDR: the kernel's _start loads the user entry from here */
    image_store(&im, 0xffff0210u, kentry);  /* SUSPEND_PC_ADDR */
    image_store(&im, 0xffff0214u, 1);       /* SUSPEND_MODE_ADDR: machine mode */

    /* leaves, then 22 sparse levels with the zero-subtree digest of each level */
    size_t n = im.n;
    uint32_t* idx = malloc(n * sizeof(uint32_t));
    uint32_t* dig = malloc(n * 8 * sizeof(uint32_t));
    for (size_t i = 0; i < n; i++) idx[i] = im.p[i].idx, page_digest(dig + 8 * i, im.p[i].w);
    uint32_t zero[8];
    {
        uint32_t zw[PAGE_WORDS] = {0};
        page_digest(zero, zw);
    }
    for (int d = 0; d < TREE_DEPTH; d++) {
        size_t m = 0;
        for (size_t i = 0; i < n;) {
            uint32_t parent = idx[i] >> 1;
            const uint32_t *lhs = zero, *rhs = zero;
            if ((idx[i] & 1) == 0) {
                lhs = dig + 8 * i;
                if (i + 1 < n && idx[i + 1] == idx[i] + 1) rhs = dig + 8 * (i + 1), i++;
            } else {
                rhs = dig + 8 * i;
            }
            i++;
            uint32_t out[8];
            bxo_hash_pair(out, rhs, lhs); /* DigestPair::digest: cells[0..8) = rhs, cells[8..16) = lhs */
            memcpy(dig + 8 * m, out, sizeof out);
            idx[m++] = parent;
        }
        n = m;
        uint32_t z2[8];
        bxo_hash_pair(z2, zero, zero);
        memcpy(zero, z2, sizeof zero);
    }
    uint32_t root[8];
    for (int k = 0; k < 8; k++) root[k] = bxo_fp_decode(n ? dig[k] : zero[k]);
    if (root_canonical) memcpy(root_canonical, root, sizeof root);
    free(idx), free(dig), free(im.p);

    /* SystemState { pc: 0, merkle_root }.digest() = tagged_struct("risc0.SystemState", [root], [pc]) */
    uint8_t buf[32 + 32 + 4 + 2];
    static const char tag[] = "risc0.SystemState";
    bxo_sha256(buf, (const uint8_t*)tag, sizeof tag - 1);
    for (int k = 0; k < 8; k++) buf[32 + 4 * k] = root[k], buf[33 + 4 * k] = root[k] >> 8, buf[34 + 4 * k] = root[k] >> 16, buf[35 + 4 * k] = root[k] >> 24;
    memset(buf + 64, 0, 4); /* pc = 0 */
    buf[68] = 1, buf[69] = 0; /* one digest below */
    bxo_sha256(id_out, buf, sizeof buf);
    return 0;
}
