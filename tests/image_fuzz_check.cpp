// image_fuzz_check.cpp — the program-binary / ELF loader (boundless_amd/csrc/image_host.cpp) under AddressSanitizer / UBSan.
//
// Built and run by tests/test_image_id_cpu.py:  image_fuzz_check <program.bin> <iterations>
// program.bin = the reference's boundless-povw-log-updater.bin (an R0BF ProgramBinary: user ELF + kernel ELF).  Program binaries
// come from outside — the reference's executor API recomputes the image ID of whatever is uploaded
// (crates/executor/src/api.rs:166-178) — so every mutation (truncation, bit flips, hostile program-header fields: offsets and
// sizes past the file, vaddr + memsz wrapping, gigabytes of .bss, 65535 headers) must come back as an error string or a valid
// image, never as an out-of-bounds access, an overflow, a leak, or minutes of CPU.
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include <vector>

#include "../include/bx_image.h"

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static uint64_t rnd() {  // splitmix64
    uint64_t z = (rng_state += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
static void wr32(std::vector<uint8_t>& b, size_t off, uint32_t v) {
    for (int k = 0; k < 4 && off + k < b.size(); ++k) b[off + k] = (uint8_t)(v >> (8 * k));
}

int main(int argc, char** argv) {
    if (argc < 3) return 2;
    FILE* f = fopen(argv[1], "rb");
    if (!f) return 2;
    std::vector<uint8_t> blob;
    uint8_t buf[65536];
    size_t n;
    while ((n = fread(buf, 1, sizeof buf, f)) > 0) blob.insert(blob.end(), buf, buf + n);
    fclose(f);
    const long iters = atol(argv[2]);
    bx_image* im = nullptr;
    if (const char* e = bx_image_from_program(nullptr, blob.data(), blob.size(), &im)) {
        printf("honest program rejected: %s\n", e);
        return 1;
    }
    const size_t honest_pages = bx_image_page_count(im);
    bx_image_free(im);
    // where the two ELFs and their program headers sit
    const uint32_t hlen = blob[8] | blob[9] << 8 | blob[10] << 16 | (uint32_t)blob[11] << 24;
    const size_t user_off = 16 + hlen;
    const uint32_t ulen = blob[12 + hlen] | blob[13 + hlen] << 8 | blob[14 + hlen] << 16 | (uint32_t)blob[15 + hlen] << 24;
    const size_t kern_off = user_off + ulen;
    const uint32_t interesting[] = {0, 1, 3, 4, 0x7fffffffu, 0x80000000u, 0xbffffffcu, 0xc0000000u, 0xfffffff0u, 0xfffffffcu, 0xffffffffu, 0x10000u, 0x3ff, 0x400};
    long accepted = 0, rejected = 0;
    const time_t t0 = time(nullptr);
    for (long it = 0; it < iters; ++it) {
        std::vector<uint8_t> m = blob;
        const size_t elf = (rnd() & 1) ? user_off : kern_off;
        switch (rnd() % 8) {
            case 0: m.resize(rnd() % m.size()); break;                                            // truncation
            case 1: for (int k = 0; k < 1 + (int)(rnd() % 8); ++k) m[rnd() % 64] ^= (uint8_t)(1u << (rnd() % 8)); break;  // R0BF header
            case 2: for (int k = 0; k < 1 + (int)(rnd() % 4); ++k) m[elf + rnd() % 52] ^= (uint8_t)(1u << (rnd() % 8)); break;  // ELF header
            case 3: {  // one program-header field of one segment set to a hostile value
                const size_t ph = elf + 52 + 32 * (rnd() % 7) + 4 * (rnd() % 8);
                wr32(m, ph, interesting[rnd() % (sizeof interesting / 4)]);
                break;
            }
            case 4: {  // gigabytes of .bss / a wrapping range
                const size_t ph = elf + 52 + 32 * (1 + rnd() % 4);
                wr32(m, ph + 8, interesting[rnd() % (sizeof interesting / 4)] & ~3u);  // vaddr
                wr32(m, ph + 20, (uint32_t)rnd());                                        // memsz
                if (rnd() & 1) wr32(m, ph + 16, (uint32_t)rnd());                         // filesz
                break;
            }
            case 5: wr32(m, elf + 44, (uint32_t)(rnd() % 65536)); wr32(m, elf + 28, (uint32_t)rnd()); break;  // phnum, phoff
            case 6: wr32(m, 12 + hlen, (uint32_t)rnd()); break;                               // user ELF length
            default: for (int k = 0; k < 16; ++k) m[rnd() % m.size()] = (uint8_t)rnd(); break;  // scattered garbage
        }
        bx_image* g = nullptr;
        const char* e = bx_image_from_program(nullptr, m.data(), m.size(), &g);
        if (e) {
            if (g) { printf("error AND an image returned\n"); return 1; }
            ++rejected;
            continue;
        }
        ++accepted;
        // walk it the way bx_image_root would: indices ascending, every page readable
        const size_t np = bx_image_page_count(g);
        std::vector<uint32_t> idx(np ? np : 1);
        if (bx_image_page_indices(g, idx.data(), np) != np) { printf("page count changed\n"); return 1; }
        uint32_t page[BX_PAGE_WORDS];
        for (size_t k = 0; k < np; k += (np / 16) + 1) {
            if (k && idx[k] <= idx[k - 1]) { printf("page indices not ascending\n"); return 1; }
            if (bx_image_get_page(g, idx[k], page)) { printf("page unreadable\n"); return 1; }
        }
        if (np > honest_pages + 4096) { printf("a %zu-byte file produced %zu pages\n", m.size(), np); return 1; }  // pages come from file bytes only
        bx_image_free(g);
    }
    const long secs = (long)(time(nullptr) - t0);
    if (secs > 120) { printf("too slow: %ld s\n", secs); return 1; }
    printf("image_fuzz_check ok: %ld accepted, %ld rejected, %ld s\n", accepted, rejected, secs);
    return 0;
}
