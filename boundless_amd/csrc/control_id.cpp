// control_id.cpp — the verifier's half of the code-group binding (include/bx_circuit.h): VerifierContext, the synthetic
// circuit's check_code, and the definition-level host computation of its control IDs.  Host code only.
//
// Reference: `segment_receipt.verify_integrity_with_context(&agent.verifier_ctx)` (bento/crates/workflow/src/tasks/prove.rs:53-55,
// `verifier_ctx` built at lib.rs:241) ends in risc0_zkp::verify::verify(circuit, suite, seal, check_code) [EXT: risc0-zkp 3.0.3],
// whose `check_code(po2, root)` closure refuses a seal whose code-group Merkle root is not one of the circuit's control IDs
// (risc0-zkvm: `verifier_parameters.control_ids.contains(control_id)`); the generated table one level up is
// contracts/src/blake3-groth16/ControlID.sol:13.  Without that comparison the selectors and control words a seal's constraints
// are evaluated with are the prover's own choice (Weak #2: a code group with `last` == 0 proves any g_1).
//
// The host computation below shares no code with the device path (ntt.hip / poseidon2.hip) nor with the test oracle: plain
// iterative radix-2 transforms in natural order — coefficients by an inverse DFT on <w_N>, the coset shift as a_j * 3^j, the 4N
// evaluations by a zero-padded forward DFT (row i = f(3 * w_4N^i)) — then the row sponge and the pair hashes of transcript.hpp.
#include <stdio.h>
#include <string.h>

#include <array>
#include <map>
#include <mutex>
#include <thread>
#include <vector>

#include "../../include/bx_circuit.h"
#include "circuit.hpp"
#include "fp.hpp"
#include "poseidon2_params.hpp"
#include "transcript.hpp"

namespace {
using namespace bx;
using Digest = std::array<uint32_t, 8>;

uint32_t root_of_unity(unsigned k) { return fp_pow(fp_encode(137u), (uint64_t)1 << (27 - k)); }  // w_{2^k}, Montgomery

// out[i] = sum_j a[j] * w^(i j), natural order in and out (bit-reversal permutation + decimation-in-time butterflies)
void dft_natural(uint32_t* a, unsigned log_n, uint32_t w) {
    const size_t n = (size_t)1 << log_n;
    for (size_t i = 0; i < n; ++i) {
        const size_t j = log_n ? (size_t)(bit_reverse32((uint32_t)i) >> (32 - log_n)) : 0;
        if (i < j) std::swap(a[i], a[j]);
    }
    std::vector<uint32_t> tw(n / 2 ? n / 2 : 1);
    for (unsigned s = 1; s <= log_n; ++s) {
        const size_t half = (size_t)1 << (s - 1);
        const uint32_t ws = fp_pow(w, (uint64_t)1 << (log_n - s));  // primitive 2^s-th root
        tw[0] = MONT_ONE;
        for (size_t k = 1; k < half; ++k) tw[k] = fp_mul(tw[k - 1], ws);
        for (size_t base = 0; base < n; base += 2 * half)
            for (size_t k = 0; k < half; ++k) {
                const uint32_t u = a[base + k], v = fp_mul(a[base + k + half], tw[k]);
                a[base + k] = fp_add(u, v);
                a[base + k + half] = fp_sub(u, v);
            }
    }
}

unsigned worker_count(size_t items) {
    unsigned hw = std::thread::hardware_concurrency();
    if (hw == 0) hw = 1;
    if (hw > 16) hw = 16;
    return (unsigned)(items < hw ? (items ? items : 1) : hw);
}
template <class F>
void parallel_ranges(size_t n, F&& body) {  // body(begin, end)
    const unsigned t = worker_count(n / 1024 + 1);
    if (t <= 1) {
        body((size_t)0, n);
        return;
    }
    std::vector<std::thread> th;
    const size_t chunk = (n + t - 1) / t;
    for (unsigned i = 0; i < t; ++i) {
        const size_t b = (size_t)i * chunk, e = b + chunk < n ? b + chunk : n;
        if (b < e) th.emplace_back([&body, b, e] { body(b, e); });
    }
    for (auto& x : th) x.join();
}

// Merkle root of the committed code group of the synthetic circuit for (po2, w_code)
const char* host_control_id(uint32_t po2, uint32_t wc, uint32_t out[8]) {
    if (po2 < 9 || po2 > 24 || wc < 1 || wc >= 65536) return "bx_synthetic_control_id_host: shape out of range";
    const size_t n = (size_t)1 << po2, dom = 4 * n;
    if ((double)wc * (double)dom * 4.0 > 6.0e9) return "bx_synthetic_control_id_host: the code group is too large for the host computation (build a verifier context from bx_prover_control_id instead)";
    const Circuit cc(po2, wc, 1, 1, 1, 1);  // only po2 / w_code enter the code group
    std::vector<uint32_t> ev((size_t)wc * dom);
    const uint32_t w_n_inv = fp_inv(root_of_unity(po2)), w_dom = root_of_unity(po2 + 2), n_inv = fp_inv(fp_encode((uint32_t)n));
    {
        std::vector<std::thread> th;
        const unsigned t = worker_count(wc);
        for (unsigned k = 0; k < t; ++k)
            th.emplace_back([&, k] {
                for (uint32_t c = k; c < wc; c += t) {
                    uint32_t* col = &ev[(size_t)c * dom];
                    for (size_t r = 0; r < n; ++r) col[r] = synth_code_cell(cc, c, (uint32_t)r);
                    dft_natural(col, po2, w_n_inv);  // coefficients * N
                    uint32_t shift = n_inv;          // 3^j / N
                    for (size_t j = 0; j < n; ++j) {
                        col[j] = fp_mul(col[j], shift);
                        shift = fp_mul(shift, MONT_THREE);
                    }
                    memset(col + n, 0, (dom - n) * 4);
                    dft_natural(col, po2 + 2, w_dom);  // col[i] = f(3 w_4N^i)
                }
            });
        for (auto& x : th) x.join();
    }
    HostPoseidon2 h;
    h.load(POSEIDON2_RC, POSEIDON2_DIAG);
    std::vector<uint32_t> layer(8 * dom);
    parallel_ranges(dom, [&](size_t b, size_t e) {
        std::vector<uint32_t> row(wc);
        for (size_t r = b; r < e; ++r) {
            for (uint32_t c = 0; c < wc; ++c) row[c] = ev[(size_t)c * dom + r];
            h.hash_elems(&layer[8 * r], row.data(), wc);
        }
    });
    ev.clear();
    ev.shrink_to_fit();
    for (size_t sz = dom; sz > 1; sz >>= 1) {
        std::vector<uint32_t> next(8 * (sz / 2));
        parallel_ranges(sz / 2, [&](size_t b, size_t e) {
            for (size_t i = b; i < e; ++i) h.hash_elems(&next[8 * i], &layer[16 * i], 16);  // hash_pair == sponge of 16 words
        });
        layer.swap(next);
    }
    memcpy(out, layer.data(), 32);
    return nullptr;
}

// ---- the generated table (tools/gen_control_ids.py): w_code = 16, po2 9..24 — upstream ships the same thing as control_id.rs ----
struct TableRow {
    uint32_t po2;
    uint32_t id[8];
};
#if !defined(BX_NO_CONTROL_TABLE)
const TableRow SYNTH_CONTROL_IDS_W16[] = {
#include "control_ids_w16.inc"
};
const size_t N_TABLE = sizeof SYNTH_CONTROL_IDS_W16 / sizeof SYNTH_CONTROL_IDS_W16[0];
#else
const TableRow* const SYNTH_CONTROL_IDS_W16 = nullptr;
const size_t N_TABLE = 0;
#endif

std::mutex cache_mu;
std::map<std::pair<uint32_t, uint32_t>, Digest> cache;  // (po2, w_code) -> host-computed control ID

const char* synth_control_id(uint32_t po2, uint32_t wc, uint32_t out[8], bool use_table) {
    if (use_table && wc == 16)
        for (size_t i = 0; i < N_TABLE; ++i)
            if (SYNTH_CONTROL_IDS_W16[i].po2 == po2) {
                memcpy(out, SYNTH_CONTROL_IDS_W16[i].id, 32);
                return nullptr;
            }
    {
        std::lock_guard<std::mutex> g(cache_mu);
        auto it = cache.find({po2, wc});
        if (it != cache.end()) {
            memcpy(out, it->second.data(), 32);
            return nullptr;
        }
    }
    Digest d;
    if (const char* e = host_control_id(po2, wc, d.data())) return e;
    std::lock_guard<std::mutex> g(cache_mu);
    cache[{po2, wc}] = d;
    memcpy(out, d.data(), 32);
    return nullptr;
}
}  // namespace

namespace bx {
const char* synth_check_code(void*, const bx_segment_params* s, const uint32_t root[8]) {
    if (!s || !root) return "check_code: null argument";
    uint32_t id[8];
    if (const char* e = synth_control_id(s->po2, s->w_code, id, true)) return e;
    return memcmp(id, root, 32) == 0 ? nullptr : "the code group's root is not this circuit's control ID for the shape (the seal was made with another code group)";
}
}  // namespace bx

struct bx_verifier_ctx {
    std::vector<std::pair<uint32_t, Digest>> ids;
};

extern "C" {

const char* bx_synthetic_control_id_host(uint32_t po2, uint32_t w_code, uint32_t id_out[8]) {
    static thread_local char err[256];
    if (!id_out) return "bx_synthetic_control_id_host: null output";
    try {
        if (const char* e = synth_control_id(po2, w_code, id_out, false)) {
            snprintf(err, sizeof err, "%s", e);
            return err;
        }
        return nullptr;
    } catch (const std::exception& e) {
        snprintf(err, sizeof err, "bx_synthetic_control_id_host: %s", e.what());
        return err;
    }
}

const char* bx_verifier_ctx_create(bx_verifier_ctx** out) {
    if (!out) return "bx_verifier_ctx_create: out is NULL";
    *out = new (std::nothrow) bx_verifier_ctx();
    return *out ? nullptr : "bx_verifier_ctx_create: out of memory";
}
void bx_verifier_ctx_destroy(bx_verifier_ctx* v) { delete v; }
const char* bx_verifier_ctx_add_control_id(bx_verifier_ctx* v, uint32_t po2, const uint32_t id[8]) {
    if (!v || !id) return "bx_verifier_ctx_add_control_id: null argument";
    if (po2 < 9 || po2 > 24) return "bx_verifier_ctx_add_control_id: po2 must be in [9, 24]";
    for (int i = 0; i < 8; ++i)
        if (id[i] >= bx::P) return "bx_verifier_ctx_add_control_id: a digest word is not a canonical field element";
    try {
        Digest d;
        memcpy(d.data(), id, 32);
        for (auto& e : v->ids)
            if (e.first == po2 && e.second == d) return nullptr;  // a set
        v->ids.emplace_back(po2, d);
        return nullptr;
    } catch (const std::exception&) {
        return "bx_verifier_ctx_add_control_id: out of memory";
    }
}
size_t bx_verifier_ctx_size(const bx_verifier_ctx* v) { return v ? v->ids.size() : 0; }
size_t bx_verifier_ctx_count(const bx_verifier_ctx* v, uint32_t po2) {
    size_t n = 0;
    if (v)
        for (auto& e : v->ids) n += e.first == po2;
    return n;
}

}  // extern "C"

namespace bx {
// what verify.cpp asks a context: is `root` one of the IDs registered for po2?
bool verifier_ctx_contains(const bx_verifier_ctx* v, uint32_t po2, const uint32_t root[8]) {
    for (auto& e : v->ids)
        if (e.first == po2 && memcmp(e.second.data(), root, 32) == 0) return true;
    return false;
}
}  // namespace bx
