// prover.hip — the segment prover under `ProverServer::prove_segment` (include/bx_prover.h), host side in C++
// over the HAL entry points of include/bx_hal.h, plus the few device kernels that stand in for circuit code.
//
// Follows risc0_zkp::prove::{Prover::commit_group, Prover::finalize, fri::fri_prove, merkle::MerkleTreeProver,
// poly_group::PolyGroup} (risc0-zkp 3.0.3, reference Cargo.lock:9155) as called by
// bento/crates/workflow/src/tasks/prove.rs:41-49.  The call sequence, constants (INV_RATE 4, FRI_FOLD 16,
// FRI_MIN_DEGREE 256, QUERIES 50, CHECK_SIZE 16), Merkle top-layer rule and transcript order are upstream's; the
// circuit (witness generation, accumulate, eval_check: circuit.hip) is the synthetic one specified in bx_prover.h.
#include <algorithm>
#include <memory>
#include <mutex>

#include "circuit.hpp"
#include "ctx.hpp"
#include "transcript.hpp"
#include "../../include/bx_circuit.h"

namespace bx {


// MerkleTreeProver::prove for a batch of queries (one workgroup per query).
__global__ void merkle_query_gather_kernel(uint32_t* __restrict__ out, const uint32_t* __restrict__ matrix,
                                           const uint32_t* __restrict__ nodes, uint32_t rows, uint32_t cols,
                                           const uint32_t* __restrict__ positions, uint32_t depth) {
    uint32_t q = blockIdx.x;
    uint32_t pos = positions[q];
    uint32_t* o = out + (size_t)q * (cols + 8u * depth);
    for (uint32_t c = threadIdx.x; c < cols; c += blockDim.x) o[c] = matrix[(size_t)c * rows + pos];
    for (uint32_t w = threadIdx.x; w < 8u * depth; w += blockDim.x) {
        uint32_t level = w >> 3;
        uint32_t idx = ((pos + rows) >> level) ^ 1u;
        o[cols + w] = nodes[(size_t)idx * 8 + (w & 7u)];
    }
}

static inline unsigned top_layer_of(unsigned layers) {
    unsigned top = 0;
    for (unsigned i = 1; i < layers; ++i) {
        if ((1u << i) > BX_QUERIES) break;
        top = i;
    }
    return top;
}

struct DevBuf {  // owning device allocation; movable, not copyable
    bx_ctx* c = nullptr;
    bx_buf b{nullptr, 0};
    DevBuf() = default;
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    DevBuf(DevBuf&& o) noexcept : c(o.c), b(o.b) { o.b = bx_buf{nullptr, 0}; }
    DevBuf& operator=(DevBuf&& o) noexcept {
        if (this != &o) {
            if (b.dptr) (void)hipFree(b.dptr);
            c = o.c;
            b = o.b;
            o.b = bx_buf{nullptr, 0};
        }
        return *this;
    }
    const char* alloc(bx_ctx* ctx, size_t words) {
        c = ctx;
        return raw_alloc(ctx, words, &b);  // long-lived (the prover's lifetime), freed with hipFree in the destructor: not pooled
    }
    ~DevBuf() {
        if (b.dptr) (void)hipFree(b.dptr);
    }
    bx_buf slice(size_t off, size_t len) const { return bx_buf{(uint32_t*)b.dptr + off, len}; }
};

struct Tree {
    size_t rows = 0, cols = 0;
    unsigned layers = 0, top_layer = 0;
    DevBuf nodes;
    uint32_t root[8];
    size_t top_size() const { return (size_t)1 << top_layer; }
    unsigned depth() const { return layers - top_layer; }
    size_t query_words() const { return cols + 8u * depth(); }
};

struct Group {
    uint32_t width = 0;
    DevBuf coeffs, evaluated, combo_ids;
    Tree tree;
    std::vector<std::vector<uint32_t>> backs;  // tap set per column: the rows back it is opened at (backs[c][0] == 0)
    std::vector<uint32_t> combo;               // combo of each column (columns with the same tap set share one)
};

// One of the prover's two staging slots for a segment's bytes (bx_prover_submit_segment): pinned host copy -> HBM copy on the
// prover's copy stream; `up` is recorded behind the upload and the compute stream waits on it, never the host.
struct SegSlot {
    uint8_t* host = nullptr;
    size_t host_cap = 0;
    uint32_t* dev = nullptr;
    size_t dev_cap = 0;  // bytes
    size_t len = 0;
    hipEvent_t up0 = nullptr, up = nullptr;
};

struct FriRound {
    size_t size = 0;  // ext coefficients entering the round
    DevBuf evaluated, out_coeffs;
    Tree tree;
};

}  // namespace bx

using namespace bx;

struct bx_prover {
    bx_ctx* c = nullptr;
    bx_segment_params shape{};  // normalised by the circuit (defaults filled in)
    const bx_circuit_ops* circ = nullptr;  // the circuit half: witgen / accumulate / eval_check / tap set (bx_circuit.h)
    void* circ_state = nullptr;
    size_t N = 0;
    bool coeffs_bitrev = false;  // trace coefficients stay in bit-reversed order (N >= 2^15), see commit_group
    HostPoseidon2 h2;
    Group groups[4];  // code, data, accum, check
    DevBuf combos, final_poly, which, xs, evals, rems, positions, qout;
    DevBuf code_w;  // the code group's WITNESS (what witgen reads); groups[0].coeffs is interpolated in place by the commit, which is
                    // enqueued before the segment's bytes have arrived (prove_prologue)
    DevBuf tap_ptrs, tap_flags;  // per tap evaluation: the device address of its coefficient column and its storage order (N >= 2^15)
    std::vector<std::vector<uint32_t>> combo_backs;  // trace combos in order of first appearance; the check combo comes after them
    std::vector<uint32_t> tap_which;                 // polynomial index of every tap evaluation (fixed per shape)
    size_t tap_first[5] = {0, 0, 0, 0, 0};           // first tap evaluation of each group
    size_t n_div = 0;                                // DEEP divisions per proof
    uint32_t n_globals = 0;                          // public words of the statement (bx_circuit_ops::n_globals)
    ~bx_prover() {
        if (circ && circ_state && circ->destroy) circ->destroy(circ->user, circ_state);
        if (copy_stream) (void)hipStreamSynchronize(copy_stream);  // a segment submitted and never proved may still be on its way up
        for (SegSlot& sl : seg) {
            if (sl.host) (void)hipHostFree(sl.host);
            if (sl.dev) (void)hipFree(sl.dev);
            if (sl.up0) (void)hipEventDestroy(sl.up0);
            if (sl.up) (void)hipEventDestroy(sl.up);
        }
        if (copy_stream) (void)hipStreamDestroy(copy_stream);
    }
    std::vector<FriRound> rounds;
    DevBuf final_coeffs;
    DevBuf tstate, dev_chal;  // device half of the transcript: 24 cells + pool counter; the challenges it drew (4 words per step)
    uint32_t last_roots[32];
    size_t seal_bound = 0;
    char err[512];
    bool prologue_done = false;  // prove_prologue was already enqueued for the next bx_prove_submitted
    char seg_err[256];  // bx_prover_submit_segment may run on another thread than the proof: its own message buffer
    // segment staging: two slots, oldest first (submit may run on another thread than prove_submitted)
    SegSlot seg[2];
    int seg_head = 0, seg_count = 0;
    std::mutex seg_mu;
    hipStream_t copy_stream = nullptr;
    float last_upload_ms = 0;
    size_t last_upload_bytes = 0;
};

namespace {

const char* perr(bx_prover* p, const char* m) {
    snprintf(p->err, sizeof p->err, "%s", m);
    return p->err;
}
#define PV(expr)                                  \
    do {                                          \
        const char* _m = (expr);                  \
        if (_m) return perr(p, _m);               \
    } while (0)

const char* tree_init(bx_ctx* c, Tree& t, size_t rows, size_t cols) {
    t.rows = rows;
    t.cols = cols;
    t.layers = (unsigned)ilog2(rows);
    t.top_layer = top_layer_of(t.layers);
    return t.nodes.alloc(c, 16 * rows);
}

// MerkleTreeProver::new: leaves + every layer, enqueued; nothing is read back
const char* tree_build(bx_prover* p, Tree& t, bx_buf matrix) {
    PV(bx_merkle_build(p->c, t.nodes.b, matrix, t.rows));
    return nullptr;
}
// the root nodes[1] and the top layer nodes[top..2*top) are the two ends of one contiguous run: one copy per tree
const char* tree_fetch(bx_prover* p, Tree& t, size_t* used, const uint32_t** host) {
    PV(d2h_batch_add(p->c, used, t.nodes.slice(8, 8 * (2 * t.top_size() - 1)), 8 * (2 * t.top_size() - 1), host));
    return nullptr;
}
// MerkleTreeProver::commit from the fetched run: the top layer goes to the seal, the root into the transcript
void tree_absorb(Tree& t, const uint32_t* host, Transcript& T) {
    const size_t top = t.top_size();
    memcpy(t.root, host, 32);
    T.write(host + 8 * (top - 1), 8 * top);
    T.commit(t.root);
}
// build + one round trip + commit (the commits whose root the next stage's challenge needs at once)
const char* tree_commit(bx_prover* p, Tree& t, bx_buf matrix, Transcript& T) {
    PV(tree_build(p, t, matrix));
    size_t used = 0;
    const uint32_t* host = nullptr;
    PV(tree_fetch(p, t, &used, &host));
    PV(d2h_batch_wait(p->c));
    tree_absorb(t, host, T);
    return nullptr;
}

// the mixed u polynomials (one ext coefficient per tap of the combo) come off the low end of the combination polynomials
struct SubLow {
    uint32_t v[BX_MAX_COMBOS * BX_MAX_TAPS * 4];
};
__global__ void sub_low_kernel(uint32_t* __restrict__ combos, uint32_t combo_words, SubLow s, uint32_t n_combos) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_combos * BX_MAX_TAPS * 4) {
        uint32_t* p = combos + (size_t)(i / (BX_MAX_TAPS * 4)) * combo_words + (i % (BX_MAX_TAPS * 4));  // natural-order AoS: coefficient t at 4t
        *p = fp_sub(*p, s.v[i]);  // unused slots hold 0
    }
}

// Prover::commit_group, device half: interpolate -> zk_shift -> PolyGroup::new (expand+evaluate, bit_reverse, Merkle); no read-back
const char* commit_group_work(bx_prover* p, Group& g) {
    bx_ctx* c = p->c;
    PV(bx_batch_interpolate_zk(c, g.coeffs.b, g.width));  // = batch_interpolate_ntt + zk_shift
    PV(bx_batch_expand_into_evaluate_ntt(c, g.evaluated.b, g.coeffs.b, g.width, 2));
    // PolyGroup::new bit-reverses the coefficients to natural order here.  At BASELINE sizes the trace coefficients stay
    // bit-reversed instead (p->coeffs_bitrev): the taps are evaluated by bx_batch_evaluate_any_bitrev, the DEEP mix is
    // element-wise and therefore order-agnostic, and only the two combination polynomials are bit-reversed afterwards —
    // 32 words per row moved instead of 336.
    if (!p->coeffs_bitrev) PV(bx_batch_bit_reverse(c, g.coeffs.b, g.width));
    return tree_build(p, g.tree, g.evaluated.b);
}
const char* commit_group(bx_prover* p, Group& g, Transcript& T) {
    PV(commit_group_work(p, g));
    size_t used = 0;
    const uint32_t* host = nullptr;
    PV(tree_fetch(p, g.tree, &used, &host));
    PV(d2h_batch_wait(p->c));
    tree_absorb(g.tree, host, T);
    return nullptr;
}

Fp4 host_pow(Fp4 a, uint64_t e) { return f4_pow(a, e); }

}  // namespace

extern "C" const char* bx_merkle_query_gather(bx_ctx* c, bx_buf out, bx_buf matrix, bx_buf nodes, size_t rows, size_t cols,
                                              bx_buf positions, size_t n_queries, size_t top_size) try {
    if (!c) return "bx_merkle_query_gather: null ctx";
    BX_REQUIRE(c, is_pow2(rows) && is_pow2(top_size) && top_size <= rows, "merkle_query_gather: rows/top_size must be powers of two");
    BX_REQUIRE(c, matrix.len == rows * cols && nodes.len == 16 * rows, "merkle_query_gather: matrix/nodes size mismatch");
    unsigned depth = (unsigned)(ilog2(rows) - ilog2(top_size));
    BX_REQUIRE(c, out.len >= n_queries * (cols + 8 * depth) && positions.len >= n_queries, "merkle_query_gather: out/positions too small");
    BX_ENTER(c);
    if (!n_queries) return nullptr;
    OpScope op(c, "merkle_query_gather", 4.0 * (double)(n_queries * (cols + 8 * depth)) * 2.0);
    hipLaunchKernelGGL(merkle_query_gather_kernel, dim3((unsigned)n_queries), dim3(256), 0, c->stream, (uint32_t*)out.dptr,
                       (const uint32_t*)matrix.dptr, (const uint32_t*)nodes.dptr, (uint32_t)rows, (uint32_t)cols,
                       (const uint32_t*)positions.dptr, depth);
    BX_LAUNCH_CHECK(c);
    return nullptr;
} BX_ABI_CATCH(c, "bx_merkle_query_gather")

extern "C" const char* bx_prover_create(bx_ctx* c, const bx_segment_params* shape, bx_prover** out) try {
    return bx_prover_create_with_circuit(c, shape, nullptr, out);
} BX_ABI_CATCH(c, "bx_prover_create")
extern "C" const char* bx_prover_create_with_circuit(bx_ctx* c, const bx_segment_params* shape, const bx_circuit_ops* circuit, bx_prover** out) try {
    if (!c) return "bx_prover_create: null ctx";
    BX_REQUIRE(c, shape && out, "bx_prover_create: null argument");
    if (!circuit) circuit = bx_synthetic_circuit();
    BX_REQUIRE(c, circuit->taps && circuit->code_group && circuit->witgen && circuit->accumulate && circuit->eval_check,
               "bx_prover_create: circuit table incomplete");
    BX_REQUIRE(c, shape->po2 >= 9 && shape->po2 <= 24, "bx_prover_create: po2 must be in [9, 24]");
    BX_REQUIRE(c, shape->w_code >= 1 && shape->w_data >= 1 && shape->w_accum >= 1, "bx_prover_create: every group needs at least one column");
    BX_REQUIRE(c, shape->w_code < 65536 && shape->w_data < 65536 && shape->w_accum < 65536, "bx_prover_create: group width out of range");

    BX_ENTER(c);
    std::unique_ptr<bx_prover> p(new (std::nothrow) bx_prover());
    BX_REQUIRE(c, p != nullptr, "bx_prover_create: out of host memory");
    p->c = c;
    p->shape = *shape;
    p->circ = circuit;
    if (circuit->normalize)
        if (const char* e = circuit->normalize(circuit->user, &p->shape)) return set_msg(c, e);
    p->N = (size_t)1 << shape->po2;
    p->coeffs_bitrev = shape->po2 >= 15 && c->deep_bitrev;
    p->err[0] = 0;
    p->h2.load(c->h_rc, c->h_diag);
    const size_t N = p->N, D = 4 * N;
    const uint32_t widths[4] = {shape->w_code, shape->w_data, shape->w_accum, BX_CHECK_SIZE};
    size_t total_taps = 0, max_w = 0;
    for (int g = 0; g < 4; ++g) {
        Group& G = p->groups[g];
        G.width = widths[g];
        max_w = std::max<size_t>(max_w, G.width);
        BX_TRY(G.coeffs.alloc(c, (size_t)G.width * N));
        BX_TRY(G.evaluated.alloc(c, (size_t)G.width * D));
        BX_TRY(tree_init(c, G.tree, D, G.width));
        // the circuit's tap set (bx_circuit.h): every column is opened at Z and at the rows back its set lists; columns with
        // the same set share a combo, combos are numbered in order of first appearance, the check group's comes last
        G.backs.resize(G.width);
        G.combo.resize(G.width);
        for (uint32_t col = 0; col < G.width; ++col) {
            if (g == 3) {
                G.backs[col] = {0};
                continue;
            }
            uint32_t bk[BX_MAX_TAPS];
            const uint32_t k = circuit->taps(circuit->user, &p->shape, g, col, bk);
            BX_REQUIRE(c, k >= 1 && k <= BX_MAX_TAPS && bk[0] == 0, "bx_prover_create: a tap set has 1..8 entries and starts with 0");
            for (uint32_t t = 1; t < k; ++t) BX_REQUIRE(c, bk[t] > bk[t - 1] && bk[t] < N, "bx_prover_create: tap sets are strictly increasing");
            G.backs[col].assign(bk, bk + k);
            size_t id = 0;
            while (id < p->combo_backs.size() && p->combo_backs[id] != G.backs[col]) ++id;
            if (id == p->combo_backs.size()) p->combo_backs.push_back(G.backs[col]);
            G.combo[col] = (uint32_t)id;
        }
    }
    BX_REQUIRE(c, p->combo_backs.size() + 1 <= BX_MAX_COMBOS, "bx_prover_create: too many distinct tap sets");
    const size_t n_combos = p->combo_backs.size() + 1;
    for (int g = 0; g < 4; ++g) {
        Group& G = p->groups[g];
        if (g == 3) G.combo.assign(G.width, (uint32_t)(n_combos - 1));
        BX_TRY(G.combo_ids.alloc(c, G.width));
        BX_TRY(bx_h2d(c, G.combo_ids.b, G.combo.data(), G.width));
        for (uint32_t col = 0; col < G.width; ++col) {
            for (size_t t = 0; t < G.backs[col].size(); ++t) p->tap_which.push_back(col);
            total_taps += G.backs[col].size();
        }
        p->tap_first[g + 1] = p->tap_which.size();
    }
    p->n_div = 1;
    for (auto& cb : p->combo_backs) p->n_div += cb.size();
    if (circuit->create)
        if (const char* e = circuit->create(circuit->user, c, &p->shape, &p->circ_state)) return e == c->err ? e : set_msg(c, e);
    p->n_globals = circuit->n_globals ? circuit->n_globals(circuit->user, &p->shape) : 0;
    BX_REQUIRE(c, p->n_globals <= BX_MAX_GLOBALS, "bx_prover_create: too many public words");
    BX_TRY(p->code_w.alloc(c, (size_t)shape->w_code * N));
    BX_TRY(p->combos.alloc(c, n_combos * 4 * N));
    BX_TRY(p->final_poly.alloc(c, 4 * N));
    // tap evaluations of all four groups go up, run and come back as one batch (one host round trip instead of twelve)
    BX_TRY(p->which.alloc(c, total_taps));
    BX_TRY(p->xs.alloc(c, 4 * total_taps));
    BX_TRY(p->evals.alloc(c, 4 * total_taps));
    BX_TRY(p->rems.alloc(c, 4 * p->n_div));
    BX_TRY(bx_h2d(c, p->which.slice(0, total_taps), p->tap_which.data(), total_taps));  // fixed per shape
    if (N >= ((size_t)1 << 15)) {  // all four groups' taps in one launch set (bx_batch_evaluate_ptrs)
        std::vector<uint32_t> ptrs(2 * total_taps), flags(total_taps);
        size_t e = 0;
        for (int g = 0; g < 4; ++g)
            for (size_t t = p->tap_first[g]; t < p->tap_first[g + 1]; ++t, ++e) {
                const unsigned long long a = (unsigned long long)(uintptr_t)((uint32_t*)p->groups[g].coeffs.b.dptr + (size_t)p->tap_which[t] * N);
                ptrs[2 * e] = (uint32_t)a;
                ptrs[2 * e + 1] = (uint32_t)(a >> 32);
                flags[e] = (p->coeffs_bitrev && g < 3) ? 1u : 0u;
            }
        BX_TRY(p->tap_ptrs.alloc(c, 2 * total_taps));
        BX_TRY(p->tap_flags.alloc(c, total_taps));
        BX_TRY(bx_h2d(c, p->tap_ptrs.b, ptrs.data(), ptrs.size()));
        BX_TRY(bx_h2d(c, p->tap_flags.b, flags.data(), flags.size()));
    }
    // FRI rounds
    size_t size = N;
    size_t fri_query_words = 0;
    p->rounds.reserve(8);
    while (size > BX_FRI_MIN_DEGREE) {
        p->rounds.emplace_back();
        FriRound& r = p->rounds.back();
        r.size = size;
        BX_TRY(r.evaluated.alloc(c, 4 * 4 * size));
        BX_TRY(r.out_coeffs.alloc(c, 4 * size / BX_FRI_FOLD));
        BX_TRY(tree_init(c, r.tree, 4 * size / BX_FRI_FOLD, 4 * BX_FRI_FOLD));
        fri_query_words += r.tree.query_words();
        size /= BX_FRI_FOLD;
    }
    BX_TRY(p->final_coeffs.alloc(c, 4 * size));
    BX_TRY(p->tstate.alloc(c, 32));
    BX_TRY(p->dev_chal.alloc(c, 4 * (p->rounds.size() + 4)));
    size_t max_q = 0, trace_query_words = 0;
    for (int g = 0; g < 4; ++g) {
        max_q = std::max(max_q, p->groups[g].tree.query_words());
        trace_query_words += p->groups[g].tree.query_words();
    }
    for (auto& r : p->rounds) max_q = std::max(max_q, r.tree.query_words());
    // the 50 openings of every tree are gathered into one buffer and copied back together
    BX_TRY(p->positions.alloc(c, BX_QUERIES * (4 + p->rounds.size())));
    BX_TRY(p->qout.alloc(c, (trace_query_words + fri_query_words) * BX_QUERIES));
    // seal bound: header + tops + coeff_u + final coeffs + queries
    size_t bound = BX_SEAL_HEADER_WORDS + p->n_globals;
    for (int g = 0; g < 4; ++g) bound += 8 * p->groups[g].tree.top_size();
    for (auto& r : p->rounds) bound += 8 * r.tree.top_size();
    bound += 4 * total_taps + 4 * size + BX_QUERIES * (trace_query_words + fri_query_words);
    p->seal_bound = bound;
    memset(p->last_roots, 0, sizeof p->last_roots);
    BX_HIP(c, hipStreamCreateWithFlags(&p->copy_stream, hipStreamNonBlocking));
    for (SegSlot& sl : p->seg) {
        BX_HIP(c, hipEventCreate(&sl.up0));
        BX_HIP(c, hipEventCreate(&sl.up));
    }
    BX_TRY(bx_sync(c));
    c->live_provers += 1;
    *out = p.release();
    return nullptr;
} BX_ABI_CATCH(c, "bx_prover_create_with_circuit")

extern "C" const char* bx_prover_destroy(bx_prover* p) try {
    if (!p) return nullptr;
    (void)hipSetDevice(p->c->device);
    (void)stream_wait(p->c);
    p->c->live_provers -= 1;
    delete p;
    return nullptr;
} BX_ABI_CATCH(nullptr, "bx_prover_destroy")  // p may be gone by then
extern "C" size_t bx_prover_seal_words(const bx_prover* p) { return p ? p->seal_bound : 0; }
extern "C" const char* bx_prover_last_roots(const bx_prover* p, uint32_t roots_out[32]) try {
    if (!p) return "bx_prover_last_roots: null prover";
    memcpy(roots_out, p->last_roots, sizeof p->last_roots);
    return nullptr;
} BX_ABI_CATCH((p ? p->c : nullptr), "bx_prover_last_roots")

static const char* prove_segment_impl(bx_prover* p, const SegSlot& seg, uint32_t* seal_out, size_t seal_cap, size_t* seal_words);

// What a proof can start before its segment has arrived: the code group (a function of the shape) and its whole commitment —
// 2.2 ms of device work at 2^20 that the upload of the segment's bytes hides behind.  The witness of the code group is kept in
// code_w for witgen (the commit interpolates groups[0].coeffs in place).  Enqueues only.
static const char* prove_prologue(bx_prover* p) {
    bx_ctx* c = p->c;
    const bx_circuit_ops* circ = p->circ;
    TraceRange tr(c, "bx:commit_code");
    PV(circ->code_group(circ->user, p->circ_state, c, p->code_w.b));
    PV(bx_eltwise_copy_elem(c, p->groups[0].coeffs.b, p->code_w.b));
    PV(commit_group_work(p, p->groups[0]));
    return nullptr;
}

// ---- the segment's bytes: pinned staging + upload on the copy stream (two slots) ----
static const char* serr(bx_prover* p, const char* m) {
    snprintf(p->seg_err, sizeof p->seg_err, "%s", m);
    return p->seg_err;
}
extern "C" const char* bx_prover_submit_segment(bx_prover* p, const uint8_t* segment, size_t len) try {
    if (!p) return "bx_prover_submit_segment: null prover";
    if (!segment || len == 0) return serr(p, "bx_prover_submit_segment: empty segment");
    if (len > ((size_t)1 << 32) - 4) return serr(p, "bx_prover_submit_segment: segment larger than 4 GiB");
    bx_ctx* c = p->c;
    std::lock_guard<std::mutex> g(p->seg_mu);
    if (p->seg_count == 2) return serr(p, "bx_prover_submit_segment: staging slots busy (two segments are already outstanding)");
    if (hipSetDevice(c->device) != hipSuccess) return serr(p, "bx_prover_submit_segment: hipSetDevice failed");
    SegSlot& sl = p->seg[(p->seg_head + p->seg_count) & 1];
    const size_t padded = (len + 3) & ~(size_t)3;
    if (sl.host_cap < padded || sl.dev_cap < padded) {  // grown on demand, kept for the prover's lifetime (a free slot has no copy in flight)
        if (sl.host) (void)hipHostFree(sl.host);
        if (sl.dev) (void)hipFree(sl.dev);
        sl.host = nullptr, sl.dev = nullptr, sl.host_cap = sl.dev_cap = 0;
        const size_t cap = padded + padded / 8;
        if (hipHostMalloc((void**)&sl.host, cap, hipHostMallocDefault) != hipSuccess) return serr(p, "bx_prover_submit_segment: out of pinned host memory");
        sl.host_cap = cap;
        if (hipMalloc((void**)&sl.dev, cap) != hipSuccess) return serr(p, "bx_prover_submit_segment: out of device memory");
        sl.dev_cap = cap;
    }
    // copy and upload in 8 MB pieces: the DMA of piece i runs while piece i+1 is copied into the slot (host copy ~34 GB/s, PCIe ~57 GB/s:
    // an 80 MB segment is in HBM ~0.2 ms after its last byte was copied instead of 1.4 ms)
    sl.len = len;
    if (hipEventRecord(sl.up0, p->copy_stream) != hipSuccess) return serr(p, "bx_prover_submit_segment: upload failed");
    constexpr size_t PIECE = (size_t)8 << 20;
    for (size_t off = 0; off < padded; off += PIECE) {
        const size_t n = padded - off < PIECE ? padded - off : PIECE, have = off + n <= len ? n : (len > off ? len - off : 0);
        if (have) memcpy(sl.host + off, segment + off, have);
        if (have < n) memset(sl.host + off + have, 0, n - have);
        if (hipMemcpyAsync((uint8_t*)sl.dev + off, sl.host + off, n, hipMemcpyHostToDevice, p->copy_stream) != hipSuccess)
            return serr(p, "bx_prover_submit_segment: upload failed");
    }
    if (hipEventRecord(sl.up, p->copy_stream) != hipSuccess) return serr(p, "bx_prover_submit_segment: upload failed");
    p->seg_count += 1;
    return nullptr;
} BX_ABI_CATCH((p ? p->c : nullptr), "bx_prover_submit_segment")

extern "C" const char* bx_prove_submitted(bx_prover* p, uint32_t* seal_out, size_t seal_cap, size_t* seal_words) try {
    if (!p) return "bx_prove_submitted: null prover";
    SegSlot* sl = nullptr;
    {
        std::lock_guard<std::mutex> g(p->seg_mu);
        if (p->seg_count == 0) return perr(p, "bx_prove_submitted: no segment was submitted");
        sl = &p->seg[p->seg_head];
    }
    const char* r = nullptr;
    // the code group's commitment is enqueued first (bx_prove_segment_bytes did that before it even staged the bytes); everything from
    // witgen on waits for the upload's event — on the stream, not on the host
    if (hipSetDevice(p->c->device) != hipSuccess) r = perr(p, "bx_prove_segment: hipSetDevice failed");
    else if (p->c->gq_n && (r = gather_flush(p->c)) != nullptr) {
    } else if (!p->prologue_done && (r = prove_prologue(p)) != nullptr) {
    } else if (hipStreamWaitEvent(p->c->stream, sl->up, 0) != hipSuccess) r = perr(p, "bx_prove_segment: could not order the proof behind the segment's upload");
    else r = prove_segment_impl(p, *sl, seal_out, seal_cap, seal_words);
    p->prologue_done = false;
    // a proof ends with a blocking read-back of the whole stream, so the upload is over: its duration is on the two events
    if (!r && hipEventElapsedTime(&p->last_upload_ms, sl->up0, sl->up) == hipSuccess) p->last_upload_bytes = sl->len;
    if (r) (void)hipEventSynchronize(sl->up);  // failed before the stream got there: the slot must be idle before it is reused
    std::lock_guard<std::mutex> g(p->seg_mu);
    p->seg_head ^= 1;
    p->seg_count -= 1;
    return r;
} BX_ABI_CATCH((p ? p->c : nullptr), "bx_prove_submitted")

extern "C" const char* bx_prove_segment_bytes(bx_prover* p, const uint8_t* segment, size_t len, uint32_t* seal_out, size_t seal_cap, size_t* seal_words) try {
    if (!p) return "bx_prove_segment_bytes: null prover";
    {
        std::lock_guard<std::mutex> g(p->seg_mu);
        if (p->seg_count != 0) return perr(p, "bx_prove_segment_bytes: segments submitted earlier are still outstanding (use bx_prove_submitted)");
    }
    // start what does not need the bytes, then stage and upload them while the device works on it
    if (hipSetDevice(p->c->device) != hipSuccess) return perr(p, "bx_prove_segment: hipSetDevice failed");
    if (const char* e = prove_prologue(p)) return e;
    p->prologue_done = true;
    if (const char* e = bx_prover_submit_segment(p, segment, len)) {
        p->prologue_done = false;  // the code commitment just enqueued is simply redone by the next proof
        return e;
    }
    return bx_prove_submitted(p, seal_out, seal_cap, seal_words);
} BX_ABI_CATCH((p ? p->c : nullptr), "bx_prove_segment_bytes")

extern "C" const char* bx_prover_last_upload(const bx_prover* p, double* ms, size_t* bytes) {
    if (!p) return "bx_prover_last_upload: null prover";
    if (ms) *ms = p->last_upload_ms;
    if (bytes) *bytes = p->last_upload_bytes;
    return nullptr;
}

extern "C" const char* bx_prover_set_noise_seed(bx_prover* p, uint64_t noise_seed) {
    if (!p) return "bx_prover_set_noise_seed: null prover";
    if (p->circ->set_noise_seed) p->circ->set_noise_seed(p->circ->user, p->circ_state, noise_seed);
    return nullptr;
}
extern "C" const char* bx_prove_segment(bx_prover* p, uint64_t seed, uint32_t* seal_out, size_t seal_cap, size_t* seal_words) try {
    if (!p) return "bx_prove_segment: null prover";
    uint8_t wire[BX_SEGMENT_WIRE_BYTES];
    bx_segment_encode(0, p->shape.po2, seed, wire);  // the circuit derives the noise seed from the seed
    return bx_prove_segment_bytes(p, wire, sizeof wire, seal_out, seal_cap, seal_words);
} BX_ABI_CATCH((p ? p->c : nullptr), "bx_prove_segment")
extern "C" const char* bx_prove_segment_zk(bx_prover* p, uint64_t seed, uint64_t noise_seed, uint32_t* seal_out, size_t seal_cap, size_t* seal_words) try {
    if (!p) return "bx_prove_segment_zk: null prover";
    (void)bx_prover_set_noise_seed(p, noise_seed);
    return bx_prove_segment(p, seed, seal_out, seal_cap, seal_words);
} BX_ABI_CATCH((p ? p->c : nullptr), "bx_prove_segment_zk")

// The circuit's control ID for this shape: the code group through the same commit as in a proof, root read back.
extern "C" const char* bx_prover_control_id(bx_prover* p, uint32_t id_out[8]) try {
    if (!p) return "bx_prover_control_id: null prover";
    if (!id_out) return perr(p, "bx_prover_control_id: null output");
    bx_ctx* c = p->c;
    if (hipSetDevice(c->device) != hipSuccess) return perr(p, "bx_prover_control_id: hipSetDevice failed");
    Group& G = p->groups[0];
    PV(prove_prologue(p));
    size_t used = 0;
    const uint32_t* host = nullptr;
    PV(tree_fetch(p, G.tree, &used, &host));
    PV(d2h_batch_wait(c));
    memcpy(id_out, host, 32);
    return nullptr;
} BX_ABI_CATCH((p ? p->c : nullptr), "bx_prover_control_id")

static const char* prove_segment_impl(bx_prover* p, const SegSlot& seg, uint32_t* seal_out, size_t seal_cap, size_t* seal_words) {
    bx_ctx* c = p->c;
    if (hipSetDevice(c->device) != hipSuccess) return perr(p, "bx_prove_segment: hipSetDevice failed");
    const size_t N = p->N, D = 4 * N;
    const uint32_t po2 = p->shape.po2;
    Transcript T(&p->h2);
    T.seal.reserve(p->seal_bound);
    TraceRange whole(c, "bx:prove_segment");

    // ---- header ----
    {
        uint32_t hdr[BX_SEAL_HEADER_WORDS] = {po2, p->shape.w_code, p->shape.w_data, p->shape.w_accum, p->shape.cons_terms, p->shape.cons_degree};
        uint32_t enc[BX_SEAL_HEADER_WORDS], dg[8];
        for (int i = 0; i < BX_SEAL_HEADER_WORDS; ++i) enc[i] = fp_encode(hdr[i]);
        T.write(hdr, BX_SEAL_HEADER_WORDS);
        p->h2.hash_elems(dg, enc, BX_SEAL_HEADER_WORDS);
        T.commit(dg);
    }
    // ---- witness generation (code + data), then the commits in transcript order; the accumulate step needs the
    //      challenge drawn after the data commit, like upstream's accum mix ----
    const bx_circuit_ops* circ = p->circ;
    Fp4 beta = f4_zero();
    uint32_t globals[BX_MAX_GLOBALS];
    memset(globals, 0, sizeof globals);
    {
        TraceRange tr(c, "bx:witgen");  // the code group and its commitment are already enqueued (prove_prologue)
        PV(circ->witgen(circ->user, p->circ_state, c, p->code_w.b, p->groups[1].coeffs.b, seg.host, seg.len,
                        bx_buf{seg.dev, (seg.len + 3) / 4}, globals));
    }
    if (p->n_globals) {  // the statement's public words: in the seal and in the transcript before any commitment
        uint32_t dg[8];
        for (uint32_t i = 0; i < p->n_globals; ++i)
            if (globals[i] >= P) return perr(p, "bx_prove_segment: the circuit produced a non-canonical public word");
        T.write(globals, p->n_globals);
        p->h2.hash_elems(dg, globals, p->n_globals);
        T.commit(dg);
    }
    {   // code and data: neither commit needs anything from the transcript, so both are enqueued and ONE round trip brings back both
        // roots and top layers; the transcript absorbs them in order (code, then data) and only then is beta drawn
        size_t used = 0;
        const uint32_t* host[2] = {nullptr, nullptr};
        TraceRange tr(c, "bx:commit_data");
        PV(commit_group_work(p, p->groups[1]));
        PV(tree_fetch(p, p->groups[0].tree, &used, &host[0]));
        PV(tree_fetch(p, p->groups[1].tree, &used, &host[1]));
        PV(d2h_batch_wait(c));
        for (int g = 0; g < 2; ++g) {
            tree_absorb(p->groups[g].tree, host[g], T);
            memcpy(p->last_roots + 8 * g, p->groups[g].tree.root, 32);
        }
    }
    {
        Group& G = p->groups[2];
        beta = T.random_ext();
        {
            TraceRange tr(c, "bx:accumulate");
            PV(circ->accumulate(circ->user, p->circ_state, c, G.coeffs.b, beta.c));  // CircuitHal::accumulate
        }
        TraceRange tr(c, "bx:commit_accum");
        PV(commit_group(p, G, T));
        memcpy(p->last_roots + 16, G.tree.root, 32);
    }
    // ---- eval_check: the constraint polynomial over the 4N domain, divided by the vanishing polynomial ----
    Group& CK = p->groups[3];
    {
        Fp4 poly_mix = T.random_ext();
        // the 16N-word check buffer holds the 4 ext planes over the 4N domain                       (CircuitHal::eval_check)
        {
            TraceRange tr(c, "bx:eval_check");
            PV(circ->eval_check(circ->user, p->circ_state, c, CK.coeffs.b, p->groups[0].evaluated.b, p->groups[1].evaluated.b,
                                p->groups[2].evaluated.b, poly_mix.c, beta.c, globals));
        }
        TraceRange tr(c, "bx:commit_check");
        PV(bx_batch_interpolate_ntt(c, CK.coeffs.b, 4));        // 4 polynomials of size 4N
        PV(bx_zk_shift(c, CK.coeffs.b, BX_CHECK_SIZE));         // viewed as 16 polynomials of size N
        PV(bx_batch_expand_into_evaluate_ntt(c, CK.evaluated.b, CK.coeffs.b, BX_CHECK_SIZE, 2));
        PV(bx_batch_bit_reverse(c, CK.coeffs.b, BX_CHECK_SIZE));
        PV(tree_commit(p, CK.tree, CK.evaluated.b, T));
        memcpy(p->last_roots + 24, CK.tree.root, 32);
    }
    // ---- DEEP: evaluate every tap at Z * back_one^back, write/commit coeff_u ----
    TraceStages stage(c);
    stage.next("bx:deep_taps");
    const Fp4 Z = T.random_ext();
    const uint32_t back_one = fp_inv(fp_pow(fp_encode(137u), (uint64_t)1 << (27 - po2)));  // ROU_REV[po2] = w_N^-1
    // check columns hold g(3z) of the split check(y) = sum_q y^q g_q(y^4), so their tap is z = Z^4 / 3: the verifier can
    // then test check(Z) against the trace taps (verify.cpp)
    const Fp4 Z4 = f4_scale(host_pow(Z, 4), fp_inv(MONT_THREE));
    const size_t n_trace_combos = p->combo_backs.size(), n_combos = n_trace_combos + 1;
    // per combo: the points Z * w_N^-b of its tap set and the matrix that turns the values there into the coefficients of the
    // interpolating polynomial (PolyGroup / Prover::finalize: `poly_interpolate` per register; the points are shared by all
    // registers of a combo, so the Lagrange basis is expanded once per combo)
    std::vector<std::vector<Fp4>> pts(n_trace_combos), interp(n_trace_combos);
    for (size_t id = 0; id < n_trace_combos; ++id) {
        const auto& B = p->combo_backs[id];
        const size_t k = B.size();
        for (uint32_t b : B) pts[id].push_back(f4_scale(Z, fp_pow(back_one, b)));
        interp[id].assign(k * k, f4_zero());  // interp[t * k + i] = coefficient t of the basis polynomial L_i
        for (size_t i = 0; i < k; ++i) {
            std::vector<Fp4> poly(1, f4_one());  // prod_{j != i} (x - x_j), low coefficient first
            Fp4 denom = f4_one();
            for (size_t j = 0; j < k; ++j) {
                if (j == i) continue;
                std::vector<Fp4> next(poly.size() + 1, f4_zero());
                for (size_t d = 0; d < poly.size(); ++d) {
                    next[d + 1] = f4_add(next[d + 1], poly[d]);
                    next[d] = f4_sub(next[d], f4_mul(poly[d], pts[id][j]));
                }
                poly.swap(next);
                denom = f4_mul(denom, f4_sub(pts[id][i], pts[id][j]));
            }
            const Fp4 inv = f4_inv(denom);
            for (size_t t = 0; t < k; ++t) interp[id][t * k + i] = f4_mul(poly[t], inv);
        }
    }
    std::vector<uint32_t> coeff_u;  // flattened ext elems, column by column, group by group
    {
        std::vector<uint32_t> xs;
        for (int g = 0; g < 4; ++g) {
            Group& G = p->groups[g];
            for (uint32_t col = 0; col < G.width; ++col)
                for (size_t t = 0; t < G.backs[col].size(); ++t) {
                    const Fp4& x = g == 3 ? Z4 : pts[G.combo[col]][t];
                    xs.insert(xs.end(), x.c, x.c + 4);
                }
        }
        const size_t ne_all = p->tap_which.size();
        PV(h2d_staged(c, p->xs.slice(0, 4 * ne_all), xs.data(), 4 * ne_all));  // no wait: the evaluations' read-back below is the round trip
        if (p->tap_ptrs.b.dptr) {
            PV(bx_batch_evaluate_ptrs(c, p->tap_ptrs.b, p->tap_flags.b, N, p->xs.slice(0, 4 * ne_all), p->evals.slice(0, 4 * ne_all)));
        } else {
            for (int g = 0; g < 4; ++g) {
                Group& G = p->groups[g];
                const size_t o = p->tap_first[g], ne = p->tap_first[g + 1] - p->tap_first[g];
                if (p->coeffs_bitrev && g < 3)
                    PV(bx_batch_evaluate_any_bitrev(c, G.coeffs.b, G.width, p->which.slice(o, ne), p->xs.slice(4 * o, 4 * ne), p->evals.slice(4 * o, 4 * ne)));
                else
                    PV(bx_batch_evaluate_any(c, G.coeffs.b, G.width, p->which.slice(o, ne), p->xs.slice(4 * o, 4 * ne), p->evals.slice(4 * o, 4 * ne)));
            }
        }
        std::vector<uint32_t> ev(4 * ne_all);
        PV(bx_d2h(c, ev.data(), p->evals.slice(0, 4 * ne_all), 4 * ne_all));
        size_t e = 0;
        for (int g = 0; g < 4; ++g) {
            Group& G = p->groups[g];
            for (uint32_t col = 0; col < G.width; ++col) {
                const size_t k = G.backs[col].size();
                if (k == 1) {  // a single tap: the "polynomial" is the value itself
                    coeff_u.insert(coeff_u.end(), ev.begin() + 4 * e, ev.begin() + 4 * e + 4);
                } else {
                    const auto& M = interp[G.combo[col]];
                    for (size_t t = 0; t < k; ++t) {
                        Fp4 ct = f4_zero();
                        for (size_t i = 0; i < k; ++i) {
                            const uint32_t* y = &ev[4 * (e + i)];
                            ct = f4_add(ct, f4_mul(M[t * k + i], Fp4{{y[0], y[1], y[2], y[3]}}));
                        }
                        coeff_u.insert(coeff_u.end(), ct.c, ct.c + 4);
                    }
                }
                e += k;
            }
        }
    }
    {
        uint32_t dg[8];
        T.write(coeff_u.data(), coeff_u.size());
        p->h2.hash_elems(dg, coeff_u.data(), coeff_u.size());
        T.commit(dg);
    }
    // ---- DEEP: mix every column into its combo, subtract the mixed u polynomials, divide by every tap point ----
    stage.next("bx:deep_combos");
    const Fp4 mix = T.random_ext();
    if (hipMemsetAsync(p->combos.b.dptr, 0, p->combos.b.len * 4, c->stream) != hipSuccess) return perr(p, "bx_prove_segment: memset failed");
    {
        Fp4 cur = f4_one();
        std::vector<Fp4> combo_u(n_combos * BX_MAX_TAPS, f4_zero());
        size_t u = 0;
        for (int g = 0; g < 4; ++g) {
            Group& G = p->groups[g];
            PV(bx_mix_poly_coeffs(c, p->combos.b, cur.c, mix.c, G.coeffs.b, G.combo_ids.b, G.width, N));
            for (uint32_t col = 0; col < G.width; ++col) {
                const uint32_t id = G.combo[col];
                for (size_t t = 0; t < G.backs[col].size(); ++t, u += 4) {
                    Fp4 cu{{coeff_u[u], coeff_u[u + 1], coeff_u[u + 2], coeff_u[u + 3]}};
                    combo_u[id * BX_MAX_TAPS + t] = f4_add(combo_u[id * BX_MAX_TAPS + t], f4_mul(cur, cu));
                }
                cur = f4_mul(cur, mix);
            }
        }
        // the trace combos collect bit-reversed coefficient storage, the last combo only the check group (natural order)
        if (p->coeffs_bitrev) PV(bx_batch_bit_reverse_ext(c, p->combos.slice(0, 4 * N * n_trace_combos), n_trace_combos));
        {   // subtract the mixed u polynomials (degree < taps of the combo) from the low coefficients of the combos, on the device
            SubLow sl;
            memset(&sl, 0, sizeof sl);
            for (size_t id = 0; id < n_combos; ++id)
                for (size_t t = 0; t < BX_MAX_TAPS; ++t)
                    for (int k = 0; k < 4; ++k) sl.v[(id * BX_MAX_TAPS + t) * 4 + k] = combo_u[id * BX_MAX_TAPS + t].c[k];
            hipLaunchKernelGGL(sub_low_kernel, dim3((unsigned)((n_combos * BX_MAX_TAPS * 4 + 63) / 64)), dim3(64), 0, c->stream,
                               (uint32_t*)p->combos.b.dptr, (uint32_t)(4 * N), sl, (uint32_t)n_combos);
            if (hipGetLastError() != hipSuccess) return perr(p, "bx_prove_segment: sub_low launch failed");
        }
        // divide every combo by each of its tap points: round r divides, in one launch, every combo that has a tap r (the check
        // combo's single point joins round 0) — max-taps launches per proof instead of one five-launch call per division
        size_t d = 0;
        for (size_t r = 0;; ++r) {
            uint32_t which[BX_MAX_COMBOS];
            uint32_t zs[4 * BX_MAX_COMBOS];
            size_t cnt = 0;
            for (size_t id = 0; id < n_trace_combos; ++id)
                if (pts[id].size() > r) {
                    which[cnt] = (uint32_t)id;
                    memcpy(zs + 4 * cnt++, pts[id][r].c, 16);
                }
            if (r == 0) {
                which[cnt] = (uint32_t)n_trace_combos;
                memcpy(zs + 4 * cnt++, Z4.c, 16);
            }
            if (!cnt) break;
            PV(bx_poly_divide_batch_indexed(c, p->combos.b, n_combos, cnt, which, zs, p->rems.slice(4 * d, 4 * cnt)));
            d += cnt;
        }
        // the remainders (all zero for a consistent proof) are looked at with the last read-back of the proof, not here: a
        // non-zero one is an internal error, and waiting for it now would cost a round trip on every proof
    }
    PV(bx_eltwise_sum_extelem(c, p->final_poly.b, p->combos.b));
    PV(bx_batch_bit_reverse(c, p->final_poly.b, 4));

    // ---- fri_prove ----
    stage.next("bx:fri_prove");
    {
        bx_buf coeffs = p->final_poly.b;
        const bool dev = c->dev_draws != 0 && !p->rounds.empty();
        if (dev) {
            // A round's challenge depends on nothing but the round's root, so the device half of the transcript draws it
            // (bx_transcript_step: 1-2 permutations on one quad, ~20 us) and the fold reads it from device memory: no host round trip
            // inside the loop.  Roots, top layers, the drawn challenges and the final coefficients come back in ONE copy after the
            // loop; the host transcript then replays the same commits (it writes the seal) and must draw the same words.
            uint32_t st[25];
            memcpy(st, T.cells, sizeof T.cells);
            st[24] = T.pool_used;
            PV(h2d_staged(c, p->tstate.slice(0, 25), st, 25));
        }
        size_t ri = 0;
        for (FriRound& r : p->rounds) {
            PV(bx_batch_expand_into_evaluate_ntt(c, r.evaluated.b, coeffs, 4, 2));
            if (dev) {
                PV(tree_build(p, r.tree, r.evaluated.b));
                PV(bx_transcript_step(c, p->tstate.b, r.tree.nodes.slice(8, 8), 1, p->dev_chal.slice(4 * ri, 4), 1));
                PV(bx_fri_fold_dev(c, r.out_coeffs.b, coeffs, p->dev_chal.slice(4 * ri, 4)));
            } else {
                PV(tree_commit(p, r.tree, r.evaluated.b, T));
                Fp4 fold_mix = T.random_ext();
                PV(bx_fri_fold(c, r.out_coeffs.b, coeffs, fold_mix.c));
            }
            coeffs = r.out_coeffs.b;
            ++ri;
        }
        PV(bx_eltwise_copy_elem(c, p->final_coeffs.b, coeffs));
        PV(bx_batch_bit_reverse(c, p->final_coeffs.b, 4));
        const size_t nfc = p->final_coeffs.b.len;
        std::vector<uint32_t> fc(nfc);
        uint32_t dg[8];
        if (dev) {
            size_t used = 0;
            std::vector<const uint32_t*> host(p->rounds.size(), nullptr);
            const uint32_t *fc_host = nullptr, *chal = nullptr;
            for (size_t i = 0; i < p->rounds.size(); ++i) PV(tree_fetch(p, p->rounds[i].tree, &used, &host[i]));
            PV(d2h_batch_add(c, &used, p->dev_chal.slice(0, 4 * p->rounds.size()), 4 * p->rounds.size(), &chal));
            PV(d2h_batch_add(c, &used, p->final_coeffs.b, nfc, &fc_host));
            PV(d2h_batch_wait(c));
            for (size_t i = 0; i < p->rounds.size(); ++i) {
                tree_absorb(p->rounds[i].tree, host[i], T);
                const Fp4 fold_mix = T.random_ext();
                if (memcmp(fold_mix.c, chal + 4 * i, 16) != 0) return perr(p, "bx_prove_segment: the device transcript drew a different FRI challenge than the host transcript");
            }
            memcpy(fc.data(), fc_host, nfc * 4);
        } else {
            PV(bx_d2h(c, fc.data(), p->final_coeffs.b, fc.size()));
        }
        T.write(fc.data(), fc.size());
        p->h2.hash_elems(dg, fc.data(), fc.size());
        T.commit(dg);
    }
    // ---- queries: positions come only from the RNG (writes do not feed it), so draw all 50 first, gather each tree
    //      for the whole batch on the device, then lay the seal out in upstream's query-major order ----
    stage.next("bx:queries");
    {
        const unsigned bits = (unsigned)ilog2(D);
        uint32_t pos0[BX_QUERIES];
        for (int q = 0; q < BX_QUERIES; ++q) pos0[q] = T.random_bits(bits) % (uint32_t)D;
        size_t n_trees = 4 + p->rounds.size();
        std::vector<size_t> qw(n_trees), off(n_trees + 1, 0);
        std::vector<uint32_t> all_pos(n_trees * BX_QUERIES);
        uint32_t pos[BX_QUERIES];
        memcpy(pos, pos0, sizeof pos);
        for (size_t t = 0; t < n_trees; ++t) {
            Tree& tr = t < 4 ? p->groups[t].tree : p->rounds[t - 4].tree;
            if (t >= 4)
                for (int q = 0; q < BX_QUERIES; ++q) pos[q] %= (uint32_t)tr.rows;  // group = pos % (domain / FRI_FOLD)
            memcpy(all_pos.data() + t * BX_QUERIES, pos, sizeof pos);
            qw[t] = tr.query_words();
            off[t + 1] = off[t] + qw[t] * BX_QUERIES;
        }
        PV(h2d_staged(c, p->positions.b, all_pos.data(), all_pos.size()));
        for (size_t t = 0; t < n_trees; ++t) {
            Tree& tr = t < 4 ? p->groups[t].tree : p->rounds[t - 4].tree;
            bx_buf matrix = t < 4 ? p->groups[t].evaluated.b : p->rounds[t - 4].evaluated.b;
            PV(bx_merkle_query_gather(c, p->qout.slice(off[t], qw[t] * BX_QUERIES), matrix, tr.nodes.b, tr.rows, tr.cols,
                                      p->positions.slice(t * BX_QUERIES, BX_QUERIES), BX_QUERIES, tr.top_size()));
        }
        size_t used = 0;
        const uint32_t *host_all = nullptr, *rems = nullptr;
        std::vector<uint32_t> big;
        if (off[n_trees] + 4 * p->n_div + 8 <= bx_ctx::STAGE_WORDS) {
            PV(d2h_batch_add(c, &used, p->qout.slice(0, off[n_trees]), off[n_trees], &host_all));
            PV(d2h_batch_add(c, &used, p->rems.b, 4 * p->n_div, &rems));
            PV(d2h_batch_wait(c));
        } else {  // seals of the largest shapes do not fit the pinned landing area: two plain copies
            big.resize(off[n_trees] + 4 * p->n_div);
            PV(bx_d2h(c, big.data(), p->qout.slice(0, off[n_trees]), off[n_trees]));
            PV(bx_d2h(c, big.data() + off[n_trees], p->rems.b, 4 * p->n_div));
            host_all = big.data();
            rems = big.data() + off[n_trees];
        }
        for (size_t i = 0; i < 4 * p->n_div; ++i)
            if (rems[i] != 0) return perr(p, "bx_prove_segment: DEEP quotient has a non-zero remainder");
        for (int q = 0; q < BX_QUERIES; ++q)
            for (size_t t = 0; t < n_trees; ++t) T.write(host_all + off[t] + (size_t)q * qw[t], qw[t]);
    }
    stage.close();
    if (seal_words) *seal_words = T.seal.size();
    if (T.seal.size() > seal_cap) return perr(p, "bx_prove_segment: seal buffer too small");
    memcpy(seal_out, T.seal.data(), T.seal.size() * 4);
    return nullptr;
}
