// ctx.hpp — internal state behind the opaque bx_ctx of include/bx_hal.h (one per device).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include <map>
#include <unordered_map>
#include <string>
#include <vector>

#include "../../include/bx_hal.h"
#include "fp.hpp"

namespace bx {

constexpr int TW_LOG = 13;  // stage-twiddle tables cover sub-transform sizes up to 2^13

struct ProfRec {
    const char* name;
    double bytes;  // algorithmic bytes of this call
    hipEvent_t e0, e1;
};
struct ProfAgg {
    double ms = 0, bytes = 0;
    long calls = 0;
};

struct TwistKey {
    int m, m_hi, inverse;
    bool operator<(const TwistKey& o) const {
        if (m != o.m) return m < o.m;
        if (m_hi != o.m_hi) return m_hi < o.m_hi;
        return inverse < o.inverse;
    }
};
struct ZkTab {
    uint32_t* lo = nullptr;  // 3^(rev_LO(i_lo) << (n-LO))
    uint32_t* hi = nullptr;  // 3^(rev_(n-LO)(i_hi))
    int lo_bits = 0;
};

}  // namespace bx

struct bx_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    hipStream_t own_stream = nullptr;
    char err[512];
    int cu_count = 256;

    // NTT tables (device): stage table index 2^(s-1)+e -> w_{2^s}^{+-e} (Montgomery)
    uint32_t* d_tw_fwd = nullptr;
    uint32_t* d_tw_inv = nullptr;
    std::map<bx::TwistKey, uint32_t*> twist;
    std::map<int, bx::ZkTab> zk;
    std::map<int, uint32_t*> zk_full;  // n -> 3^bitrev_n(i), i < 2^n (fused interpolate + zk_shift)

    // Poseidon2 parameters: host canonical copy + device Montgomery copy [213 rc | 24 diag]
    uint32_t h_rc[BX_POSEIDON2_RC_COUNT];
    uint32_t h_diag[24];
    uint32_t* d_p2 = nullptr;
    int live_provers = 0;  // bx_prover objects created on this ctx and not yet destroyed

    // deferred device-side errors (e.g. a scatter offset out of range): h_flag is pinned, host-coherent memory the kernels
    // write straight into (one word per kind of error, so plain stores suffice); the blocking entry points (bx_d2h, bx_sync)
    // read it after their own stream synchronisation — no copy, no extra launch
    uint32_t* h_flag = nullptr;
    // pinned landing area of bx_d2h: a device-to-host copy into pageable memory is staged (and serialised) by the runtime;
    // copies up to this size land here at pinned-memory latency and are handed to the caller with one memcpy
    uint32_t* h_stage = nullptr;
    static constexpr size_t STAGE_WORDS = (size_t)1 << 20;  // 4 MiB: every read-back of a proof (tops, taps, queries) fits
    // pinned ring for small host-to-device copies that must not cost a wait (the prover's tap points and query positions): the data
    // is copied into the ring and the async copy reads it from there, so the caller's buffer is free at once
    uint32_t* h_up = nullptr;
    size_t up_used = 0;
    static constexpr size_t UP_WORDS = (size_t)1 << 18;  // 1 MiB

    // look-back scan state (scan.hip): two alternating buffers, each launch clears what the other one was last used with
    uint32_t* d_scan[2] = {nullptr, nullptr};
    size_t scan_cap = 0, scan_used[2] = {0, 0};
    int scan_next = 0;
    long scan_lookback = 1;  // poly_divide / prefix_products: single-pass decoupled look-back kernels (0 = the three-phase kernels)

    // Deferred Hal::gather_sample calls (poly.hip).  MerkleTreeProver::prove issues one gather per opened row and one per path digest:
    // ~5 000 launches of a few words each per proof, 3.7 us of host time apiece on an idle GPU (19 of the 70 ms of a proof driven
    // through the plain trait calls, profiles/r06_plain_hal.json).  A small gather is therefore queued as a 32-byte descriptor and
    // the queue is launched as ONE kernel by whatever touches the ctx next (BX_ENTER at the top of every entry point, OpScope,
    // stream_wait, the staged copies, bx_get_stream), which keeps the stream order the caller sees.  A gather that reads or writes
    // memory an already queued one writes flushes first.  Off while profiling / tracing (per-call events), or with gather_defer = 0.
    std::vector<uint32_t> gq;  // 8 words per descriptor: dst lo/hi, src + idx lo/hi, size, stride, 0, 0
    size_t gq_n = 0;
    uint32_t* d_gq = nullptr;
    uintptr_t gq_dst_lo = 0, gq_dst_hi = 0, gq_src_lo = 0, gq_src_hi = 0;  // bounding intervals of what the queue writes / reads
    long gather_defer = 1;
    static constexpr size_t GQ_MAX = 8192;

    // bx_alloc / bx_release pool (hal.hip).  risc0-zkp's prover allocates every buffer inside a proof (hal.alloc_* in commit_group,
    // finalize, fri_prove: ~9 GB in ~45 buffers at 2^20 / 16-256-64) and drops them at its end; raw hipMalloc / hipFree cost
    // milliseconds per GB-sized block and hipFree drains the WHOLE device, every other lane's stream included.  A released block
    // therefore goes to a per-ctx free list and the next request of about that size takes it back: no driver call, no wait —
    // every use of the memory, old and new, is ordered on the ctx's one stream.  Capped by the tunable alloc_cache_mb (0 = off:
    // hipMalloc / stream wait + hipFree as before); an allocation that fails empties the list and retries.
    std::multimap<size_t, void*> pool_free;        // capacity in bytes -> block
    std::unordered_map<void*, size_t> pool_live;   // blocks handed out by bx_alloc, with their capacity
    size_t pool_cached = 0;
    long alloc_cache_mb = 16384;

    // scratch (grown on demand)
    uint32_t* d_scratch = nullptr;
    size_t scratch_words = 0;

    // tunables
    long ntt_block_log = 12;   // log2 of the contiguous-pass sub-transform (13 when the size needs it)
    long ntt_tile_log = 14;    // log2 of the strided-pass LDS tile (elements)
    long ntt_fast = 1;         // 1 = register-radix-16 passes (ntt_r16.hpp), 0 = one-stage-per-barrier v1 kernels
    long ntt_tile_a_log = 12;  // log2 elements per pass-A workgroup (fast path)
    long ntt_tile_b_log = 13;  // log2 elements per pass-B workgroup (fast path)
    long ntt_cols_per_wg = 8;  // forward pass A (2^12 tiles): columns sharing one load of the tile's twist + twiddles
    long ntt_group_cols = 0;   // forward transform: columns per pass-A + pass-B group (0 = all columns per pass)
    long ntt_tile_b_wide = 1;  // grow the pass-B tile (up to 2^14) so that rows are at least 16 words wide
    long hash_rows_block = 256;
    long fold_deep = 2;              // large Merkle layers: up to this many levels per launch, depth first per lane (1 = one launch per layer)
    long dev_draws = 0;  // 1 = the prover draws the FRI challenges (which depend only on a Merkle root) on the device (bx_transcript_step): three blocking
                         // waits fewer per proof, same seal — and measured no faster while the host polls 25-45 % longer (profiles/r03_ab_dev_draws.jsonl), so off
    long fold_deep_min_lanes = 1 << 17;  // ... as long as the launch still has this many lanes (measured: tools/foldbench2.py, profiles/r03_foldbench.jsonl)
    long fold_quad = 1;              // small Merkle layers: four lanes per node (hash_fold_quad_kernel) instead of one
    long fold_quad_wg = 512;         // ... input digests per workgroup of that kernel (a power of two, 16..512)
    long fold_fuse_below = 1 << 17;  // Merkle layers with at most this many inputs are folded 9 levels per launch
    // How a host thread waits for its stream (bx_d2h, bx_sync, bx_h2d ...).  Measured on this ROCm (tools/waitbench.hip,
    // tools/host_budget.py; profiles/r03_waitbench.jsonl, r03_host_budget.json), 3 lanes on one GPU at 2^20:
    //   0 spin   hipStreamSynchronize under the default device schedule busy-polls (the hipEventBlockingSync event flag is
    //            ignored): one CPU per lane thread + one runtime thread = 4.0 CPUs busy, 0.162 CPU-s per proof;
    //   1 block  hipSetDeviceFlags(hipDeviceScheduleBlockingSync) before the stream is created (a stream keeps the mode the
    //            device had at its creation): waits sleep on the completion interrupt, but the runtime's event thread then
    //            handles one interrupt per kernel: 0.52 CPUs busy, 0.021 CPU-s per proof (flag is per device and process; if
    //            the runtime refuses it, falls back to 2).  Only BX_WAIT=block at bx_init sets the flag; asked for at run time
    //            (bx_set_tunable) it IS policy 2 — flipping the mode under live streams deadlocks a later hipFree (hal.hip);
    //   2 poll   record an event and hipEventQuery it every wait_poll_us (usleep in between; the waiting thread's timer slack is
    //            lowered to 1 us so that the period is real): no interrupts, no spinning: 0.13 CPUs busy, 0.0055 CPU-s per
    //            proof, 24.90 against 24.95 proofs/s busy-polling (a 20 us period doubles the CPU for +0.0).
    // With 3 lanes x 8 GPUs against the 16-CPU quota of a GPU box, spinning lanes would take CPUs from each other and from the
    // seal verifiers, so 2 is the default; BX_WAIT=spin|block|poll overrides it at bx_init.
    long wait_blocking = 2;
    long wait_poll_us = 50;
    long wait_spin_us = 60;  // policy 2: poll without sleeping for this long before the first usleep (short read-backs end within it)
    bool wait_poll = false;  // blocking requested but the device flag could not be set: sleep-poll an event instead
    hipEvent_t wait_ev = nullptr;
    long eval_x4 = 1;                // batch_evaluate_any: 16-byte loads for whole 2^15-coefficient segments (0 = the dword kernel)
    long deep_bitrev = 1;            // segment prover: keep trace coefficients bit-reversed through the DEEP phase (read at bx_prover_create)

    // timing
    hipEvent_t t0 = nullptr, t1 = nullptr;
    bool profile = false;
    std::vector<bx::ProfRec> prof_pending;
    std::map<std::string, bx::ProfAgg> prof_agg;
    std::vector<hipEvent_t> event_pool;
};

namespace bx {

inline const char* set_err(bx_ctx* c, const char* what, hipError_t e, const char* file, int line) {
    snprintf(c->err, sizeof c->err, "%s: %s (%s:%d)", what, hipGetErrorString(e), file, line);
    return c->err;
}
inline const char* set_msg(bx_ctx* c, const char* msg) {
    snprintf(c->err, sizeof c->err, "%s", msg);
    return c->err;
}

// "No exception crosses the ABI" (bx_hal.h): every multi-statement extern "C" entry point is a function-try-block closed by
// BX_ABI_CATCH — a std::bad_alloc from a host-side container (or anything else thrown below) comes back as the entry point's
// error string like any other failure.  The message lives in the ctx when there is one, otherwise in a thread-local buffer.
const char* abi_caught(bx_ctx* c, const char* fn) noexcept;
#define BX_ABI_CATCH(c, fn) \
    catch (...) { return bx::abi_caught((c), fn); }

#define BX_HIP(c, call)                                                        \
    do {                                                                       \
        hipError_t _e = (call);                                                \
        if (_e != hipSuccess) return bx::set_err((c), #call, _e, __FILE__, __LINE__); \
    } while (0)
#define BX_LAUNCH_CHECK(c) BX_HIP(c, hipGetLastError())
// top of every entry point that touches the device: select it, and launch the gathers queued by earlier bx_gather_sample calls
#define BX_ENTER(c)                                          \
    do {                                                     \
        BX_HIP(c, hipSetDevice((c)->device));                \
        if ((c)->gq_n) BX_TRY(bx::gather_flush(c));          \
    } while (0)
#define BX_REQUIRE(c, cond, msg)                  \
    do {                                          \
        if (!(cond)) return bx::set_msg((c), msg); \
    } while (0)
#define BX_TRY(expr)                      \
    do {                                  \
        const char* _m = (expr);          \
        if (_m) return _m;                \
    } while (0)

inline bool is_pow2(size_t n) { return n && !(n & (n - 1)); }
// a * b <= limit without wrapping: the scalars of an entry point come from the caller, and a product that wraps must not pass a bound
// check (a wrapped `4 * count` is also a division by zero)
inline bool mul_le(size_t a, size_t b, size_t limit) {
    size_t p;
    return !__builtin_mul_overflow(a, b, &p) && p <= limit;
}
inline int ilog2(size_t n) {
    int k = 0;
    while (((size_t)1 << k) < n) k++;
    return k;
}

// roctx ranges for rocprofv3 --marker-trace (bx_trace_enable, bx_hal.h): process-wide, off by default.  The roctx library
// is looked up with dlopen when tracing is switched on, so the HAL has no link-time dependency on the profiler.
int trace_level();  // 0 off, 1 ranges, 2 ranges + a stream sync at the end of every prover stage
void trace_push(const char* name);
void trace_pop();
// A named stage of bx_prove_segment (upstream brackets the same stages with nvtx ranges / tracing spans).  At level 2 the
// ctx's stream is drained before the range closes, so the host-side range is the stage's device time.
struct TraceRange {
    bx_ctx* c;
    bool on;
    TraceRange(bx_ctx* ctx, const char* name) : c(ctx), on(trace_level() > 0) {
        if (on) trace_push(name);
    }
    ~TraceRange();
    TraceRange(const TraceRange&) = delete;
    TraceRange& operator=(const TraceRange&) = delete;
};

// The same for a run of consecutive stages in straight-line code: next() closes the open range and opens another.
struct TraceStages {
    bx_ctx* c;
    bool open = false;
    explicit TraceStages(bx_ctx* ctx) : c(ctx) {}
    void close();
    void next(const char* name) {
        close();
        if (trace_level() > 0) {
            trace_push(name);
            open = true;
        }
    }
    ~TraceStages() { close(); }
    TraceStages(const TraceStages&) = delete;
    TraceStages& operator=(const TraceStages&) = delete;
};

// RAII bracket: records hipEvents around an entry point when profiling is on, and a roctx range when tracing is.
struct OpScope {
    bx_ctx* c;
    const char* name;
    double bytes;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    bool traced = false;
    OpScope(bx_ctx* ctx, const char* n, double b);
    ~OpScope();
};

// internal launchers shared between translation units (each returns NULL or an error string)
const char* ensure_scratch(bx_ctx* c, size_t words);
const char* raw_alloc(bx_ctx* c, size_t words, bx_buf* out);  // hal.hip: plain hipMalloc for the library's own long-lived buffers (not pooled)
const char* gather_flush(bx_ctx* c);  // poly.hip: launch the queued gather_sample descriptors (no-op when the queue is empty)
constexpr uint32_t FLAG_SLOT_SCATTER_RANGE = 0u;  // words of bx_ctx::h_flag
constexpr uint32_t FLAG_SLOT_SCATTER_INDEX = 1u;
constexpr uint32_t FLAG_SLOTS = 4u;
void apply_wait_policy(bx_ctx* c, bool at_init);  // bx_ctx::wait_blocking -> the device's schedule flag (at bx_init only) or the event poll
hipError_t stream_wait(bx_ctx* c);           // wait for the ctx's stream under the ctx's wait policy (bx_ctx::wait_blocking)
const char* sync_and_check_flag(bx_ctx* c);  // stream_wait + deferred device errors
// h2d without a wait (words <= UP_WORDS): through the pinned ring; when the ring wraps, the stream is drained first
const char* h2d_staged(bx_ctx* c, bx_buf dst, const uint32_t* src, size_t words);
// several device-to-host copies, ONE wait: d2h_batch_add enqueues a copy into the pinned landing area (bx_ctx::h_stage) and returns
// where it will land; the data is there after d2h_batch_wait.  `used` starts at 0; nothing else may use bx_d2h in between.
const char* d2h_batch_add(bx_ctx* c, size_t* used, bx_buf src, size_t words, const uint32_t** host);
const char* d2h_batch_wait(bx_ctx* c);
const char* poly_divide_lookback(bx_ctx* c, uint32_t* polys, size_t size, size_t count, const uint32_t* zs, uint32_t* rems, const uint32_t* which);  // scan.hip
const char* prefix_products_lookback(bx_ctx* c, uint32_t* io, size_t n, size_t count);
const char* ntt_init_tables(bx_ctx* c);
void ntt_free_tables(bx_ctx* c);
const char* poseidon2_upload_params(bx_ctx* c);

// the synthetic circuit's device stages (circuit.hip), driven by prover.hip
struct Circuit;
const char* circuit_perm_tables(bx_ctx* c, const Circuit& cc, bx_buf offsets, bx_buf index);
const char* circuit_code(bx_ctx* c, const Circuit& cc, bx_buf code);
const char* circuit_witness(bx_ctx* c, const Circuit& cc, bx_buf code, bx_buf data, uint64_t seed_data, uint64_t seed_noise, bx_buf perm_offsets,
                            bx_buf perm_index);
const char* circuit_accum_gather(bx_ctx* c, const Circuit& cc, bx_buf srcvals, bx_buf data);
const char* circuit_accumulate(bx_ctx* c, const Circuit& cc, bx_buf accum, bx_buf run, bx_buf srcvals, bx_buf betas_dev, uint64_t seed_accum);
const char* circuit_mix_table(bx_ctx* c, const Circuit& cc, bx_buf mixpows, const uint32_t poly_mix[4]);
const char* circuit_eval_check(bx_ctx* c, const Circuit& cc, bx_buf check, bx_buf ecode, bx_buf edata, bx_buf eacc, bx_buf mixpows, bx_buf betas_dev,
                               const uint32_t zinv[4], const uint32_t* globals);

}  // namespace bx
