// hal.hip — context, memory, timing and profiling entry points of include/bx_hal.h.
//
// Counterpart of the device/buffer half of risc0_zkp::hal::Hal (alloc_*, copy_from_*, Buffer::view) that the
// agent's prover object owns for the process lifetime (bento/crates/workflow/src/lib.rs:192,246-249: one
// `Rc<dyn ProverServer>` per agent process = one per GPU, compose.yml:113).
#include <dlfcn.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <exception>
#include <new>
#include <mutex>
#include <string>

#include <sys/prctl.h>
#include <time.h>
#include <unistd.h>

#include "ctx.hpp"
#include "poseidon2_params.hpp"

namespace bx {

static hipEvent_t get_event(bx_ctx* c) {
    if (!c->event_pool.empty()) {
        hipEvent_t e = c->event_pool.back();
        c->event_pool.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    return e;
}

// ---- roctx ranges ----
namespace {
struct TraceApi {
    std::atomic<int> level{0};
    std::mutex mu;
    void* lib = nullptr;
    int (*push)(const char*) = nullptr;
    int (*pop)() = nullptr;
} g_trace;
}  // namespace
int trace_level() { return g_trace.level.load(std::memory_order_acquire); }
void trace_push(const char* name) { (void)g_trace.push(name); }
void trace_pop() { (void)g_trace.pop(); }
TraceRange::~TraceRange() {
    if (!on) return;
    if (trace_level() >= 2 && c) (void)stream_wait(c);
    trace_pop();
}
void TraceStages::close() {
    if (!open) return;
    if (trace_level() >= 2 && c) (void)stream_wait(c);
    trace_pop();
    open = false;
}
static const char* trace_set(int level) {
    if (level < 0 || level > 2) return "bx_trace_enable: level must be 0, 1 or 2";
    std::lock_guard<std::mutex> g(g_trace.mu);
    if (level > 0 && !g_trace.push) {
        // rocprofv3 intercepts the rocprofiler-sdk flavour; libroctx64 is roctracer's (rocprof v1/v2)
        static const char* names[] = {"librocprofiler-sdk-roctx.so", "librocprofiler-sdk-roctx.so.1", "libroctx64.so", "libroctx64.so.4"};
        for (const char* n : names) {
            void* h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
            if (!h) continue;
            auto pu = (int (*)(const char*))dlsym(h, "roctxRangePushA");
            auto po = (int (*)())dlsym(h, "roctxRangePop");
            if (pu && po) {
                g_trace.lib = h;
                g_trace.pop = po;
                g_trace.push = pu;
                break;
            }
            dlclose(h);
        }
        if (!g_trace.push) return "bx_trace_enable: no roctx library (librocprofiler-sdk-roctx.so / libroctx64.so) could be loaded";
    }
    g_trace.level.store(level, std::memory_order_release);
    return nullptr;
}

OpScope::OpScope(bx_ctx* ctx, const char* n, double b) : c(ctx), name(n), bytes(b) {
    if (c->gq_n) (void)gather_flush(c);  // queued gathers go first (a failure is sticky: the launch check of this op reports it)
    if (trace_level() > 0) {
        traced = true;
        trace_push(name);
    }
    if (!c->profile) return;
    e0 = get_event(c);
    e1 = get_event(c);
    if (e0) (void)hipEventRecord(e0, c->stream);
}
OpScope::~OpScope() {
    if (traced) trace_pop();
    if (!c->profile || !e0 || !e1) return;
    (void)hipEventRecord(e1, c->stream);
    c->prof_pending.push_back(ProfRec{name, bytes, e0, e1});
}

static void drain_profile(bx_ctx* c) {
    if (c->prof_pending.empty()) return;
    (void)stream_wait(c);
    for (auto& r : c->prof_pending) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, r.e0, r.e1) == hipSuccess) {
            ProfAgg& a = c->prof_agg[r.name];
            a.ms += ms;
            a.bytes += r.bytes;
            a.calls += 1;
        }
        c->event_pool.push_back(r.e0);
        c->event_pool.push_back(r.e1);
    }
    c->prof_pending.clear();
}

}  // namespace bx

using namespace bx;

namespace bx {
const char* abi_caught(bx_ctx* c, const char* fn) noexcept {
    const char* what = "unexpected C++ exception";
    char copy[160];
    try {
        throw;
    } catch (const std::bad_alloc&) {
        what = "out of host memory";
    } catch (const std::exception& e) {
        snprintf(copy, sizeof copy, "%s", e.what());
        what = copy;
    } catch (...) {
    }
    static thread_local char tl[224];
    char* dst = c ? c->err : tl;
    snprintf(dst, c ? sizeof c->err : sizeof tl, "%s: %s", fn, what);
    return dst;
}
}  // namespace bx

extern "C" const char* bx_trace_enable(int level) { return trace_set(level); }
extern "C" int bx_trace_level(void) { return trace_level(); }

extern "C" const char* bx_init(int device, bx_ctx** out) try {
    if (!out) return "bx_init: null out pointer";
    *out = nullptr;
    if (const char* env = getenv("BX_TRACE")) {  // BX_TRACE=1|2: ranges on from the first context on
        static std::once_flag once;
        const char* terr = nullptr;
        std::call_once(once, [&] { terr = trace_set(atoi(env)); });
        if (terr) return terr;
    }
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return "bx_init: no HIP device visible (the HIP HAL has no CPU fallback)";
    if (device < 0 || device >= n) return "bx_init: device index out of range";
    if (hipSetDevice(device) != hipSuccess) return "bx_init: hipSetDevice failed";
    bx_ctx* c = new (std::nothrow) bx_ctx();
    if (!c) return "bx_init: out of host memory";
    c->device = device;
    c->err[0] = 0;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess) c->cu_count = prop.multiProcessorCount;
    if (const char* w = getenv("BX_WAIT")) {  // BX_WAIT=spin|block: how host threads wait for their stream (ctx.hpp)
        if (!strcmp(w, "spin")) c->wait_blocking = 0;
        else if (!strcmp(w, "block")) c->wait_blocking = 1;
        else if (!strcmp(w, "poll")) c->wait_blocking = 2;
        else {
            delete c;
            return "bx_init: BX_WAIT must be 'spin', 'block' or 'poll'";
        }
    }
    // before the stream exists: a stream keeps the wait mode the device had when it was created (measured: the first ctx of a
    // process, whose stream predated the flag, kept busy-polling while later ones slept)
    apply_wait_policy(c, true);
    if (hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking) != hipSuccess) {
        delete c;
        return "bx_init: hipStreamCreate failed";
    }
    c->stream = c->own_stream;
    if (hipEventCreate(&c->t0) != hipSuccess || hipEventCreate(&c->t1) != hipSuccess ||
        hipEventCreateWithFlags(&c->wait_ev, hipEventDisableTiming) != hipSuccess) {
        delete c;
        return "bx_init: hipEventCreate failed";
    }
    if (hipHostMalloc((void**)&c->h_flag, FLAG_SLOTS * 4, hipHostMallocDefault) != hipSuccess) {
        bx_free(c);
        return "bx_init: allocating the deferred error flags failed";
    }
    for (uint32_t i = 0; i < FLAG_SLOTS; ++i) c->h_flag[i] = 0;
    if (hipHostMalloc((void**)&c->h_up, bx_ctx::UP_WORDS * 4, hipHostMallocDefault) != hipSuccess) {
        bx_free(c);
        return "bx_init: allocating the pinned upload ring failed";
    }
    if (hipHostMalloc((void**)&c->h_stage, bx_ctx::STAGE_WORDS * 4, hipHostMallocDefault) != hipSuccess) {
        bx_free(c);
        return "bx_init: allocating the pinned read-back buffer failed";
    }
    // constants sanity (fp.hpp literals vs. computed)
    if (fp_encode(1u) != MONT_ONE || fp_encode(P - 11u) != MONT_NBETA || fp_encode(11u) != MONT_BETA ||
        fp_encode(3u) != MONT_THREE) {
        delete c;
        return "bx_init: field constant self-check failed";
    }
    static char init_err[512];
    const char* m = ntt_init_tables(c);
    if (!m) {
        for (int i = 0; i < 213; ++i) c->h_rc[i] = POSEIDON2_RC[i];
        for (int i = 0; i < 24; ++i) c->h_diag[i] = POSEIDON2_DIAG[i];
        m = poseidon2_upload_params(c);
    }
    if (!m) {
        // BX_TUNABLES="name=value,name=value": bx_set_tunable for every ctx of the process (A/B runs of an unmodified caller)
        if (const char* tv = getenv("BX_TUNABLES")) {
            std::string all(tv);
            size_t pos = 0;
            while (!m && pos < all.size()) {
                size_t end = all.find(',', pos);
                if (end == std::string::npos) end = all.size();
                const std::string item = all.substr(pos, end - pos);
                const size_t eq = item.find('=');
                if (eq == std::string::npos || eq == 0) m = "BX_TUNABLES: expected name=value[,name=value...]";
                else m = bx_set_tunable(c, item.substr(0, eq).c_str(), strtol(item.c_str() + eq + 1, nullptr, 10));
                pos = end + 1;
            }
        }
    }
    if (m) {
        snprintf(init_err, sizeof init_err, "bx_init: %s", m);
        bx_free(c);
        return init_err;
    }
    *out = c;
    return nullptr;
} BX_ABI_CATCH(nullptr, "bx_init")

namespace bx {
hipError_t stream_wait(bx_ctx* c) {
    if (c->gq_n && gather_flush(c) != nullptr) return hipErrorLaunchFailure;
    if (!((c->wait_blocking == 2 || (c->wait_blocking == 1 && c->wait_poll)) && c->wait_ev))
        return hipStreamSynchronize(c->stream);  // sleeps on the interrupt or busy-polls, per the device's schedule flag
    hipError_t e = hipEventRecord(c->wait_ev, c->stream);
    if (e != hipSuccess) return e;
    // usleep rounds up by the thread's timer slack (50 us by default): ask for 1 us once per waiting thread, so that
    // wait_poll_us is the real period (measured: 50 us period + default slack cost 1 % of the 3-lane rate against busy-polling)
    static thread_local bool slack_set = false;
    if (!slack_set) {
        (void)prctl(PR_SET_TIMERSLACK, 1000UL, 0UL, 0UL, 0UL);
        slack_set = true;
    }
    // Short waits first: a read-back of a few words on an idle stream is over in 10-20 us, and usleep() cannot wake up earlier than
    // ~60 us after it was called.  A trait-level caller makes ~30 such waits per proof (one per Buffer::view), which cost it 1.5 ms of
    // a 48 ms proof.  So the event is polled without sleeping for the first wait_spin_us (bounded: at most that much CPU per wait), then
    // with the sleep as before.
    if (c->wait_spin_us > 0) {
        timespec t0, t1;
        clock_gettime(CLOCK_MONOTONIC, &t0);
        for (;;) {
            if ((e = hipEventQuery(c->wait_ev)) != hipErrorNotReady) return e;
            clock_gettime(CLOCK_MONOTONIC, &t1);
            if ((t1.tv_sec - t0.tv_sec) * 1000000L + (t1.tv_nsec - t0.tv_nsec) / 1000L >= c->wait_spin_us) break;
        }
    }
    while ((e = hipEventQuery(c->wait_ev)) == hipErrorNotReady) usleep((useconds_t)c->wait_poll_us);
    return e;
}
// Apply bx_ctx::wait_blocking (see ctx.hpp).  The schedule flag is per DEVICE AND PROCESS, not per ctx: it is touched only for the one
// policy that needs it ("block"); "poll" (the default) and "spin" leave the process's setting alone — a ctx in the default mode must
// not force Spin on torch's, RCCL's or another ctx's streams, nor undo a Blocking flag an earlier ctx asked for.
// The flag is only ever set from bx_init (BX_WAIT=block), i.e. before this ctx's stream exists.  Asked for at RUN TIME
// (bx_set_tunable "wait_blocking" 1) the ctx sleep-polls its event instead and the device flag is left alone: flipping the schedule
// mode of a process whose streams were created under the other mode is accepted by this runtime and then deadlocks a later hipFree —
// hip::Device::SyncAllStreams waits on a condition variable for streams whose completions are never signalled that way (found by the
// GPU suite, round 6: an agent destroyed after such a flip hung in synth_destroy -> hipFree -> amd::Event::awaitCompletion).
void apply_wait_policy(bx_ctx* c, bool at_init) {
    c->wait_poll = false;
    if (c->wait_blocking != 1) return;
    if (!at_init || hipSetDeviceFlags(hipDeviceScheduleBlockingSync) != hipSuccess) {
        if (at_init) (void)hipGetLastError();
        c->wait_poll = true;  // not at init, or the runtime refused: sleep-poll an event instead
    }
}
const char* h2d_staged(bx_ctx* c, bx_buf dst, const uint32_t* src, size_t words) {
    BX_REQUIRE(c, words <= dst.len && words <= bx_ctx::UP_WORDS, "h2d_staged: copy too large");
    if (c->gq_n) BX_TRY(gather_flush(c));
    if (!words) return nullptr;
    if (c->up_used + words > bx_ctx::UP_WORDS) {  // wrap: earlier copies may still be reading the ring
        BX_HIP(c, stream_wait(c));
        c->up_used = 0;
    }
    uint32_t* slot = c->h_up + c->up_used;
    memcpy(slot, src, words * 4);
    c->up_used += (words + 3) & ~(size_t)3;
    BX_HIP(c, hipMemcpyAsync(dst.dptr, slot, words * 4, hipMemcpyHostToDevice, c->stream));
    return nullptr;
}
const char* d2h_batch_add(bx_ctx* c, size_t* used, bx_buf src, size_t words, const uint32_t** host) {
    BX_REQUIRE(c, words <= src.len && *used + words <= bx_ctx::STAGE_WORDS, "d2h_batch_add: the pinned landing area is full");
    if (c->gq_n) BX_TRY(gather_flush(c));
    *host = c->h_stage + *used;
    if (words) BX_HIP(c, hipMemcpyAsync(c->h_stage + *used, src.dptr, words * 4, hipMemcpyDeviceToHost, c->stream));
    *used += (words + 3) & ~(size_t)3;
    return nullptr;
}
const char* d2h_batch_wait(bx_ctx* c) { return sync_and_check_flag(c); }
const char* sync_and_check_flag(bx_ctx* c) {
    BX_HIP(c, stream_wait(c));
    volatile uint32_t* f = c->h_flag;
    const bool range = f[FLAG_SLOT_SCATTER_RANGE] != 0, index = f[FLAG_SLOT_SCATTER_INDEX] != 0;
    if (range || index) {
        for (uint32_t i = 0; i < FLAG_SLOTS; ++i) f[i] = 0;  // the stream is idle: nothing races with the reset
        if (range) return set_msg(c, "scatter: an offset is outside the destination buffer");
        return set_msg(c, "scatter: index range exceeds offsets/values");
    }
    return nullptr;
}
}  // namespace bx

extern "C" const char* bx_free(bx_ctx* c) try {
    if (!c) return nullptr;
    (void)hipSetDevice(c->device);
    (void)stream_wait(c);
    drain_profile(c);
    ntt_free_tables(c);
    if (c->d_p2) (void)hipFree(c->d_p2);
    if (c->d_scratch) (void)hipFree(c->d_scratch);
    if (c->d_gq) (void)hipFree(c->d_gq);
    for (auto& kv : c->pool_free) (void)hipFree(kv.second);
    for (int b = 0; b < 2; ++b)
        if (c->d_scan[b]) (void)hipFree(c->d_scan[b]);
    if (c->h_flag) (void)hipHostFree(c->h_flag);
    if (c->h_stage) (void)hipHostFree(c->h_stage);
    if (c->h_up) (void)hipHostFree(c->h_up);
    for (hipEvent_t e : c->event_pool) (void)hipEventDestroy(e);
    if (c->t0) (void)hipEventDestroy(c->t0);
    if (c->t1) (void)hipEventDestroy(c->t1);
    if (c->wait_ev) (void)hipEventDestroy(c->wait_ev);
    if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
    delete c;
    return nullptr;
} BX_ABI_CATCH(nullptr, "bx_free")  // c may be gone by then

extern "C" const char* bx_device_name(bx_ctx* c, char* out, size_t cap) try {
    if (!c) return "bx_device_name: null ctx";
    hipDeviceProp_t prop;
    BX_HIP(c, hipGetDeviceProperties(&prop, c->device));
    snprintf(out, cap, "%s (%s, %d CUs)", prop.name, prop.gcnArchName, prop.multiProcessorCount);
    return nullptr;
} BX_ABI_CATCH(c, "bx_device_name")

// Hal::get_hash_suite / Hal::has_unified_memory: the one suite this HAL implements is the reference's default `poseidon2`
// (ProverOpts::default(), bento/crates/workflow/src/lib.rs:246-249); MI355X HBM is not host-coherent unified memory.
extern "C" const char* bx_hash_suite_name(void) { return "poseidon2"; }
extern "C" int bx_has_unified_memory(bx_ctx*) { return 0; }

extern "C" const char* bx_set_stream(bx_ctx* c, void* s) try {
    if (!c) return "bx_set_stream: null ctx";
    BX_HIP(c, stream_wait(c));
    c->stream = s ? (hipStream_t)s : c->own_stream;
    return nullptr;
} BX_ABI_CATCH(c, "bx_set_stream")
extern "C" void* bx_get_stream(bx_ctx* c) {
    if (c && c->gq_n) (void)gather_flush(c);  // the caller is about to enqueue work of its own behind ours
    return c ? (void*)c->stream : nullptr;
}

namespace bx {
// the library's own long-lived buffers (a prover's buffer set, the circuit's tables): straight from the driver, freed with hipFree
const char* raw_alloc(bx_ctx* c, size_t words, bx_buf* out) {
    BX_REQUIRE(c, out != nullptr, "raw_alloc: null out");
    BX_REQUIRE(c, words <= ((size_t)1 << 40), "raw_alloc: more than 2^40 words asked for");
    BX_HIP(c, hipSetDevice(c->device));
    void* p = nullptr;
    BX_HIP(c, hipMalloc(&p, (words ? words : 1) * 4));
    out->dptr = p;
    out->len = words;
    return nullptr;
}
// ---- bx_alloc / bx_release pool (ctx.hpp) ----
static size_t pool_round(size_t bytes) {  // 256 B granules for small blocks, 2 MiB for large ones (so that nearby sizes match)
    const size_t g = bytes <= ((size_t)1 << 20) ? 256 : ((size_t)2 << 20);
    return (bytes + g - 1) / g * g;
}
static void pool_trim(bx_ctx* c) {  // give every cached block back to the driver (the stream is drained first: hipFree would do it anyway)
    if (c->pool_free.empty()) return;
    (void)stream_wait(c);
    for (auto& kv : c->pool_free) (void)hipFree(kv.second);
    c->pool_free.clear();
    c->pool_cached = 0;
}
static const char* pool_alloc(bx_ctx* c, size_t words, void** out) {
    BX_REQUIRE(c, words <= ((size_t)1 << 40), "bx_alloc: more than 2^40 words asked for");  // and words * 4 cannot wrap below
    const size_t bytes = pool_round((words ? words : 1) * 4);
    if (c->alloc_cache_mb > 0) {
        auto it = c->pool_free.lower_bound(bytes);
        if (it != c->pool_free.end() && it->first <= bytes + bytes / 8) {  // at most 12.5 % larger than asked for
            *out = it->second;
            c->pool_cached -= it->first;
            c->pool_live[*out] = it->first;
            c->pool_free.erase(it);
            return nullptr;
        }
    }
    void* p = nullptr;
    hipError_t e = hipMalloc(&p, bytes);
    if (e != hipSuccess && !c->pool_free.empty()) {  // out of memory with blocks idling in the list: release them and try again
        (void)hipGetLastError();
        pool_trim(c);
        e = hipMalloc(&p, bytes);
    }
    if (e != hipSuccess) return set_err(c, "hipMalloc", e, __FILE__, __LINE__);
    c->pool_live[p] = bytes;
    *out = p;
    return nullptr;
}
}  // namespace bx

extern "C" const char* bx_alloc(bx_ctx* c, size_t words, bx_buf* out) try {
    if (!c) return "bx_alloc: null ctx";
    BX_REQUIRE(c, out != nullptr, "bx_alloc: null out");
    BX_ENTER(c);
    void* p = nullptr;
    BX_TRY(pool_alloc(c, words, &p));
    out->dptr = p;
    out->len = words;
    return nullptr;
} BX_ABI_CATCH(c, "bx_alloc")
extern "C" const char* bx_alloc_zeroed(bx_ctx* c, size_t words, bx_buf* out) try {
    if (!c) return "bx_alloc_zeroed: null ctx";
    BX_REQUIRE(c, out != nullptr, "bx_alloc_zeroed: null out");
    BX_ENTER(c);
    void* p = nullptr;
    BX_TRY(pool_alloc(c, words, &p));
    out->dptr = p;
    out->len = words;
    if (hipMemsetAsync(p, 0, (words ? words : 1) * 4, c->stream) != hipSuccess) {
        (void)bx_release(c, *out);
        out->dptr = nullptr;
        return set_msg(c, "bx_alloc_zeroed: clearing the allocation failed");
    }
    return nullptr;
} BX_ABI_CATCH(c, "bx_alloc_zeroed")
extern "C" const char* bx_alloc_init(bx_ctx* c, size_t words, uint32_t value, bx_buf* out) try {
    if (!c) return "bx_alloc_init: null ctx";
    BX_REQUIRE(c, out != nullptr, "bx_alloc_init: null out");
    BX_ENTER(c);
    void* p = nullptr;
    BX_TRY(pool_alloc(c, words, &p));
    out->dptr = p;
    out->len = words;
    if (words && hipMemsetD32Async((hipDeviceptr_t)p, (int)value, words, c->stream) != hipSuccess) {
        (void)bx_release(c, *out);
        out->dptr = nullptr;
        return set_msg(c, "bx_alloc_init: filling the allocation failed");
    }
    return nullptr;
} BX_ABI_CATCH(c, "bx_alloc_init")
extern "C" const char* bx_release(bx_ctx* c, bx_buf b) try {
    if (!c) return "bx_release: null ctx";
    if (!b.dptr) return nullptr;
    BX_ENTER(c);
    auto it = c->pool_live.find(b.dptr);
    if (it == c->pool_live.end() && !c->pool_free.empty()) {  // not handed out by bx_alloc (a raw_alloc / foreign block) — or released TWICE:
        for (auto& kv : c->pool_free)                         // a block that idles in the pool must not reach hipFree behind the pool's back
            BX_REQUIRE(c, kv.second != b.dptr, "bx_release: this buffer was already released");
    }
    if (it != c->pool_live.end()) {
        const size_t cap = it->second;
        c->pool_live.erase(it);
        // keep it for the next request of that size: whoever gets it uses it on this ctx's stream, behind everything enqueued so far.
        // (Not on an adopted stream: its owner may still be using the memory in work this library has not seen.)
        if (c->alloc_cache_mb > 0 && c->stream == c->own_stream && c->pool_cached + cap <= ((size_t)c->alloc_cache_mb << 20)) {
            c->pool_free.emplace(cap, b.dptr);
            c->pool_cached += cap;
            return nullptr;
        }
    }
    BX_HIP(c, stream_wait(c));
    BX_HIP(c, hipFree(b.dptr));
    return nullptr;
} BX_ABI_CATCH(c, "bx_release")
extern "C" const char* bx_h2d(bx_ctx* c, bx_buf dst, const uint32_t* src, size_t words) try {
    if (!c) return "bx_h2d: null ctx";
    BX_REQUIRE(c, words <= dst.len, "bx_h2d: copy larger than the buffer");
    BX_ENTER(c);
    BX_HIP(c, hipMemcpyAsync(dst.dptr, src, words * 4, hipMemcpyHostToDevice, c->stream));
    BX_HIP(c, stream_wait(c));  // src may be pageable and freed by the caller right after
    return nullptr;
} BX_ABI_CATCH(c, "bx_h2d")
extern "C" const char* bx_d2h(bx_ctx* c, uint32_t* dst, bx_buf src, size_t words) try {
    if (!c) return "bx_d2h: null ctx";
    BX_REQUIRE(c, words <= src.len, "bx_d2h: copy larger than the buffer");
    BX_ENTER(c);
    if (words == 0) return sync_and_check_flag(c);
    if (words <= bx_ctx::STAGE_WORDS) {
        BX_HIP(c, hipMemcpyAsync(c->h_stage, src.dptr, words * 4, hipMemcpyDeviceToHost, c->stream));
        BX_TRY(sync_and_check_flag(c));
        memcpy(dst, c->h_stage, words * 4);
        return nullptr;
    }
    BX_HIP(c, hipMemcpyAsync(dst, src.dptr, words * 4, hipMemcpyDeviceToHost, c->stream));
    return sync_and_check_flag(c);
} BX_ABI_CATCH(c, "bx_d2h")
extern "C" const char* bx_d2d(bx_ctx* c, bx_buf dst, bx_buf src, size_t words) try {
    if (!c) return "bx_d2d: null ctx";
    BX_REQUIRE(c, words <= src.len && words <= dst.len, "bx_d2d: copy larger than a buffer");
    BX_ENTER(c);
    BX_HIP(c, hipMemcpyAsync(dst.dptr, src.dptr, words * 4, hipMemcpyDeviceToDevice, c->stream));
    return nullptr;
} BX_ABI_CATCH(c, "bx_d2d")
extern "C" const char* bx_eltwise_copy_elem_slice(bx_ctx* c, bx_buf into, const uint32_t* from, size_t from_len, size_t from_rows, size_t from_cols,
                                                  size_t from_offset, size_t from_stride, size_t into_offset, size_t into_stride) try {
    if (!c) return "bx_eltwise_copy_elem_slice: null ctx";
    if (!from_rows || !from_cols) return nullptr;
    BX_REQUIRE(c, from != nullptr && into.dptr != nullptr, "eltwise_copy_elem_slice: null buffer");
    // last word touched on each side, without wrapping: offset + (rows - 1) * stride + cols <= len
    BX_REQUIRE(c, from_cols <= from_len && from_offset <= from_len - from_cols && mul_le(from_rows - 1, from_stride, from_len - from_cols - from_offset),
               "eltwise_copy_elem_slice: the region leaves the source slice");
    BX_REQUIRE(c, from_cols <= into.len && into_offset <= into.len - from_cols && mul_le(from_rows - 1, into_stride, into.len - from_cols - into_offset),
               "eltwise_copy_elem_slice: the region leaves the destination buffer");
    BX_REQUIRE(c, from_rows == 1 || into_stride >= from_cols, "eltwise_copy_elem_slice: destination rows overlap (into_stride < from_cols)");
    BX_REQUIRE(c, from_stride <= (SIZE_MAX >> 2) && into_stride <= (SIZE_MAX >> 2), "eltwise_copy_elem_slice: stride too large");
    BX_ENTER(c);
    OpScope op(c, "eltwise_copy_elem_slice", 8.0 * (double)from_rows * (double)from_cols);
    if (from_rows == 1 || (from_stride == from_cols && into_stride == from_cols)) {  // one contiguous run
        BX_HIP(c, hipMemcpyAsync((uint32_t*)into.dptr + into_offset, from + from_offset, from_rows * from_cols * 4, hipMemcpyHostToDevice, c->stream));
    } else {
        // the pitches of a 2-D copy must be at least its width: a source whose rows overlap (from_stride < from_cols) goes row by row
        if (from_stride >= from_cols) {
            BX_HIP(c, hipMemcpy2DAsync((uint32_t*)into.dptr + into_offset, into_stride * 4, from + from_offset, from_stride * 4, from_cols * 4, from_rows,
                                       hipMemcpyHostToDevice, c->stream));
        } else {
            for (size_t r = 0; r < from_rows; ++r)
                BX_HIP(c, hipMemcpyAsync((uint32_t*)into.dptr + into_offset + r * into_stride, from + from_offset + r * from_stride, from_cols * 4,
                                         hipMemcpyHostToDevice, c->stream));
        }
    }
    BX_HIP(c, stream_wait(c));  // `from` may be pageable and freed by the caller right after
    return nullptr;
} BX_ABI_CATCH(c, "bx_eltwise_copy_elem_slice")
extern "C" const char* bx_sync(bx_ctx* c) try {
    if (!c) return "bx_sync: null ctx";
    BX_ENTER(c);
    return sync_and_check_flag(c);
} BX_ABI_CATCH(c, "bx_sync")

extern "C" const char* bx_timer_start(bx_ctx* c) try {
    if (!c) return "bx_timer_start: null ctx";
    BX_ENTER(c);
    BX_HIP(c, hipEventRecord(c->t0, c->stream));
    return nullptr;
} BX_ABI_CATCH(c, "bx_timer_start")
extern "C" const char* bx_timer_stop(bx_ctx* c, float* ms) try {
    if (!c) return "bx_timer_stop: null ctx";
    BX_ENTER(c);
    BX_HIP(c, hipEventRecord(c->t1, c->stream));
    BX_HIP(c, hipEventSynchronize(c->t1));
    BX_HIP(c, hipEventElapsedTime(ms, c->t0, c->t1));
    return nullptr;
} BX_ABI_CATCH(c, "bx_timer_stop")

extern "C" const char* bx_profile_enable(bx_ctx* c, int on) try {
    if (!c) return "bx_profile_enable: null ctx";
    if (!on) drain_profile(c);
    c->profile = on != 0;
    return nullptr;
} BX_ABI_CATCH(c, "bx_profile_enable")
extern "C" const char* bx_profile_reset(bx_ctx* c) try {
    if (!c) return "bx_profile_reset: null ctx";
    drain_profile(c);
    c->prof_agg.clear();
    return nullptr;
} BX_ABI_CATCH(c, "bx_profile_reset")
extern "C" const char* bx_profile_report(bx_ctx* c, char* out, size_t cap) try {
    if (!c) return "bx_profile_report: null ctx";
    drain_profile(c);
    std::string s = "{";
    bool first = true;
    char line[256];
    for (auto& kv : c->prof_agg) {
        snprintf(line, sizeof line, "%s\"%s\":{\"calls\":%ld,\"ms\":%.6f,\"alg_bytes\":%.0f}", first ? "" : ",", kv.first.c_str(),
                 kv.second.calls, kv.second.ms, kv.second.bytes);
        s += line;
        first = false;
    }
    s += "}";
    BX_REQUIRE(c, s.size() + 1 <= cap, "bx_profile_report: output buffer too small");
    memcpy(out, s.c_str(), s.size() + 1);
    return nullptr;
} BX_ABI_CATCH(c, "bx_profile_report")

extern "C" const char* bx_set_tunable(bx_ctx* c, const char* name, long value) try {
    if (!c) return "bx_set_tunable: null ctx";
    BX_REQUIRE(c, name != nullptr, "bx_set_tunable: null name");
    if (!strcmp(name, "ntt_block_log")) {
        BX_REQUIRE(c, value >= 1 && value <= TW_LOG, "ntt_block_log out of range [1,13]");
        c->ntt_block_log = value;
    } else if (!strcmp(name, "ntt_tile_log")) {
        BX_REQUIRE(c, value >= 6 && value <= 15, "ntt_tile_log out of range [6,15]");
        c->ntt_tile_log = value;
    } else if (!strcmp(name, "hash_rows_block")) {
        BX_REQUIRE(c, value == 64 || value == 128 || value == 256, "hash_rows_block must be 64, 128 or 256");
        c->hash_rows_block = value;
    } else if (!strcmp(name, "ntt_fast")) {
        c->ntt_fast = value != 0;
    } else if (!strcmp(name, "ntt_tile_a_log")) {
        BX_REQUIRE(c, value >= 10 && value <= 13, "ntt_tile_a_log out of range [10,13]");
        c->ntt_tile_a_log = value;
    } else if (!strcmp(name, "ntt_tile_b_log")) {
        BX_REQUIRE(c, value >= 10 && value <= 14, "ntt_tile_b_log out of range [10,14]");
        c->ntt_tile_b_log = value;
    } else if (!strcmp(name, "ntt_cols_per_wg")) {
        BX_REQUIRE(c, value >= 1 && value <= 16, "ntt_cols_per_wg out of range [1,16]");
        c->ntt_cols_per_wg = value;
    } else if (!strcmp(name, "ntt_group_cols")) {
        BX_REQUIRE(c, value >= 0 && value <= 4096, "ntt_group_cols out of range [0,4096]");
        c->ntt_group_cols = value;
    } else if (!strcmp(name, "ntt_tile_b_wide")) {
        c->ntt_tile_b_wide = value != 0;
    } else if (!strcmp(name, "wait_spin_us")) {
        BX_REQUIRE(c, value >= 0 && value <= 10000, "wait_spin_us out of range [0, 10000]");
        c->wait_spin_us = value;
    } else if (!strcmp(name, "alloc_cache_mb")) {
        BX_REQUIRE(c, value >= 0 && value <= (288 << 10), "alloc_cache_mb out of range [0, 294912]");
        BX_ENTER(c);
        c->alloc_cache_mb = value;
        while (!c->pool_free.empty() && c->pool_cached > ((size_t)value << 20)) pool_trim(c);  // a lower cap takes effect at once
    } else if (!strcmp(name, "gather_defer")) {
        BX_ENTER(c);
        c->gather_defer = value != 0;
    } else if (!strcmp(name, "scan_lookback")) {
        c->scan_lookback = value != 0;
    } else if (!strcmp(name, "eval_x4")) {
        c->eval_x4 = value != 0;
    } else if (!strcmp(name, "deep_bitrev")) {
        c->deep_bitrev = value != 0;
    } else if (!strcmp(name, "fold_deep")) {
        BX_REQUIRE(c, value >= 1 && value <= 3, "fold_deep must be 1, 2 or 3");
        c->fold_deep = value;
    } else if (!strcmp(name, "dev_draws")) {
        BX_REQUIRE(c, value == 0 || value == 1, "dev_draws must be 0 or 1");
        c->dev_draws = value;
    } else if (!strcmp(name, "fold_deep_min_lanes")) {
        BX_REQUIRE(c, value >= 1, "fold_deep_min_lanes must be positive");
        c->fold_deep_min_lanes = value;
    } else if (!strcmp(name, "fold_quad")) {
        BX_REQUIRE(c, value == 0 || value == 1, "fold_quad must be 0 or 1");
        c->fold_quad = value;
    } else if (!strcmp(name, "fold_quad_wg")) {
        BX_REQUIRE(c, value >= 16 && value <= 512 && (value & (value - 1)) == 0, "fold_quad_wg must be a power of two in [16, 512]");
        c->fold_quad_wg = value;
    } else if (!strcmp(name, "wait_blocking")) {
        BX_REQUIRE(c, value >= 0 && value <= 2, "wait_blocking must be 0 (busy-poll), 1 (sleep until the completion interrupt) or 2 (sleep-poll an event)");
        c->wait_blocking = value;
        BX_ENTER(c);
        apply_wait_policy(c, false);
    } else if (!strcmp(name, "wait_poll_us")) {
        BX_REQUIRE(c, value >= 1 && value <= 10000, "wait_poll_us out of range [1, 10000]");
        c->wait_poll_us = value;
    } else if (!strcmp(name, "fold_fuse_below")) {
        BX_REQUIRE(c, value >= 0, "fold_fuse_below must be >= 0");
        c->fold_fuse_below = value;
    } else {
        return set_msg(c, "bx_set_tunable: unknown tunable");
    }
    return nullptr;
} BX_ABI_CATCH(c, "bx_set_tunable")
