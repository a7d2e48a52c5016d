// Race / leak check of the native feed loop (boundless_amd/csrc/agent.cpp): built by tests/test_agent_sanitizers_cpu.py with
// -fsanitize=thread (and again with address,undefined) from agent.cpp + planner.cpp + this file, no GPU and no HIP runtime.
// 3 "devices" x 2 lanes x 2 threads each hammer the ONE in-memory hot store / task db with a prover that fails at random;
// every task must end `done` exactly once or `failed` after its retries, and the sanitizer must stay silent.  The lanes of
// device 1 are ten times slower: since every lane claims when idle from the shared task db (the work-stealing queue of
// BASELINE configs[2], in native code), they end up with a small share of the batch and nobody waits for them.
#include <atomic>
#include <cstdio>
#include <cstring>
#include <chrono>
#include <string>
#include <thread>

#include "bx_agent.h"
#include "bx_circuit.h"

// Link stubs for the device entry points the default (HIP) prover ops reference; this test injects its own prover ops, so
// none of them is ever called.
extern "C" {
const char* bx_init(int, bx_ctx**) { return "stub"; }
const char* bx_free(bx_ctx*) { return "stub"; }
const char* bx_prover_create(bx_ctx*, const bx_segment_params*, bx_prover**) { return "stub"; }
const char* bx_prover_destroy(bx_prover*) { return "stub"; }
size_t bx_prover_seal_words(const bx_prover*) { return 0; }
const char* bx_prove_segment_bytes(bx_prover*, const uint8_t*, size_t, uint32_t*, size_t, size_t*) { return "stub"; }
const char* bx_prover_control_id(bx_prover*, uint32_t*) { return "stub"; }
const bx_circuit_ops* bx_synthetic_circuit(void) { return nullptr; }
const char* bx_verifier_ctx_create(bx_verifier_ctx**) { return "stub"; }
void bx_verifier_ctx_destroy(bx_verifier_ctx*) {}
const char* bx_verifier_ctx_add_control_id(bx_verifier_ctx*, uint32_t, const uint32_t*) { return "stub"; }
size_t bx_verifier_ctx_count(const bx_verifier_ctx*, uint32_t) { return 0; }
const char* bx_verify_segment_with_context(const uint32_t* seal, size_t words, const bx_circuit_ops*, const bx_verifier_ctx*) {
    return (words == 64 && seal[0] == 7u) ? nullptr : "bad seal";
}
}

namespace bx {  // the library-internal lookup hip_prover_for consults (csrc/control_id.cpp): never reached with injected prover ops
bool verifier_ctx_contains(const bx_verifier_ctx*, uint32_t, const uint32_t*) { return false; }
}

static std::atomic<uint64_t> g_calls{0};
static size_t seal_words(void*, uint32_t, uint32_t) { return 64; }
static const char* prove(void*, uint32_t lane, uint32_t, const uint8_t* segment, size_t len, uint32_t* seal, size_t cap, size_t* words) {
    uint64_t index = 0, seed = 0;
    if (bx_segment_decode(segment, len, &index, nullptr, &seed)) return "not a segment";
    uint64_t n = g_calls.fetch_add(1);
    if (cap < 64) return "cap";
    std::this_thread::sleep_for(std::chrono::microseconds(lane / 2 == 1 ? 10000 : 100));  // device 1 is the slow GPU
    if ((n * 2654435761u >> 7) % 5 == 0) return "hipErrorLaunchFailure (injected)";
    for (uint32_t i = 0; i < 64; ++i) seal[i] = (uint32_t)(seed + index + i + lane * 0);
    seal[0] = ((n >> 3) % 7 == 0) ? 9u : 7u;  // some seals fail verification -> retried from the finisher thread
    *words = 64;
    return nullptr;
}

// Second phase: a PLANNED job (bx_plan_job: K proves -> log-depth tail of stand-in joins -> resolve -> finalize) through the same
// 6 lanes with the same randomly failing prover: every task ends done exactly once, and no task was claimed before each of its
// prerequisites was done (timestamps of the task db), whatever the interleaving of lanes, finishers and retries.  Run twice: serial
// lanes, then with the claim-ahead fetcher (bx_agent_config.prefetch) — a third thread per lane that claims and GETs.
static int planned_job_phase(int prefetch) {
    bx_mem_store* store = nullptr;
    bx_mem_taskdb* db = nullptr;
    if (bx_mem_store_create(&store) || bx_mem_taskdb_create(&db)) return 1;
    bx_hot_store_ops sops = bx_mem_store_ops(store);
    bx_taskdb_ops tops = bx_mem_taskdb_ops(db);
    const int K = 97;
    for (int i = 0; i < K; ++i) {
        uint8_t wire[BX_SEGMENT_WIRE_BYTES];
        bx_segment_encode((uint64_t)i, 10, 5000 + (uint64_t)i, wire);
        std::string key = "job:dag:segments:" + std::to_string(i);
        if (sops.set_ex(sops.user, key.c_str(), wire, sizeof wire, 0, nullptr, 0) != 0) return 1;
    }
    bx_job_plan plan;
    memset(&plan, 0, sizeof plan);
    plan.prove_retries = plan.join_retries = plan.resolve_retries = plan.finalize_retries = 50;  // the injected failures never exhaust them
    uint64_t created = 0;
    uint64_t root = 0;
    if (const char* e = bx_plan_job(db, "dag", K, &plan, &created, &root)) {
        fprintf(stderr, "plan: %s\n", e);
        return 1;
    }
    if (created != (uint64_t)(2 * K - 1 + 2) || root != (uint64_t)(2 * K - 2)) return 1;
    bx_agent_config cfg;
    memset(&cfg, 0, sizeof cfg);
    cfg.inflight = 2;
    cfg.n_devices = 3;
    cfg.devices[0] = 0, cfg.devices[1] = 1, cfg.devices[2] = 2;
    cfg.synthetic = 1;
    cfg.poll_time = 0.001;
    cfg.join_po2 = 9;
    cfg.prefetch = prefetch;  // second run: every lane has a fetcher thread claiming one task ahead and doing its GET ...
    cfg.monitor_requeue = prefetch;  // ... and the requeue monitor sweeps the table every millisecond beside them (nothing times out)
    cfg.requeue_poll_interval = 0.001;
    cfg.lift_po2 = 9;  // every Prove task also runs the stand-in lift: the segment seal is verified on a helper thread beside it
    snprintf(cfg.also_streams, sizeof cfg.also_streams, "aux");
    bx_segment_prover_ops pops{nullptr, seal_words, prove, nullptr, nullptr};
    bx_agent* agent = nullptr;
    if (const char* e = bx_agent_create(&cfg, &sops, &tops, &pops, &agent)) {
        fprintf(stderr, "create: %s\n", e);
        return 1;
    }
    uint64_t done = 0;
    if (const char* e = bx_agent_poll_work(agent, 5, &done)) {
        fprintf(stderr, "poll: %s\n", e);
        return 1;
    }
    bx_job_info job;
    if (bx_mem_taskdb_job_info(db, "dag", &job)) return 1;
    if (job.state != BX_JOB_DONE || job.done != created || done != created) {
        fprintf(stderr, "planned job: state %d done %llu of %llu (agent %llu) failed %llu: %s\n", job.state, (unsigned long long)job.done,
                (unsigned long long)created, (unsigned long long)done, (unsigned long long)job.failed, job.error);
        return 1;
    }
    // dependencies in time: replay the planner and compare claim / completion times
    bx_planner* pl = nullptr;
    if (bx_planner_create(&pl)) return 1;
    uint64_t n = 0;
    for (int i = 0; i < K; ++i)
        if (bx_planner_enqueue_segment(pl, &n)) return 1;
    if (bx_planner_finish(pl, &n)) return 1;
    int joins = 0;
    for (uint64_t k = 0; k < bx_planner_task_count(pl); ++k) {
        bx_plan_task t;
        if (bx_planner_get_task(pl, k, &t)) return 1;
        if (t.command != BX_PLAN_JOIN) continue;
        ++joins;
        bx_task_info me, pre;
        if (bx_mem_taskdb_task_info(db, "dag", std::to_string(t.task_number).c_str(), &me)) return 1;
        for (uint32_t d = 0; d < t.n_depends_on; ++d) {
            if (bx_mem_taskdb_task_info(db, "dag", std::to_string(t.depends_on[d]).c_str(), &pre)) return 1;
            if (pre.state != BX_TASK_DONE || me.started_s < pre.updated_s) {
                fprintf(stderr, "join %llu was claimed at %.6f before its prerequisite %llu was done at %.6f\n", (unsigned long long)t.task_number,
                        me.started_s, (unsigned long long)t.depends_on[d], pre.updated_s);
                return 1;
            }
        }
    }
    bx_planner_destroy(pl);
    if (joins != K - 1) return 1;
    bx_task_info res, fin;
    if (bx_mem_taskdb_task_info(db, "dag", "resolve", &res) || bx_mem_taskdb_task_info(db, "dag", "finalize", &fin)) return 1;
    if (fin.started_s < res.updated_s) return 1;
    if (bx_agent_destroy(agent)) return 1;
    // what is left in the store: the root receipt and the rollup (every join unlinked its children, every prove its segment)
    size_t keys = bx_mem_store_key_count(store);
    bx_mem_taskdb_destroy(db);
    bx_mem_store_destroy(store);
    if (keys != 2) {
        fprintf(stderr, "planned job: %zu keys left in the hot store\n", keys);
        return 1;
    }
    printf("planned job ok%s: %llu tasks (%d proves, %d joins, resolve, finalize)\n", prefetch ? " (prefetch)" : "", (unsigned long long)created, K, joins);
    return 0;
}

// Third phase: bx_agent_stop in the middle of a planned job, with fetchers and the requeue monitor running: poll_work returns, no
// task is left 'running' (what a fetcher had claimed is run before its lane leaves), and a second poll_work finishes the job.
static int stop_and_resume_phase(void) {
    bx_mem_store* store = nullptr;
    bx_mem_taskdb* db = nullptr;
    if (bx_mem_store_create(&store) || bx_mem_taskdb_create(&db)) return 1;
    bx_hot_store_ops sops = bx_mem_store_ops(store);
    bx_taskdb_ops tops = bx_mem_taskdb_ops(db);
    const int K = 120;
    for (int i = 0; i < K; ++i) {
        uint8_t wire[BX_SEGMENT_WIRE_BYTES];
        bx_segment_encode((uint64_t)i, 10, 9000 + (uint64_t)i, wire);
        std::string key = "job:halt:segments:" + std::to_string(i);
        if (sops.set_ex(sops.user, key.c_str(), wire, sizeof wire, 0, nullptr, 0) != 0) return 1;
    }
    bx_job_plan plan;
    memset(&plan, 0, sizeof plan);
    plan.prove_retries = plan.join_retries = plan.resolve_retries = plan.finalize_retries = 50;
    uint64_t created = 0;
    if (bx_plan_job(db, "halt", K, &plan, &created, nullptr)) return 1;
    bx_agent_config cfg;
    memset(&cfg, 0, sizeof cfg);
    cfg.inflight = 2;
    cfg.n_devices = 3;
    cfg.devices[0] = 0, cfg.devices[1] = 1, cfg.devices[2] = 2;
    cfg.synthetic = 1;
    cfg.poll_time = 0.001;
    cfg.join_po2 = 9;
    cfg.prefetch = 1;
    cfg.monitor_requeue = 1;
    cfg.requeue_poll_interval = 0.001;
    snprintf(cfg.also_streams, sizeof cfg.also_streams, "aux");
    bx_segment_prover_ops pops{nullptr, seal_words, prove, nullptr, nullptr};
    bx_agent* agent = nullptr;
    if (bx_agent_create(&cfg, &sops, &tops, &pops, &agent)) return 1;
    uint64_t first = 0, second = 0;
    std::thread stopper([agent] {
        std::this_thread::sleep_for(std::chrono::milliseconds(40));
        bx_agent_stop(agent);
    });
    const char* e = bx_agent_poll_work(agent, -1, &first);  // no idle limit: only the stop ends it
    stopper.join();
    if (e) {
        fprintf(stderr, "stop phase: %s\n", e);
        return 1;
    }
    bx_job_info job;
    if (bx_mem_taskdb_job_info(db, "halt", &job)) return 1;
    if (job.running != 0 || job.failed != 0 || job.done != first || job.done + job.ready + job.pending != created) {
        fprintf(stderr, "after the stop: running %llu failed %llu done %llu (agent %llu) ready %llu pending %llu of %llu\n", (unsigned long long)job.running,
                (unsigned long long)job.failed, (unsigned long long)job.done, (unsigned long long)first, (unsigned long long)job.ready,
                (unsigned long long)job.pending, (unsigned long long)created);
        return 1;
    }
    if (bx_agent_destroy(agent)) return 1;
    agent = nullptr;
    if (bx_agent_create(&cfg, &sops, &tops, &pops, &agent)) return 1;  // a fresh agent: its stop flag is clear
    if (const char* e2 = bx_agent_poll_work(agent, 5, &second)) {
        fprintf(stderr, "resume phase: %s\n", e2);
        return 1;
    }
    if (bx_mem_taskdb_job_info(db, "halt", &job)) return 1;
    if (job.state != BX_JOB_DONE || first + second != created) {
        fprintf(stderr, "after the resume: state %d, %llu + %llu of %llu\n", job.state, (unsigned long long)first, (unsigned long long)second,
                (unsigned long long)created);
        return 1;
    }
    if (bx_agent_destroy(agent)) return 1;
    bx_mem_taskdb_destroy(db);
    bx_mem_store_destroy(store);
    printf("stop and resume ok: %llu tasks before the stop, %llu after\n", (unsigned long long)first, (unsigned long long)second);
    return 0;
}

int main() {
    if (planned_job_phase(0) || planned_job_phase(1) || stop_and_resume_phase()) return 1;
    bx_mem_store* store = nullptr;
    bx_mem_taskdb* db = nullptr;
    if (bx_mem_store_create(&store) || bx_mem_taskdb_create(&db)) return 1;
    bx_hot_store_ops sops = bx_mem_store_ops(store);
    bx_taskdb_ops tops = bx_mem_taskdb_ops(db);
    const int N = 600;
    for (int i = 0; i < N; ++i) {
        uint8_t wire[BX_SEGMENT_WIRE_BYTES];
        bx_segment_encode((uint64_t)i, 10, 1000 + (uint64_t)i, wire);
        std::string key = "job:race:segments:" + std::to_string(i), task = "p" + std::to_string(i);
        std::string def = "{\"Prove\":{\"index\":" + std::to_string(i) + "}}";
        if (i % 97 != 13 && sops.set_ex(sops.user, key.c_str(), wire, sizeof wire, 0, nullptr, 0) != 0) return 1;  // a few blobs are missing
        if (bx_mem_taskdb_create_task(db, "prove", "race", task.c_str(), def.c_str(), 4)) return 1;
    }
    bx_agent_config cfg;
    memset(&cfg, 0, sizeof cfg);
    cfg.inflight = 2;
    cfg.n_devices = 3;
    cfg.devices[0] = 0, cfg.devices[1] = 1, cfg.devices[2] = 2;
    cfg.synthetic = 1;
    cfg.poll_time = 0.001;
    bx_segment_prover_ops pops{nullptr, seal_words, prove, nullptr, nullptr};
    {   // without the opt-in the synthetic wire format is refused
        bx_agent_config off = cfg;
        off.synthetic = 0;
        bx_agent* none = nullptr;
        const char* e = bx_agent_create(&off, &sops, &tops, &pops, &none);
        if (!e || !strstr(e, "synthetic")) return 1;
    }
    bx_agent* agent = nullptr;
    if (const char* e = bx_agent_create(&cfg, &sops, &tops, &pops, &agent)) {
        fprintf(stderr, "create: %s\n", e);
        return 1;
    }
    uint64_t done = 0;
    if (const char* e = bx_agent_poll_work(agent, 3, &done)) {
        fprintf(stderr, "poll: %s\n", e);
        return 1;
    }
    size_t n_done = bx_mem_taskdb_count(db, BX_TASK_DONE), n_failed = bx_mem_taskdb_count(db, BX_TASK_FAILED);
    size_t n_other = bx_mem_taskdb_count(db, BX_TASK_READY) + bx_mem_taskdb_count(db, BX_TASK_RUNNING);
    char metrics[1 << 15];
    bx_agent_metrics(agent, metrics, sizeof metrics);
    uint64_t per_dev[3] = {0, 0, 0}, lane_sum = 0;
    if (bx_agent_lane_count(agent) != 6) return 1;
    for (uint32_t l = 0; l < 6; ++l) {
        if (bx_agent_lane_device(agent, l) != (int32_t)(l / 2)) return 1;
        per_dev[l / 2] += bx_agent_lane_tasks_done(agent, l);
        lane_sum += bx_agent_lane_tasks_done(agent, l);
    }
    // the slow device ends with the smallest share (how much smaller depends on the sanitizer's own overhead per task: not asserted)
    if (lane_sum != done || per_dev[1] >= per_dev[0] || per_dev[1] >= per_dev[2] || per_dev[1] == 0) {
        fprintf(stderr, "work stealing: per device %llu %llu %llu of %llu\n", (unsigned long long)per_dev[0], (unsigned long long)per_dev[1],
                (unsigned long long)per_dev[2], (unsigned long long)done);
        return 1;
    }
    if (bx_agent_destroy(agent)) return 1;
    // every stored receipt has its segment unlinked; failed tasks keep (or never had) their segment blob
    size_t keys = bx_mem_store_key_count(store);
    bx_mem_taskdb_destroy(db);
    bx_mem_store_destroy(store);
    if (done != n_done || n_done + n_failed != (size_t)N || n_other != 0 || n_done < 500 || n_failed < 6) {
        fprintf(stderr, "done=%llu n_done=%zu n_failed=%zu other=%zu keys=%zu\n", (unsigned long long)done, n_done, n_failed, n_other, keys);
        return 1;
    }
    if (!strstr(metrics, "task_operations_total{task_name=\"prove\",operation_type=\"complete\",status=\"success\"}")) return 1;
    printf("agent_race_check ok: done %zu failed %zu prove calls %llu keys %zu, per device %llu/%llu/%llu\n", n_done, n_failed,
           (unsigned long long)g_calls.load(), keys, (unsigned long long)per_dev[0], (unsigned long long)per_dev[1], (unsigned long long)per_dev[2]);
    return 0;
}
