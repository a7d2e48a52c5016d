// Prove-stream feed loop: the native host runtime between the task queue and bx_prove_segment.
//
// Restates, for the `prove` task stream only:
//   Agent::poll_work      bento/crates/workflow/src/lib.rs:279-442   (claim, run, retry / fail bookkeeping, SIGTERM flag)
//   Agent::process_work   bento/crates/workflow/src/lib.rs:445-530   (TaskType dispatch, update_task_done)
//   tasks::prove::prover  bento/crates/workflow/src/tasks/prove.rs:18-135 (fetch -> prove -> verify -> store -> cleanup)
//   redis helpers         bento/crates/workflow/src/redis.rs:19-63   (operation names + redis_operations metrics)
//   metric definitions    bento/crates/workflow-common/src/metrics.rs:61-70,108-117
// Redis/Postgres themselves are out of scope; they are the callback tables of include/bx_agent.h.  The `lift` step needs the
// recursion circuit (not available offline) and the built-in prover proves the SYNTHETIC circuit of
// bx_prover.h: its seals are stored under job:{id}:synthetic_receipts:{task}, never under the reference's recursion-receipt
// key, and only when the agent was created with cfg.synthetic = 1.  A real prover plugs in through
// bx_segment_prover_ops::prove_blob (raw bytes in, raw bytes out) and then the reference's keys are used.
#include <atomic>
#include <cerrno>
#include <chrono>
#include <condition_variable>
#include <future>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <map>
#include <memory>
#include <mutex>
#include <set>
#include <shared_mutex>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include "../../include/bx_agent.h"
#include "../../include/bx_circuit.h"

namespace bx {
bool verifier_ctx_contains(const bx_verifier_ctx* v, uint32_t po2, const uint32_t root[8]);  // control_id.cpp
}

namespace {

using Clock = std::chrono::steady_clock;
double secs_since(Clock::time_point t0) { return std::chrono::duration<double>(Clock::now() - t0).count(); }

thread_local std::string tl_err;
const char* fail(const std::string& m) {
    tl_err = m;
    return tl_err.c_str();
}

// ------------------------------------------------------------------------------------------------------ tiny JSON ----
// Enough of JSON to read an externally tagged serde enum: {"Variant": {...fields...}}.
struct JVal {
    enum Kind { Null, Bool, Num, Str, Arr, Obj } kind = Null;
    bool b = false;
    bool is_uint = false;
    uint64_t u = 0;
    std::string s;
    std::vector<JVal> arr;
    std::vector<std::pair<std::string, JVal>> obj;
    const JVal* find(const char* k) const {
        for (auto& kv : obj)
            if (kv.first == k) return &kv.second;
        return nullptr;
    }
};

struct JParser {
    const char* p;
    const char* end;
    bool ok = true;
    void ws() {
        while (p < end && (*p == ' ' || *p == '\t' || *p == '\n' || *p == '\r')) ++p;
    }
    bool lit(const char* t) {
        size_t n = strlen(t);
        if ((size_t)(end - p) >= n && memcmp(p, t, n) == 0) {
            p += n;
            return true;
        }
        return false;
    }
    std::string str() {
        std::string out;
        ++p;  // opening quote
        while (p < end && *p != '"') {
            if (*p == '\\' && p + 1 < end) {
                ++p;
                switch (*p) {
                    case 'n': out += '\n'; break;
                    case 't': out += '\t'; break;
                    case 'r': out += '\r'; break;
                    case 'b': out += '\b'; break;
                    case 'f': out += '\f'; break;
                    case 'u':  // keep \uXXXX escapes verbatim: no key or value this agent reads contains one
                        out += "\\u";
                        break;
                    default: out += *p;
                }
                ++p;
            } else {
                out += *p++;
            }
        }
        if (p >= end) {
            ok = false;
            return out;
        }
        ++p;
        return out;
    }
    JVal value(int depth = 0) {
        JVal v;
        ws();
        if (p >= end || depth > 32) {
            ok = false;
            return v;
        }
        if (*p == '{') {
            v.kind = JVal::Obj;
            ++p;
            ws();
            if (p < end && *p == '}') {
                ++p;
                return v;
            }
            while (ok) {
                ws();
                if (p >= end || *p != '"') {
                    ok = false;
                    break;
                }
                std::string k = str();
                ws();
                if (p >= end || *p != ':') {
                    ok = false;
                    break;
                }
                ++p;
                v.obj.emplace_back(std::move(k), value(depth + 1));
                ws();
                if (p < end && *p == ',') {
                    ++p;
                    continue;
                }
                if (p < end && *p == '}') {
                    ++p;
                    break;
                }
                ok = false;
            }
        } else if (*p == '[') {
            v.kind = JVal::Arr;
            ++p;
            ws();
            if (p < end && *p == ']') {
                ++p;
                return v;
            }
            while (ok) {
                v.arr.push_back(value(depth + 1));
                ws();
                if (p < end && *p == ',') {
                    ++p;
                    continue;
                }
                if (p < end && *p == ']') {
                    ++p;
                    break;
                }
                ok = false;
            }
        } else if (*p == '"') {
            v.kind = JVal::Str;
            v.s = str();
        } else if (lit("true")) {
            v.kind = JVal::Bool;
            v.b = true;
        } else if (lit("false")) {
            v.kind = JVal::Bool;
        } else if (lit("null")) {
            v.kind = JVal::Null;
        } else {
            v.kind = JVal::Num;
            const char* s = p;
            bool integral = true;
            if (p < end && *p == '-') {
                integral = false;
                ++p;
            }
            while (p < end && ((*p >= '0' && *p <= '9') || *p == '.' || *p == 'e' || *p == 'E' || *p == '+' || *p == '-')) {
                if (*p < '0' || *p > '9') integral = false;
                ++p;
            }
            if (p == s) {
                ok = false;
                return v;
            }
            v.s.assign(s, p);
            if (integral && v.s.size() <= 20) {
                errno = 0;
                v.u = strtoull(v.s.c_str(), nullptr, 10);
                v.is_uint = errno == 0;
            }
        }
        return v;
    }
};

bool parse_json(const char* text, JVal* out) {
    JParser jp{text, text + strlen(text)};
    *out = jp.value();
    jp.ws();
    return jp.ok && jp.p == jp.end;
}

// ---------------------------------------------------------------------------------------------------------- metrics ----
const double TASK_BUCKETS[] = {0.1, 0.5, 1.0, 2.5, 5.0, 10.0, 25.0, 50.0, 100.0, 250.0, 500.0};  // metrics.rs:108-112
// next-gen worker loop (prover/crates/workflow-common/src/metrics.rs:55-61)
const double E2E_BUCKETS[] = {0.01, 0.05, 0.1, 0.25, 0.5, 1.0, 2.5, 5.0, 10.0, 25.0, 50.0, 100.0, 250.0, 500.0};
const double REDIS_BUCKETS[] = {0.001, 0.005, 0.01, 0.025, 0.05, 0.1, 0.25, 0.5, 1.0};            // metrics.rs:66-70

struct Hist {
    std::vector<uint64_t> buckets;
    double sum = 0;
    uint64_t count = 0;
};

struct Family {
    const char* counter_name;
    const char* hist_name;
    const char* counter_help;
    const char* hist_help;
    std::vector<const char*> labels;
    const double* bounds;
    size_t n_bounds;
    std::map<std::vector<std::string>, Hist> series;

    void observe(std::vector<std::string> key, double v) {
        Hist& h = series[std::move(key)];
        if (h.buckets.empty()) h.buckets.assign(n_bounds, 0);
        for (size_t i = 0; i < n_bounds; ++i)
            if (v <= bounds[i]) h.buckets[i]++;
        h.sum += v;
        h.count++;
    }
    std::string labels_of(const std::vector<std::string>& key) const {
        std::string s;
        for (size_t i = 0; i < labels.size(); ++i) {
            if (i) s += ",";
            s += labels[i];
            s += "=\"" + key[i] + "\"";
        }
        return s;
    }
    void render(std::string& out) const {
        char buf[96];
        out += std::string("# HELP ") + counter_name + " " + counter_help + "\n# TYPE " + counter_name + " counter\n";
        for (auto& kv : series) {
            snprintf(buf, sizeof buf, "} %llu\n", (unsigned long long)kv.second.count);
            out += std::string(counter_name) + "{" + labels_of(kv.first) + buf;
        }
        if (!hist_name) return;  // a plain counter family
        out += std::string("# HELP ") + hist_name + " " + hist_help + "\n# TYPE " + hist_name + " histogram\n";
        for (auto& kv : series) {
            std::string l = labels_of(kv.first);
            for (size_t i = 0; i < n_bounds; ++i) {
                snprintf(buf, sizeof buf, ",le=\"%g\"} %llu\n", bounds[i], (unsigned long long)kv.second.buckets[i]);
                out += std::string(hist_name) + "_bucket{" + l + buf;
            }
            snprintf(buf, sizeof buf, ",le=\"+Inf\"} %llu\n", (unsigned long long)kv.second.count);
            out += std::string(hist_name) + "_bucket{" + l + buf;
            snprintf(buf, sizeof buf, "} %.6f\n", kv.second.sum);
            out += std::string(hist_name) + "_sum{" + l + buf;
            snprintf(buf, sizeof buf, "} %llu\n", (unsigned long long)kv.second.count);
            out += std::string(hist_name) + "_count{" + l + buf;
        }
    }
};

struct Metrics {
    std::mutex mu;
    Family task{"task_operations_total", "task_duration_seconds", "Total number of task operations by type and status",
                "Duration of task execution", {"task_name", "operation_type", "status"}, TASK_BUCKETS,
                sizeof TASK_BUCKETS / sizeof *TASK_BUCKETS, {}};
    Family redis{"redis_operations_total", "redis_operation_duration_seconds", "Total number of Redis operations by type",
                 "Duration of Redis operations", {"operation_type", "status"}, REDIS_BUCKETS,
                 sizeof REDIS_BUCKETS / sizeof *REDIS_BUCKETS, {}};
    // the next-gen worker loop's own series (prover/crates/workflow-common/src/metrics.rs:44-80; recorded from
    // prover/crates/workflow/src/lib.rs:613-672)
    Family claims{"task_claims_total", nullptr, "Total number of task claim attempts by stream and result", nullptr,
                  {"task_stream", "result"}, nullptr, 0, {}};
    Family processing{"task_processing_total", "task_processing_end_to_end_seconds",
                      "Total number of task processing attempts by type and status",
                      "End-to-end duration of task processing in the worker loop", {"task_type", "status"}, E2E_BUCKETS,
                      sizeof E2E_BUCKETS / sizeof *E2E_BUCKETS, {}};
    Family retries{"task_retry_attempts_total", nullptr, "Total number of task retry transitions by type", nullptr,
                   {"task_type"}, nullptr, 0, {}};
    Family exhausted{"task_max_retries_exhausted_total", nullptr,
                     "Total number of tasks that stopped retrying and failed permanently", nullptr, {"task_type"}, nullptr, 0, {}};
    void record_task_claim(const char* stream, const char* result) {
        std::lock_guard<std::mutex> g(mu);
        claims.observe({stream, result}, 0);
    }
    void record_task_processing(const std::string& type, const char* status, double s) {
        std::lock_guard<std::mutex> g(mu);
        processing.observe({type, status}, s);
    }
    void record_task_retry_attempt(const std::string& type) {
        std::lock_guard<std::mutex> g(mu);
        retries.observe({type}, 0);
    }
    void record_task_max_retries_exhausted(const std::string& type) {
        std::lock_guard<std::mutex> g(mu);
        exhausted.observe({type}, 0);
    }
    // helpers::record_task_operation / record_task (metrics.rs:298-300,323-335)
    void record_task_operation(const char* task_name, const char* op, const char* status, double s) {
        std::lock_guard<std::mutex> g(mu);
        task.observe({task_name, op, status}, s);
    }
    void record_redis_operation(const char* op, const char* status, double s) {
        std::lock_guard<std::mutex> g(mu);
        redis.observe({op, status}, s);
    }
    std::string exposition() {
        std::lock_guard<std::mutex> g(mu);
        std::string out;
        task.render(out);
        redis.render(out);
        claims.render(out);
        processing.render(out);
        retries.render(out);
        exhausted.render(out);
        return out;
    }
};

void put_le(uint8_t* p, uint64_t v, int n) {
    for (int i = 0; i < n; ++i) p[i] = (uint8_t)(v >> (8 * i));
}
uint64_t get_le(const uint8_t* p, int n) {
    uint64_t v = 0;
    for (int i = 0; i < n; ++i) v |= (uint64_t)p[i] << (8 * i);
    return v;
}

}  // namespace

// ------------------------------------------------------------------------------------------------ in-memory stores ----
struct bx_mem_store {
    std::mutex mu;
    // a value is immutable once set: GET hands out the stored bytes themselves (no copy of an ~80 MB segment), kept alive by a
    // reference until free_value even when the key is overwritten or unlinked meanwhile
    using Bytes = std::shared_ptr<const std::vector<uint8_t>>;
    struct Val {
        Bytes bytes;
        bool expires = false;
        Clock::time_point deadline;
    };
    std::map<std::string, Val> kv;
    std::multimap<const uint8_t*, Bytes> lent;  // values handed out by get and not yet given back
    void sweep_locked() {
        auto now = Clock::now();
        for (auto it = kv.begin(); it != kv.end();) it = (it->second.expires && it->second.deadline < now) ? kv.erase(it) : ++it;
    }
};

static int mem_get(void* user, const char* key, uint8_t** value, size_t* len, char*, size_t) {
    auto* s = (bx_mem_store*)user;
    std::lock_guard<std::mutex> g(s->mu);
    auto it = s->kv.find(key);
    if (it == s->kv.end()) return 1;
    if (it->second.expires && it->second.deadline < Clock::now()) {
        s->kv.erase(it);
        return 1;
    }
    try {
        const bx_mem_store::Bytes& b = it->second.bytes;
        *len = b->size();
        *value = const_cast<uint8_t*>(b->data());  // read-only by contract (bx_hot_store_ops::get)
        s->lent.emplace(b->data(), b);
        return 0;
    } catch (...) {
        return -1;
    }
}
static void mem_free_value(void* user, uint8_t* v) {
    auto* s = (bx_mem_store*)user;
    std::lock_guard<std::mutex> g(s->mu);
    auto it = s->lent.find(v);
    if (it != s->lent.end()) s->lent.erase(it);
}
static int mem_set_ex(void* user, const char* key, const uint8_t* value, size_t len, uint64_t ttl, char*, size_t) {
    auto* s = (bx_mem_store*)user;
    try {  // these tables are a C interface: an allocation failure is a transport error, not an exception
        std::lock_guard<std::mutex> g(s->mu);
        bx_mem_store::Val v;
        auto owned = std::make_shared<std::vector<uint8_t>>(len ? len : 1);  // never an empty vector: data() is the key of `lent`
        owned->resize(len);
        if (len) memcpy(owned->data(), value, len);
        v.bytes = std::move(owned);
        v.expires = ttl != 0;
        if (ttl) v.deadline = Clock::now() + std::chrono::seconds(ttl);
        s->kv[key] = std::move(v);
        return 0;
    } catch (...) {
        return -1;
    }
}
static int mem_unlink(void* user, const char* key, char*, size_t) {
    auto* s = (bx_mem_store*)user;
    std::lock_guard<std::mutex> g(s->mu);
    s->kv.erase(key);
    return 0;
}

struct bx_mem_taskdb {
    std::mutex mu;
    struct Row {
        std::string stream, job, task, def, error, output;
        int32_t max_retries = 0, retries = 0, state = BX_TASK_READY;
        int32_t timeout_secs = INT32_MAX;  // tasks.timeout_secs: how long the task may stay 'running' before requeue_tasks retries it
        uint32_t job_ix = 0;
        std::vector<uint32_t> dependants;  // task_deps rows with this task as pre_task_id: the rows to release when it is done
        int32_t waiting_on = 0;            // prerequisites not yet done
        double created = 0, started = 0, updated = 0;  // seconds since `epoch`
    };
    struct Job {  // the `jobs` row (1_taskdb.sql:52-58) + per-state task counts
        std::string id, error;
        int32_t state = BX_JOB_RUNNING;
        uint64_t counts[5] = {0, 0, 0, 0, 0}, tasks = 0;
    };
    // Every operation is O(log n) in the number of rows: a job of 2^20 segments is 2^21 rows, and every lane of every device
    // claims from and reports to this one table under this one mutex.
    std::deque<Row> rows;  // creation order
    std::vector<Job> jobs;
    std::unordered_map<std::string, uint32_t> row_index, job_index;  // "job\0task" -> row, job -> jobs[]
    // worker type -> ready rows as (job, row): the oldest JOB first, then the oldest task of it — `ORDER BY job_created_at ASC,
    // created_at ASC` (9_request_work.sql:139-141; jobs and rows are numbered in creation order)
    std::map<std::string, std::set<std::pair<uint32_t, uint32_t>>> ready;
    uint64_t counts[5] = {0, 0, 0, 0, 0};
    Clock::time_point epoch = Clock::now();
    double clock_skew = 0;  // bx_mem_taskdb_advance_clock: tests move time instead of sleeping
    double now_s() const { return secs_since(epoch) + clock_skew; }
    std::set<uint32_t> running;  // rows in state 'running': what requeue_tasks scans
    static std::string key(const char* job, const char* task) {
        std::string k(job);
        k.push_back('\0');
        k += task;
        return k;
    }
    Row* find_locked(const char* job, const char* task) {
        auto it = row_index.find(key(job, task));
        return it == row_index.end() ? nullptr : &rows[it->second];
    }
    uint32_t index_of(const Row* r) const { return row_index.at(key(r->job.c_str(), r->task.c_str())); }
    void set_state_locked(Row* r, int32_t s) {
        if (r->state == s) return;
        const uint32_t ix = index_of(r);
        if (s == BX_TASK_READY) ready[r->stream].insert({r->job_ix, ix});  // may throw: before anything else changes
        if (s == BX_TASK_RUNNING) running.insert(ix);                      // (at most one of the two inserts happens per call)
        if (r->state == BX_TASK_READY) ready[r->stream].erase({r->job_ix, ix});
        if (r->state == BX_TASK_RUNNING) running.erase(ix);
        Job& j = jobs[r->job_ix];
        counts[r->state]--, j.counts[r->state]--;
        counts[s]++, j.counts[s]++;
        r->state = s;
    }
    // update_task_failed, 1_taskdb.sql:316-347: ready, running and PENDING rows can fail; the first failure (in time) is the job's error
    int fail_locked(Row* r, const char* error) {
        if (!r || (r->state != BX_TASK_READY && r->state != BX_TASK_RUNNING && r->state != BX_TASK_PENDING)) return 0;
        r->error = error;
        Job& j = jobs[r->job_ix];
        if (j.state != BX_JOB_FAILED) j.error = error;  // both strings assigned before any state changes
        set_state_locked(r, BX_TASK_FAILED);
        r->updated = now_s();
        j.state = BX_JOB_FAILED;
        return 1;
    }
    // the second half of update_task_done (1_taskdb.sql:296-311): every task waiting on `done` loses one prerequisite and becomes
    // ready when that was its last (failed dependants stay failed); a job none of whose tasks is anything but done is done
    void release_dependants_locked(Row* done) {
        for (uint32_t ix : done->dependants) {
            Row& r = rows[ix];
            if (r.state == BX_TASK_FAILED) continue;
            if (r.waiting_on > 0 && --r.waiting_on == 0 && r.state == BX_TASK_PENDING) set_state_locked(&r, BX_TASK_READY);
        }
        Job& j = jobs[done->job_ix];
        if (j.counts[BX_TASK_DONE] == j.tasks) j.state = BX_JOB_DONE;
    }
};

static int tdb_request_work(void* user, const char* stream, bx_ready_task* out, char* errbuf, size_t cap) {
    auto* t = (bx_mem_taskdb*)user;
    try {
        std::lock_guard<std::mutex> g(t->mu);
        auto q = t->ready.find(stream);
        if (q == t->ready.end() || q->second.empty()) return 0;
        bx_mem_taskdb::Row& r = t->rows[q->second.begin()->second];  // ORDER BY job_created_at, created_at LIMIT 1
        if (r.job.size() >= sizeof out->job_id || r.task.size() >= sizeof out->task_id || r.def.size() >= sizeof out->task_def) {
            snprintf(errbuf, cap, "task %s:%s does not fit bx_ready_task", r.job.c_str(), r.task.c_str());
            return -1;
        }
        t->set_state_locked(&r, BX_TASK_RUNNING);
        r.started = t->now_s();
        memset(out, 0, sizeof *out);
        memcpy(out->job_id, r.job.c_str(), r.job.size());
        memcpy(out->task_id, r.task.c_str(), r.task.size());
        memcpy(out->task_def, r.def.c_str(), r.def.size());
        out->max_retries = r.max_retries;
        return 1;
    } catch (const std::exception&) {
        snprintf(errbuf, cap, "request_work: out of memory");
        return -1;
    }
}
static int tdb_done(void* user, const char* job, const char* task, const char* output, char*, size_t) {
    auto* t = (bx_mem_taskdb*)user;
    try {
        std::lock_guard<std::mutex> g(t->mu);
        auto* r = t->find_locked(job, task);
        if (!r || (r->state != BX_TASK_READY && r->state != BX_TASK_RUNNING)) return 0;
        r->output = output ? output : "null";
        t->set_state_locked(r, BX_TASK_DONE);
        r->updated = t->now_s();
        t->release_dependants_locked(r);
        return 1;
    } catch (...) {
        return -1;
    }
}
static int tdb_failed(void* user, const char* job, const char* task, const char* error, char*, size_t) {
    auto* t = (bx_mem_taskdb*)user;
    try {
        std::lock_guard<std::mutex> g(t->mu);
        return t->fail_locked(t->find_locked(job, task), error);
    } catch (...) {
        return -1;
    }
}
// 1_taskdb.sql:361-391
static int retry_locked(bx_mem_taskdb* t, bx_mem_taskdb::Row* r) {
    if (!r || r->state != BX_TASK_RUNNING) return 0;
    t->set_state_locked(r, BX_TASK_READY);
    r->retries += 1;
    r->updated = t->now_s();
    r->error.clear();
    if (r->retries > r->max_retries) {
        t->fail_locked(r, "retry max hit");
        return 0;
    }
    return 1;
}
static int tdb_retry(void* user, const char* job, const char* task, char*, size_t) {
    auto* t = (bx_mem_taskdb*)user;
    try {
        std::lock_guard<std::mutex> g(t->mu);
        return retry_locked(t, t->find_locked(job, task));
    } catch (...) {
        return -1;
    }
}
// taskdb::requeue_tasks (bento/crates/taskdb/src/lib.rs:328-358): up to `limit` tasks that have been 'running' for longer than their
// timeout_secs since GREATEST(started_at, updated_at) go through update_task_retry — back to 'ready' with one more retry, or
// 'failed' ("retry max hit") when that exceeds max_retries.  Returns the number of timed-out tasks found, < 0 on error.
static int64_t tdb_requeue(void* user, int64_t limit, char* errbuf, size_t cap) {
    auto* t = (bx_mem_taskdb*)user;
    try {
        std::lock_guard<std::mutex> g(t->mu);
        const double now = t->now_s();
        std::vector<uint32_t> timed_out;
        for (uint32_t ix : t->running) {
            if (limit >= 0 && (int64_t)timed_out.size() >= limit) break;
            const bx_mem_taskdb::Row& r = t->rows[ix];
            if ((double)r.timeout_secs < now - std::max(r.started, r.updated)) timed_out.push_back(ix);
        }
        for (uint32_t ix : timed_out) (void)retry_locked(t, &t->rows[ix]);
        return (int64_t)timed_out.size();
    } catch (const std::exception&) {
        snprintf(errbuf, cap, "requeue_tasks: out of memory");
        return -1;
    }
}
static int tdb_current_retries(void* user, const char* job, const char* task, int32_t* retries, char*, size_t) {
    auto* t = (bx_mem_taskdb*)user;
    std::lock_guard<std::mutex> g(t->mu);
    try {
        auto* r = t->find_locked(job, task);
        if (!r || r->state != BX_TASK_RUNNING) return 0;
        *retries = r->retries;
        return 1;
    } catch (...) {
        return -1;
    }
}

// ------------------------------------------------------------------------------------------------------------ agent ----
namespace {
struct Lane {
    bx_ctx* ctx = nullptr;
    int32_t device = -1;
    // one prover (= one set of device buffers, several GB at po2 20) per segment size seen, least recently used first;
    // at most cfg.max_shapes are kept
    std::vector<std::pair<uint32_t, bx_prover*>> provers;
    std::atomic<uint64_t> done{0};
    Lane() = default;
    Lane(const Lane&) = delete;
};

// a proved segment on its way through the host half of the task
struct Pending {
    bx_ready_task task;
    Clock::time_point start, claimed;
    std::string job_prefix, segment_key;
    uint64_t seg_index = 0;
    uint32_t po2 = 0;
    bool opaque = false;  // proved through prove_blob: `wire` already holds the receipt bytes to store
    bool lifted = false;  // `seal` holds the stand-in lift of the segment seal kept in child_seal[0] (cfg.lift_po2)
    enum Kind { Prove, Join, HostOnly } kind = Prove;  // HostOnly: the whole task ran in the first half (resolve / finalize stand-ins)
    std::vector<std::string> cleanup_keys;  // a join unlinks its children's receipts once its own is stored (join.rs:94-104)
    std::vector<uint32_t> child_seal[2];    // a join's children ...
    std::future<std::string> child_check[2];  // ... verified on helper threads WHILE the join is proved; collected in the host half
    std::vector<uint32_t> seal;
    size_t words = 0;
    double prove_s = 0;
    std::vector<uint8_t> wire;
};

// one-slot mailbox between a lane and its finisher thread
struct Finisher {
    std::mutex mu;
    std::condition_variable cv;
    Pending* slot = nullptr;
    bool busy = false, quitting = false;
    void post(Pending* p) {
        std::unique_lock<std::mutex> l(mu);
        cv.wait(l, [&] { return !busy; });
        slot = p;
        busy = true;
        cv.notify_all();
    }
    Pending* take() {  // finisher side; nullptr = quit
        std::unique_lock<std::mutex> l(mu);
        cv.wait(l, [&] { return slot || quitting; });
        Pending* p = slot;
        slot = nullptr;
        return p;
    }
    void release() {
        std::lock_guard<std::mutex> l(mu);
        busy = false;
        cv.notify_all();
    }
    bool wait_idle() {  // returns true when something was pending
        std::unique_lock<std::mutex> l(mu);
        bool was = busy;
        cv.wait(l, [&] { return !busy; });
        return was;
    }
    void quit() {
        std::unique_lock<std::mutex> l(mu);
        cv.wait(l, [&] { return !busy; });
        quitting = true;
        cv.notify_all();
    }
};
}  // namespace

namespace {
// "BXSYNRCP" | index u64 | po2 u32 | seal_words u32 | seal u32[] (bx_agent.h)
void receipt_encode(std::vector<uint8_t>& wire, uint64_t index, uint32_t po2, const uint32_t* seal, size_t words) {
    wire.resize(BX_RECEIPT_HEADER_BYTES + 4 * words);
    memcpy(wire.data(), BX_RECEIPT_MAGIC, 8);
    put_le(wire.data() + 8, index, 8);
    put_le(wire.data() + 16, po2, 4);
    put_le(wire.data() + 20, words, 4);
    for (size_t i = 0; i < words; ++i) put_le(wire.data() + BX_RECEIPT_HEADER_BYTES + 4 * i, seal[i], 4);
}
bool receipt_decode(const uint8_t* p, size_t n, uint64_t* index, uint32_t* po2, std::vector<uint32_t>* seal) {
    if (!p || n < BX_RECEIPT_HEADER_BYTES || memcmp(p, BX_RECEIPT_MAGIC, 8) != 0) return false;
    const uint64_t words = get_le(p + 20, 4);
    if (n != BX_RECEIPT_HEADER_BYTES + 4 * words) return false;
    if (index) *index = get_le(p + 8, 8);
    if (po2) *po2 = (uint32_t)get_le(p + 16, 4);
    seal->resize(words);
    for (size_t i = 0; i < words; ++i) (*seal)[i] = (uint32_t)get_le(p + BX_RECEIPT_HEADER_BYTES + 4 * i, 4);
    return true;
}
}  // namespace

extern "C" uint64_t bx_join_seed(const uint32_t* left, size_t nl, const uint32_t* right, size_t nr) {
    uint64_t h = 0xCBF29CE484222325ull;  // FNV-1a, 64 bit
    auto eat = [&h](const uint32_t* w, size_t n) {
        for (size_t i = 0; i < n; ++i)
            for (int b = 0; b < 4; ++b) {
                h ^= (uint8_t)(w[i] >> (8 * b));
                h *= 0x100000001B3ull;
            }
    };
    if (left) eat(left, nl);
    if (right) eat(right, nr);
    uint64_t z = h + 0x9E3779B97F4A7C15ull;  // splitmix64 finaliser
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

struct bx_agent;
struct bx_agent {
    bx_agent_config cfg;
    bx_hot_store_ops store;
    bx_taskdb_ops taskdb;
    bx_segment_prover_ops prover;
    bool hip = false;
    std::vector<std::unique_ptr<Lane>> lanes;
    Metrics metrics;
    std::atomic<int> stop{0};
    std::vector<std::string> streams;  // task_stream first, then also_streams: a lane claims from the first that has work
    std::mutex create_mu;  // device buffer allocation of a new shape is serialised across lanes
    // VerifierContext of the HIP prover's agent (`verifier_ctx`, lib.rs:241): the control ID of every buffer set a lane creates.
    // Finishers verify under the shared lock, a lane adding a new shape takes it exclusively.
    bx_verifier_ctx* vctx = nullptr;
    std::shared_mutex vctx_mu;

    // ---- default prover ops: the HIP segment prover, one ctx per lane ----
    // Errors of bx_init / bx_prover_create are kept per lane (the text is owned by the failing call) and surface as the
    // task's error instead of a bare "no prover".
    const char* hip_prover_for(uint32_t lane_idx, uint32_t po2, bx_prover** out, std::string* err) {
        Lane& lane = *lanes[lane_idx];
        if (po2 < cfg.po2_min || po2 > cfg.po2_max) {
            *err = "segment po2 " + std::to_string(po2) + " is outside the sizes this agent accepts [" + std::to_string(cfg.po2_min) + ", " +
                   std::to_string(cfg.po2_max) + "]";
            return err->c_str();
        }
        for (size_t i = 0; i < lane.provers.size(); ++i)
            if (lane.provers[i].first == po2) {
                auto hit = lane.provers[i];
                lane.provers.erase(lane.provers.begin() + (long)i);
                lane.provers.push_back(hit);  // most recently used last
                *out = hit.second;
                return nullptr;
            }
        std::lock_guard<std::mutex> g(create_mu);
        if (!lane.ctx) {
            if (const char* e = bx_init(lane.device, &lane.ctx)) {
                *err = std::string("bx_init(device ") + std::to_string(lane.device) + "): " + e;
                return err->c_str();
            }
        }
        while (lane.provers.size() >= cfg.max_shapes) {  // evict before allocating the new buffer set
            (void)bx_prover_destroy(lane.provers.front().second);
            lane.provers.erase(lane.provers.begin());
        }
        bx_segment_params shape{po2, cfg.w_code, cfg.w_data, cfg.w_accum, cfg.cons_terms, cfg.cons_degree};
        bx_prover* p = nullptr;
        if (const char* e = bx_prover_create(lane.ctx, &shape, &p)) {
            *err = std::string("bx_prover_create(po2 ") + std::to_string(po2) + "): " + e;
            return err->c_str();
        }
        // the shape's control ID, computed by the device that will prove it, must be the one the circuit publishes; it then
        // goes into the agent's verifier context
        uint32_t id[8];
        const char* ce = bx_prover_control_id(p, id);
        bool known = false;  // a buffer set re-created after an eviction: its ID was validated when it was first made
        if (!ce) {
            std::shared_lock<std::shared_mutex> r(vctx_mu);
            known = bx::verifier_ctx_contains(vctx, po2, id);
        }
        if (!ce && !known) ce = bx_synthetic_circuit()->check_code(nullptr, &shape, id);  // may be seconds of host work off the table
        if (!ce && !known) {
            std::unique_lock<std::shared_mutex> w(vctx_mu);
            ce = bx_verifier_ctx_add_control_id(vctx, po2, id);
        }
        if (ce) {
            *err = std::string("control ID of the new buffer set (po2 ") + std::to_string(po2) + "): " + ce;
            (void)bx_prover_destroy(p);
            return err->c_str();
        }
        lane.provers.emplace_back(po2, p);
        *out = p;
        return nullptr;
    }
    static inline thread_local std::string hip_err;
    static size_t hip_seal_words(void* user, uint32_t lane, uint32_t po2) {
        auto* a = (bx_agent*)user;
        bx_prover* p = nullptr;
        if (a->hip_prover_for(lane, po2, &p, &hip_err)) return 0;
        hip_err.clear();
        return bx_prover_seal_words(p);
    }
    static const char* hip_prove(void* user, uint32_t lane, uint32_t po2, const uint8_t* segment, size_t len, uint32_t* seal, size_t cap,
                                 size_t* words) {
        auto* a = (bx_agent*)user;
        bx_prover* p = nullptr;
        if (const char* e = a->hip_prover_for(lane, po2, &p, &hip_err)) return e;
        return bx_prove_segment_bytes(p, segment, len, seal, cap, words);  // the stored blob, as it is
    }

    // ---- redis.rs helpers with their metrics ----
    // returns "" on success, otherwise the error text
    // a value as the store handed it out; given back through free_value when it goes out of scope (a segment is ~80 MB:
    // it is read where the store put it, not copied into a container of ours)
    struct StoreValue {
        const bx_hot_store_ops* ops = nullptr;
        uint8_t* p = nullptr;
        size_t n = 0;
        StoreValue() = default;
        StoreValue(const StoreValue&) = delete;
        StoreValue& operator=(const StoreValue&) = delete;
        ~StoreValue() {
            if (p && ops && ops->free_value) ops->free_value(ops->user, p);
        }
        const uint8_t* data() const { return p; }
        size_t size() const { return n; }
    };
    // A claimed task on its way to the lane, with its segment already fetched when it is a Prove task and cfg.prefetch is on.
    struct Fetched {
        bx_ready_task task;
        int rc = 0;  // request_work: 1 claimed, 0 nothing ready, < 0 error (eb)
        char eb[256] = {0};
        Clock::time_point claimed;
        bool fetched = false;  // a GET of the segment was attempted: blob / blob_err hold its outcome
        uint64_t index = 0;
        StoreValue blob;
        std::string blob_err;
    };
    // one-slot mailbox between a lane's fetcher thread and the lane (cfg.prefetch)
    struct Prefetcher {
        std::mutex mu;
        std::condition_variable cv;
        std::unique_ptr<Fetched> slot;
        bool quitting = false;
        bool dead = false;  // the fetcher thread gave up (it could not even allocate a mailbox entry): take() must not wait for it
        void fetcher_died() {
            std::lock_guard<std::mutex> l(mu);
            dead = true;
            cv.notify_all();
        }
        void put(std::unique_ptr<Fetched> w) {
            std::lock_guard<std::mutex> l(mu);
            slot = std::move(w);
            cv.notify_all();
        }
        bool wait_empty() {  // fetcher side: false = quit
            std::unique_lock<std::mutex> l(mu);
            cv.wait(l, [&] { return !slot || quitting; });
            return !quitting;
        }
        std::unique_ptr<Fetched> take(std::atomic<int>& stop_flag) {  // lane side; nullptr = stop was requested while waiting
            std::unique_lock<std::mutex> l(mu);
            while (!slot) {
                if (dead || stop_flag.load(std::memory_order_relaxed)) return nullptr;
                // system_clock on purpose: a steady-clock wait is pthread_cond_clockwait, which the ThreadSanitizer of this
                // toolchain does not intercept (it then reports the re-lock inside the wait as a double lock)
                cv.wait_until(l, std::chrono::system_clock::now() + std::chrono::milliseconds(20));
            }
            auto w = std::move(slot);
            cv.notify_all();
            return w;
        }
        std::unique_ptr<Fetched> quit() {  // returns what was fetched and never taken (a claimed task must not be lost)
            std::lock_guard<std::mutex> l(mu);
            quitting = true;
            cv.notify_all();
            return std::move(slot);
        }
    };
    // request_work over the lane's streams (its own first) + the claim metrics (lib.rs:613-623)
    int claim_once(bx_ready_task* task, char* eb, size_t cap) {
        int rc = 0;
        for (const std::string& st : streams) {
            rc = taskdb.request_work(taskdb.user, st.c_str(), task, eb, cap);
            metrics.record_task_claim(st.c_str(), rc < 0 ? "error" : rc == 0 ? "empty" : "claimed");
            if (rc != 0) break;
        }
        return rc;
    }
    // {"Prove":{"index":n}} -> n
    static bool prove_index_of(const bx_ready_task& task, uint64_t* index) {
        JVal def;
        if (!parse_json(task.task_def, &def) || def.kind != JVal::Obj || def.obj.size() != 1 || def.obj[0].first != "Prove") return false;
        const JVal& body = def.obj[0].second;
        const JVal* f = body.kind == JVal::Obj ? body.find("index") : nullptr;
        if (!f || f->kind != JVal::Num || !f->is_uint) return false;
        *index = f->u;
        return true;
    }

    std::string store_get(const std::string& key, StoreValue* out) {
        auto t0 = Clock::now();
        char eb[256] = {0};
        int rc = store.get(store.user, key.c_str(), &out->p, &out->n, eb, sizeof eb);
        std::string err;
        if (rc == 0) {
            out->ops = &store;
        } else {
            out->p = nullptr, out->n = 0;
            if (rc == 1) err = "Key not found (nil response): " + key;  // redis.rs:51-55
            else err = eb[0] ? eb : "hot store get failed";
        }
        metrics.record_redis_operation("get", rc == 0 ? "success" : "error", secs_since(t0));
        return err;
    }
    std::string store_set(const std::string& key, const std::vector<uint8_t>& val, uint64_t ttl) {
        auto t0 = Clock::now();
        char eb[256] = {0};
        int rc = store.set_ex(store.user, key.c_str(), val.data(), val.size(), ttl, eb, sizeof eb);
        metrics.record_redis_operation(ttl ? "set_ex" : "set", rc == 0 ? "success" : "error", secs_since(t0));
        return rc == 0 ? "" : (eb[0] ? eb : "hot store set failed");
    }
    std::string store_unlink(const std::string& key) {
        auto t0 = Clock::now();
        char eb[256] = {0};
        int rc = store.unlink(store.user, key.c_str(), eb, sizeof eb);
        metrics.record_redis_operation("unlink", rc == 0 ? "success" : "error", secs_since(t0));
        return rc == 0 ? "" : (eb[0] ? eb : "hot store unlink failed");
    }

    // tasks::prove::prover (prove.rs:18-135) in two halves so the host-side half of one segment (verify, store, cleanup,
    // update_task_done) can overlap the device half of the lane's next segment.  Each returns "" or the error chain
    // ("outer: inner", anyhow's `{:#}` form).
    //   first half: fetch -> deserialize -> prove_segment
    std::string prove_stage(uint32_t lane_idx, const bx_ready_task& task, uint64_t index, Pending* out, Fetched* pre = nullptr) {
        out->start = Clock::now();
        out->task = task;
        out->job_prefix = std::string("job:") + task.job_id;
        out->segment_key = out->job_prefix + ":segments:" + std::to_string(index);  // SEGMENTS_PATH, tasks/mod.rs:25
        StoreValue fetched_here;
        const bool use_pre = pre && pre->fetched && pre->index == index;  // the lane's fetcher already did the GET (cfg.prefetch)
        StoreValue& blob = use_pre ? pre->blob : fetched_here;
        std::string e = use_pre ? pre->blob_err : store_get(out->segment_key, &fetched_here);
        if (!e.empty()) return "segment data not found for segment key: " + out->segment_key + ": " + e;
        out->opaque = prover.prove_blob != nullptr;
        if (out->opaque) {  // a real prover: bytes in, bytes out (prove.rs:36-109 happens inside the callback)
            auto prove_start = Clock::now();
            uint8_t* rec = nullptr;
            size_t rec_len = 0;
            if (const char* pe = prover.prove_blob(prover.user, lane_idx, blob.data(), blob.size(), &rec, &rec_len)) return pe;
            out->wire.assign(rec, rec + rec_len);
            if (prover.free_blob) prover.free_blob(prover.user, rec);
            out->prove_s = secs_since(prove_start);
            metrics.record_task_operation("prove", "prove_segment", "success", out->prove_s);
            return "";
        }
        // only the routing fields are read here; the prover gets the stored bytes
        if (const char* de = bx_segment_decode(blob.data(), blob.size(), &out->seg_index, &out->po2, nullptr)) return de;

        auto prove_start = Clock::now();
        if (!prover.prove_segment) return "[BENTO-PROVE-002] Missing prover from prove task";
        size_t cap = prover.seal_words(prover.user, lane_idx, out->po2);
        if (cap == 0) {
            if (hip && !hip_err.empty()) return "prove_segment: " + hip_err;
            return "prove_segment: no prover for a segment of po2 " + std::to_string(out->po2);
        }
        if (out->seal.size() < cap) out->seal.resize(cap);
        out->words = 0;
        if (const char* pe = prover.prove_segment(prover.user, lane_idx, out->po2, blob.data(), blob.size(), out->seal.data(), cap, &out->words))
            return pe;
        out->prove_s = secs_since(prove_start);
        metrics.record_task_operation("prove", "prove_segment", "success", out->prove_s);
        out->lifted = false;
        if (cfg.lift_po2) {
            // the `lift` leg (prove.rs:60-113), as a labelled stand-in: the segment seal is set aside — its verification (BENTO-PROVE-004)
            // runs on a helper thread beside the second proof, as for a join's children — and a synthetic proof of 2^lift_po2 cycles
            // seeded by it takes its place
            out->child_seal[0].assign(out->seal.begin(), out->seal.begin() + (long)out->words);
            if (!cfg.no_verify)
                out->child_check[0] = std::async(std::launch::async, [this, out]() -> std::string {
                    const char* ve = verify_seal(out->child_seal[0].data(), out->child_seal[0].size());
                    return ve ? std::string("[BENTO-PROVE-004] Failed to verify segment receipt integrity: ") + ve : std::string();
                });
            uint8_t wire[BX_SEGMENT_WIRE_BYTES];
            bx_segment_encode(out->seg_index, cfg.lift_po2, bx_join_seed(out->child_seal[0].data(), out->child_seal[0].size(), nullptr, 0), wire);
            auto lift_start = Clock::now();
            size_t cap2 = prover.seal_words(prover.user, lane_idx, cfg.lift_po2);
            std::string err;
            if (cap2 == 0) err = hip && !hip_err.empty() ? "lift: " + hip_err : "lift: no prover for a stand-in lift of po2 " + std::to_string(cfg.lift_po2);
            if (err.empty()) {
                if (out->seal.size() < cap2) out->seal.resize(cap2);
                out->words = 0;
                if (const char* pe = prover.prove_segment(prover.user, lane_idx, cfg.lift_po2, wire, sizeof wire, out->seal.data(), cap2, &out->words)) err = pe;
            }
            if (!err.empty()) {
                drain_children(out);
                return err;
            }
            out->po2 = cfg.lift_po2;
            out->lifted = true;
            metrics.record_task_operation("prove", "lift", "success", secs_since(lift_start));
        }
        return "";
    }
    //   second half: verify -> store under the recursion-receipt key -> unlink the segment
    std::string finish_stage(Pending* p) {
        std::string output_key;
        if (p->opaque) {
            // the callback returned the serialized lifted receipt: reference key (RECUR_RECEIPT_PATH, tasks/mod.rs:23)
            output_key = p->job_prefix + ":" BX_RECUR_RECEIPT_PATH ":" + p->task.task_id;
        } else {
            if (!cfg.no_verify) {  // segment_receipt.verify_integrity_with_context (prove.rs:53-55)
                // the HIP prover's agent checks the code root against its own context; an injected prover's seals against the
                // circuit's published IDs (check_code)
                const char* ve = verify_seal(p->seal.data(), p->words);
                std::string own = ve ? ve : "";
                if (p->lifted) {  // the segment seal's verdict first (it was checked beside the lift), then the lifted receipt's
                    std::string seg = p->child_check[0].valid() ? p->child_check[0].get() : std::string();
                    if (!seg.empty()) return seg;
                    if (!own.empty()) return "[BENTO-PROVE-010] Failed to verify lift receipt integrity: " + own;
                } else if (!own.empty()) {
                    return "[BENTO-PROVE-004] Failed to verify segment receipt integrity: " + own;
                }
            }
            // a synthetic seal is not a lifted receipt: it never goes under the key Join workers read
            output_key = p->job_prefix + ":" BX_SYNTHETIC_RECEIPT_PATH ":" + p->task.task_id;
            receipt_encode(p->wire, p->seg_index, p->po2, p->seal.data(), p->words);
        }
        metrics.record_task_operation("prove", "prove_segment", "success", p->prove_s);  // helpers::record_task, prove.rs:57

        std::string e = store_set(output_key, p->wire, cfg.redis_ttl);
        if (!e.empty()) return "Failed to set receipt key with expiry: " + e;

        e = store_unlink(p->segment_key);
        if (!e.empty()) return "Failed to delete segment key: " + e;
        metrics.record_task_operation("prove", "complete", "success", secs_since(p->start));
        return "";
    }

    // verify_integrity_with_context against the agent's context (HIP prover) or the circuit's published IDs (injected prover)
    // The HIP agent's context holds the control ID of every buffer set its lanes created (each checked against the circuit's
    // published IDs when it was added); a receipt of a size no lane has created yet — a child receipt another agent proved, the
    // root receipt an aux agent finalizes — is checked against the published IDs themselves (check_code).
    const char* verify_seal(const uint32_t* seal, size_t words) {
        std::shared_lock<std::shared_mutex> r(vctx_mu);
        const bool own = hip && vctx && words > 0 && bx_verifier_ctx_count(vctx, seal[0]) > 0;
        return bx_verify_segment_with_context(seal, words, nullptr, own ? vctx : nullptr);
    }
    // a stored synthetic receipt: GET + deserialize (+ verify when a code is given)
    std::string load_receipt(const std::string& key, const char* which, const char* code_deser, const char* code_verify, uint64_t* index, uint32_t* po2,
                             std::vector<uint32_t>* seal) {
        StoreValue blob;
        std::string e = store_get(key, &blob);
        if (!e.empty()) return e;
        if (!receipt_decode(blob.data(), blob.size(), index, po2, seal)) return std::string(code_deser) + " Failed to deserialize " + which + " receipt";
        if (code_verify && !cfg.no_verify)
            if (const char* ve = verify_seal(seal->data(), seal->size())) return std::string(code_verify) + " Failed to verify " + which + " receipt integrity: " + ve;
        return "";
    }

    // tasks::join::join (join.rs:18-113) with a STAND-IN for `prover.join(&left, &right)`: one synthetic segment of 2^join_po2
    // cycles seeded by the two children's seals (bx_agent.h, "Stand-ins for the recursion tasks").  First half: fetch, deserialize,
    // prove.  The reference verifies both children before joining (join.rs:44-49); here those two checks (5-15 ms of CPU each) start
    // on helper threads before the proof is launched and run beside it — the seed needs the children's bytes, not their validity —
    // and the second half collects their verdicts: a join whose child does not verify fails with the same code and stores nothing.
    // On the critical path of a job's join tail (log2 K levels, each a join whose parent is released by update_task_done) that
    // leaves one proof + one verification per level instead of one proof + three verifications.
    static void drain_children(Pending* p) {
        for (auto& f : p->child_check)
            if (f.valid()) (void)f.get();
    }
    std::string join_stage(uint32_t lane_idx, const bx_ready_task& task, uint64_t idx, uint64_t left, uint64_t right, Pending* out) {
        out->start = Clock::now();
        out->task = task;
        out->kind = Pending::Join;
        out->opaque = false;
        out->job_prefix = std::string("job:") + task.job_id;
        const std::string prefix = out->job_prefix + ":" BX_SYNTHETIC_RECEIPT_PATH ":";
        const std::string lk = prefix + std::to_string(left), rk = prefix + std::to_string(right);
        std::vector<uint32_t>&ls = out->child_seal[0], &rs = out->child_seal[1];
        std::string e = load_receipt(lk, "left", "[BENTO-JOIN-001]", nullptr, nullptr, nullptr, &ls);
        if (e.empty()) e = load_receipt(rk, "right", "[BENTO-JOIN-002]", nullptr, nullptr, nullptr, &rs);
        if (!e.empty()) return e.rfind("[BENTO-JOIN", 0) == 0 ? e : "failed to get receipts for keys: " + lk + ", " + rk + ": " + e;
        out->cleanup_keys = {lk, rk};
        out->seg_index = idx;
        out->po2 = cfg.join_po2;
        uint8_t wire[BX_SEGMENT_WIRE_BYTES];
        bx_segment_encode(idx, cfg.join_po2, bx_join_seed(ls.data(), ls.size(), rs.data(), rs.size()), wire);
        auto join_start = Clock::now();
        if (!prover.prove_segment) return "Missing prover from join task";
        size_t cap = prover.seal_words(prover.user, lane_idx, out->po2);
        if (cap == 0) {
            if (hip && !hip_err.empty()) return "join_receipts: " + hip_err;
            return "join_receipts: no prover for a stand-in join of po2 " + std::to_string(out->po2);
        }
        if (out->seal.size() < cap) out->seal.resize(cap);
        out->words = 0;
        if (!cfg.no_verify) {  // the children's checks run beside the proof; child_seal[] stays untouched until they are collected
            static const char* const codes[2] = {"[BENTO-JOIN-003] Failed to verify left receipt integrity: ",
                                                 "[BENTO-JOIN-004] Failed to verify right receipt integrity: "};
            for (int k = 0; k < 2; ++k)
                out->child_check[k] = std::async(std::launch::async, [this, out, k]() -> std::string {
                    const char* ve = verify_seal(out->child_seal[k].data(), out->child_seal[k].size());
                    return ve ? std::string(codes[k]) + ve : std::string();
                });
        }
        if (const char* pe = prover.prove_segment(prover.user, lane_idx, out->po2, wire, sizeof wire, out->seal.data(), cap, &out->words)) {
            metrics.record_task_operation("join", "join_receipts", "error", secs_since(join_start));
            std::string msg = pe;
            drain_children(out);  // the slot is reused by the lane's next task: nothing may still be reading it
            return msg;
        }
        out->prove_s = secs_since(join_start);
        metrics.record_task_operation("join", "join_receipts", "success", out->prove_s);
        return "";
    }
    //   second half: verify the children and the joined receipt, store it, unlink the children
    std::string join_finish(Pending* p) {
        if (!cfg.no_verify) {
            // the joined receipt is checked here while the children's checks (started before the proof) finish; both verdicts first,
            // so that no helper is left running when this slot is handed back
            std::string own;
            if (const char* ve = verify_seal(p->seal.data(), p->words)) own = std::string("[BENTO-JOIN-006] Failed to verify join receipt integrity: ") + ve;
            std::string left = p->child_check[0].valid() ? p->child_check[0].get() : std::string();
            std::string right = p->child_check[1].valid() ? p->child_check[1].get() : std::string();
            if (!left.empty()) return left;
            if (!right.empty()) return right;
            if (!own.empty()) return own;
        }
        receipt_encode(p->wire, p->seg_index, p->po2, p->seal.data(), p->words);
        std::string e = store_set(p->job_prefix + ":" BX_SYNTHETIC_RECEIPT_PATH ":" + std::to_string(p->seg_index), p->wire, cfg.redis_ttl);
        if (!e.empty()) return "Failed to store joined receipt: " + e;
        for (auto& k : p->cleanup_keys) {
            e = store_unlink(k);
            if (!e.empty()) return "Failed to delete join receipt keys: " + e;
        }
        metrics.record_task_operation("join", "complete", "success", secs_since(p->start));
        return "";
    }
    // tasks::resolve::resolver without assumptions (resolve.rs:18-180): the root receipt is read, checked and written back
    std::string resolve_stage(const bx_ready_task& task, uint64_t max_idx, Pending* out) {
        out->start = Clock::now();
        out->task = task;
        out->kind = Pending::HostOnly;
        out->opaque = false;
        const std::string key = std::string("job:") + task.job_id + ":" BX_SYNTHETIC_RECEIPT_PATH ":" + std::to_string(max_idx);
        uint64_t index = 0;
        uint32_t po2 = 0;
        std::vector<uint32_t> seal;
        std::string e = load_receipt(key, "root", "[BENTO-RESOLVE-001]", "[BENTO-RESOLVE-001]", &index, &po2, &seal);
        if (!e.empty()) return "segment data not found for root receipt key: " + key + ": " + e;
        receipt_encode(out->wire, index, po2, seal.data(), seal.size());
        e = store_set(key, out->wire, cfg.redis_ttl);
        if (!e.empty()) return "Failed to set root receipt key with expiry: " + e;
        metrics.record_task_operation("resolve", "complete", "success", secs_since(out->start));
        return "";
    }
    // tasks::finalize::finalize (finalize.rs:21-95): the root receipt is verified and becomes the job's rollup receipt (the reference
    // uploads receipts/stark/{job}.bincode to S3; here the hot store holds receipts/stark/{job}.synthetic)
    std::string finalize_stage(const bx_ready_task& task, uint64_t max_idx, Pending* out) {
        out->start = Clock::now();
        out->task = task;
        out->kind = Pending::HostOnly;
        out->opaque = false;
        const std::string key = std::string("job:") + task.job_id + ":" BX_SYNTHETIC_RECEIPT_PATH ":" + std::to_string(max_idx);
        uint64_t index = 0;
        uint32_t po2 = 0;
        std::vector<uint32_t> seal;
        std::string e = load_receipt(key, "root", "[BENTO-FINALIZE-001]", "[BENTO-FINALIZE-001]", &index, &po2, &seal);
        if (!e.empty()) return "failed to get the root receipt key: " + key + ": " + e;
        receipt_encode(out->wire, index, po2, seal.data(), seal.size());
        e = store_set(std::string(BX_SYNTHETIC_ROLLUP_PREFIX) + task.job_id + BX_SYNTHETIC_ROLLUP_SUFFIX, out->wire, cfg.redis_ttl);
        if (!e.empty()) return "Failed to upload final receipt to obj store: " + e;
        metrics.record_task_operation("finalize", "complete", "success", secs_since(out->start));
        return "";
    }

    // Agent::process_work (lib.rs:445-530), first half: TaskType dispatch + the device half of the prove task.
    std::string dispatch(uint32_t lane, const bx_ready_task& task, Pending* out, Fetched* pre = nullptr) {
        drain_children(out);  // a slot that was abandoned on an exception may still have helpers reading it
        JVal def;
        std::string bad = std::string("Invalid task_def: ") + task.job_id + ":" + task.task_id;
        if (!parse_json(task.task_def, &def) || def.kind != JVal::Obj || def.obj.size() != 1) return bad;
        const std::string& variant = def.obj[0].first;
        const JVal& body = def.obj[0].second;
        auto uint_field = [&](const char* name, uint64_t* v) {
            const JVal* f = body.kind == JVal::Obj ? body.find(name) : nullptr;
            if (!f || f->kind != JVal::Num || !f->is_uint) return false;
            *v = f->u;
            return true;
        };
        if (variant == "Prove") {
            uint64_t idx = 0;
            if (!uint_field("index", &idx)) return bad;
            out->kind = Pending::Prove;
            out->cleanup_keys.clear();
            std::string e = prove_stage(lane, task, idx, out, pre);
            if (!e.empty()) return "[BENTO-WF-115] Prove failed: " + e;
            return "";
        }
        // the recursion tasks of a planned job: served by their stand-ins in synthetic mode (bx_agent.h), refused otherwise
        if (cfg.synthetic && !prover.prove_blob && (variant == "Join" || variant == "Resolve" || variant == "Finalize")) {
            uint64_t idx = 0, left = 0, right = 0, max_idx = 0;
            if (variant == "Join") {
                if (!uint_field("idx", &idx) || !uint_field("left", &left) || !uint_field("right", &right)) return bad;
                std::string e = join_stage(lane, task, idx, left, right, out);
                return e.empty() ? "" : "[BENTO-WF-119] Join failed: " + e;
            }
            if (!uint_field("max_idx", &max_idx)) return bad;
            if (variant == "Resolve") {
                std::string e = resolve_stage(task, max_idx, out);
                return e.empty() ? "" : "[BENTO-WF-123] Resolve failed: " + e;
            }
            std::string e = finalize_stage(task, max_idx, out);
            return e.empty() ? "" : "[BENTO-WF-125] Finalize failed: " + e;
        }
        static const char* others[] = {"Executor", "Join", "Resolve", "Finalize", "Snark", "Keccak", "Union"};
        for (const char* o : others)
            if (variant == o) return "task type " + variant + " reached a prove-stream agent (not served: DESIGN.md §2)";
        return bad;
    }
    // second half: host half of the prove task + update_task_done.  `fatal` is set when the task-db call itself failed.
    std::string complete(Pending* p, bool* fatal) {
        if (p->kind == Pending::Join) {
            std::string e = join_finish(p);
            if (!e.empty()) return "[BENTO-WF-119] Join failed: " + e;
        } else if (p->kind == Pending::Prove) {
            std::string e = finish_stage(p);
            if (!e.empty()) return "[BENTO-WF-115] Prove failed: " + e;
        }
        char eb[256] = {0};
        int rc = taskdb.update_task_done(taskdb.user, p->task.job_id, p->task.task_id, "null", eb, sizeof eb);  // prover returns ()
        if (rc < 0) {
            *fatal = true;
            return std::string("[BENTO-WF-133] Failed to report task done: ") + eb;
        }
        return "";
    }

    // Agent::task_type_label (prover/crates/workflow/src/lib.rs:299-310) + TaskType::to_job_type_str (workflow-common lib.rs:176-187)
    static std::string task_type_label(const bx_ready_task& task) {
        JVal def;
        if (!parse_json(task.task_def, &def)) return "invalid_task";
        std::string variant;
        if (def.kind == JVal::Obj && def.obj.size() == 1) variant = def.obj[0].first;
        else if (def.kind == JVal::Str) variant = def.s;  // unit variant: "Finalize"
        static const char* names[][2] = {{"Executor", "executor"}, {"Prove", "prove-lift"}, {"Join", "join"},   {"Resolve", "resolve"},
                                         {"Finalize", "finalize"}, {"Snark", "snark"},      {"Keccak", "keccak"}, {"Union", "union"}};
        for (auto& n : names)
            if (variant == n[0]) return n[1];
        return "invalid_task";
    }

    // the error arm of poll_work (lib.rs:381-436); returns "" or a fatal task-db error
    // cut at a character boundary: a multi-byte UTF-8 sequence split in half makes the failure report itself invalid JSON
    static void truncate_utf8(std::string& s, size_t cap) {
        if (s.size() <= cap) return;
        size_t n = cap;
        while (n > 0 && ((unsigned char)s[n] & 0xC0) == 0x80) --n;  // s[n] is a continuation byte: back up to its lead byte
        s.resize(n);
    }
    std::string handle_failure(const bx_ready_task& task, std::string err) {
        char eb[256] = {0};
        const std::string type = task_type_label(task);
        if (task.max_retries > 0) {
            int32_t cur = 0;
            int found = taskdb.current_retries(taskdb.user, task.job_id, task.task_id, &cur, eb, sizeof eb);
            if (found < 0) return std::string("[BENTO-WF-109] Failed to read current retries: ") + eb;
            if (found == 1 && cur + 1 > task.max_retries) {
                metrics.record_task_max_retries_exhausted(type);
                truncate_utf8(err, 1024);
                std::string final_err = err.empty() ? "retry max hit" : "retry max hit: " + err;
                if (taskdb.update_task_failed(taskdb.user, task.job_id, task.task_id, final_err.c_str(), eb, sizeof eb) < 0)
                    return std::string("[BENTO-WF-110] Failed to report task failure: ") + eb;
                return "";
            }
            int rc = taskdb.update_task_retry(taskdb.user, task.job_id, task.task_id, eb, sizeof eb);
            if (rc < 0) return std::string("[BENTO-WF-111] Failed to update task retries: ") + eb;
            // update_task_retry's bool (lib.rs:664-669): requeued, or the task db itself stopped retrying
            if (rc > 0) metrics.record_task_retry_attempt(type);
            else metrics.record_task_max_retries_exhausted(type);
        } else {
            metrics.record_task_max_retries_exhausted(type);
            truncate_utf8(err, 1024);
            if (taskdb.update_task_failed(taskdb.user, task.job_id, task.task_id, err.c_str(), eb, sizeof eb) < 0)
                return std::string("[BENTO-WF-112] Failed to report task failure: ") + eb;
        }
        return "";
    }

    // One lane of poll_work's main loop (lib.rs:369-438).  The lane thread claims and runs the device half; its finisher
    // thread runs the host half of the previous segment meanwhile (one segment of look-ahead per lane, two seal buffers).
    void lane_loop(uint32_t lane, int64_t max_idle_polls, std::atomic<uint64_t>* done, std::string* fatal_out) {
        Finisher fin;
        std::mutex fatal_mu;
        auto set_fatal = [&](const std::string& m) {
            std::lock_guard<std::mutex> g(fatal_mu);
            if (fatal_out->empty()) *fatal_out = m;
            stop.store(1);
        };
        // neither thread lets a C++ exception escape (bad_alloc from a seal / string / vector growth would otherwise
        // reach std::terminate): it becomes the task's error, or the loop's fatal error when even that cannot be recorded
        std::thread finisher([&] {
            while (Pending* p = fin.take()) {
                try {
                    bool fatal = false;
                    std::string err;
                    try {
                        err = complete(p, &fatal);
                    } catch (const std::exception& e) {
                        err = std::string("[BENTO-WF-115] Prove failed: exception in the host half: ") + e.what();
                    }
                    // end to end = both halves of process_work; with the host half of segment k overlapping the device half
                    // of k+1 this is the task's latency, not the lane's occupancy
                    metrics.record_task_processing(task_type_label(p->task), err.empty() ? "success" : "error", secs_since(p->claimed));
                    if (err.empty()) {
                        done->fetch_add(1);
                        lanes[lane]->done.fetch_add(1);
                    } else if (fatal) {
                        set_fatal(err);
                    } else {
                        std::string f = handle_failure(p->task, err);
                        if (!f.empty()) set_fatal(f);
                    }
                } catch (...) {
                    try {
                        set_fatal("lane finisher: exception while recording a task result");
                    } catch (...) {
                        stop.store(1);
                    }
                }
                fin.release();
            }
        });
        // cfg.prefetch: a fetcher thread claims the lane's NEXT task and fetches its segment while the lane proves the current one.
        // With a store behind a network (an ~80 MB GET over the REST worker protocol) the lane's GPU share would otherwise idle for
        // the length of every GET; with the in-memory store there is nothing to hide, which is why it is opt-in: the price is one
        // task claimed ahead per lane (a tail imbalance of at most one proof at the end of a batch).
        Prefetcher pf;
        std::thread fetcher;
        if (cfg.prefetch) {
            fetcher = std::thread([&] {
                while (pf.wait_empty()) {
                    if (stop.load(std::memory_order_relaxed)) {  // no new claims once a stop was requested
                        std::this_thread::sleep_for(std::chrono::milliseconds(5));
                        continue;
                    }
                    std::unique_ptr<Fetched> w;
                    try {
                        w.reset(new Fetched());
                        w->rc = claim_once(&w->task, w->eb, sizeof w->eb);
                        w->claimed = Clock::now();
                        if (w->rc == 1 && prove_index_of(w->task, &w->index)) {  // SEGMENTS_PATH, tasks/mod.rs:25
                            w->blob_err = store_get(std::string("job:") + w->task.job_id + ":segments:" + std::to_string(w->index), &w->blob);
                            w->fetched = true;
                        }
                    } catch (const std::exception& e) {
                        if (!w) {  // not even the mailbox entry could be allocated: tell the lane, which then ends like on a stop
                            pf.fetcher_died();
                            return;
                        }
                        if (w->rc != 1) {
                            w->rc = -1;
                            snprintf(w->eb, sizeof w->eb, "exception in the fetcher: %s", e.what());
                        } else {
                            w->fetched = true;
                            w->blob_err = std::string("exception while fetching the segment: ") + e.what();
                        }
                    }
                    pf.put(std::move(w));
                }
            });
        }
        Pending slots[2];
        int cur = 0;
        int64_t idle = 0;
        // the device half of one claimed task + the failure bookkeeping of poll_work; false = fatal (the loop ends)
        auto run_claimed = [&](Fetched& w) -> bool {
            idle = 0;
            Pending* p = &slots[cur];
            p->claimed = w.claimed;  // processing_start, lib.rs:629
            std::string err;
            try {
                err = dispatch(lane, w.task, p, &w);
            } catch (const std::exception& e) {
                err = std::string("[BENTO-WF-115] Prove failed: exception in the device half: ") + e.what();
            }
            if (!err.empty()) {
                std::string f;
                try {
                    metrics.record_task_processing(task_type_label(w.task), "error", secs_since(p->claimed));
                    f = handle_failure(w.task, err);
                } catch (const std::exception& e) {
                    f = std::string("exception while recording a task failure: ") + e.what();
                }
                if (!f.empty()) {
                    set_fatal(f);
                    return false;
                }
                return true;
            }
            fin.post(p);  // blocks while the previous segment's host half is still running
            cur ^= 1;
            return true;
        };
        bool prefetching = cfg.prefetch != 0;
        // ends the fetcher and returns the task it claimed that the lane never took (still 'running' in the task db), if any
        auto stop_fetcher = [&]() -> std::unique_ptr<Fetched> {
            if (!prefetching) return nullptr;
            prefetching = false;
            std::unique_ptr<Fetched> left = pf.quit();
            fetcher.join();
            if (!left) {
                std::lock_guard<std::mutex> l(pf.mu);
                left = std::move(pf.slot);  // put() raced with quit()
            }
            if (left && left->rc != 1) left.reset();
            return left;
        };
        bool fatal_seen = false;
        try {
            while (!stop.load(std::memory_order_relaxed)) {
                std::unique_ptr<Fetched> w;
                if (prefetching) {
                    w = pf.take(stop);
                    if (!w) break;  // stop requested
                    if (w->rc == 0) {
                        // the fetcher's "nothing ready" is as old as the proof that ran meanwhile, and a finish in flight may requeue
                        // its task: the poll that counts as idle is one made now, by the lane, with nothing pending
                        (void)fin.wait_idle();
                        w.reset(new Fetched());
                        w->rc = claim_once(&w->task, w->eb, sizeof w->eb);
                        w->claimed = Clock::now();
                    }
                } else {
                    w.reset(new Fetched());
                    w->rc = claim_once(&w->task, w->eb, sizeof w->eb);
                    w->claimed = Clock::now();
                }
                if (w->rc < 0) {
                    set_fatal(std::string("[BENTO-WF-107] Failed to request_work: ") + w->eb);
                    fatal_seen = true;
                    break;
                }
                if (w->rc == 0) {
                    // a finish still in flight may requeue its task (retry): only a poll made with nothing pending counts as idle
                    if (fin.wait_idle()) continue;
                    if (max_idle_polls >= 0 && ++idle >= max_idle_polls) {
                        // out of idle polls.  A task the fetcher holds is not idleness: its completion may release dependants that no
                        // other lane is left to claim, so the lane runs it and goes on polling (serially from here) until idle again
                        if (std::unique_ptr<Fetched> left = stop_fetcher()) {
                            if (!run_claimed(*left)) {
                                fatal_seen = true;
                                break;
                            }
                            continue;
                        }
                        break;
                    }
                    // sleep poll_time in slices so a stop request is honoured promptly
                    auto until = Clock::now() + std::chrono::duration<double>(cfg.poll_time);
                    while (!stop.load(std::memory_order_relaxed) && Clock::now() < until)
                        std::this_thread::sleep_for(std::chrono::duration<double>(std::min(cfg.poll_time, 0.05)));
                    continue;
                }
                if (!run_claimed(*w)) {
                    fatal_seen = true;
                    break;
                }
            }
        } catch (const std::exception& e) {  // bad_alloc while claiming: the lane ends, its threads are still joined below
            try {
                set_fatal(std::string("[BENTO-WF-107] Failed to request_work: exception in the lane loop: ") + e.what());
            } catch (...) {
                stop.store(1);
            }
            fatal_seen = true;
        }
        // stop request or fatal error: a task the fetcher claimed is still run (not lost), unless the task db itself is failing
        if (std::unique_ptr<Fetched> left = stop_fetcher())
            if (!fatal_seen) (void)run_claimed(*left);
        fin.quit();
        finisher.join();
    }
};

extern "C" {

// ---- mem store / taskdb ----
const char* bx_mem_store_create(bx_mem_store** out) {
    if (!out) return "bx_mem_store_create: out is NULL";
    *out = new (std::nothrow) bx_mem_store();
    return *out ? nullptr : "bx_mem_store_create: out of memory";
}
void bx_mem_store_destroy(bx_mem_store* s) { delete s; }
bx_hot_store_ops bx_mem_store_ops(bx_mem_store* s) { return bx_hot_store_ops{s, mem_get, mem_free_value, mem_set_ex, mem_unlink}; }
size_t bx_mem_store_key_count(bx_mem_store* s) {
    if (!s) return 0;
    std::lock_guard<std::mutex> g(s->mu);
    s->sweep_locked();
    return s->kv.size();
}
const char* bx_mem_store_keys(bx_mem_store* s, char* out, size_t cap) {
    try {
        if (!s || !out || cap == 0) return "bx_mem_store_keys: NULL argument";
        std::lock_guard<std::mutex> g(s->mu);
        s->sweep_locked();
        std::string all;
        for (auto& kv : s->kv) {
            if (!all.empty()) all += "\n";
            all += kv.first;
        }
        snprintf(out, cap, "%s", all.c_str());
        return nullptr;
    } catch (const std::exception&) {
        return "bx_mem_store_keys: out of memory";
    }
}

const char* bx_mem_taskdb_create(bx_mem_taskdb** out) {
    if (!out) return "bx_mem_taskdb_create: out is NULL";
    *out = new (std::nothrow) bx_mem_taskdb();
    return *out ? nullptr : "bx_mem_taskdb_create: out of memory";
}
void bx_mem_taskdb_destroy(bx_mem_taskdb* t) { delete t; }
bx_taskdb_ops bx_mem_taskdb_ops(bx_mem_taskdb* t) {
    return bx_taskdb_ops{t, tdb_request_work, tdb_done, tdb_failed, tdb_retry, tdb_current_retries, tdb_requeue};
}
const char* bx_mem_taskdb_create_task(bx_mem_taskdb* t, const char* stream, const char* job, const char* task, const char* def,
                                      int32_t max_retries) {
    return bx_mem_taskdb_create_task_ex(t, stream, job, task, def, nullptr, 0, max_retries, INT32_MAX);
}
const char* bx_mem_taskdb_create_task_with_prereqs(bx_mem_taskdb* t, const char* stream, const char* job, const char* task, const char* def,
                                                   const char* const* prereqs, size_t n_prereqs, int32_t max_retries) {
    return bx_mem_taskdb_create_task_ex(t, stream, job, task, def, prereqs, n_prereqs, max_retries, INT32_MAX);
}
const char* bx_mem_taskdb_advance_clock(bx_mem_taskdb* t, double seconds) {
    if (!t) return "bx_mem_taskdb_advance_clock: NULL argument";
    if (!(seconds >= 0)) return "bx_mem_taskdb_advance_clock: time only moves forward";
    std::lock_guard<std::mutex> g(t->mu);
    t->clock_skew += seconds;
    return nullptr;
}
const char* bx_mem_taskdb_requeue_tasks(bx_mem_taskdb* t, int64_t limit, uint64_t* timed_out) {
    if (!t) return "bx_mem_taskdb_requeue_tasks: NULL argument";
    char eb[128] = {0};
    int64_t n = tdb_requeue(t, limit, eb, sizeof eb);
    if (n < 0) return fail(eb);
    if (timed_out) *timed_out = (uint64_t)n;
    return nullptr;
}
// taskdb::create_task, 1_taskdb.sql:197-228
const char* bx_mem_taskdb_create_task_ex(bx_mem_taskdb* t, const char* stream, const char* job, const char* task, const char* def,
                                         const char* const* prereqs, size_t n_prereqs, int32_t max_retries, int32_t timeout_secs) {
    try {
        if (!t || !stream || !job || !task || !def || (n_prereqs && !prereqs)) return "bx_mem_taskdb_create_task: NULL argument";
        std::lock_guard<std::mutex> g(t->mu);
        if (t->find_locked(job, task)) return fail(std::string("task already exists: ") + job + ":" + task);
        if (t->rows.size() >= 0xFFFFFFF0u) return "bx_mem_taskdb_create_task: too many rows";
        bx_mem_taskdb::Row r;
        r.stream = stream;
        r.job = job;
        r.task = task;
        r.def = def;
        r.max_retries = max_retries;
        r.timeout_secs = timeout_secs;
        r.created = t->now_s();
        std::vector<uint32_t> pres;
        for (size_t i = 0; i < n_prereqs; ++i) {
            if (!prereqs[i]) return "bx_mem_taskdb_create_task: NULL prerequisite";
            const bx_mem_taskdb::Row* pre = t->find_locked(job, prereqs[i]);
            // task_deps has a foreign key on (job_id, pre_task_id)
            if (!pre) return fail(std::string("prerequisite task does not exist: ") + job + ":" + prereqs[i]);
            pres.push_back(t->index_of(pre));
            if (pre->state != BX_TASK_DONE) r.waiting_on += 1;
        }
        r.state = r.waiting_on ? BX_TASK_PENDING : BX_TASK_READY;
        // everything that can throw happens before the tables change, in an order that is undone on failure
        const uint32_t ix = (uint32_t)t->rows.size();
        auto jit = t->job_index.find(job);
        const bool new_job = jit == t->job_index.end();
        if (new_job) {  // the reference's create_job makes the row (1_taskdb.sql:172-191); here a job starts with its first task
            bx_mem_taskdb::Job j;
            j.id = job;
            t->jobs.push_back(std::move(j));
            try {
                jit = t->job_index.emplace(job, (uint32_t)(t->jobs.size() - 1)).first;
            } catch (...) {
                t->jobs.pop_back();
                throw;
            }
        }
        r.job_ix = jit->second;
        const std::string k = bx_mem_taskdb::key(job, task);
        const std::string stream_name = r.stream;
        const int32_t state = r.state;
        size_t linked = 0;
        bool indexed = false, pushed = false, queued = false;
        try {
            for (uint32_t p : pres) t->rows[p].dependants.reserve(t->rows[p].dependants.size() + 1);
            t->row_index.emplace(k, ix);
            indexed = true;
            t->rows.push_back(std::move(r));
            pushed = true;
            if (state == BX_TASK_READY) {
                t->ready[stream_name].insert({jit->second, ix});
                queued = true;
            }
            for (uint32_t p : pres) {
                t->rows[p].dependants.push_back(ix);  // reserved above: does not throw
                ++linked;
            }
        } catch (...) {
            (void)queued;
            (void)linked;
            if (pushed) t->rows.pop_back();
            if (indexed) t->row_index.erase(k);
            if (new_job) {
                t->job_index.erase(job);
                t->jobs.pop_back();
            }
            throw;
        }
        bx_mem_taskdb::Job& j = t->jobs[jit->second];
        j.tasks++, j.counts[state]++, t->counts[state]++;
        return nullptr;
    } catch (const std::exception&) {
        return "bx_mem_taskdb_create_task: out of memory";
    }
}
const char* bx_mem_taskdb_job_info(bx_mem_taskdb* t, const char* job, bx_job_info* out) {
    try {
        if (!t || !job || !out) return "bx_mem_taskdb_job_info: NULL argument";
        std::lock_guard<std::mutex> g(t->mu);
        memset(out, 0, sizeof *out);
        auto it = t->job_index.find(job);
        if (it == t->job_index.end()) return fail(std::string("no such job: ") + job);
        const bx_mem_taskdb::Job& j = t->jobs[it->second];
        out->state = j.state;  // the jobs row: 'failed' from the first failure on, 'done' when an update_task_done left nothing undone
        out->tasks = j.tasks;
        out->pending = j.counts[BX_TASK_PENDING], out->ready = j.counts[BX_TASK_READY], out->running = j.counts[BX_TASK_RUNNING];
        out->done = j.counts[BX_TASK_DONE], out->failed = j.counts[BX_TASK_FAILED];
        snprintf(out->error, sizeof out->error, "%s", j.error.c_str());
        return nullptr;
    } catch (const std::exception&) {
        return "bx_mem_taskdb_job_info: out of memory";
    }
}

// The executor's writer task (executor.rs:566-698) + process_task (:56-250) for a job whose segments are already in the hot store.
const char* bx_plan_job(bx_mem_taskdb* t, const char* job, uint64_t n_segments, const bx_job_plan* plan_in, uint64_t* tasks_created,
                        uint64_t* root_task) {
    if (!t || !job) return "bx_plan_job: NULL argument";
    if (n_segments == 0) return "bx_plan_job: a job has at least one segment";
    if (n_segments > ((uint64_t)1 << 20)) return "bx_plan_job: more than 2^20 segments in one job";  // 2^40 cycles at po2 20: a typo, not a job
    bx_planner* pl = nullptr;
    try {
        bx_job_plan plan;
        memset(&plan, 0, sizeof plan);
        if (plan_in) plan = *plan_in;
        else plan.prove_retries = plan.join_retries = plan.resolve_retries = plan.finalize_retries = 3;
        // the agent's defaults (bento/crates/workflow/src/lib.rs:108-136), handed to create_task as timeout_secs (executor.rs:82,144,206,226)
        if (plan.prove_timeout <= 0) plan.prove_timeout = 30;
        if (plan.join_timeout <= 0) plan.join_timeout = 10;
        if (plan.resolve_timeout <= 0) plan.resolve_timeout = 120;
        if (plan.finalize_timeout <= 0) plan.finalize_timeout = 10;
        plan.prove_stream[sizeof plan.prove_stream - 1] = plan.join_stream[sizeof plan.join_stream - 1] = plan.aux_stream[sizeof plan.aux_stream - 1] = 0;
        const std::string prove_stream = plan.prove_stream[0] ? plan.prove_stream : "prove";
        const std::string join_stream = plan.join_stream[0] ? plan.join_stream : prove_stream;
        const std::string aux_stream = plan.aux_stream[0] ? plan.aux_stream : "aux";
        if (const char* e = bx_planner_create(&pl)) return e;
        uint64_t created = 0;
        std::string err;
        auto process_task = [&](const bx_plan_task& tt, uint64_t segment_index) -> bool {
            const std::string name = std::to_string(tt.task_number);
            const char* e = nullptr;
            switch (tt.command) {
                case BX_PLAN_SEGMENT:  // the segment INDEX, not the planner's task number, goes into the request (executor.rs:94-104)
                    e = bx_mem_taskdb_create_task_ex(t, prove_stream.c_str(), job, name.c_str(),
                                                     ("{\"Prove\":{\"index\":" + std::to_string(segment_index) + "}}").c_str(), nullptr, 0,
                                                     plan.prove_retries, plan.prove_timeout);
                    created += !e;
                    break;
                case BX_PLAN_JOIN: {
                    const std::string l = std::to_string(tt.depends_on[0]), r = std::to_string(tt.depends_on[1]);
                    const char* pre[2] = {l.c_str(), r.c_str()};
                    e = bx_mem_taskdb_create_task_ex(t, join_stream.c_str(), job, name.c_str(),
                                                     ("{\"Join\":{\"idx\":" + name + ",\"left\":" + l + ",\"right\":" + r + "}}").c_str(), pre, 2,
                                                     plan.join_retries, plan.join_timeout);
                    created += !e;
                    break;
                }
                case BX_PLAN_FINALIZE: {
                    if (root_task) *root_task = tt.depends_on[0];
                    if (plan.subtree_only) break;  // the root receipt is handed to whoever joins the subtrees
                    const std::string m = std::to_string(tt.depends_on[0]);
                    const char* pre[1] = {m.c_str()};
                    e = bx_mem_taskdb_create_task_ex(t, join_stream.c_str(), job, "resolve",
                                                     ("{\"Resolve\":{\"max_idx\":" + m + ",\"union_max_idx\":null}}").c_str(), pre, 1,
                                                     plan.resolve_retries, plan.resolve_timeout);
                    // DEVIATION from executor.rs:179-205: the reference multiplies resolve_timeout by assumption_count =
                    // assumptions + keccak requests, which is 0 for a job like this one — a timeout of 0 s, i.e. a row its
                    // requeue monitor would put back at once.  This planner keeps resolve_timeout x 1.
                    created += !e;
                    if (!e) {
                        const char* pre2[1] = {"resolve"};
                        e = bx_mem_taskdb_create_task_ex(t, aux_stream.c_str(), job, "finalize",
                                                         ("{\"Finalize\":{\"max_idx\":" + m + "}}").c_str(), pre2, 1, plan.finalize_retries,
                                                         plan.finalize_timeout);
                        created += !e;
                    }
                    break;
                }
                default: e = "bx_plan_job: the planner produced a keccak/union task for a job without coprocessor requests";
            }
            if (e) err = e;
            return !e;
        };
        auto drain = [&](uint64_t segment_index) -> bool {
            for (;;) {
                bx_plan_task tt;
                int has = 0;
                if (const char* e = bx_planner_next_task(pl, &tt, &has)) {
                    err = e;
                    return false;
                }
                if (!has) return true;
                if (!process_task(tt, segment_index)) return false;
            }
        };
        bool ok = true;
        for (uint64_t i = 0; ok && i < n_segments; ++i) {
            uint64_t n = 0;
            if (const char* e = bx_planner_enqueue_segment(pl, &n)) {
                err = e;
                ok = false;
            } else {
                ok = drain(i);
            }
        }
        if (ok) {
            uint64_t n = 0;
            if (const char* e = bx_planner_finish(pl, &n)) {
                err = e;
                ok = false;
            } else {
                ok = drain(0);
            }
        }
        bx_planner_destroy(pl);
        pl = nullptr;
        if (tasks_created) *tasks_created = created;
        return ok ? nullptr : fail("bx_plan_job: " + err);
    } catch (const std::exception& e) {
        if (pl) bx_planner_destroy(pl);
        return fail(std::string("bx_plan_job: ") + e.what());
    }
}
const char* bx_mem_taskdb_task_info(bx_mem_taskdb* t, const char* job, const char* task, bx_task_info* out) {
    try {
        if (!t || !job || !task || !out) return "bx_mem_taskdb_task_info: NULL argument";
        std::lock_guard<std::mutex> g(t->mu);
        auto* r = t->find_locked(job, task);
        if (!r) return fail(std::string("no such task: ") + job + ":" + task);
        out->state = r->state;
        out->retries = r->retries;
        out->max_retries = r->max_retries;
        out->waiting_on = r->waiting_on;
        out->timeout_secs = r->timeout_secs;
        out->created_s = r->created;
        out->started_s = r->started;
        out->updated_s = r->updated;
        snprintf(out->error, sizeof out->error, "%s", r->error.c_str());
        snprintf(out->output, sizeof out->output, "%s", r->output.c_str());
        return nullptr;
    } catch (const std::exception&) {
        return "bx_mem_taskdb_task_info: out of memory";
    }
}
// clear_completed_jobs (4_clear_completed_streams.sql): the rows of every 'done' job leave the table; returns the jobs cleared
const char* bx_mem_taskdb_clear_completed_jobs(bx_mem_taskdb* t, uint64_t* cleared) {
    if (!t) return "bx_mem_taskdb_clear_completed_jobs: NULL argument";
    try {
        std::lock_guard<std::mutex> g(t->mu);
        uint64_t n_done = 0;
        for (auto& j : t->jobs) n_done += j.state == BX_JOB_DONE;
        if (cleared) *cleared = n_done;
        if (!n_done) return nullptr;
        // build the surviving tables beside the old ones, then swap: a failure half way leaves the table as it was
        const uint32_t GONE = 0xFFFFFFFFu;
        std::vector<uint32_t> job_map(t->jobs.size(), GONE), row_map(t->rows.size(), GONE);
        std::vector<bx_mem_taskdb::Job> jobs;
        std::unordered_map<std::string, uint32_t> job_index, row_index;
        for (size_t i = 0; i < t->jobs.size(); ++i)
            if (t->jobs[i].state != BX_JOB_DONE) {
                job_map[i] = (uint32_t)jobs.size();
                job_index.emplace(t->jobs[i].id, job_map[i]);
                jobs.push_back(t->jobs[i]);
            }
        uint32_t kept = 0;
        for (size_t i = 0; i < t->rows.size(); ++i)
            if (job_map[t->rows[i].job_ix] != GONE) row_map[i] = kept++;
        std::deque<bx_mem_taskdb::Row> rows;
        std::map<std::string, std::set<std::pair<uint32_t, uint32_t>>> ready;
        std::set<uint32_t> running;
        uint64_t counts[5] = {0, 0, 0, 0, 0};
        for (size_t i = 0; i < t->rows.size(); ++i) {
            if (row_map[i] == GONE) continue;
            bx_mem_taskdb::Row r = t->rows[i];  // a copy: the old table stays whole until the swap
            r.job_ix = job_map[r.job_ix];
            for (uint32_t& d : r.dependants) d = row_map[d];  // dependants are rows of the same job: they survive with it
            row_index.emplace(bx_mem_taskdb::key(r.job.c_str(), r.task.c_str()), row_map[i]);
            if (r.state == BX_TASK_READY) ready[r.stream].insert({r.job_ix, row_map[i]});
            if (r.state == BX_TASK_RUNNING) running.insert(row_map[i]);
            counts[r.state]++;
            rows.push_back(std::move(r));
        }
        t->rows.swap(rows);
        t->jobs.swap(jobs);
        t->job_index.swap(job_index);
        t->row_index.swap(row_index);
        t->ready.swap(ready);
        t->running.swap(running);
        memcpy(t->counts, counts, sizeof counts);
        return nullptr;
    } catch (const std::exception&) {
        return "bx_mem_taskdb_clear_completed_jobs: out of memory";
    }
}
size_t bx_mem_taskdb_count(bx_mem_taskdb* t, int32_t state) {
    if (!t) return 0;
    if (state < 0 || state > 4) return 0;
    std::lock_guard<std::mutex> g(t->mu);
    return (size_t)t->counts[state];
}

// ---- wire ----
void bx_segment_encode(uint64_t index, uint32_t po2, uint64_t seed, uint8_t out[BX_SEGMENT_WIRE_BYTES]) {
    memcpy(out, BX_SEGMENT_MAGIC, 8);
    put_le(out + 8, index, 8);
    put_le(out + 16, po2, 4);
    put_le(out + 20, seed, 8);
}
const char* bx_segment_decode(const uint8_t* blob, size_t len, uint64_t* index, uint32_t* po2, uint64_t* seed) {
    if (!blob || len < BX_SEGMENT_WIRE_BYTES || memcmp(blob, BX_SEGMENT_MAGIC, 8) != 0)
        return "Failed to deserialize segment data from redis: not a synthetic segment blob (the built-in prover proves the "
               "synthetic circuit only; a bincode(Segment) needs a prover plugged in through prove_blob)";
    if (index) *index = get_le(blob + 8, 8);
    if (po2) *po2 = (uint32_t)get_le(blob + 16, 4);
    if (seed) *seed = get_le(blob + 20, 8);
    return nullptr;
}

// ---- agent ----
const char* bx_agent_create(const bx_agent_config* cfg, const bx_hot_store_ops* store, const bx_taskdb_ops* taskdb,
                            const bx_segment_prover_ops* prover, bx_agent** out) {
    if (!cfg || !store || !taskdb || !out) return "bx_agent_create: NULL argument";
    if (!store->get || !store->set_ex || !store->unlink) return "bx_agent_create: hot store ops incomplete";
    if (!taskdb->request_work || !taskdb->update_task_done || !taskdb->update_task_failed || !taskdb->update_task_retry ||
        !taskdb->current_retries)
        return "bx_agent_create: task db ops incomplete";
    if (cfg->no_prover) prover = nullptr;  // `prover: None` (lib.rs:242-252): whatever table was passed is not used
    if (prover && !prover->prove_blob && (!prover->prove_segment || !prover->seal_words)) return "bx_agent_create: prover ops incomplete";
    const bool opaque = prover && prover->prove_blob;
    if (!opaque && !cfg->synthetic)
        return "bx_agent_create: the built-in prover proves the synthetic circuit of bx_prover.h, not rv32im segments: set "
               "cfg.synthetic = 1 to run it (seals go to job:{id}:synthetic_receipts:{task}), or plug a real prover in through "
               "bx_segment_prover_ops::prove_blob";
    bx_agent* a = nullptr;
    try {
        a = new bx_agent();
        a->cfg = *cfg;
        a->cfg.task_stream[sizeof a->cfg.task_stream - 1] = 0;
        if (!a->cfg.task_stream[0]) snprintf(a->cfg.task_stream, sizeof a->cfg.task_stream, "prove");
        if (a->cfg.inflight == 0) a->cfg.inflight = 3;
        if (a->cfg.inflight > 16 || a->cfg.n_devices > 16) {
            delete a;
            return "bx_agent_create: inflight and n_devices must be <= 16";
        }
        if (!a->cfg.w_code) a->cfg.w_code = 16;
        if (!a->cfg.w_data) a->cfg.w_data = 256;
        if (!a->cfg.w_accum) a->cfg.w_accum = 64;
        if (!a->cfg.redis_ttl) a->cfg.redis_ttl = 8 * 60 * 60;
        if (!(a->cfg.poll_time > 0)) a->cfg.poll_time = 1.0;
        if (!a->cfg.po2_min) a->cfg.po2_min = 9;
        // the prover takes po2 up to 24, but a lane's buffers grow with the size (8.5 GB at 2^20, 34 GB at 2^22, 135 GB at 2^24 for
        // 16/256/64): the default keeps `inflight` x 2 cached shapes inside one 288 GB GPU; raise po2_max with fewer lanes
        if (!a->cfg.po2_max) a->cfg.po2_max = 22;
        if (a->cfg.po2_min < 9 || a->cfg.po2_max > 24 || a->cfg.po2_min > a->cfg.po2_max) {
            delete a;
            return "bx_agent_create: po2_min / po2_max must satisfy 9 <= po2_min <= po2_max <= 24";
        }
        // widths x the largest accepted size, checked here instead of at the first oversized task: bx_prover_create takes widths below
        // 65536 (16-bit tap bookkeeping), and ONE buffer set of a lane — (w + 16) columns x 2^po2_max rows x (1 coefficient + 4
        // evaluation words) — must at least fit the 288 GB of one MI355X
        if (!prover && !a->cfg.no_prover && (a->cfg.w_code >= 65536 || a->cfg.w_data >= 65536 || a->cfg.w_accum >= 65536)) {
            delete a;
            return "bx_agent_create: group widths must be below 65536";
        }
        if (!prover && !a->cfg.no_prover && (((uint64_t)a->cfg.w_code + a->cfg.w_data + a->cfg.w_accum + 16u) * 20u << a->cfg.po2_max) > (288ull << 30)) {
            delete a;
            return "bx_agent_create: the group widths at po2_max need more than the 288 GB of one GPU for a single buffer set: lower po2_max or the widths";
        }
        if (!a->cfg.max_shapes) a->cfg.max_shapes = 2;
        if (a->cfg.lift_po2 && (a->cfg.lift_po2 < 9 || a->cfg.lift_po2 > 24)) {
            delete a;
            return "bx_agent_create: lift_po2 must be 0 or in [9, 24]";
        }
        if (!a->cfg.join_po2) a->cfg.join_po2 = 18;
        a->cfg.prefetch = a->cfg.prefetch ? 1 : 0;
        if (a->cfg.join_po2 < 9 || a->cfg.join_po2 > 24) {
            delete a;
            return "bx_agent_create: join_po2 must be in [9, 24]";
        }
        a->cfg.also_streams[sizeof a->cfg.also_streams - 1] = 0;
        a->streams.emplace_back(a->cfg.task_stream);
        for (const char* q = a->cfg.also_streams; *q;) {
            const char* e = strchr(q, ',');
            std::string name(q, e ? (size_t)(e - q) : strlen(q));
            if (!name.empty() && name != a->streams[0]) a->streams.push_back(name);
            q = e ? e + 1 : q + strlen(q);
        }
        if (a->cfg.n_devices == 0) {
            a->cfg.n_devices = 1;
            a->cfg.devices[0] = a->cfg.device;
        }
        a->store = *store;
        a->taskdb = *taskdb;
        // lane l proves on device devices[l / inflight]
        for (uint32_t l = 0; l < a->cfg.n_devices * a->cfg.inflight; ++l) {
            a->lanes.emplace_back(new Lane());
            a->lanes.back()->device = a->cfg.devices[l / a->cfg.inflight];
        }
        if (a->cfg.no_prover) {
            // Agent::new gives only the prove / join / coproc worker types a prover (lib.rs:242-252); an aux agent has none, needs no
            // GPU, and a Prove or Join task that reaches it fails with the reference's "Missing prover" errors
            a->prover = bx_segment_prover_ops{nullptr, nullptr, nullptr, nullptr, nullptr};
        } else if (prover) {
            a->prover = *prover;
        } else {
            a->hip = true;
            a->prover = bx_segment_prover_ops{a, bx_agent::hip_seal_words, bx_agent::hip_prove, nullptr, nullptr};
            if (const char* e = bx_verifier_ctx_create(&a->vctx)) {
                std::string m = std::string("bx_agent_create: ") + e;
                (void)bx_agent_destroy(a);
                return fail(m);
            }
            // like Agent::new (lib.rs:241-252) the device contexts are created up front so a missing GPU fails here, loudly
            for (uint32_t d = 0; d < a->cfg.n_devices; ++d) {
                Lane& first = *a->lanes[(size_t)d * a->cfg.inflight];
                if (const char* e = bx_init(first.device, &first.ctx)) {
                    std::string m = std::string("bx_agent_create: device ") + std::to_string(first.device) + ": " + e;
                    (void)bx_agent_destroy(a);
                    return fail(m);
                }
            }
        }
    } catch (const std::exception& e) {
        if (a) (void)bx_agent_destroy(a);
        return fail(std::string("bx_agent_create: ") + e.what());
    }
    *out = a;
    return nullptr;
}

const char* bx_agent_destroy(bx_agent* a) {
    if (!a) return nullptr;
    const char* first = nullptr;
    try {
        for (auto& lp : a->lanes) {
            Lane& lane = *lp;
            for (auto& kv : lane.provers)
                if (const char* e = bx_prover_destroy(kv.second))
                    if (!first) first = fail(e);
            if (lane.ctx)
                if (const char* e = bx_free(lane.ctx))
                    if (!first) first = fail(e);
        }
    } catch (...) {
        first = "bx_agent_destroy: exception";
    }
    bx_verifier_ctx_destroy(a->vctx);
    delete a;
    return first;
}

uint32_t bx_agent_lane_count(const bx_agent* a) { return a ? (uint32_t)a->lanes.size() : 0; }
int32_t bx_agent_lane_device(const bx_agent* a, uint32_t lane) { return a && lane < a->lanes.size() ? a->lanes[lane]->device : -1; }
uint64_t bx_agent_lane_tasks_done(const bx_agent* a, uint32_t lane) { return a && lane < a->lanes.size() ? a->lanes[lane]->done.load() : 0; }

void bx_agent_stop(bx_agent* a) {
    if (a) a->stop.store(1, std::memory_order_relaxed);
}

const char* bx_agent_prewarm(bx_agent* a, uint32_t po2) {
    if (!a) return "bx_agent_prewarm: NULL agent";
    if (!a->hip) return nullptr;
    try {
        for (uint32_t l = 0; l < a->lanes.size(); ++l) {
            bx_prover* p = nullptr;
            std::string err;
            if (a->hip_prover_for(l, po2, &p, &err)) return fail("bx_agent_prewarm: lane " + std::to_string(l) + ": " + err);
        }
        return nullptr;
    } catch (const std::exception& e) {
        return fail(std::string("bx_agent_prewarm: ") + e.what());
    }
}

const char* bx_agent_poll_work(bx_agent* a, int64_t max_idle_polls, uint64_t* tasks_done) {
    if (!a) return "bx_agent_poll_work: NULL agent";
    std::atomic<uint64_t> done{0};
    std::vector<std::string> fatal;
    std::vector<std::thread> threads;
    // the requeue monitor (lib.rs:283-303 -> poll_for_requeue :536-551): tasks that stayed 'running' past their timeout — their lane
    // hung, or the process that claimed them died — go back to 'ready' for any other lane; runs beside the lanes until they are done
    std::atomic<int> monitor_quit{0};
    std::thread monitor;
    if (a->cfg.monitor_requeue && a->taskdb.requeue_tasks) {
        try {
            monitor = std::thread([a, &monitor_quit] {
                const double interval = a->cfg.requeue_poll_interval > 0 ? a->cfg.requeue_poll_interval : 5.0;
                while (!monitor_quit.load(std::memory_order_relaxed) && !a->stop.load(std::memory_order_relaxed)) {
                    char eb[256] = {0};
                    // an error is retried at the next tick, as in the reference, which logs both outcomes (lib.rs:536-551)
                    const int n = a->taskdb.requeue_tasks(a->taskdb.user, 100, eb, sizeof eb);
                    if (n < 0) fprintf(stderr, "[bx_agent] requeue monitor: requeue_tasks failed: %s\n", eb);
                    else if (n > 0) fprintf(stderr, "[bx_agent] requeue monitor: %d task(s) requeued after their timeout\n", n);
                    auto until = Clock::now() + std::chrono::duration<double>(interval);
                    while (!monitor_quit.load(std::memory_order_relaxed) && !a->stop.load(std::memory_order_relaxed) && Clock::now() < until)
                        std::this_thread::sleep_for(std::chrono::duration<double>(std::min(interval, 0.02)));
                }
            });
        } catch (...) {
            return "bx_agent_poll_work: could not start the requeue monitor";
        }
    }
    struct MonitorJoin {
        std::atomic<int>& quit;
        std::thread& th;
        ~MonitorJoin() {
            quit.store(1);
            if (th.joinable()) th.join();
        }
    } monitor_join{monitor_quit, monitor};
    try {
        fatal.resize(a->lanes.size());
        for (uint32_t l = 1; l < a->lanes.size(); ++l)
            threads.emplace_back([a, l, max_idle_polls, &done, &fatal] { a->lane_loop(l, max_idle_polls, &done, &fatal[l]); });
    } catch (...) {
        a->stop.store(1);
        for (auto& t : threads) t.join();
        return "bx_agent_poll_work: could not start lane threads";
    }
    try {
        a->lane_loop(0, max_idle_polls, &done, &fatal[0]);
    } catch (const std::exception& e) {
        a->stop.store(1);
        fatal[0] = std::string("bx_agent_poll_work: ") + e.what();
    }
    for (auto& t : threads) t.join();
    if (tasks_done) *tasks_done = done.load();
    for (auto& f : fatal)
        if (!f.empty()) return fail(f);
    return nullptr;
}

const char* bx_agent_process_one(bx_agent* a, const bx_ready_task* task, int* ok) {
    if (!a || !task) return "bx_agent_process_one: NULL argument";
    try {
        Pending p;
        p.claimed = Clock::now();
        bool fatal = false;
        std::string err = a->dispatch(0, *task, &p);
        if (err.empty()) err = a->complete(&p, &fatal);
        a->metrics.record_task_processing(bx_agent::task_type_label(*task), err.empty() ? "success" : "error", secs_since(p.claimed));
        if (ok) *ok = err.empty();
        if (err.empty()) {
            a->lanes[0]->done.fetch_add(1);
            return nullptr;
        }
        if (fatal) return fail(err);
        std::string f = a->handle_failure(*task, err);
        return f.empty() ? nullptr : fail(f);
    } catch (const std::exception& e) {
        if (ok) *ok = 0;
        return fail(std::string("bx_agent_process_one: ") + e.what());
    }
}

size_t bx_agent_metrics(bx_agent* a, char* out, size_t cap) {
    if (!a) return 0;
    try {
        std::string s = a->metrics.exposition();
        if (out && cap) snprintf(out, cap, "%s", s.c_str());
        return s.size() + 1;
    } catch (...) {
        return 0;
    }
}

}  // extern "C"
