/*
 * bx_agent.h — C ABI of the native (C++) host runtime around the segment prover: the prove-stream feed loop and the
 * join-tree planner.  Host code only; lives in the same libbx_hip_hal.so.
 *
 * Reference interfaces restated here
 * ----------------------------------
 *   Agent::poll_work              bento/crates/workflow/src/lib.rs:279-442   claim -> run -> done | retry | failed
 *   Agent::process_work           bento/crates/workflow/src/lib.rs:445-530   TaskType dispatch ("Invalid task_def", WF-115)
 *   tasks::prove::prover          bento/crates/workflow/src/tasks/prove.rs:18-135
 *   key scheme                    bento/crates/workflow/src/tasks/mod.rs:23-29 (job:{id}:segments:{i}, ...:recursion_receipts:{task})
 *   hot-store calls               bento/crates/workflow/src/redis.rs:19-63   (GET / SETEX / UNLINK + redis_operations metrics)
 *   task db calls                 bento/crates/taskdb/src/lib.rs:236-326     request_work / update_task_done|failed|retry
 *                                 bento/crates/taskdb/migrations/1_taskdb.sql:308-391 (state transitions those functions make)
 *   metrics                       bento/crates/workflow-common/src/metrics.rs:61-70,108-117,288-335
 *   Planner                       bento/crates/taskdb/src/planner/mod.rs:20-252, planner/task.rs:8-81
 *
 * The reference's Redis and Postgres clients are control plane and out of scope: they enter only as two
 * small tables of callbacks (`bx_hot_store_ops`, `bx_taskdb_ops`) a deployment fills with its own clients.  The library
 * ships in-memory implementations of both for tests, the bench's queue mode and single-box runs.
 *
 * The agent owns the device context(s) for the process lifetime, like `Agent.prover` (lib.rs:192,241-252).  Unlike the
 * reference (one task in flight per process, one agent process per GPU), one bx_agent runs `inflight` prover lanes on
 * each of its GPUs — each lane a host thread with its own bx_ctx/stream/prover claiming tasks independently — because the
 * segment prover is VALU-issue-bound with latency-bound tails and 3 lanes fill a chip; with
 * `n_devices` > 1 the lanes of all GPUs of the node claim from the same task db, which is BASELINE configs[2]'s
 * "batch work-stolen across 8 GPUs" in native code.  Every callback may be invoked from several threads at once.
 *
 * Conventions as in bx_hal.h: every call returns NULL on success or a message owned by the object it was called on
 * (valid until the next call on that object from the same thread); nothing aborts, no exception crosses the ABI.
 */
#ifndef BX_AGENT_H
#define BX_AGENT_H
#include "bx_prover.h"
#ifdef __cplusplus
extern "C" {
#endif
/* the library is built with -fvisibility=hidden: only what these headers declare is exported */
#if defined(__GNUC__)
#pragma GCC visibility push(default)
#endif

/* ------------------------------------------------------------------------------------------------ planner ---- */
typedef struct bx_planner bx_planner;
enum bx_plan_command { BX_PLAN_SEGMENT = 0, BX_PLAN_KECCAK = 1, BX_PLAN_JOIN = 2, BX_PLAN_UNION = 3, BX_PLAN_FINALIZE = 4 };
typedef struct bx_plan_task { /* planner/task.rs:8-26 */
    uint64_t task_number;
    uint32_t task_height;
    uint32_t command; /* enum bx_plan_command */
    uint32_t n_depends_on;
    uint32_t n_keccak_depends_on;
    uint64_t depends_on[2];
    uint64_t keccak_depends_on[2];
} bx_plan_task;

const char* bx_planner_create(bx_planner** out);
void bx_planner_destroy(bx_planner* p);
/* Planner::enqueue_segment / enqueue_keccak: errors "Cannot add segment to finished plan" after finish(). */
const char* bx_planner_enqueue_segment(bx_planner* p, uint64_t* task_number);
const char* bx_planner_enqueue_keccak(bx_planner* p, uint64_t* task_number);
/* Planner::finish: error "Planning not yet started" when no segment was enqueued; idempotent afterwards. */
const char* bx_planner_finish(bx_planner* p, uint64_t* task_number);
/* Planner::next_task: *has = 0 when every planned task has been handed out. */
const char* bx_planner_next_task(bx_planner* p, bx_plan_task* out, int* has);
size_t bx_planner_task_count(const bx_planner* p);
const char* bx_planner_get_task(bx_planner* p, uint64_t task_number, bx_plan_task* out);

/* ------------------------------------------------------------------------------------- stores (callbacks) ---- */
/* Hot store = the three Redis operations the prove task issues.  Return 0 = ok, 1 = key not found (get only),
 * negative = transport error; on error `errbuf` (cap bytes) may be filled with a message.  A value handed out by `get` is
 * read-only and stays valid until `free_value` (the in-memory store lends its own copy of the bytes: an ~80 MB segment is not
 * duplicated on its way to the prover). */
typedef struct bx_hot_store_ops {
    void* user;
    int (*get)(void* user, const char* key, uint8_t** value, size_t* len, char* errbuf, size_t cap);
    void (*free_value)(void* user, uint8_t* value);
    int (*set_ex)(void* user, const char* key, const uint8_t* value, size_t len, uint64_t ttl_secs /* 0 = no expiry */,
                  char* errbuf, size_t cap);
    int (*unlink)(void* user, const char* key, char* errbuf, size_t cap);
} bx_hot_store_ops;

typedef struct bx_ready_task { /* taskdb::ReadyTask, bento/crates/taskdb/src/lib.rs:58-64 */
    char job_id[40];            /* uuid text */
    char task_id[128];
    char task_def[1024];        /* JSON TaskType, e.g. {"Prove":{"index":3}} */
    int32_t max_retries;
} bx_ready_task;

/* Task db = the calls poll_work makes.  request_work: 1 = task claimed (now 'running'), 0 = none ready, <0 error.
 * current_retries: 1 = found a running row and wrote *retries, 0 = no such running row, <0 error.
 * update_*: 1 = row updated, 0 = not found / not in a state the update applies to, <0 error. */
typedef struct bx_taskdb_ops {
    void* user;
    int (*request_work)(void* user, const char* task_stream, bx_ready_task* out, char* errbuf, size_t cap);
    int (*update_task_done)(void* user, const char* job_id, const char* task_id, const char* output_json, char* errbuf,
                            size_t cap);
    int (*update_task_failed)(void* user, const char* job_id, const char* task_id, const char* error, char* errbuf,
                              size_t cap);
    int (*update_task_retry)(void* user, const char* job_id, const char* task_id, char* errbuf, size_t cap);
    int (*current_retries)(void* user, const char* job_id, const char* task_id, int32_t* retries, char* errbuf,
                           size_t cap);
    /* taskdb::requeue_tasks (bento/crates/taskdb/src/lib.rs:328-358), what the agent's requeue monitor calls
     * (cfg.monitor_requeue): up to `limit` tasks 'running' for longer than their timeout_secs go through update_task_retry.
     * Returns the number of timed-out tasks found, < 0 on error.  May be NULL (a task db that requeues by itself, like the
     * API server behind the REST worker protocol): the monitor is then not started. */
    int64_t (*requeue_tasks)(void* user, int64_t limit, char* errbuf, size_t cap);
} bx_taskdb_ops;

/* In-memory implementations (thread-safe). The returned ops borrow the object; destroy it after the agent. */
typedef struct bx_mem_store bx_mem_store;
const char* bx_mem_store_create(bx_mem_store** out);
void bx_mem_store_destroy(bx_mem_store* s);
bx_hot_store_ops bx_mem_store_ops(bx_mem_store* s);
size_t bx_mem_store_key_count(bx_mem_store* s);
/* Writes the sorted keys separated by '\n' (NUL-terminated, truncated to cap). */
const char* bx_mem_store_keys(bx_mem_store* s, char* out, size_t cap);

typedef struct bx_mem_taskdb bx_mem_taskdb;
/* task_state, 1_taskdb.sql:12-19 (pending = waiting on the completion of prerequisites) */
enum bx_task_state { BX_TASK_READY = 0, BX_TASK_RUNNING = 1, BX_TASK_DONE = 2, BX_TASK_FAILED = 3, BX_TASK_PENDING = 4 };
typedef struct bx_task_info {
    int32_t state; /* enum bx_task_state */
    int32_t retries;
    int32_t max_retries;
    char error[1100];
    char output[256];
    int32_t waiting_on;  /* prerequisites not yet done (tasks.waiting_on) */
    int32_t timeout_secs; /* tasks.timeout_secs (INT32_MAX = created without one) */
    double created_s, started_s, updated_s; /* seconds since the task db was created: created_at, started_at (most recent claim),
                                             * updated_at (done | failed | retry); 0 = not yet */
} bx_task_info;
const char* bx_mem_taskdb_create(bx_mem_taskdb** out);
void bx_mem_taskdb_destroy(bx_mem_taskdb* t);
bx_taskdb_ops bx_mem_taskdb_ops(bx_mem_taskdb* t);
/* taskdb::create_task for a task with no prerequisites (it is 'ready' at once). */
const char* bx_mem_taskdb_create_task(bx_mem_taskdb* t, const char* task_stream, const char* job_id, const char* task_id,
                                      const char* task_def_json, int32_t max_retries);
/* taskdb::create_task (1_taskdb.sql:197-228): the task is 'pending' while any of its prerequisites (task ids of the same job,
 * which must exist) is not 'done', 'ready' otherwise; update_task_done on a prerequisite decrements waiting_on and releases
 * the task when it reaches zero (1_taskdb.sql:296-306); update_task_failed also applies to pending tasks (:324).
 * request_work hands out the oldest ready task of the OLDEST job of the worker type (job_created_at ASC, created_at ASC:
 * 9_request_work.sql:139-141 — job-level FIFO: a job's late-created joins go before a younger job's proves).  Every operation is O(log rows):
 * a 2^16-segment job (131 075 rows) is planned and drained in a second (tests/test_taskdb_model_cpu.py, which also checks the
 * table against a row-by-row restatement of the SQL on random operation sequences).  One difference: a prerequisite listed
 * twice is released twice here; the SQL counts it twice and releases it once, which leaves the task pending for ever. */
const char* bx_mem_taskdb_create_task_with_prereqs(bx_mem_taskdb* t, const char* task_stream, const char* job_id, const char* task_id,
                                                   const char* task_def_json, const char* const* prerequisites, size_t n_prerequisites,
                                                   int32_t max_retries);
/* The full form (taskdb::create_task, bento/crates/taskdb/src/lib.rs:201-210): timeout_secs = how long the task may stay 'running'
 * (since its claim or last update) before requeue_tasks sends it through update_task_retry.  The two shorter forms above create
 * tasks that never time out (INT32_MAX). */
const char* bx_mem_taskdb_create_task_ex(bx_mem_taskdb* t, const char* task_stream, const char* job_id, const char* task_id,
                                         const char* task_def_json, const char* const* prerequisites, size_t n_prerequisites,
                                         int32_t max_retries, int32_t timeout_secs);
/* taskdb::requeue_tasks on this table (see bx_taskdb_ops::requeue_tasks); limit < 0 = no limit; *timed_out may be NULL. */
const char* bx_mem_taskdb_requeue_tasks(bx_mem_taskdb* t, int64_t limit, uint64_t* timed_out);
/* Test hook: the table's clock (created_at / started_at / updated_at / now) jumps `seconds` >= 0 forward. */
const char* bx_mem_taskdb_advance_clock(bx_mem_taskdb* t, double seconds);
const char* bx_mem_taskdb_task_info(bx_mem_taskdb* t, const char* job_id, const char* task_id, bx_task_info* out);
size_t bx_mem_taskdb_count(bx_mem_taskdb* t, int32_t state);
/* clear_completed_jobs (bento/crates/taskdb/migrations/4_clear_completed_streams.sql): every row of every 'done' job leaves the
 * table (the maintenance call that keeps a long-lived table bounded); *cleared (may be NULL) = the number of jobs removed.
 * Failed and running jobs stay. */
const char* bx_mem_taskdb_clear_completed_jobs(bx_mem_taskdb* t, uint64_t* cleared);
/* job_state (1_taskdb.sql:5-9), a stored row as in the reference: running; -> done by the update_task_done that leaves no task
 * of the job in another state (:308-311; a task created afterwards does not reopen it); -> failed, with that task's error, by the
 * FIRST update_task_failed in time (:333-340).  The row is created with the job's first task (the reference's create_job). */
enum bx_job_state { BX_JOB_RUNNING = 0, BX_JOB_DONE = 1, BX_JOB_FAILED = 2 };
typedef struct bx_job_info {
    int32_t state; /* enum bx_job_state */
    uint64_t tasks, pending, ready, running, done, failed;
    char error[1100];
} bx_job_info;
const char* bx_mem_taskdb_job_info(bx_mem_taskdb* t, const char* job_id, bx_job_info* out);

/* ------------------------------------------------------------------------------------ the planner's DAG as tasks ---- */
/* What the executor's writer task does with the planner (bento/crates/workflow/src/tasks/executor.rs:566-698 driving
 * process_task, :56-250) for a job of `n_segments` segments already flushed to the hot store: enqueue_segment per segment,
 * then finish(), creating one task row per planner task —
 *   Segment  -> task "{n}"       {"Prove":{"index":i}}                     prove stream, no prerequisites      (:92-127)
 *   Join     -> task "{n}"       {"Join":{"idx":n,"left":l,"right":r}}     join stream, prerequisites [l, r]   (:128-153)
 *   Finalize -> task "resolve"   {"Resolve":{"max_idx":m,"union_max_idx":null}}  join stream, prerequisite [m] (:176-209)
 *               task "finalize"  {"Finalize":{"max_idx":m}}                aux stream, prerequisite ["resolve"] (:211-229)
 * (no keccak / union / snark tasks: a synthetic job has no coprocessor requests and compress = None).  join_stream "" = the
 * prove stream, as in the reference without JOIN_STREAM (executor.rs:515-524). */
typedef struct bx_job_plan {
    char prove_stream[64]; /* "" = "prove" */
    char join_stream[64];  /* "" = prove_stream */
    char aux_stream[64];   /* "" = "aux" */
    int32_t prove_retries, join_retries, resolve_retries, finalize_retries; /* the reference's defaults are 3 */
    int32_t subtree_only; /* 1 = stop at the root join: no resolve / finalize tasks.  The job is one GPU's share of a larger job whose
                           * top levels another agent joins from the subtree roots (one process per GPU: the roots cross GPUs, the
                           * segments never do) */
    int32_t prove_timeout, join_timeout, resolve_timeout, finalize_timeout; /* timeout_secs of the tasks created; <= 0 = the
                           * agent's defaults 30 / 10 / 120 / 10 (bento/crates/workflow/src/lib.rs:108-136).  The resolve row gets
                           * resolve_timeout x 1: the reference multiplies by its assumption count (executor.rs:179-205), which is 0
                           * for a job without assumptions or keccak requests — a deliberate deviation, a 0 s timeout requeues at once */
} bx_job_plan;
/* root_task (may be NULL) receives the task number whose receipt is the job's root: the last join, or task 0 for a single segment. */
const char* bx_plan_job(bx_mem_taskdb* t, const char* job_id, uint64_t n_segments, const bx_job_plan* plan /* NULL = defaults */,
                        uint64_t* tasks_created, uint64_t* root_task);

/* ---------------------------------------------------------------------------------- segment / receipt wire ---- */
/* The reference moves bincode(risc0_zkvm::Segment) in and bincode(receipt) out (tasks/mod.rs:40-47); both need risc0's type
 * layouts and a real circuit.  Two modes, chosen by the prover table:
 *   opaque     ops->prove_blob != NULL: the agent hands the stored bytes to the prover untouched and stores what it
 *              returns under the reference's key  job:{id}:recursion_receipts:{task}  (tasks/mod.rs:23).  This is the
 *              drop-in path for a real prover.
 *   synthetic  otherwise (the built-in HIP prover, or an injected prove_segment): blobs are the tagged stand-ins below and
 *              the seal goes to  job:{id}:synthetic_receipts:{task}  — a different key on purpose, so that a Join worker
 *              of a real cluster can never pick a synthetic seal up as a lifted SuccinctReceipt.  Needs cfg.synthetic = 1.
 *   segment  = "BXSYNSEG" | index u64 | po2 u32 | seed u64 | payload ...        (28 bytes + payload, little endian; bx_prover.h)
 *   receipt  = "BXSYNRCP" | index u64 | po2 u32 | seal_words u32 | seal u32[]   (24 + 4*n bytes, little endian)
 * A blob without the tag (e.g. a real bincode Segment) fails the task with a message naming the mismatch. */
#define BX_RECEIPT_HEADER_BYTES 24
#define BX_RECEIPT_MAGIC "BXSYNRCP"
/* Stand-ins for the recursion tasks (synthetic mode only; the recursion circuit is not available offline), so
 * that the DAG the planner produces — K proves, a log-depth join tail, resolve, finalize — runs through the same lanes:
 *   Join{idx,left,right}  (join.rs:18-113)  reads both child receipts, verifies both, proves ONE synthetic segment of
 *                         2^join_po2 cycles whose seed is bx_join_seed(left seal, right seal) — a stand-in for
 *                         `prover.join(&left, &right)`, NOT a recursion proof — verifies it, stores it under
 *                         job:{id}:synthetic_receipts:{idx} and unlinks the children;
 *   Resolve{max_idx}      (resolve.rs:18-180 without assumptions) reads, verifies and re-stores the root receipt;
 *   Finalize{max_idx}     (finalize.rs:21-95) reads and verifies the root receipt and stores it under
 *                         receipts/stark/{job}.synthetic (the reference writes receipts/stark/{job}.bincode to S3).
 * Everything these tasks store is tagged BXSYNRCP and lives under synthetic_* keys. */
#define BX_SYNTHETIC_RECEIPT_PATH "synthetic_receipts"
#define BX_SYNTHETIC_ROLLUP_PREFIX "receipts/stark/"
#define BX_SYNTHETIC_ROLLUP_SUFFIX ".synthetic"
/* seed of a stand-in join: FNV-1a (64 bit) over the little-endian bytes of the left seal then the right seal, through splitmix64 */
uint64_t bx_join_seed(const uint32_t* left_seal, size_t left_words, const uint32_t* right_seal, size_t right_words);
#define BX_RECUR_RECEIPT_PATH "recursion_receipts"
/* bx_segment_encode / bx_segment_decode and the segment's layout: bx_prover.h ("the segment on the wire"). */

/* ------------------------------------------------------------------------------------------------- agent ---- */
/* The prover a lane calls.  NULL ops in bx_agent_create = the HIP prover (bx_prove_segment_bytes on the lane's own ctx).
 * A custom table lets tests inject failures without a GPU.  prove returns NULL or an error message.
 * The agent reads the stand-in's header only to route the segment (po2 -> buffer set, index -> receipt header); the stored
 * bytes go to the prover as they are: `prover.prove_segment(&ctx, &segment)` (prove.rs:41-49). */
typedef struct bx_segment_prover_ops {
    void* user;
    size_t (*seal_words)(void* user, uint32_t lane, uint32_t po2); /* seal capacity in words, 0 = unsupported shape */
    const char* (*prove_segment)(void* user, uint32_t lane, uint32_t po2, const uint8_t* segment, size_t segment_len, uint32_t* seal_out,
                                 size_t seal_cap, size_t* seal_words);
    /* Opaque mode (optional, may be NULL): everything tasks::prove::prover does between the GET and the SETEX
     * (prove.rs:36-109: deserialize, prove_segment, verify, lift, verify, serialize) on the raw stored bytes.  *receipt is
     * owned by the prover until free_blob.  When set, seal_words / prove_segment are not used. */
    const char* (*prove_blob)(void* user, uint32_t lane, const uint8_t* segment, size_t len, uint8_t** receipt, size_t* receipt_len);
    void (*free_blob)(void* user, uint8_t* receipt);
} bx_segment_prover_ops;

typedef struct bx_agent_config {
    int32_t device;          /* HIP device ordinal (after HIP_VISIBLE_DEVICES), ignored with custom prover ops */
    uint32_t inflight;       /* prover lanes on this GPU; 0 = default (3) */
    uint32_t w_code, w_data, w_accum; /* synthetic segment group widths; 0 = BASELINE config (16/256/64) */
    uint64_t redis_ttl;      /* seconds; 0 = 8 h (the reference's default `redis_ttl`) */
    double poll_time;        /* idle sleep between empty claims, seconds; <= 0 = 1 s (`poll_time`) */
    int32_t no_verify;       /* 0 = verify each seal before storing it, as the reference does (prove.rs:53-55); 1 = skip.  The HIP
                              * prover's agent verifies against its VerifierContext: the control ID of every buffer set it creates
                              * (bx_prover_control_id, checked against the circuit's own check_code) — `verifier_ctx`, lib.rs:241 */
    char task_stream[64];    /* worker type passed to request_work; "" = "prove" */
    /* ---- one agent, several GPUs: n_devices * inflight lanes claim from the ONE task db (request_work is the work-stealing
     * queue, 9_request_work.sql:126-153); 0 = the single `device` above.  The reference starts one process per GPU
     * (compose.yml:113); here the lanes of all devices live in one process and share nothing but the two callback tables. */
    uint32_t n_devices;
    int32_t devices[16];
    int32_t synthetic;       /* 1 = accept the synthetic wire format and write synthetic_receipts (see above); without it an
                              * agent whose prover table has no prove_blob refuses to start */
    uint32_t cons_terms, cons_degree; /* the synthetic circuit's knobs for the built-in prover (0 = defaults) */
    uint32_t po2_min, po2_max;        /* segment sizes the built-in prover accepts (within 9..24); 0 = 9 / 22.  Others fail the task.  A lane's buffers are 8.5 GB at 2^20, 34 GB at 2^22, 135 GB at 2^24 (16/256/64): size po2_max x inflight for the GPU */
    uint32_t max_shapes;     /* buffer sets (one per segment size, several GB at po2 20) cached per lane, least recently used
                              * evicted; 0 = 2 */
    uint32_t join_po2;       /* size of the stand-in join proofs (see "Stand-ins for the recursion tasks"); 0 = 18, the size of
                              * the reference's recursion proofs */
    char also_streams[128];  /* comma-separated worker types a lane also claims from when its task_stream is empty, in order
                              * (e.g. "aux" to serve the finalize task of a planned job from the same process); "" = none */
    uint32_t lift_po2;       /* 0 = a Prove task stores the segment's own seal (default).  N = the task also runs a STAND-IN for the
                              * `lift` leg of the reference's prove task (prove.rs:60-113: prove_segment -> verify -> lift -> verify ->
                              * store the lifted receipt): one more synthetic proof of 2^N cycles seeded by bx_join_seed(segment seal),
                              * and THAT is what is stored under synthetic_receipts:{task} for the joins to consume.  Like the join
                              * stand-in it is not a recursion proof; it gives a Prove task its real anatomy (two proofs) */
    int32_t prefetch;        /* 1 = every lane runs a fetcher thread that claims the lane's NEXT task and GETs its segment while the lane
                              * proves the current one (pull-when-idle + 2-deep pipelining): with a store behind a
                              * network — an ~80 MB GET over the REST worker protocol — the lane's share of the GPU no longer idles for
                              * the length of a GET.  Costs one task claimed ahead per lane (at the end of a batch a claimed task may wait
                              * for its lane while another lane idles).  0 (default) = claim, fetch, prove, in that order, like the
                              * reference's agent */
    int32_t monitor_requeue; /* 1 = poll_work also runs the requeue monitor (the reference's --monitor-requeue, lib.rs:101-103,283-303):
                              * every requeue_poll_interval seconds, taskdb requeue_tasks(100) sends tasks that stayed 'running' past
                              * their timeout_secs back through update_task_retry, so another lane (another GPU) picks up what a hung
                              * lane or a dead process had claimed.  Needs bx_taskdb_ops::requeue_tasks; 0 (default) = off */
    double requeue_poll_interval; /* seconds; 0 = 5 (lib.rs:148-150) */
    int32_t no_prover;       /* 1 = the agent has no prover, like the reference's agents of every worker type but prove / join / coproc
                              * (`prover: None`, lib.rs:242-252): it needs no GPU, the `prover` argument of bx_agent_create is ignored, the
                              * Resolve / Finalize stand-ins are served (an aux agent beside the GPU agents), and a Prove or Join task that
                              * reaches it fails with "[BENTO-PROVE-002] Missing prover from prove task" / "Missing prover from join task" */
} bx_agent_config;

typedef struct bx_agent bx_agent;
const char* bx_agent_create(const bx_agent_config* cfg, const bx_hot_store_ops* store, const bx_taskdb_ops* taskdb,
                            const bx_segment_prover_ops* prover /* NULL = HIP */, bx_agent** out);
const char* bx_agent_destroy(bx_agent* a);
/* Create, on EVERY lane, the buffer set (and the verifier-context entry) for segments of 2^po2 cycles now instead of at the lane's
 * first such task: a deployment that knows its segment size (`--segment-po2`, lib.rs:61-63) and its join size pays the allocations and
 * the control-ID computation at start-up, like Agent::new creates its prover up front (lib.rs:241-252).  HIP prover only (an injected
 * prover has nothing to create: returns NULL). */
const char* bx_agent_prewarm(bx_agent* a, uint32_t po2);
/* Agent::poll_work: runs the lanes until bx_agent_stop, or until every lane has seen `max_idle_polls` consecutive empty
 * claims (max_idle_polls < 0 = run until stopped).  Blocks.  Task failures are reported to the task db and never end
 * the loop; a failing task-db call does (the reference `?`-returns there: WF-107, WF-109..112, WF-133). */
const char* bx_agent_poll_work(bx_agent* a, int64_t max_idle_polls, uint64_t* tasks_done);
/* The SIGTERM flag of create_sig_monitor (lib.rs:266-270): async-signal-safe, may be called from a signal handler. */
void bx_agent_stop(bx_agent* a);
/* Run one already-claimed task on lane 0 (process_work + the error bookkeeping of poll_work). *ok = 1 when it succeeded. */
const char* bx_agent_process_one(bx_agent* a, const bx_ready_task* task, int* ok);
/* Prometheus text exposition, with the reference's names, labels, help strings and buckets, of
 *   task_operations_total / task_duration_seconds, redis_operations_total / redis_operation_duration_seconds
 *       (bento/crates/workflow-common/src/metrics.rs:60-120), and
 *   task_claims_total{task_stream,result}, task_processing_total / task_processing_end_to_end_seconds{task_type,status},
 *   task_retry_attempts_total{task_type}, task_max_retries_exhausted_total{task_type}
 *       (the next-gen worker loop: prover/crates/workflow-common/src/metrics.rs:44-80, recorded as in
 *       prover/crates/workflow/src/lib.rs:613-675; task_type is TaskType::to_job_type_str, "prove-lift" for a Prove task).
 * Returns the needed size. */
size_t bx_agent_metrics(bx_agent* a, char* out, size_t cap);
/* Lanes of the agent (n_devices * inflight; lane l belongs to devices[l / inflight]), the device of a lane (informational
 * with an injected prover, which receives the lane index) and how many tasks the lane completed. */
uint32_t bx_agent_lane_count(const bx_agent* a);
int32_t bx_agent_lane_device(const bx_agent* a, uint32_t lane);
uint64_t bx_agent_lane_tasks_done(const bx_agent* a, uint32_t lane);

#if defined(__GNUC__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif
