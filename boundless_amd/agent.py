"""Agent-equivalent feed loop for the prove stream.

Mirrors, for the `prove` task stream only, what a Bento agent process does around the hot path:
  * `tasks::prove::prover`      bento/crates/workflow/src/tasks/prove.rs:18-135  -> `prove_task`
  * key scheme                  bento/crates/workflow/src/tasks/mod.rs:23-29     -> SEGMENTS_PATH / RECUR_RECEIPT_PATH
  * task JSON                   bento/crates/workflow-common/src/lib.rs:88-92,160-178 (`{"Prove":{"index":n}}`)
  * poll / process / retry      bento/crates/workflow/src/lib.rs:369-438,445-530 (claim -> run -> done | retry | failed)
  * metrics                     bento/crates/workflow-common/src/metrics.rs:108-117,323-335
    (`task_operations_total`, `task_duration_seconds{task_name,operation_type,status}`)
The reference's stores (Redis hot store, Postgres taskdb) are control plane and out of scope; they appear here only as
the minimal in-process interfaces the loop needs (`HotStore`, `TaskStream`).  The `lift` step (recursion circuit) is not
reproducible offline, so the verified segment receipt itself is stored under the recursion-receipt key.
"""
import json
import struct
import threading
import time
from dataclasses import dataclass, field

import numpy as np

from .prover import Segment, SegmentReceipt

RECUR_RECEIPT_PATH = "recursion_receipts"
SEGMENTS_PATH = "segments"
RECEIPT_PATH = "receipts"
_DURATION_BUCKETS = (0.1, 0.5, 1.0, 2.5, 5.0, 10.0, 25.0, 50.0, 100.0, 250.0, 500.0)


# ---------------------------------------------------------------------------------------------------------------- wire
def serialize_segment(seg: Segment) -> bytes:
    """Stand-in for `bincode(risc0_zkvm::Segment)` (tasks/mod.rs:40-47): index u64 | po2 u32 | seed u64, little endian."""
    return struct.pack("<QIQ", seg.index, seg.po2, seg.seed & (2**64 - 1))


def deserialize_segment(blob: bytes) -> Segment:
    try:
        index, po2, seed = struct.unpack("<QIQ", blob)
    except struct.error as e:
        raise ValueError("Failed to deserialize segment data from redis") from e
    return Segment(index=index, po2=po2, seed=seed)


def serialize_receipt(r: SegmentReceipt) -> bytes:
    head = struct.pack("<QII", r.index, r.po2, r.seal.size)
    return head + np.ascontiguousarray(r.seal, dtype="<u4").tobytes()


def deserialize_receipt(blob: bytes) -> SegmentReceipt:
    index, po2, n = struct.unpack_from("<QII", blob)
    seal = np.frombuffer(blob, dtype="<u4", count=n, offset=16).copy()
    return SegmentReceipt(seal=seal, index=index, po2=po2)


def parse_task(task_def: str):
    """`serde_json::from_value::<TaskType>` for the variants this agent serves."""
    d = json.loads(task_def) if isinstance(task_def, str) else task_def
    if not isinstance(d, dict) or len(d) != 1:
        raise ValueError("Invalid task_def")
    (kind, body), = d.items()
    if kind == "Prove":
        return "prove", ProveReq(index=int(body["index"]))
    raise ValueError(f"task type {kind} is not served by the prove agent")


def job_type_str(kind):
    return {"prove": "prove-lift"}[kind]  # TaskType::to_job_type_str


@dataclass
class ProveReq:
    index: int


# ---------------------------------------------------------------------------------------------------------------- stores
class HotStore:
    """The four Redis operations the prove task uses (workflow/src/redis.rs:19-63): GET, SETEX, UNLINK (+ SET)."""

    def __init__(self):
        self._d = {}
        self._lock = threading.Lock()

    def get(self, key):
        with self._lock:
            v = self._d.get(key)
            if v is None:
                raise KeyError(key)
            value, expires = v
            if expires is not None and expires < time.monotonic():
                del self._d[key]
                raise KeyError(key)
            return value

    def set_key_with_expiry(self, key, value, ttl_secs=None):
        with self._lock:
            self._d[key] = (bytes(value), None if ttl_secs is None else time.monotonic() + ttl_secs)

    def unlink(self, key):
        with self._lock:
            self._d.pop(key, None)

    def keys(self):
        with self._lock:
            return sorted(self._d)


@dataclass
class _TaskRow:
    job_id: str
    task_id: str
    task_def: str
    max_retries: int = 3
    retries: int = 0
    state: str = "ready"  # ready | running | done | failed
    error: str = ""


class TaskStream:
    """Claim-when-idle task stream with retry bookkeeping (the `request_work` / `update_task_*` calls of lib.rs:369-438)."""

    def __init__(self):
        self._rows = []
        self._lock = threading.Lock()

    def create_task(self, job_id, task_id, task_def, max_retries=3):
        with self._lock:
            self._rows.append(_TaskRow(str(job_id), str(task_id), json.dumps(task_def) if not isinstance(task_def, str) else task_def,
                                       max_retries))

    def request_work(self):
        with self._lock:
            for r in self._rows:
                if r.state == "ready":
                    r.state = "running"
                    return r
        return None

    def update_task_done(self, row):
        with self._lock:
            row.state = "done"

    def update_task_retry(self, row, err):
        """Returns True if the task was requeued, False if it ran out of retries and was failed."""
        with self._lock:
            row.error = err
            if row.retries < row.max_retries:
                row.retries += 1
                row.state = "ready"
                return True
            row.state = "failed"
            return False

    def rows(self):
        with self._lock:
            return list(self._rows)


# ---------------------------------------------------------------------------------------------------------------- metrics
class Metrics:
    """`task_operations_total` + `task_duration_seconds` with the reference's labels and buckets."""

    def __init__(self):
        self.ops = {}
        self.hist = {}
        self._lock = threading.Lock()

    def record_task_operation(self, task_name, operation_type, status, seconds):
        key = (task_name, operation_type, status)
        with self._lock:
            self.ops[key] = self.ops.get(key, 0) + 1
            h = self.hist.setdefault(key, {"buckets": [0] * len(_DURATION_BUCKETS), "sum": 0.0, "count": 0})
            for i, le in enumerate(_DURATION_BUCKETS):
                if seconds <= le:
                    h["buckets"][i] += 1
            h["sum"] += seconds
            h["count"] += 1

    record_task = record_task_operation

    def exposition(self):
        """Prometheus text format (what the agent's exporter serves on PROMETHEUS_METRICS_ADDR)."""
        out = ["# TYPE task_operations_total counter"]
        lab = lambda k: f'task_name="{k[0]}",operation_type="{k[1]}",status="{k[2]}"'
        for k, v in sorted(self.ops.items()):
            out.append(f"task_operations_total{{{lab(k)}}} {v}")
        out.append("# TYPE task_duration_seconds histogram")
        for k, h in sorted(self.hist.items()):
            for le, n in zip(_DURATION_BUCKETS, h["buckets"]):
                out.append(f'task_duration_seconds_bucket{{{lab(k)},le="{le}"}} {n}')
            out.append(f'task_duration_seconds_bucket{{{lab(k)},le="+Inf"}} {h["count"]}')
            out.append(f"task_duration_seconds_sum{{{lab(k)}}} {h['sum']:.6f}")
            out.append(f"task_duration_seconds_count{{{lab(k)}}} {h['count']}")
        return "\n".join(out) + "\n"


# ---------------------------------------------------------------------------------------------------------------- agent
@dataclass
class Agent:
    """One per process / GPU (lib.rs:180-195): holds the prover object for the process lifetime."""

    prover: object  # anything with prove_segment(Segment) -> SegmentReceipt (HipProverServer in production)
    store: HotStore = field(default_factory=HotStore)
    stream: TaskStream = field(default_factory=TaskStream)
    metrics: Metrics = field(default_factory=Metrics)
    redis_ttl: int = 8 * 60 * 60
    verify: bool = True
    poll_time: float = 1.0


def prove_task(agent: Agent, job_id, task_id, request: ProveReq):
    """`tasks::prove::prover` (prove.rs:18-135): fetch -> deserialize -> prove -> verify -> store -> cleanup."""
    start = time.perf_counter()
    job_prefix = f"job:{job_id}"
    segment_key = f"{job_prefix}:{SEGMENTS_PATH}:{request.index}"
    try:
        blob = agent.store.get(segment_key)
    except KeyError as e:
        raise RuntimeError(f"segment data not found for segment key: {segment_key}") from e
    segment = deserialize_segment(blob)
    if agent.prover is None:
        raise RuntimeError("[BENTO-PROVE-002] Missing prover from prove task")
    t0 = time.perf_counter()
    receipt = agent.prover.prove_segment(segment)
    dt = time.perf_counter() - t0
    agent.metrics.record_task_operation("prove", "prove_segment", "success", dt)
    if agent.verify:
        try:
            receipt.verify_integrity()
        except Exception as e:
            raise RuntimeError(f"[BENTO-PROVE-004] Failed to verify segment receipt integrity: {e}") from e
    agent.metrics.record_task("prove", "prove_segment", "success", dt)
    output_key = f"{job_prefix}:{RECUR_RECEIPT_PATH}:{task_id}"
    agent.store.set_key_with_expiry(output_key, serialize_receipt(receipt), agent.redis_ttl)
    agent.store.unlink(segment_key)
    agent.metrics.record_task_operation("prove", "complete", "success", time.perf_counter() - start)


def process_work(agent: Agent, row):
    """`Agent::process_work` (lib.rs:445-530) for the prove stream."""
    kind, req = parse_task(row.task_def)
    if kind == "prove":
        prove_task(agent, row.job_id, row.task_id, req)
    agent.stream.update_task_done(row)


def poll_work(agent: Agent, stop=None, max_idle_polls=None):
    """`Agent::poll_work` (lib.rs:279-442): claim, run, mark done; on error retry up to max_retries, then fail the task.
    A failing task never takes the agent down.  Returns the number of tasks completed."""
    done = 0
    idle = 0
    while stop is None or not stop.is_set():
        row = agent.stream.request_work()
        if row is None:
            idle += 1
            if max_idle_polls is not None and idle >= max_idle_polls:
                break
            time.sleep(agent.poll_time)
            continue
        idle = 0
        try:
            process_work(agent, row)
            done += 1
        except Exception as e:  # noqa: BLE001 - mirrors the agent's catch-all around process_work
            requeued = agent.stream.update_task_retry(row, f"{type(e).__name__}: {e}")
            agent.metrics.record_task_operation("prove", "complete", "retry" if requeued else "failed", 0.0)
    return done
