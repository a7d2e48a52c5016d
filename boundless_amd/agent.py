"""Prove-stream agent — ctypes binding of the native feed loop (boundless_amd/csrc/agent.cpp, include/bx_agent.h).

What it mirrors, for the `prove` task stream only (all logic is C++; this file marshals and adapts Python callables):
  * `Agent::poll_work` / `process_work`   bento/crates/workflow/src/lib.rs:279-442, 445-530
  * `tasks::prove::prover`                bento/crates/workflow/src/tasks/prove.rs:18-135
  * key scheme                            bento/crates/workflow/src/tasks/mod.rs:23-29
  * task JSON                             bento/crates/workflow-common/src/lib.rs:88-92,160-178 (`{"Prove":{"index":n}}`)
  * metrics                               bento/crates/workflow-common/src/metrics.rs:61-70,108-117,288-335
The reference's Redis/Postgres clients are out of scope; the library's in-memory `HotStore` / `TaskDb` stand in for them.
"""
import ctypes as C
import json
import struct

import numpy as np

from .hal import HalError, load_library
from .prover import Segment, SegmentReceipt

RECUR_RECEIPT_PATH = "recursion_receipts"  # the reference's key (tasks/mod.rs:23): written only by an opaque (real) prover
SYNTHETIC_RECEIPT_PATH = "synthetic_receipts"  # where seals of the synthetic circuit go (include/bx_agent.h)
SEGMENTS_PATH = "segments"
TASK_STATES = ("ready", "running", "done", "failed", "pending")


class _HotStoreOps(C.Structure):
    _fields_ = [("user", C.c_void_p), ("get", C.c_void_p), ("free_value", C.c_void_p), ("set_ex", C.c_void_p),
                ("unlink", C.c_void_p)]


class _TaskDbOps(C.Structure):
    _fields_ = [("user", C.c_void_p), ("request_work", C.c_void_p), ("update_task_done", C.c_void_p),
                ("update_task_failed", C.c_void_p), ("update_task_retry", C.c_void_p), ("current_retries", C.c_void_p),
                ("requeue_tasks", C.c_void_p)]


_SEAL_WORDS_FN = C.CFUNCTYPE(C.c_size_t, C.c_void_p, C.c_uint32, C.c_uint32)
_PROVE_FN = C.CFUNCTYPE(C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint8), C.c_size_t, C.POINTER(C.c_uint32),
                        C.c_size_t, C.POINTER(C.c_size_t))


_PROVE_BLOB_FN = C.CFUNCTYPE(C.c_void_p, C.c_void_p, C.c_uint32, C.POINTER(C.c_uint8), C.c_size_t, C.POINTER(C.POINTER(C.c_uint8)),
                             C.POINTER(C.c_size_t))
_FREE_BLOB_FN = C.CFUNCTYPE(None, C.c_void_p, C.POINTER(C.c_uint8))


class _ProverOps(C.Structure):
    _fields_ = [("user", C.c_void_p), ("seal_words", _SEAL_WORDS_FN), ("prove_segment", _PROVE_FN),
                ("prove_blob", _PROVE_BLOB_FN), ("free_blob", _FREE_BLOB_FN)]


class _ReadyTask(C.Structure):
    _fields_ = [("job_id", C.c_char * 40), ("task_id", C.c_char * 128), ("task_def", C.c_char * 1024),
                ("max_retries", C.c_int32)]


class _TaskInfo(C.Structure):
    _fields_ = [("state", C.c_int32), ("retries", C.c_int32), ("max_retries", C.c_int32), ("error", C.c_char * 1100),
                ("output", C.c_char * 256), ("waiting_on", C.c_int32), ("timeout_secs", C.c_int32), ("created_s", C.c_double), ("started_s", C.c_double),
                ("updated_s", C.c_double)]


class _JobInfo(C.Structure):
    _fields_ = [("state", C.c_int32), ("tasks", C.c_uint64), ("pending", C.c_uint64), ("ready", C.c_uint64), ("running", C.c_uint64),
                ("done", C.c_uint64), ("failed", C.c_uint64), ("error", C.c_char * 1100)]


class _JobPlan(C.Structure):
    _fields_ = [("prove_stream", C.c_char * 64), ("join_stream", C.c_char * 64), ("aux_stream", C.c_char * 64), ("prove_retries", C.c_int32),
                ("join_retries", C.c_int32), ("resolve_retries", C.c_int32), ("finalize_retries", C.c_int32), ("subtree_only", C.c_int32),
                ("prove_timeout", C.c_int32), ("join_timeout", C.c_int32), ("resolve_timeout", C.c_int32), ("finalize_timeout", C.c_int32)]


class _AgentConfig(C.Structure):
    _fields_ = [("device", C.c_int32), ("inflight", C.c_uint32), ("w_code", C.c_uint32), ("w_data", C.c_uint32),
                ("w_accum", C.c_uint32), ("redis_ttl", C.c_uint64), ("poll_time", C.c_double), ("no_verify", C.c_int32),
                ("task_stream", C.c_char * 64), ("n_devices", C.c_uint32), ("devices", C.c_int32 * 16), ("synthetic", C.c_int32),
                ("cons_terms", C.c_uint32), ("cons_degree", C.c_uint32), ("po2_min", C.c_uint32), ("po2_max", C.c_uint32),
                ("max_shapes", C.c_uint32), ("join_po2", C.c_uint32), ("also_streams", C.c_char * 128), ("lift_po2", C.c_uint32), ("prefetch", C.c_int32), ("monitor_requeue", C.c_int32),
                ("requeue_poll_interval", C.c_double), ("no_prover", C.c_int32)]


def _lib():
    lib = load_library()
    if getattr(lib, "_bx_agent_declared", False):
        return lib
    vp, cp, sz = C.c_void_p, C.c_char_p, C.c_size_t
    sigs = {
        "bx_mem_store_create": ([C.POINTER(vp)], cp), "bx_mem_store_destroy": ([vp], None),
        "bx_mem_store_ops": ([vp], _HotStoreOps), "bx_mem_store_key_count": ([vp], sz),
        "bx_mem_store_keys": ([vp, cp, sz], cp),
        "bx_mem_taskdb_create": ([C.POINTER(vp)], cp), "bx_mem_taskdb_destroy": ([vp], None),
        "bx_mem_taskdb_ops": ([vp], _TaskDbOps),
        "bx_mem_taskdb_create_task": ([vp, cp, cp, cp, cp, C.c_int32], cp),
        "bx_mem_taskdb_task_info": ([vp, cp, cp, C.POINTER(_TaskInfo)], cp), "bx_mem_taskdb_count": ([vp, C.c_int32], sz),
        "bx_mem_taskdb_clear_completed_jobs": ([vp, C.POINTER(C.c_uint64)], cp),
        "bx_mem_taskdb_create_task_ex": ([vp, cp, cp, cp, cp, C.POINTER(C.c_char_p), sz, C.c_int32, C.c_int32], cp),
        "bx_mem_taskdb_requeue_tasks": ([vp, C.c_int64, C.POINTER(C.c_uint64)], cp),
        "bx_mem_taskdb_advance_clock": ([vp, C.c_double], cp),
        "bx_mem_taskdb_create_task_with_prereqs": ([vp, cp, cp, cp, cp, C.POINTER(cp), sz, C.c_int32], cp),
        "bx_mem_taskdb_job_info": ([vp, cp, C.POINTER(_JobInfo)], cp),
        "bx_plan_job": ([vp, cp, C.c_uint64, C.POINTER(_JobPlan), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)], cp),
        "bx_join_seed": ([vp, sz, vp, sz], C.c_uint64),
        "bx_segment_encode": ([C.c_uint64, C.c_uint32, C.c_uint64, vp], None),
        "bx_segment_decode": ([vp, sz, C.POINTER(C.c_uint64), C.POINTER(C.c_uint32), C.POINTER(C.c_uint64)], cp),
        "bx_agent_create": ([C.POINTER(_AgentConfig), C.POINTER(_HotStoreOps), C.POINTER(_TaskDbOps), vp, C.POINTER(vp)], cp),
        "bx_agent_destroy": ([vp], cp), "bx_agent_poll_work": ([vp, C.c_int64, C.POINTER(C.c_uint64)], cp),
        "bx_agent_stop": ([vp], None), "bx_agent_process_one": ([vp, C.POINTER(_ReadyTask), C.POINTER(C.c_int)], cp),
        "bx_agent_metrics": ([vp, cp, sz], sz),
        "bx_agent_prewarm": ([vp, C.c_uint32], cp),
        "bx_rest_client_create": ([cp, C.c_uint64, C.c_uint64, C.POINTER(vp)], cp), "bx_rest_client_destroy": ([vp], None),
        "bx_rest_taskdb_ops": ([vp], _TaskDbOps), "bx_rest_hot_store_ops": ([vp], _HotStoreOps),
        "bx_rest_client_requests": ([vp], C.c_uint64),
        "bx_rest_client_connects": ([vp], C.c_uint64),
        "bx_agent_lane_count": ([vp], C.c_uint32), "bx_agent_lane_device": ([vp, C.c_uint32], C.c_int32),
        "bx_agent_lane_tasks_done": ([vp, C.c_uint32], C.c_uint64),
    }
    for name, (args, res) in sigs.items():
        fn = getattr(lib, name)
        fn.argtypes, fn.restype = args, res
    lib._bx_agent_declared = True
    return lib


def _check(msg):
    if msg:
        raise HalError(msg.decode())


# ---------------------------------------------------------------------------------------------------------------- wire
def serialize_segment(seg: Segment) -> bytes:
    """Stand-in for `bincode(risc0_zkvm::Segment)` (tasks/mod.rs:40-47): bx_segment_encode."""
    out = (C.c_uint8 * 28)()
    _lib().bx_segment_encode(seg.index, seg.po2, seg.seed & (2**64 - 1), out)
    return bytes(out) + bytes(seg.payload)  # the payload (the stand-in's "preflight trace") follows the 28-byte header


def deserialize_segment(blob: bytes) -> Segment:
    i, p, s = C.c_uint64(), C.c_uint32(), C.c_uint64()
    buf = (C.c_uint8 * max(len(blob), 1)).from_buffer_copy(blob or b"\0")
    msg = _lib().bx_segment_decode(buf, len(blob), C.byref(i), C.byref(p), C.byref(s))
    if msg:
        raise ValueError(msg.decode())
    return Segment(index=i.value, po2=p.value, seed=s.value, payload=bytes(blob[28:]))


def serialize_receipt(r: SegmentReceipt) -> bytes:
    """"BXSYNRCP" | index u64 | po2 u32 | seal_words u32 | seal (include/bx_agent.h)"""
    seal = np.ascontiguousarray(r.seal, dtype="<u4")
    return (b"BXSYNRCP" + int(r.index).to_bytes(8, "little") + int(r.po2).to_bytes(4, "little") + int(seal.size).to_bytes(4, "little")
            + seal.tobytes())


def join_seed(left_seal, right_seal):
    """bx_join_seed: the seed of a stand-in join (include/bx_agent.h)."""
    a = np.ascontiguousarray(left_seal, dtype=np.uint32)
    b = np.ascontiguousarray(right_seal, dtype=np.uint32)
    return _lib().bx_join_seed(a.ctypes.data, a.size, b.ctypes.data, b.size)


def deserialize_receipt(blob: bytes) -> SegmentReceipt:
    """"BXSYNRCP" | index u64 | po2 u32 | seal_words u32 | seal (include/bx_agent.h)"""
    if blob[:8] != b"BXSYNRCP":
        raise ValueError("not a synthetic receipt blob")
    index, po2, n = struct.unpack_from("<QII", blob, 8)
    return SegmentReceipt(seal=np.frombuffer(blob, dtype="<u4", count=n, offset=24).copy(), index=index, po2=po2)


# -------------------------------------------------------------------------------------------------------------- stores
class HotStore:
    """The library's in-memory hot store (GET / SETEX / UNLINK, workflow/src/redis.rs:19-63)."""

    def __init__(self):
        self._lib = _lib()
        self._h = C.c_void_p()
        _check(self._lib.bx_mem_store_create(C.byref(self._h)))
        self.ops = self._lib.bx_mem_store_ops(self._h)
        self._get = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_char_p, C.POINTER(C.POINTER(C.c_uint8)), C.POINTER(C.c_size_t),
                                C.c_char_p, C.c_size_t)(self.ops.get)
        self._free = C.CFUNCTYPE(None, C.c_void_p, C.POINTER(C.c_uint8))(self.ops.free_value)
        self._set = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_char_p, C.c_char_p, C.c_size_t, C.c_uint64, C.c_char_p,
                                C.c_size_t)(self.ops.set_ex)
        self._unlink = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_char_p, C.c_char_p, C.c_size_t)(self.ops.unlink)

    def get(self, key):
        v, n = C.POINTER(C.c_uint8)(), C.c_size_t()
        rc = self._get(self._h, key.encode(), C.byref(v), C.byref(n), None, 0)
        if rc != 0:
            raise KeyError(key)
        out = C.string_at(v, n.value)
        self._free(self._h, v)
        return out

    def set_key_with_expiry(self, key, value, ttl_secs=None):
        self._set(self._h, key.encode(), bytes(value), len(value), int(ttl_secs or 0), None, 0)

    def unlink(self, key):
        self._unlink(self._h, key.encode(), None, 0)

    def keys(self):
        buf = C.create_string_buffer(1 << 16)
        _check(self._lib.bx_mem_store_keys(self._h, buf, len(buf)))
        return [k for k in buf.value.decode().split("\n") if k]

    def __del__(self):
        try:
            self._lib.bx_mem_store_destroy(self._h)
        except Exception:
            pass


class TaskRow:
    def __init__(self, job_id, task_id, info):
        self.job_id, self.task_id = job_id, task_id
        self.state = TASK_STATES[info.state]
        self.retries, self.max_retries = info.retries, info.max_retries
        self.error, self.output = info.error.decode(), info.output.decode()
        self.waiting_on = info.waiting_on
        self.timeout_secs = info.timeout_secs
        self.created_s, self.started_s, self.updated_s = info.created_s, info.started_s, info.updated_s


class TaskDb:
    """The library's in-memory task table: request_work / update_task_done|failed|retry with the state transitions of
    bento/crates/taskdb/migrations/1_taskdb.sql:308-391."""

    def __init__(self):
        self._lib = _lib()
        self._h = C.c_void_p()
        _check(self._lib.bx_mem_taskdb_create(C.byref(self._h)))
        self.ops = self._lib.bx_mem_taskdb_ops(self._h)
        self._ids = []

    def create_task(self, job_id, task_id, task_def, max_retries=3, stream="prove", prerequisites=(), timeout_secs=None):
        """taskdb::create_task (1_taskdb.sql:197-228): 'pending' while a prerequisite (task ids of the same job) is not done.
        timeout_secs: how long the task may stay running before `requeue_tasks` retries it (None = never)."""
        d = task_def if isinstance(task_def, str) else json.dumps(task_def)
        pre = [str(p).encode() for p in prerequisites]
        arr = (C.c_char_p * max(len(pre), 1))(*pre)
        _check(self._lib.bx_mem_taskdb_create_task_ex(self._h, stream.encode(), str(job_id).encode(), str(task_id).encode(), d.encode(), arr,
                                                      len(pre), max_retries, 0x7FFFFFFF if timeout_secs is None else int(timeout_secs)))
        self._ids.append((str(job_id), str(task_id)))

    def requeue_tasks(self, limit=100):
        """taskdb::requeue_tasks (bento/crates/taskdb/src/lib.rs:328-358): running tasks past their timeout go through update_task_retry;
        returns how many had timed out."""
        n = C.c_uint64()
        _check(self._lib.bx_mem_taskdb_requeue_tasks(self._h, limit, C.byref(n)))
        return n.value

    def advance_clock(self, seconds):
        """Test hook: the table's clock jumps forward."""
        _check(self._lib.bx_mem_taskdb_advance_clock(self._h, float(seconds)))

    def plan_job(self, job_id, n_segments, prove_stream="", join_stream="", aux_stream="", retries=3, subtree_only=False, timeouts=(0, 0, 0, 0)):
        """bx_plan_job: the executor's planner loop (executor.rs:566-698) — one task row per planner task, prerequisites as the
        planner's dependencies.  Returns the task ids created, in creation order; `self.root_task` = the task whose receipt is the
        job's root.  subtree_only: stop at the root join (no resolve / finalize): one GPU's share of a larger job."""
        plan = _JobPlan(prove_stream.encode(), join_stream.encode(), aux_stream.encode(), retries, retries, retries, retries, int(subtree_only),
                        *timeouts)  # prove, join, resolve, finalize timeout_secs; 0 = the reference's defaults 30 / 10 / 120 / 10
        n, root = C.c_uint64(), C.c_uint64()
        _check(self._lib.bx_plan_job(self._h, str(job_id).encode(), n_segments, C.byref(plan), C.byref(n), C.byref(root)))
        self.root_task = root.value
        ids = [str(i) for i in range(n.value)] if subtree_only else [str(i) for i in range(n.value - 2)] + ["resolve", "finalize"]
        self._ids += [(str(job_id), t) for t in ids]
        return ids

    def update_task_done(self, job_id, task_id, output="null"):
        """taskdb::update_task_done (1_taskdb.sql:278-314): marks the task done and releases what waited on it."""
        fn = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_size_t)(self.ops.update_task_done)
        return fn(self.ops.user, str(job_id).encode(), str(task_id).encode(), output.encode(), None, 0) == 1

    def job(self, job_id):
        info = _JobInfo()
        _check(self._lib.bx_mem_taskdb_job_info(self._h, str(job_id).encode(), C.byref(info)))
        return {"state": ("running", "done", "failed")[info.state], "tasks": info.tasks, "pending": info.pending, "ready": info.ready,
                "running": info.running, "done": info.done, "failed": info.failed, "error": info.error.decode()}

    def task(self, job_id, task_id):
        info = _TaskInfo()
        _check(self._lib.bx_mem_taskdb_task_info(self._h, str(job_id).encode(), str(task_id).encode(), C.byref(info)))
        return TaskRow(str(job_id), str(task_id), info)

    def rows(self):
        return [self.task(j, t) for j, t in self._ids]

    def count(self, state):
        return self._lib.bx_mem_taskdb_count(self._h, TASK_STATES.index(state))

    def clear_completed_jobs(self):
        """taskdb `clear_completed_jobs` (4_clear_completed_streams.sql): drops every row of every done job; returns the jobs cleared."""
        n = C.c_uint64()
        _check(self._lib.bx_mem_taskdb_clear_completed_jobs(self._h, C.byref(n)))
        if n.value:  # rows() lists what is still in the table
            alive, info = {}, _JobInfo()
            for j, _ in self._ids:
                if j not in alive:
                    alive[j] = self._lib.bx_mem_taskdb_job_info(self._h, j.encode(), C.byref(info)) is None
            self._ids = [(j, t) for j, t in self._ids if alive[j]]
        return n.value

    def __del__(self):
        try:
            self._lib.bx_mem_taskdb_destroy(self._h)
        except Exception:
            pass


# ---------------------------------------------------------------------------------------------------- REST worker tables
class _Ops:
    def __init__(self, ops):
        self.ops = ops


class RestWorker:
    """The next-generation Bento worker protocol (prover/crates/api/src/lib.rs:922-1040; client prover/crates/workflow/src/assets.rs)
    as the agent's two callback tables (include/bx_rest.h): `Agent(store=w.store, taskdb=w.taskdb, ...)` makes the native feed
    loop claim tasks from and move blobs through a Bento API over HTTP."""

    def __init__(self, base_url, claim_wait_secs=0, io_timeout_secs=0):
        self._lib = _lib()
        self._h = C.c_void_p()
        _check(self._lib.bx_rest_client_create(base_url.encode(), claim_wait_secs, io_timeout_secs, C.byref(self._h)))
        self.store = _Ops(self._lib.bx_rest_hot_store_ops(self._h))
        self.taskdb = _Ops(self._lib.bx_rest_taskdb_ops(self._h))

    @property
    def requests(self):
        return self._lib.bx_rest_client_requests(self._h)

    @property
    def connects(self):
        """TCP connections opened so far: connections are kept alive and pooled, so this stays near the number of lanes."""
        return self._lib.bx_rest_client_connects(self._h)

    def close(self):
        if getattr(self, "_h", None):
            self._lib.bx_rest_client_destroy(self._h)
            self._h = None


# --------------------------------------------------------------------------------------------------------------- agent
class Agent:
    """One per process (lib.rs:180-195).  `prover=None` = the HIP segment prover with `inflight` lanes on `device`, or on each
    of `devices` (one agent, several GPUs, one shared task db = the work-stealing queue).  A Python object with
    `prove_segment(Segment) -> SegmentReceipt` may be injected instead (tests, no GPU); `blob_prover` (bytes -> bytes) is the
    opaque mode a real prover would use (reference keys).  The built-in and injected provers speak the SYNTHETIC wire format
    and need `synthetic=True` (the default here; the C ABI's default is off, see include/bx_agent.h)."""

    def __init__(self, prover=None, device=0, inflight=None, widths=(16, 256, 64), redis_ttl=8 * 60 * 60, poll_time=1.0,
                 verify=True, store=None, taskdb=None, task_stream="prove", seal_cap=1 << 20, devices=None, synthetic=True,
                 terms=0, degree=0, po2_range=(0, 0), max_shapes=0, blob_prover=None, join_po2=0, also_streams="", lift_po2=0, prefetch=False,
                 monitor_requeue=False, requeue_poll_interval=0.0, no_prover=False):
        self._lib = _lib()
        self.store = store or HotStore()
        self.taskdb = taskdb or TaskDb()
        self.prover = prover
        injected = prover is not None or blob_prover is not None or no_prover
        cfg = _AgentConfig(device=device, inflight=inflight or (1 if injected else 3), w_code=widths[0],
                           w_data=widths[1], w_accum=widths[2], redis_ttl=redis_ttl, poll_time=poll_time, no_verify=int(not verify),
                           task_stream=task_stream.encode(), synthetic=int(bool(synthetic)), cons_terms=terms, cons_degree=degree,
                           po2_min=po2_range[0], po2_max=po2_range[1], max_shapes=max_shapes, join_po2=join_po2,
                           also_streams=also_streams.encode(), lift_po2=lift_po2, prefetch=int(bool(prefetch)),
                           monitor_requeue=int(bool(monitor_requeue)), requeue_poll_interval=requeue_poll_interval,
                           no_prover=int(bool(no_prover)))
        if devices:
            cfg.n_devices = len(devices)
            for i, d in enumerate(devices):
                cfg.devices[i] = d
        self._errs = {}
        self._blobs = {}
        ops_ptr = None
        if blob_prover is not None:
            def prove_blob(_user, lane, seg, n, out, out_len):
                try:
                    rec = bytes(blob_prover(bytes(bytearray(seg[:n]))))
                    buf = (C.c_uint8 * max(len(rec), 1)).from_buffer_copy(rec or b"\0")
                    self._blobs[lane] = buf  # owned here until free_blob
                    out[0] = C.cast(buf, C.POINTER(C.c_uint8))
                    out_len[0] = len(rec)
                    return None
                except Exception as e:  # noqa: BLE001
                    self._errs[lane] = C.create_string_buffer(f"{e}".encode("utf-8", "surrogateescape"))
                    return C.cast(self._errs[lane], C.c_void_p).value

            def free_blob(_user, _ptr):
                return None

            self._ops = _ProverOps(None, _SEAL_WORDS_FN(), _PROVE_FN(), _PROVE_BLOB_FN(prove_blob), _FREE_BLOB_FN(free_blob))
            ops_ptr = C.cast(C.pointer(self._ops), C.c_void_p)
        elif prover is not None:
            def seal_words(_user, _lane, _po2):
                return seal_cap

            def prove(_user, lane, po2, seg, seg_len, seal_out, cap, words):
                try:
                    r = prover.prove_segment(Segment.from_bytes(C.string_at(seg, seg_len)))  # the stored bytes, as the prover gets them
                    seal = np.ascontiguousarray(r.seal, dtype=np.uint32)
                    if seal.size > cap:
                        raise HalError("seal does not fit")
                    C.memmove(seal_out, seal.ctypes.data, seal.nbytes)
                    words[0] = seal.size
                    return None
                except Exception as e:  # noqa: BLE001 - the error crosses the ABI as a string, like the HIP prover's
                    self._errs[lane] = C.create_string_buffer(f"{e}".encode("utf-8", "surrogateescape"))
                    return C.cast(self._errs[lane], C.c_void_p).value

            self._ops = _ProverOps(None, _SEAL_WORDS_FN(seal_words), _PROVE_FN(prove), _PROVE_BLOB_FN(), _FREE_BLOB_FN())
            ops_ptr = C.cast(C.pointer(self._ops), C.c_void_p)
        self._h = C.c_void_p()
        _check(self._lib.bx_agent_create(C.byref(cfg), C.byref(self.store.ops), C.byref(self.taskdb.ops), ops_ptr,
                                         C.byref(self._h)))

    def prewarm(self, po2):
        """bx_agent_prewarm: every lane creates its buffer set for 2^po2-cycle segments now (HIP prover only)."""
        _check(self._lib.bx_agent_prewarm(self._h, po2))

    def lane_stats(self):
        """[(device, tasks completed)] per lane: which GPU's lanes claimed how much of the shared queue."""
        n = self._lib.bx_agent_lane_count(self._h)
        return [(self._lib.bx_agent_lane_device(self._h, i), self._lib.bx_agent_lane_tasks_done(self._h, i)) for i in range(n)]

    def poll_work(self, max_idle_polls=None):
        """`Agent::poll_work`: returns the number of tasks completed; raises only when the task db itself fails."""
        done = C.c_uint64()
        _check(self._lib.bx_agent_poll_work(self._h, -1 if max_idle_polls is None else max_idle_polls, C.byref(done)))
        return done.value

    def stop(self):
        self._lib.bx_agent_stop(self._h)

    def process_one(self, job_id, task_id, task_def, max_retries=0):
        """process_work + poll_work's error bookkeeping for one claimed task; returns True when the task succeeded."""
        d = task_def if isinstance(task_def, str) else json.dumps(task_def)
        t = _ReadyTask(str(job_id).encode(), str(task_id).encode(), d.encode(), max_retries)
        ok = C.c_int()
        _check(self._lib.bx_agent_process_one(self._h, C.byref(t), C.byref(ok)))
        return bool(ok.value)

    def metrics_text(self):
        n = self._lib.bx_agent_metrics(self._h, None, 0)
        buf = C.create_string_buffer(n)
        self._lib.bx_agent_metrics(self._h, buf, n)
        return buf.value.decode()

    def close(self):
        if getattr(self, "_h", None):
            self._lib.bx_agent_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
