"""CPU: the planner's DAG as task rows with prerequisites, run through the native feed loop (no GPU: injected prover).

Reference: the executor feeds `Planner` output into the task db (bento/crates/workflow/src/tasks/executor.rs:566-698 -> process_task
:56-250); `create_task` / `update_task_done` keep `waiting_on` and move tasks pending -> ready
(bento/crates/taskdb/migrations/1_taskdb.sql:197-228, 278-314); agents then claim Prove tasks, the log-depth tail of Join tasks
(join.rs:18-113), Resolve and Finalize.  The recursion proofs themselves are stand-ins here (include/bx_agent.h): a Join proves one
synthetic segment seeded by the hash of its children's seals, so the ROOT seal commits to every seal below it — the test recomputes
that chain.
"""
import json

import numpy as np
import pytest

from boundless_amd import agent as ag
from boundless_amd.hal import HalError
from boundless_amd.planner import Planner
from boundless_amd.prover import Segment, SegmentReceipt


def fake_seal(po2, seed):
    return (np.arange(16, dtype=np.uint32) * np.uint32(2654435761) + np.uint32(seed & 0xFFFFFFFF) + np.uint32(po2)).astype(np.uint32)


class FakeProver:
    """Deterministic seals; can fail the first attempt of chosen (po2, index) pairs."""

    def __init__(self, fail_once=()):
        self.calls = []
        self.fail_once = set(fail_once)

    def prove_segment(self, seg):
        self.calls.append((seg.po2, seg.index, seg.seed))
        if (seg.po2, seg.index) in self.fail_once:
            self.fail_once.discard((seg.po2, seg.index))
            raise RuntimeError("hipErrorLaunchFailure (injected)")
        return SegmentReceipt(seal=fake_seal(seg.po2, seg.seed), index=seg.index, po2=seg.po2)


# ------------------------------------------------------------------------------------------------ prerequisites in the task db
def test_pending_ready_transitions_follow_the_reference_sql():
    db = ag.TaskDb()
    db.create_task("j", "a", {"Prove": {"index": 0}})
    db.create_task("j", "b", {"Prove": {"index": 1}})
    db.create_task("j", "c", {"Join": {"idx": 2, "left": 0, "right": 1}}, prerequisites=["a", "b"])
    db.create_task("j", "d", {"Resolve": {"max_idx": 2, "union_max_idx": None}}, prerequisites=["c"])
    assert [db.task("j", t).state for t in "abcd"] == ["ready", "ready", "pending", "pending"]
    assert db.task("j", "c").waiting_on == 2 and db.task("j", "d").waiting_on == 1
    assert db.job("j") == {"state": "running", "tasks": 4, "pending": 2, "ready": 2, "running": 0, "done": 0, "failed": 0, "error": ""}
    ops = db.ops
    import ctypes as C

    done = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_size_t)(ops.update_task_done)
    failed = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_size_t)(ops.update_task_failed)
    assert done(ops.user, b"j", b"a", b"null", None, 0) == 1
    assert db.task("j", "c").state == "pending" and db.task("j", "c").waiting_on == 1
    assert done(ops.user, b"j", b"a", b"null", None, 0) == 0  # done twice is not a second release
    assert db.task("j", "c").waiting_on == 1
    assert done(ops.user, b"j", b"b", b"null", None, 0) == 1
    assert db.task("j", "c").state == "ready" and db.task("j", "c").waiting_on == 0 and db.task("j", "d").state == "pending"
    # a task created after its prerequisites are done is ready at once (create_task counts the not-done ones)
    db.create_task("j", "e", {"Prove": {"index": 9}}, prerequisites=["a", "b"])
    assert db.task("j", "e").state == "ready"
    # update_task_failed applies to pending rows too, and fails the job with that error (1_taskdb.sql:324-347)
    assert failed(ops.user, b"j", b"d", b"boom", None, 0) == 1
    assert db.task("j", "d").state == "failed" and db.job("j")["state"] == "failed" and db.job("j")["error"] == "boom"
    assert done(ops.user, b"j", b"c", b"null", None, 0) == 1
    assert db.task("j", "d").state == "failed"  # a failed dependant is not resurrected
    with pytest.raises(HalError, match="prerequisite task does not exist"):
        db.create_task("j", "f", {"Prove": {"index": 1}}, prerequisites=["nope"])
    with pytest.raises(HalError, match="no such job"):
        db.job("other")
    with pytest.raises(HalError, match="at least one segment"):
        db.plan_job("empty", 0)
    with pytest.raises(HalError, match="2\\^20 segments"):
        db.plan_job("typo", 1 << 40)


@pytest.mark.parametrize("n", [1, 2, 3, 4, 5, 8, 37] + [6, 7, 9, 15, 16, 17, 31, 32, 33, 63, 64, 65, 100, 127, 128, 129, 255, 257])
def test_plan_job_creates_the_planners_tasks_with_the_planners_dependencies(n):
    """bx_plan_job == driving Planner by hand the way the executor does: same task numbers, defs, streams and prerequisites."""
    db = ag.TaskDb()
    ids = db.plan_job("J", n, join_stream="join")
    p = Planner()
    want = {}
    seg = 0

    def drain(segment_index):
        while True:
            t = p.next_task()
            if t is None:
                return
            name = str(t.task_number)
            if t.command == "Segment":
                want[name] = ("prove", {"Prove": {"index": segment_index}}, [])
            elif t.command == "Join":
                want[name] = ("join", {"Join": {"idx": t.task_number, "left": t.depends_on[0], "right": t.depends_on[1]}}, [str(d) for d in t.depends_on])
            else:
                assert t.command == "Finalize"
                want["resolve"] = ("join", {"Resolve": {"max_idx": t.depends_on[0], "union_max_idx": None}}, [str(t.depends_on[0])])
                want["finalize"] = ("aux", {"Finalize": {"max_idx": t.depends_on[0]}}, ["resolve"])

    for seg in range(n):
        p.enqueue_segment()
        drain(seg)
    p.finish()
    drain(None)
    assert sorted(ids) == sorted(want) and len(ids) == 2 * n - 1 + 2
    # claim everything stream by stream, completing tasks as they come: the order must respect the dependencies
    import ctypes as C

    ops = db.ops
    req = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_char_p, C.POINTER(ag._ReadyTask), C.c_char_p, C.c_size_t)(ops.request_work)
    done = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_size_t)(ops.update_task_done)
    finished, progress = set(), True
    while progress:
        progress = False
        for stream in ("prove", "join", "aux"):
            t = ag._ReadyTask()
            while req(ops.user, stream.encode(), C.byref(t), None, 0) == 1:
                name = t.task_id.decode()
                w_stream, w_def, w_pre = want[name]
                assert w_stream == stream and json.loads(t.task_def.decode()) == w_def
                assert all(q in finished for q in w_pre), (name, w_pre)
                assert done(ops.user, b"J", name.encode(), b"null", None, 0) == 1
                finished.add(name)
                progress = True
    assert finished == set(want) and db.job("J")["state"] == "done"


# ------------------------------------------------------------------------------------------------------- through the feed loop
def expected_root(n, seg_po2, join_po2, base_seed):
    """Recompute the chain of stand-in joins bottom-up: leaf seals, then every Join's seed from its children's seals."""
    p = Planner()
    seals = {}
    for i in range(n):
        p.enqueue_segment()
        while True:
            t = p.next_task()
            if t is None:
                break
            if t.command == "Segment":
                seals[t.task_number] = fake_seal(seg_po2, base_seed + i)
            else:
                l, r = t.depends_on
                seals[t.task_number] = fake_seal(join_po2, ag.join_seed(seals[l], seals[r]))
    p.finish()
    root = None
    while True:
        t = p.next_task()
        if t is None:
            break
        if t.command == "Join":
            l, r = t.depends_on
            seals[t.task_number] = fake_seal(join_po2, ag.join_seed(seals[l], seals[r]))
        elif t.command == "Finalize":
            root = t.depends_on[0]
    return root, seals[root]


@pytest.mark.parametrize("prefetch", [False, True])
@pytest.mark.parametrize("n,lanes", [(1, 1), (2, 2), (7, 3), (16, 4), (37, 5)])
def test_a_planned_job_runs_to_its_rollup_receipt_through_the_lanes(n, lanes, prefetch):
    """prefetch: every lane claims one task ahead (include/bx_agent.h).  A lane that runs out of idle polls while its fetcher holds
    a task runs that task and keeps polling — the task's completion may release the job's last tasks when every other lane has
    already left."""
    base = 0xB0D1E550000
    prover = FakeProver(fail_once={(11, 2 * n - 2)} if n > 1 else ())  # the ROOT join fails once and is retried
    a = ag.Agent(prover=prover, verify=False, poll_time=0.002, inflight=lanes, join_po2=11, also_streams="aux", prefetch=prefetch)
    try:
        for i in range(n):
            a.store.set_key_with_expiry(f"job:J:segments:{i}", ag.serialize_segment(Segment.synthetic(i, po2=13)), 600)
        ids = a.taskdb.plan_job("J", n)
        assert a.poll_work(max_idle_polls=3) == len(ids)
        job = a.taskdb.job("J")
        assert job["state"] == "done" and job["done"] == 2 * n - 1 + 2
        rows = {t: a.taskdb.task("J", t) for t in ids}
        # every prove once, every join once (+ the injected retry), nothing else reached the prover
        proves = [c for c in prover.calls if c[0] == 13]
        joins = [c for c in prover.calls if c[0] == 11]
        assert sorted(c[1] for c in proves) == list(range(n)) and len(joins) == (n - 1) + (1 if n > 1 else 0)
        if n > 1:
            assert rows[str(2 * n - 2)].retries == 1
        # dependencies respected in time: a task was claimed only after each of its prerequisites was done
        pl = Planner()
        for _ in range(n):
            pl.enqueue_segment()
        pl.finish()
        for k in range(pl.task_count()):
            t = pl.get_task(k)
            if t.command == "Join":
                assert rows[str(t.task_number)].started_s >= max(rows[str(d)].updated_s for d in t.depends_on)
            elif t.command == "Finalize":
                assert rows["resolve"].started_s >= rows[str(t.depends_on[0])].updated_s
                assert rows["finalize"].started_s >= rows["resolve"].updated_s
        # the rollup receipt is the root of the hash chain over every seal below it; intermediate receipts were cleaned up
        root_idx, root_seal = expected_root(n, 13, 11, base)
        rollup = ag.deserialize_receipt(a.store.get("receipts/stark/J.synthetic"))
        assert np.array_equal(rollup.seal, root_seal) and rollup.index == (root_idx if n > 1 else 0)
        assert sorted(a.store.keys()) == sorted([f"job:J:synthetic_receipts:{root_idx}", "receipts/stark/J.synthetic"])
        text = a.metrics_text()
        if n > 1:
            assert f'task_operations_total{{task_name="join",operation_type="join_receipts",status="success"}} {n - 1}' in text
            assert f'task_operations_total{{task_name="join",operation_type="complete",status="success"}} {n - 1}' in text
        assert 'task_operations_total{task_name="finalize",operation_type="complete",status="success"} 1' in text
        assert 'task_processing_total{task_type="resolve",status="success"} 1' in text
    finally:
        a.close()


def test_without_an_aux_worker_the_job_waits_at_finalize_and_a_second_agent_finishes_it():
    """Streams are worker types (1_taskdb.sql:25-33): a prove-stream agent does not claim the aux stream's finalize task unless told
    to (`also_streams`); an aux agent on the same stores does."""
    store, db = ag.HotStore(), ag.TaskDb()
    a = ag.Agent(prover=FakeProver(), verify=False, poll_time=0.002, inflight=2, join_po2=11, store=store, taskdb=db)
    for i in range(4):
        store.set_key_with_expiry(f"job:S:segments:{i}", ag.serialize_segment(Segment.synthetic(i, po2=13)), 600)
    ids = db.plan_job("S", 4)
    assert a.poll_work(max_idle_polls=3) == len(ids) - 1
    assert db.task("S", "finalize").state == "ready" and db.job("S")["state"] == "running"
    a.close()
    aux = ag.Agent(prover=FakeProver(), verify=False, poll_time=0.002, inflight=1, store=store, taskdb=db, task_stream="aux")
    assert aux.poll_work(max_idle_polls=3) == 1
    assert db.job("S")["state"] == "done"
    aux.close()


def test_a_join_whose_child_receipt_is_missing_or_corrupt_fails_with_the_references_error_chain():
    store, db = ag.HotStore(), ag.TaskDb()
    a = ag.Agent(prover=FakeProver(), verify=False, poll_time=0.002, inflight=1, join_po2=11, store=store, taskdb=db)
    good = ag.serialize_receipt(SegmentReceipt(seal=fake_seal(13, 1), index=0, po2=13))
    store.set_key_with_expiry("job:X:synthetic_receipts:0", good, 600)
    db.create_task("X", "2", {"Join": {"idx": 2, "left": 0, "right": 1}}, max_retries=0)
    assert a.process_one("X", "2", {"Join": {"idx": 2, "left": 0, "right": 1}}) is False
    err = db.task("X", "2").error
    assert err.startswith("[BENTO-WF-119] Join failed: failed to get receipts for keys: job:X:synthetic_receipts:0, job:X:synthetic_receipts:1")
    assert "Key not found (nil response)" in err
    store.set_key_with_expiry("job:X:synthetic_receipts:1", b"\x00" * 40, 600)
    db.create_task("X", "3", {"Join": {"idx": 3, "left": 0, "right": 1}}, max_retries=0)
    assert a.process_one("X", "3", {"Join": {"idx": 3, "left": 0, "right": 1}}) is False
    assert db.task("X", "3").error == "[BENTO-WF-119] Join failed: [BENTO-JOIN-002] Failed to deserialize right receipt"
    # a malformed request is an invalid task_def, as for Prove
    db.create_task("X", "4", {"Join": {"idx": 4, "left": 0}}, max_retries=0)
    assert a.process_one("X", "4", {"Join": {"idx": 4, "left": 0}}) is False
    assert db.task("X", "4").error == "Invalid task_def: X:4"
    a.close()


def test_recursion_tasks_are_refused_outside_synthetic_mode():
    """An agent with a real (opaque) prover never runs the stand-ins: a Join that reaches it fails loudly."""
    a = ag.Agent(blob_prover=lambda b: b, poll_time=0.002, synthetic=False)
    a.taskdb.create_task("R", "5", {"Join": {"idx": 5, "left": 1, "right": 2}}, max_retries=0)
    assert a.process_one("R", "5", {"Join": {"idx": 5, "left": 1, "right": 2}}) is False
    assert "task type Join reached a prove-stream agent" in a.taskdb.task("R", "5").error
    a.close()


def test_the_lift_leg_of_the_prove_task_as_a_stand_in():
    """prove.rs:41-113: prove_segment -> verify -> lift -> verify -> store the LIFTED receipt.  With cfg.lift_po2 a Prove task runs a
    second (stand-in) proof seeded by the segment seal and stores that; the joins consume the lifted receipts."""
    n, base = 7, 0xB0D1E550000
    prover = FakeProver(fail_once={(12, 3)})  # the lift of segment 3 fails once: the whole Prove task is retried
    a = ag.Agent(prover=prover, verify=False, poll_time=0.002, inflight=3, join_po2=11, lift_po2=12, also_streams="aux")
    try:
        for i in range(n):
            a.store.set_key_with_expiry(f"job:L:segments:{i}", ag.serialize_segment(Segment.synthetic(i, po2=13)), 600)
        ids = a.taskdb.plan_job("L", n)
        assert a.poll_work(max_idle_polls=3) == len(ids) and a.taskdb.job("L")["state"] == "done"
        proves = sorted(c[1] for c in prover.calls if c[0] == 13)
        lifts = sorted(c[1] for c in prover.calls if c[0] == 12)
        joins = [c for c in prover.calls if c[0] == 11]
        assert proves == sorted(list(range(n)) + [3]) and lifts == sorted(list(range(n)) + [3]) and len(joins) == n - 1
        # the chain over LIFTED leaves
        p, seals, root, leaf = Planner(), {}, None, 0
        for _ in range(n):
            p.enqueue_segment()
        p.finish()
        for k in range(p.task_count()):
            t = p.get_task(k)
            if t.command == "Segment":
                seals[t.task_number] = fake_seal(12, ag.join_seed(fake_seal(13, base + leaf), np.zeros(0, np.uint32)))
                leaf += 1
        for k in range(p.task_count()):
            t = p.get_task(k)
            if t.command == "Join":
                seals[t.task_number] = fake_seal(11, ag.join_seed(seals[t.depends_on[0]], seals[t.depends_on[1]]))
            elif t.command == "Finalize":
                root = t.depends_on[0]
        rollup = ag.deserialize_receipt(a.store.get("receipts/stark/L.synthetic"))
        assert np.array_equal(rollup.seal, seals[root])
        assert f'task_operations_total{{task_name="prove",operation_type="lift",status="success"}} {n}' in a.metrics_text()
    finally:
        a.close()


def test_an_agent_without_a_prover_serves_the_aux_tasks_and_refuses_proofs_with_the_references_errors():
    """`prover: None` (bento/crates/workflow/src/lib.rs:242-252): only the prove / join / coproc worker types get a prover; an aux
    agent (cfg.no_prover) needs no GPU, finishes the job's finalize task, and a Prove or Join task that reaches such an agent fails with
    "[BENTO-PROVE-002] Missing prover from prove task" (prove.rs:41-45) / "Missing prover from join task" (join.rs:51-55)."""
    store, db = ag.HotStore(), ag.TaskDb()
    gpu = ag.Agent(prover=FakeProver(), verify=False, poll_time=0.002, inflight=2, join_po2=11, store=store, taskdb=db)
    aux = ag.Agent(no_prover=True, verify=False, poll_time=0.002, inflight=1, store=store, taskdb=db, task_stream="aux")
    try:
        for i in range(4):
            store.set_key_with_expiry(f"job:N:segments:{i}", ag.serialize_segment(Segment.synthetic(i, po2=13)), 600)
        ids = db.plan_job("N", 4)
        assert gpu.poll_work(max_idle_polls=3) == len(ids) - 1 and db.task("N", "finalize").state == "ready"
        assert aux.poll_work(max_idle_polls=3) == 1 and db.job("N")["state"] == "done"
        assert "receipts/stark/N.synthetic" in store.keys()
        # proofs that reach the prover-less agent
        store.set_key_with_expiry("job:M:segments:0", ag.serialize_segment(Segment.synthetic(0, po2=13)), 600)
        db.create_task("M", "0", {"Prove": {"index": 0}}, max_retries=0, stream="aux")
        assert aux.poll_work(max_idle_polls=3) == 0
        row = db.task("M", "0")
        assert row.state == "failed" and row.error == "[BENTO-WF-115] Prove failed: [BENTO-PROVE-002] Missing prover from prove task"
        assert "job:M:segments:0" in store.keys()  # nothing was consumed
        for k in (1, 2):
            store.set_key_with_expiry(f"job:M2:synthetic_receipts:{k}", ag.serialize_receipt(SegmentReceipt(seal=fake_seal(11, k), index=k, po2=11)), 600)
        db.create_task("M2", "3", {"Join": {"idx": 3, "left": 1, "right": 2}}, max_retries=0, stream="aux")
        assert aux.poll_work(max_idle_polls=3) == 0
        assert db.task("M2", "3").error == "[BENTO-WF-119] Join failed: Missing prover from join task"
        with pytest.raises(HalError, match="prover"):
            ag.Agent(no_prover=True, synthetic=False, store=store, taskdb=db)  # still the synthetic key scheme: must be asked for
    finally:
        gpu.close()
        aux.close()
