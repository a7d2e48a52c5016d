"""CPU: planner known-answer tests restated from the reference's unit tests, and the prove-agent feed loop."""
import json

import numpy as np
import pytest

from boundless_amd import agent as ag
from boundless_amd import planner as pl
from boundless_amd.prover import Segment, SegmentReceipt


# ---- bento/crates/taskdb/src/planner/mod.rs:258-290 (test_simple_plan) ----
def test_simple_plan():
    p = pl.Planner()
    assert p.enqueue_segment() == 0
    t = p.next_task()
    assert (t.command, t.task_number, t.depends_on, t.keccak_depends_on) == (pl.SEGMENT, 0, [], [])
    assert p.next_task() is None
    assert p.enqueue_keccak() == 1
    t = p.next_task()
    assert (t.command, t.task_number, t.depends_on, t.keccak_depends_on) == (pl.KECCAK, 1, [], [])
    assert p.next_task() is None
    p.finish()
    t = p.next_task()
    assert t.command == pl.FINALIZE and t.task_number == 2 and t.task_height == 1
    assert len(t.depends_on) == 1 and len(t.keccak_depends_on) == 1
    assert p.task_count() == 3 and p.get_task(0).task_number == 0


# ---- mod.rs:292-351 (test_balanced) ----
def test_balanced():
    p = pl.Planner()
    p.enqueue_segment()
    t = p.next_task()
    assert (t.command, t.task_number, t.task_height) == (pl.SEGMENT, 0, 0) and p.next_task() is None
    p.enqueue_keccak()
    t = p.next_task()
    assert (t.command, t.task_number, t.task_height) == (pl.KECCAK, 1, 0)
    p.enqueue_segment()
    t = p.next_task()
    assert (t.command, t.task_number, t.task_height, t.depends_on) == (pl.SEGMENT, 2, 0, [])
    j = p.next_task()
    assert (j.command, j.task_number, j.task_height, len(j.depends_on)) == (pl.JOIN, 3, 1, 2)
    p.enqueue_keccak()
    t = p.next_task()
    assert (t.command, t.task_number, t.task_height) == (pl.KECCAK, 4, 0)
    u = p.next_task()
    assert (u.command, u.task_number, u.task_height, len(u.keccak_depends_on)) == (pl.UNION, 5, 1, 2)
    p.finish()
    f = p.next_task()
    assert (f.command, f.task_number, f.task_height, len(f.depends_on), len(f.keccak_depends_on)) == (pl.FINALIZE, 6, 2, 1, 1)


# ---- mod.rs:353-394 (test_unbalanced_keccak) ----
def test_unbalanced_keccak():
    p = pl.Planner()
    p.enqueue_keccak(); p.enqueue_keccak(); p.enqueue_keccak(); p.enqueue_segment()
    p.finish()
    got = [(t.task_number, t.command, t.task_height) for t in iter(p.next_task, None)]
    assert got == [(0, pl.KECCAK, 0), (1, pl.KECCAK, 0), (2, pl.UNION, 1), (3, pl.KECCAK, 0), (4, pl.SEGMENT, 0),
                   (5, pl.UNION, 2), (6, pl.FINALIZE, 3)]


# ---- mod.rs:396-430 (test_unbalanced) ----
def test_unbalanced():
    p = pl.Planner()
    p.enqueue_segment(); p.enqueue_segment(); p.enqueue_segment()
    p.finish()
    got = [(t.task_number, t.command) for t in iter(p.next_task, None)]
    assert got == [(0, pl.SEGMENT), (1, pl.SEGMENT), (2, pl.JOIN), (3, pl.SEGMENT), (4, pl.JOIN), (5, pl.FINALIZE)]
    assert p.get_task(5).task_height == 3
    assert p.get_task(2).depends_on == [0, 1] and p.get_task(4).depends_on == [2, 3]


# ---- mod.rs:432-452 (error cases) ----
def test_planner_errors():
    with pytest.raises(pl.PlanNotStarted, match="Planning not yet started"):
        pl.Planner().finish()
    p = pl.Planner()
    p.enqueue_segment()
    p.finish()
    with pytest.raises(pl.PlanFinalized, match="Cannot add segment to finished plan"):
        p.enqueue_segment()
    with pytest.raises(IndexError, match="Invalid task number 100"):
        pl.Planner().get_task(100)


def test_join_tree_is_log_depth_for_64_segments():
    """BASELINE configs[2]/[3] shape: 64 segments -> 63 joins, height 6, every join's children are complete."""
    p = pl.Planner()
    for _ in range(64):
        p.enqueue_segment()
    fin = p.finish()
    assert p.task_count() == 64 + 63 + 1
    root = p.get_task(p.get_task(fin).depends_on[0])
    assert root.command == pl.JOIN and root.task_height == 6
    for t in p.tasks:
        for d in t.depends_on:
            assert d < t.task_number


# ---------------------------------------------------------------------------------------------------------- agent
class FakeProver:
    def __init__(self, fail_times=0, bad_seal=False):
        self.calls = 0
        self.fail_times = fail_times
        self.bad_seal = bad_seal

    def prove_segment(self, seg):
        self.calls += 1
        if self.calls <= self.fail_times:
            raise RuntimeError("hipErrorLaunchFailure (injected)")
        r = SegmentReceipt(seal=np.arange(10, dtype=np.uint32) + seg.seed % 7, index=seg.index, po2=seg.po2)
        if not self.bad_seal:
            r.verify_integrity = lambda: None
        return r


def test_task_json_and_keys_roundtrip():
    assert ag.parse_task('{"Prove":{"index":7}}') == ("prove", ag.ProveReq(7))
    assert ag.job_type_str("prove") == "prove-lift"
    with pytest.raises(ValueError):
        ag.parse_task('{"Join":{"idx":1,"left":2,"right":3}}')
    seg = Segment.synthetic(5, po2=20)
    assert ag.deserialize_segment(ag.serialize_segment(seg)) == seg
    r = SegmentReceipt(seal=np.arange(33, dtype=np.uint32), index=5, po2=20)
    r2 = ag.deserialize_receipt(ag.serialize_receipt(r))
    assert np.array_equal(r2.seal, r.seal) and (r2.index, r2.po2) == (5, 20)


def test_prove_task_key_scheme_cleanup_and_metrics():
    a = ag.Agent(prover=FakeProver(), poll_time=0.0)
    job = "0b1e55-job"
    a.store.set_key_with_expiry(f"job:{job}:segments:3", ag.serialize_segment(Segment.synthetic(3, po2=12)), 60)
    ag.prove_task(a, job, "task-3", ag.ProveReq(index=3))
    assert a.store.keys() == [f"job:{job}:recursion_receipts:task-3"]  # receipt stored, segment unlinked
    rec = ag.deserialize_receipt(a.store.get(f"job:{job}:recursion_receipts:task-3"))
    assert rec.index == 3 and rec.po2 == 12
    assert a.metrics.ops[("prove", "prove_segment", "success")] == 2  # record_task_operation + record_task, like the reference
    assert a.metrics.ops[("prove", "complete", "success")] == 1
    text = a.metrics.exposition()
    assert 'task_operations_total{task_name="prove",operation_type="complete",status="success"} 1' in text
    assert 'task_duration_seconds_bucket{task_name="prove",operation_type="prove_segment",status="success",le="0.1"} 2' in text
    with pytest.raises(RuntimeError, match="segment data not found for segment key: job:x:segments:9"):
        ag.prove_task(a, "x", "t", ag.ProveReq(index=9))


def test_poll_loop_retries_then_succeeds_and_fails_after_max_retries():
    a = ag.Agent(prover=FakeProver(fail_times=2), poll_time=0.0)
    a.store.set_key_with_expiry("job:j:segments:0", ag.serialize_segment(Segment.synthetic(0, po2=10)))
    a.stream.create_task("j", "t0", {"Prove": {"index": 0}}, max_retries=3)
    assert ag.poll_work(a, max_idle_polls=1) == 1
    row = a.stream.rows()[0]
    assert row.state == "done" and row.retries == 2 and "injected" in row.error
    # a task whose segment blob is missing exhausts its retries and is failed; the agent survives and serves the next task
    a2 = ag.Agent(prover=FakeProver(), poll_time=0.0)
    a2.stream.create_task("j", "missing", {"Prove": {"index": 5}}, max_retries=1)
    a2.store.set_key_with_expiry("job:j:segments:6", ag.serialize_segment(Segment.synthetic(6, po2=10)))
    a2.stream.create_task("j", "ok", {"Prove": {"index": 6}})
    assert ag.poll_work(a2, max_idle_polls=1) == 1
    states = {r.task_id: (r.state, r.retries) for r in a2.stream.rows()}
    assert states == {"missing": ("failed", 1), "ok": ("done", 0)}
    assert a2.metrics.ops[("prove", "complete", "failed")] == 1


def test_receipt_that_fails_verification_is_not_stored():
    a = ag.Agent(prover=FakeProver(bad_seal=True), poll_time=0.0)
    a.store.set_key_with_expiry("job:j:segments:0", ag.serialize_segment(Segment.synthetic(0, po2=10)))
    with pytest.raises(RuntimeError, match=r"\[BENTO-PROVE-004\]"):
        ag.prove_task(a, "j", "t", ag.ProveReq(0))
    assert a.store.keys() == ["job:j:segments:0"]  # nothing written, segment kept for the retry


def test_hot_store_expiry():
    s = ag.HotStore()
    s.set_key_with_expiry("k", b"v", ttl_secs=-1)
    with pytest.raises(KeyError):
        s.get("k")
