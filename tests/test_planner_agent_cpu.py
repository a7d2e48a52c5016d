"""CPU: planner known-answer tests restated from the reference's unit tests, and the native prove-agent feed loop."""
import json

import numpy as np
import pytest

from boundless_amd.hal import HalError

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
# The feed loop is native code (boundless_amd/csrc/agent.cpp); a Python prover is injected through the prover-ops callbacks so
# the control flow can be exercised without a GPU.
class FakeProver:
    def __init__(self, fail_times=0):
        self.calls = 0
        self.fail_times = fail_times

    def prove_segment(self, seg):
        self.calls += 1
        if self.calls <= self.fail_times:
            raise RuntimeError("hipErrorLaunchFailure (injected)")
        return SegmentReceipt(seal=np.arange(10, dtype=np.uint32) + seg.seed % 7, index=seg.index, po2=seg.po2)


def test_task_json_and_wire_roundtrip():
    seg = Segment.synthetic(5, po2=20)
    blob = ag.serialize_segment(seg)
    assert len(blob) == 28 and blob[:8] == b"BXSYNSEG" and ag.deserialize_segment(blob) == seg
    with pytest.raises(ValueError, match="Failed to deserialize segment data from redis"):
        ag.deserialize_segment(blob[:-1])
    with pytest.raises(ValueError, match="not a synthetic segment blob"):  # e.g. a real bincode(Segment): refused loudly
        ag.deserialize_segment(b"\x00" * 28)
    a = ag.Agent(prover=FakeProver(), verify=False, poll_time=0.01)
    # serde's externally tagged TaskType (workflow-common/src/lib.rs:160-178): anything but one known variant is invalid
    for bad in ('{"Prove":{"idx":7}}', '{"Prove":{"index":-1}}', '{"Prove":{"index":1.5}}', '[]', '{"Prove":{"index":1},"x":{}}',
                '{"Nope":{}}', '{"Prove":{"index":1}', ''):
        a.taskdb.create_task("jj", f"t{bad}", {"Prove": {"index": 0}}, max_retries=0)
        assert a.process_one("jj", f"t{bad}", bad, max_retries=0) is False
        assert a.taskdb.task("jj", f"t{bad}").error == f"Invalid task_def: jj:t{bad}"  # lib.rs:446-447
    assert a.taskdb.count("failed") == 8  # update_task_failed applies to ready rows too (1_taskdb.sql:324)
    a.close()


def test_prove_task_key_scheme_cleanup_and_metrics():
    a = ag.Agent(prover=FakeProver(), verify=False, poll_time=0.01)
    job = "0b1e55-job"
    a.store.set_key_with_expiry(f"job:{job}:segments:3", ag.serialize_segment(Segment.synthetic(3, po2=12)), 60)
    a.taskdb.create_task(job, "task-3", {"Prove": {"index": 3}})
    assert a.poll_work(max_idle_polls=1) == 1
    # receipt stored, segment unlinked.  A synthetic seal never goes under the key Join workers read (recursion_receipts)
    assert a.store.keys() == [f"job:{job}:synthetic_receipts:task-3"]
    rec = ag.deserialize_receipt(a.store.get(f"job:{job}:synthetic_receipts:task-3"))
    assert rec.index == 3 and rec.po2 == 12 and np.array_equal(rec.seal, np.arange(10, dtype=np.uint32) + Segment.synthetic(3).seed % 7)
    row = a.taskdb.task(job, "task-3")
    assert (row.state, row.retries, row.output) == ("done", 0, "null")
    text = a.metrics_text()
    # record_task_operation + record_task both count the prove (prove.rs:50-51,57)
    assert 'task_operations_total{task_name="prove",operation_type="prove_segment",status="success"} 2' in text
    assert 'task_operations_total{task_name="prove",operation_type="complete",status="success"} 1' in text
    assert 'task_duration_seconds_bucket{task_name="prove",operation_type="prove_segment",status="success",le="0.1"} 2' in text
    assert 'task_duration_seconds_bucket{task_name="prove",operation_type="complete",status="success",le="+Inf"} 1' in text
    for op in ("get", "set_ex", "unlink"):
        assert f'redis_operations_total{{operation_type="{op}",status="success"}} 1' in text
    assert 'redis_operation_duration_seconds_bucket{operation_type="get",status="success",le="0.001"}' in text
    # the next-gen worker loop's series (prover/crates/workflow/src/lib.rs:613-637): one claim served, then the queue is empty
    assert 'task_claims_total{task_stream="prove",result="claimed"} 1' in text
    # ... 1 or 2 times: a poll that finds the queue empty while the finisher still holds a task is repeated once it is idle
    import re
    assert int(re.search(r'task_claims_total\{task_stream="prove",result="empty"\} (\d+)', text).group(1)) in (1, 2)
    assert 'task_processing_total{task_type="prove-lift",status="success"} 1' in text  # to_job_type_str, workflow-common lib.rs:179
    assert 'task_processing_end_to_end_seconds_bucket{task_type="prove-lift",status="success",le="0.01"}' in text
    assert 'task_processing_end_to_end_seconds_count{task_type="prove-lift",status="success"} 1' in text
    assert "# TYPE task_claims_total counter" in text and "# TYPE task_processing_end_to_end_seconds histogram" in text
    assert "task_retry_attempts_total{" not in text and "task_max_retries_exhausted_total{" not in text
    a.close()


def test_missing_segment_blob_error_chain_and_no_retries():
    a = ag.Agent(prover=FakeProver(), verify=False, poll_time=0.01)
    a.taskdb.create_task("x", "t", {"Prove": {"index": 9}}, max_retries=0)
    assert a.poll_work(max_idle_polls=1) == 0
    row = a.taskdb.task("x", "t")
    assert row.state == "failed" and row.retries == 0
    # anyhow `{:#}` chain: WF-115 context, prove.rs:30-33 context, redis.rs:51-55 nil error
    assert row.error == ("[BENTO-WF-115] Prove failed: segment data not found for segment key: job:x:segments:9: "
                         "Key not found (nil response): job:x:segments:9")
    assert 'redis_operations_total{operation_type="get",status="error"} 1' in a.metrics_text()
    a.close()


def test_poll_loop_retries_then_succeeds_and_fails_after_max_retries():
    a = ag.Agent(prover=FakeProver(fail_times=2), verify=False, poll_time=0.01)
    a.store.set_key_with_expiry("job:j:segments:0", ag.serialize_segment(Segment.synthetic(0, po2=10)))
    a.taskdb.create_task("j", "t0", {"Prove": {"index": 0}}, max_retries=3)
    assert a.poll_work(max_idle_polls=1) == 1
    row = a.taskdb.rows()[0]
    assert row.state == "done" and row.retries == 2
    text = a.metrics_text()  # lib.rs:664-666: every requeue is a retry attempt; each attempt is one processing sample
    assert 'task_retry_attempts_total{task_type="prove-lift"} 2' in text
    assert 'task_processing_total{task_type="prove-lift",status="error"} 2' in text
    assert 'task_processing_total{task_type="prove-lift",status="success"} 1' in text
    assert 'task_claims_total{task_stream="prove",result="claimed"} 3' in text
    assert "task_max_retries_exhausted_total{" not in text
    a.close()
    # a task whose segment blob is missing exhausts its retries and is failed with "retry max hit: ..." (lib.rs:395-413);
    # the agent survives and serves the next task
    a2 = ag.Agent(prover=FakeProver(), verify=False, poll_time=0.01)
    a2.taskdb.create_task("j", "missing", {"Prove": {"index": 5}}, max_retries=1)
    a2.store.set_key_with_expiry("job:j:segments:6", ag.serialize_segment(Segment.synthetic(6, po2=10)))
    a2.taskdb.create_task("j", "ok", {"Prove": {"index": 6}})
    assert a2.poll_work(max_idle_polls=1) == 1
    states = {r.task_id: (r.state, r.retries) for r in a2.taskdb.rows()}
    assert states == {"missing": ("failed", 1), "ok": ("done", 0)}
    err = a2.taskdb.task("j", "missing").error
    assert err.startswith("retry max hit: [BENTO-WF-115] Prove failed: segment data not found for segment key: job:j:segments:5")
    text = a2.metrics_text()  # one requeue, then the limit (lib.rs:650-661)
    assert 'task_retry_attempts_total{task_type="prove-lift"} 1' in text
    assert 'task_max_retries_exhausted_total{task_type="prove-lift"} 1' in text
    a2.close()
    # max_retries == 0: the first failure is final (lib.rs:670-675); a task_def that does not parse is labelled invalid_task
    a3 = ag.Agent(prover=FakeProver(), verify=False, poll_time=0.01)
    a3.taskdb.create_task("j", "t", {"Prove": {"index": 0}}, max_retries=0)
    assert a3.process_one("j", "t", '{"Nope":{}}', max_retries=0) is False
    a3.taskdb.create_task("j", "u", {"Prove": {"index": 0}}, max_retries=0)
    assert a3.process_one("j", "u", '{"Join":{"idx":1,"left":2,"right":3}}', max_retries=0) is False
    text = a3.metrics_text()
    assert 'task_max_retries_exhausted_total{task_type="invalid_task"} 1' in text
    assert 'task_max_retries_exhausted_total{task_type="join"} 1' in text
    assert 'task_processing_total{task_type="invalid_task",status="error"} 1' in text
    a3.close()


def test_long_errors_are_truncated_before_reaching_the_db():
    class Noisy:
        def prove_segment(self, seg):
            raise RuntimeError("x" * 5000)

    a = ag.Agent(prover=Noisy(), verify=False, poll_time=0.01)
    a.store.set_key_with_expiry("job:j:segments:0", ag.serialize_segment(Segment.synthetic(0, po2=10)))
    a.taskdb.create_task("j", "t", {"Prove": {"index": 0}}, max_retries=0)
    assert a.poll_work(max_idle_polls=1) == 0
    assert len(a.taskdb.task("j", "t").error) == 1024  # err_str.truncate(1024), lib.rs:424
    a.close()


def test_receipt_that_fails_verification_is_not_stored():
    a = ag.Agent(prover=FakeProver(), verify=True, poll_time=0.01)  # arange(10) is not a seal the verifier accepts
    a.store.set_key_with_expiry("job:j:segments:0", ag.serialize_segment(Segment.synthetic(0, po2=10)))
    a.taskdb.create_task("j", "t", {"Prove": {"index": 0}}, max_retries=0)
    assert a.poll_work(max_idle_polls=1) == 0
    assert "[BENTO-PROVE-004] Failed to verify segment receipt integrity" in a.taskdb.task("j", "t").error
    assert a.store.keys() == ["job:j:segments:0"]  # nothing written, segment kept for a retry
    a.close()


def test_verification_failure_found_by_the_finisher_thread_is_retried_up_to_the_limit():
    p = FakeProver()
    a = ag.Agent(prover=p, verify=True, poll_time=0.01)
    a.store.set_key_with_expiry("job:j:segments:0", ag.serialize_segment(Segment.synthetic(0, po2=10)))
    a.taskdb.create_task("j", "t", {"Prove": {"index": 0}}, max_retries=2)
    assert a.poll_work(max_idle_polls=1) == 0
    row = a.taskdb.task("j", "t")
    assert (row.state, row.retries, p.calls) == ("failed", 2, 3)
    assert row.error.startswith("retry max hit: [BENTO-WF-115] Prove failed: [BENTO-PROVE-004]")
    a.close()


def test_other_task_streams_are_left_alone_and_stop_flag_ends_the_loop():
    import threading
    import time

    a = ag.Agent(prover=FakeProver(), verify=False, poll_time=0.02)
    a.taskdb.create_task("j", "join-1", {"Join": {"idx": 1, "left": 2, "right": 3}}, stream="join")
    t0 = time.time()
    th = threading.Thread(target=lambda: a.poll_work(max_idle_polls=None))
    th.start()
    time.sleep(0.2)
    a.stop()  # the SIGTERM flag
    th.join(timeout=5)
    assert not th.is_alive() and time.time() - t0 < 5
    assert a.taskdb.task("j", "join-1").state == "ready"
    a.close()


def test_hot_store_expiry_and_several_lanes():
    import time

    s = ag.HotStore()
    s.set_key_with_expiry("k", b"v", ttl_secs=1)
    assert s.get("k") == b"v"
    time.sleep(1.1)
    with pytest.raises(KeyError):
        s.get("k")
    # 4 lanes claim 32 tasks between them; every task is done exactly once
    p = FakeProver()
    a = ag.Agent(prover=p, verify=False, poll_time=0.01, inflight=4)
    for i in range(32):
        a.store.set_key_with_expiry(f"job:m:segments:{i}", ag.serialize_segment(Segment.synthetic(i, po2=10)))
        a.taskdb.create_task("m", f"p{i}", {"Prove": {"index": i}})
    assert a.poll_work(max_idle_polls=2) == 32
    assert a.taskdb.count("done") == 32 and p.calls == 32
    assert a.store.keys() == sorted(f"job:m:synthetic_receipts:p{i}" for i in range(32))
    assert sum(n for _, n in a.lane_stats()) == 32 and len(a.lane_stats()) == 4
    a.close()


def test_the_synthetic_prover_is_opt_in_and_a_real_prover_gets_the_reference_keys():
    """Round-1 advisor finding: nothing stopped the agent from being pointed at a real `prove` stream.  Now the synthetic wire
    format needs cfg.synthetic, non-synthetic blobs fail loudly, and only an opaque prover (bytes in, bytes out: what a real
    `impl ProverServer` shim provides) writes job:{id}:recursion_receipts:{task} (tasks/mod.rs:23)."""
    with pytest.raises(HalError, match="synthetic"):
        ag.Agent(prover=FakeProver(), synthetic=False)
    # a blob that is not a synthetic segment (here: what a real executor would have stored) fails the task, names the reason
    a = ag.Agent(prover=FakeProver(), verify=False, poll_time=0.01)
    a.store.set_key_with_expiry("job:j:segments:0", b"\x01\x02" * 40000)
    a.taskdb.create_task("j", "t", {"Prove": {"index": 0}}, max_retries=0)
    assert a.poll_work(max_idle_polls=1) == 0
    assert "not a synthetic segment blob" in a.taskdb.task("j", "t").error
    assert a.store.keys() == ["job:j:segments:0"]
    a.close()
    # opaque mode: the stored bytes go to the prover untouched, its bytes are stored under the reference's key
    seen = []

    def real_prover(blob):
        seen.append(blob)
        return b"lifted:" + blob[::-1]

    b = ag.Agent(blob_prover=real_prover, synthetic=False, poll_time=0.01, inflight=2)
    for i in range(5):
        b.store.set_key_with_expiry(f"job:r:segments:{i}", bytes([i]) * (1000 + i))
        b.taskdb.create_task("r", f"p{i}", {"Prove": {"index": i}})
    assert b.poll_work(max_idle_polls=2) == 5
    assert b.store.keys() == [f"job:r:recursion_receipts:p{i}" for i in range(5)]
    assert b.store.get("job:r:recursion_receipts:p3") == b"lifted:" + bytes([3]) * 1003
    assert sorted(seen) == sorted(bytes([i]) * (1000 + i) for i in range(5))
    b.close()


def test_one_agent_several_devices_share_one_queue():
    """bx_agent_config.n_devices: lanes of every GPU of the node claim from the same task db (BASELINE configs[2] in native
    code; the reference runs one agent process per GPU against one Postgres queue, compose.yml:113)."""
    import time

    class Slow:
        def __init__(self):
            self.calls = 0

        def prove_segment(self, seg):
            self.calls += 1
            time.sleep(0.002)
            return SegmentReceipt(seal=np.arange(4, dtype=np.uint32), index=seg.index, po2=seg.po2)

    a = ag.Agent(prover=Slow(), verify=False, poll_time=0.005, inflight=2, devices=[0, 1, 2, 3])
    for i in range(64):
        a.store.set_key_with_expiry(f"job:b:segments:{i}", ag.serialize_segment(Segment.synthetic(i, po2=10)))
        a.taskdb.create_task("b", f"p{i}", {"Prove": {"index": i}})
    assert a.poll_work(max_idle_polls=3) == 64
    stats = a.lane_stats()
    assert [d for d, _ in stats] == [0, 0, 1, 1, 2, 2, 3, 3]
    assert sum(n for _, n in stats) == 64 and a.taskdb.count("done") == 64
    per_dev = [sum(n for d, n in stats if d == k) for k in range(4)]
    assert all(n > 0 for n in per_dev), per_dev  # every device's lanes got work from the one queue
    a.close()
