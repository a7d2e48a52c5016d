"""CPU: the lane's claim-ahead fetcher (`bx_agent_config.prefetch`, include/bx_agent.h) — the GET of segment k+1 overlaps proof k.

Reference: an agent claims a task, GETs `job:{id}:segments:{n}` from the hot store, proves, stores the receipt — serially per
process (bento/crates/workflow/src/lib.rs:603-654 -> tasks/prove.rs:22-49); the deployment hides the GET by running several agents
per GPU.  Behind the REST worker protocol a segment is an ~80 MB download, so a lane that fetches only when
idle leaves its share of the GPU idle for every download.  Here the store is the library's in-memory one behind a wrapper that
makes every GET slow; the prover is injected (no GPU).
"""
import ctypes as C
import os
import sys
import threading
import time

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from rest_stub_server import StubServer  # noqa: E402

from boundless_amd import agent as ag  # noqa: E402
from boundless_amd.prover import Segment, SegmentReceipt

_GET = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_char_p, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.c_void_p, C.c_size_t)


class SlowStore:
    """The in-memory hot store with `delay` seconds added to every GET of a segment (set/unlink/free pass straight through)."""

    def __init__(self, delay, fail_key=None):
        self.inner = ag.HotStore()
        self.delay, self.fail_key = delay, fail_key
        self.gets = []  # (key, t_start, t_end)
        self.mu = threading.Lock()
        inner_get = _GET(self.inner.ops.get)
        inner_user = self.inner.ops.user

        def get(_user, key, value, n, errbuf, cap):
            t0 = time.monotonic()
            if b":segments:" in key:
                time.sleep(self.delay)
            if self.fail_key and key == self.fail_key:
                msg = b"connection reset by peer (injected)"[: max(cap - 1, 0)]
                C.memmove(errbuf, msg + b"\0", len(msg) + 1)
                return -1
            rc = inner_get(inner_user, key, value, n, errbuf, cap)
            with self.mu:
                self.gets.append((key.decode(), t0, time.monotonic()))
            return rc

        self._get = _GET(get)
        self.ops = ag._HotStoreOps(None, C.cast(self._get, C.c_void_p).value, self.inner.ops.free_value, self.inner.ops.set_ex,
                                   self.inner.ops.unlink)
        # set/unlink/free are the inner store's functions: they need the inner store's `user`; get ignores its own
        self.ops.user = inner_user

    def __getattr__(self, name):
        return getattr(self.inner, name)


class SleepyProver:
    def __init__(self, seconds):
        self.seconds = seconds
        self.proofs = []  # (index, t_start, t_end)
        self.mu = threading.Lock()

    def prove_segment(self, seg):
        t0 = time.monotonic()
        time.sleep(self.seconds)
        with self.mu:
            self.proofs.append((seg.index, t0, time.monotonic()))
        return SegmentReceipt(seal=(np.arange(16, dtype=np.uint32) + np.uint32(seg.seed & 0xFFFF)), index=seg.index, po2=seg.po2)


def run(prefetch, n=12, lanes=1, get_s=0.03, prove_s=0.03, fail_key=None):
    store = SlowStore(get_s, fail_key=fail_key)
    prover = SleepyProver(prove_s)
    a = ag.Agent(prover=prover, verify=False, poll_time=0.002, inflight=lanes, store=store, prefetch=prefetch)
    try:
        for i in range(n):
            store.set_key_with_expiry(f"job:P:segments:{i}", ag.serialize_segment(Segment.synthetic(i, po2=13)), 600)
            a.taskdb.create_task("P", str(i), {"Prove": {"index": i}}, max_retries=0)
        t0 = time.monotonic()
        done = a.poll_work(max_idle_polls=3)
        wall = time.monotonic() - t0
        rows = {i: a.taskdb.task("P", str(i)) for i in range(n)}
        keys = set(store.keys())
        receipts = {i: store.get(k) if (k := f"job:P:synthetic_receipts:{i}") in keys else None for i in range(n)}
        return done, wall, store, prover, rows, receipts
    finally:
        a.close()


def overlapped_pairs(store, prover):
    """(proof, get) pairs whose time intervals intersect."""
    n = 0
    for _, p0, p1 in prover.proofs:
        for key, g0, g1 in store.gets:
            if ":segments:" in key and g0 < p1 and p0 < g1:
                n += 1
    return n


def test_prefetch_overlaps_the_next_get_with_the_current_proof():
    n = 12
    done0, wall0, store0, prover0, rows0, rec0 = run(False, n)
    done1, wall1, store1, prover1, rows1, rec1 = run(True, n)
    assert done0 == n and done1 == n
    assert all(r.state == "done" for r in rows0.values()) and all(r.state == "done" for r in rows1.values())
    assert rec0 == rec1 and all(v is not None for v in rec1.values())  # the same receipts either way
    assert sorted(i for i, _, _ in prover1.proofs) == list(range(n))  # every segment proved exactly once
    assert len([k for k, _, _ in store1.gets if ":segments:" in k]) == n  # and fetched exactly once
    # serial: no GET runs during a proof.  prefetch: (nearly) every proof has the next GET under it
    assert overlapped_pairs(store0, prover0) == 0
    assert overlapped_pairs(store1, prover1) >= n - 3
    # n * (get + prove) against get + n * max(get, prove): 0.72 s vs 0.39 s nominal; the bound is loose on purpose
    assert wall0 >= n * 0.06 * 0.95
    assert wall1 < 0.8 * wall0, (wall0, wall1)


def test_prefetch_with_several_lanes_loses_and_duplicates_nothing():
    n = 25
    done, _, store, prover, rows, receipts = run(True, n, lanes=4, get_s=0.004, prove_s=0.003)
    assert done == n
    assert sorted(i for i, _, _ in prover.proofs) == list(range(n))
    assert all(r.state == "done" and r.retries == 0 for r in rows.values())
    assert all(v is not None for v in receipts.values())


def test_a_get_that_fails_in_the_fetcher_fails_that_task_only_with_the_serial_paths_error():
    fail = b"job:P:segments:3"
    errs = []
    for prefetch in (False, True):
        done, _, store, prover, rows, receipts = run(prefetch, 6, get_s=0.002, prove_s=0.002, fail_key=fail)
        assert done == 5
        assert rows[3].state == "failed" and all(rows[i].state == "done" for i in range(6) if i != 3)
        assert receipts[3] is None
        errs.append(rows[3].error)
    assert errs[0] == errs[1]
    assert "segment data not found for segment key: job:P:segments:3" in errs[0] and "connection reset by peer (injected)" in errs[0]


def test_stop_does_not_strand_a_task_the_fetcher_claimed():
    """bx_agent_stop while the fetcher holds a claimed task: the lane runs it before it exits — nothing stays 'running'."""
    store = SlowStore(0.01)
    prover = SleepyProver(0.02)
    a = ag.Agent(prover=prover, verify=False, poll_time=0.002, inflight=2, store=store, prefetch=True)
    try:
        n = 40
        for i in range(n):
            store.set_key_with_expiry(f"job:P:segments:{i}", ag.serialize_segment(Segment.synthetic(i, po2=13)), 600)
            a.taskdb.create_task("P", str(i), {"Prove": {"index": i}}, max_retries=0)
        threading.Timer(0.15, a.stop).start()
        done = a.poll_work()
        states = [a.taskdb.task("P", str(i)).state for i in range(n)]
        assert 0 < done < n
        assert states.count("running") == 0 and states.count("done") == done and states.count("ready") == n - done
    finally:
        a.close()


def test_prefetch_over_the_rest_worker_protocol_hides_the_download():
    """The case it exists for: claims, GETs, PUTs and status reports go over HTTP (include/bx_rest.h) and every segment GET is a
    slow download.  Same requests, same final state; the lane no longer waits for a download while it has nothing to prove."""
    job = "0b1e55ed-0000-4000-8000-0000000000fe"
    n, get_s, prove_s = 10, 0.04, 0.04
    walls, logs = [], []
    for prefetch in (False, True):
        server = StubServer()
        try:
            st = server.state
            st.get_delay = get_s
            for i in range(n):
                st.hot[f"job:{job}:segments:{i}"] = (ag.serialize_segment(Segment.synthetic(i, po2=12)), None)
                st.create_task("prove", job, f"prove-{i}", {"Prove": {"index": i}}, max_retries=1)
            w = ag.RestWorker(server.url, claim_wait_secs=0)
            prover = SleepyProver(prove_s)
            a = ag.Agent(prover=prover, verify=False, poll_time=0.005, inflight=1, store=w.store, taskdb=w.taskdb, prefetch=prefetch)
            try:
                t0 = time.monotonic()
                assert a.poll_work(max_idle_polls=2) == n
                walls.append(time.monotonic() - t0)
            finally:
                a.close()
                w.close()
            assert [t["state"] for t in st.tasks] == ["done"] * n
            assert sorted(st.hot) == sorted(f"job:{job}:synthetic_receipts:prove-{i}" for i in range(n))
            assert sorted(i for i, _, _ in prover.proofs) == list(range(n))
            logs.append(sorted(r for r in st.log if "/claim/" not in r[1]))  # the number of empty polls differs, nothing else
        finally:
            server.close()
    assert logs[0] == logs[1]
    assert walls[0] >= n * (get_s + prove_s) * 0.95
    assert walls[1] < 0.8 * walls[0], walls


# ------------------------------------------------------------------------------------------------- the requeue monitor
class HangingProver:
    """The first attempt at `hang_index` takes `hang_s` (a wedged device); everything else is quick."""

    def __init__(self, hang_index, hang_s):
        self.hang_index, self.hang_s = hang_index, hang_s
        self.calls = []
        self.mu = threading.Lock()

    def prove_segment(self, seg):
        with self.mu:
            first = (seg.index == self.hang_index) and not any(i == seg.index for i, _ in self.calls)
            self.calls.append((seg.index, time.monotonic()))
        time.sleep(self.hang_s if first else 0.005)
        return SegmentReceipt(seal=(np.arange(16, dtype=np.uint32) + np.uint32(seg.seed & 0xFFFF)), index=seg.index, po2=seg.po2)


def test_the_requeue_monitor_hands_a_hung_lanes_task_to_another_lane():
    """--monitor-requeue (bento/crates/workflow/src/lib.rs:101-103,283-303 -> poll_for_requeue :536-551 -> taskdb::requeue_tasks): a task
    that stayed 'running' past its timeout_secs goes back to 'ready' with one more retry and the other lane proves it, long before
    the wedged lane returns; the late finisher's update_task_done finds the task done already and changes nothing."""
    n, hang_s = 8, 2.5
    prover = HangingProver(hang_index=2, hang_s=hang_s)
    a = ag.Agent(prover=prover, verify=False, poll_time=0.01, inflight=2, monitor_requeue=True, requeue_poll_interval=0.05)
    try:
        for i in range(n):
            a.store.set_key_with_expiry(f"job:Q:segments:{i}", ag.serialize_segment(Segment.synthetic(i, po2=13)), 600)
            a.taskdb.create_task("Q", str(i), {"Prove": {"index": i}}, max_retries=2, timeout_secs=1)
        done_at = {}

        def watch():
            t0 = time.monotonic()
            while time.monotonic() - t0 < 10 and len(done_at) < 1:
                if a.taskdb.job("Q")["state"] == "done":
                    done_at["job"] = time.monotonic() - t0
                time.sleep(0.01)

        w = threading.Thread(target=watch)
        w.start()
        t0 = time.monotonic()
        done = a.poll_work(max_idle_polls=150)  # the healthy lane keeps polling (1.5 s of idleness) while the other one is wedged
        wall = time.monotonic() - t0
        w.join()
        rows = [a.taskdb.task("Q", str(i)) for i in range(n)]
        assert all(r.state == "done" for r in rows) and a.taskdb.job("Q")["state"] == "done"
        assert [r.retries for r in rows] == [0, 0, 1, 0, 0, 0, 0, 0]  # requeued once, by the monitor
        assert sorted(i for i, _ in prover.calls) == sorted(list(range(n)) + [2])  # proved twice: by the wedged lane and by the other
        assert done >= n
        assert 1.0 < done_at["job"] < hang_s - 0.5  # the job was done ~1 s (the timeout) in, not when the wedged lane came back
        assert wall >= hang_s * 0.95  # poll_work itself waits for its lanes
        assert sorted(a.store.keys()) == sorted(f"job:Q:synthetic_receipts:{i}" for i in range(n))
    finally:
        a.close()


def test_without_the_monitor_a_hung_lane_keeps_its_task():
    prover = HangingProver(hang_index=1, hang_s=1.6)
    a = ag.Agent(prover=prover, verify=False, poll_time=0.002, inflight=2)
    try:
        for i in range(4):
            a.store.set_key_with_expiry(f"job:Q:segments:{i}", ag.serialize_segment(Segment.synthetic(i, po2=13)), 600)
            a.taskdb.create_task("Q", str(i), {"Prove": {"index": i}}, max_retries=2, timeout_secs=1)
        assert a.poll_work(max_idle_polls=3) == 4
        assert [a.taskdb.task("Q", str(i)).retries for i in range(4)] == [0, 0, 0, 0]
        assert sorted(i for i, _ in prover.calls) == [0, 1, 2, 3]
    finally:
        a.close()
