"""CPU: the next-gen Bento worker protocol (include/bx_rest.h) against a local stub of the API's worker routes.

The native agent (C++ feed loop) claims, proves through an injected prover, stores and reports over HTTP exactly as
prover/crates/workflow does (assets.rs:193-420): same URLs, same JSON bodies, same status handling.
"""
import json
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from rest_stub_server import StubServer  # noqa: E402

from boundless_amd import agent as ag  # noqa: E402
from boundless_amd.hal import HalError  # noqa: E402
from boundless_amd.prover import Segment, SegmentReceipt  # noqa: E402

JOB = "0b1e55ed-0000-4000-8000-00000000c0de"


class FakeProver:
    def __init__(self, fail_times=0):
        self.calls, self.fail_times = 0, fail_times

    def prove_segment(self, seg):
        self.calls += 1
        if self.calls <= self.fail_times:
            raise RuntimeError("hipErrorLaunchFailure (injected)")
        return SegmentReceipt(seal=np.arange(10, dtype=np.uint32) + seg.index, index=seg.index, po2=seg.po2)


@pytest.fixture()
def server():
    s = StubServer()
    yield s
    s.close()


def test_claim_prove_store_done_over_http(server):
    st = server.state
    for i in range(6):
        st.hot[f"job:{JOB}:segments:{i}"] = (ag.serialize_segment(Segment.synthetic(i, po2=12)), None)
        st.create_task("prove", JOB, f"prove-{i}", {"Prove": {"index": i}}, max_retries=2)
    st.create_task("join", JOB, "join-1", {"Join": {"idx": 1, "left": 2, "right": 3}})
    w = ag.RestWorker(server.url, claim_wait_secs=0)
    a = ag.Agent(prover=FakeProver(), verify=False, poll_time=0.01, inflight=2, store=w.store, taskdb=w.taskdb, redis_ttl=3600)
    try:
        assert a.poll_work(max_idle_polls=2) == 6
    finally:
        a.close()
    assert sorted(st.hot) == sorted(f"job:{JOB}:synthetic_receipts:prove-{i}" for i in range(6))  # receipts stored, segments deleted
    for i in range(6):
        blob, deadline = st.hot[f"job:{JOB}:synthetic_receipts:prove-{i}"]
        rec = ag.deserialize_receipt(blob)
        assert rec.index == i and np.array_equal(rec.seal, np.arange(10, dtype=np.uint32) + i)
        assert deadline is not None  # PUT ...?ttl_secs=3600 (hot_set_bytes with a TTL, assets.rs:386-399)
    assert [t["state"] for t in st.tasks] == ["done"] * 6 + ["ready"]  # the join task is another stream's
    assert all(t["output"] is None for t in st.tasks[:6])  # {"output": null}: the prove task returns ()
    paths = [p for _, p in st.log]
    assert ("POST", "/worker/gpu/tasks/claim/prove") in st.log
    assert ("POST", f"/worker/gpu/tasks/{JOB}/prove-3/done") in st.log
    assert ("GET", f"/worker/hot/job:{JOB}:segments:3") in st.log and ("DELETE", f"/worker/hot/job:{JOB}:segments:3") in st.log
    assert ("PUT", f"/worker/hot/job:{JOB}:synthetic_receipts:prove-3") in st.log
    assert w.requests == len(paths)
    w.close()


def test_retry_failed_and_missing_blob_semantics_over_http(server):
    st = server.state
    st.hot[f"job:{JOB}:segments:0"] = (ag.serialize_segment(Segment.synthetic(0, po2=10)), None)
    st.create_task("prove", JOB, "flaky", {"Prove": {"index": 0}}, max_retries=3)
    st.create_task("prove", JOB, "missing", {"Prove": {"index": 7}}, max_retries=1)
    st.create_task("prove", JOB, "nonsense", {"Nope": {}}, max_retries=0)
    w = ag.RestWorker(server.url)
    p = FakeProver(fail_times=2)
    a = ag.Agent(prover=p, verify=False, poll_time=0.01, store=w.store, taskdb=w.taskdb)
    try:
        assert a.poll_work(max_idle_polls=2) == 1
    finally:
        a.close()
    by = {t["task_id"]: t for t in st.tasks}
    assert (by["flaky"]["state"], by["flaky"]["retries"]) == ("done", 2)  # retries-running -> retry -> claimed again (lib.rs:381-436)
    assert by["missing"]["state"] == "failed" and by["missing"]["retries"] == 1
    # 404 HotDataMissing arrives as the reference's nil-key error chain
    assert by["missing"]["error"].startswith("retry max hit: [BENTO-WF-115] Prove failed: segment data not found for segment key: "
                                             f"job:{JOB}:segments:7: Key not found (nil response)")
    assert by["nonsense"]["state"] == "failed" and by["nonsense"]["error"] == f"Invalid task_def: {JOB}:nonsense"
    assert ("GET", f"/worker/gpu/tasks/{JOB}/flaky/retries-running") in st.log and ("POST", f"/worker/gpu/tasks/{JOB}/flaky/retry") in st.log
    w.close()


def test_error_text_is_json_escaped_and_keys_are_path_encoded(server):
    st = server.state
    odd_task = 'we ird"task\\1'
    st.hot[f"job:{JOB}:segments:0"] = (ag.serialize_segment(Segment.synthetic(0, po2=10)), None)
    st.create_task("prove", JOB, odd_task, {"Prove": {"index": 0}}, max_retries=0)

    class Noisy:
        def prove_segment(self, seg):
            raise RuntimeError('line1\n"quoted" \\ tab\t end')

    w = ag.RestWorker(server.url)
    a = ag.Agent(prover=Noisy(), verify=False, poll_time=0.01, store=w.store, taskdb=w.taskdb)
    try:
        assert a.poll_work(max_idle_polls=1) == 0
    finally:
        a.close()
    t = st.tasks[0]
    assert t["state"] == "failed" and t["error"] == '[BENTO-WF-115] Prove failed: line1\n"quoted" \\ tab\t end'
    w.close()


def test_transport_and_server_errors_end_the_loop_like_the_reference(server):
    """A failing claim is fatal to poll_work (the reference `?`-returns with BENTO-WF-105/107); the agent object survives."""
    w = ag.RestWorker(server.url)
    a = ag.Agent(prover=FakeProver(), verify=False, poll_time=0.01, store=w.store, taskdb=w.taskdb)
    server.state.fail_next = 1
    with pytest.raises(HalError, match=r"\[BENTO-WF-107\] Failed to request_work: GPU work claim failed for stream prove: HTTP 500"):
        a.poll_work(max_idle_polls=1)
    assert a.poll_work(max_idle_polls=1) == 0  # the next poll works again
    a.close()
    w.close()
    dead = ag.RestWorker("http://127.0.0.1:1")  # nothing listens there
    b = ag.Agent(prover=FakeProver(), verify=False, poll_time=0.01, store=dead.store, taskdb=dead.taskdb)
    with pytest.raises(HalError, match="failed to claim GPU work for stream prove: connect 127.0.0.1:1"):
        b.poll_work(max_idle_polls=1)
    b.close()
    dead.close()
    for bad in ("", "https://x", "http://", "http://host:port"):
        with pytest.raises(HalError):
            ag.RestWorker(bad)


def test_long_poll_claim_and_invalid_stream(server):
    import threading
    import time

    st = server.state
    w = ag.RestWorker(server.url, claim_wait_secs=2)
    a = ag.Agent(prover=FakeProver(), verify=False, poll_time=0.01, store=w.store, taskdb=w.taskdb)
    st.hot[f"job:{JOB}:segments:0"] = (ag.serialize_segment(Segment.synthetic(0, po2=10)), None)

    def later():
        time.sleep(0.3)
        st.create_task("prove", JOB, "late", {"Prove": {"index": 0}})

    threading.Thread(target=later).start()
    t0 = time.time()
    assert a.poll_work(max_idle_polls=1) == 1  # the claim blocks server-side (request_work_wait) until the task appears
    assert 0.25 < time.time() - t0 < 6
    a.close()
    bad = ag.Agent(prover=FakeProver(), verify=False, poll_time=0.01, store=w.store, taskdb=w.taskdb, task_stream="exec")
    with pytest.raises(HalError, match="HTTP 400"):  # AppError::InvalidGpuWorkerStream
        bad.poll_work(max_idle_polls=1)
    bad.close()
    w.close()


def test_a_long_non_ascii_error_still_reaches_the_server_as_valid_json(server):
    """The 1024-byte truncation (lib.rs:424) must not cut a multi-byte UTF-8 sequence in half, and bytes that are not UTF-8 at
    all must not make the failure report itself fail (the lane used to stop with [BENTO-WF-112])."""
    st = server.state
    for k, (task, text) in enumerate((("t-utf8", "é" * 2000), ("t-bytes", None))):
        st.hot[f"job:{JOB}:segments:{k}"] = (ag.serialize_segment(Segment.synthetic(k, po2=10)), None)
        st.create_task("prove", JOB, task, {"Prove": {"index": k}}, max_retries=0)

    class Noisy:
        def prove_segment(self, seg):
            if seg.index == 0:
                raise RuntimeError("é" * 2000)
            raise RuntimeError(b"bad \xff\xfe bytes \xc3".decode("latin-1").encode("latin-1").decode("utf-8", "surrogateescape"))

    w = ag.RestWorker(server.url)
    a = ag.Agent(prover=Noisy(), verify=False, poll_time=0.01, store=w.store, taskdb=w.taskdb)
    try:
        assert a.poll_work(max_idle_polls=1) == 0
    finally:
        a.close()
    by = {t["task_id"]: t for t in st.tasks}
    assert by["t-utf8"]["state"] == "failed" and by["t-bytes"]["state"] == "failed"
    e = by["t-utf8"]["error"]
    assert e.startswith("[BENTO-WF-115] Prove failed: é") and e.endswith("é") and len(e.encode()) in (1023, 1024)
    w.close()


def _store_ops(w):
    """The hot-store callback table of a RestWorker as plain Python calls (what the agent's lanes do through C)."""
    import ctypes as C

    ops = w.store.value if hasattr(w.store, "value") else w.store
    return ops, C


def test_connections_are_kept_alive_and_pooled(server):
    """30+ requests of a run travel over a handful of TCP connections (the reference shares one pooling reqwest client,
    assets.rs:76); with `Connection: close` on every answer the same run opens one connection per request and still works."""
    st = server.state
    for mode in ("keep-alive", "close"):
        st.keep_alive = mode == "keep-alive"
        st.tasks.clear(), st.hot.clear(), st.log.clear()
        for i in range(6):
            st.hot[f"job:{JOB}:segments:{i}"] = (ag.serialize_segment(Segment.synthetic(i, po2=10)), None)
            st.create_task("prove", JOB, f"p-{mode}-{i}", {"Prove": {"index": i}}, max_retries=0)
        w = ag.RestWorker(server.url)
        a = ag.Agent(prover=FakeProver(), verify=False, poll_time=0.01, inflight=2, store=w.store, taskdb=w.taskdb)
        try:
            assert a.poll_work(max_idle_polls=2) == 6
        finally:
            a.close()
        assert w.requests == len(st.log) >= 30
        if mode == "keep-alive":
            assert w.connects <= 6, (w.connects, w.requests)  # lanes + finishers, not requests
        else:
            assert w.connects == w.requests
        w.close()


def test_an_idle_connection_dropped_by_the_server_is_replaced_transparently(server):
    """A server-side idle timeout closes a pooled connection without notice; the request that finds it dead (no byte of an
    answer yet) goes out again on a fresh connection instead of failing the task."""
    st = server.state
    st.drop_every, st.drop_silently = 3, True
    for i in range(8):
        st.hot[f"job:{JOB}:segments:{i}"] = (ag.serialize_segment(Segment.synthetic(i, po2=10)), None)
        st.create_task("prove", JOB, f"d-{i}", {"Prove": {"index": i}}, max_retries=0)
    w = ag.RestWorker(server.url)
    a = ag.Agent(prover=FakeProver(), verify=False, poll_time=0.01, store=w.store, taskdb=w.taskdb)
    try:
        assert a.poll_work(max_idle_polls=2) == 8
    finally:
        a.close()
    assert [t["state"] for t in st.tasks] == ["done"] * 8
    assert 1 < w.connects < w.requests
    w.close()


def test_a_receive_timeout_on_a_pooled_connection_is_not_retried(server):
    """only a connection the server really CLOSED is replaced.  A request that times out on a pooled connection
    (SO_RCVTIMEO -> EAGAIN) is that request's failure: re-sending the claim POST on a fresh socket would double the timeout and could
    claim a task twice — the first claim may still complete server-side."""
    import ctypes as C
    import time

    st = server.state
    w = ag.RestWorker(server.url, io_timeout_secs=1)
    ops = w.taskdb.ops
    req = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_char_p, C.POINTER(ag._ReadyTask), C.c_char_p, C.c_size_t)(ops.request_work)
    t, eb = ag._ReadyTask(), C.create_string_buffer(256)
    assert req(ops.user, b"prove", C.byref(t), eb, 256) == 0  # opens the connection that goes back to the pool
    assert w.connects == 1
    st.log.clear()
    st.stall_next = 2.5  # the pooled connection is alive; the server just does not answer within the client's 1 s
    t0 = time.time()
    assert req(ops.user, b"prove", C.byref(t), eb, 256) == -1
    dt = time.time() - t0
    assert b"timed out" in eb.value, eb.value
    assert dt < 1.9, dt  # one timeout, not two
    time.sleep(2.0)  # let the stalled handler finish
    assert len([m for m in st.log if m[0] == "POST"]) == 1, st.log  # the request was sent ONCE
    assert w.connects == 1
    w.close()


def test_interim_1xx_responses_are_skipped_and_do_not_desync_the_pooled_connection(server):
    """a `100 Continue` is not the answer.  The real response that follows is the one returned, and the next request
    on the same pooled connection reads ITS OWN answer, not a leftover."""
    import ctypes as C

    st = server.state
    st.create_task("prove", JOB, "i-1", {"Prove": {"index": 4}}, max_retries=2)
    w = ag.RestWorker(server.url)
    ops = w.taskdb.ops
    req = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_char_p, C.POINTER(ag._ReadyTask), C.c_char_p, C.c_size_t)(ops.request_work)
    cur = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_char_p, C.c_char_p, C.POINTER(C.c_int32), C.c_char_p, C.c_size_t)(ops.current_retries)
    t, eb = ag._ReadyTask(), C.create_string_buffer(256)
    st.interim_next = 2
    assert req(ops.user, b"prove", C.byref(t), eb, 256) == 1 and t.task_id == b"i-1" and t.max_retries == 2
    n = C.c_int32(-1)
    assert cur(ops.user, JOB.encode(), b"i-1", C.byref(n), eb, 256) == 1 and n.value == 0  # same connection, its own answer
    assert w.connects == 1
    w.close()
