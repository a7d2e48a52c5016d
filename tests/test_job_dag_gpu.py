"""GPU: a planned job — 16 segment proofs, the log-depth tail of stand-in joins, resolve, finalize — through the HIP prover's lanes.

The Join tasks are STAND-INS for the recursion proofs (include/bx_agent.h: one synthetic segment seeded by the hash of the two
children's seals), so the rollup seal commits to every seal below it: recomputing that chain with the CPU oracle's prover checks
every one of the 31 proofs of the job word for word.  Reference flow: executor.rs:566-698 (planner -> task rows), join.rs:18-113,
resolve.rs, finalize.rs; prerequisites bento/crates/taskdb/migrations/1_taskdb.sql:197-228,296-306.
"""
import os

import numpy as np
import pytest

from oracle import oracle_lib as ol

pytestmark = pytest.mark.gpu


def oracle_chain(n, seg_po2, join_po2, widths, seed_of):
    from boundless_amd import agent as ag
    from boundless_amd.planner import Planner

    p = Planner()
    seals, root = {}, None

    def drain(i):
        nonlocal root
        while True:
            t = p.next_task()
            if t is None:
                return
            if t.command == "Segment":
                seals[t.task_number] = ol.prove_segment(seg_po2, *widths, seed_of(i))[0]
            elif t.command == "Join":
                l, r = t.depends_on
                seals[t.task_number] = ol.prove_segment(join_po2, *widths, ag.join_seed(seals[l], seals[r]))[0]
            else:
                root = t.depends_on[0]

    for i in range(n):
        p.enqueue_segment()
        drain(i)
    p.finish()
    drain(None)
    return root, seals


@pytest.mark.parametrize("n,lanes,devices", [(16, 3, None), (5, 2, None), (11, 2, [0, 0])])
def test_planned_job_on_the_gpu_equals_the_oracle_chain(n, lanes, devices):
    """devices = [0, 0]: one agent over two device slots (both on the one GPU of the test box) = the `--gpus N` form of bench.py --job:
    the lanes of every device claim proves and joins from the one task db."""
    from boundless_amd import agent as ag
    from boundless_amd.prover import Segment

    widths, seg_po2, join_po2 = (4, 8, 4), 12, 10
    a = ag.Agent(prover=None, device=0, devices=devices, inflight=lanes, widths=widths, poll_time=0.002, join_po2=join_po2, also_streams="aux")
    try:
        a.prewarm(seg_po2)  # buffer sets (and verifier-context entries) created up front on every lane
        a.prewarm(join_po2)
        with pytest.raises(Exception, match="outside the sizes this agent accepts"):
            a.prewarm(8)
        segs = [Segment.synthetic(i, po2=seg_po2) for i in range(n)]
        for s in segs:
            a.store.set_key_with_expiry(f"job:G:segments:{s.index}", ag.serialize_segment(s), 600)
        ids = a.taskdb.plan_job("G", n)
        assert a.poll_work(max_idle_polls=5) == len(ids) == 2 * n - 1 + 2
        assert a.taskdb.job("G")["state"] == "done"
        root, seals = oracle_chain(n, seg_po2, join_po2, widths, lambda i: segs[i].seed)
        rollup = ag.deserialize_receipt(a.store.get("receipts/stark/G.synthetic"))
        assert rollup.po2 == join_po2 and rollup.index == root
        assert np.array_equal(rollup.seal, seals[root])  # the root of the hash chain: all 2n-1 proofs were the oracle's
        rollup.verify_integrity()
        assert sorted(a.store.keys()) == sorted([f"job:G:synthetic_receipts:{root}", "receipts/stark/G.synthetic"])
        text = a.metrics_text()
        assert f'task_operations_total{{task_name="join",operation_type="join_receipts",status="success"}} {n - 1}' in text
        per_lane = [d for _, d in a.lane_stats()]
        assert sum(per_lane) == len(ids) and len(per_lane) == lanes * (len(devices) if devices else 1)
    finally:
        a.close()


def test_a_tampered_child_receipt_fails_the_join_at_verification():
    """join.rs:44-49: both children are verified before they are joined ([BENTO-JOIN-003/004])."""
    from boundless_amd import agent as ag
    from boundless_amd.prover import SegmentReceipt

    widths = (4, 8, 4)
    a = ag.Agent(prover=None, device=0, inflight=1, widths=widths, poll_time=0.002, join_po2=10)
    try:
        good, _ = ol.prove_segment(10, *widths, 1)
        bad = good.copy()
        bad[-1] ^= 1
        a.store.set_key_with_expiry("job:T:synthetic_receipts:0", ag.serialize_receipt(SegmentReceipt(seal=good, index=0, po2=10)), 600)
        a.store.set_key_with_expiry("job:T:synthetic_receipts:1", ag.serialize_receipt(SegmentReceipt(seal=bad, index=1, po2=10)), 600)
        a.taskdb.create_task("T", "2", {"Join": {"idx": 2, "left": 0, "right": 1}}, max_retries=0)
        assert a.process_one("T", "2", {"Join": {"idx": 2, "left": 0, "right": 1}}) is False
        assert a.taskdb.task("T", "2").error.startswith("[BENTO-WF-119] Join failed: [BENTO-JOIN-004] Failed to verify right receipt integrity")
        # with both children intact the join goes through and cleans up
        a.store.set_key_with_expiry("job:T:synthetic_receipts:1", ag.serialize_receipt(SegmentReceipt(seal=good, index=1, po2=10)), 600)
        a.taskdb.create_task("T", "3", {"Join": {"idx": 3, "left": 0, "right": 1}}, max_retries=0)
        assert a.process_one("T", "3", {"Join": {"idx": 3, "left": 0, "right": 1}}) is True
        joined = ag.deserialize_receipt(a.store.get("job:T:synthetic_receipts:3"))
        want, _ = ol.prove_segment(10, *widths, ag.join_seed(good, good))
        assert np.array_equal(joined.seal, want)
        assert a.store.keys() == ["job:T:synthetic_receipts:3"]
    finally:
        a.close()


def _bench_job(extra, dump, ranks=None, timeout=900):
    import json
    import os
    import socket
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    base = [os.path.join(root, "bench.py"), "--po2", "12", "--widths", "4,8,4", "--join-po2", "10", "--inflight", "2", "--dump", dump] + extra
    if ranks:
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={ranks}", "--master-addr", "127.0.0.1",
               "--master-port", str(port)] + base
    else:
        cmd = [sys.executable] + base
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=env, cwd=root)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    return out, np.load(os.path.join(dump, "rollup.npz"))["seal"]


def test_bench_job_one_process_and_process_per_gpu_give_the_oracles_rollup(tmp_path):
    """`bench.py --job 8` three ways on the one GPU of the test box: one process (the native agent over its lanes); two ranks under
    torchrun (gloo, both on device 0) = one process per GPU, every rank joining its own subtree, ONE all_gather of the two subtree
    roots, rank 0 joining them; and one forced rank under the default backend (nccl = RCCL: two ranks cannot share a GPU under it),
    which runs the same all_gather on the device.  8 and 2 are powers of two, so all three build the planner's tree: the same rollup
    seal, equal to the oracle's chain of 15 proofs."""
    from boundless_amd.prover import Segment

    root, seals = oracle_chain(8, 12, 10, (4, 8, 4), lambda i: Segment.synthetic(i, po2=12).seed)
    one, seal_one = _bench_job(["--job", "8"], str(tmp_path / "one"))
    assert one["join"] == "synthetic stand-in" and one["job"]["tasks"] == 17 and one["job"]["joins"] == 7
    assert np.array_equal(seal_one, seals[root])
    two, seal_two = _bench_job(["--job", "8", "--gpus", "2", "--dist-backend", "gloo", "--device", "0"], str(tmp_path / "two"), ranks=2)
    assert two["n_gpus"] == 2 and two["join"] == "synthetic stand-in" and two["collective"]["op"] == "all_gather"
    assert two["collective"]["world_size"] == 2 and two["job"]["top_joins"] == 1
    assert [r["segments"] for r in two["per_rank"]] == [4, 4] and [r["rank"] for r in two["per_rank"]] == [0, 1]
    assert np.array_equal(seal_two, seals[root])
    rccl, seal_rccl = _bench_job(["--job", "8", "--gpus", "1", "--force-dist"], str(tmp_path / "rccl"), ranks=1)
    assert rccl["collective"]["backend"] == "nccl" and rccl["collective"]["world_size"] == 1 and rccl["job"]["top_joins"] == 0
    assert np.array_equal(seal_rccl, seals[root])


def test_prove_tasks_with_the_stand_in_lift_leg_on_the_gpu():
    """cfg.lift_po2: prove_segment -> (verify beside) lift -> verify -> store the lifted receipt (prove.rs:41-113, the lift a labelled
    stand-in).  Five segments: the rollup is the oracle's chain over the LIFTED leaves — 5 + 5 + 4 proofs."""
    from boundless_amd import agent as ag
    from boundless_amd.planner import Planner
    from boundless_amd.prover import Segment

    widths, seg_po2, small = (4, 8, 4), 12, 10
    a = ag.Agent(prover=None, device=0, inflight=2, widths=widths, poll_time=0.002, join_po2=small, lift_po2=small, also_streams="aux")
    try:
        n = 5
        segs = [Segment.synthetic(i, po2=seg_po2) for i in range(n)]
        for s in segs:
            a.store.set_key_with_expiry(f"job:LG:segments:{s.index}", ag.serialize_segment(s), 600)
        ids = a.taskdb.plan_job("LG", n)
        assert a.poll_work(max_idle_polls=5) == len(ids) and a.taskdb.job("LG")["state"] == "done"
        p, seals, root, leaf = Planner(), {}, None, 0
        for _ in range(n):
            p.enqueue_segment()
        p.finish()
        for k in range(p.task_count()):
            t = p.get_task(k)
            if t.command == "Segment":
                seg_seal = ol.prove_segment(seg_po2, *widths, segs[leaf].seed)[0]
                seals[t.task_number] = ol.prove_segment(small, *widths, ag.join_seed(seg_seal, np.zeros(0, np.uint32)))[0]
                leaf += 1
        for k in range(p.task_count()):
            t = p.get_task(k)
            if t.command == "Join":
                seals[t.task_number] = ol.prove_segment(small, *widths, ag.join_seed(seals[t.depends_on[0]], seals[t.depends_on[1]]))[0]
            elif t.command == "Finalize":
                root = t.depends_on[0]
        rollup = ag.deserialize_receipt(a.store.get("receipts/stark/LG.synthetic"))
        assert np.array_equal(rollup.seal, seals[root])
        assert f'task_operations_total{{task_name="prove",operation_type="lift",status="success"}} {n}' in a.metrics_text()
    finally:
        a.close()


class _RecordingStore:
    """The library's in-memory hot store behind a `bx_hot_store_ops` table whose SETEX also keeps a copy of the values of chosen keys:
    a planned job unlinks every intermediate receipt as soon as its parent has joined it, and this test wants to look at three of them.
    Test-side only (ctypes callbacks that forward to the native function pointers); the product's store and agent are unchanged."""

    def __init__(self, watch):
        import ctypes as C

        from boundless_amd import agent as ag

        self.inner = ag.HotStore()
        self.watch, self.seen = set(watch), {}
        io = self.inner.ops
        GET = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_char_p, C.POINTER(C.POINTER(C.c_uint8)), C.POINTER(C.c_size_t), C.c_void_p, C.c_size_t)
        FREE = C.CFUNCTYPE(None, C.c_void_p, C.POINTER(C.c_uint8))
        SET = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_char_p, C.POINTER(C.c_uint8), C.c_size_t, C.c_uint64, C.c_void_p, C.c_size_t)
        UNLINK = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_char_p, C.c_void_p, C.c_size_t)
        get, free, set_ex, unlink = GET(io.get), FREE(io.free_value), SET(io.set_ex), UNLINK(io.unlink)
        user = io.user

        def rec_set(_u, key, value, n, ttl, errbuf, cap):
            k = key.decode()
            if k in self.watch:
                self.seen[k] = C.string_at(value, n)
            return set_ex(user, key, value, n, ttl, errbuf, cap)

        self._keep = (GET(lambda _u, k, v, n, e, c: get(user, k, v, n, e, c)), FREE(lambda _u, p: free(user, p)), SET(rec_set),
                      UNLINK(lambda _u, k, e, c: unlink(user, k, e, c)))
        self.ops = ag._HotStoreOps(None, *[C.cast(f, C.c_void_p) for f in self._keep])
        for name in ("get", "set_key_with_expiry", "unlink", "keys"):
            setattr(self, name, getattr(self.inner, name))


@pytest.mark.fullsize
def test_batch_of_64_segments_at_2_20_cycles_through_the_native_agent():
    """BASELINE.json configs[2] at the metric's size: 64 independent 2^20-cycle segments (16/256/64) planned as one job
    (bento/crates/taskdb/src/planner/mod.rs:91-116) and work-stolen by the three lanes of the native agent, verification on: 64 Prove
    + 63 stand-in Join (2^18 synthetic proofs, NOT recursion proofs) + Resolve + Finalize = 129 tasks.  Every receipt is verified by
    the agent before it is stored (a failure would fail its task and the job), the job ends `done`, the lanes' counts add up, and
    three stored receipts are the oracle's word for word: the Prove seals of segments 0 and 1 — the first join's two inputs — and
    that join's output, seeded by the hash of both.  (The whole chain of 127 oracle proofs would take half an hour; the small-size
    tests above check it in full.)  Deselect locally with -m "gpu and not fullsize"."""
    from boundless_amd import agent as ag
    from boundless_amd.planner import Planner
    from boundless_amd.prover import Segment

    n, lanes, widths, seg_po2, join_po2 = 64, 3, (16, 256, 64), 20, 18
    p = Planner()
    for _ in range(n):
        p.enqueue_segment()
    p.finish()
    tasks = [p.get_task(k) for k in range(p.task_count())]
    first_join = next(t for t in tasks if t.command == "Join" and tuple(t.depends_on) == (0, 1))
    watch = [f"job:B:synthetic_receipts:{i}" for i in (0, 1, first_join.task_number)]
    store = _RecordingStore(watch)
    a = ag.Agent(prover=None, device=0, inflight=lanes, widths=widths, poll_time=0.002, join_po2=join_po2, also_streams="aux", store=store,
                 verify=True)
    try:
        a.prewarm(seg_po2)
        a.prewarm(join_po2)
        segs = [Segment.synthetic(i, po2=seg_po2) for i in range(n)]
        for s in segs:
            store.set_key_with_expiry(f"job:B:segments:{s.index}", ag.serialize_segment(s), 600)
        ids = a.taskdb.plan_job("B", n)
        assert len(ids) == 129
        assert a.poll_work(max_idle_polls=5) == 129
        assert a.taskdb.job("B")["state"] == "done"
        per_lane = [d for _, d in a.lane_stats()]
        assert len(per_lane) == lanes and sum(per_lane) == 129 and min(per_lane) > 0
        text = a.metrics_text()
        assert f'task_operations_total{{task_name="join",operation_type="join_receipts",status="success"}} {n - 1}' in text
        assert 'status="failed"' not in text and 'status="error"' not in text
        rollup = ag.deserialize_receipt(store.get("receipts/stark/B.synthetic"))
        assert rollup.po2 == join_po2
        rollup.verify_integrity()
        assert sorted(store.keys()) == sorted([f"job:B:synthetic_receipts:{rollup.index}", "receipts/stark/B.synthetic"])
        assert sorted(store.seen) == sorted(watch)
        got = {k: ag.deserialize_receipt(v) for k, v in store.seen.items()}
        ol.lib().bxo_set_threads(min(os.cpu_count() or 1, 16))
        s0, _ = ol.prove_segment(seg_po2, *widths, segs[0].seed)
        assert np.array_equal(got[watch[0]].seal, s0), "Prove seal of segment 0 differs from the oracle's"
        s1, _ = ol.prove_segment(seg_po2, *widths, segs[1].seed)
        assert np.array_equal(got[watch[1]].seal, s1), "Prove seal of segment 1 differs from the oracle's"
        j, _ = ol.prove_segment(join_po2, *widths, ag.join_seed(s0, s1))
        assert got[watch[2]].po2 == join_po2 and np.array_equal(got[watch[2]].seal, j), "first join's output differs from the oracle's"
    finally:
        a.close()
