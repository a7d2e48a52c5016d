"""Round-4 soak on one MI355X (not part of the test suite; `PYTHONPATH=. python tools/soak_r04.py [out.json]` on the GPU box).

1. The bytes boundary, two deep: three provers, each with its own feeder thread submitting segments of RANDOM payload sizes
   (0 .. 6 MB, so the staging slots grow and are reused) while the previous one is proved; 4 500 proofs at po2 14; every seal equals the
   seal of the same seed proved through the seed entry point, and verifies.
2. Planned jobs through the native agent: 60 jobs of 9..40 segments (po2 12, stand-in joins at po2 10) one after the other on the
   same agent (buffer sets reused, verifier context filled once); every job ends `done`, every rollup verifies; the hot store holds
   exactly two keys per job afterwards.
3. Device memory: free HBM before == after (provers, staging slots, copy streams, agent contexts all released).
"""
import json
import queue
import sys
import threading
import time

import numpy as np
import torch

from boundless_amd import agent as ag
from boundless_amd.prover import HipProverServer, Segment, verify_seal


def free_bytes():
    torch.cuda.synchronize()
    return torch.cuda.mem_get_info(0)[0]


def two_deep(po2, widths, n, lanes):
    servers = [HipProverServer(0, po2=po2, widths=widths) for _ in range(lanes)]
    want = {}
    ref = HipProverServer(0, po2=po2, widths=widths)
    for i in range(n):
        want[i] = ref.prove_segment(Segment.synthetic(i, po2=po2)).seal
    ref.close()
    bad = []

    def lane(l, sv):
        rng = np.random.default_rng(l)
        mine = list(range(l, n, lanes)) * 3
        handed, free = queue.Queue(), threading.Semaphore(2)

        def feeder():
            for i in mine:
                free.acquire()
                seg = Segment.synthetic(i, po2=po2)
                seg.payload = bytes(int(rng.integers(0, 6_000_000)))
                sv.submit_segment(seg.to_bytes())
                handed.put(i)
            handed.put(None)

        t = threading.Thread(target=feeder)
        t.start()
        while True:
            i = handed.get()
            if i is None:
                break
            seal = sv.prove_submitted(index=i).seal
            free.release()
            if not np.array_equal(seal, want[i]):
                bad.append(i)
        t.join()

    t0 = time.time()
    ts = [threading.Thread(target=lane, args=(l, sv)) for l, sv in enumerate(servers)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    dt = time.time() - t0
    for s in servers:
        s.close()
    for i in range(0, n, 7):
        verify_seal(want[i])
    return {"proofs": 3 * n, "mismatches": len(bad), "seconds": round(dt, 2)}


def jobs():
    a = ag.Agent(prover=None, device=0, inflight=3, widths=(4, 8, 4), poll_time=0.002, join_po2=10, also_streams="aux")
    rng = np.random.default_rng(4)
    done_jobs, tasks = 0, 0
    t0 = time.time()
    try:
        for j in range(60):
            k = int(rng.integers(9, 41))
            for i in range(k):
                a.store.set_key_with_expiry(f"job:S{j}:segments:{i}", ag.serialize_segment(Segment.synthetic(1000 * j + i, po2=12)), 600)
            ids = a.taskdb.plan_job(f"S{j}", k)
            assert a.poll_work(max_idle_polls=3) == len(ids)
            assert a.taskdb.job(f"S{j}")["state"] == "done"
            ag.deserialize_receipt(a.store.get(f"receipts/stark/S{j}.synthetic")).verify_integrity()
            done_jobs += 1
            tasks += len(ids)
        keys = a.store.keys()
        assert len(keys) == 2 * 60, keys
    finally:
        a.close()
    return {"jobs": done_jobs, "tasks": tasks, "seconds": round(time.time() - t0, 2)}


def warm():
    """One cycle of everything first: the runtime's one-time allocations (code objects, its pools: ~0.2 GB, constant afterwards) are
    not what this soak is looking for."""
    sv = HipProverServer(0, po2=14, widths=(4, 24, 8))
    sv.prove_segment(Segment.synthetic(0, po2=14))
    sv.close()
    a = ag.Agent(prover=None, device=0, inflight=3, widths=(4, 8, 4), poll_time=0.002, join_po2=10, also_streams="aux")
    for i in range(6):
        a.store.set_key_with_expiry(f"job:W:segments:{i}", ag.serialize_segment(Segment.synthetic(i, po2=12)), 600)
    a.taskdb.plan_job("W", 6)
    a.poll_work(max_idle_polls=3)
    a.close()


def main(out):
    warm()
    before = free_bytes()
    res = {"two_deep_bytes_po2_14": two_deep(14, (4, 24, 8), 1500, 3)}
    assert res["two_deep_bytes_po2_14"]["mismatches"] == 0
    res["planned_jobs_po2_12"] = jobs()
    after = free_bytes()
    res["hbm_free_before"], res["hbm_free_after"] = before, after
    assert abs(before - after) < (64 << 20), (before, after)
    print(json.dumps(res))
    if out:
        json.dump(res, open(out, "w"), indent=1)


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else None)
