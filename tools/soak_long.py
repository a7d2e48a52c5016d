"""Long soak at the metric's size (`PYTHONPATH=. python tools/soak_long.py SECONDS [out.json]`): three provers prove segments
0, 1, 2, ... of 2^20 cycles for SECONDS; every seal is verified on the CPU by a pool of verifier threads; every 8th segment is
proved a second time by a fourth prover and the two seals compared word for word."""
import json
import queue
import sys
import threading
import time

import numpy as np

from boundless_amd.prover import HipProverServer, Segment, verify_seal

seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
out = sys.argv[2] if len(sys.argv) > 2 else None
lanes = 3
servers = [HipProverServer(0) for _ in range(lanes)]
todo = queue.Queue()
stats = {"proved": 0, "verified": 0, "rechecked": 0, "mismatches": 0, "verify_failures": 0}
lock = threading.Lock()
keep = {}
deadline = time.time() + seconds
counter = [0]


def prove(l):
    while time.time() < deadline:
        with lock:
            i = counter[0]
            counter[0] += 1
        seal = servers[l].prove_segment(Segment.synthetic(i)).seal
        with lock:
            stats["proved"] += 1
            if i % 8 == 0:
                keep[i] = seal
        todo.put((i, seal))


def recheck():
    # a fourth prover re-proves every 8th segment and compares
    srv = HipProverServer(0)
    done = set()
    while time.time() < deadline or len(done) < len(keep):
        with lock:
            pending = [i for i in keep if i not in done]
        if not pending:
            time.sleep(0.05)
            if time.time() > deadline + 30:
                break
            continue
        i = pending[0]
        seal = srv.prove_segment(Segment.synthetic(i)).seal
        with lock:
            stats["rechecked"] += 1
            stats["mismatches"] += int(not np.array_equal(seal, keep[i]))
        done.add(i)
    srv.close()


def verifier():
    while True:
        item = todo.get()
        if item is None:
            return
        try:
            verify_seal(item[1])
            with lock:
                stats["verified"] += 1
        except Exception:
            with lock:
                stats["verify_failures"] += 1


vt = [threading.Thread(target=verifier) for _ in range(6)]
[t.start() for t in vt]
t0 = time.time()
pt = [threading.Thread(target=prove, args=(l,)) for l in range(lanes)]
rt = threading.Thread(target=recheck)
[t.start() for t in pt]
rt.start()
[t.join() for t in pt]
elapsed = time.time() - t0
rt.join()
for _ in vt:
    todo.put(None)
[t.join() for t in vt]
for s in servers:
    s.close()
stats.update({"seconds": round(elapsed, 1), "segment_proofs_per_s_incl_recheck_lane": round((stats["proved"] + stats["rechecked"]) / elapsed, 2)})
print(json.dumps(stats))
assert stats["mismatches"] == 0 and stats["verify_failures"] == 0 and stats["verified"] == stats["proved"]
if out:
    json.dump(stats, open(out, "w"), indent=1)
