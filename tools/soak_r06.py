"""Round-6 soak at the metric's size (`PYTHONPATH=. python tools/soak_r06.py SECONDS [out.json]`): two in-library provers
(bx_prove_segment) and two trait-level drivers (tests/plain_hal_prover.c: plain Hal entry points only, gather_sample queue on) prove
segments 0, 1, 2, ... of 2^20 cycles side by side for SECONDS (one of the two drivers allocates and releases its big buffers inside every
proof: bx_alloc / bx_release pool).  Every seal is verified on the CPU; every segment a plain driver proved
is proved again by an in-library prover and the two seals compared word for word.  Then 20 x create / prove / destroy of a plain driver
(its own ctx each time): free HBM before == after."""
import json
import os
import queue
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import plain_hal  # noqa: E402
from boundless_amd.prover import HipProverServer, Segment, verify_seal  # noqa: E402

seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
out = sys.argv[2] if len(sys.argv) > 2 else None
servers = [HipProverServer(0) for _ in range(3)]  # two lanes + the re-prover
drivers = [plain_hal.PlainHalProver(0), plain_hal.PlainHalProver(0, flags=plain_hal.ALLOC_PER_PROOF)]  # the second allocates its ~9 GB of
# buffers inside every proof, through the library's per-ctx pool (as risc0-zkp's prover would)
todo, again = queue.Queue(), queue.Queue()
stats = {"proved_in_library": 0, "proved_plain_hal": 0, "verified": 0, "recompared": 0, "mismatches": 0, "verify_failures": 0}
lock = threading.Lock()
deadline = time.time() + seconds
counter = [0]


def claim():
    with lock:
        i = counter[0]
        counter[0] += 1
        return i


def lane(k):
    while time.time() < deadline:
        i = claim()
        seal = servers[k].prove_segment(Segment.synthetic(i)).seal
        with lock:
            stats["proved_in_library"] += 1
        todo.put(seal)


def plain(k):
    while time.time() < deadline:
        i = claim()
        seal, _ = drivers[k].prove(Segment.synthetic(i).seed)
        with lock:
            stats["proved_plain_hal"] += 1
        todo.put(seal)
        again.put((i, seal))


def recompare():
    while True:
        item = again.get()
        if item is None:
            return
        i, seal = item
        ref = servers[2].prove_segment(Segment.synthetic(i)).seal
        with lock:
            stats["recompared"] += 1
            if not np.array_equal(ref, seal):
                stats["mismatches"] += 1


def verifier():
    while True:
        seal = todo.get()
        if seal is None:
            return
        try:
            verify_seal(seal)
            with lock:
                stats["verified"] += 1
        except Exception:
            with lock:
                stats["verify_failures"] += 1


t0 = time.time()
vs = [threading.Thread(target=verifier) for _ in range(4)]
rc = threading.Thread(target=recompare)
ws = [threading.Thread(target=lane, args=(k,)) for k in range(2)] + [threading.Thread(target=plain, args=(k,)) for k in range(2)]
[t.start() for t in vs + [rc] + ws]
[t.join() for t in ws]
elapsed = time.time() - t0
again.put(None)
rc.join()
for _ in vs:
    todo.put(None)
[t.join() for t in vs]
for d in drivers:
    d.close()

import torch  # noqa: E402

torch.cuda.synchronize()
free0 = torch.cuda.mem_get_info(0)[0]
for k in range(20):
    d = plain_hal.PlainHalProver(0, po2=16, widths=(16, 64, 16), flags=(k % 2) * plain_hal.EXT_ALL)
    d.prove(k)
    d.close()
free1 = torch.cuda.mem_get_info(0)[0]
from boundless_amd.build import device_code_hash  # noqa: E402

res = {"device_code_sha": device_code_hash(), "seconds": round(elapsed, 1), **stats,
       "proofs_per_s": round((stats["proved_in_library"] + stats["proved_plain_hal"]) / elapsed, 2),
       "hbm_free_before_after_20_create_destroy": [free0, free1], "leak_bytes": free0 - free1}
print(json.dumps(res))
if out:
    json.dump(res, open(out, "w"), indent=1)
for s in servers:
    s.close()
assert stats["mismatches"] == 0 and stats["verify_failures"] == 0 and stats["verified"] == stats["proved_in_library"] + stats["proved_plain_hal"]
