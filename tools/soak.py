"""Soak run on one MI355X (not part of the test suite; `python tools/soak.py [out.json]` on the GPU box).

1. Determinism under concurrency: three provers on their own threads prove the same 300 segments (po2 14) in different
   orders, twice; every seal must verify and every pair of seals of the same segment must be identical.
2. The metric's size: 45 segments at po2 20 (16/256/64), three in flight, every seal verified on the CPU.
3. Device memory: create/prove/destroy a prover (context included) 40 times at po2 16; free HBM before == after.
"""
import json
import sys
import threading
import time

import numpy as np
import torch

from boundless_amd.prover import HipProverServer, Segment, verify_seal


def free_bytes():
    torch.cuda.synchronize()
    return torch.cuda.mem_get_info(0)[0]


def concurrent(po2, widths, n_segments, lanes, passes):
    seals = [dict() for _ in range(passes * lanes)]
    servers = [HipProverServer(0, po2=po2, widths=widths) for _ in range(lanes)]

    def run(slot, srv, order):
        for i in order:
            seals[slot][i] = srv.prove_segment(Segment.synthetic(i, po2=po2)).seal

    t0 = time.time()
    for p in range(passes):
        threads = []
        for l in range(lanes):
            order = np.random.default_rng(100 * p + l).permutation(n_segments).tolist()
            threads.append(threading.Thread(target=run, args=(p * lanes + l, servers[l], order)))
        for t in threads:
            t.start()
        for t in threads:
            t.join()
    elapsed = time.time() - t0
    for s in servers:
        s.close()
    mismatches = 0
    for i in range(n_segments):
        for k in range(1, len(seals)):
            mismatches += int(not np.array_equal(seals[0][i], seals[k][i]))
    return seals[0], mismatches, elapsed


def main(out):
    res = {}
    first, mism, el = concurrent(14, (4, 24, 8), 300, 3, 2)
    for i, s in first.items():
        verify_seal(s)
    res["determinism_po2_14"] = {"segments": 300, "copies_of_each": 6, "mismatching_copies": mism, "all_verified": True,
                                 "proofs": 1800, "seconds": round(el, 2)}
    assert mism == 0

    servers = [HipProverServer(0) for _ in range(3)]
    seals = [None] * 45

    def run20(l):
        for i in range(l, 45, 3):
            seals[i] = servers[l].prove_segment(Segment.synthetic(i)).seal

    t0 = time.time()
    th = [threading.Thread(target=run20, args=(l,)) for l in range(3)]
    [t.start() for t in th]
    [t.join() for t in th]
    el = time.time() - t0
    for s in seals:
        verify_seal(s)
    assert len({s.tobytes() for s in seals}) == 45
    for s in servers:
        s.close()
    res["baseline_config_po2_20"] = {"segments": 45, "in_flight": 3, "all_verified": True, "all_distinct": True,
                                     "segment_proofs_per_s": round(45 / el, 2)}

    HipProverServer(0, po2=16, widths=(4, 24, 8)).close()  # warm allocator pools and the module
    before = free_bytes()
    for k in range(40):
        srv = HipProverServer(0, po2=16, widths=(4, 24, 8))
        srv.prove_segment(Segment.synthetic(k, po2=16))
        srv.close()
    after = free_bytes()
    res["create_destroy_x40_po2_16"] = {"free_bytes_before": before, "free_bytes_after": after, "leaked_bytes": before - after}
    assert before - after < (8 << 20), res
    print(json.dumps(res))
    if out:
        json.dump(res, open(out, "w"), indent=1)


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else None)
