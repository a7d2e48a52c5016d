#!/usr/bin/env python3
"""How much of a segment download does a lane hide?  The HIP prover as a REST worker (include/bx_rest.h) of the local API stub,
2^20-cycle segments, every GET of a segment delayed by --get-ms (an ~80 MB download on a 10-25 Gb/s link), lanes x prefetch.
One JSON line per configuration.  Run on a GPU box from the repo root:  python tools/rest_prefetch_bench.py > gpurun_out/x.jsonl"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from rest_stub_server import StubServer  # noqa: E402

from boundless_amd import agent as ag  # noqa: E402
from boundless_amd import build  # noqa: E402
from boundless_amd.prover import Segment  # noqa: E402

JOB = "0b1e55ed-0000-4000-8000-00000000be7c"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--po2", type=int, default=20)
    ap.add_argument("--segments", type=int, default=36)
    ap.add_argument("--get-ms", type=float, default=40.0)
    ap.add_argument("--lanes", default="1,3")
    args = ap.parse_args()
    for lanes in [int(x) for x in args.lanes.split(",")]:
        for prefetch in (False, True):
            srv = StubServer()
            w = ag.RestWorker(srv.url, claim_wait_secs=0)
            a = ag.Agent(prover=None, device=0, inflight=lanes, poll_time=0.002, store=w.store, taskdb=w.taskdb, prefetch=prefetch, verify=True)
            try:
                a.prewarm(args.po2)
                # warm-up: one proof per lane, no delay
                for i in range(lanes):
                    srv.state.hot[f"job:{JOB}:segments:{1000 + i}"] = (ag.serialize_segment(Segment.synthetic(1000 + i, po2=args.po2)), None)
                    srv.state.create_task("prove", JOB, f"warm-{i}", {"Prove": {"index": 1000 + i}}, max_retries=0)
                assert a.poll_work(max_idle_polls=3) == lanes
                srv.state.get_delay = args.get_ms / 1e3
                for i in range(args.segments):
                    srv.state.hot[f"job:{JOB}:segments:{i}"] = (ag.serialize_segment(Segment.synthetic(i, po2=args.po2)), None)
                    srv.state.create_task("prove", JOB, f"prove-{i}", {"Prove": {"index": i}}, max_retries=0)
                t0 = time.monotonic()
                done = a.poll_work(max_idle_polls=3)
                wall = time.monotonic() - t0
                ok = all(t["state"] == "done" for t in srv.state.tasks)
                print(json.dumps({"what": "HIP prover as a REST worker of the API stub; every segment GET delayed", "po2": args.po2, "segments": args.segments,
                                  "get_ms": args.get_ms, "lanes": lanes, "prefetch": prefetch, "done": done, "all_done": ok,
                                  "seconds": round(wall, 3), "proofs_per_s": round(args.segments / wall, 2),
                                  "ms_per_proof_per_lane": round(wall / args.segments * lanes * 1e3, 1), "device_code_sha": build.device_code_hash(), "csrc_sha": build.csrc_hash()}), flush=True)
            finally:
                a.close()
                w.close()
                srv.close()


if __name__ == "__main__":
    main()
