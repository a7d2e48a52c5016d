"""The LDE alone (Hal::batch_expand_into_evaluate_ntt, N = 2^po2 -> 4N) through the C ABI: ms per call from the library's HIP-event
profiler, algorithmic GB/s (20*W*N bytes) and the fraction of the 8 TB/s roofline; --check compares one call with the oracle.

    python tools/ldebench.py [--po2 20] [--cols 64] [--reps 10] [--tunables a=1,b=2] [--check] [--interp]

Used under rocprofv3 for the per-kernel stall counters of the two LDE kernels (tools/r05_lde_probe.sh).
"""
import argparse
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from boundless_amd.hal import HipHal  # noqa: E402

P = 2013265921


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--po2", type=int, default=20)
    ap.add_argument("--cols", type=int, default=64)
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--tunables", default="")
    ap.add_argument("--check", action="store_true")
    ap.add_argument("--interp", action="store_true", help="also time batch_interpolate_ntt (+ fused zk_shift) on the same columns")
    ap.add_argument("--tag", default="")
    ap.add_argument("--warm-seconds", type=float, default=1.5, help="keep the GPU busy with the same call this long before timing (clocks ramp up)")
    a = ap.parse_args()
    hal = HipHal(0)
    for item in filter(None, a.tunables.split(",")):
        k, v = item.split("=")
        hal.set_tunable(k, int(v))
    n, cols = 1 << a.po2, a.cols
    rng = np.random.default_rng(7)
    x = rng.integers(0, P, n * cols, dtype=np.uint32)
    src = hal.copy_from(x)
    out = hal.alloc(4 * n * cols)
    io = hal.alloc(n * cols) if a.interp else None
    hal.batch_expand_into_evaluate_ntt(out, src, cols, 2)
    hal.sync()
    if a.check:
        from oracle import oracle_lib as ol

        cc = min(cols, 4)
        ref = np.zeros(4 * n * cc, np.uint32)
        ol.lib().bxo_batch_expand_into_evaluate_ntt(ref, x[: n * cc].copy(), cc, n, 2)
        got = out.view()[: 4 * n * cc]
        assert np.array_equal(got, ref), "LDE differs from the oracle"
    import time

    t_end = time.time() + a.warm_seconds
    while time.time() < t_end:
        for _ in range(8):
            hal.batch_expand_into_evaluate_ntt(out, src, cols, 2)
        hal.sync()
    hal.profile_reset()
    hal.profile_enable(True)
    for _ in range(a.reps):
        hal.batch_expand_into_evaluate_ntt(out, src, cols, 2)
        if a.interp:
            hal.eltwise_copy_elem(io, src)
            hal.batch_interpolate_ntt(io, cols)
    hal.sync()
    rep = hal.profile_report()
    hal.profile_enable(False)
    for name, r in sorted(rep.items()):
        if "ntt" not in name:
            continue
        ms = r["ms"] / r["calls"]
        gbs = r["alg_bytes"] / r["calls"] / (ms * 1e-3) / 1e9 if ms > 0 else 0
        print(json.dumps({"op": name, "tag": a.tag, "tunables": a.tunables, "po2": a.po2, "cols": cols, "ms": round(ms, 4), "alg_GBps": round(gbs, 1),
                          "frac_8TBps": round(gbs / 8000, 4), "checked": bool(a.check)}))
    hal.close()


if __name__ == "__main__":
    main()
