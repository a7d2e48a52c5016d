"""Isolated timing of the HBM-bound helper entry points at the sizes one 2^20 proof calls them with.

    python tools/helperbench.py [--po2 20] [--reps 5] [--tunables name=value,...]

One JSON line per op: ms per call, algorithmic GB/s (the bytes DESIGN.md section 4 states for the call) and the fraction of
the 8 TB/s HBM roofline; HIP events on the ctx's stream (bx_profile_*), the ops launched back to back as a proof does.
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
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--tunables", type=str, default="")
    a = ap.parse_args()
    hal = HipHal(0)
    for item in filter(None, a.tunables.split(",")):
        k, _, v = item.partition("=")
        hal.set_tunable(k, int(v))
    n = 1 << a.po2
    rng = np.random.default_rng(3)
    cols, evals = 64, 96  # a 64-column slice of a group with 1.5 taps per column
    coeffs = hal.copy_from(rng.integers(0, P, n * cols, dtype=np.uint32))
    which = hal.copy_from((np.arange(evals) * cols // evals).astype(np.uint32))
    xs = hal.copy_from(rng.integers(0, P, 4 * evals, dtype=np.uint32))
    ev_out = hal.alloc(4 * evals)
    poly = hal.copy_from(rng.integers(0, P, 4 * n, dtype=np.uint32))
    rem = hal.alloc(4)
    z = rng.integers(0, P, 4, dtype=np.uint32)
    accs = 16
    run = hal.copy_from(rng.integers(1, P, 4 * n * accs, dtype=np.uint32))
    br = hal.copy_from(rng.integers(0, P, n * 16, dtype=np.uint32))
    brx = hal.copy_from(rng.integers(0, P, 4 * n * 3, dtype=np.uint32))
    fin = hal.copy_from(rng.integers(0, P, 4 * n, dtype=np.uint32))
    fout = hal.alloc(4 * n // 16)

    def once():
        hal.batch_evaluate_any_bitrev(coeffs, cols, which, xs, ev_out)
        hal.batch_evaluate_any(coeffs, cols, which, xs, ev_out)
        hal.poly_divide(poly, z, rem)
        hal.batch_prefix_products(run, accs)
        hal.batch_bit_reverse(br, 16)
        hal.batch_bit_reverse_ext(brx, 3)
        hal.fri_fold(fout, fin, z)

    once()
    hal.sync()
    hal.profile_reset()
    hal.profile_enable(True)
    for _ in range(a.reps):
        once()
    hal.sync()
    rep = hal.profile_report()
    hal.profile_enable(False)
    print(json.dumps({"device": hal.device_name(), "po2": a.po2, "tunables": a.tunables}))
    for name, r in sorted(rep.items()):
        ms = r["ms"] / r["calls"]
        gbs = r["alg_bytes"] / r["calls"] / (ms * 1e-3) / 1e9 if ms > 0 else 0
        print(json.dumps({"op": name, "calls": r["calls"], "ms": round(ms, 4), "alg_GBps": round(gbs, 1), "frac_8TBps": round(gbs / 8000, 4)}))


if __name__ == "__main__":
    main()
