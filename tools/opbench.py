"""Per-entry-point timing at BASELINE sizes (N = 2^po2 rows) through the C ABI, with the library's HIP-event profiler.

    python tools/opbench.py [--po2 20] [--cols 64] [--reps 5] [--sweep]

Prints one JSON line per op: ms per call, algorithmic GB/s and the fraction of the 8 TB/s HBM roofline.
"""
import argparse
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from boundless_amd.hal import HipHal  # noqa: E402

P = 2013265921


def run(hal, po2, cols, reps, tag=""):
    n = 1 << po2
    rng = np.random.default_rng(1)
    x = rng.integers(0, P, n * cols, dtype=np.uint32)
    src = hal.copy_from(x)
    io = hal.alloc(n * cols)
    out = hal.alloc(4 * n * cols)
    nodes = hal.alloc_digest(2 * 4 * n)
    ext = hal.copy_from(rng.integers(0, P, 4 * n, dtype=np.uint32))
    fold_out = hal.alloc(4 * n // 16)
    mix = rng.integers(0, P, 4, dtype=np.uint32)
    # warm-up (table builds)
    hal.eltwise_copy_elem(io, src)
    hal.batch_interpolate_ntt(io, cols)
    hal.zk_shift(io, cols)
    hal.batch_expand_into_evaluate_ntt(out, io, cols, 2)
    hal.sync()
    hal.profile_reset()
    hal.profile_enable(True)
    for _ in range(reps):
        hal.eltwise_copy_elem(io, src)
        hal.batch_interpolate_ntt(io, cols)
        hal.zk_shift(io, cols)
        hal.batch_expand_into_evaluate_ntt(out, io, cols, 2)
        hal.batch_bit_reverse(io, cols)
        hal.merkle_build(nodes, out, 4 * n)
        hal.fri_fold(fold_out, ext, mix)
    hal.sync()
    rep = hal.profile_report()
    hal.profile_enable(False)
    for name, r in sorted(rep.items()):
        ms = r["ms"] / r["calls"]
        gbs = r["alg_bytes"] / r["calls"] / (ms * 1e-3) / 1e9 if ms > 0 else 0
        print(json.dumps({"op": name, "tag": tag, "po2": po2, "cols": cols, "ms": round(ms, 4), "alg_GBps": round(gbs, 1),
                          "frac_8TBps": round(gbs / 8000, 4)}))
    for b in (src, io, out, nodes, ext, fold_out):
        b.free()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--po2", type=int, default=20)
    ap.add_argument("--cols", type=int, default=64)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--sweep", action="store_true")
    a = ap.parse_args()
    hal = HipHal(0)
    print(json.dumps({"device": hal.device_name()}))
    if a.sweep:
        for fast, blk, ta, tb in ((0, 12, 12, 13), (1, 12, 12, 13), (1, 12, 12, 14), (1, 12, 12, 12), (1, 12, 13, 13), (1, 13, 13, 13), (1, 13, 13, 12),
                                  (1, 11, 12, 13), (1, 11, 11, 13), (1, 10, 12, 13), (1, 12, 12, 11)):
            hal.set_tunable("ntt_fast", fast)
            hal.set_tunable("ntt_block_log", blk)
            hal.set_tunable("ntt_tile_a_log", ta)
            hal.set_tunable("ntt_tile_b_log", tb)
            run(hal, a.po2, a.cols, a.reps, tag=f"fast{fast}_blk{blk}_ta{ta}_tb{tb}")
    else:
        run(hal, a.po2, a.cols, a.reps)


if __name__ == "__main__":
    main()
