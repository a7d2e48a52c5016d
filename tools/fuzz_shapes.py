#!/usr/bin/env python3
"""Seeded fuzz of the whole segment proof over random shapes, on the GPU: for every drawn (po2, widths, circuit knobs, seed, noise
seed) the seal of `bx_prove_segment` must equal the CPU oracle's word for word and verify, and (every other shape) so must the seal
of the trait-level driver (tests/plain_hal_prover.c) with a random subset of the extension entry points swapped in.

tests/test_prover_gpu.py holds 32 such shapes under one seed; this runs as many as the time allows under any seed:

    python tools/fuzz_shapes.py --iters 200 --seed 2 --max-po2 16        # prints one JSON line, exit code 1 on any failure

Test infrastructure (uses oracle/): not part of the product.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from oracle import oracle_lib as ol  # noqa: E402


def run(args):
    import plain_hal
    from boundless_amd.hal import HalError
    from boundless_amd.prover import HipProverServer, Segment

    ol.build()
    rng = np.random.default_rng(args.seed)
    failures, done, plain_done = [], 0, 0
    t0 = time.time()
    for it in range(args.iters):
        if args.seconds and time.time() - t0 > args.seconds:
            break
        po2 = int(rng.integers(args.min_po2, args.max_po2 + 1))
        cap = 1 if po2 >= 15 else 2  # keep the oracle's share of the time bounded
        widths = (int(rng.integers(1, 40 // cap)), int(rng.integers(1, 96 // cap)), int(rng.integers(1, 28 // cap)))
        knobs = (int(rng.integers(0, 65)), int(rng.integers(1, 6)))
        if knobs[0] == 0:
            knobs = (0, 0)
        seed = int(rng.integers(0, 2**63))
        noise = int(rng.integers(0, 2**63)) if rng.random() < 0.5 else None
        flags = int(rng.integers(0, 128)) if it % 2 == 0 else None  # plain driver: random extensions, random ALLOC_PER_PROOF
        what = {"po2": po2, "widths": widths, "knobs": knobs, "seed": seed, "noise_seed": noise, "plain_flags": flags}
        try:
            seal, roots = ol.prove_segment(po2, *widths, seed, terms=knobs[0], degree=knobs[1], noise_seed=noise)
            srv = HipProverServer(0, po2=po2, widths=widths, terms=knobs[0], degree=knobs[1])
            try:
                receipt = srv.prove_segment(Segment(index=int(rng.integers(0, 1000)), po2=po2, seed=seed, noise_seed=noise))
            finally:
                srv.close()
            if receipt.seal.size != seal.size or not np.array_equal(receipt.seal, seal):
                raise AssertionError("bx_prove_segment: seal differs from the oracle's")
            if not np.array_equal(receipt.roots, roots):
                raise AssertionError("bx_prove_segment: roots differ from the oracle's")
            receipt.verify_integrity()
            done += 1
            if flags is not None and noise is None:
                pp = plain_hal.PlainHalProver(0, po2=po2, widths=widths, terms=knobs[0], degree=knobs[1], flags=flags)
                try:
                    s2, _ = pp.prove(seed)
                    s3, _ = pp.prove(seed)  # the same object again: buffers reused (or re-allocated through the pool)
                finally:
                    pp.close()
                if not np.array_equal(s2, seal) or not np.array_equal(s3, seal):
                    raise AssertionError("plain-Hal driver: seal differs from the oracle's")
                plain_done += 1
            if args.verbose:
                print(it, what, flush=True)
        except (AssertionError, HalError, RuntimeError) as e:
            failures.append({"iter": it, **what, "error": f"{type(e).__name__}: {e}"[:400]})
    out = {"tool": "fuzz_shapes", "seed": args.seed, "shapes_proved_and_verified": done, "of_which_also_through_the_plain_driver": plain_done,
           "seconds": round(time.time() - t0, 1), "failures": failures[:20], "n_failures": len(failures)}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=100)
    ap.add_argument("--seed", type=int, default=2)
    ap.add_argument("--min-po2", type=int, default=9)
    ap.add_argument("--max-po2", type=int, default=15)
    ap.add_argument("--seconds", type=float, default=0, help="stop after this long (0 = run all iterations)")
    ap.add_argument("--verbose", action="store_true")
    out = run(ap.parse_args())
    print(json.dumps(out))
    return 1 if out["n_failures"] else 0


if __name__ == "__main__":
    sys.exit(main())
