#!/usr/bin/env python3
"""Seeded fuzz of the plain `Hal` entry points (include/bx_hal.h) against the CPU oracle, on the GPU.

What the hand-picked parametrisations of tests/test_hal_gpu.py do not reach: every operand is a SLICE of a larger device buffer that
starts at a random word offset (as `Buffer::slice` may hand out), the words in front of and behind every operand are canaries that
must come back untouched, and sizes / counts / strides are drawn at random (powers of two and not, where the entry point allows).
A call either returns the oracle's words exactly, or returns an error string (a shape the library refuses is refused loudly) —
silent differences and canary damage are the failures.

    python tools/fuzz_hal.py --iters 400 --seed 1            # on the GPU box; prints one JSON line, exit code 1 on any failure

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

from oracle import oracle_lib as ol  # noqa: E402

P = ol.P
PAD = 48  # canary words on each side


def c(a):
    return np.ascontiguousarray(a, dtype=np.uint32)


class Placed:
    """`data` inside a larger device buffer at word offset PAD + off, canaries around it."""

    def __init__(self, hal, rng, data, granule=1, aligned=False):
        data = c(data)
        self.n = data.size
        self.off = PAD + (0 if aligned else int(rng.integers(0, 8 // granule)) * granule)
        self.host = rng.integers(0, 1 << 32, self.n + self.off + PAD + 8, dtype=np.uint32)
        self.host[self.off:self.off + self.n] = data
        self.whole = hal.copy_from(self.host)
        self.buf = self.whole.slice(self.off, self.n)

    def check(self, want, what):
        got = self.whole.view()
        exp = self.host.copy()
        exp[self.off:self.off + self.n] = want
        if os.environ.get("FUZZ_SELFTEST") == "1" and self.n:  # the checker checked: one wrong word must be reported
            exp[self.off + self.n - 1] ^= 1
        if not np.array_equal(got[:self.off], exp[:self.off]) or not np.array_equal(got[self.off + self.n:], exp[self.off + self.n:]):
            raise AssertionError(f"{what}: words outside the operand were written (offset {self.off - PAD})")
        bad = np.nonzero(got != exp)[0]
        if bad.size:
            raise AssertionError(f"{what}: {bad.size} words differ from the oracle, first at {int(bad[0]) - self.off} (offset {self.off - PAD})")

    def free(self):
        self.whole.free()


def elems(rng, n):
    """field elements; one time in four, drawn from the extreme words"""
    if rng.random() < 0.25:
        pool = np.array([0, 1, 2, P - 1, P - 2, (P - 1) // 2, 268435454, 1172168163], dtype=np.uint32)
        return pool[rng.integers(0, len(pool), n)]
    return rng.integers(0, P, n, dtype=np.uint32)


# ---- one function per entry point: build operands, call, compare.  Each returns a short description of what it drew. ----
def big(rng, lo, mid, hi):
    """mostly [lo, mid), one time in six [mid, hi]: the large sizes take the multi-pass kernels"""
    return int(rng.integers(mid, hi + 1)) if rng.random() < 1 / 6 else int(rng.integers(lo, mid))


def op_interpolate(hal, O, rng):
    bits = big(rng, 1, 17, 22)
    count = int(rng.integers(1, 7 if bits < 17 else 3))
    n = 1 << bits
    x = elems(rng, n * count)
    io = Placed(hal, rng, x)
    hal.batch_interpolate_ntt(io.buf, count)
    ref = x.copy()
    O.bxo_batch_interpolate_ntt(ref, count, n)
    io.check(ref, f"batch_interpolate_ntt bits={bits} count={count}")
    hal.zk_shift(io.buf, count)
    O.bxo_zk_shift(ref, count, n)
    io.check(ref, f"zk_shift bits={bits} count={count}")
    hal.batch_bit_reverse(io.buf, count)
    O.bxo_batch_bit_reverse(ref, count, n)
    io.check(ref, f"batch_bit_reverse bits={bits} count={count}")
    io.free()
    return f"interpolate/zk_shift/bit_reverse 2^{bits} x {count}"


def op_expand(hal, O, rng):
    bits, eb = big(rng, 1, 15, 20), int(rng.integers(1, 4))
    count = int(rng.integers(1, 7 if bits < 15 else 3))
    n = 1 << bits
    x = elems(rng, n * count)
    inp = Placed(hal, rng, x)
    out = Placed(hal, rng, np.zeros(count * (n << eb), np.uint32))
    hal.batch_expand_into_evaluate_ntt(out.buf, inp.buf, count, eb)
    ref = np.zeros(count * (n << eb), np.uint32)
    O.bxo_batch_expand_into_evaluate_ntt(ref, x, count, n, eb)
    out.check(ref, f"batch_expand_into_evaluate_ntt bits={bits} count={count} expand={eb}")
    inp.check(x, "batch_expand_into_evaluate_ntt (input)")
    inp.free(), out.free()
    return f"expand 2^{bits} x {count} by 2^{eb}"


def op_evaluate(hal, O, rng):
    bits = big(rng, 1, 16, 22)
    count = int(rng.integers(1, 6 if bits < 16 else 3))
    eb = int(rng.integers(0, min(bits, 3) + 1))
    n = 1 << bits
    x = elems(rng, n * count)
    io = Placed(hal, rng, x)
    hal.batch_evaluate_ntt(io.buf, count, eb)
    ref = x.copy()
    O.bxo_batch_evaluate_ntt(ref, count, n, eb)
    io.check(ref, f"batch_evaluate_ntt bits={bits} count={count} expand={eb}")
    io.free()
    return f"evaluate 2^{bits} x {count} skip {eb}"


def op_hash_rows(hal, O, rng):
    rows = int(rng.integers(1, 3000)) if rng.random() < 0.5 else 1 << big(rng, 0, 13, 18)
    cols = int(rng.integers(1, 70 if rows <= 4096 else 20))
    x = elems(rng, rows * cols)
    m = Placed(hal, rng, x)
    out = Placed(hal, rng, np.zeros(8 * rows, np.uint32), granule=8)
    hal.hash_rows(out.buf, m.buf)
    ref = np.zeros(8 * rows, np.uint32)
    O.bxo_hash_rows(ref, x, rows, cols)
    out.check(ref, f"hash_rows rows={rows} cols={cols}")
    m.free(), out.free()
    return f"hash_rows {rows} x {cols}"


def op_hash_fold(hal, O, rng):
    out_size = 1 << big(rng, 0, 12, 19)
    nodes = elems(rng, 8 * 4 * out_size)  # digests [0, 4 out): inputs at [2 out, 4 out), outputs at [out, 2 out)
    io = Placed(hal, rng, nodes, granule=8)
    hal.hash_fold(io.buf, 2 * out_size, out_size)
    ref = nodes.copy()
    O.bxo_hash_fold(ref, 2 * out_size, out_size)
    io.check(ref, f"hash_fold out={out_size}")
    io.free()
    return f"hash_fold {2 * out_size} -> {out_size}"


def op_merkle(hal, O, rng):
    rows = 1 << big(rng, 0, 13, 19)
    cols = int(rng.integers(1, 40 if rows <= 4096 else 12))
    x = elems(rng, rows * cols)
    m = Placed(hal, rng, x)
    nodes = Placed(hal, rng, np.zeros(16 * rows, np.uint32), granule=8)
    hal.merkle_build(nodes.buf, m.buf, rows)
    ref = np.zeros(16 * rows, np.uint32)
    leaves = np.zeros(8 * rows, np.uint32)
    O.bxo_hash_rows(leaves, x, rows, cols)
    ref[8 * rows:] = leaves
    size = rows
    while size > 1:
        O.bxo_hash_fold(ref, size, size // 2)
        size //= 2
    got = nodes.whole.view()[nodes.off:nodes.off + 16 * rows]
    ref[:8] = got[:8]  # node 0 is unused
    nodes.check(ref, f"merkle_build rows={rows} cols={cols}")
    m.free(), nodes.free()
    return f"merkle_build {rows} x {cols}"


def op_fri_fold(hal, O, rng):
    count = int(rng.integers(1, 5000)) if rng.random() < 0.5 else 1 << int(rng.integers(0, 14))
    x = elems(rng, 64 * count)
    mix = elems(rng, 4)
    inp = Placed(hal, rng, x)
    out = Placed(hal, rng, np.zeros(4 * count, np.uint32))
    hal.fri_fold(out.buf, inp.buf, mix)
    ref = np.zeros(4 * count, np.uint32)
    O.bxo_fri_fold(ref, x, c(mix), count)
    out.check(ref, f"fri_fold count={count}")
    inp.free(), out.free()
    return f"fri_fold {count}"


def op_mix_poly(hal, O, rng):
    count = int(rng.integers(1, 20000)) if rng.random() < 0.5 else 1 << int(rng.integers(2, 15))
    npoly, ncombo = int(rng.integers(1, 40)), int(rng.integers(1, 6))
    inp_h = elems(rng, npoly * count)
    combos_h = rng.integers(0, ncombo, npoly, dtype=np.uint32)
    mix, start = elems(rng, 4), elems(rng, 4)
    init = elems(rng, ncombo * count * 4)
    out = Placed(hal, rng, init, granule=4)
    inp = Placed(hal, rng, inp_h)
    combos = Placed(hal, rng, combos_h)
    hal.mix_poly_coeffs(out.buf, start, mix, inp.buf, combos.buf, npoly, count)
    ref = init.copy()
    O.bxo_mix_poly_coeffs(ref, c(start), c(mix), inp_h, c(combos_h), npoly, count)
    out.check(ref, f"mix_poly_coeffs count={count} npoly={npoly} ncombo={ncombo}")
    out.free(), inp.free(), combos.free()
    return f"mix_poly_coeffs {count} x {npoly} -> {ncombo}"


def op_evaluate_any(hal, O, rng):
    size = 1 << int(rng.integers(0, 18))
    npoly, evals = int(rng.integers(1, 9)), int(rng.integers(1, 40))
    coeffs_h = elems(rng, npoly * size)
    which_h = rng.integers(0, npoly, evals, dtype=np.uint32)
    xs_h = elems(rng, 4 * evals)
    coeffs = Placed(hal, rng, coeffs_h)
    which = Placed(hal, rng, which_h)
    xs = Placed(hal, rng, xs_h, granule=4)
    out = Placed(hal, rng, np.zeros(4 * evals, np.uint32), granule=4)
    hal.batch_evaluate_any(coeffs.buf, npoly, which.buf, xs.buf, out.buf)
    ref = np.zeros(4 * evals, np.uint32)
    O.bxo_batch_evaluate_any(coeffs_h, size, c(which_h), xs_h, ref, evals)
    out.check(ref, f"batch_evaluate_any size={size} npoly={npoly} evals={evals}")
    for p in (coeffs, which, xs, out):
        p.free()
    return f"batch_evaluate_any {size} x {npoly}, {evals} points"


def op_eltwise(hal, O, rng):
    n = int(rng.integers(1, 200000))
    a_h, b_h = elems(rng, n), elems(rng, n)
    a, b = Placed(hal, rng, a_h), Placed(hal, rng, b_h)
    out = Placed(hal, rng, np.zeros(n, np.uint32))
    hal.eltwise_add_elem(out.buf, a.buf, b.buf)
    ref = np.zeros(n, np.uint32)
    O.bxo_eltwise_add(ref, a_h, b_h, n)
    out.check(ref, f"eltwise_add_elem n={n}")
    hal.eltwise_copy_elem(out.buf, a.buf)
    out.check(a_h, f"eltwise_copy_elem n={n}")
    z = a_h.copy()
    z[rng.integers(0, n, max(1, n // 5))] = 0xFFFFFFFF
    zb = Placed(hal, rng, z)
    hal.eltwise_zeroize_elem(zb.buf)
    O.bxo_eltwise_zeroize(z, n)
    zb.check(z, f"eltwise_zeroize_elem n={n}")
    for p in (a, b, out, zb):
        p.free()
    return f"eltwise add/copy/zeroize {n}"


def op_sum_ext(hal, O, rng):
    count, to_add = int(rng.integers(1, 30000)), int(rng.integers(1, 12))
    e_h = elems(rng, 4 * count * to_add)
    e = Placed(hal, rng, e_h, granule=4)
    out = Placed(hal, rng, np.zeros(4 * count, np.uint32))
    hal.eltwise_sum_extelem(out.buf, e.buf)
    ref = np.zeros(4 * count, np.uint32)
    O.bxo_eltwise_sum_extelem(ref, e_h, count, to_add)
    out.check(ref, f"eltwise_sum_extelem count={count} to_add={to_add}")
    e.free(), out.free()
    return f"eltwise_sum_extelem {count} x {to_add}"


def op_gather(hal, O, rng):
    n = int(rng.integers(64, 200000))
    src_h = elems(rng, n)
    src = Placed(hal, rng, src_h)
    k = int(rng.integers(1, 12))
    outs = []
    for _ in range(k):  # several in a row: the library queues small gathers and launches them together
        size = int(rng.integers(1, min(300, n)))
        stride = int(rng.integers(1, max(2, n // size)))
        idx = int(rng.integers(0, n - (size - 1) * stride))
        dst = Placed(hal, rng, np.zeros(size, np.uint32))
        hal.gather_sample(dst.buf, src.buf, idx, size, stride)
        ref = np.zeros(size, np.uint32)
        O.bxo_gather_sample(ref, src_h, idx, size, stride)
        outs.append((dst, ref, (idx, size, stride)))
    for dst, ref, what in outs:
        dst.check(ref, f"gather_sample idx/size/stride={what}")
        dst.free()
    src.free()
    return f"gather_sample x {k} from {n}"


def op_poly_divide(hal, O, rng):
    size = int(rng.integers(1, 300000)) if rng.random() < 0.6 else 1 << int(rng.integers(0, 19))
    poly_h = elems(rng, 4 * size)
    z = elems(rng, 4)
    poly = Placed(hal, rng, poly_h, granule=4)
    rem = Placed(hal, rng, np.zeros(4, np.uint32), granule=4)
    hal.poly_divide(poly.buf, z, rem.buf)
    ref, ref_rem = poly_h.copy(), np.zeros(4, np.uint32)
    O.bxo_poly_divide(ref, size, c(z), ref_rem)
    poly.check(ref, f"poly_divide size={size}")
    rem.check(ref_rem, f"poly_divide size={size} (remainder)")
    poly.free(), rem.free()
    return f"poly_divide {size}"


def op_prefix(hal, O, rng):
    n = int(rng.integers(1, 300000)) if rng.random() < 0.6 else 1 << int(rng.integers(0, 19))
    x = elems(rng, 4 * n)
    io = Placed(hal, rng, x, granule=4)
    hal.prefix_products(io.buf)
    ref = x.copy()
    O.bxo_prefix_products(ref, n)
    io.check(ref, f"prefix_products n={n}")
    io.free()
    return f"prefix_products {n}"


def op_scatter(hal, O, rng):
    into_len, cycles = int(rng.integers(16, 100000)), int(rng.integers(1, 300))
    counts = rng.integers(0, 9, cycles)
    index_h = np.concatenate([[0], np.cumsum(counts)]).astype(np.uint32)
    total = int(index_h[-1])
    if total > into_len:
        return "scatter (skipped: more values than cells)"
    offsets_h = rng.permutation(into_len)[:total].astype(np.uint32)
    values_h = elems(rng, max(total, 1))[:total]
    init = elems(rng, into_len)
    dst = Placed(hal, rng, init)
    index, offsets, values = Placed(hal, rng, index_h), Placed(hal, rng, c(offsets_h) if total else np.zeros(1, np.uint32)), Placed(hal, rng, c(values_h) if total else np.zeros(1, np.uint32))
    hal.scatter(dst.buf, index.buf, offsets.buf, values.buf)
    ref = init.copy()
    O.bxo_scatter(ref, index_h, c(offsets_h) if total else np.zeros(1, np.uint32), c(values_h) if total else np.zeros(1, np.uint32), cycles)
    dst.check(ref, f"scatter into={into_len} cycles={cycles} total={total}")
    for p in (dst, index, offsets, values):
        p.free()
    return f"scatter {total} into {into_len}"


def op_copy_slice(hal, O, rng):
    rows, cols = int(rng.integers(1, 200)), int(rng.integers(1, 3000))
    fs = int(rng.integers(0 if rows == 1 else 1, cols + 50))
    is_ = int(rng.integers(cols, cols + 100))
    fo, io = int(rng.integers(0, 40)), int(rng.integers(0, 40))
    src = elems(rng, fo + (rows - 1) * fs + cols + int(rng.integers(0, 9)))
    init = elems(rng, io + (rows - 1) * is_ + cols + int(rng.integers(0, 9)))
    dst = Placed(hal, rng, init)
    hal.eltwise_copy_elem_slice(dst.buf, src, rows, cols, fo, fs, io, is_)
    want = init.copy()
    for r in range(rows):
        want[io + r * is_: io + r * is_ + cols] = src[fo + r * fs: fo + r * fs + cols]
    dst.check(want, f"eltwise_copy_elem_slice rows={rows} cols={cols} fs={fs} is={is_} fo={fo} io={io}")
    dst.free()
    return f"eltwise_copy_elem_slice {rows} x {cols}"


OPS = [op_copy_slice, op_interpolate, op_expand, op_evaluate, op_hash_rows, op_hash_fold, op_merkle, op_fri_fold, op_mix_poly, op_evaluate_any, op_eltwise,
       op_sum_ext, op_gather, op_poly_divide, op_prefix, op_scatter]


def run(iters=300, seed=1, only="", verbose=False):
    from boundless_amd.hal import HalError, HipHal

    ol.build()
    O = ol.lib()
    hal = HipHal(0)
    rng = np.random.default_rng(seed)
    ops = [f for f in OPS if not only or only in f.__name__]
    ran, refused, failures = {}, {}, []
    t0 = time.time()
    try:
        for it in range(iters):
            f = ops[int(rng.integers(0, len(ops)))]
            sub = np.random.default_rng(int(rng.integers(0, 1 << 62)))
            try:
                what = f(hal, O, sub)
                ran[f.__name__] = ran.get(f.__name__, 0) + 1
                if verbose:
                    print(it, what, flush=True)
            except HalError as e:  # a refusal is loud: allowed, but counted and shown
                refused.setdefault(f.__name__, []).append(str(e))
                try:
                    hal.sync()
                except HalError:
                    pass
            except AssertionError as e:
                failures.append({"iter": it, "op": f.__name__, "error": str(e)})
            except Exception as e:  # a bug of this tool (or of the Python mirror): reported, and it fails the run
                failures.append({"iter": it, "op": f.__name__, "error": f"{type(e).__name__}: {e}"})
    finally:
        hal.close()
    return {"tool": "fuzz_hal", "seed": seed, "iters": iters, "seconds": round(time.time() - t0, 1), "ran": ran,
            "refused": {k: {"count": len(v), "first": v[0]} for k, v in refused.items()}, "failures": failures[:20], "n_failures": len(failures)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=300)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--only", default="")
    ap.add_argument("--verbose", action="store_true")
    args = ap.parse_args()
    out = run(args.iters, args.seed, args.only, args.verbose)
    print(json.dumps(out))
    return 1 if out["n_failures"] else 0


if __name__ == "__main__":
    sys.exit(main())
