"""CPU: pin the C oracle against the golden fixtures (published constants/KAT + definition-level vectors)."""
import json
import os

import numpy as np
import pytest

from oracle import np_oracle as npo
from oracle import oracle_lib as ol

P = ol.P


def load(golden_dir, name):
    with open(os.path.join(golden_dir, name)) as f:
        return json.load(f)


def enc(x):
    return np.ascontiguousarray(ol.encode(np.asarray(x, dtype=np.uint64)))


def dec(x):
    return ol.decode(x).tolist()


def bitrev(i, bits):
    return npo.bitrev(i, bits)


def test_field_constants(oracle, golden_dir):
    g = load(golden_dir, "babybear_consts.json")
    assert g["P"] == P == 15 * 2**27 + 1
    assert (g["M"] * P) % 2**32 == 1
    for k in range(28):
        assert oracle.bxo_fp_decode(oracle.bxo_rou_fwd(k)) == g["rou_fwd"][k]
        assert oracle.bxo_fp_decode(oracle.bxo_rou_rev(k)) == g["rou_rev"][k]
    assert oracle.bxo_fp_encode(1) == g["R"]
    assert oracle.bxo_fp_mul(g["R2"], 1) == g["R"]


def test_field_ops_random(oracle):
    rng = np.random.default_rng(7)
    a = rng.integers(0, P, 2000, dtype=np.uint64)
    b = rng.integers(0, P, 2000, dtype=np.uint64)
    am, bm = ol.encode(a), ol.encode(b)
    for x, y, xm, ym in zip(a.tolist(), b.tolist(), am.tolist(), bm.tolist()):
        assert oracle.bxo_fp_decode(oracle.bxo_fp_mul(xm, ym)) == (x * y) % P
        assert oracle.bxo_fp_decode(oracle.bxo_fp_add(xm, ym)) == (x + y) % P
        assert oracle.bxo_fp_decode(oracle.bxo_fp_sub(xm, ym)) == (x - y) % P
    # edge values
    for x in (0, 1, P - 1):
        for y in (0, 1, P - 1):
            xm, ym = int(ol.encode([x])[0]), int(ol.encode([y])[0])
            assert oracle.bxo_fp_decode(oracle.bxo_fp_mul(xm, ym)) == (x * y) % P
            assert oracle.bxo_fp_decode(oracle.bxo_fp_sub(xm, ym)) == (x - y) % P
    assert oracle.bxo_fp_decode(oracle.bxo_fp_inv(int(ol.encode([31])[0]))) == pow(31, -1, P)


def test_fp4(oracle, golden_dir):
    g = load(golden_dir, "fri_vectors.json")["fp4"]
    out = np.zeros(4, np.uint32)
    oracle.bxo_fp4_mul(out, enc(g["x"]), enc(g["y"]))
    assert dec(out) == g["xy"]
    oracle.bxo_fp4_inv(out, enc(g["x"]))
    assert dec(out) == g["x_inv"]


def test_poseidon2_published_kat(oracle, golden_dir):
    g = load(golden_dir, "poseidon2_kat.json")
    rc = np.zeros(213, np.uint32)
    diag = np.zeros(24, np.uint32)
    oracle.bxo_poseidon2_get_params(rc, diag)
    assert rc.tolist() == g["round_constants"] and diag.tolist() == g["internal_diag"]
    assert [hex(x) for x in rc[:4]] == ["0xfa20c37", "0x795bb97", "0x12c60b9c", "0xeabd88e"]
    st = enc(g["kat_in"])
    oracle.bxo_poseidon2_mix(st)
    assert dec(st) == g["kat_out"]
    for case in g["seeded"]:
        st = enc(case["in"])
        oracle.bxo_poseidon2_mix(st)
        assert dec(st) == case["out"]


def test_poseidon2_sponge_pair_merkle(oracle, golden_dir):
    g = load(golden_dir, "poseidon2_sponge.json")
    d = np.zeros(8, np.uint32)
    for case in g["sponge"]:
        el = enc(case["in"]) if case["in"] else np.zeros(1, np.uint32)
        oracle.bxo_hash_elem_slice(d, el, len(case["in"]), 1)
        assert dec(d) == case["digest"]
    oracle.bxo_hash_pair(d, enc(g["pair"]["a"]), enc(g["pair"]["b"]))
    assert dec(d) == g["pair"]["out"]
    m = g["merkle"]
    rows, cols = m["rows"], m["cols"]
    mat = enc(np.array(m["matrix_colmajor"], dtype=np.uint64).reshape(-1))
    nodes = np.zeros(2 * rows * 8, np.uint32)
    leaves = np.zeros(rows * 8, np.uint32)
    oracle.bxo_hash_rows(leaves, mat, rows, cols)
    assert dec(leaves.reshape(rows, 8)) == m["leaves"]
    nodes[rows * 8 :] = leaves
    size = rows
    while size > 1:
        oracle.bxo_hash_fold(nodes, size, size // 2)
        size //= 2
    assert dec(nodes[8:16]) == m["root"]


def test_ntt_vectors(oracle, golden_dir):
    for case in load(golden_dir, "ntt_vectors.json")["cases"]:
        n = 1 << case["bits"]
        io = enc(case["evals"])
        oracle.bxo_batch_interpolate_ntt(io, 1, n)
        assert dec(io) == case["coeffs_bitrev"]
        oracle.bxo_zk_shift(io, 1, n)
        assert dec(io) == case["shifted_bitrev"]
        out = np.zeros(4 * n, np.uint32)
        oracle.bxo_batch_expand_into_evaluate_ntt(out, io, 1, n, 2)
        assert dec(out) == case["lde4"]
        # round trip + bit reverse
        back = enc(case["evals"])
        oracle.bxo_batch_interpolate_ntt(back, 1, n)
        oracle.bxo_batch_evaluate_ntt(back, 1, n, 0)
        assert dec(back) == case["evals"]
        br = enc(case["coeffs_bitrev"])
        oracle.bxo_batch_bit_reverse(br, 1, n)
        assert dec(br) == [case["coeffs_bitrev"][bitrev(i, case["bits"])] for i in range(n)]


def test_lde_is_coset_evaluation(oracle):
    """interpolate -> zk_shift -> expand+evaluate == evaluations of the same polynomial on 3*<w_4N> (Horner)."""
    rng = np.random.default_rng(11)
    bits = 5
    n = 1 << bits
    ev = rng.integers(0, P, n, dtype=np.uint64).tolist()
    io = enc(ev)
    oracle.bxo_batch_interpolate_ntt(io, 1, n)
    co_br = dec(io)
    coeffs = [co_br[bitrev(j, bits)] for j in range(n)]
    oracle.bxo_zk_shift(io, 1, n)
    out = np.zeros(4 * n, np.uint32)
    oracle.bxo_batch_expand_into_evaluate_ntt(out, io, 1, n, 2)
    w = npo.rou(bits + 2)
    want = [npo.poly_eval(coeffs, 3 * pow(w, k, P) % P) for k in range(4 * n)]
    assert dec(out) == want


def test_fri_fold_identity(oracle, golden_dir):
    g = load(golden_dir, "fri_vectors.json")["fold"]
    f = g["coeffs_natural"]
    total = len(f)
    count = total // 16
    bits = npo.log2(total)
    # HAL layout: SoA ext planes of bit-reversed coefficients
    planes = np.zeros(4 * total, np.uint64)
    for j in range(total):
        for k in range(4):
            planes[k * total + bitrev(j, bits)] = f[j][k]
    out = np.zeros(4 * count, np.uint32)
    oracle.bxo_fri_fold(out, enc(planes), enc(g["mix"]), count)
    o = dec(out)
    got = [[o[k * count + bitrev(q, bits - 4)] for k in range(4)] for q in range(count)]
    assert got == g["out_natural"]


def test_mix_evaluate_sum_divide(oracle):
    rng = np.random.default_rng(5)
    count, npoly, ncombo = 32, 5, 2
    inp = rng.integers(0, P, npoly * count, dtype=np.uint64)
    combos = np.array([0, 1, 0, 1, 1], np.uint32)
    mix = rng.integers(0, P, 4, dtype=np.uint64).tolist()
    start = rng.integers(0, P, 4, dtype=np.uint64).tolist()
    out = np.zeros(ncombo * count * 4, np.uint32)
    oracle.bxo_mix_poly_coeffs(out, enc(start), enc(mix), enc(inp), combos, npoly, count)
    want = [[[0, 0, 0, 0] for _ in range(count)] for _ in range(ncombo)]
    cur = start
    for i in range(npoly):
        for idx in range(count):
            want[combos[i]][idx] = npo.f4_add(want[combos[i]][idx], [c * int(inp[i * count + idx]) % P for c in cur])
        cur = npo.f4_mul(cur, mix)
    assert dec(out.reshape(ncombo, count, 4)) == want
    # batch_evaluate_any vs Horner
    which = np.array([3, 0, 3], np.uint32)
    xs = rng.integers(0, P, (3, 4), dtype=np.uint64).tolist()
    ev = np.zeros(12, np.uint32)
    oracle.bxo_batch_evaluate_any(enc(inp), count, which, enc(np.array(xs).reshape(-1)), ev, 3)
    for e in range(3):
        co = [[int(inp[which[e] * count + i]), 0, 0, 0] for i in range(count)]
        assert dec(ev[4 * e : 4 * e + 4]) == npo.f4_poly_eval(co, xs[e])
    # eltwise_sum_extelem
    summed = np.zeros(4 * count, np.uint32)
    oracle.bxo_eltwise_sum_extelem(summed, out, count, ncombo)
    s = dec(summed)
    for idx in range(count):
        assert [s[k * count + idx] for k in range(4)] == npo.f4_add(want[0][idx], want[1][idx])
    # poly_divide: (f - f(z)) / (x - z) has zero remainder and q*(x-z)+f(z) == f
    z = xs[0]
    poly = [want[0][i] for i in range(count)]
    fz = npo.f4_poly_eval(poly, z)
    poly[0] = [(a - b) % P for a, b in zip(poly[0], fz)]
    buf = enc(np.array(poly, dtype=np.uint64).reshape(-1))
    rem = np.zeros(4, np.uint32)
    assert oracle.bxo_poly_divide(buf, count, enc(z), rem) == 1
    q = dec(buf.reshape(count, 4))
    y = xs[1]
    lhs = npo.f4_add(npo.f4_mul(npo.f4_poly_eval(q, y), [(a - b) % P for a, b in zip(y, z)]), fz)
    assert lhs == npo.f4_poly_eval(want[0], y)


def test_eltwise_gather_zeroize(oracle):
    a = np.array([1, P - 1, 5, 0xFFFFFFFF], np.uint32)
    io = a.copy()
    oracle.bxo_eltwise_zeroize(io, 4)
    assert io.tolist() == [1, P - 1, 5, 0]
    src = np.arange(40, dtype=np.uint32)
    dst = np.zeros(5, np.uint32)
    oracle.bxo_gather_sample(dst, src, 3, 5, 8)
    assert dst.tolist() == [3, 11, 19, 27, 35]


def test_prefix_products_and_scatter(oracle):
    rng = np.random.default_rng(21)
    n = 37
    x = rng.integers(0, P, (n, 4), dtype=np.uint64).tolist()
    buf = enc(np.array(x, dtype=np.uint64).reshape(-1))
    oracle.bxo_prefix_products(buf, n)
    want = [x[0]]
    for i in range(1, n):
        want.append(npo.f4_mul(x[i], want[-1]))
    assert dec(buf.reshape(n, 4)) == want
    into = np.zeros(10, np.uint32)
    oracle.bxo_scatter(into, np.array([0, 2, 2, 5], np.uint32), np.array([7, 1, 3, 9, 0], np.uint32),
                       np.array([11, 22, 33, 44, 55], np.uint32), 3)
    assert into.tolist() == [55, 22, 0, 33, 0, 0, 0, 11, 0, 44]
