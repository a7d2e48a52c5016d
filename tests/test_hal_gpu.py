"""GPU parity: every HAL entry point (through the C ABI) against the CPU oracle, bit-exact."""
import json
import os

import numpy as np
import pytest

from oracle import np_oracle as npo
from oracle import oracle_lib as ol

pytestmark = pytest.mark.gpu
P = ol.P


@pytest.fixture(scope="module")
def hal():
    from boundless_amd.hal import HipHal

    h = HipHal(0)
    yield h
    h.close()


oracle_P = 2013265921


def rnd(seed, n):
    return ol.random_elems(np.random.default_rng(seed), n)


def c(a):
    return np.ascontiguousarray(a, dtype=np.uint32)


def adversarial_columns(n):
    """Extreme columns of n rows for the NTT kernels (the default kernels carry values in [0, 2P) across their LDS regroupings, so
    the words that sit at the ends of every intermediate range must go through the COMPILED kernels, not only through the host
    check of the arithmetic source): all 0, all P - 1, an impulse at row 0, an impulse at row n - 1, alternating 0 / P - 1."""
    z = np.zeros(n, np.uint32)
    top = np.full(n, oracle_P - 1, np.uint32)
    first, last, alt = z.copy(), z.copy(), z.copy()
    first[0] = oracle_P - 1
    last[n - 1] = oracle_P - 1
    alt[1::2] = oracle_P - 1
    return np.concatenate([z, top, first, last, alt])


N_ADV = 5


# ------------------------------------------------------------------ NTT family
@pytest.fixture(params=[1, 0], ids=["r16", "v1"])
def ntt_path(hal, request):
    """Run the NTT tests on both kernel families: register-radix-16 (default) and the one-stage-per-barrier v1."""
    hal.set_tunable("ntt_fast", request.param)
    yield request.param
    hal.set_tunable("ntt_fast", 1)


@pytest.mark.parametrize("bits", [1, 2, 3, 5, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18])
@pytest.mark.parametrize("count", [1, 3, "3+adv"])
def test_interpolate_zkshift_lde_bitreverse(hal, oracle, ntt_path, bits, count):
    n = 1 << bits
    if count == "3+adv":  # three random columns, then the five extreme ones
        count = 3 + N_ADV
        x = np.concatenate([rnd(100 + bits, n * 3), adversarial_columns(n)])
    else:
        x = rnd(100 + bits, n * count)
    ref = x.copy()
    io = hal.copy_from(x)
    hal.batch_interpolate_ntt(io, count)
    oracle.bxo_batch_interpolate_ntt(ref, count, n)
    assert np.array_equal(io.view(), ref), "batch_interpolate_ntt"
    hal.zk_shift(io, count)
    oracle.bxo_zk_shift(ref, count, n)
    assert np.array_equal(io.view(), ref), "zk_shift"
    out = hal.alloc(4 * n * count)
    hal.batch_expand_into_evaluate_ntt(out, io, count, 2)
    ref_out = np.zeros(4 * n * count, np.uint32)
    oracle.bxo_batch_expand_into_evaluate_ntt(ref_out, ref, count, n, 2)
    assert np.array_equal(out.view(), ref_out), "batch_expand_into_evaluate_ntt"
    hal.batch_bit_reverse(io, count)
    oracle.bxo_batch_bit_reverse(ref, count, n)
    assert np.array_equal(io.view(), ref), "batch_bit_reverse"
    if count > 3:  # the fused interpolate + zk_shift entry point and the in-place evaluate on the same extreme columns
        io2 = hal.copy_from(x)
        hal.batch_interpolate_zk(io2, count)
        ref2 = x.copy()
        oracle.bxo_batch_interpolate_ntt(ref2, count, n)
        oracle.bxo_zk_shift(ref2, count, n)
        assert np.array_equal(io2.view(), ref2), "batch_interpolate_zk"
        io3 = hal.copy_from(x)
        hal.batch_evaluate_ntt(io3, count, 0)
        ref3 = x.copy()
        oracle.bxo_batch_evaluate_ntt(ref3, count, n, 0)
        assert np.array_equal(io3.view(), ref3), "batch_evaluate_ntt"


@pytest.mark.parametrize("bits,expand", [(4, 0), (9, 0), (11, 0), (14, 0), (6, 2), (12, 2), (14, 2), (15, 1)])
def test_evaluate_ntt_in_place(hal, oracle, ntt_path, bits, expand):
    n = 1 << bits
    x = rnd(7 + bits, n * 2)
    ref = x.copy()
    io = hal.copy_from(x)
    hal.batch_evaluate_ntt(io, 2, expand)
    oracle.bxo_batch_evaluate_ntt(ref, 2, n, expand)
    assert np.array_equal(io.view(), ref)


def test_ntt_golden_vectors(hal, golden_dir):
    cases = json.load(open(os.path.join(golden_dir, "ntt_vectors.json")))["cases"]
    for case in cases:
        n = 1 << case["bits"]
        io = hal.copy_from(ol.encode(case["evals"]))
        hal.batch_interpolate_ntt(io, 1)
        assert ol.decode(io.view()).tolist() == case["coeffs_bitrev"]
        hal.zk_shift(io, 1)
        assert ol.decode(io.view()).tolist() == case["shifted_bitrev"]
        out = hal.alloc(4 * n)
        hal.batch_expand_into_evaluate_ntt(out, io, 1, 2)
        assert ol.decode(out.view()).tolist() == case["lde4"]


@pytest.mark.parametrize("fast,block_log,tile_log,tile_a,tile_b", [
    (0, 12, 14, 12, 13), (0, 13, 14, 12, 13), (0, 11, 13, 12, 13), (0, 10, 12, 12, 13),
    (1, 12, 14, 12, 13), (1, 12, 14, 12, 14), (1, 13, 14, 13, 13), (1, 11, 14, 12, 12), (1, 10, 14, 11, 13), (1, 12, 14, 13, 11)])
def test_ntt_full_size_vs_oracle_and_roundtrip(hal, oracle, fast, block_log, tile_log, tile_a, tile_b):
    """BASELINE size: N = 2^20 rows -> 2^22 LDE, for both kernel families and every pass-split / tile tunable."""
    hal.set_tunable("ntt_fast", fast)
    hal.set_tunable("ntt_block_log", block_log)
    hal.set_tunable("ntt_tile_log", tile_log)
    hal.set_tunable("ntt_tile_a_log", tile_a)
    hal.set_tunable("ntt_tile_b_log", tile_b)
    try:
        n, count = 1 << 20, 2 + N_ADV
        x = np.concatenate([rnd(2020, n * 2), adversarial_columns(n)])  # two random columns + all-0, all-(P-1), two impulses, 0/P-1
        ref = x.copy()
        io = hal.copy_from(x)
        hal.batch_interpolate_ntt(io, count)
        oracle.bxo_batch_interpolate_ntt(ref, count, n)
        assert np.array_equal(io.view(), ref)
        zk = hal.copy_from(x)
        hal.batch_interpolate_zk(zk, count)
        ref_zk = ref.copy()
        oracle.bxo_zk_shift(ref_zk, count, n)
        assert np.array_equal(zk.view(), ref_zk)
        out = hal.alloc(4 * n * count)
        hal.batch_expand_into_evaluate_ntt(out, io, count, 2)
        lde = out.view()
        # size-independent property: without zk_shift every 4th LDE point is the original evaluation
        assert np.array_equal(lde.reshape(count, 4 * n)[:, ::4].reshape(-1), x)
        ref_out = np.zeros(4 * n * count, np.uint32)
        oracle.bxo_batch_expand_into_evaluate_ntt(ref_out, ref, count, n, 2)
        assert np.array_equal(lde, ref_out)
        # evaluate(interpolate(x)) == x
        hal.batch_evaluate_ntt(io, count, 0)
        assert np.array_equal(io.view(), x)
    finally:
        hal.set_tunable("ntt_fast", 1)
        hal.set_tunable("ntt_block_log", 12)
        hal.set_tunable("ntt_tile_log", 14)
        hal.set_tunable("ntt_tile_a_log", 12)
        hal.set_tunable("ntt_tile_b_log", 13)


@pytest.mark.parametrize("bits", [21, 22, 23, 24])
def test_ntt_largest_segment_sizes(hal, oracle, bits):
    """po2 21-24 segments: 2^23 ... 2^26-point LDEs (pass B with 2^11 ... 2^13 rows, pass A with 2^13 at the top), one column, vs
    the oracle."""
    n = 1 << bits
    x = rnd(bits, n)
    ref = x.copy()
    io = hal.copy_from(x)
    hal.batch_interpolate_ntt(io, 1)
    oracle.bxo_batch_interpolate_ntt(ref, 1, n)
    assert np.array_equal(io.view(), ref)
    out = hal.alloc(4 * n)
    hal.batch_expand_into_evaluate_ntt(out, io, 1, 2)
    ref_out = np.zeros(4 * n, np.uint32)
    oracle.bxo_batch_expand_into_evaluate_ntt(ref_out, ref, 1, n, 2)
    assert np.array_equal(out.view(), ref_out)
    hal.batch_bit_reverse(io, 1)
    oracle.bxo_batch_bit_reverse(ref, 1, n)
    assert np.array_equal(io.view(), ref)


@pytest.mark.parametrize("cpw,count", [(1, 5), (2, 5), (4, 7), (8, 3), (8, 17), (16, 16)])
def test_lde_columns_per_workgroup(hal, oracle, cpw, count):
    """Forward pass A shares one load of the tile's twist/twiddles between `cpw` columns: ragged column counts."""
    hal.set_tunable("ntt_cols_per_wg", cpw)
    try:
        n = 1 << 14
        x = rnd(cpw * 100 + count, n * count)
        out = hal.alloc(4 * n * count)
        hal.batch_expand_into_evaluate_ntt(out, hal.copy_from(x), count, 2)
        ref = np.zeros(4 * n * count, np.uint32)
        oracle.bxo_batch_expand_into_evaluate_ntt(ref, x, count, n, 2)
        assert np.array_equal(out.view(), ref)
        io = hal.copy_from(x)
        hal.batch_evaluate_ntt(io, count, 0)
        ref2 = x.copy()
        oracle.bxo_batch_evaluate_ntt(ref2, count, n, 0)
        assert np.array_equal(io.view(), ref2)
    finally:
        hal.set_tunable("ntt_cols_per_wg", 8)


def test_ntt_linearity_full_size(hal):
    n = 1 << 20
    a, b = rnd(1, n), rnd(2, n)
    s = ((a.astype(np.uint64) + b) % P).astype(np.uint32)
    outs = []
    for v in (a, b, s):
        io = hal.copy_from(v)
        hal.batch_interpolate_ntt(io, 1)
        outs.append(io.view().astype(np.uint64))
    assert np.array_equal((outs[0] + outs[1]) % P, outs[2])


def test_ntt_errors(hal):
    from boundless_amd.hal import HalError

    io = hal.alloc(24)
    with pytest.raises(HalError):
        hal.batch_interpolate_ntt(io, 1)  # 24 is not a power of two
    with pytest.raises(HalError):
        hal.batch_interpolate_ntt(io, 5)  # not divisible
    out = hal.alloc(64)
    with pytest.raises(HalError):
        hal.batch_expand_into_evaluate_ntt(out, hal.alloc(32), 1, 2)  # 32<<2 != 64


# ------------------------------------------------------------------ Poseidon2 / Merkle
def test_poseidon2_params_are_the_published_instance(hal, golden_dir):
    g = json.load(open(os.path.join(golden_dir, "poseidon2_kat.json")))
    rc, diag = hal.poseidon2_get_params()
    assert rc.tolist() == g["round_constants"] and diag.tolist() == g["internal_diag"]


def test_poseidon2_golden_sponge_pair_merkle(hal, golden_dir):
    g = json.load(open(os.path.join(golden_dir, "poseidon2_sponge.json")))
    # sponge cases as 1-row matrices (cols = len)
    for case in g["sponge"]:
        if not case["in"]:
            continue
        m = hal.copy_from(ol.encode(case["in"]))
        d = hal.alloc_digest(1)
        hal.hash_rows(d, m)
        assert ol.decode(d.view()).tolist() == case["digest"]
    # pair
    io = hal.alloc_digest(4)
    host = np.zeros(32, np.uint32)
    host[16:24] = ol.encode(g["pair"]["a"])
    host[24:32] = ol.encode(g["pair"]["b"])
    io.copy_from(host)
    hal.hash_fold(io, 2, 1)
    assert ol.decode(io.view()[8:16]).tolist() == g["pair"]["out"]
    # merkle
    mk = g["merkle"]
    rows = mk["rows"]
    mat = hal.copy_from(ol.encode(np.array(mk["matrix_colmajor"], dtype=np.uint64).reshape(-1)))
    nodes = hal.alloc_digest(2 * rows)
    hal.merkle_build(nodes, mat, rows)
    v = nodes.view()
    assert ol.decode(v[8 * rows :].reshape(rows, 8)).tolist() == mk["leaves"]
    assert ol.decode(v[8:16]).tolist() == mk["root"]


@pytest.mark.parametrize("rows,cols", [(1, 1), (64, 15), (64, 16), (100, 17), (1000, 40), (4096, 64), (1 << 14, 33)])
def test_hash_rows_vs_oracle(hal, oracle, rows, cols):
    x = rnd(rows * 31 + cols, rows * cols)
    out = hal.alloc_digest(rows)
    hal.hash_rows(out, hal.copy_from(x))
    ref = np.zeros(8 * rows, np.uint32)
    oracle.bxo_hash_rows(ref, x, rows, cols)
    assert np.array_equal(out.view(), ref)


def test_hash_rows_extreme_values(hal, oracle):
    """Lazy-reduction bounds: rows made of the extreme words 0, 1, P-1 and mixtures must still match exactly."""
    rows, cols = 256, 48
    rng = np.random.default_rng(99)
    pool = np.array([0, 1, 2, P - 1, P - 2, (P - 1) // 2, 268435454, 1172168163], dtype=np.uint32)
    x = pool[rng.integers(0, len(pool), rows * cols)]
    x[:cols] = P - 1  # note: column-major, so this sets the first rows of column 0; set whole rows explicitly below
    m = x.reshape(cols, rows)
    m[:, 0] = P - 1
    m[:, 1] = 0
    m[:, 2] = 1
    m[:, 3] = P - 2
    x = np.ascontiguousarray(m.reshape(-1))
    out = hal.alloc_digest(rows)
    hal.hash_rows(out, hal.copy_from(x))
    ref = np.zeros(8 * rows, np.uint32)
    oracle.bxo_hash_rows(ref, x, rows, cols)
    assert np.array_equal(out.view(), ref)
    # digests of extreme words through hash_fold as well
    nodes = np.zeros(4 * 8, np.uint32)
    nodes[16:24] = P - 1
    nodes[24:32] = P - 1
    io = hal.copy_from(nodes)
    hal.hash_fold(io, 2, 1)
    want = np.zeros(8, np.uint32)
    oracle.bxo_hash_pair(want, c(nodes[16:24]), c(nodes[24:32]))
    assert np.array_equal(io.view()[8:16], want)


@pytest.fixture(params=[1, 0], ids=["quad", "lane"])
def fold_path(hal, request):
    """Small Merkle layers on both kernels: four lanes per node (default) and one lane per node."""
    hal.set_tunable("fold_quad", request.param)
    yield request.param
    hal.set_tunable("fold_quad", 1)


@pytest.mark.parametrize("rows", [2, 4, 8, 256, 512, 1024, 4096, 1 << 15, 1 << 17, 1 << 19])
def test_merkle_build_vs_oracle(hal, oracle, fold_path, rows):
    cols = 20
    x = rnd(rows, rows * cols)
    nodes = hal.alloc_digest(2 * rows)
    nodes.copy_from(np.zeros(16 * rows, np.uint32))
    hal.merkle_build(nodes, hal.copy_from(x), rows)
    ref = np.zeros(16 * rows, np.uint32)
    leaves = np.zeros(8 * rows, np.uint32)
    oracle.bxo_hash_rows(leaves, x, rows, cols)
    ref[8 * rows :] = leaves
    size = rows
    while size > 1:
        oracle.bxo_hash_fold(ref, size, size // 2)
        size //= 2
    got = nodes.view()
    assert np.array_equal(got[8:], ref[8:])
    # the same tree through the layer-by-layer Hal::hash_fold calls
    nodes2 = hal.alloc_digest(2 * rows)
    nodes2.copy_from(np.concatenate([np.zeros(8 * rows, np.uint32), leaves]))
    size = rows
    while size > 1:
        hal.hash_fold(nodes2, size, size // 2)
        size //= 2
    assert np.array_equal(nodes2.view()[8:], ref[8:])


def test_poseidon2_set_params_roundtrip(hal, oracle):
    """A different parameter table changes the digests identically on both sides, and restoring works."""
    rc0, d0 = hal.poseidon2_get_params()
    rng = np.random.default_rng(3)
    rc1 = rng.integers(0, P, 213, dtype=np.uint32)
    d1 = rng.integers(0, P, 24, dtype=np.uint32)
    x = rnd(5, 64 * 20)
    try:
        hal.poseidon2_set_params(rc1, d1)
        oracle.bxo_poseidon2_set_params(c(rc1), c(d1))
        out = hal.alloc_digest(64)
        hal.hash_rows(out, hal.copy_from(x))
        ref = np.zeros(8 * 64, np.uint32)
        oracle.bxo_hash_rows(ref, x, 64, 20)
        assert np.array_equal(out.view(), ref)
    finally:
        hal.poseidon2_set_params(rc0, d0)
        oracle.bxo_poseidon2_set_params(c(rc0), c(d0))


# ------------------------------------------------------------------ FRI / DEEP
@pytest.mark.parametrize("count", [1, 16, 256, 1 << 12, 1 << 16])
def test_fri_fold_vs_oracle(hal, oracle, count):
    x = rnd(count, 64 * count)
    mix = rnd(9, 4)
    out = hal.alloc(4 * count)
    hal.fri_fold(out, hal.copy_from(x), mix)
    ref = np.zeros(4 * count, np.uint32)
    oracle.bxo_fri_fold(ref, x, c(mix), count)
    assert np.array_equal(out.view(), ref)


@pytest.mark.parametrize("n,count,offset", [(12, 3, 0), (13, 2, 0), (16, 5, 0), (20, 2, 0), (14, 2, 1), (12, 1, 3), (15, 3, 4)])
def test_batch_bit_reverse_tiled_and_unaligned(hal, oracle, n, count, offset):
    """The tiled kernel moves 16 bytes per lane; a buffer that does not start on a 16-byte boundary (a slice) takes the word-wise
    kernel instead.  Both against the oracle, and twice = identity."""
    size = 1 << n
    x = rnd(n * 31 + count, size * count)
    whole = hal.copy_from(np.concatenate([np.zeros(offset, np.uint32), x]))
    io = whole.slice(offset, size * count)
    hal.batch_bit_reverse(io, count)
    ref = x.copy()
    oracle.bxo_batch_bit_reverse(ref, count, size)
    assert np.array_equal(whole.view()[offset:], ref)
    assert not whole.view()[:offset].any()
    hal.batch_bit_reverse(io, count)
    assert np.array_equal(whole.view()[offset:], x)


def test_transcript_step_and_fri_fold_dev_vs_oracle(hal, oracle):
    """The device half of the Fiat-Shamir transcript (Poseidon2Rng: commit digests, hand out rate cells, permute when the pool runs
    dry or was touched) word for word against the oracle's, through chained steps of every shape; a fold fed by a challenge drawn
    on the device equals the fold fed the same words from the host."""
    rng = np.random.default_rng(77)
    state = np.zeros(25, np.uint32)
    d_state = hal.copy_from(state)
    for n_commit, n_ext in [(1, 1), (2, 1), (1, 4), (0, 3), (1, 0), (3, 5), (0, 1), (2, 2), (1, 16), (1, 1)]:
        digests = rng.integers(0, ol.P, (max(n_commit, 1), 8), dtype=np.uint32)
        out = hal.alloc(4 * max(n_ext, 1))
        hal.transcript_step(d_state, hal.copy_from(digests.reshape(-1)), n_commit, out, n_ext)
        state, want = ol.transcript_step(state, digests[:n_commit], 4 * n_ext)
        assert np.array_equal(d_state.view(), state), (n_commit, n_ext)
        assert np.array_equal(out.view()[: 4 * n_ext], want), (n_commit, n_ext)
    count = 1000
    x = rnd(count, 64 * count)
    got, ref = hal.alloc(4 * count), hal.alloc(4 * count)
    hal.fri_fold_dev(got, hal.copy_from(x), out)  # `out` holds the last challenge drawn above
    hal.fri_fold(ref, hal.copy_from(x), out.view()[:4].copy())
    assert np.array_equal(got.view(), ref.view())


def test_fri_fold_golden(hal, golden_dir):
    g = json.load(open(os.path.join(golden_dir, "fri_vectors.json")))["fold"]
    f = g["coeffs_natural"]
    total = len(f)
    count = total // 16
    bits = npo.log2(total)
    planes = np.zeros(4 * total, np.uint64)
    for j in range(total):
        for k in range(4):
            planes[k * total + npo.bitrev(j, bits)] = f[j][k]
    out = hal.alloc(4 * count)
    hal.fri_fold(out, hal.copy_from(ol.encode(planes)), ol.encode(g["mix"]))
    o = ol.decode(out.view()).tolist()
    got = [[o[k * count + npo.bitrev(q, bits - 4)] for k in range(4)] for q in range(count)]
    assert got == g["out_natural"]


@pytest.mark.parametrize("count,npoly,ncombo", [(64, 5, 2), (1000, 33, 4), (1 << 14, 16, 3), (1001, 9, 3), (6, 4, 2), (4, 1, 1), (1 << 12, 35, 3)])
def test_mix_poly_coeffs_vs_oracle(hal, oracle, count, npoly, ncombo):
    rng = np.random.default_rng(count)
    inp = rnd(count + 1, npoly * count)
    combos = rng.integers(0, ncombo, npoly, dtype=np.uint32)
    mix, start = rnd(3, 4), rnd(4, 4)
    init = rnd(5, ncombo * count * 4)
    out = hal.copy_from(init)
    hal.mix_poly_coeffs(out, start, mix, hal.copy_from(inp), hal.copy_from(combos), npoly, count)
    ref = init.copy()
    oracle.bxo_mix_poly_coeffs(ref, c(start), c(mix), inp, c(combos), npoly, count)
    assert np.array_equal(out.view(), ref)


@pytest.mark.parametrize("size,npoly,evals", [(16, 3, 4), (256, 2, 3), (8192, 3, 5), (1 << 15, 2, 3), (1 << 18, 2, 2)])
def test_batch_evaluate_any_vs_oracle(hal, oracle, size, npoly, evals):
    rng = np.random.default_rng(size)
    coeffs = rnd(size, npoly * size)
    which = rng.integers(0, npoly, evals, dtype=np.uint32)
    xs = rnd(11, 4 * evals)
    out = hal.alloc(4 * evals)
    hal.batch_evaluate_any(hal.copy_from(coeffs), npoly, hal.copy_from(which), hal.copy_from(xs), out)
    ref = np.zeros(4 * evals, np.uint32)
    oracle.bxo_batch_evaluate_any(coeffs, size, c(which), xs, ref, evals)
    assert np.array_equal(out.view(), ref)


def _bitrev_perm(n):
    idx = np.arange(1 << n, dtype=np.uint64)
    r = np.zeros_like(idx)
    for b in range(n):
        r |= ((idx >> np.uint64(b)) & np.uint64(1)) << np.uint64(n - 1 - b)
    return r.astype(np.int64)


@pytest.mark.parametrize("n,npoly,evals", [(15, 3, 5), (16, 2, 4), (18, 2, 3), (20, 2, 3)])
def test_batch_evaluate_any_over_bit_reversed_storage(hal, oracle, n, npoly, evals):
    """Extension entry point: evaluating the bit-reversed array equals the oracle's evaluation of the natural-order one."""
    size = 1 << n
    rng = np.random.default_rng(n)
    coeffs = rnd(n, npoly * size)
    perm = _bitrev_perm(n)
    stored = coeffs.reshape(npoly, size)[:, perm].reshape(-1).copy()  # stored[j] = coeffs[bitrev(j)]
    which = rng.integers(0, npoly, evals, dtype=np.uint32)
    xs = rnd(13, 4 * evals)
    out = hal.alloc(4 * evals)
    hal.batch_evaluate_any_bitrev(hal.copy_from(stored), npoly, hal.copy_from(which), hal.copy_from(xs), out)
    ref = np.zeros(4 * evals, np.uint32)
    oracle.bxo_batch_evaluate_any(coeffs, size, c(which), xs, ref, evals)
    assert np.array_equal(out.view(), ref)
    from boundless_amd.hal import HalError

    with pytest.raises(HalError, match="2\\^15"):
        hal.batch_evaluate_any_bitrev(hal.copy_from(rnd(1, 1 << 10)), 1, hal.copy_from(which[:1] * 0), hal.copy_from(xs[:4]), hal.alloc(4))


@pytest.mark.parametrize("n,count", [(0, 3), (3, 4), (11, 3), (12, 2), (13, 5), (16, 3), (20, 2), (22, 1)])
def test_batch_interpolate_zk_equals_the_two_calls(hal, n, count):
    """Extension entry point: interpolate + zk_shift fused into the inverse transform's final store."""
    size = 1 << n
    data = rnd(n + 70, size * count)
    a, b = hal.copy_from(data), hal.copy_from(data)
    hal.batch_interpolate_ntt(a, count)
    hal.zk_shift(a, count)
    hal.batch_interpolate_zk(b, count)
    assert np.array_equal(a.view(), b.view())


@pytest.mark.parametrize("n,count", [(1, 3), (5, 2), (12, 3), (16, 2), (20, 2)])
def test_batch_bit_reverse_ext(hal, n, count):
    size = 1 << n
    data = rnd(n + 40, 4 * size * count)
    buf = hal.copy_from(data)
    hal.batch_bit_reverse_ext(buf, count)
    ref = data.reshape(count, size, 4)[:, _bitrev_perm(n), :].reshape(-1)
    assert np.array_equal(buf.view(), ref)
    hal.batch_bit_reverse_ext(buf, count)  # an involution
    assert np.array_equal(buf.view(), data)


@pytest.mark.parametrize("size", [1, 5, 64, 100, 2047, 2048, 2049, 4096, 65535, 1 << 16, (1 << 16) + 33, 1 << 20, (1 << 21) + 7])
def test_poly_divide_vs_oracle(hal, oracle, size):
    poly = rnd(size, 4 * size)
    z = rnd(17, 4)
    buf = hal.copy_from(poly)
    rem = hal.alloc(4)
    hal.poly_divide(buf, z, rem)
    ref = poly.copy()
    ref_rem = np.zeros(4, np.uint32)
    oracle.bxo_poly_divide(ref, size, c(z), ref_rem)
    assert np.array_equal(buf.view(), ref)
    assert np.array_equal(rem.view(), ref_rem)


def test_eltwise_and_gather(hal, oracle):
    n = 10000
    a, b = rnd(1, n), rnd(2, n)
    out = hal.alloc(n)
    hal.eltwise_add_elem(out, hal.copy_from(a), hal.copy_from(b))
    ref = np.zeros(n, np.uint32)
    oracle.bxo_eltwise_add(ref, a, b, n)
    assert np.array_equal(out.view(), ref)
    cp = hal.alloc(n)
    hal.eltwise_copy_elem(cp, out)
    assert np.array_equal(cp.view(), ref)
    z = a.copy()
    z[::7] = 0xFFFFFFFF
    zb = hal.copy_from(z)
    hal.eltwise_zeroize_elem(zb)
    oracle.bxo_eltwise_zeroize(z, n)
    assert np.array_equal(zb.view(), z)
    # alloc_elem_init (rv32im's witness generator: every cell INVALID, then eltwise_zeroize_elem clears what was never written)
    inv = hal.alloc_elem_init(100003, 0xFFFFFFFF)
    assert np.all(inv.view() == 0xFFFFFFFF)
    inv.slice(10, 50).copy_from(a[:50])
    hal.eltwise_zeroize_elem(inv)
    v = inv.view()
    assert np.array_equal(v[10:60], a[:50]) and not v[:10].any() and not v[60:].any()
    inv.free()
    assert np.all(hal.alloc_elem_init(77, 12345).view() == 12345) and hal.alloc_elem_init(0, 5).view().size == 0
    # alloc_zeroed (Hal::alloc_extelem_zeroed): zeros even when the allocator hands back memory that was just dirtied and freed
    for _ in range(3):
        d = hal.copy_from(np.full(1 << 20, 0xDEADBEEF, np.uint32))
        d.free()
        zz = hal.alloc_zeroed(1 << 20)
        assert not zz.view().any()
        zz.free()
    count, to_add = 1000, 5
    e = rnd(3, 4 * count * to_add)
    so = hal.alloc(4 * count)
    hal.eltwise_sum_extelem(so, hal.copy_from(e))
    sref = np.zeros(4 * count, np.uint32)
    oracle.bxo_eltwise_sum_extelem(sref, e, count, to_add)
    assert np.array_equal(so.view(), sref)
    g = hal.alloc(17)
    hal.gather_sample(g, hal.copy_from(a), 5, 17, 500)
    gref = np.zeros(17, np.uint32)
    oracle.bxo_gather_sample(gref, a, 5, 17, 500)
    assert np.array_equal(g.view(), gref)
    from boundless_amd.hal import HalError

    with pytest.raises(HalError):
        hal.gather_sample(g, hal.copy_from(a), 5, 17, 1000)  # out of range


def test_eltwise_mul_factor(hal, oracle):
    n = 100003
    x = rnd(5, n)
    f = int(rnd(6, 1)[0])
    buf = hal.copy_from(x)
    hal.eltwise_mul_factor(buf, f)
    want = np.array([oracle.bxo_fp_mul(int(v), f) for v in x[:2000]], dtype=np.uint32)
    got = buf.view()
    assert np.array_equal(got[:2000], want)
    # whole array through the field definition (Montgomery product = a * b * 2^-32 mod P)
    rinv = pow(1 << 32, -1, P)
    assert np.array_equal(got, ((x.astype(object) * f * rinv) % P).astype(np.uint32))
    from boundless_amd.hal import HalError

    with pytest.raises(HalError):
        hal.eltwise_mul_factor(buf, P)


@pytest.mark.parametrize("n", [1, 2, 31, 32, 33, 63, 64, 65, 1000, 2048, 2049, 4096, 65537, 1 << 16, (1 << 18) + 3, 1 << 20, (1 << 21) + 5])
def test_prefix_products_vs_oracle(hal, oracle, n):
    x = rnd(n, 4 * n)
    buf = hal.copy_from(x)
    hal.prefix_products(buf)
    ref = x.copy()
    oracle.bxo_prefix_products(ref, n)
    assert np.array_equal(buf.view(), ref)


def test_scatter_vs_oracle(hal, oracle):
    rng = np.random.default_rng(4)
    into_len, cycles = 5000, 40
    counts = rng.integers(0, 9, cycles)
    index = np.concatenate([[0], np.cumsum(counts)]).astype(np.uint32)
    total = int(index[-1])
    offsets = rng.permutation(into_len)[:total].astype(np.uint32)  # distinct destinations, as the circuits produce
    values = rnd(8, total)
    init = rnd(9, into_len)
    dst = hal.copy_from(init)
    hal.scatter(dst, hal.copy_from(index), hal.copy_from(offsets), hal.copy_from(values))
    ref = init.copy()
    oracle.bxo_scatter(ref, index, offsets, c(values), cycles)
    assert np.array_equal(dst.view(), ref)
    from boundless_amd.hal import HalError

    # scatter is asynchronous: a bad offset / index range is detected on the device and reported by the next blocking call
    bad = offsets.copy()
    bad[3] = into_len + 7
    hal.scatter(dst, hal.copy_from(index), hal.copy_from(bad), hal.copy_from(values))
    with pytest.raises(HalError, match="offset is outside"):
        hal.sync()
    hal.sync()  # the flag is cleared once reported
    bad_index = index.copy()
    bad_index[-1] = total + 1
    hal.scatter(dst, hal.copy_from(bad_index), hal.copy_from(offsets), hal.copy_from(values))
    with pytest.raises(HalError, match="index range"):
        dst.view()
    # a sub-range [index[0], index[last]) of the entries: only those are written
    dst2 = hal.copy_from(init)
    hal.scatter(dst2, hal.copy_from(index[5:21]), hal.copy_from(offsets), hal.copy_from(values))
    ref2 = init.copy()
    oracle.bxo_scatter(ref2, c(index[5:21]), offsets, c(values), 15)
    assert np.array_equal(dst2.view(), ref2)


@pytest.mark.parametrize("n,count", [(2, 3), (64, 2), (1000, 5), (2049, 4), (4096, 16), ((1 << 16) + 1, 3), (65537, 2), (1 << 20, 3)])
def test_batch_prefix_products_vs_oracle(hal, oracle, n, count):
    """`count` independent sequences back to back (the accumulate step's shape): each equals the single-sequence result."""
    x = rnd(n + count, 4 * n * count)
    buf = hal.copy_from(x)
    hal.batch_prefix_products(buf, count)
    ref = x.copy()
    for k in range(count):
        seq = np.ascontiguousarray(ref[4 * n * k: 4 * n * (k + 1)])
        oracle.bxo_prefix_products(seq, n)
        ref[4 * n * k: 4 * n * (k + 1)] = seq
    assert np.array_equal(buf.view(), ref)


def test_torch_memory_and_stream_interop(hal, oracle):
    """bx_buf is a plain (pointer, length) pair and a ctx can adopt an external stream: run the LDE on a torch tensor
    on torch's current stream (PyTorch is plumbing here: device memory + streams)."""
    import torch

    n, cols = 1 << 12, 4
    x = rnd(77, n * cols)
    t_in = torch.from_numpy(x.view(np.int32)).to("cuda")
    t_out = torch.empty(4 * n * cols, dtype=torch.int32, device="cuda")
    stream = torch.cuda.Stream()
    with torch.cuda.stream(stream):
        hal.set_stream(stream.cuda_stream)
        try:
            a = hal.wrap(t_in.data_ptr(), t_in.numel())
            b = hal.wrap(t_out.data_ptr(), t_out.numel())
            hal.batch_interpolate_ntt(a, cols)
            hal.zk_shift(a, cols)
            hal.batch_expand_into_evaluate_ntt(b, a, cols, 2)
            # a gather followed by torch's own work on the adopted stream: the library must not hold the gather back in its queue
            # (it cannot see what the stream's owner enqueues), so torch reads the gathered words
            t_row = torch.zeros(cols, dtype=torch.int32, device="cuda")
            hal.gather_sample(hal.wrap(t_row.data_ptr(), cols), b, 5, cols, 4 * n)
            t_copy = t_row.clone()  # enqueued by torch on the same stream, without passing through the library
        finally:
            stream.synchronize()
            hal.set_stream(None)
    ref = x.copy()
    oracle.bxo_batch_interpolate_ntt(ref, cols, n)
    oracle.bxo_zk_shift(ref, cols, n)
    ref_out = np.zeros(4 * n * cols, np.uint32)
    oracle.bxo_batch_expand_into_evaluate_ntt(ref_out, ref, cols, n, 2)
    assert np.array_equal(t_out.cpu().numpy().view(np.uint32), ref_out)
    assert np.array_equal(t_copy.cpu().numpy().view(np.uint32), ref_out.reshape(cols, 4 * n)[:, 5])


# ------------------------------------------------------------------ degenerate inputs
def test_empty_and_null_inputs_return_errors_or_do_nothing(hal, oracle):
    """The boundary contract (bx_hal.h: "nothing aborts"): every entry point, handed empty buffers or a NULL context, returns —
    NULL for a well-formed no-op, an error string otherwise — and leaves the context usable."""
    import ctypes as C

    from boundless_amd.hal import BxBuf

    L, ctx = hal.lib, hal.ctx
    E = BxBuf(None, 0)
    mix = (C.c_uint32 * 4)(1, 2, 3, 4)
    calls = {
        "bx_batch_interpolate_ntt": (E, 1), "bx_batch_interpolate_zk": (E, 1), "bx_batch_evaluate_ntt": (E, 1, 0),
        "bx_batch_expand_into_evaluate_ntt": (E, E, 1, 2), "bx_batch_bit_reverse": (E, 1), "bx_batch_bit_reverse_ext": (E, 1),
        "bx_zk_shift": (E, 1), "bx_hash_rows": (E, E), "bx_hash_fold": (E, 0, 0), "bx_merkle_build": (E, E, 0),
        "bx_fri_fold": (E, E, mix), "bx_mix_poly_coeffs": (E, mix, mix, E, E, 0, 1),
        "bx_batch_evaluate_any": (E, 1, E, E, E), "bx_batch_evaluate_any_bitrev": (E, 1, E, E, E),
        "bx_eltwise_add_elem": (E, E, E), "bx_eltwise_copy_elem": (E, E), "bx_eltwise_zeroize_elem": (E,),
        "bx_eltwise_sum_extelem": (E, E), "bx_eltwise_mul_factor": (E, 1), "bx_gather_sample": (E, E, 0, 0, 1),
        "bx_poly_divide": (E, mix, E), "bx_prefix_products": (E,), "bx_batch_prefix_products": (E, 1), "bx_scatter": (E, E, E, E),
    }
    outcomes = {}
    for name, args in calls.items():
        fn = getattr(L, name)
        msg = fn(ctx, *args)  # must return: NULL or a string
        outcomes[name] = None if not msg else msg.decode()
        null_ctx = fn(None, *args)  # a NULL context is an error string, never a crash
        assert null_ctx and b"null" in null_ctx.lower(), name
    # zero-count batches are rejected by name, not by a HIP launch failure
    for name in ("bx_batch_interpolate_ntt", "bx_batch_bit_reverse", "bx_zk_shift", "bx_hash_rows", "bx_merkle_build"):
        assert outcomes[name] is not None and "hip" not in outcomes[name].lower(), (name, outcomes[name])
    # well-formed no-ops
    for name in ("bx_fri_fold", "bx_eltwise_add_elem", "bx_eltwise_copy_elem", "bx_eltwise_zeroize_elem", "bx_eltwise_mul_factor",
                 "bx_gather_sample", "bx_prefix_products", "bx_scatter"):
        assert outcomes[name] is None, (name, outcomes[name])
    assert not any(v and "hip" in v.lower() for v in outcomes.values()), outcomes  # no launch ever failed
    # count = 0 with real buffers
    buf = hal.copy_from(rnd(1, 64))
    for name in ("bx_batch_interpolate_ntt", "bx_batch_bit_reverse", "bx_zk_shift"):
        assert getattr(L, name)(ctx, buf.raw, 0)
    # a matrix with rows but no columns hashes the empty slice (one permutation of the zero state), as the oracle does
    dg = hal.alloc_digest(8)
    hal._check(L.bx_hash_rows(ctx, dg.raw, E))
    ref = np.zeros(64, np.uint32)
    oracle.bxo_hash_rows(ref, np.zeros(1, np.uint32), 8, 0)
    assert np.array_equal(dg.view(), ref) and len(set(ref.reshape(8, 8)[:, 0].tolist())) == 1 and ref.any()
    # the context still works
    hal.sync()
    want = rnd(1, 64)
    oracle.bxo_batch_interpolate_ntt(want, 1, 64)
    hal.batch_interpolate_ntt(buf, 1)
    assert np.array_equal(buf.view(), want)


@pytest.mark.parametrize("lookback", [1, 0])
def test_scans_single_pass_and_three_phase_agree_with_the_oracle_across_tile_boundaries(hal, oracle, lookback):
    """poly_divide / prefix_products: the look-back kernels (scan.hip) and the three-phase kernels, on sizes around the 2048-element
    tile and the 64-tile look-back window (131072), back to back on one ctx so that every launch finds the state the previous
    one left (two alternating buffers, cleared by the launch that does not use them)."""
    hal.set_tunable("scan_lookback", lookback)
    try:
        for size in (1, 7, 2047, 2048, 2049, 4096, 10000, 131072, 131073, 300001, 1 << 20, 5, 70000):
            poly = rnd(size, 4 * size)
            z = rnd(size + 1, 4)
            buf, rem = hal.copy_from(poly), hal.alloc(4)
            hal.poly_divide(buf, z, rem)
            ref, ref_rem = poly.copy(), np.zeros(4, np.uint32)
            oracle.bxo_poly_divide(ref, size, c(z), ref_rem)
            assert np.array_equal(buf.view(), ref) and np.array_equal(rem.view(), ref_rem), ("divide", size)
            if size >= 2:
                x = rnd(size + 2, 4 * size)
                b2 = hal.copy_from(x)
                hal.prefix_products(b2)
                r2 = x.copy()
                oracle.bxo_prefix_products(r2, size)
                assert np.array_equal(b2.view(), r2), ("prefix", size)
    finally:
        hal.set_tunable("scan_lookback", 1)


def test_poly_divide_batch_divides_every_polynomial_by_its_own_point(hal, oracle):
    for size, count in ((5000, 3), (1 << 16, 9), (2048, 1), (100, 16)):
        polys = rnd(size + count, 4 * size * count)
        zs = rnd(count, 4 * count)
        buf, rems = hal.copy_from(polys), hal.alloc(4 * count)
        hal.poly_divide_batch(buf, count, zs, rems)
        got, got_rems = buf.view().reshape(count, 4 * size), rems.view().reshape(count, 4)
        for q in range(count):
            ref, ref_rem = polys.reshape(count, 4 * size)[q].copy(), np.zeros(4, np.uint32)
            oracle.bxo_poly_divide(ref, size, c(zs[4 * q:4 * q + 4]), ref_rem)
            assert np.array_equal(got[q], ref) and np.array_equal(got_rems[q], ref_rem), (size, count, q)
    # a quotient times (x - z) plus the remainder is the polynomial again: dividing (x - z) * q leaves remainder zero
    with pytest.raises(Exception):
        hal.poly_divide_batch(hal.alloc(4 * 10), 3, rnd(1, 12), hal.alloc(12))


def test_batch_evaluate_ptrs_over_several_buffers_matches_per_buffer_calls(hal, oracle):
    """The DEEP step's one-call form: evaluations over columns of several coefficient buffers, some stored bit-reversed."""
    n, size = 15, 1 << 15
    bufs = [rnd(40 + k, size * w) for k, w in enumerate((2, 3, 1))]
    rev = _bitrev_perm(n)
    dev, ptrs, flags, which_ref, xs = [], [], [], [], rnd(77, 4 * 9)
    stored_bitrev = (True, False, True)
    for k, b in enumerate(bufs):
        cols = b.reshape(-1, size)
        stored = cols[:, rev] if stored_bitrev[k] else cols  # position j holds the coefficient of x^rev(j)
        dev.append(hal.copy_from(np.ascontiguousarray(stored).reshape(-1)))
    for e in range(9):
        k = e % 3
        col = (e // 3) % (bufs[k].size // size)
        addr = dev[k].raw.dptr + 4 * col * size
        ptrs += [addr & 0xFFFFFFFF, addr >> 32]
        flags.append(1 if stored_bitrev[k] else 0)
        which_ref.append((k, col))
    out = hal.alloc(4 * 9)
    hal.batch_evaluate_ptrs(hal.copy_from(np.array(ptrs, np.uint32)), hal.copy_from(np.array(flags, np.uint32)), size, hal.copy_from(xs), out)
    got = out.view().reshape(9, 4)
    for e, (k, col) in enumerate(which_ref):
        ref = np.zeros(4, np.uint32)
        oracle.bxo_batch_evaluate_any(np.ascontiguousarray(bufs[k].reshape(-1, size)[col]), size, np.zeros(1, np.uint32), c(xs[4 * e:4 * e + 4]), ref, 1)
        assert np.array_equal(got[e], ref), e


@pytest.mark.parametrize("defer", [1, 0])
def test_gather_sample_queue_keeps_the_stream_order_the_caller_sees(hal, defer):
    """bx_gather_sample queues small gathers and launches them together (ctx.hpp: ~5 000 openings per proof driven through the plain
    trait calls).  Whatever could observe the difference must not: chains of gathers through the same memory, a source overwritten
    after it was gathered from, a destination read back at once, more gathers than the queue holds, and the un-queued path."""
    hal.set_tunable("gather_defer", defer)
    try:
        rows, cols = 1 << 10, 24
        m = rnd(910, rows * cols)
        mat = hal.copy_from(m)
        # 1. many independent gathers (rows of a column-major matrix), one read-back
        out = hal.alloc(cols * 300)
        pos = np.random.default_rng(5).integers(0, rows, 300)
        for q, r in enumerate(pos):
            hal.gather_sample(out.slice(q * cols, cols), mat, int(r), cols, rows)
        assert np.array_equal(out.view().reshape(300, cols), m.reshape(cols, rows)[:, pos].T)
        # 2. a chain: B <- gather(A), C <- gather(B), A <- gather(C)   (each reads what the previous one wrote)
        a = hal.copy_from(np.arange(64, dtype=np.uint32))
        b, c2 = hal.alloc_zeroed(32), hal.alloc_zeroed(16)
        hal.gather_sample(b, a, 1, 32, 2)      # b[i] = a[1 + 2i] = 1 + 2i
        hal.gather_sample(c2, b, 0, 16, 2)     # c[i] = b[2i] = 1 + 4i
        hal.gather_sample(a.slice(0, 16), c2, 0, 16, 1)  # a[0..16) = c
        assert np.array_equal(a.view()[:16], 1 + 4 * np.arange(16, dtype=np.uint32))
        assert np.array_equal(b.view(), 1 + 2 * np.arange(32, dtype=np.uint32))
        # 3. the source is overwritten AFTER the gather was issued: the gather saw the old words
        src = hal.copy_from(np.full(128, 7, np.uint32))
        dst = hal.alloc_zeroed(128)
        hal.gather_sample(dst, src, 0, 128, 1)
        src.copy_from(np.full(128, 9, np.uint32))
        assert np.all(dst.view() == 7) and np.all(src.view() == 9)
        # 4. a destination that a later NON-gather call reads: eltwise_copy of the gathered words
        g1, g2 = hal.alloc_zeroed(cols), hal.alloc_zeroed(cols)
        hal.gather_sample(g1, mat, 3, cols, rows)
        hal.eltwise_copy_elem(g2, g1)
        assert np.array_equal(g2.view(), m.reshape(cols, rows)[:, 3])
        # 5. more gathers than the queue holds (8192), 8 words each like the path digests of an opening
        nodes = rnd(911, 8 * 4096)
        nb = hal.copy_from(nodes)
        big = hal.alloc(8 * 9000)
        idx = np.random.default_rng(6).integers(0, 4096, 9000)
        for q, i in enumerate(idx):
            hal.gather_sample(big.slice(8 * q, 8), nb, 8 * int(i), 8, 1)
        assert np.array_equal(big.view().reshape(9000, 8), nodes.reshape(4096, 8)[idx])
    finally:
        hal.set_tunable("gather_defer", 1)


def test_alloc_pool_recycles_blocks_without_changing_what_callers_see():
    """bx_alloc / bx_release go through a per-ctx pool (ctx.hpp): a released block serves the next request of about its size.  What a
    caller can observe must not change: alloc_zeroed is zero on recycled (dirty) memory, work enqueued before a release still sees its
    data, sizes that do not match get their own block, the cap and the off switch hold, and closing the ctx returns everything."""
    import torch

    from boundless_amd.hal import HipHal

    HipHal(0).close()  # the first ctx of a process loads the code objects (~150 MB that stay with the runtime): not part of what is measured
    torch.cuda.synchronize()
    free_before = torch.cuda.mem_get_info(0)[0]
    h = HipHal(0)
    try:
        n = 1 << 22
        a = h.copy_from(np.full(n, 0xDEADBEEF, np.uint32))
        ptr = a.raw.dptr
        out = h.alloc(n)
        h.eltwise_copy_elem(out, a)  # enqueued, reads `a` ...
        a.free()                     # ... released right behind it: no wait, the block goes to the pool
        z = h.alloc_zeroed(n)        # the same block comes back (same size), cleared on the stream behind the copy
        assert z.raw.dptr == ptr
        assert not z.view().any() and np.all(out.view() == 0xDEADBEEF)
        z.free()
        small = h.alloc(n // 2)      # a block twice as large as asked for is not handed out
        assert small.raw.dptr != ptr
        close = h.alloc(n - 1000)    # one within 12.5 % is
        assert close.raw.dptr == ptr
        small.free(), close.free(), out.free()
        # many sizes, interleaved lifetimes, contents checked: nothing aliases while live
        rng = np.random.default_rng(9)
        live = []
        for it in range(200):
            if live and rng.random() < 0.45:
                buf, val = live.pop(int(rng.integers(len(live))))
                assert np.all(buf.view() == val)
                buf.free()
            else:
                words = int(rng.integers(1, 1 << 18))
                buf = h.alloc(words)
                buf.copy_from(np.full(words, it, np.uint32))
                live.append((buf, it))
        ptrs = sorted((b.raw.dptr, b.raw.len) for b, _ in live)
        assert all(p0 + 4 * l0 <= p1 for (p0, l0), (p1, _) in zip(ptrs, ptrs[1:]))
        for buf, val in live:
            assert np.all(buf.view() == val)
            buf.free()
        # a block released twice is refused (it idles in the pool: a second release must not hipFree it behind the pool's back)
        from boundless_amd.hal import BxBuf

        dbl = h.alloc(n)
        raw = BxBuf(dbl.raw.dptr, dbl.raw.len)
        dbl.free()
        msg = h.lib.bx_release(h.ctx, raw)
        assert msg and b"already released" in msg
        again = [h.alloc(n) for _ in range(4)]  # still in the pool (with the other blocks of its size), still handed out
        assert raw.dptr in [b.raw.dptr for b in again]
        for b in again:
            b.free()
        # off switch: the cached blocks go back to the driver at once, releases free again
        h.set_tunable("alloc_cache_mb", 0)
        b1 = h.alloc(n)
        p1 = b1.raw.dptr
        b1.free()
        torch.cuda.synchronize()
        held = free_before - torch.cuda.mem_get_info(0)[0]
        assert held < (256 << 20), held  # only the ctx's own tables are left
        h.set_tunable("alloc_cache_mb", 1)  # a 1 MiB cap: a 16 MiB block is not kept
        b2 = h.alloc(n)
        b2.free()
        assert h.alloc(n).raw.dptr is not None
    finally:
        h.close()
    torch.cuda.synchronize()
    assert torch.cuda.mem_get_info(0)[0] >= free_before - (64 << 20)


def test_wait_policies_return_the_same_words(hal):
    """How a host thread waits for its stream (ctx.hpp: busy-poll, interrupt, event poll with a bounded spin first — wait_spin_us) is
    not observable in results: a read-back right behind a long kernel and a run of tiny read-backs give the same words under each."""
    from boundless_amd.hal import HalError

    n = 1 << 22
    src = rnd(920, n)
    try:
        for blocking, spin in [(2, 60), (2, 0), (2, 5000), (0, 60), (1, 60)]:
            hal.set_tunable("wait_blocking", blocking)
            hal.set_tunable("wait_spin_us", spin)
            a, out = hal.copy_from(src), hal.alloc(n)
            for _ in range(8):  # a few ms of device work in front of the read-back
                hal.eltwise_copy_elem(out, a)
                hal.batch_bit_reverse(out, 1)
                hal.batch_bit_reverse(out, 1)
            assert np.array_equal(out.view(), src), (blocking, spin)
            for i in range(0, 4096, 97):  # short waits: 8 words each on an idle stream
                assert np.array_equal(out.slice(i, 8).view(), src[i:i + 8]), (blocking, spin, i)
            a.free(), out.free()
        with pytest.raises(HalError, match="wait_spin_us out of range"):
            hal.set_tunable("wait_spin_us", -1)
    finally:
        hal.set_tunable("wait_blocking", 2)
        hal.set_tunable("wait_spin_us", 60)


def test_scalars_whose_products_wrap_are_refused(hal):
    """The scalars of an entry point come from the caller.  A count / size / stride so large that `4 * count` or `(size - 1) * stride`
    wraps around 2^64 must fail the bound check like any other out-of-range value — not pass it (and then divide by zero on the host or
    read far outside the buffer on the device).  Every call returns an error string; the buffers are untouched; the ctx keeps working."""
    from boundless_amd.hal import HalError

    x = rnd(930, 1 << 12)
    buf, out = hal.copy_from(x), hal.alloc_zeroed(1 << 14)
    mix = rnd(931, 4)
    combos = hal.copy_from(np.zeros(4, np.uint32))
    hostile = [
        lambda: hal.mix_poly_coeffs(out, mix, mix, buf, combos, 4, 1 << 62),              # 4 * count wraps to 0: was a division by zero
        lambda: hal.mix_poly_coeffs(out, mix, mix, buf, combos, 1 << 63, 2),              # input_size * count wraps to 0
        lambda: hal.gather_sample(out, buf, 2, 9, 1 << 61),                               # (size - 1) * stride wraps to 0
        lambda: hal.gather_sample(out, buf, 3, 3, (1 << 64) - 1),
        lambda: hal.gather_sample(out, buf, 1 << 63, 1, 1),
        lambda: hal.hash_fold(out, 0, 1 << 63),                                           # 2 * output_size wraps to 0 == input_size
        lambda: hal.merkle_build(out.slice(0, 0), buf, 1 << 60),                          # 16 * rows wraps to 0 == nodes.len
        lambda: hal.batch_expand_into_evaluate_ntt(out, buf, 1, 66),                      # a shift by more than the word
        lambda: hal.batch_expand_into_evaluate_ntt(out, buf, 1, 64),
        lambda: hal.batch_evaluate_ntt(buf, 1, (1 << 32) + 1),                            # truncated to int it was 1
        lambda: hal.batch_bit_reverse_ext(buf, 1 << 62),
        lambda: hal.batch_interpolate_ntt(buf, 1 << 63),
        lambda: hal.fri_fold(out.slice(0, 4), buf.slice(0, 63), mix),
        lambda: hal.alloc(1 << 62),                                                       # words * 4 wraps to 0 bytes
        lambda: hal.alloc_zeroed((1 << 62) + 1),
    ]
    for k, call in enumerate(hostile):
        with pytest.raises(HalError):
            call()
            pytest.fail(f"hostile call {k} was accepted")
    assert np.array_equal(buf.view(), x) and not out.view().any()
    hal.gather_sample(out, buf, 5, 7, 100)  # the largest in-range stride pattern still works
    assert np.array_equal(out.view()[:7], x[5:5 + 700:100])


def test_eltwise_copy_elem_slice_places_a_host_region_with_strides(hal):
    """Hal::eltwise_copy_elem_slice: into[io + r*is + c] = from[fo + r*fs + c] from a HOST slice — how the prover places a witness of
    `steps` rows per column into buffers of 2^po2 rows per column.  Against numpy, words outside the region untouched, every run
    shape (one contiguous run, a 2-D copy, overlapping source rows), and the regions that leave either side refused."""
    from boundless_amd.hal import HalError

    rng = np.random.default_rng(940)
    for rows, cols, fs, is_, fo, io in [(1, 17, 0, 0, 3, 5), (7, 100, 100, 100, 0, 0), (5, 33, 40, 64, 11, 9), (256, 1000, 1000, 1024, 0, 0),
                                        (4, 16, 8, 16, 2, 1), (3, 1, 1, 1, 0, 0), (1 << 10, 3, 5, 7, 4, 4)]:
        src = rnd(rows * 31 + cols, fo + (rows - 1) * fs + cols + 13)
        init = rnd(cols, io + (rows - 1) * is_ + cols + 29)
        dst = hal.copy_from(init)
        hal.eltwise_copy_elem_slice(dst, src, rows, cols, fo, fs, io, is_)
        want = init.copy()
        for r in range(rows):
            want[io + r * is_: io + r * is_ + cols] = src[fo + r * fs: fo + r * fs + cols]
        assert np.array_equal(dst.view(), want), (rows, cols, fs, is_, fo, io)
    # the witness shape: 24 columns of 1000 steps into columns of 1024 rows (column-major on both sides)
    steps, n, w = 1000, 1024, 24
    wit = rnd(941, steps * w)
    data = hal.alloc_zeroed(n * w)
    hal.eltwise_copy_elem_slice(data, wit, w, steps, 0, steps, 0, n)
    got = data.view().reshape(w, n)
    assert np.array_equal(got[:, :steps], wit.reshape(w, steps)) and not got[:, steps:].any()
    dst = hal.alloc_zeroed(100)
    src = rnd(942, 100)
    for bad in [(2, 60, 0, 60, 0, 60), (2, 10, 95, 1, 0, 10), (2, 10, 0, 10, 85, 10), (3, 10, 0, 10, 0, 5), (2, 1, 0, 1 << 63, 0, 1),
                (1 << 62, 4, 0, 4, 0, 4), (1, 101, 0, 0, 0, 0)]:
        with pytest.raises(HalError):
            hal.eltwise_copy_elem_slice(dst, src, *bad)
    assert not dst.view().any()
    hal.eltwise_copy_elem_slice(dst, src, 0, 5, 0, 0, 0, 0)  # nothing to copy: not an error
