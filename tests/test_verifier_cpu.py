"""CPU: the product's host-side verifier (bx_verify_segment) accepts honest seals and rejects tampered ones.

The seals come from the CPU oracle's prover, so this also cross-checks the product's host transcript / Poseidon2 code
(transcript.hpp, used by both the HIP prover and the verifier) against the oracle without a GPU.
"""
import numpy as np
import pytest

from boundless_amd.hal import HalError
from boundless_amd.prover import verify_seal
from oracle import oracle_lib as ol


@pytest.mark.parametrize("po2,widths,seed", [(9, (1, 1, 1), 3), (10, (4, 8, 4), 1234), (12, (3, 17, 5), 99), (13, (2, 9, 6), 5),
                                             (11, (16, 40, 12), 8), (10, (2, 16, 8), 2), (10, (1, 16, 8), 2), (10, (2, 2, 4), 5),
                                             (9, (3, 3, 8), 6), (10, (40, 6, 16), 7)])
def test_honest_seal_is_accepted(po2, widths, seed):
    seal, _ = ol.prove_segment(po2, *widths, seed)
    verify_seal(seal)


def _regions(n):
    """Word offsets into the seal of the (10, 4/8/4) segment, one or two per region."""
    taps = 4 + (8 + 1 + 2) + (4 + 4) + 16
    h = 6 + 2  # header + the two public words
    return {
        "header": 1, "header_terms": 4, "global_start": 6, "global_end": 7, "code_top": h + 5, "data_top": h + 256 + 9, "accum_top": h + 512 + 3, "check_top": h + 768 + 100,
        "coeff_u": h + 1024 + 7, "coeff_u_2tap": h + 1024 + 4 * 4 + 5, "coeff_u_3tap": h + 1024 + 4 * (4 + 5 + 2) + 1, "coeff_u_check": h + 1024 + 4 * (taps - 3), "fri_top": h + 1024 + 4 * taps + 11,
        "final": h + 1024 + 4 * taps + 256 + 5, "query_first": h + 1024 + 4 * taps + 256 + 256 + 2,
        "query_sibling": h + 1024 + 4 * taps + 256 + 256 + 4 + 3, "last": n - 1,
    }


def test_honest_seals_of_random_shapes_are_accepted_and_one_flipped_bit_is_not():
    """The same seeded shape sweep the GPU parity test proves (tests/test_prover_gpu.py), here oracle -> verifier."""
    rng = np.random.default_rng(20260927)
    for _ in range(12):
        po2 = int(rng.integers(9, 14))
        widths = (int(rng.integers(1, 24)), int(rng.integers(1, 48)), int(rng.integers(1, 14)))
        terms, degree, seed = int(rng.integers(1, 65)), int(rng.integers(1, 6)), int(rng.integers(0, 2**63))
        seal, _ = ol.prove_segment(po2, *widths, seed, terms=terms, degree=degree)
        verify_seal(seal)
        bad = seal.copy()
        at = int(rng.integers(0, seal.size))
        bad[at] ^= np.uint32(1 << int(rng.integers(0, 30)))
        with pytest.raises(HalError):
            verify_seal(bad)


def test_truncated_padded_and_garbage_seals_are_rejected_without_reading_out_of_bounds():
    """Every prefix of a seal, a seal with trailing words, and arbitrary word strings are errors — the reader is bounded
    (tests/test_verifier_sanitizers_cpu.py runs the same mutations under ASan/UBSan)."""
    seal, _ = ol.prove_segment(9, 2, 3, 2, 11)
    rng = np.random.default_rng(5)
    lengths = sorted(set(list(range(0, 48)) + rng.integers(48, seal.size, 150).tolist() + [seal.size - 1]))
    for n in lengths:
        with pytest.raises(HalError):
            verify_seal(seal[:n].copy())
    with pytest.raises(HalError):
        verify_seal(np.concatenate([seal, np.zeros(1, np.uint32)]))
    for _ in range(50):
        junk = rng.integers(0, 2**32, int(rng.integers(1, 4096)), dtype=np.uint32)
        junk[0] = rng.integers(0, 30)  # a plausible po2, so the header gets past the first check sometimes
        with pytest.raises(HalError):
            verify_seal(junk)
    # header fields at their extremes: the size computation must not overflow into acceptance
    for hdr in ([22, 65535, 65535, 65535, 64, 5], [9, 0, 1, 1, 64, 4], [9, 1, 1, 1, 65, 4], [9, 1, 1, 1, 64, 6], [23, 1, 1, 1, 1, 1],
                [0xFFFFFFFF] * 6):
        with pytest.raises(HalError):
            verify_seal(np.concatenate([np.array(hdr, np.uint32), seal[6:]]))


def test_tampering_anywhere_is_rejected():
    seal, _ = ol.prove_segment(10, 4, 8, 4, 1234)
    verify_seal(seal)
    n = seal.size
    rng = np.random.default_rng(0)
    # header, code top layer, data top layer, coeff_u, FRI top, final coefficients, query openings, last word
    offs = _regions(n)
    for name, off in offs.items():
        bad = seal.copy()
        bad[off] = (int(bad[off]) + 1) % ol.P
        with pytest.raises(HalError):
            verify_seal(bad)
    for off in rng.integers(0, n, 40):
        bad = seal.copy()
        bad[off] ^= 1
        with pytest.raises(HalError):
            verify_seal(bad)
    with pytest.raises(HalError):
        verify_seal(seal[:-1])
    with pytest.raises(HalError):
        verify_seal(np.concatenate([seal, seal[:1]]))


def test_non_canonical_words_are_rejected():
    """x + P behaves like x in the lazily reduced arithmetic (hashes, sums and products agree), so the verifier has to refuse
    every word >= P explicitly: otherwise a seal could be re-encoded without invalidating it (round-1 advisor finding)."""
    seal, _ = ol.prove_segment(10, 4, 8, 4, 1234)
    n = seal.size
    offs = {k: v for k, v in _regions(n).items() if not k.startswith("header")}
    rng = np.random.default_rng(1)
    for off in list(offs.values()) + rng.integers(6, n, 200).tolist():
        bad = seal.copy()
        bad[off] = int(bad[off]) + ol.P  # < 2^32 for every canonical word
        with pytest.raises(HalError, match="non-canonical|header"):
            verify_seal(bad)
    bad = seal.copy()
    bad[10] = 0xFFFFFFFF  # the INVALID marker of unset cells is not a field element either
    with pytest.raises(HalError, match="non-canonical"):
        verify_seal(bad)


def test_the_public_words_are_bound_to_the_trace():
    """The seal carries the statement's public words (the first cell of data column 0 and the last ACTIVE cell of the last data
    column: the ZK noise rows come after it) and proves a trace that starts and ends there: claiming other values for the same proof is refused, and a prover
    that honestly proves a trace with another end cell gets another (valid) claim, not the old one."""
    L = ol.lib()
    widths = (4, 16, 8)
    seal, _ = ol.prove_segment(10, *widths, 99)
    verify_seal(seal)
    g0, g1 = int(seal[6]), int(seal[7])
    for off in (6, 7):  # a different claim for the same proof
        bad = seal.copy()
        bad[off] = (int(bad[off]) + 1) % ol.P
        with pytest.raises(HalError):
            verify_seal(bad)
    try:  # the witness changed in the very cell g_1 reports: the constraint that defines that derived cell now fails
        L.bxo_set_witness_fault(1, widths[1] - 1, (1 << 10) - (1 << 8) - 1)  # last active row: N - min(1994, N/4) - 1
        forged, _ = ol.prove_segment(10, *widths, 99)
    finally:
        L.bxo_set_witness_fault(-1, 0, 0)
    assert int(forged[6]) == g0 and int(forged[7]) != g1
    with pytest.raises(HalError, match="constraint identity"):
        verify_seal(forged)
    try:  # a trace with another first cell reports another g_0 ...
        L.bxo_set_witness_fault(1, 0, 0)
        other, _ = ol.prove_segment(10, *widths, 99)
    finally:
        L.bxo_set_witness_fault(-1, 0, 0)
    assert int(other[6]) != g0
    swapped = other.copy()
    swapped[6] = g0  # ... and rewriting the claim back to the original g_0 is refused
    with pytest.raises(HalError):
        verify_seal(swapped)


def test_a_seal_of_a_different_circuit_is_rejected():
    """The circuit's knobs are part of the statement: a proof for (terms, degree) = (5, 4) does not verify as one for the defaults."""
    seal, _ = ol.prove_segment(10, 4, 8, 4, 1234, terms=5, degree=4)
    verify_seal(seal)
    bad = seal.copy()
    bad[4], bad[5] = 64, 4
    with pytest.raises(HalError):
        verify_seal(bad)
    for hdr in ((0, 3), (65, 3), (16, 6), (16, 0)):
        bad = seal.copy()
        bad[4], bad[5] = hdr
        with pytest.raises(HalError, match="header"):
            verify_seal(bad)


@pytest.mark.parametrize("group,col,row", [(1, 5, 17), (1, 2, 0), (1, 0, 1023), (2, 1, 500), (2, 4, 1023), (0, 0, 3), (0, 1, 9), (1, 3, 77)])
def test_a_proof_of_a_false_statement_is_rejected(group, col, row):
    """Soundness, end to end: the oracle proves honestly a witness in which ONE cell is wrong (a derived cell, a free cell a
    constraint reads, a cell of a permuted copy, an accumulator cell, a selector).  Merkle openings, DEEP and FRI are all
    consistent with that witness; only the constraint identity at Z can catch it."""
    L = ol.lib()
    widths = (4, 16, 8)  # F = 8, J = 8, two accumulators = one pair: columns 2 and 3 are a permuted copy of each other
    try:
        L.bxo_set_witness_fault(group, col, row)
        seal, _ = ol.prove_segment(10, *widths, 4321)
    finally:
        L.bxo_set_witness_fault(-1, 0, 0)
    with pytest.raises(HalError, match="constraint identity"):
        verify_seal(seal)
    good, _ = ol.prove_segment(10, *widths, 4321)
    verify_seal(good)


def test_seal_of_another_segment_shape_is_rejected():
    a, _ = ol.prove_segment(10, 4, 8, 4, 1)
    b, _ = ol.prove_segment(10, 4, 8, 4, 2)
    mixed = a.copy()
    mixed[-500:] = b[-500:]
    with pytest.raises(HalError):
        verify_seal(mixed)


def test_a_short_seal_claiming_huge_widths_is_refused_before_any_per_column_work():
    """An ~8 KB seal whose header claims three 65535-column groups used to cost ~1.7 s of tap-set construction before it was
    rejected as truncated (the agent verifies every seal it is handed): the widths are now checked against the words left."""
    import time

    seal, _ = ol.prove_segment(9, 1, 1, 1, 3)
    bad = seal[:2048].copy()
    bad[1] = bad[2] = bad[3] = 65535
    t0 = time.perf_counter()
    with pytest.raises(HalError):
        verify_seal(bad)
    assert time.perf_counter() - t0 < 0.25


def test_zk_noise_rows_are_seeded_and_change_nothing_the_verifier_checks():
    """The last min(1994, N/4) rows of the free data columns are ZK noise drawn from their own seed (upstream: a thread RNG,
    hence its run-to-run different seals).  Same (seed, noise seed) -> same seal; another noise seed -> another seal that
    proves the same statement (same public words) and verifies; the default noise seed is a function of the seed."""
    a, _ = ol.prove_segment(10, 4, 16, 8, 77, noise_seed=1)
    b, _ = ol.prove_segment(10, 4, 16, 8, 77, noise_seed=1)
    c, _ = ol.prove_segment(10, 4, 16, 8, 77, noise_seed=2)
    d, _ = ol.prove_segment(10, 4, 16, 8, 77)
    assert np.array_equal(a, b) and not np.array_equal(a, c) and not np.array_equal(a, d)
    for s in (a, c, d):
        verify_seal(s)
    assert np.array_equal(a[:8], c[:8]) and np.array_equal(a[:8], d[:8])  # header + public words: the statement
    assert np.array_equal(a[8:8 + 256], c[8:8 + 256])  # the code group is public and not blinded: same top layer
    assert not np.array_equal(a[8 + 256:8 + 512], c[8 + 256:8 + 512])  # the data commitment differs


def test_the_verdict_does_not_depend_on_the_number_of_verification_threads():
    """bx_verify_set_threads: the queries of one seal are checked independently on n threads; the error reported is that of the
    first failing query in seal order — the same text a front-to-back read gives (tests/verify_fuzz_check.cpp fuzzes this under
    the sanitizers; here a few hand-made cases through the Python binding)."""
    from boundless_amd.prover import set_verify_threads

    seal, _ = ol.prove_segment(11, 4, 12, 4, 77)
    cases = [seal.copy() for _ in range(5)]
    cases[1][len(seal) - 3] ^= 1             # the last query's last Merkle path
    cases[2][len(seal) // 2] ^= 1            # somewhere in the middle of the queries
    cases[3] = seal[: len(seal) - 40].copy()  # truncated inside the last query
    cases[4] = np.concatenate([seal, seal[:8]])  # trailing words
    verdicts = {}
    try:
        for n in (1, 2, 3, 7, 50, 64):
            set_verify_threads(n)
            row = []
            for c in cases:
                try:
                    verify_seal(c)
                    row.append("")
                except HalError as e:
                    row.append(str(e))
            verdicts[n] = row
    finally:
        set_verify_threads(0)
    assert verdicts[1][0] == "" and all(v for v in verdicts[1][1:])
    assert "trailing words" in verdicts[1][4] and "truncated" in verdicts[1][3]
    for n, row in verdicts.items():
        assert row == verdicts[1], n
    with pytest.raises(HalError, match="0 .default. .. 64"):
        set_verify_threads(65)
