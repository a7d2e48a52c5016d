"""CPU: the oracle's segment prover is self-consistent (DEEP quotients exact, deterministic, seal layout)."""
import numpy as np

from oracle import oracle_lib as ol


def test_oracle_prover_runs_and_is_deterministic():
    a, ra = ol.prove_segment(10, 4, 8, 4, 1234)
    b, rb = ol.prove_segment(10, 4, 8, 4, 1234)
    c, _ = ol.prove_segment(10, 4, 8, 4, 1235)
    assert np.array_equal(a, b) and np.array_equal(ra, rb)
    assert not np.array_equal(a, c)
    assert a[:6].tolist() == [10, 4, 8, 4, 64, 4]  # po2, widths, the circuit's default knobs (terms, degree)
    # layout: header 6 | 2 public words | 4 trace tops (32 digests each) | coeff_u | fri tops | final coeffs | 50 queries
    n = 1 << 10
    taps = 4 + (8 + 1 + 2) + (4 + 4) + 16  # data column 0 and the accumulator's columns also one row back, data column 4 one and two
    rows_fri = 4 * n // 16
    per_query = sum(w + 8 * (12 - 5) for w in (4, 8, 4, 16)) + (64 + 8 * (8 - 5))
    expect = 6 + 2 + 4 * 32 * 8 + 4 * taps + 32 * 8 + 4 * (n // 16) + 50 * per_query
    assert rows_fri == 256 and a.size == expect


def test_oracle_prover_threads_do_not_change_the_seal():
    L = ol.lib()
    L.bxo_set_threads(1)
    a, _ = ol.prove_segment(9, 2, 3, 2, 9)
    L.bxo_set_threads(0)
    b, _ = ol.prove_segment(9, 2, 3, 2, 9)
    assert np.array_equal(a, b)


def test_circuit_knobs_change_the_seal_and_are_part_of_the_header():
    a, _ = ol.prove_segment(10, 4, 8, 4, 7)
    b, _ = ol.prove_segment(10, 4, 8, 4, 7, terms=64, degree=4)  # the defaults, spelled out
    c, _ = ol.prove_segment(10, 4, 8, 4, 7, terms=5, degree=4)
    assert np.array_equal(a, b)
    assert c[:6].tolist() == [10, 4, 8, 4, 5, 4] and not np.array_equal(a[8:], c[8:])
