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


def test_transcript_rng_step_against_the_definition_level_permutation():
    """`bxo_transcript_step` (the oracle's Poseidon2Rng, the checker of the device-side `bx_transcript_step`) against the big-int
    permutation of np_oracle: commit = (permute first if any rate cell was handed out) add the digest into cells 0..7, permute;
    draws hand out cells 0..15 in order and permute when they run out."""
    from oracle import np_oracle as npo

    rng = np.random.default_rng(5)
    cells = [0] * 24  # canonical values; the C side holds Montgomery words
    used = 0
    state = np.zeros(25, np.uint32)
    for n_commit, n_elems in [(1, 4), (2, 4), (0, 12), (1, 0), (1, 20), (0, 16), (3, 1)]:
        digests = rng.integers(0, ol.P, (max(n_commit, 1), 8), dtype=np.uint32)
        state, got = ol.transcript_step(state, ol.encode(digests[:n_commit]) if n_commit else digests[:0], n_elems)
        want = []
        for d in digests[:n_commit].tolist():
            if used:
                cells, used = npo.poseidon2_permute(cells), 0
            cells = npo.poseidon2_permute([(c + x) % ol.P for c, x in zip(cells[:8], d)] + cells[8:])
        for _ in range(n_elems):
            if used == 16:
                cells, used = npo.poseidon2_permute(cells), 0
            want.append(cells[used])
            used += 1
        assert ol.decode(got).tolist() == want
        assert ol.decode(state[:24]).tolist() == cells and int(state[24]) == used
