"""CPU: the product's host-side verifier (bx_verify_segment) accepts honest seals and rejects tampered ones.

The seals come from the CPU oracle's prover, so this also cross-checks the product's host transcript / Poseidon2 code
(transcript.hpp, used by both the HIP prover and the verifier) against the oracle without a GPU.
"""
import numpy as np
import pytest

from boundless_amd.hal import HalError
from boundless_amd.prover import verify_seal
from oracle import oracle_lib as ol


@pytest.mark.parametrize("po2,widths,seed", [(9, (1, 1, 1), 3), (10, (4, 8, 4), 1234), (12, (3, 17, 5), 99), (13, (2, 9, 6), 5)])
def test_honest_seal_is_accepted(po2, widths, seed):
    seal, _ = ol.prove_segment(po2, *widths, seed)
    verify_seal(seal)


def test_tampering_anywhere_is_rejected():
    seal, _ = ol.prove_segment(10, 4, 8, 4, 1234)
    verify_seal(seal)
    n = seal.size
    rng = np.random.default_rng(0)
    # header, code top layer, data top layer, coeff_u, FRI top, final coefficients, query openings, last word
    taps = 4 + (8 + 2) + (4 + 1) + 16
    offs = {
        "header": 1, "code_top": 4 + 5, "data_top": 4 + 256 + 9, "accum_top": 4 + 512 + 3, "check_top": 4 + 768 + 100,
        "coeff_u": 4 + 1024 + 7, "coeff_u_check": 4 + 1024 + 4 * (taps - 3), "fri_top": 4 + 1024 + 4 * taps + 11,
        "final": 4 + 1024 + 4 * taps + 256 + 5, "query_first": 4 + 1024 + 4 * taps + 256 + 256 + 2, "last": n - 1,
    }
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


def test_seal_of_another_segment_shape_is_rejected():
    a, _ = ol.prove_segment(10, 4, 8, 4, 1)
    b, _ = ol.prove_segment(10, 4, 8, 4, 2)
    mixed = a.copy()
    mixed[-500:] = b[-500:]
    with pytest.raises(HalError):
        verify_seal(mixed)
