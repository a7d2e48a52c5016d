"""GPU: a short, seeded run of the two fuzzers under tools/ (the long runs are `python tools/fuzz_hal.py`, `tools/fuzz_shapes.py`;
their outputs are under profiles/).  Every `Hal` entry point on operands that are slices at random word offsets with canaries
around them, random sizes; and whole proofs over random shapes through `bx_prove_segment` and the trait-level driver."""
import argparse
import os
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def test_hal_entry_points_on_random_slices_and_sizes():
    import fuzz_hal

    out = fuzz_hal.run(iters=150, seed=20261001)
    assert out["n_failures"] == 0, out["failures"]
    assert len(out["ran"]) == len(fuzz_hal.OPS), out  # every entry point was drawn at least once and none was refused every time


def test_whole_proofs_over_random_shapes():
    import fuzz_shapes

    out = fuzz_shapes.run(argparse.Namespace(iters=10, seed=20261001, min_po2=9, max_po2=13, seconds=0, verbose=False))
    assert out["n_failures"] == 0, out["failures"]
    assert out["shapes_proved_and_verified"] == 10 and out["of_which_also_through_the_plain_driver"] >= 1
