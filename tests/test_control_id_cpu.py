"""CPU: the code group is bound to the circuit — a seal's code root must be the circuit's control ID (risc0's `check_code`).

Reference: `segment_receipt.verify_integrity_with_context(&agent.verifier_ctx)` (bento/crates/workflow/src/tasks/prove.rs:53-55;
`verifier_ctx` at lib.rs:241) compares, inside risc0, the code group's Merkle root with the circuit's control IDs; the generated
table one level up is contracts/src/blake3-groth16/ControlID.sol:13.  VERDICT r03 (Weak #2) showed what happens without it: the
`first` / `last` selectors are tap values of a group the PROVER committed, so a prover that commits `last == 0` satisfies every
constraint for any claimed g_1.  profiles/r04_soundness_at_head_872f07d.log records both forgeries being ACCEPTED by the
verifier of round 3; here they must be refused, for the right reason.

Independent computations of a control ID agree: the oracle's commit_group (recursive NTTs), big-int Python from the definition
(tests/golden/control_ids.json), the library's host path (iterative natural-order DFTs, csrc/control_id.cpp) and its generated
table; the HIP path joins them in test_control_id_gpu.py.
"""
import os
import re

import numpy as np
import pytest

from boundless_amd.hal import HalError
from boundless_amd.prover import VerifierContext, synthetic_control_id_host, verify_seal
from oracle import oracle_lib as ol

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def table_w16():
    text = open(os.path.join(ROOT, "boundless_amd", "csrc", "control_ids_w16.inc")).read()
    return {int(po2): np.array([int(w.strip().rstrip("u"), 16) for w in words.split(",")], np.uint32)
            for po2, words in re.findall(r"\{(\d+), \{([^}]*)\}\}", text)}


def forged_seal(mode, shape=(10, 4, 16, 8), seed=99):
    L = ol.lib()
    L.bxo_set_cheat(mode)
    try:
        seal, roots = ol.prove_segment(*shape, seed)
    finally:
        L.bxo_set_cheat(0)
    return seal, roots


@pytest.mark.parametrize("mode,word", [(1, 7), (2, 6)])
def test_a_dishonest_code_group_no_longer_proves_false_public_words(mode, word):
    """mode 1: `last` == 0 and g_1 + 1;  mode 2: `first` == 0, zero accumulators and g_0 + 1 (oracle/bx_oracle_prover.c,
    bxo_set_cheat).  Accepted at HEAD 872f07d (profiles/r04_soundness_at_head_872f07d.log); refused now — and by the control-ID
    check alone: against a context that holds the forger's own code root the seal passes every other check."""
    shape = (10, 4, 16, 8)
    honest, honest_roots = ol.prove_segment(*shape, 99)
    forged, forged_roots = forged_seal(mode, shape)
    assert int(forged[word]) == (int(honest[word]) + int(ol.encode([1])[0])) % ol.P  # the claim is false by one
    assert not np.array_equal(forged_roots[0], honest_roots[0])
    with pytest.raises(HalError, match="control ID"):
        verify_seal(forged)
    verify_seal(forged, ctx=VerifierContext().add_control_id(10, forged_roots[0]))  # nothing else is wrong with it
    with pytest.raises(HalError, match="control ID"):
        verify_seal(forged, ctx=VerifierContext().add_control_id(10, honest_roots[0]))
    verify_seal(honest)
    verify_seal(honest, ctx=VerifierContext().add_control_id(10, honest_roots[0]))


def test_the_code_root_of_every_honest_seal_is_the_control_id_of_its_shape():
    """The code group depends on (po2, w_code) only — not on the segment, the other widths or the circuit's knobs."""
    a, ra = ol.prove_segment(10, 4, 8, 4, 1)
    b, rb = ol.prove_segment(10, 4, 16, 8, 2, terms=5, degree=3)
    c, rc = ol.prove_segment(10, 5, 8, 4, 1)
    d, rd = ol.prove_segment(11, 4, 8, 4, 1)
    assert np.array_equal(ra[0], rb[0]) and np.array_equal(ra[0], ol.control_id(10, 4))
    assert not np.array_equal(ra[0], rc[0]) and not np.array_equal(ra[0], rd[0])
    assert np.array_equal(rc[0], ol.control_id(10, 5)) and np.array_equal(rd[0], ol.control_id(11, 4))
    for s in (a, b, c, d):
        verify_seal(s)


@pytest.mark.parametrize("po2,w_code", [(9, 1), (9, 2), (10, 3), (11, 16), (12, 5), (13, 24), (14, 16)])
def test_host_control_id_equals_the_oracles(po2, w_code):
    assert np.array_equal(synthetic_control_id_host(po2, w_code), ol.control_id(po2, w_code))


def test_definition_level_golden_control_ids(golden_dir):
    """tests/golden/control_ids.json: the IDs from the mathematical definition in big-int Python (tests/golden/make_golden.py
    control_ids: O(n^2) interpolation, Horner evaluation on the coset, the sponge, the tree — oracle/np_oracle.py).  The C oracle's
    commit_group and the library's host path reproduce them; the HIP prover does in tests/test_control_id_gpu.py."""
    import json

    cases = json.load(open(os.path.join(golden_dir, "control_ids.json")))["cases"]
    assert len(cases) >= 3
    for c in cases:
        want = ol.encode(c["control_id"])
        assert np.array_equal(ol.control_id(c["po2"], c["w_code"]), want), c
        assert np.array_equal(synthetic_control_id_host(c["po2"], c["w_code"]), want), c


def test_the_generated_table_equals_the_oracle_and_the_host_path():
    """control_ids_w16.inc (tools/gen_control_ids.py) for w_code = 16: every entry up to po2 16 against the oracle here (po2 20
    on the GPU box, tests/test_control_id_gpu.py), po2 9..14 also against a fresh host computation."""
    table = table_w16()
    assert sorted(table) == list(range(9, 25))
    for po2 in range(9, 17):
        assert np.array_equal(table[po2], ol.control_id(po2, 16)), po2
    for po2 in range(9, 15):
        assert np.array_equal(table[po2], synthetic_control_id_host(po2, 16)), po2
    assert len({t.tobytes() for t in table.values()}) == 16


def test_verifier_context_semantics():
    """A set of (po2, id): a seal is accepted iff its code root is in the set for ITS po2; an empty context accepts nothing."""
    seal, roots = ol.prove_segment(10, 4, 8, 4, 1234)
    other, other_roots = ol.prove_segment(11, 4, 8, 4, 1234)
    ctx = VerifierContext()
    assert len(ctx) == 0
    with pytest.raises(HalError, match="control ID"):
        verify_seal(seal, ctx=ctx)
    ctx.add_control_id(11, roots[0])  # the right digest under the wrong size
    with pytest.raises(HalError, match="control ID"):
        verify_seal(seal, ctx=ctx)
    ctx.add_control_id(10, roots[0]).add_control_id(10, roots[0]).add_control_id(11, other_roots[0])
    assert len(ctx) == 3  # duplicates are not stored twice
    verify_seal(seal, ctx=ctx)
    verify_seal(other, ctx=ctx)
    with pytest.raises(HalError, match="po2"):
        ctx.add_control_id(8, roots[0])
    with pytest.raises(HalError, match="canonical"):
        ctx.add_control_id(10, np.full(8, 0xFFFFFFFF, np.uint32))
    bad = seal.copy()
    bad[8 + 3] = (int(bad[8 + 3]) + 1) % ol.P  # a word of the code group's top layer: another root
    with pytest.raises(HalError, match="control ID"):
        verify_seal(bad, ctx=ctx)


def test_host_control_id_refuses_shapes_it_cannot_hold():
    with pytest.raises(HalError, match="out of range"):
        synthetic_control_id_host(8, 4)
    with pytest.raises(HalError, match="too large"):
        synthetic_control_id_host(24, 4096)
