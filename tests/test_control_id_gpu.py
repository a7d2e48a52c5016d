"""GPU: control IDs from the HIP prover, and the bytes-in boundary of bx_prove_segment.

bx_prover_control_id runs the circuit's code group through the kernels of a proof (witgen_code -> inverse NTT + zk shift -> 4x LDE ->
hash_rows -> Merkle tree) and returns the root.  It must equal the CPU oracle's (oracle/bx_oracle_prover.c: bxo_control_id), the
library's definition-level host computation and the generated table the verifier consults — three implementations that share no
transform code with the device path.  Reference: risc0's control IDs / check_code behind
`verify_integrity_with_context` (bento/crates/workflow/src/tasks/prove.rs:53-55, lib.rs:241).
"""
import threading

import numpy as np
import pytest

from oracle import oracle_lib as ol

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hal():
    from boundless_amd.hal import HipHal

    h = HipHal(0)
    yield h
    h.close()


def _server(hal, po2, widths, **kw):
    from boundless_amd.prover import HipProverServer

    return HipProverServer(0, po2=po2, widths=widths, hal=hal, **kw)


@pytest.mark.parametrize("po2,widths", [(10, (16, 8, 4)), (16, (16, 8, 4)), (20, (16, 2, 4)), (9, (1, 1, 1)), (12, (3, 17, 5)), (13, (24, 9, 6))])
def test_device_control_id_equals_the_oracles(hal, po2, widths):
    srv = _server(hal, po2, widths)
    try:
        cid = srv.control_id()
        assert np.array_equal(cid, ol.control_id(po2, widths[0]))
        # and it IS the code root of a proof of that shape, whatever the segment
        from boundless_amd.prover import Segment

        r = srv.prove_segment(Segment(index=0, po2=po2, seed=41))
        assert np.array_equal(r.roots[0], cid)
        assert np.array_equal(srv.control_id(), cid)  # idempotent, also right after a proof
        r.verify_integrity()
        r.verify_integrity(ctx=srv.verifier_context())
    finally:
        srv.close()


def test_device_control_ids_equal_the_definition_level_golden(hal, golden_dir):
    import json
    import os

    for c in json.load(open(os.path.join(golden_dir, "control_ids.json")))["cases"]:
        srv = _server(hal, c["po2"], (c["w_code"], 2, 4))
        try:
            assert np.array_equal(srv.control_id(), ol.encode(c["control_id"])), c
        finally:
            srv.close()


def test_device_control_ids_equal_the_generated_table_up_to_the_largest_segment(hal):
    """w_code = 16, po2 9..24: the table bx_verify_segment consults (csrc/control_ids_w16.inc, generated on the host) against the
    device, entry by entry.  The other groups are one column wide here: the control ID depends on (po2, w_code) only."""
    import os
    import re

    text = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "boundless_amd", "csrc", "control_ids_w16.inc")).read()
    table = {int(po2): np.array([int(w.strip().rstrip("u"), 16) for w in words.split(",")], np.uint32)
             for po2, words in re.findall(r"\{(\d+), \{([^}]*)\}\}", text)}
    assert sorted(table) == list(range(9, 25))
    for po2 in range(9, 25):
        srv = _server(hal, po2, (16, 1, 1))
        try:
            assert np.array_equal(srv.control_id(), table[po2]), po2
        finally:
            srv.close()


def test_the_payload_crosses_pcie_and_leaves_the_synthetic_seal_alone(hal):
    """bx_prove_segment_bytes: "BXSYNSEG" header + payload.  The built-in circuit's witness is a function of the seed, so the seal
    is the oracle's whatever the payload; the upload (pinned staging slot -> copy stream -> HBM) is timed by its own events."""
    from boundless_amd.prover import Segment

    po2, widths = 12, (4, 8, 4)
    srv = _server(hal, po2, widths)
    try:
        want, _ = ol.prove_segment(po2, *widths, 77)
        plain = srv.prove_segment(Segment(index=3, po2=po2, seed=77))
        assert np.array_equal(plain.seal, want)
        ms0, nb0 = srv.last_upload()
        assert nb0 == 28
        payload = np.random.default_rng(1).integers(0, 256, 8_000_001, dtype=np.uint8).tobytes()  # not a multiple of 4
        big = srv.prove_segment(Segment(index=3, po2=po2, seed=77, payload=payload))
        assert np.array_equal(big.seal, want)
        ms, nb = srv.last_upload()
        assert nb == 28 + len(payload) and ms > 0
        assert nb / (ms * 1e-3) > 1e9  # a DMA from pinned memory, not a pageable crawl (PCIe gen5 x16: ~50 GB/s)
        # a smaller one afterwards reuses the grown slot
        again = srv.prove_segment_bytes(Segment(index=0, po2=po2, seed=77).to_bytes())
        assert np.array_equal(again.seal, want) and srv.last_upload()[1] == 28
    finally:
        srv.close()


def test_malformed_segment_bytes_are_errors_not_crashes(hal):
    from boundless_amd.hal import HalError
    from boundless_amd.prover import Segment

    srv = _server(hal, 10, (4, 8, 4))
    try:
        good = Segment(index=0, po2=10, seed=5).to_bytes()
        with pytest.raises(HalError, match="Failed to deserialize segment data"):
            srv.prove_segment_bytes(good[:27])
        with pytest.raises(HalError, match="not a synthetic segment blob"):
            srv.prove_segment_bytes(b"\x00" * 64)  # e.g. a real bincode(Segment)
        with pytest.raises(HalError, match="po2 11"):
            srv.prove_segment_bytes(Segment(index=0, po2=11, seed=5).to_bytes())
        with pytest.raises(HalError, match="empty segment"):
            srv.prove_segment_bytes(b"")
        with pytest.raises(HalError, match="no segment was submitted"):
            srv.prove_submitted()
        want, _ = ol.prove_segment(10, 4, 8, 4, 5)
        assert np.array_equal(srv.prove_segment_bytes(good).seal, want)  # the prover is fine afterwards
    finally:
        srv.close()


def test_two_deep_staging_uploads_the_next_segment_while_this_one_is_proved(hal):
    """bx_prover_submit_segment from a feeder thread while bx_prove_submitted runs; segments come out in
    submission order, each seal the oracle's, and a third outstanding segment is refused."""
    from boundless_amd.hal import HalError
    from boundless_amd.prover import Segment

    po2, widths = 12, (4, 8, 4)
    srv = _server(hal, po2, widths)
    try:
        seeds = [100 + i for i in range(6)]
        want = [ol.prove_segment(po2, *widths, s)[0] for s in seeds]
        pad = bytes(1 << 20)
        blobs = [Segment(index=i, po2=po2, seed=s, payload=pad).to_bytes() for i, s in enumerate(seeds)]
        srv.submit_segment(blobs[0])
        srv.submit_segment(blobs[1])
        with pytest.raises(HalError, match="staging slots busy"):
            srv.submit_segment(blobs[2])
        with pytest.raises(HalError, match="still outstanding"):
            srv.prove_segment_bytes(blobs[2])
        got, errors = [], []
        slots = threading.Semaphore(0)  # a slot frees each time a proof returns

        def feeder():
            try:
                for b in blobs[2:]:
                    slots.acquire()
                    srv.submit_segment(b)
            except Exception as e:  # noqa: BLE001
                errors.append(e)

        t = threading.Thread(target=feeder)
        t.start()
        for i in range(len(blobs)):
            got.append(srv.prove_submitted(index=i).seal)
            slots.release()
        t.join()
        assert not errors
        for g, w in zip(got, want):
            assert np.array_equal(g, w)
    finally:
        srv.close()


def test_the_native_agent_verifies_against_its_own_verifier_context(hal):
    """The HIP prover's agent adds the control ID of every buffer set it creates to its VerifierContext (checked against the
    circuit's published IDs) and verifies every seal against it; segments with a payload go to the prover as stored."""
    from boundless_amd import agent as ag
    from boundless_amd.prover import Segment

    a = ag.Agent(prover=None, device=0, inflight=2, widths=(4, 8, 4), poll_time=0.01)
    try:
        segs = [Segment(index=i, po2=10 + (i % 2), seed=900 + i, payload=bytes(1000 * i)) for i in range(6)]
        for s in segs:
            a.store.set_key_with_expiry(f"job:cid:segments:{s.index}", ag.serialize_segment(s), 600)
            a.taskdb.create_task("cid", f"p{s.index}", {"Prove": {"index": s.index}}, max_retries=0)
        assert a.poll_work(max_idle_polls=3) == len(segs)
        for s in segs:
            rec = ag.deserialize_receipt(a.store.get(f"job:cid:synthetic_receipts:p{s.index}"))
            want, _ = ol.prove_segment(s.po2, 4, 8, 4, s.seed)
            assert np.array_equal(rec.seal, want)
    finally:
        a.close()
