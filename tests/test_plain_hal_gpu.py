"""One WHOLE proof through nothing but the plain `Hal`-trait entry points, sequenced from outside the library.

The boundary north_star names is `risc0_zkp::hal::Hal`; the in-library prover (csrc/prover.hip) reaches
the kernels through extension entry points (bx_batch_interpolate_zk, bx_merkle_build, bx_batch_evaluate_ptrs, ...).  A Rust
`impl Hal for HipHal` shim is driven by risc0-zkp's own prover, which calls the trait methods one by one
(reached from bento/crates/workflow/src/tasks/prove.rs:41-49 via lib.rs:246-249).  tests/plain_hal_prover.c restates that call
order in C over the section-8(b2) list only — one bx_hash_fold per layer, bx_poly_divide per combo and point, bx_batch_evaluate_any
per group, separate bx_zk_shift, bx_gather_sample for the openings — with its own host transcript, and these tests require

    seal(plain driver) == seal(bx_prove_segment) == seal(CPU oracle)        word for word.
"""
import numpy as np
import pytest

from oracle import oracle_lib as ol

pytestmark = pytest.mark.gpu


def _product_seal(po2, widths, seed, terms=0, degree=0):
    from boundless_amd.prover import HipProverServer, Segment

    srv = HipProverServer(0, po2=po2, widths=widths, terms=terms, degree=degree)
    try:
        r = srv.prove_segment(Segment(index=0, po2=po2, seed=seed))
        r.verify_integrity()
        return r.seal
    finally:
        srv.close()


@pytest.mark.parametrize("po2,widths", [(12, (4, 16, 8)), (16, (16, 32, 8)), (9, (2, 6, 4)), (13, (3, 9, 4))])
def test_plain_hal_calls_compose_into_the_same_seal(po2, widths):
    import plain_hal

    seed = 0xB0D1E550000 + po2
    pp = plain_hal.PlainHalProver(0, po2=po2, widths=widths)
    try:
        seal, _ms = pp.prove(seed)
        again, _ = pp.prove(seed)  # buffers are reused: a second proof on the same driver is the same seal
        other, _ = pp.prove(seed + 1)
    finally:
        pp.close()
    want, _ = ol.prove_segment(po2, *widths, seed)
    assert seal.size == want.size
    bad = np.nonzero(seal != want)[0]
    assert bad.size == 0, f"plain-Hal seal differs from the oracle's in {bad.size} words, first at {bad[:5]}"
    assert np.array_equal(again, seal) and not np.array_equal(other, seal)
    assert np.array_equal(_product_seal(po2, widths, seed), seal), "plain-Hal seal differs from bx_prove_segment's"
    assert np.array_equal(other, ol.prove_segment(po2, *widths, seed + 1)[0])


@pytest.mark.parametrize("flag", [1, 2, 4, 8, 16, 32, 63])
def test_each_extension_entry_point_swapped_in_alone_gives_the_same_seal(flag):
    """The driver's `flags` replace one plain call sequence each by the extension entry point the in-library prover uses (what
    INTEGRATION.md section 1 prices): every single swap, and all of them together, leave the seal unchanged."""
    import plain_hal

    po2, widths, seed = 15, (4, 24, 8), 77  # 2^15: the smallest size the bit-reversed-coefficient entry points accept
    base = plain_hal.PlainHalProver(0, po2=po2, widths=widths, flags=0)
    try:
        want, _ = base.prove(seed)
        plain_calls = base.calls
    finally:
        base.close()
    pp = plain_hal.PlainHalProver(0, po2=po2, widths=widths, flags=flag)
    try:
        seal, _ = pp.prove(seed)
        assert pp.calls < plain_calls or flag == 4  # an extension is fewer calls (keeping coefficients bit-reversed: as many)
    finally:
        pp.close()
    assert np.array_equal(seal, want), plain_hal.EXT_NAMES.get(flag, "all extensions")
    if flag == 63:
        assert np.array_equal(seal, ol.prove_segment(po2, *widths, seed)[0])


def test_plain_hal_nondefault_circuit_knobs():
    import plain_hal

    po2, widths, seed = 11, (4, 12, 8), 5
    pp = plain_hal.PlainHalProver(0, po2=po2, widths=widths, terms=3, degree=2)
    try:
        seal, _ = pp.prove(seed)
    finally:
        pp.close()
    assert np.array_equal(seal, ol.prove_segment(po2, *widths, seed, terms=3, degree=2)[0])


@pytest.mark.fullsize
def test_plain_hal_seal_at_the_baseline_config():
    """BASELINE.json configs[1] — 2^20 cycles, widths 16/256/64 — through the plain entry points alone: the seal bx_prove_segment
    writes and the one the CPU oracle writes."""
    import os

    import plain_hal

    po2, widths, seed = 20, (16, 256, 64), 0xB0D1E550000
    pp = plain_hal.PlainHalProver(0, po2=po2, widths=widths)
    try:
        seal, ms = pp.prove(seed)
        calls = pp.calls
        _, ms2 = pp.prove(seed)
    finally:
        pp.close()
    got = _product_seal(po2, widths, seed)
    assert seal.size == got.size and np.array_equal(seal, got), "plain-Hal seal differs from bx_prove_segment's at 2^20"
    L = ol.lib()
    old = L.bxo_get_threads()
    L.bxo_set_threads(min(os.cpu_count() or 1, 32))
    try:
        want, _ = ol.prove_segment(po2, *widths, seed, L)
    finally:
        L.bxo_set_threads(old)
    bad = np.nonzero(seal != want)[0]
    assert bad.size == 0, f"{bad.size} differing seal words vs the oracle, first at {bad[:5]}"
    print(f"plain-Hal proof at 2^20: {ms:.1f} ms cold, {ms2:.1f} ms warm, {calls} entry-point calls")


def test_three_plain_hal_drivers_in_flight_stay_bit_exact():
    """Three trait-level drivers (three ctxs, three host threads — what a process holding three `HipHal` objects does) prove
    different segments concurrently, each with its own queue of deferred gathers: every seal is the oracle's."""
    import threading

    import plain_hal

    po2, widths = 12, (4, 12, 4)
    drivers = [plain_hal.PlainHalProver(0, po2=po2, widths=widths) for _ in range(3)]
    got = {}

    def work(k):
        for j in range(4):
            seed = 1000 + 4 * k + j
            got[seed] = drivers[k].prove(seed)[0]

    try:
        ts = [threading.Thread(target=work, args=(k,)) for k in range(3)]
        [t.start() for t in ts]
        [t.join() for t in ts]
    finally:
        for d in drivers:
            d.close()
    assert sorted(got) == list(range(1000, 1012))
    for seed, seal in got.items():
        assert np.array_equal(seal, ol.prove_segment(po2, *widths, seed)[0]), seed


@pytest.mark.parametrize("po2,widths,cache_mb", [(12, (4, 16, 8), None), (16, (16, 32, 8), None), (16, (16, 32, 8), 0)])
def test_plain_hal_with_every_buffer_allocated_inside_the_proof(po2, widths, cache_mb, monkeypatch):
    """risc0-zkp's prover allocates its buffers inside a proof and drops them at its end.  The driver does the same with
    ALLOC_PER_PROOF: bx_alloc / bx_release are then on the proof's path — served by the library's per-ctx pool (a released block is
    handed to the next request of its size, no driver call, no device-wide wait), or by hipMalloc / hipFree with alloc_cache_mb = 0.
    Recycled memory is dirty, so this is also the test that nothing in the sequence relies on fresh zeros: three proofs, same seal."""
    import plain_hal

    if cache_mb is not None:
        monkeypatch.setenv("BX_TUNABLES", f"alloc_cache_mb={cache_mb}")
    seed = 4242 + po2
    pp = plain_hal.PlainHalProver(0, po2=po2, widths=widths, flags=plain_hal.ALLOC_PER_PROOF)
    try:
        seals = [pp.prove(seed)[0] for _ in range(3)]
        other = pp.prove(seed + 1)[0]
    finally:
        pp.close()
    want = ol.prove_segment(po2, *widths, seed)[0]
    for s in seals:
        assert np.array_equal(s, want)
    assert np.array_equal(other, ol.prove_segment(po2, *widths, seed + 1)[0])
