"""GPU: a circuit written OUTSIDE the library plugs into bx_prove_segment / bx_verify_segment through include/bx_circuit.h.

`bx_circuit_ops` is the C ABI counterpart of risc0's `CircuitHal` (+ witness generation): the slot where the generated rv32im
kernels would go.  To show that the prover really is circuit-agnostic behind that table, this test implements a small AIR in
numpy (host arithmetic, device buffers moved with bx_h2d / bx_d2h) and proves / verifies it with the library's pipeline:

    data[1][r] = data[0][r]^2 + code[0][r] * data[0][r-1] + data[0][r-3]   (degree 2; data column 0 is opened at rows back {0, 1, 3},
                                                                            a tap set the built-in circuit does not have)

plus one public word g = data[0][0], tied to the trace by the boundary constraint first(r) * (data[0][r] - g) = 0 with the
selector first = code[1] (weight poly_mix, so the check polynomial is genuinely ext-valued).

Everything else — commits, transcript, DEEP, FRI, queries — is the library's.  The seal must verify against this circuit,
must NOT verify against the built-in synthetic circuit, and a witness that violates the constraint must be rejected.

The circuit's code group (a control column and the `first` selector) is public and a function of the shape alone; its committed
root is the circuit's control ID, which the verifier is given in a VerifierContext (or through the table's check_code).  Witness
generation receives the SEGMENT'S BYTES — host copy and HBM copy — as the reference's prover receives a `Segment`
(bento/crates/workflow/src/tasks/prove.rs:41-49): data column 0 is read back from the uploaded payload when there is one.
"""
import ctypes as C

import numpy as np
import pytest

from oracle import oracle_lib as ol

pytestmark = pytest.mark.gpu
P = ol.P
ROU_FWD2 = 284861408  # w_4 (canonical), SURVEY.md App. A.1


def fmul(a, b):
    return (np.asarray(a, np.uint64) * np.asarray(b, np.uint64)) % np.uint64(P)


class Fp4:
    """Canonical-integer arithmetic in Fp[X]/(X^4 + 11) for the verifier half."""

    def __init__(self, c):
        self.c = [int(v) % P for v in c]

    @staticmethod
    def from_mont(words):
        return Fp4(ol.decode(np.array(words, np.uint32)).tolist())

    def to_mont(self):
        return ol.encode(self.c).tolist()

    def __add__(self, o):
        return Fp4([(a + b) % P for a, b in zip(self.c, o.c)])

    def __sub__(self, o):
        return Fp4([(a - b) % P for a, b in zip(self.c, o.c)])

    def __mul__(self, o):
        a, b, nb = self.c, o.c, P - 11
        return Fp4([a[0] * b[0] + nb * (a[1] * b[3] + a[2] * b[2] + a[3] * b[1]),
                    a[0] * b[1] + a[1] * b[0] + nb * (a[2] * b[3] + a[3] * b[2]),
                    a[0] * b[2] + a[1] * b[1] + a[2] * b[0] + nb * (a[3] * b[3]),
                    a[0] * b[3] + a[1] * b[2] + a[2] * b[1] + a[3] * b[0]])


class SquareCircuit:
    def __init__(self, lib, cheat_row=None, claim=None, zero_first=False):
        self.lib = lib
        self.cheat_row = cheat_row
        self.claim = claim  # report this public word instead of the true data[0][0]
        self.zero_first = zero_first  # a DISHONEST code group: the `first` selector is the zero column
        self.calls = []

    # ---- shape
    def normalize(self, shape):
        shape.cons_terms, shape.cons_degree = 1, 2

    def taps(self, shape, group, col):
        return [0, 1, 3] if (group == 1 and col == 0) else [0]

    def n_globals(self, shape):
        return 1

    # ---- device buffers <-> numpy (canonical integers on the host side)
    def _put(self, ctx, buf, canon):
        m = ol.encode(canon)
        assert m.size == buf.len
        msg = self.lib.bx_h2d(ctx, buf, m.ctypes.data, m.size)
        assert not msg, msg

    def _get(self, ctx, buf, offset, n):
        out = np.empty(n, np.uint32)
        sub = type(buf)(buf.dptr + 4 * offset, n)
        msg = self.lib.bx_d2h(ctx, out.ctypes.data, sub, n)
        assert not msg, msg
        return ol.decode(out).astype(np.uint64)

    # ---- prover side
    def _code(self):
        """The code group: public, a function of the shape alone (its committed root is the control ID)."""
        rng = np.random.default_rng(0xC0DE + self.n + 7 * self.wc)
        code_w = rng.integers(0, P, (self.wc, self.n), dtype=np.uint64)
        code_w[1] = 0
        if not self.zero_first:
            code_w[1][0] = 1  # the `first` selector
        return code_w

    def code_group(self, ctx, code):
        self.calls.append("code_group")
        self._put(ctx, code, self._code().reshape(-1))

    def witgen(self, ctx, code, data, segment, segment_dev):
        self.calls.append("witgen")
        n, wd = self.n, self.wd
        assert segment[:8] == b"BXSYNSEG"  # this circuit reuses the stand-in's 28-byte header; the payload is its own
        self.seed = seed = int.from_bytes(segment[20:28], "little")
        rng = np.random.default_rng(seed & 0xFFFFFFFF)
        code_w = self._code()
        data_w = rng.integers(0, P, (wd, n), dtype=np.uint64)
        if len(segment) >= 28 + 4 * n:
            # the "preflight trace": data column 0 comes from the payload — read back from the copy the prover uploaded to HBM
            assert segment_dev.len == (len(segment) + 3) // 4
            words = np.empty(segment_dev.len, np.uint32)
            msg = self.lib.bx_d2h(ctx, words.ctypes.data, segment_dev, words.size)
            assert not msg, msg
            assert words.tobytes()[:len(segment)] == segment  # HBM copy == host copy
            data_w[0] = words[7:7 + n].astype(np.uint64) % np.uint64(P)
        x = data_w[0]
        data_w[1] = (fmul(x, x) + fmul(code_w[0], np.roll(x, 1)) + np.roll(x, 3)) % np.uint64(P)
        if self.cheat_row is not None:
            data_w[1][self.cheat_row] = (data_w[1][self.cheat_row] + np.uint64(1)) % np.uint64(P)
        self._put(ctx, data, data_w.reshape(-1))
        g = int(data_w[0][0]) if self.claim is None else self.claim
        return ol.encode([g]).tolist()  # the public word, as a Montgomery word

    def accumulate(self, ctx, accum, mix):
        self.calls.append("accumulate")
        rng = np.random.default_rng((self.seed ^ mix[0]) & 0xFFFFFFFF)
        self._put(ctx, accum, rng.integers(0, P, self.wa * self.n, dtype=np.uint64))  # unconstrained columns

    def eval_check(self, ctx, check, code_eval, data_eval, accum_eval, poly_mix, mix, globals_):
        self.calls.append("eval_check")
        dom = 4 * self.n
        c0 = self._get(ctx, code_eval, 0, dom)
        first = self._get(ctx, code_eval, dom, dom)
        g = np.uint64(ol.decode(np.array(globals_, np.uint32))[0])
        pm = ol.decode(np.array(poly_mix, np.uint32)).astype(np.uint64)
        d0 = self._get(ctx, data_eval, 0, dom)
        d1 = self._get(ctx, data_eval, dom, dom)
        # one row back = 4 domain points
        cons = (d1 + np.uint64(3 * P) - fmul(d0, d0) - fmul(c0, np.roll(d0, 4)) - np.roll(d0, 12)) % np.uint64(P)
        t3n = pow(3, self.n, P)
        zinv = np.array([pow((t3n * pow(ROU_FWD2, m, P) - 1) % P, -1, P) for m in range(4)], np.uint64)
        bound = fmul(first, (d0 + np.uint64(P) - g) % np.uint64(P))  # first * (data[0] - g), weight poly_mix^1
        zi = zinv[np.arange(dom) % 4]
        planes = np.zeros(4 * dom, np.uint64)
        for k in range(4):
            tot = fmul(bound, pm[k])
            if k == 0:
                tot = (tot + cons) % np.uint64(P)  # weight poly_mix^0 = 1 of the first constraint
            planes[k * dom:(k + 1) * dom] = fmul(tot, zi)
        self._put(ctx, check, planes)

    # ---- verifier side
    def constraints_at(self, shape, tap, poly_mix, mix, globals_):
        d0, d0b, d0b3 = Fp4.from_mont(tap(1, 0, 0)), Fp4.from_mont(tap(1, 0, 1)), Fp4.from_mont(tap(1, 0, 3))
        d1, c0, first = Fp4.from_mont(tap(1, 1, 0)), Fp4.from_mont(tap(0, 0, 0)), Fp4.from_mont(tap(0, 1, 0))
        g = Fp4([int(ol.decode(np.array(globals_, np.uint32))[0]), 0, 0, 0])
        return (d1 - d0 * d0 - c0 * d0b - d0b3 + Fp4.from_mont(poly_mix) * (first * (d0 - g))).to_mont()

    def bind(self, po2, widths):
        self.n, (self.wc, self.wd, self.wa) = 1 << po2, widths


def _prove(lib, circ_obj, po2, widths, seed, payload=b""):
    """-> (receipt, ops table, VerifierContext holding the circuit's control ID as the device computed it)"""
    from boundless_amd.circuit import CircuitOps
    from boundless_amd.prover import HipProverServer, Segment

    circ_obj.bind(po2, widths)
    ops = CircuitOps.from_object(circ_obj, b"square-plus-back")
    srv = HipProverServer(0, po2=po2, widths=widths, circuit=ops)
    try:
        vctx = srv.verifier_context()
        circ_obj.calls.clear()
        return srv.prove_segment(Segment(index=0, po2=po2, seed=seed, payload=payload)), ops, vctx
    finally:
        srv.close()


def test_a_foreign_circuit_is_proved_and_verified_through_the_plugin_table():
    from boundless_amd.hal import HalError, load_library
    from boundless_amd.prover import verify_seal

    lib = load_library()
    po2, widths = 10, (2, 3, 2)
    circ = SquareCircuit(lib)
    receipt, ops, vctx = _prove(lib, circ, po2, widths, seed=77)
    assert circ.calls == ["code_group", "witgen", "accumulate", "eval_check"]
    assert receipt.seal[:6].tolist() == [po2, 2, 3, 2, 1, 2]  # the circuit's own knobs travel in the header
    assert int(receipt.seal[6]) < P  # ... followed by its public word
    verify_seal(receipt.seal, circuit=ops, ctx=vctx)  # accepted against the circuit it was made for and its control ID
    with pytest.raises(HalError, match="no control IDs"):  # this table has no check_code: without a context nothing binds the code group
        verify_seal(receipt.seal, circuit=ops)
    with pytest.raises(HalError):  # ... and it is not a proof of the built-in synthetic circuit
        verify_seal(receipt.seal)
    bad = receipt.seal.copy()
    bad[-1] ^= 1
    with pytest.raises(HalError):
        verify_seal(bad, circuit=ops, ctx=vctx)
    # deterministic: the same segment twice gives the same seal; another segment (seed) leaves the code root == control ID
    again, _, vctx2 = _prove(lib, SquareCircuit(lib), po2, widths, seed=77)
    assert np.array_equal(again.seal, receipt.seal)
    other, _, _ = _prove(lib, SquareCircuit(lib), po2, widths, seed=78)
    assert not np.array_equal(other.seal, receipt.seal) and np.array_equal(other.roots[0], receipt.roots[0])
    verify_seal(other.seal, circuit=ops, ctx=vctx2)


def test_the_segments_bytes_reach_witgen_on_the_host_and_in_hbm():
    """`prove_segment(&ctx, &segment)`: the circuit's witgen is handed the serialized segment — the host copy and the copy the
    prover uploaded on its copy stream.  Here data column 0 IS the payload, read back from HBM, so the public word g = data[0][0]
    is the payload's first word: the seal proves a statement about bytes that travelled through bx_prove_segment_bytes."""
    from boundless_amd.hal import load_library
    from boundless_amd.prover import verify_seal

    lib = load_library()
    po2, widths = 10, (2, 3, 2)
    trace = np.random.default_rng(9).integers(0, P, 1 << po2, dtype=np.uint32)
    circ = SquareCircuit(lib)
    receipt, ops, vctx = _prove(lib, circ, po2, widths, seed=3, payload=trace.tobytes())
    assert int(ol.decode(receipt.seal[6:7])[0]) == int(trace[0])
    verify_seal(receipt.seal, circuit=ops, ctx=vctx)
    plain, _, _ = _prove(lib, SquareCircuit(lib), po2, widths, seed=3)
    assert not np.array_equal(plain.seal, receipt.seal)


def test_a_foreign_circuit_with_a_false_witness_is_rejected():
    from boundless_amd.hal import HalError, load_library
    from boundless_amd.prover import verify_seal

    lib = load_library()
    receipt, ops, vctx = _prove(lib, SquareCircuit(lib, cheat_row=123), 10, (2, 3, 2), seed=5)
    with pytest.raises(HalError, match="constraint identity"):
        verify_seal(receipt.seal, circuit=ops, ctx=vctx)


def test_a_foreign_circuit_cannot_claim_a_public_word_its_trace_does_not_have():
    from boundless_amd.hal import HalError, load_library
    from boundless_amd.prover import verify_seal

    lib = load_library()
    receipt, ops, vctx = _prove(lib, SquareCircuit(lib, claim=12345), 10, (2, 3, 2), seed=5)
    with pytest.raises(HalError, match="constraint identity"):
        verify_seal(receipt.seal, circuit=ops, ctx=vctx)


def test_a_dishonest_code_group_is_refused_by_the_control_id_and_by_nothing_else():
    """VERDICT r03 Weak #2, for a plug-in circuit: a prover that commits its own code group — the `first` selector as the zero
    column — switches the boundary constraint first * (data[0] - g) off and may then claim ANY public word.  Constraint identity,
    Merkle openings, DEEP and FRI all hold for such a seal: it verifies against a context holding the cheater's own "control ID".
    Against the circuit's real control ID (or its check_code) it is refused."""
    from boundless_amd.circuit import CircuitOps
    from boundless_amd.hal import HalError, load_library
    from boundless_amd.prover import verify_seal

    lib = load_library()
    po2, widths = 10, (2, 3, 2)
    honest, ops, honest_ctx = _prove(lib, SquareCircuit(lib), po2, widths, seed=5)
    forged, _, forged_ctx = _prove(lib, SquareCircuit(lib, zero_first=True, claim=12345), po2, widths, seed=5)
    assert int(ol.decode(forged.seal[6:7])[0]) == 12345 != int(ol.decode(honest.seal[6:7])[0])
    verify_seal(forged.seal, circuit=ops, ctx=forged_ctx)  # every other check passes
    with pytest.raises(HalError, match="control ID"):
        verify_seal(forged.seal, circuit=ops, ctx=honest_ctx)
    verify_seal(honest.seal, circuit=ops, ctx=honest_ctx)

    class Checked(SquareCircuit):  # the same circuit publishing its control ID through the table (upstream: check_code)
        def check_code(self, shape, root):
            if list(root) != honest.roots[0].tolist():
                raise ValueError("not the control ID of square-plus-back")

    c2 = Checked(lib)
    c2.bind(po2, widths)
    ops2 = CircuitOps.from_object(c2)
    verify_seal(honest.seal, circuit=ops2)
    with pytest.raises(HalError, match="control ID"):
        verify_seal(forged.seal, circuit=ops2)


def test_errors_of_a_plugged_circuit_surface_as_error_strings():
    from boundless_amd.circuit import CircuitOps
    from boundless_amd.hal import HalError, load_library
    from boundless_amd.prover import HipProverServer, Segment

    class Broken(SquareCircuit):
        def accumulate(self, ctx, accum, mix):
            raise RuntimeError("no accumulate kernel for this shape")

    lib = load_library()
    circ = Broken(lib)
    circ.bind(10, (2, 3, 2))
    ops = CircuitOps.from_object(circ)
    srv = HipProverServer(0, po2=10, widths=(2, 3, 2), circuit=ops)
    try:
        with pytest.raises(HalError, match="no accumulate kernel for this shape"):
            srv.prove_segment(Segment(index=0, po2=10, seed=1))
    finally:
        srv.close()
