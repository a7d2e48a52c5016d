"""ctypes mirror of include/bx_circuit.h: the circuit half of the segment prover as a plug-in table.

Reference: below `ProverServer::prove_segment` (bento/crates/workflow/src/tasks/prove.rs:41-49) the prover reaches
`risc0_zkp::hal::CircuitHal` (eval_check, accumulate) and the circuit crate's witness generation; `bx_circuit_ops` is that
boundary as a C table.  `CircuitOps.from_object` adapts a Python object with the same method names (used by the tests to plug a
circuit written outside the library into bx_prove_segment / bx_verify_segment); production circuits are native tables.
"""
import ctypes as C

from .hal import BxBuf, load_library
from .prover import SegmentParams


MAX_TAPS = 8  # BX_MAX_TAPS (include/bx_circuit.h)


class TapReader(C.Structure):
    pass


_TAP_AT = C.CFUNCTYPE(C.c_void_p, C.c_void_p, C.c_int, C.c_uint32, C.c_int, C.POINTER(C.c_uint32))
TapReader._fields_ = [("ctx", C.c_void_p), ("at", _TAP_AT)]

_NORMALIZE = C.CFUNCTYPE(C.c_void_p, C.c_void_p, C.POINTER(SegmentParams))
_TAPS = C.CFUNCTYPE(C.c_uint32, C.c_void_p, C.POINTER(SegmentParams), C.c_int, C.c_uint32, C.POINTER(C.c_uint32))
_NGLOBALS = C.CFUNCTYPE(C.c_uint32, C.c_void_p, C.POINTER(SegmentParams))
_CREATE = C.CFUNCTYPE(C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(SegmentParams), C.POINTER(C.c_void_p))
_DESTROY = C.CFUNCTYPE(None, C.c_void_p, C.c_void_p)
_CODE = C.CFUNCTYPE(C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, BxBuf)
_WITGEN = C.CFUNCTYPE(C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, BxBuf, BxBuf, C.POINTER(C.c_uint8), C.c_size_t, BxBuf, C.POINTER(C.c_uint32))
_ACCUM = C.CFUNCTYPE(C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, BxBuf, C.POINTER(C.c_uint32))
_CHECK_CODE = C.CFUNCTYPE(C.c_void_p, C.c_void_p, C.POINTER(SegmentParams), C.POINTER(C.c_uint32))
_EVAL = C.CFUNCTYPE(C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, BxBuf, BxBuf, BxBuf, BxBuf, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32),
                    C.POINTER(C.c_uint32))
_CONS = C.CFUNCTYPE(C.c_void_p, C.c_void_p, C.POINTER(SegmentParams), C.POINTER(TapReader), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32),
                    C.POINTER(C.c_uint32), C.POINTER(C.c_uint32))


class CircuitOps(C.Structure):
    _fields_ = [("user", C.c_void_p), ("name", C.c_char_p), ("normalize", _NORMALIZE), ("taps", _TAPS), ("n_globals", _NGLOBALS), ("create", _CREATE),
                ("destroy", _DESTROY), ("code_group", _CODE), ("witgen", _WITGEN), ("accumulate", _ACCUM), ("eval_check", _EVAL),
                ("constraints_at", _CONS),
                ("set_noise_seed", C.c_void_p),  # optional (include/bx_circuit.h); NULL for circuits written in Python
                ("check_code", _CHECK_CODE)]

    @staticmethod
    def from_object(obj, name=b"python-circuit"):
        """obj provides normalize(shape), taps(shape, group, col) -> list of rows back (first 0), n_globals(shape),
        code_group(ctx, code) (fills the public code group: a function of the shape alone),
        witgen(ctx, code, data, segment_bytes, segment_dev) -> list of the n_globals public words, accumulate(ctx, accum, mix),
        optionally check_code(shape, root_words) (raise to refuse a code root; without it seals of this circuit verify only
        against an explicit VerifierContext),
        eval_check(ctx, check, code_eval, data_eval, accum_eval, poly_mix, mix, globals) and
        constraints_at(shape, tap, poly_mix, mix, globals) -> 4 words; ctx is the raw bx_ctx pointer, buffers are BxBuf, mixes are lists of 4 Montgomery words, tap(group, col, back)
        returns 4 Montgomery words.  Exceptions become the error string of the call."""
        errs = []

        def guard(fn):
            def wrapped(*a):
                try:
                    fn(*a)
                    return None
                except Exception as e:  # noqa: BLE001 - crosses the ABI as a string
                    errs.append(C.create_string_buffer(f"{type(e).__name__}: {e}".encode()))
                    return C.cast(errs[-1], C.c_void_p).value
            return wrapped

        def w4(p):
            return [p[i] for i in range(4)]

        n_glob = {}

        def n_globals(_u, shape):
            n_glob["n"] = int(obj.n_globals(shape.contents)) if hasattr(obj, "n_globals") else 0
            return n_glob["n"]

        def witgen(_u, _s, ctx, code, data, seg, seg_len, seg_dev, globals_out):
            g = list(obj.witgen(ctx, code, data, C.string_at(seg, seg_len), seg_dev) or [])
            # the C side hands over an array of n_globals words: more would be written past it before any check could run
            if len(g) > n_glob.get("n", 0):
                raise ValueError(f"witgen returned {len(g)} public words, the circuit declared {n_glob.get('n', 0)}")
            for i, v in enumerate(g):
                globals_out[i] = int(v)

        def glist(p):
            return [p[i] for i in range(n_glob.get("n", 0))]

        def constraints_at(_u, shape, reader, poly_mix, mix, globals_, out):
            def tap(group, col, back):
                buf = (C.c_uint32 * 4)()
                msg = reader.contents.at(reader.contents.ctx, group, col, back, buf)
                if msg:
                    raise RuntimeError(C.cast(msg, C.c_char_p).value.decode())
                return list(buf)

            r = obj.constraints_at(shape.contents, tap, w4(poly_mix), w4(mix), glist(globals_))
            for i in range(4):
                out[i] = int(r[i])

        def taps(_u, shape, g, c, out):
            backs = list(obj.taps(shape.contents, g, c))  # e.g. [0] or [0, 1, 3]: the rows back the column is opened at
            # `out` is a BX_MAX_TAPS-word array on the caller's stack: report an oversized set by its length (the C side refuses
            # k > BX_MAX_TAPS) without writing past the array
            for i, b in enumerate(backs[:MAX_TAPS]):
                out[i] = int(b)
            return len(backs)

        ops = CircuitOps(None, name,
                         _NORMALIZE(guard(lambda _u, shape: obj.normalize(shape.contents))),
                         _TAPS(taps), _NGLOBALS(n_globals),
                         _CREATE(), _DESTROY(),
                         _CODE(guard(lambda _u, _s, ctx, code: obj.code_group(ctx, code))),
                         _WITGEN(guard(witgen)),
                         _ACCUM(guard(lambda _u, _s, ctx, accum, mix: obj.accumulate(ctx, accum, w4(mix)))),
                         _EVAL(guard(lambda _u, _s, ctx, check, ce, de, ae, pm, mix, gl: obj.eval_check(ctx, check, ce, de, ae, w4(pm), w4(mix), glist(gl)))),
                         _CONS(guard(constraints_at)), None,
                         _CHECK_CODE(guard(lambda _u, shape, root: obj.check_code(shape.contents, [root[i] for i in range(8)])))
                         if hasattr(obj, "check_code") else _CHECK_CODE())
        ops._keepalive = (obj, errs)
        return ops


def synthetic_circuit():
    """Pointer to the library's built-in table (bx_synthetic_circuit)."""
    lib = load_library()
    lib.bx_synthetic_circuit.restype = C.POINTER(CircuitOps)
    return lib.bx_synthetic_circuit()
