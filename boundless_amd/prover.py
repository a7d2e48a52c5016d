"""Host-side mirror of the agent's prover object for the segment-prove path.

Reference: `Agent.prover: Option<Rc<dyn ProverServer>>` (bento/crates/workflow/src/lib.rs:192), built once per
process by `get_prover_server` (lib.rs:246-249) and called as `prover.prove_segment(&verifier_ctx, &segment)`
(bento/crates/workflow/src/tasks/prove.rs:41-49).  `HipProverServer.prove_segment` keeps that shape: one blocking
call per segment, returns the receipt (seal words + claim stub) or raises; all arithmetic happens in
libbx_hip_hal.so (include/bx_prover.h).  No CPU fallback exists.
"""
import ctypes as C
from dataclasses import dataclass

import numpy as np

from .hal import HalError, HipHal, load_library


class SegmentParams(C.Structure):
    _fields_ = [("po2", C.c_uint32), ("w_code", C.c_uint32), ("w_data", C.c_uint32), ("w_accum", C.c_uint32),
                ("cons_terms", C.c_uint32), ("cons_degree", C.c_uint32)]


SEGMENT_MAGIC = b"BXSYNSEG"
SEGMENT_WIRE_BYTES = 28


@dataclass
class Segment:
    """Synthetic stand-in for `risc0_zkvm::Segment` (the executor's output blob, tasks/executor.rs:476-503):
    2^po2 cycles whose witness is derived from `seed` (see include/bx_prover.h).  `payload` stands for the preflight trace: bytes
    that cross PCIe with the segment and reach the circuit's witgen (the built-in circuit does not read them)."""

    index: int
    po2: int = 20
    seed: int = 0xB0D1E550000
    noise_seed: int = None  # generator of the ZK noise rows (None = derived from `seed`, include/bx_prover.h)
    payload: bytes = b""

    @staticmethod
    def synthetic(index, po2=20, base_seed=0xB0D1E550000):
        return Segment(index=index, po2=po2, seed=base_seed + index)

    def to_bytes(self):
        """The segment on the wire (include/bx_prover.h): "BXSYNSEG" | index u64 | po2 u32 | seed u64 | payload."""
        return (SEGMENT_MAGIC + (self.index & (2**64 - 1)).to_bytes(8, "little") + int(self.po2).to_bytes(4, "little")
                + (self.seed & (2**64 - 1)).to_bytes(8, "little") + bytes(self.payload))

    @staticmethod
    def from_bytes(blob):
        if len(blob) < SEGMENT_WIRE_BYTES or blob[:8] != SEGMENT_MAGIC:
            raise HalError("Failed to deserialize segment data: not a synthetic segment blob")
        return Segment(index=int.from_bytes(blob[8:16], "little"), po2=int.from_bytes(blob[16:20], "little"),
                       seed=int.from_bytes(blob[20:28], "little"), payload=bytes(blob[28:]))


@dataclass
class SegmentReceipt:
    """Stand-in for `risc0_zkvm::SegmentReceipt`: the seal (Vec<u32>) plus what identifies the segment."""

    seal: np.ndarray
    index: int
    po2: int
    hashfn: str = "poseidon2"
    roots: np.ndarray = None

    def get_seal_bytes(self):
        return self.seal.tobytes()

    def public_words(self, n=2):
        """The statement's public words ("globals", include/bx_circuit.h) as Montgomery words: they follow the 6-word header.
        The built-in synthetic circuit has two (the first cell of data column 0, the last cell of the last data column)."""
        return self.seal[6:6 + n].copy()

    def verify_integrity(self, ctx=None):
        """`SegmentReceipt::verify_integrity_with_context` (bento/crates/workflow/src/tasks/prove.rs:53-55): CPU check of the
        whole seal (transcript, check identity, Merkle openings, DEEP quotients, FRI chain, code root == control ID).  `ctx` = a
        VerifierContext; None = the built-in circuit's own control IDs.  Raises on rejection."""
        verify_seal(self.seal, ctx=ctx)


class VerifierContext:
    """`risc0_zkvm::VerifierContext` for the segment path (built once per agent, bento/crates/workflow/src/lib.rs:241): the set of
    control IDs — Merkle roots of a circuit's code group, one per segment size — a seal's code root may be (include/bx_circuit.h)."""

    def __init__(self):
        self.lib = load_library()
        _declare(self.lib)
        h = C.c_void_p()
        msg = self.lib.bx_verifier_ctx_create(C.byref(h))
        if msg:
            raise HalError(msg.decode())
        self.handle = h

    def add_control_id(self, po2, digest_words):
        d = np.ascontiguousarray(digest_words, dtype=np.uint32)
        if d.size != 8:
            raise HalError("a control ID is 8 digest words")
        msg = self.lib.bx_verifier_ctx_add_control_id(self.handle, po2, d.ctypes.data)
        if msg:
            raise HalError(msg.decode())
        return self

    def __len__(self):
        return self.lib.bx_verifier_ctx_size(self.handle)

    def close(self):
        if getattr(self, "handle", None):
            self.lib.bx_verifier_ctx_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _declare(lib):
    if getattr(lib, "_bx_prover_declared", False):
        return
    ctx, sz = C.c_void_p, C.c_size_t
    lib.bx_prover_create.argtypes = [ctx, C.POINTER(SegmentParams), C.POINTER(C.c_void_p)]
    lib.bx_prover_create.restype = C.c_char_p
    lib.bx_prover_destroy.argtypes = [C.c_void_p]
    lib.bx_prover_destroy.restype = C.c_char_p
    lib.bx_prover_seal_words.argtypes = [C.c_void_p]
    lib.bx_prover_seal_words.restype = sz
    lib.bx_prove_segment.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, sz, C.POINTER(sz)]
    lib.bx_prove_segment.restype = C.c_char_p
    lib.bx_prove_segment_zk.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p, sz, C.POINTER(sz)]
    lib.bx_prove_segment_zk.restype = C.c_char_p
    lib.bx_prover_last_roots.argtypes = [C.c_void_p, C.c_void_p]
    lib.bx_prover_last_roots.restype = C.c_char_p
    lib.bx_verify_segment.argtypes = [C.c_void_p, sz]
    lib.bx_verify_segment.restype = C.c_char_p
    lib.bx_prover_create_with_circuit.argtypes = [ctx, C.POINTER(SegmentParams), C.c_void_p, C.POINTER(C.c_void_p)]
    lib.bx_prover_create_with_circuit.restype = C.c_char_p
    lib.bx_verify_segment_with_circuit.argtypes = [C.c_void_p, sz, C.c_void_p]
    lib.bx_verify_segment_with_circuit.restype = C.c_char_p
    lib.bx_verify_segment_with_context.argtypes = [C.c_void_p, sz, C.c_void_p, C.c_void_p]
    lib.bx_verify_segment_with_context.restype = C.c_char_p
    lib.bx_verifier_ctx_create.argtypes = [C.POINTER(C.c_void_p)]
    lib.bx_verifier_ctx_create.restype = C.c_char_p
    lib.bx_verifier_ctx_destroy.argtypes = [C.c_void_p]
    lib.bx_verifier_ctx_destroy.restype = None
    lib.bx_verifier_ctx_add_control_id.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p]
    lib.bx_verifier_ctx_add_control_id.restype = C.c_char_p
    lib.bx_verifier_ctx_size.argtypes = [C.c_void_p]
    lib.bx_verifier_ctx_size.restype = sz
    lib.bx_verifier_ctx_count.argtypes = [C.c_void_p, C.c_uint32]
    lib.bx_verifier_ctx_count.restype = sz
    lib.bx_synthetic_control_id_host.argtypes = [C.c_uint32, C.c_uint32, C.c_void_p]
    lib.bx_synthetic_control_id_host.restype = C.c_char_p
    lib.bx_prover_control_id.argtypes = [C.c_void_p, C.c_void_p]
    lib.bx_prover_control_id.restype = C.c_char_p
    lib.bx_prove_segment_bytes.argtypes = [C.c_void_p, C.c_char_p, sz, C.c_void_p, sz, C.POINTER(sz)]
    lib.bx_prove_segment_bytes.restype = C.c_char_p
    lib.bx_prover_submit_segment.argtypes = [C.c_void_p, C.c_char_p, sz]
    lib.bx_prover_submit_segment.restype = C.c_char_p
    lib.bx_prove_submitted.argtypes = [C.c_void_p, C.c_void_p, sz, C.POINTER(sz)]
    lib.bx_prove_submitted.restype = C.c_char_p
    lib.bx_prover_set_noise_seed.argtypes = [C.c_void_p, C.c_uint64]
    lib.bx_prover_set_noise_seed.restype = C.c_char_p
    lib.bx_prover_last_upload.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(sz)]
    lib.bx_prover_last_upload.restype = C.c_char_p
    lib._bx_prover_declared = True


def verify_seal(seal_words, circuit=None, ctx=None):
    """Host-side verifier (include/bx_prover.h: bx_verify_segment); needs no GPU.  `circuit` = a bx_circuit_ops table
    (boundless_amd.circuit.CircuitOps) when the seal was made for another circuit than the built-in synthetic one; `ctx` = a
    VerifierContext holding the control IDs the code root may be (None = the circuit's own check_code)."""
    lib = load_library()
    _declare(lib)
    a = np.ascontiguousarray(seal_words, dtype=np.uint32)
    msg = lib.bx_verify_segment_with_context(a.ctypes.data, a.size, C.addressof(circuit) if circuit is not None else None,
                                             ctx.handle if ctx is not None else None)
    if msg:
        raise HalError(msg.decode())


def set_verify_threads(threads):
    """bx_verify_set_threads: host threads that share the 50 queries of one verification (0 = default: BX_VERIFY_THREADS, else
    min(4, cores); 1 = the calling thread alone).  The verdict does not depend on it."""
    lib = load_library()
    _declare(lib)
    lib.bx_verify_set_threads.restype = C.c_char_p
    lib.bx_verify_set_threads.argtypes = [C.c_int]
    msg = lib.bx_verify_set_threads(int(threads))
    if msg:
        raise HalError(msg.decode())


def synthetic_control_id_host(po2, w_code):
    """The built-in circuit's control ID for (po2, w_code) computed on the host (include/bx_circuit.h); no GPU."""
    lib = load_library()
    _declare(lib)
    out = np.zeros(8, np.uint32)
    msg = lib.bx_synthetic_control_id_host(po2, w_code, out.ctypes.data)
    if msg:
        raise HalError(msg.decode())
    return out


class HipProverServer:
    """`impl ProverServer` for one GPU.  `widths` = (code, data, accum) trace-group widths of the synthetic circuit (include/bx_prover.h)."""

    DEFAULT_WIDTHS = (16, 256, 64)  # SURVEY.md §8d synthetic segment

    def __init__(self, device=0, po2=20, widths=DEFAULT_WIDTHS, hal=None, terms=0, degree=0, circuit=None):
        """terms / degree: the circuit's knobs (synthetic circuit: product terms per constraint, factors per term); 0 = defaults.
        circuit: a bx_circuit_ops table (boundless_amd.circuit.CircuitOps) to prove another circuit than the built-in one."""
        self._own_hal = hal is None  # a context created here is released by close(); a caller's is the caller's
        self.hal = hal or HipHal(device)
        self.lib = load_library()
        _declare(self.lib)
        self.po2 = po2
        self.widths = tuple(widths)
        shape = SegmentParams(po2, *self.widths, terms, degree)
        handle = C.c_void_p()
        self._circuit = circuit  # the table must outlive the prover
        msg = self.lib.bx_prover_create_with_circuit(self.hal.ctx, C.byref(shape), C.addressof(circuit) if circuit is not None else None,
                                                     C.byref(handle))
        if msg:
            raise HalError(msg.decode())
        self.handle = handle
        self._seal = np.empty(self.lib.bx_prover_seal_words(handle), dtype=np.uint32)

    def prove_segment(self, segment, ctx=None):
        """`ProverServer::prove_segment(&self, ctx: &VerifierContext, segment: &Segment) -> Result<SegmentReceipt>`"""
        if segment.po2 != self.po2:
            raise HalError(f"segment po2 {segment.po2} does not match the prover's allocation ({self.po2})")
        n = C.c_size_t(0)
        if segment.noise_seed is not None:
            self.lib.bx_prover_set_noise_seed(self.handle, segment.noise_seed & (2**64 - 1))
        blob = segment.to_bytes()  # what the reference deserializes from the hot store (prove.rs:36-37) and hands to prove_segment
        msg = self.lib.bx_prove_segment_bytes(self.handle, blob, len(blob), self._seal.ctypes.data, self._seal.size, C.byref(n))
        if msg:
            raise HalError(msg.decode())
        roots = np.zeros(32, np.uint32)
        self.lib.bx_prover_last_roots(self.handle, roots.ctypes.data)
        return SegmentReceipt(seal=self._seal[: n.value].copy(), index=segment.index, po2=segment.po2, roots=roots.reshape(4, 8))

    def prove_segment_bytes(self, blob, index=0):
        """The same from the serialized segment (`bincode::deserialize` + prove_segment in one call, prove.rs:36-49)."""
        n = C.c_size_t(0)
        blob = bytes(blob)
        msg = self.lib.bx_prove_segment_bytes(self.handle, blob, len(blob), self._seal.ctypes.data, self._seal.size, C.byref(n))
        if msg:
            raise HalError(msg.decode())
        roots = np.zeros(32, np.uint32)
        self.lib.bx_prover_last_roots(self.handle, roots.ctypes.data)
        return SegmentReceipt(seal=self._seal[: n.value].copy(), index=index, po2=self.po2, roots=roots.reshape(4, 8))

    def prove_segment_buffer(self, buf, index=0):
        """prove_segment_bytes from a writable buffer (bytearray / numpy uint8) without the copy `bytes()` would make: what a
        caller holding an ~80 MB segment in memory does (the C entry point takes a pointer and a length)."""
        n = C.c_size_t(0)
        mv = memoryview(buf)
        arr = (C.c_char * mv.nbytes).from_buffer(buf)
        msg = self.lib.bx_prove_segment_bytes(self.handle, arr, mv.nbytes, self._seal.ctypes.data, self._seal.size, C.byref(n))
        if msg:
            raise HalError(msg.decode())
        return SegmentReceipt(seal=self._seal[: n.value].copy(), index=index, po2=self.po2)

    def submit_segment_buffer(self, buf):
        mv = memoryview(buf)
        arr = (C.c_char * mv.nbytes).from_buffer(buf)
        msg = self.lib.bx_prover_submit_segment(self.handle, arr, mv.nbytes)
        if msg:
            raise HalError(msg.decode())

    def submit_segment(self, blob):
        """Stage the next segment: pinned copy + upload on the copy stream (two deep, include/bx_prover.h)."""
        blob = bytes(blob)
        msg = self.lib.bx_prover_submit_segment(self.handle, blob, len(blob))
        if msg:
            raise HalError(msg.decode())

    def prove_submitted(self, index=0):
        n = C.c_size_t(0)
        msg = self.lib.bx_prove_submitted(self.handle, self._seal.ctypes.data, self._seal.size, C.byref(n))
        if msg:
            raise HalError(msg.decode())
        roots = np.zeros(32, np.uint32)
        self.lib.bx_prover_last_roots(self.handle, roots.ctypes.data)
        return SegmentReceipt(seal=self._seal[: n.value].copy(), index=index, po2=self.po2, roots=roots.reshape(4, 8))

    def last_upload(self):
        """(milliseconds on the copy stream, bytes) of the upload of the segment proved last."""
        ms, nb = C.c_double(0), C.c_size_t(0)
        self.lib.bx_prover_last_upload(self.handle, C.byref(ms), C.byref(nb))
        return ms.value, nb.value

    def control_id(self):
        """The circuit's control ID for this prover's shape, computed on the device (bx_prover_control_id)."""
        out = np.zeros(8, np.uint32)
        msg = self.lib.bx_prover_control_id(self.handle, out.ctypes.data)
        if msg:
            raise HalError(msg.decode())
        return out

    def verifier_context(self):
        """A VerifierContext holding this prover's control ID (what an agent builds at start-up, lib.rs:241)."""
        return VerifierContext().add_control_id(self.po2, self.control_id())

    def close(self):
        if getattr(self, "handle", None):
            self.lib.bx_prover_destroy(self.handle)
            self.handle = None
        if getattr(self, "_own_hal", False):
            self._own_hal = False
            self.hal.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def get_prover_server(device=0, po2=20, widths=HipProverServer.DEFAULT_WIDTHS):
    """Counterpart of `risc0_zkvm::get_prover_server(&ProverOpts::default())` (bento/crates/workflow/src/lib.rs:247)."""
    return HipProverServer(device=device, po2=po2, widths=widths)
