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


@dataclass
class Segment:
    """Synthetic stand-in for `risc0_zkvm::Segment` (the executor's output blob, tasks/executor.rs:476-503):
    2^po2 cycles whose witness is derived from `seed` (see include/bx_prover.h)."""

    index: int
    po2: int = 20
    seed: int = 0xB0D1E550000
    noise_seed: int = None  # generator of the ZK noise rows (None = derived from `seed`, include/bx_prover.h)

    @staticmethod
    def synthetic(index, po2=20, base_seed=0xB0D1E550000):
        return Segment(index=index, po2=po2, seed=base_seed + index)


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

    def verify_integrity(self):
        """`SegmentReceipt::verify_integrity_with_context` (bento/crates/workflow/src/tasks/prove.rs:53-55): CPU check of the
        whole seal (transcript, check identity, Merkle openings, DEEP quotients, FRI chain).  Raises on rejection."""
        verify_seal(self.seal)


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
    lib._bx_prover_declared = True


def verify_seal(seal_words, circuit=None):
    """Host-side verifier (include/bx_prover.h: bx_verify_segment); needs no GPU.  `circuit` = a bx_circuit_ops table
    (boundless_amd.circuit.CircuitOps) when the seal was made for another circuit than the built-in synthetic one."""
    lib = load_library()
    _declare(lib)
    a = np.ascontiguousarray(seal_words, dtype=np.uint32)
    msg = lib.bx_verify_segment_with_circuit(a.ctypes.data, a.size, C.addressof(circuit) if circuit is not None else None)
    if msg:
        raise HalError(msg.decode())


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
        if segment.noise_seed is None:
            msg = self.lib.bx_prove_segment(self.handle, segment.seed & (2**64 - 1), self._seal.ctypes.data, self._seal.size, C.byref(n))
        else:
            msg = self.lib.bx_prove_segment_zk(self.handle, segment.seed & (2**64 - 1), segment.noise_seed & (2**64 - 1),
                                               self._seal.ctypes.data, self._seal.size, C.byref(n))
        if msg:
            raise HalError(msg.decode())
        roots = np.zeros(32, np.uint32)
        self.lib.bx_prover_last_roots(self.handle, roots.ctypes.data)
        return SegmentReceipt(seal=self._seal[: n.value].copy(), index=segment.index, po2=segment.po2, roots=roots.reshape(4, 8))

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
