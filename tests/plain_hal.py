"""ctypes loader of tests/plain_hal_prover.c — the test-side driver that sequences one whole proof through the plain `Hal`
entry points of SURVEY.md section 8(b2).  Plain C on the library's C ABI: built with gcc (no hipcc), linked against the in-tree
libbx_hip_hal.so.  Used by tests/test_plain_hal_gpu.py and by bench.py's untimed `single_proof_ms.plain_hal` extra.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
SRC = os.path.join(HERE, "plain_hal_prover.c")
OUT = os.path.join(HERE, "_build", "libplain_hal.so")

EXT_INTERPOLATE_ZK, EXT_MERKLE_BUILD, EXT_COEFFS_BITREV, EXT_DIVIDE_BATCH, EXT_QUERY_GATHER, EXT_EVAL_PTRS = 1, 2, 4, 8, 16, 32
EXT_ALL = 63
ALLOC_PER_PROOF = 64  # not an extension: the big buffers are allocated and released inside every proof, as risc0-zkp's prover does
EXT_NAMES = {EXT_INTERPOLATE_ZK: "bx_batch_interpolate_zk", EXT_MERKLE_BUILD: "bx_merkle_build",
             EXT_COEFFS_BITREV: "bx_batch_evaluate_any_bitrev+bx_batch_bit_reverse_ext", EXT_DIVIDE_BATCH: "bx_poly_divide_batch_indexed",
             EXT_QUERY_GATHER: "bx_merkle_query_gather", EXT_EVAL_PTRS: "bx_batch_evaluate_ptrs"}


class _Params(C.Structure):
    _fields_ = [("po2", C.c_uint32), ("w_code", C.c_uint32), ("w_data", C.c_uint32), ("w_accum", C.c_uint32), ("cons_terms", C.c_uint32),
                ("cons_degree", C.c_uint32)]


def build(force=False):
    from boundless_amd.hal import LIB_PATH

    libdir = os.path.dirname(LIB_PATH)
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    if not force and os.path.exists(OUT) and os.path.getmtime(OUT) >= max(os.path.getmtime(SRC), os.path.getmtime(LIB_PATH)):
        return OUT
    cmd = ["gcc", "-std=c99", "-O2", "-Wall", "-Wextra", "-shared", "-fPIC", f"-I{os.path.join(ROOT, 'include')}", SRC, f"-L{libdir}", "-lbx_hip_hal",
           "-Wl,-rpath,$ORIGIN/../../boundless_amd/lib", "-Wl,-rpath-link,/opt/rocm/lib", "-o", OUT]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("building tests/plain_hal_prover.c failed:\n" + r.stdout + r.stderr)
    return OUT


_lib = None


def lib():
    global _lib
    if _lib is None:
        from boundless_amd.hal import load_library

        load_library()  # the product library first (same soname, already mapped)
        L = C.CDLL(build())
        L.ph_create.restype = C.c_char_p
        L.ph_create.argtypes = [C.c_void_p, C.c_int, C.POINTER(_Params), C.c_uint, C.POINTER(C.c_void_p)]
        L.ph_prove.restype = C.c_char_p
        L.ph_prove.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t), C.POINTER(C.c_double)]
        L.ph_destroy.restype = C.c_char_p
        L.ph_destroy.argtypes = [C.c_void_p]
        L.ph_seal_words.restype = C.c_size_t
        L.ph_seal_words.argtypes = [C.c_void_p]
        L.ph_last_calls.restype = C.c_size_t
        L.ph_last_calls.argtypes = [C.c_void_p]
        _lib = L
    return _lib


class PlainHalProver:
    """One prover of the driver: every device buffer allocated up front, `prove(seed)` -> (seal words, wall ms)."""

    def __init__(self, device=0, po2=20, widths=(16, 256, 64), terms=0, degree=0, flags=0):
        self.L = lib()
        self.h = C.c_void_p()
        prm = _Params(po2, widths[0], widths[1], widths[2], terms, degree)
        e = self.L.ph_create(None, device, C.byref(prm), flags, C.byref(self.h))
        if e:
            raise RuntimeError(e.decode())
        self.cap = self.L.ph_seal_words(self.h)
        self.calls = 0

    def prove(self, seed):
        seal = np.zeros(self.cap, np.uint32)
        n, ms = C.c_size_t(0), C.c_double(0)
        e = self.L.ph_prove(self.h, seed, seal.ctypes.data, self.cap, C.byref(n), C.byref(ms))
        if e:
            raise RuntimeError(e.decode())
        self.calls = self.L.ph_last_calls(self.h)
        return seal[: n.value].copy(), ms.value

    def close(self):
        if self.h:
            self.L.ph_destroy(self.h)
            self.h = C.c_void_p()
