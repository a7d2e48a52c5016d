"""Host-side mirror of `risc0_zkp::hal::Hal` over the C ABI of libbx_hip_hal.so (include/bx_hal.h).

The reference's prover reaches its GPU through a `Hal` trait object (risc0-zkp 3.0.3, Cargo.lock:9155) created by
`get_prover_server` (bento/crates/workflow/src/lib.rs:246-249).  `HipHal` keeps that trait's method names, argument
order and error behaviour (`HalError` <-> Rust `Err`), so the parity tests read like upstream's HAL tests.  It is a
thin ctypes binding: no arithmetic happens in Python and there is no CPU fallback — if the HIP library or a GPU is
missing, construction raises.
"""
import ctypes as C
import json
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("BX_HAL_LIB") or os.path.join(_HERE, "lib", "libbx_hip_hal.so")  # BX_HAL_LIB: A/B builds of the same ABI

P = 2013265921
DIGEST_WORDS = 8
EXT_SIZE = 4
FRI_FOLD = 16
INV_RATE = 4
QUERIES = 50
FRI_MIN_DEGREE = 256
CHECK_SIZE = 16


class HalError(RuntimeError):
    """An error string returned across the C ABI (the Rust shim maps it to `anyhow::Error`)."""


class BxBuf(C.Structure):
    _fields_ = [("dptr", C.c_void_p), ("len", C.c_size_t)]


_lib = None


def load_library():
    """dlopen the in-tree HIP library and declare every entry point of include/bx_hal.h / bx_prover.h."""
    global _lib
    if _lib is not None:
        return _lib
    # PyTorch-ROCm wheels bundle their own libamdhip64/libhsa-runtime64 with the same SONAMEs as /opt/rocm's.  Whichever
    # copy is loaded first serves the whole process; if ours (linked against /opt/rocm) comes first, torch's later HIP
    # initialisation finds no devices.  Loading torch first makes the order deterministic (measured on the GPU box).
    if os.environ.get("BX_NO_TORCH") != "1":
        try:
            import torch

            if torch.cuda.is_available():
                torch.cuda.init()
        except ImportError:
            pass
    if not os.path.exists(LIB_PATH):
        raise HalError(f"{LIB_PATH} is missing: run `python -m boundless_amd.build` (hipcc, gfx950). There is no CPU fallback.")
    L = C.CDLL(LIB_PATH)
    ctx, sz, u32p, cp = C.c_void_p, C.c_size_t, C.POINTER(C.c_uint32), C.c_char_p
    sigs = {
        "bx_init": [C.c_int, C.POINTER(C.c_void_p)],
        "bx_free": [ctx],
        "bx_device_name": [ctx, C.c_char_p, sz],
        "bx_set_stream": [ctx, C.c_void_p],
        "bx_alloc": [ctx, sz, C.POINTER(BxBuf)],
        "bx_alloc_zeroed": [ctx, sz, C.POINTER(BxBuf)],
        "bx_alloc_init": [ctx, sz, C.c_uint32, C.POINTER(BxBuf)],
        "bx_release": [ctx, BxBuf],
        "bx_h2d": [ctx, BxBuf, C.c_void_p, sz],
        "bx_d2h": [ctx, C.c_void_p, BxBuf, sz],
        "bx_d2d": [ctx, BxBuf, BxBuf, sz],
        "bx_eltwise_copy_elem_slice": [ctx, BxBuf, C.c_void_p, sz, sz, sz, sz, sz, sz, sz],
        "bx_sync": [ctx],
        "bx_batch_interpolate_ntt": [ctx, BxBuf, sz],
        "bx_batch_evaluate_ntt": [ctx, BxBuf, sz, sz],
        "bx_batch_expand_into_evaluate_ntt": [ctx, BxBuf, BxBuf, sz, sz],
        "bx_batch_bit_reverse": [ctx, BxBuf, sz],
        "bx_zk_shift": [ctx, BxBuf, sz],
        "bx_poseidon2_set_params": [ctx, u32p, u32p],
        "bx_poseidon2_get_params": [ctx, u32p, u32p],
        "bx_hash_rows": [ctx, BxBuf, BxBuf],
        "bx_hash_fold": [ctx, BxBuf, sz, sz],
        "bx_merkle_build": [ctx, BxBuf, BxBuf, sz],
        "bx_merkle_fold": [ctx, BxBuf, sz],
        "bx_fri_fold": [ctx, BxBuf, BxBuf, u32p],
        "bx_fri_fold_dev": [ctx, BxBuf, BxBuf, BxBuf],
        "bx_transcript_step": [ctx, BxBuf, BxBuf, sz, BxBuf, sz],
        "bx_mix_poly_coeffs": [ctx, BxBuf, u32p, u32p, BxBuf, BxBuf, sz, sz],
        "bx_batch_evaluate_any": [ctx, BxBuf, sz, BxBuf, BxBuf, BxBuf],
        "bx_batch_evaluate_any_bitrev": [ctx, BxBuf, sz, BxBuf, BxBuf, BxBuf],
        "bx_batch_bit_reverse_ext": [ctx, BxBuf, sz],
        "bx_batch_evaluate_ptrs": [ctx, BxBuf, BxBuf, sz, BxBuf, BxBuf],
        "bx_batch_interpolate_zk": [ctx, BxBuf, sz],
        "bx_eltwise_add_elem": [ctx, BxBuf, BxBuf, BxBuf],
        "bx_eltwise_copy_elem": [ctx, BxBuf, BxBuf],
        "bx_eltwise_zeroize_elem": [ctx, BxBuf],
        "bx_eltwise_sum_extelem": [ctx, BxBuf, BxBuf],
        "bx_gather_sample": [ctx, BxBuf, BxBuf, sz, sz, sz],
        "bx_poly_divide": [ctx, BxBuf, u32p, BxBuf],
        "bx_eltwise_mul_factor": [ctx, BxBuf, C.c_uint32],
        "bx_poly_divide_batch": [ctx, BxBuf, sz, u32p, BxBuf],
        "bx_prefix_products": [ctx, BxBuf],
        "bx_batch_prefix_products": [ctx, BxBuf, sz],
        "bx_scatter": [ctx, BxBuf, BxBuf, BxBuf, BxBuf],
        "bx_timer_start": [ctx],
        "bx_timer_stop": [ctx, C.POINTER(C.c_float)],
        "bx_profile_enable": [ctx, C.c_int],
        "bx_profile_reset": [ctx],
        "bx_profile_report": [ctx, C.c_char_p, sz],
        "bx_set_tunable": [ctx, C.c_char_p, C.c_long],
    }
    for name, args in sigs.items():
        fn = getattr(L, name)  # AttributeError here = the library does not export a declared symbol
        fn.argtypes = args
        fn.restype = cp
    L.bx_get_stream.argtypes = [ctx]
    L.bx_get_stream.restype = C.c_void_p
    L.bx_trace_enable.argtypes = [C.c_int]
    L.bx_trace_enable.restype = cp
    L.bx_trace_level.argtypes = []
    L.bx_trace_level.restype = C.c_int
    _lib = L
    return L




def trace_enable(level=1):
    """roctx ranges around every HAL entry point and prover stage, for `rocprofv3 --marker-trace` (bx_trace_enable, bx_hal.h):
    0 off, 1 ranges, 2 ranges + a stream sync at the end of every prover stage.  Process-wide; no GPU needed to switch."""
    msg = load_library().bx_trace_enable(int(level))
    if msg:
        raise HalError(msg.decode())


def trace_level():
    return int(load_library().bx_trace_level())


def _words(a):
    a = np.ascontiguousarray(a, dtype=np.uint32)
    return a, a.ctypes.data_as(C.POINTER(C.c_uint32))


class Buffer:
    """Device buffer of u32 words (`Buffer<Elem>`, `Buffer<ExtElem>` AoS, `Buffer<Digest>`, `Buffer<u32>`)."""

    def __init__(self, hal, raw, owned=True):
        self.hal = hal
        self.raw = raw
        self.owned = owned

    def size(self):
        return self.raw.len

    def slice(self, offset, size):
        assert offset + size <= self.raw.len
        return Buffer(self.hal, BxBuf(self.raw.dptr + 4 * offset, size), owned=False)

    def view(self):
        """Blocking device->host copy (upstream `Buffer::view`)."""
        out = np.empty(self.raw.len, dtype=np.uint32)
        self.hal._check(self.hal.lib.bx_d2h(self.hal.ctx, out.ctypes.data, self.raw, self.raw.len))
        return out

    def copy_from(self, host):
        a, _ = _words(host)
        assert a.size <= self.raw.len
        self.hal._check(self.hal.lib.bx_h2d(self.hal.ctx, self.raw, a.ctypes.data, a.size))

    def free(self):
        if self.owned and self.raw.dptr:
            self.hal._check(self.hal.lib.bx_release(self.hal.ctx, self.raw))
            self.raw = BxBuf(None, 0)

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class HipHal:
    """`impl Hal for HipHal` — one instance per GPU (one per agent process in the reference, compose.yml:113)."""

    def __init__(self, device=0):
        self.lib = load_library()
        ctx = C.c_void_p()
        msg = self.lib.bx_init(device, C.byref(ctx))
        if msg:
            raise HalError(msg.decode())
        self.ctx = ctx
        self.device = device
        # BX_TUNABLES="name=value,..." applies bx_set_tunable to every context (A/B runs of the same command)
        for item in filter(None, os.environ.get("BX_TUNABLES", "").split(",")):
            name, _, value = item.partition("=")
            self._check(self.lib.bx_set_tunable(self.ctx, name.strip().encode(), int(value)))

    def close(self):
        if getattr(self, "ctx", None):
            self.lib.bx_free(self.ctx)
            self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, msg):
        if msg:
            raise HalError(msg.decode())

    # ---- allocation ----
    def alloc(self, words):
        raw = BxBuf()
        self._check(self.lib.bx_alloc(self.ctx, words, C.byref(raw)))
        return Buffer(self, raw)

    alloc_elem = alloc
    alloc_u32 = alloc

    def alloc_zeroed(self, words):
        """Hal::alloc_extelem_zeroed / alloc_elem_init(.., 0): cleared on the ctx's stream (eltwise_zeroize_elem is not a clear)."""
        raw = BxBuf()
        self._check(self.lib.bx_alloc_zeroed(self.ctx, words, C.byref(raw)))
        return Buffer(self, raw)

    def alloc_elem_init(self, words, value):
        """Hal::alloc_elem_init: every word holds `value` (e.g. the INVALID marker 0xffffffff), filled on the ctx's stream."""
        raw = BxBuf()
        self._check(self.lib.bx_alloc_init(self.ctx, words, int(value) & 0xFFFFFFFF, C.byref(raw)))
        return Buffer(self, raw)

    def alloc_extelem(self, n):
        return self.alloc(EXT_SIZE * n)

    def alloc_digest(self, n):
        return self.alloc(DIGEST_WORDS * n)

    def copy_from(self, host):
        a, _ = _words(host)
        b = self.alloc(a.size)
        if a.size:
            b.copy_from(a)
        return b

    copy_from_elem = copy_from
    copy_from_u32 = copy_from
    copy_from_extelem = copy_from
    copy_from_digest = copy_from

    def wrap(self, data_ptr, words):
        """Adopt device memory owned by another allocator (e.g. a torch tensor's data_ptr())."""
        return Buffer(self, BxBuf(data_ptr, words), owned=False)

    def sync(self):
        self._check(self.lib.bx_sync(self.ctx))

    def set_stream(self, hip_stream_ptr):
        self._check(self.lib.bx_set_stream(self.ctx, hip_stream_ptr))

    def device_name(self):
        buf = C.create_string_buffer(256)
        self._check(self.lib.bx_device_name(self.ctx, buf, 256))
        return buf.value.decode()

    # ---- Hal trait methods ----
    def batch_interpolate_ntt(self, io, count):
        self._check(self.lib.bx_batch_interpolate_ntt(self.ctx, io.raw, count))

    def batch_evaluate_ntt(self, io, count, expand_bits):
        self._check(self.lib.bx_batch_evaluate_ntt(self.ctx, io.raw, count, expand_bits))

    def batch_expand_into_evaluate_ntt(self, out, inp, count, expand_bits):
        self._check(self.lib.bx_batch_expand_into_evaluate_ntt(self.ctx, out.raw, inp.raw, count, expand_bits))

    def batch_bit_reverse(self, io, count):
        self._check(self.lib.bx_batch_bit_reverse(self.ctx, io.raw, count))

    def zk_shift(self, io, count):
        self._check(self.lib.bx_zk_shift(self.ctx, io.raw, count))

    def hash_rows(self, output, matrix):
        self._check(self.lib.bx_hash_rows(self.ctx, output.raw, matrix.raw))

    def hash_fold(self, io, input_size, output_size):
        self._check(self.lib.bx_hash_fold(self.ctx, io.raw, input_size, output_size))

    def merkle_build(self, nodes, matrix, rows):
        self._check(self.lib.bx_merkle_build(self.ctx, nodes.raw, matrix.raw, rows))

    def fri_fold(self, output, inp, mix):
        _, m = _words(mix)
        self._check(self.lib.bx_fri_fold(self.ctx, output.raw, inp.raw, m))

    def fri_fold_dev(self, output, inp, mix_buf):
        """fri_fold with the challenge in device memory (4 words), e.g. one drawn by `transcript_step`."""
        self._check(self.lib.bx_fri_fold_dev(self.ctx, output.raw, inp.raw, mix_buf.raw))

    def transcript_step(self, state, digests, n_commit, out_ext, n_ext):
        """Poseidon2Rng on the device: commit n_commit digests, then draw n_ext ext challenges into out_ext; state = 25 words."""
        self._check(self.lib.bx_transcript_step(self.ctx, state.raw, digests.raw, n_commit, out_ext.raw, n_ext))

    def mix_poly_coeffs(self, output, mix_start, mix, inp, combos, input_size, count):
        _, ms = _words(mix_start)
        _, m = _words(mix)
        self._check(self.lib.bx_mix_poly_coeffs(self.ctx, output.raw, ms, m, inp.raw, combos.raw, input_size, count))

    def batch_evaluate_any(self, coeffs, poly_count, which, xs, out):
        self._check(self.lib.bx_batch_evaluate_any(self.ctx, coeffs.raw, poly_count, which.raw, xs.raw, out.raw))

    def batch_evaluate_any_bitrev(self, coeffs, poly_count, which, xs, out):
        """Extension: the same over bit-reversed coefficient storage (include/bx_hal.h)."""
        self._check(self.lib.bx_batch_evaluate_any_bitrev(self.ctx, coeffs.raw, poly_count, which.raw, xs.raw, out.raw))

    def batch_evaluate_ptrs(self, poly_ptrs, flags, poly_size, xs, out):
        """Extension: evaluations over several coefficient buffers in one launch set (poly_ptrs: one u64 device address per
        evaluation as two u32 words; flags & 1 = bit-reversed storage)."""
        self._check(self.lib.bx_batch_evaluate_ptrs(self.ctx, poly_ptrs.raw, flags.raw, poly_size, xs.raw, out.raw))

    def batch_interpolate_zk(self, io, count):
        """Extension: batch_interpolate_ntt + zk_shift in one call."""
        self._check(self.lib.bx_batch_interpolate_zk(self.ctx, io.raw, count))

    def batch_bit_reverse_ext(self, io, count):
        """Extension: batch_bit_reverse for AoS Buffer<ExtElem> (16-byte elements)."""
        self._check(self.lib.bx_batch_bit_reverse_ext(self.ctx, io.raw, count))

    def eltwise_add_elem(self, output, a, b):
        self._check(self.lib.bx_eltwise_add_elem(self.ctx, output.raw, a.raw, b.raw))

    def eltwise_copy_elem(self, output, inp):
        self._check(self.lib.bx_eltwise_copy_elem(self.ctx, output.raw, inp.raw))

    def eltwise_copy_elem_slice(self, into, from_host, from_rows, from_cols, from_offset, from_stride, into_offset, into_stride):
        """Hal::eltwise_copy_elem_slice: strided 2-D copy of a HOST slice into a device buffer (into[io + r*is + c] = from[fo + r*fs + c])."""
        a, _ = _words(from_host)
        self._check(self.lib.bx_eltwise_copy_elem_slice(self.ctx, into.raw, a.ctypes.data, a.size, from_rows, from_cols, from_offset, from_stride,
                                                        into_offset, into_stride))

    def eltwise_zeroize_elem(self, io):
        self._check(self.lib.bx_eltwise_zeroize_elem(self.ctx, io.raw))

    def eltwise_sum_extelem(self, output, inp):
        self._check(self.lib.bx_eltwise_sum_extelem(self.ctx, output.raw, inp.raw))

    def eltwise_mul_factor(self, io, factor_mont):
        self._check(self.lib.bx_eltwise_mul_factor(self.ctx, io.raw, int(factor_mont)))

    def gather_sample(self, dst, src, idx, size, stride):
        self._check(self.lib.bx_gather_sample(self.ctx, dst.raw, src.raw, idx, size, stride))

    def poly_divide(self, poly, z, rem_out):
        _, zz = _words(z)
        self._check(self.lib.bx_poly_divide(self.ctx, poly.raw, zz, rem_out.raw))

    def poly_divide_batch(self, polys, count, zs, rems_out):
        """Extension: `count` polynomials back to back, each divided by its own point, in one launch."""
        _, zz = _words(zs)
        self._check(self.lib.bx_poly_divide_batch(self.ctx, polys.raw, count, zz, rems_out.raw))

    def prefix_products(self, io):
        self._check(self.lib.bx_prefix_products(self.ctx, io.raw))

    def batch_prefix_products(self, io, count):
        self._check(self.lib.bx_batch_prefix_products(self.ctx, io.raw, count))

    def scatter(self, into, index, offsets, values):
        self._check(self.lib.bx_scatter(self.ctx, into.raw, index.raw, offsets.raw, values.raw))

    def poseidon2_set_params(self, rc213, diag24):
        _, r = _words(rc213)
        _, d = _words(diag24)
        self._check(self.lib.bx_poseidon2_set_params(self.ctx, r, d))

    def poseidon2_get_params(self):
        rc = np.zeros(213, np.uint32)
        d = np.zeros(24, np.uint32)
        self._check(self.lib.bx_poseidon2_get_params(self.ctx, rc.ctypes.data_as(C.POINTER(C.c_uint32)),
                                                     d.ctypes.data_as(C.POINTER(C.c_uint32))))
        return rc, d

    def get_hash_suite(self):
        self.lib.bx_hash_suite_name.restype = C.c_char_p
        return self.lib.bx_hash_suite_name().decode()

    def has_unified_memory(self):
        self.lib.bx_has_unified_memory.argtypes, self.lib.bx_has_unified_memory.restype = [C.c_void_p], C.c_int
        return bool(self.lib.bx_has_unified_memory(self.ctx))

    # ---- measurement ----
    def timer_start(self):
        self._check(self.lib.bx_timer_start(self.ctx))

    def timer_stop(self):
        ms = C.c_float()
        self._check(self.lib.bx_timer_stop(self.ctx, C.byref(ms)))
        return ms.value

    def profile_enable(self, on=True):
        self._check(self.lib.bx_profile_enable(self.ctx, 1 if on else 0))

    def profile_reset(self):
        self._check(self.lib.bx_profile_reset(self.ctx))

    def profile_report(self):
        buf = C.create_string_buffer(1 << 16)
        self._check(self.lib.bx_profile_report(self.ctx, buf, len(buf)))
        return json.loads(buf.value.decode())

    def set_tunable(self, name, value):
        self._check(self.lib.bx_set_tunable(self.ctx, name.encode(), value))
