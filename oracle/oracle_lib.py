"""ctypes loader for oracle/libbx_oracle.so (the C restatement).  Test infrastructure only."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

u32p = np.ctypeslib.ndpointer(dtype=np.uint32, flags="C_CONTIGUOUS")


def build(force=False, native=False, out=None):
    """Compile the C oracle.  native=True builds a -march=native copy (for the cpu_baseline timing leg)."""
    out = out or os.path.join(_HERE, "libbx_oracle.so")
    srcs = [os.path.join(_HERE, f) for f in ("bx_oracle.c", "bx_oracle_prover.c", "bx_oracle_image.c")]
    deps = srcs + [os.path.join(_HERE, "bx_oracle.h")]
    if not force and os.path.exists(out) and all(os.path.getmtime(out) >= os.path.getmtime(d) for d in deps):
        return out
    cmd = ["gcc", "-O3", "-fopenmp", "-fPIC", "-std=c11", "-shared", "-o", out] + srcs
    if native:
        cmd.insert(2, "-march=native")
    subprocess.run(cmd, check=True, cwd=_HERE)
    return out


def lib(path=None):
    global _LIB
    if _LIB is not None and path is None:
        return _LIB
    p = path or os.path.join(_HERE, "libbx_oracle.so")
    if not os.path.exists(p):
        build()
    L = C.CDLL(p)
    sz = C.c_size_t
    sig = {
        "bxo_init": ([], None),
        "bxo_set_threads": ([C.c_int], None),
        "bxo_get_threads": ([], C.c_int),
        "bxo_fp_encode": ([C.c_uint32], C.c_uint32),
        "bxo_fp_decode": ([C.c_uint32], C.c_uint32),
        "bxo_fp_add": ([C.c_uint32, C.c_uint32], C.c_uint32),
        "bxo_fp_sub": ([C.c_uint32, C.c_uint32], C.c_uint32),
        "bxo_fp_mul": ([C.c_uint32, C.c_uint32], C.c_uint32),
        "bxo_fp_pow": ([C.c_uint32, C.c_uint64], C.c_uint32),
        "bxo_fp_inv": ([C.c_uint32], C.c_uint32),
        "bxo_fp4_mul": ([u32p, u32p, u32p], None),
        "bxo_fp4_inv": ([u32p, u32p], None),
        "bxo_rou_fwd": ([C.c_uint], C.c_uint32),
        "bxo_rou_rev": ([C.c_uint], C.c_uint32),
        "bxo_batch_interpolate_ntt": ([u32p, sz, sz], None),
        "bxo_batch_evaluate_ntt": ([u32p, sz, sz, C.c_uint], None),
        "bxo_batch_expand_into_evaluate_ntt": ([u32p, u32p, sz, sz, C.c_uint], None),
        "bxo_batch_bit_reverse": ([u32p, sz, sz], None),
        "bxo_zk_shift": ([u32p, sz, sz], None),
        "bxo_poseidon2_get_params": ([u32p, u32p], None),
        "bxo_poseidon2_set_params": ([u32p, u32p], None),
        "bxo_poseidon2_mix": ([u32p], None),
        "bxo_hash_elem_slice": ([u32p, u32p, sz, sz], None),
        "bxo_hash_pair": ([u32p, u32p, u32p], None),
        "bxo_hash_rows": ([u32p, u32p, sz, sz], None),
        "bxo_hash_fold": ([u32p, sz, sz], None),
        "bxo_fri_fold": ([u32p, u32p, u32p, sz], None),
        "bxo_mix_poly_coeffs": ([u32p, u32p, u32p, u32p, u32p, sz, sz], None),
        "bxo_batch_evaluate_any": ([u32p, sz, u32p, u32p, u32p, sz], None),
        "bxo_eltwise_add": ([u32p, u32p, u32p, sz], None),
        "bxo_eltwise_sum_extelem": ([u32p, u32p, sz, sz], None),
        "bxo_eltwise_zeroize": ([u32p, sz], None),
        "bxo_gather_sample": ([u32p, u32p, sz, sz, sz], None),
        "bxo_poly_divide": ([u32p, sz, u32p, u32p], C.c_int),
        "bxo_prefix_products": ([u32p, sz], None),
        "bxo_scatter": ([u32p, u32p, u32p, u32p, sz], None),
        "bxo_prove_segment": ([C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint64, C.POINTER(sz), u32p], C.c_void_p),
        "bxo_prove_segment_ex": ([C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint64, C.POINTER(sz), u32p], C.c_void_p),
        "bxo_prove_segment_zk": ([C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint64, C.c_uint64, C.POINTER(sz), u32p], C.c_void_p),
        "bxo_transcript_step": ([u32p, u32p, sz, u32p, sz], None),
        "bxo_rng_random_bits": ([u32p, C.c_uint], C.c_uint32),
        "bxo_set_witness_fault": ([C.c_int, C.c_uint32, C.c_uint32], None),
        "bxo_set_cheat": ([C.c_int], None),
        "bxo_control_id": ([C.c_uint32, C.c_uint32, u32p], None),
        "bxo_free": ([C.c_void_p], None),
        "bxo_compute_image_id": ([C.c_char_p, sz, C.c_char_p, u32p], C.c_int),
        "bxo_sha256": ([C.c_char_p, C.c_char_p, sz], None),
    }
    for name, (args, res) in sig.items():
        fn = getattr(L, name)
        fn.argtypes = args
        fn.restype = res
    L.bxo_init()
    # The oracle's OpenMP regions are short; on many-core hosts (the GPU box has 128 hardware threads) fork/join overhead
    # dominates small problems (a 2^12 proof took 15 s with 128 threads vs 0.3 s with 8).  Default to at most 16 threads;
    # bench.py's cpu_baseline leg picks its own count explicitly.
    L.bxo_set_threads(min(os.cpu_count() or 1, int(os.environ.get("BXO_THREADS", "16"))))
    if path is None:
        _LIB = L
    return L


P = 2013265921


def encode(x):
    """canonical ints (array-like) -> Montgomery u32 array"""
    a = np.asarray(x, dtype=np.uint64) % np.uint64(P)
    return ((a << np.uint64(32)) % np.uint64(P)).astype(np.uint32)


def decode(m):
    """Montgomery u32 array -> canonical (numpy uint32)."""
    a = np.asarray(m, dtype=np.uint64)
    rinv = np.uint64(pow(1 << 32, -1, P))
    return ((a * rinv) % np.uint64(P)).astype(np.uint32)


def random_elems(rng, shape):
    """uniform field elements as Montgomery words (any word < P is a valid Montgomery element)"""
    return rng.integers(0, P, size=shape, dtype=np.uint32)


def prove_segment(po2, w_code, w_data, w_accum, seed, L=None, terms=0, degree=0, noise_seed=None):
    """Run the oracle's segment prover; returns (seal as uint32 array, roots[4][8]).  terms/degree = the synthetic
    circuit's knobs (0 = defaults); noise_seed = the ZK rows' generator (None = derived from seed)."""
    L = L or lib()
    n = C.c_size_t(0)
    roots = np.zeros(32, np.uint32)
    if noise_seed is None:
        ptr = L.bxo_prove_segment_ex(po2, w_code, w_data, w_accum, terms, degree, seed, C.byref(n), roots)
    else:
        ptr = L.bxo_prove_segment_zk(po2, w_code, w_data, w_accum, terms, degree, seed, noise_seed, C.byref(n), roots)
    if not ptr:
        raise RuntimeError("oracle prover: DEEP remainder non-zero")
    seal = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint32)), shape=(n.value,)).copy()
    L.bxo_free(ptr)
    return seal, roots.reshape(4, 8)


def control_id(po2, w_code, L=None):
    """The synthetic circuit's control ID for (po2, w_code): 8 Montgomery digest words."""
    L = L or lib()
    out = np.zeros(8, np.uint32)
    L.bxo_control_id(po2, w_code, out)
    return out


def transcript_step(state25, digests, n_elems, L=None):
    """Poseidon2Rng: commit every digest (rows of 8 words), then draw n_elems field elements -> (new state, elements)."""
    L = L or lib()
    st = np.ascontiguousarray(state25, dtype=np.uint32).copy()
    dg = np.ascontiguousarray(digests, dtype=np.uint32).reshape(-1)
    out = np.zeros(max(n_elems, 1), np.uint32)
    L.bxo_transcript_step(st, dg if dg.size else np.zeros(8, np.uint32), dg.size // 8, out, n_elems)
    return st, out[:n_elems]


def compute_image_id(blob, L=None):
    """risc0_zkvm::compute_image_id of an R0BF program binary -> (32-byte id, canonical Merkle root words)."""
    L = L or lib()
    out = C.create_string_buffer(32)
    root = np.zeros(8, np.uint32)
    rc = L.bxo_compute_image_id(bytes(blob), len(blob), out, root)
    if rc != 0:
        raise ValueError(f"oracle compute_image_id: malformed program binary (code {rc})")
    return out.raw, root
