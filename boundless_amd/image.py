"""Host-side mirror of `risc0_binfmt::{ProgramBinary, MemoryImage}` / `risc0_zkvm::compute_image_id` over the C ABI of
include/bx_image.h.

The reference calls `compute_image_id(blob)` wherever it admits a guest program (crates/risc0-backend/src/lib.rs:590-718,
crates/executor/src/api.rs:178) and tests it on its own data (crates/povw/src/log_updater.rs:383-388).  `MemoryImage` keeps
the upstream method names (`set_page`, `get_page`, `image_id`); page hashing and the 22 tree levels run on the GPU through the
same Poseidon2 kernels as the prover's Merkle trees.  ctypes marshalling only: no arithmetic in Python, no CPU fallback.
"""
import ctypes as C

import numpy as np

from .hal import BxBuf, HalError, HipHal, load_library

PAGE_WORDS = 256
PAGE_BYTES = 1024
MERKLE_TREE_DEPTH = 22

_declared = False


def _lib():
    global _declared
    L = load_library()
    if not _declared:
        ctx, sz, u32p, cp, img = C.c_void_p, C.c_size_t, C.POINTER(C.c_uint32), C.c_char_p, C.c_void_p
        sigs = {
            "bx_image_from_program": ([ctx, C.c_char_p, sz, C.POINTER(img)], cp),
            "bx_image_new": ([C.POINTER(img)], cp),
            "bx_image_set_page": ([img, C.c_uint32, u32p], cp),
            "bx_image_get_page": ([img, C.c_uint32, u32p], cp),
            "bx_image_page_count": ([img], sz),
            "bx_image_set_digest": ([img, C.c_uint32, u32p], cp),
            "bx_image_digest_count": ([img], sz),
            "bx_image_node_digest": ([ctx, img, C.c_uint32, u32p], cp),
            "bx_image_page_indices": ([img, u32p, sz], sz),
            "bx_image_free": ([img], None),
            "bx_image_root": ([ctx, img, u32p], cp),
            "bx_system_state_digest": ([u32p, C.c_uint32, C.c_char_p], None),
            "bx_compute_image_id": ([ctx, C.c_char_p, sz, C.c_char_p], cp),
            "bx_hash_fold_indexed": ([ctx, BxBuf, BxBuf, BxBuf, sz], cp),
            "bx_image_page_cells": ([ctx, BxBuf, BxBuf, sz], cp),
        }
        for name, (args, res) in sigs.items():
            fn = getattr(L, name)
            fn.argtypes = args
            fn.restype = res
        _declared = True
    return L


def _check(msg):
    if msg:
        raise HalError(msg.decode())


def system_state_digest(root_canonical, pc=0):
    """`SystemState { pc, merkle_root }.digest()` (host SHA-256 of the tagged struct)."""
    root = np.ascontiguousarray(root_canonical, dtype=np.uint32)
    assert root.size == 8
    out = C.create_string_buffer(32)
    _lib().bx_system_state_digest(root.ctypes.data_as(C.POINTER(C.c_uint32)), int(pc), out)
    return out.raw


class MemoryImage:
    """Sparse 4 GiB zkVM memory image of 1 KiB pages (`risc0_binfmt::MemoryImage`)."""

    def __init__(self, handle=None):
        L = _lib()
        if handle is None:
            handle = C.c_void_p()
            _check(L.bx_image_new(C.byref(handle)))
        self._h = handle

    @classmethod
    def from_program(cls, blob):
        """`ProgramBinary::decode(blob)?.to_image()`: an "R0BF" blob holding the user and the kernel ELF.  Host only."""
        L = _lib()
        h = C.c_void_p()
        blob = bytes(blob)
        _check(L.bx_image_from_program(None, blob, len(blob), C.byref(h)))
        return cls(h)

    def close(self):
        if self._h:
            _lib().bx_image_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __len__(self):
        return int(_lib().bx_image_page_count(self._h))

    def page_indices(self):
        n = len(self)
        out = np.zeros(max(n, 1), np.uint32)
        _lib().bx_image_page_indices(self._h, out.ctypes.data_as(C.POINTER(C.c_uint32)), n)
        return out[:n]

    def set_page(self, page_idx, words):
        w = np.ascontiguousarray(words, dtype=np.uint32)
        if w.size != PAGE_WORDS:
            raise ValueError("a page is 256 words")
        _check(_lib().bx_image_set_page(self._h, int(page_idx), w.ctypes.data_as(C.POINTER(C.c_uint32))))

    def get_page(self, page_idx):
        out = np.zeros(PAGE_WORDS, np.uint32)
        _check(_lib().bx_image_get_page(self._h, int(page_idx), out.ctypes.data_as(C.POINTER(C.c_uint32))))
        return out

    def set_digest(self, node_idx, digest_canonical):
        """Partial image: the digest (8 canonical words) of a subtree the segment does not touch; root = node 1, page p = node
        2^22 + p."""
        d = np.ascontiguousarray(digest_canonical, dtype=np.uint32)
        if d.size != 8:
            raise ValueError("a digest is 8 words")
        _check(_lib().bx_image_set_digest(self._h, int(node_idx), d.ctypes.data_as(C.POINTER(C.c_uint32))))

    def node_digest(self, hal: HipHal, node_idx):
        """Digest of any node (8 canonical words), computed on `hal`'s GPU from the pages and digests below it."""
        out = np.zeros(8, np.uint32)
        _check(_lib().bx_image_node_digest(hal.ctx, self._h, int(node_idx), out.ctypes.data_as(C.POINTER(C.c_uint32))))
        return out

    def root(self, hal: HipHal):
        """Poseidon2 Merkle root of the 2^22-leaf page tree, 8 canonical words, computed on `hal`'s GPU."""
        out = np.zeros(8, np.uint32)
        _check(_lib().bx_image_root(hal.ctx, self._h, out.ctypes.data_as(C.POINTER(C.c_uint32))))
        return out

    def image_id(self, hal: HipHal):
        """`MemoryImage::image_id`: digest of `SystemState { pc: 0, merkle_root }`, 32 bytes."""
        return system_state_digest(self.root(hal), 0)


def compute_image_id(blob, hal: HipHal):
    """`risc0_zkvm::compute_image_id(blob)` on `hal`'s GPU -> 32 bytes (the `.iid` file contents)."""
    blob = bytes(blob)
    out = C.create_string_buffer(32)
    _check(_lib().bx_compute_image_id(hal.ctx, blob, len(blob), out))
    return out.raw
