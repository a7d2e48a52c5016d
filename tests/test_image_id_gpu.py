"""GPU: the reference's own image-ID vector through the HIP Poseidon2 kernels (C ABI of include/bx_image.h and bx_hal.h).

crates/povw/src/log_updater.rs:383-388 asserts compute_image_id(LOG_UPDATER_ELF) == LOG_UPDATER_ID on
crates/povw/elfs/boundless-povw-log-updater.{bin,iid} (copied as data to tests/golden/reference/).  The ID is a SHA-256 over
the Poseidon2 Merkle root of the program's memory image, so reproducing it pins hash_rows' sponge kernel and the pair-hash
kernels — the ones the prover's Merkle commitments run on — to the reference byte for byte."""
import os

import numpy as np
import pytest

from oracle import oracle_lib as ol

pytestmark = pytest.mark.gpu
REF = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference")


def blob():
    return open(os.path.join(REF, "boundless-povw-log-updater.bin"), "rb").read()


def iid():
    return open(os.path.join(REF, "boundless-povw-log-updater.iid"), "rb").read()


@pytest.fixture(scope="module")
def hal():
    from boundless_amd.hal import HipHal

    h = HipHal(0)
    yield h
    h.close()


def test_compute_image_id_on_the_gpu_equals_the_reference_vector(hal):
    from boundless_amd import image

    assert image.compute_image_id(blob(), hal) == iid()
    im = image.MemoryImage.from_program(blob())
    root = im.root(hal)
    _, oracle_root = ol.compute_image_id(blob())
    assert np.array_equal(root, oracle_root)
    assert im.image_id(hal) == iid()


def test_the_vector_through_plain_hash_rows_and_hash_fold(hal):
    """The same ID with nothing but the two Hal entry points the prover's Merkle trees use: bx_hash_rows over the pages as a
    (pages x 512) column-major matrix and bx_hash_fold level by level; the host only gathers (right, left) pairs and converts
    canonical <-> Montgomery."""
    from boundless_amd import image

    im = image.MemoryImage.from_program(blob())
    idx = [int(i) for i in im.page_indices()]
    pages = np.stack([im.get_page(i) for i in idx] + [np.zeros(256, np.uint32)])  # + the zero page
    n1 = len(pages)
    cells = np.empty((n1, 512), np.uint64)
    cells[:, 0::2] = pages & 0xFFFF
    cells[:, 1::2] = pages >> 16
    matrix = hal.copy_from(ol.encode(cells).T.reshape(-1))  # column-major: column c = cell c of every page
    leaves = hal.alloc_digest(n1)
    hal.hash_rows(leaves, matrix)
    dig = leaves.view().reshape(n1, 8)
    level = {i: dig[k] for k, i in enumerate(idx)}
    zero = dig[-1]
    for _ in range(22):
        parents = sorted({i >> 1 for i in level})
        m = len(parents) + 1
        # hash_fold(io, input_size = 2m, output_size = m): io[m + j] = H(io[2m + 2j] | io[2m + 2j + 1])
        io = np.zeros((4 * m, 8), np.uint32)
        for j, p in enumerate(parents):
            io[2 * m + 2 * j] = level.get(2 * p + 1, zero)  # right child first (DigestPair::digest)
            io[2 * m + 2 * j + 1] = level.get(2 * p, zero)
        io[2 * m + 2 * (m - 1)] = zero
        io[2 * m + 2 * (m - 1) + 1] = zero
        buf = hal.copy_from(io.reshape(-1))
        hal.hash_fold(buf, 2 * m, m)
        out = buf.view().reshape(4 * m, 8)[m : 2 * m]
        level = {p: out[j].copy() for j, p in enumerate(parents)}
        zero = out[m - 1].copy()
    root = ol.decode(level[0])
    assert image.system_state_digest(root, 0) == iid()


def test_indexed_fold_and_page_cells_against_the_oracle(hal, oracle):
    from boundless_amd import image
    from boundless_amd.hal import BxBuf  # noqa: F401

    L = image._lib()
    rng = np.random.default_rng(11)
    for n, count in ((1, 1), (37, 300), (4096, 5000)):
        digs = ol.random_elems(rng, (n, 8))
        sel = rng.integers(0, n, (count, 2), dtype=np.uint32)
        d_in, d_sel, d_out = hal.copy_from(digs.reshape(-1)), hal.copy_from(sel.reshape(-1)), hal.alloc_digest(count)
        hal._check(L.bx_hash_fold_indexed(hal.ctx, d_out.raw, d_in.raw, d_sel.raw, count))
        got = d_out.view().reshape(count, 8)
        for j in rng.choice(count, min(count, 64), replace=False):
            want = np.zeros(8, np.uint32)
            oracle.bxo_hash_pair(want, np.ascontiguousarray(digs[sel[j, 0]]), np.ascontiguousarray(digs[sel[j, 1]]))
            assert np.array_equal(got[j], want), (n, j)
    # a sel entry beyond the pool is clamped to the last digest by the kernel (never an out-of-bounds read), and `out` may be a range of
    # the pool `in` names as long as no sel entry points into it — how the image tree builds its levels
    digs = ol.random_elems(rng, (10, 8))
    pool = hal.copy_from(np.concatenate([digs.reshape(-1), np.zeros(16, np.uint32)]))  # 10 digests + room for 2 outputs
    sel = np.array([[1, 2], [9, 0xFFFFFFF0]], np.uint32)
    BxB = type(pool.raw)
    out_view = BxB(pool.raw.dptr + 4 * 80, 16)
    in_view = BxB(pool.raw.dptr, 80)
    hal._check(L.bx_hash_fold_indexed(hal.ctx, out_view, in_view, hal.copy_from(sel.reshape(-1)).raw, 2))
    got = pool.view()[80:96].reshape(2, 8)
    for j, (a, b) in enumerate(((1, 2), (9, 9))):
        want = np.zeros(8, np.uint32)
        oracle.bxo_hash_pair(want, np.ascontiguousarray(digs[a]), np.ascontiguousarray(digs[b]))
        assert np.array_equal(got[j], want)
    for n in (1, 63, 64, 65, 300):
        raw = rng.integers(0, 1 << 32, (n, 256), dtype=np.uint32)
        d_raw, d_mat = hal.copy_from(raw.reshape(-1)), hal.alloc(n * 512)
        hal._check(L.bx_image_page_cells(hal.ctx, d_mat.raw, d_raw.raw, n))
        cells = np.empty((n, 512), np.uint64)
        cells[:, 0::2] = raw & 0xFFFF
        cells[:, 1::2] = raw >> 16
        assert np.array_equal(d_mat.view().reshape(512, n), ol.encode(cells).T), n


def test_an_empty_image_and_a_dirtied_page(hal, oracle):
    """MemoryImage::set_page + image_id: the all-zero image has the zero-subtree root; rewriting one page changes the root the
    way the oracle's tree says (a sparse Merkle update, what happens to a segment's image after execution)."""
    from boundless_amd import image

    im = image.MemoryImage()
    z = np.zeros(8, np.uint32)
    cells = ol.encode(np.zeros(512, np.uint64))
    oracle.bxo_hash_elem_slice(z, cells, 512, 1)
    for _ in range(22):
        nz = np.zeros(8, np.uint32)
        oracle.bxo_hash_pair(nz, z, z)
        z = nz
    assert np.array_equal(im.root(hal), ol.decode(z))
    im2 = image.MemoryImage.from_program(blob())
    r0 = im2.root(hal)
    page = im2.get_page(2050)
    page[7] ^= 0x10000
    im2.set_page(2050, page)
    r1 = im2.root(hal)
    assert not np.array_equal(r0, r1)
    page[7] ^= 0x10000
    im2.set_page(2050, page)
    assert np.array_equal(im2.root(hal), r0)


def test_partial_images_pruned_subtrees_as_digests_give_the_same_root(hal):
    """`Segment.partial_image`: the pages a segment touches plus the digests of the subtrees it does not.  Prune random subtrees
    of the reference program's image into their digests (bx_image_node_digest), drop their pages, and the root — hence the image
    ID the reference pins — must not change; a digest that overlaps pages, or a wrong digest, must not go unnoticed."""
    from boundless_amd import image
    from boundless_amd.hal import HalError

    full = image.MemoryImage.from_program(blob())
    root = full.root(hal)
    assert image.system_state_digest(root, 0) == iid()
    idx = [int(i) for i in full.page_indices()]
    rng = np.random.default_rng(7)
    for trial in range(6):
        part = image.MemoryImage()
        keep = set(rng.choice(idx, size=int(rng.integers(1, 40)), replace=False).tolist()) if trial else set()
        # walk down from the root: a subtree without kept pages becomes one digest, the rest is descended into
        stack, pruned = [1], 0
        while stack:
            node = stack.pop()
            lvl = 22 - (node.bit_length() - 1)
            lo, hi = (node << lvl) - (1 << 22), ((node + 1) << lvl) - (1 << 22)  # page range below the node
            if not any(lo <= p < hi for p in keep):
                if any(lo <= p < hi for p in idx) or rng.random() < 0.3:  # zero subtrees may be given or left implicit
                    part.set_digest(node, full.node_digest(hal, node))
                    pruned += 1
                continue
            if lvl == 0:
                part.set_page(node - (1 << 22), full.get_page(node - (1 << 22)))
            else:
                stack += [2 * node, 2 * node + 1]
        assert len(part) == len(keep) and pruned > 0
        assert np.array_equal(part.root(hal), root), trial
        assert part.image_id(hal) == iid()
    # a page inside a subtree that is already given by its digest: refused
    part.set_page(idx[0], full.get_page(idx[0]))
    part.set_digest(2, full.node_digest(hal, 2))
    part.set_digest(3, full.node_digest(hal, 3))
    with pytest.raises(HalError, match="both by its digest"):
        part.root(hal)
    # a wrong digest changes the root (nothing is silently recomputed)
    two = image.MemoryImage()
    two.set_digest(2, full.node_digest(hal, 2))
    d3 = full.node_digest(hal, 3)
    two.set_digest(3, d3)
    assert np.array_equal(two.root(hal), root)
    d3[0] = (int(d3[0]) + 1) % ol.P
    two.set_digest(3, d3)
    assert not np.array_equal(two.root(hal), root)
    with pytest.raises(HalError):
        two.set_digest(3, np.full(8, ol.P, np.uint32))  # not a field element
    with pytest.raises(HalError, match="only given by its digest"):
        two.node_digest(hal, 7)  # inside node 3, which is only a digest here
    # node indices outside the tree are errors, not a hang: node 0 does not exist (0 << k never reaches the leaf layer — the level
    # search used to spin on it), and nothing lies beyond the 2^22 leaves
    for bad in (0, 2 << 22, 0xFFFFFFFF):
        with pytest.raises(HalError, match="outside the tree"):
            full.node_digest(hal, bad)
    assert np.array_equal(full.node_digest(hal, 1), root)
