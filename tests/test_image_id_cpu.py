"""CPU: the reference's own image-ID vector (crates/povw/src/log_updater.rs:383-388: compute_image_id(LOG_UPDATER_ELF) == LOG_UPDATER_ID
on crates/povw/elfs/boundless-povw-log-updater.{bin,iid}, copied as data to tests/golden/reference/).

The 32 bytes depend on every Poseidon2 parameter, on the rate-16 overwrite sponge (32 blocks per page) and on the pair hash, so
this is the test that pins the oracle's Poseidon2 to the reference — not to a recollection of it."""
import hashlib
import os
import struct

import numpy as np
import pytest

from oracle import oracle_lib as ol

REF = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference")


def blob():
    return open(os.path.join(REF, "boundless-povw-log-updater.bin"), "rb").read()


def iid(name="boundless-povw-log-updater"):
    return open(os.path.join(REF, name + ".iid"), "rb").read()


def test_fixture_files_are_the_reference_s(oracle):
    m = __import__("json").load(open(os.path.join(os.path.dirname(REF), "MANIFEST.json")))["reference_held"]
    for name, sha in m["files"].items():
        assert hashlib.sha256(open(os.path.join(REF, name), "rb").read()).hexdigest() == sha, name


def test_oracle_compute_image_id_equals_the_reference_vector(oracle):
    got, root = ol.compute_image_id(blob(), oracle)
    assert got == iid(), f"{got.hex()} != {iid().hex()}"
    assert (root < ol.P).all()


def test_oracle_sha256_matches_hashlib(oracle):
    import ctypes as C

    rng = np.random.default_rng(5)
    for n in (0, 1, 55, 56, 63, 64, 65, 119, 120, 1000):
        msg = rng.integers(0, 256, n, dtype=np.uint8).tobytes()
        out = C.create_string_buffer(32)
        oracle.bxo_sha256(out, msg, n)
        assert out.raw == hashlib.sha256(msg).digest(), n


def test_every_bit_of_the_vector_matters(oracle):
    """The match is not an accident of a weak check: one flipped bit in the program, in a Poseidon2 round constant or in the
    internal diagonal changes the ID."""
    b = bytearray(blob())
    b[40_000] ^= 1  # inside the user ELF's text segment
    assert ol.compute_image_id(bytes(b), oracle)[0] != iid()
    rc = np.zeros(213, np.uint32)
    dg = np.zeros(24, np.uint32)
    oracle.bxo_poseidon2_get_params(rc, dg)
    try:
        for table, i in ((rc, 0), (rc, 100), (rc, 212), (dg, 0), (dg, 23)):
            saved = table[i]
            table[i] = (int(saved) + 1) % ol.P
            oracle.bxo_poseidon2_set_params(rc, dg)
            assert ol.compute_image_id(blob(), oracle)[0] != iid(), (i,)
            table[i] = saved
    finally:
        oracle.bxo_poseidon2_set_params(rc, dg)
    assert ol.compute_image_id(blob(), oracle)[0] == iid()


def test_malformed_program_binaries_are_refused(oracle):
    b = blob()
    for bad in (b"", b"R0BF", b"XXXX" + b[4:], b[:100], b[:4] + struct.pack("<I", 2) + b[8:]):
        with pytest.raises(ValueError):
            ol.compute_image_id(bad, oracle)


def test_product_host_half_agrees_with_the_oracle():
    """Without a GPU: the product's ProgramBinary/ELF loader builds the same pages as the oracle's image would need, and its
    SystemState digest of the (oracle-computed) root is the reference's ID.  The Merkle half is the -m gpu test."""
    from boundless_amd import build, image

    build.build(verbose=False)
    im = image.MemoryImage.from_program(blob())
    _, root = ol.compute_image_id(blob())
    assert image.system_state_digest(root, 0) == iid()
    assert image.system_state_digest(root, 4) != iid()
    idx = im.page_indices()
    assert 200 < len(im) == len(idx) <= 286 and (np.diff(idx.astype(np.int64)) > 0).all()  # all-zero pages are never materialised
    # the three system words the loader writes
    user = blob()[32:]
    assert im.get_page(0x10000 >> 10)[0] == struct.unpack("<I", user[24:28])[0]  # user entry at USER_START_ADDR
    sysp = im.get_page(0xFFFF0000 >> 10)
    assert sysp[0x210 >> 2] == 0xC0000000 and sysp[0x214 >> 2] == 1
    assert not im.get_page(5).any()  # an absent page reads as zeros
    from boundless_amd.hal import HalError

    for bad in (b"", b"R0BF" + b"\0" * 20, blob()[:1000]):
        with pytest.raises(HalError, match="image:"):
            image.MemoryImage.from_program(bad)


def test_program_loader_survives_hostile_program_binaries(tmp_path):
    """The ProgramBinary / ELF loader parses bytes that arrive from outside (the reference's executor API recomputes the image ID of
    whatever is uploaded, crates/executor/src/api.rs:166-178): under AddressSanitizer / UBSan, truncations, bit flips and hostile
    program-header fields — offsets past the file, wrapping ranges, gigabytes of .bss — are error strings or valid images, and a
    huge p_memsz costs neither memory nor time (absent pages are zero pages)."""
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "image_fuzz_check")
    r = subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-omit-frame-pointer",
                        f"-I{os.path.join(root, 'include')}", os.path.join(root, "boundless_amd", "csrc", "image_host.cpp"),
                        os.path.join(root, "tests", "image_fuzz_check.cpp"), "-o", exe], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    r = subprocess.run([exe, os.path.join(REF, "boundless-povw-log-updater.bin"), "400"], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "image_fuzz_check ok" in r.stdout and "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr
