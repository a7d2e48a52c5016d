"""The stamp of the committed PMC summaries is a hash of the device code that ran, not of source text:
a host-only edit of include/bx_prover.h recompiles circuit.hip / prover.hip but must not make a profile look stale; an edit of a
kernel must.  Runs hipcc for gfx950 on one small translation unit (no GPU needed)."""
import os
import shutil
import subprocess

import pytest

from boundless_amd import build as b

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _compile(tree, out):
    src = os.path.join(tree, "boundless_amd", "csrc", "circuit.hip")
    flags = [f for f in b.FLAGS if not f.startswith("-I")] + [f"-I{tree}/include", f"-I{tree}/boundless_amd/csrc"]
    r = subprocess.run(["hipcc", "-x", "hip"] + flags + [b.cuid_flag(src), "-c", src, "-o", out], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return b.device_code_hash(out)


@pytest.mark.skipif(shutil.which("hipcc") is None, reason="hipcc not on PATH")
def test_device_code_stamp_ignores_host_only_edits(tmp_path):
    tree = str(tmp_path / "tree")
    shutil.copytree(os.path.join(ROOT, "include"), os.path.join(tree, "include"))
    shutil.copytree(os.path.join(ROOT, "boundless_amd", "csrc"), os.path.join(tree, "boundless_amd", "csrc"))
    obj = str(tmp_path / "circuit.o")
    base = _compile(tree, obj)
    assert base == _compile(tree, obj), "the device code object is not reproducible"

    # 0. the same sources in another directory give the same code object (hipcc's path-derived cuid is replaced by -cuid=<file name>, passed per translation unit by build.cuid_flag)
    tree2 = str(tmp_path / "elsewhere" / "deeper" / "tree")
    shutil.copytree(tree, tree2)
    assert _compile(tree2, str(tmp_path / "circuit2.o")) == base, "the device-code stamp depends on the build directory"

    # 1. a host-only declaration in bx_prover.h (what staled every r04 profile): text hash would change, the code object does not
    hdr = os.path.join(tree, "include", "bx_prover.h")
    text = open(hdr).read()
    assert "circuit.hpp" in open(os.path.join(tree, "boundless_amd", "csrc", "circuit.hip")).read()
    marker = "#ifdef __cplusplus\n}"
    assert marker in text
    open(hdr, "w").write(text.replace(marker, "/* host-only */ const char* bx_stamp_probe_decl(int);\n" + marker, 1))
    assert _compile(tree, obj) == base, "a host-only header edit changed the device-code stamp"

    # 2. an edit inside one kernel changes it
    hip = os.path.join(tree, "boundless_amd", "csrc", "circuit.hip")
    src = open(hip).read()
    line = "        code[i] = synth_code_cell(cc, (uint32_t)(i >> cc.po2), (uint32_t)(i & (n - 1)));"
    assert line in src
    open(hip, "w").write(src.replace(line, line.replace("(n - 1)));", "(n - 1))) ^ 1u;"), 1))
    assert _compile(tree, obj) != base, "a kernel edit did not change the device-code stamp"


def test_library_stamp_is_the_fatbin_section():
    if not os.path.exists(b.LIB):
        pytest.skip("library not built")
    sec = b.elf_section(b.LIB, ".hip_fatbin")
    assert sec is not None and len(sec) > 1 << 16
    assert sec[:24].startswith(b"__CLANG_OFFLOAD_BUNDLE__")
    assert len(b.device_code_hash()) == 16
    assert b.elf_section(b.LIB, ".no_such_section") is None
