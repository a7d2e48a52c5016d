"""CPU: the C-ABI library builds for gfx950, loads, and exports every symbol include/*.h declares."""
import ctypes
import glob
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    syms = set()
    for h in glob.glob(os.path.join(ROOT, "include", "*.h")):
        text = open(h).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        syms.update(re.findall(r"\b(bx_[a-z0-9_]+)\s*\(", text))
    return sorted(syms)


@pytest.fixture(scope="module")
def libpath():
    from boundless_amd import build

    return build.build(verbose=False)


def test_library_exports_every_declared_symbol(libpath):
    lib = ctypes.CDLL(libpath)
    syms = declared_symbols()
    assert len(syms) >= 30
    missing = [s for s in syms if not hasattr(lib, s)]
    assert not missing, f"declared in include/*.h but not exported: {missing}"


def test_library_exports_only_the_c_abi(libpath):
    """-fvisibility=hidden + csrc/exports.map: no C++ internal (bx_planner::merge, bx_mem_taskdb::find_locked, kernel host stubs ...)
    is a defined dynamic symbol; every exported name is a declared bx_* entry point."""
    import subprocess

    out = subprocess.run(["nm", "-D", "--defined-only", libpath], capture_output=True, text=True, check=True).stdout
    names = [line.split()[-1] for line in out.splitlines() if line.strip()]
    assert names, "no dynamic symbols at all?"
    mangled = [n for n in names if n.startswith("_Z")]
    assert not mangled, f"C++ symbols exported: {mangled[:5]} ... ({len(mangled)})"
    declared = set(declared_symbols())
    extra = [n for n in names if n not in declared]
    assert not extra, f"exported but not declared in include/*.h: {extra[:10]}"


def test_python_binding_declares_the_same_surface(libpath):
    from boundless_amd import hal

    lib = hal.load_library()
    for s in declared_symbols():
        assert getattr(lib, s) is not None


def test_tracing_switch_needs_no_gpu_and_rejects_bad_levels(libpath):
    """bx_trace_enable dlopen's the roctx library of the ROCm install; without a profiler attached the ranges are no-ops."""
    from boundless_amd import hal

    assert hal.trace_level() == 0  # off unless asked for (BX_TRACE is read by bx_init, which this process never reaches)
    hal.trace_enable(2)
    assert hal.trace_level() == 2
    with pytest.raises(hal.HalError, match="level must be 0, 1 or 2"):
        hal.trace_enable(3)
    assert hal.trace_level() == 2
    hal.trace_enable(0)
    assert hal.trace_level() == 0


def test_init_fails_loudly_without_gpu(libpath):
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from boundless_amd import hal

    with pytest.raises(hal.HalError):
        hal.HipHal(0)


def test_product_never_imports_oracle():
    """The product package must not reference the oracle (parity would be void)."""
    for path in glob.glob(os.path.join(ROOT, "boundless_amd", "**", "*"), recursive=True):
        if os.path.isfile(path) and path.endswith((".py", ".hip", ".cpp", ".hpp", ".h")):
            text = open(path).read()
            assert "bx_oracle" not in text and "from oracle" not in text and "import oracle" not in text, path


def test_headers_are_valid_c99():
    """The boundary is a C ABI: a plain-C translation unit including both headers must compile with gcc."""
    import subprocess

    src = os.path.join(ROOT, "tests", "c_abi_smoke.c")
    r = subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-fsyntax-only", f"-I{os.path.join(ROOT, 'include')}", src],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    r = subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-fsyntax-only", f"-I{os.path.join(ROOT, 'include')}",
                        "-x", "c", "-"], input='#include "bx_agent.h"\nint main(void){bx_agent_config c; (void)c; return 0;}\n',
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


def test_native_agent_fails_loudly_without_gpu(libpath):
    """The agent's default prover is the HIP one; with no GPU it must refuse to start (no CPU fallback)."""
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from boundless_amd import agent as ag
    from boundless_amd.hal import HalError

    with pytest.raises(HalError, match="bx_agent_create"):
        ag.Agent(prover=None)


def test_native_agent_validates_widths_against_po2_max_at_create(libpath):
    """a configuration whose buffers cannot exist is refused by bx_agent_create, not by the first oversized task."""
    from boundless_amd import agent as ag
    from boundless_amd.hal import HalError

    with pytest.raises(HalError, match="group widths must be below 65536"):
        ag.Agent(prover=None, widths=(16, 70000, 64))
    with pytest.raises(HalError, match="more than the 288 GB of one GPU"):
        ag.Agent(prover=None, widths=(16, 4096, 64), po2_range=(9, 24))


def test_plain_hal_driver_is_plain_c_on_the_declared_abi(libpath):
    """tests/plain_hal_prover.c (the trait-level driver of tests/test_plain_hal_gpu.py and of bench.py's single_proof_ms.plain_hal)
    compiles as pedantic C99 against include/*.h alone, links against the library, exports its six entry points, and references
    nothing of the library but declared bx_* symbols — in particular not bx_prove_segment, the in-library prover it is compared with,
    and nothing of oracle/."""
    import subprocess
    import sys

    src = os.path.join(ROOT, "tests", "plain_hal_prover.c")
    r = subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-fsyntax-only", f"-I{os.path.join(ROOT, 'include')}", src],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import plain_hal

    so = plain_hal.build()
    out = subprocess.run(["nm", "-D", so], capture_output=True, text=True, check=True).stdout
    defined = {ln.split()[-1] for ln in out.splitlines() if " T " in ln}
    assert {"ph_create", "ph_prove", "ph_destroy", "ph_seal_words", "ph_last_calls", "ph_error"} <= defined
    used = {ln.split()[-1] for ln in out.splitlines() if " U " in ln and ln.split()[-1].startswith("bx")}
    assert used and used <= set(declared_symbols())
    assert not any(s.startswith("bx_prove") or s.startswith("bx_prover_") or s.startswith("bxo_") for s in used), used
    text = open(src).read()
    assert "oracle" not in text.split("*/", 1)[1]  # the header comment says it does not use oracle/; the code must not either
