"""CPU: the arithmetic *source* of the gfx950 kernels (fp.hpp, poseidon2_arith.hpp), compiled for the host with every
documented bound and 64-bit accumulation asserted, against exact 128-bit arithmetic — extreme operands included."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_kernel_arithmetic_bounds_and_exactness(tmp_path):
    exe = str(tmp_path / "host_arith_check")
    csrc = os.path.join(ROOT, "boundless_amd", "csrc")
    r = subprocess.run(["g++", "-std=c++17", "-O2", "-DBX_CHECK_BOUNDS", f"-I{csrc}", f"-I{os.path.join(ROOT, 'include')}",
                        os.path.join(ROOT, "tests", "host_arith_check.cpp"), "-o", exe], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "host_arith_check ok" in r.stdout
