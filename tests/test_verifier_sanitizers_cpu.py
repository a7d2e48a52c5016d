"""CPU: the host verifier under AddressSanitizer / UBSan against mutated seals (tests/verify_fuzz_check.cpp).

`bx_verify_segment` parses bytes that arrive from outside (a receipt read back from the hot store, prove.rs:53-55 verifies
what the prover returned): a malformed seal must be an error string, never an out-of-bounds read.  The reference gets that
from Rust's bounds checks; the C++ restatement gets it from the sanitizers."""
import os
import subprocess

import pytest

from oracle import oracle_lib as ol

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "boundless_amd", "csrc")


@pytest.mark.parametrize("san,scale", [("address,undefined", 1.0), ("thread", 0.3)])
def test_verifier_rejects_mutated_seals_without_memory_or_ub_errors(tmp_path, san, scale):
    """Every mutation is verified twice — on one thread and on 2..8 threads sharing the queries (bx_verify_set_threads) — and both
    verdicts, text included, must agree; under ThreadSanitizer the same run checks that the query workers share nothing mutable."""
    exe = str(tmp_path / "verify_fuzz_check")
    r = subprocess.run(["g++", "-std=c++17", "-O1", "-g", f"-fsanitize={san}", "-fno-omit-frame-pointer",
                        "-pthread", f"-I{os.path.join(ROOT, 'include')}", os.path.join(CSRC, "verify.cpp"), os.path.join(CSRC, "control_id.cpp"),
                        os.path.join(ROOT, "tests", "verify_fuzz_check.cpp"), "-o", exe], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1", TSAN_OPTIONS="halt_on_error=1")
    # two shapes: accumulator pairs + public words, and the smallest legal widths
    for k, (shape, knobs, iters) in enumerate((((9, 2, 5, 8), (0, 0), 500), ((10, 1, 1, 1), (3, 2), 300))):
        seal, _ = ol.prove_segment(*shape, 11 + k, terms=knobs[0], degree=knobs[1])
        path = str(tmp_path / f"seal{k}.bin")
        seal.astype("<u4").tofile(path)
        r = subprocess.run([exe, path, str(max(20, int(iters * scale)))], capture_output=True, text=True, timeout=600, env=env)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
        assert "verify_fuzz_check ok" in r.stdout and "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr
        assert "ThreadSanitizer" not in r.stderr


def test_rest_client_survives_hostile_server_answers(tmp_path):
    """The REST worker client parses what a server sends: truncated and oversized bodies, chunk sizes that wrap, unterminated
    and 100 000-deep JSON, wrong types where an i32 belongs — all are errors, none is a crash, a hang or a silent zero."""
    exe = str(tmp_path / "rest_fuzz_check")
    r = subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-omit-frame-pointer", "-pthread",
                        f"-I{os.path.join(ROOT, 'include')}", os.path.join(CSRC, "rest_worker.cpp"),
                        os.path.join(ROOT, "tests", "rest_fuzz_check.cpp"), "-o", exe], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "rest_fuzz_check ok" in r.stdout and "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr
