"""N lone proofs at the metric's size through the trait-level driver (tests/plain_hal_prover.c), for rocprofv3:
    rocprofv3 --kernel-trace --stats --output-format csv -d out -o kt -- python tools/plain_hal_run.py [N] [flags]
BX_TUNABLES=gather_defer=0 shows the un-queued openings (one gather_sample_kernel launch per row / digest)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import plain_hal  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 5
flags = int(sys.argv[2]) if len(sys.argv) > 2 else 0
pp = plain_hal.PlainHalProver(0, po2=20, widths=(16, 256, 64), flags=flags)
ts = []
for k in range(n):
    _, ms = pp.prove(0xB0D1E550000 + k)
    ts.append(ms)
print({"proofs": n, "flags": flags, "ms": [round(t, 2) for t in ts], "calls": pp.calls})
pp.close()
