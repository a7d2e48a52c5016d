"""Sustained clock / power under one kernel: loops an entry point for a few seconds and samples rocm-smi meanwhile.

    python tools/sustained_clock.py hash_rows | lde | mix
Prints the median sclk (MHz), package power (W), ms per call.  Used to turn "ns per wave-instruction" into cycles.
"""
import re
import subprocess
import sys
import threading
import time

import numpy as np

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from boundless_amd.hal import HipHal  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "hash_rows"
hal = HipHal(0)
rows, cols = 1 << 22, int(sys.argv[2]) if len(sys.argv) > 2 else 64
x = hal.copy_from(np.random.default_rng(1).integers(0, 2013265921, rows * cols, dtype=np.uint32))
out = hal.alloc_digest(rows)
coef = hal.alloc((rows // 4) * cols)
samples = []
stop = False


def sampler():
    while not stop:
        t = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True).stdout
        m = re.search(r"sclk clock level: \S+ \((\d+)Mhz\)", t)
        p = re.search(r"Package Power \(W\): ([\d.]+)", t)
        if m and p:
            samples.append((int(m.group(1)), float(p.group(1))))
        time.sleep(0.3)


def call():
    if which == "hash_rows":
        hal.hash_rows(out, x)
    elif which == "lde":
        hal.batch_expand_into_evaluate_ntt(x, coef, cols, 2)
    else:
        hal.zk_shift(x, cols)


for _ in range(3):
    call()
hal.sync()
th = threading.Thread(target=sampler)
th.start()
t0 = time.time()
n = 0
while time.time() - t0 < 5.0:
    for _ in range(20):
        call()
    hal.sync()
    n += 20
dt = time.time() - t0
stop = True
th.join()
s = samples[2:] or samples
print({"kernel": which, "cols": cols, "ms_per_call": round(1e3 * dt / n, 4), "sclk_MHz_median": float(np.median([a for a, _ in s])) if s else None,
       "power_W_median": float(np.median([b for _, b in s])) if s else None, "samples": len(s)})
