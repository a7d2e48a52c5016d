"""CPU oracle prover vs thread count on this box (`PYTHONPATH=. python tools/cpu_threads.py [po2]`): picks cpu_baseline's thread count."""
import os
import sys
import time

from oracle import oracle_lib as ol

po2 = int(sys.argv[1]) if len(sys.argv) > 1 else 18
path = ol.build(force=True, native=True, out="/tmp/libbx_oracle_native.so")
L = ol.lib(path)
ol.prove_segment(10, 2, 4, 2, 1, L)
print("hardware threads", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
for threads in (8, 16, 32, 64, 128, 256):
    if threads > (os.cpu_count() or 1):
        break
    L.bxo_set_threads(threads)
    t0 = time.time()
    ol.prove_segment(po2, 16, 256, 64, 0xB0D1E550000, L)
    print(f"po2 {po2}: {threads:4d} threads {time.time() - t0:7.2f} s", flush=True)
