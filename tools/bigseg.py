"""Proof time and memory of the largest segment sizes at the BASELINE widths (`PYTHONPATH=. python tools/bigseg.py [po2 ...]`)."""
import sys
import time

import numpy as np
import torch

from boundless_amd.prover import HipProverServer, Segment, verify_seal

for po2 in [int(a) for a in sys.argv[1:]] or [22, 23, 24]:
    srv = HipProverServer(0, po2=po2)
    free, total = torch.cuda.mem_get_info(0)
    srv.prove_segment(Segment.synthetic(0, po2=po2))
    t = time.time()
    r = srv.prove_segment(Segment.synthetic(1, po2=po2))
    dt = time.time() - t
    srv.close()
    verify_seal(r.seal)
    print(f"po2 {po2}: {(total - free) / 2**30:.1f} GB, {dt:.3f} s per proof = {(1 << po2) / dt / 1e6:.1f} M cycles/s, verified", flush=True)
