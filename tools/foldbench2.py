"""Large Merkle layers: one launch per layer vs several levels per launch (fold_deep), trees of the sizes a 2^20 proof builds.
`python tools/foldbench2.py` on the GPU box; one JSON line per setting (microseconds per tree, hash_rows excluded)."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from boundless_amd.hal import HipHal  # noqa: E402

hal = HipHal(0)
for rows in (1 << 22, 1 << 18):
    digs = np.random.default_rng(1).integers(0, 2013265921, 16 * rows, dtype=np.uint32)
    nodes = hal.copy_from(digs)
    for deep, minl in ((1, 1 << 18), (2, 1 << 18), (3, 1 << 18), (3, 1 << 17), (3, 1 << 16), (2, 1 << 17)):
        hal.set_tunable("fold_deep", deep)
        hal.set_tunable("fold_deep_min_lanes", minl)

        def fold():
            size = rows
            # merkle_build's loop without the leaf hashing: fold every layer down to the root through bx_merkle_build's policy
            hal._check(hal.lib.bx_merkle_fold(hal.ctx, nodes.raw, rows))

        fold()
        hal.sync()
        hal.timer_start()
        for _ in range(10):
            fold()
        us = hal.timer_stop() * 100
        print(json.dumps({"rows": rows, "fold_deep": deep, "min_lanes": minl, "us_per_tree": round(us, 1)}))
