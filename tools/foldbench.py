"""Small-layer Merkle folding: one lane per node vs four (fold_quad), and where the fused kernel takes over (fold_fuse_below).
`PYTHONPATH=. python tools/foldbench.py` on the GPU box."""
import numpy as np

from boundless_amd.hal import HipHal
from oracle import oracle_lib as ol

hal = HipHal(0)
for rows, cols in ((1 << 15, 16), (1 << 18, 16), (1 << 22, 16)):
    x = ol.random_elems(np.random.default_rng(1), rows * cols)
    m = hal.copy_from(x)
    nodes = hal.alloc_digest(2 * rows)
    for quad, fuse, wg in ((0, 15, 512), (1, 15, 512), (1, 17, 512), (1, 17, 256), (1, 17, 128), (1, 17, 64), (1, 18, 128), (1, 19, 128), (1, 20, 128),
                           (1, 18, 64), (1, 19, 256)):
        hal.set_tunable("fold_quad", quad)
        hal.set_tunable("fold_fuse_below", 1 << fuse)
        hal.set_tunable("fold_quad_wg", wg)
        hal.merkle_build(nodes, m, rows)
        hal.sync()
        hal.timer_start()
        for _ in range(10):
            hal.merkle_build(nodes, m, rows)
        print(f"rows 2^{rows.bit_length() - 1} x {cols}: fold_quad {quad} fuse_below 2^{fuse} wg {wg}: {hal.timer_stop() * 1000 / 10:8.1f} us")
