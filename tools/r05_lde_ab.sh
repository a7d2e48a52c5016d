#!/bin/bash
# A/B of the LDE kernel variants on a warmed-up GPU: ntt_fused = 0 (canonical), 1 (pass A fused), 2 (pass B fused), 3 (both)
set -u
O=gpurun_out/r5; mkdir -p $O
TAG=${1:-ab}
for rep in 1 2; do for f in 0 1 2 3; do
  python tools/ldebench.py --cols 256 --reps 20 --tunables ntt_fused=$f --tag fused$f 2>/dev/null | grep expand
done; done | tee $O/${TAG}_lde_ab.jsonl
