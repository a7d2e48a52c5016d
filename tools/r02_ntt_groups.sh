#!/bin/bash
# NTT column-group experiment: pass A + pass B back to back on G columns at a time.
# Run on the GPU box from the repo root; writes gpurun_out/r2e/.
set -u
O=gpurun_out/r2e; mkdir -p $O
export TMPDIR=/tmp
echo "== parity with groups" > $O/log.txt
BX_TUNABLES=ntt_group_cols=8 python -m pytest tests/test_hal_gpu.py -q -x -k "interpolate_zkshift_lde or lde" 2>&1 | tail -2 >> $O/log.txt
BX_TUNABLES=ntt_group_cols=8 python -m pytest tests/test_fullsize_gpu.py -q -x -k "lde_at" 2>&1 | tail -2 >> $O/log.txt
for g in 0 8 16 32 64; do
  BX_TUNABLES=ntt_group_cols=$g python tools/opbench.py --po2 20 --cols 256 --reps 4 2>/dev/null | grep -E "expand_into|interpolate" | sed "s/^/g=$g /" >> $O/opbench.txt
done
cat $O/opbench.txt
for g in 0 8 32; do
  BX_TUNABLES=ntt_group_cols=$g python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-agent-mode > $O/bench_g$g.json 2>/dev/null
  python - <<PY
import json
j=json.loads(open("$O/bench_g$g.json").read().strip().splitlines()[-1])
print("g=$g value", round(j["value"],3), "lde in-region avg_ms", j["roofline_in_region"]["avg_ms_per_launch"], "isolated", j["roofline"]["avg_ms_per_launch"])
PY
done
# HBM-side traffic of the LDE kernels with and without groups (separate PMC passes, no tracing domains)
for g in 0 8; do
  for ctr in FETCH_SIZE WRITE_SIZE; do
    BX_TUNABLES=ntt_group_cols=$g rocprofv3 --pmc $ctr --output-format csv -d $O/pmc_g${g}_$ctr -o pmc -- python tools/opbench.py --po2 20 --cols 256 --reps 1 > /dev/null 2>&1
  done
  F=$(find $O/pmc_g${g}_FETCH_SIZE -name "*counter_collection.csv" | head -1); W=$(find $O/pmc_g${g}_WRITE_SIZE -name "*counter_collection.csv" | head -1)
  python tools/pmc_traffic.py "$F" "$W" $O/traffic_g$g.json | grep -i ntt
  rm -rf $O/pmc_g${g}_FETCH_SIZE $O/pmc_g${g}_WRITE_SIZE
done
