#!/bin/bash
# Round-6 retry of the Infinity-Cache column-group LDE on the round-5 kernels: pass A + pass B back to back on G
# columns at a time (tunable ntt_group_cols; 0 = one pass A and one pass B over all columns), so that pass B finds pass A's output in
# the 256 MB memory-side cache.  No new kernels.  Run on the GPU box from the repo root; writes gpurun_out/r6g/ and
# profiles/r06_ntt_column_groups.json.
set -u
O=gpurun_out/r6g; mkdir -p $O
export TMPDIR=/tmp
# parity with groups on
BX_TUNABLES=ntt_group_cols=8 timeout 900 python -m pytest tests/test_hal_gpu.py -q -x -k "interpolate_zkshift or lde or full_size_vs_oracle" 2>&1 | tail -1 > $O/parity.txt
BX_TUNABLES=ntt_group_cols=12 timeout 600 python -m pytest tests/test_fullsize_gpu.py -q -x -k "lde_at" 2>&1 | tail -1 >> $O/parity.txt
cat $O/parity.txt
# the LDE alone, 2^20 -> 2^22 x 256 columns
: > $O/ldebench.jsonl
for g in 0 4 8 12 16 32; do python tools/ldebench.py --cols 256 --reps 20 --tunables ntt_group_cols=$g --tag g$g 2>/dev/null | grep expand >> $O/ldebench.jsonl; done
cat $O/ldebench.jsonl
# inside the job: one and three segments in flight
: > $O/bench.jsonl
for l in 1 3; do for g in 0 4 8 12; do
  BX_TUNABLES=ntt_group_cols=$g python bench.py --steps 8 --warmup 3 --inflight $l --no-cpu-baseline --no-agent-mode --no-pcie-extra --no-plain-hal 2>/dev/null | tail -1 > $O/b.json
  python - <<PY >> $O/bench.jsonl
import json
j=json.load(open("$O/b.json"))
print(json.dumps({"inflight": $l, "group_cols": $g, "value": round(j["value"],3), "lde_isolated_ms": j["roofline"].get("avg_ms_per_launch"), "frac": j["roofline"].get("frac"),
                  "single_proof_ms": j["single_proof_ms"]["min"]}))
PY
done; done
cat $O/bench.jsonl
# HBM-side traffic and clocks of the two LDE kernels, with and without groups (separate counter passes, no tracing domains)
for g in 0 8; do
  for ctr in FETCH_SIZE WRITE_SIZE GRBM_GUI_ACTIVE; do
    rocprofv3 --pmc $ctr --output-format csv -d $O/pmc_g${g}_$ctr -o pmc -- python tools/ldebench.py --cols 256 --reps 4 --warm-seconds 0.5 --tunables ntt_group_cols=$g > /dev/null 2>&1
  done
  python tools/pmc_traffic.py "$(find $O/pmc_g${g}_FETCH_SIZE -name '*counter_collection.csv' | head -1)" "$(find $O/pmc_g${g}_WRITE_SIZE -name '*counter_collection.csv' | head -1)" $O/traffic_g$g.json > /dev/null
  python tools/pmc_clock.py "$(find $O/pmc_g${g}_GRBM_GUI_ACTIVE -name '*counter_collection.csv' | head -1)" $O/clock_g$g.json "ntt_" > /dev/null
  rm -rf $O/pmc_g${g}_FETCH_SIZE $O/pmc_g${g}_WRITE_SIZE $O/pmc_g${g}_GRBM_GUI_ACTIVE
done
python - <<'PY'
import json, os
O = "gpurun_out/r6g"
from boundless_amd.build import device_code_hash
out = {"what": "Infinity-Cache column-group LDE on the round-5 kernels (VERDICT r05 item 3): pass A + pass B back to back on G columns at a time "
               "(tunable ntt_group_cols) vs one pass A and one pass B over all 256 columns (G = 0); 2^20 -> 2^22 rows",
       "command": "bash tools/r06_ntt_groups.sh", "device_code_sha": device_code_hash(), "parity": open(f"{O}/parity.txt").read().split("\n"),
       "ldebench_256_columns": [json.loads(l) for l in open(f"{O}/ldebench.jsonl") if l.strip()],
       "bench": [json.loads(l) for l in open(f"{O}/bench.jsonl") if l.strip()], "pmc": {}}
for g in (0, 8):
    t = json.load(open(f"{O}/traffic_g{g}.json"))["kernels"]
    c = json.load(open(f"{O}/clock_g{g}.json"))["kernels"]
    out["pmc"][f"group_cols_{g}"] = {"traffic": {k: v for k, v in t.items() if "ntt_" in k}, "clock": c}
json.dump(out, open("profiles/r06_ntt_column_groups.json", "w"), indent=1)
json.dump(out, open(f"{O}/r06_ntt_column_groups.json", "w"), indent=1)
PY
