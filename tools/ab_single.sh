#!/bin/bash
# A/B of a tunable on the latency of a LONE proof and on the 3-in-flight rate: tools/ab_single.sh <tunable> <v0> <v1> [repeats]
# (alternating runs, so that clock drift of the box hits both settings alike)
T=$1; A=$2; B=$3; R=${4:-3}
for i in $(seq $R); do for v in $A $B; do
  BX_TUNABLES=$T=$v python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-agent-mode --no-pcie-extra 2>/dev/null | tail -1 | \
    python -c "import json,sys; j=json.loads(sys.stdin.read()); print(json.dumps({'tunable':'$T','value':$v,'segment_proofs_per_s':round(j['value'],3),'single_proof_ms':j['single_proof_ms']}))"
done; done
