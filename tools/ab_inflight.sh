#!/bin/bash
# A/B of one tunable over 1..3 segments in flight: tools/ab_inflight.sh <name> <value A> <value B>   (run on the GPU box)
n=${1:-dev_draws}; a=${2:-0}; b=${3:-1}
for rep in 1 2; do for v in $a $b; do for l in 1 2 3; do
  BX_TUNABLES=$n=$v python bench.py --steps 10 --warmup 3 --inflight $l --no-cpu-baseline --no-agent-mode 2>/dev/null | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.read()); print(json.dumps({'$n':$v,'inflight':$l,'rep':$rep,'segment_proofs_per_s':round(j['value'],3),'ms_per_proof_per_lane':round(1e3*$l/j['value'],2),'host_cpu_s_per_proof':j['host_cpu_s_per_proof']}))"
done; done; done
