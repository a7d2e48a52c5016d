#!/bin/bash
# Re-collect only the two PMC summaries that bench.py replays (steps 2 and 3 of tools/r03_profile.sh), after a source change that
# did not touch the kernels they describe but did change the csrc stamp.  Run on the GPU box from the repo root.
set -u
O=gpurun_out/r3p; mkdir -p $O
export TMPDIR=/tmp
B="python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-agent-mode"
for ctr in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $ctr --output-format csv -d $O/pmc_$ctr -o pmc -- $B --inflight 1 > /dev/null 2>&1
done
python tools/pmc_traffic.py "$(find $O/pmc_FETCH_SIZE -name '*counter_collection.csv' | head -1)" "$(find $O/pmc_WRITE_SIZE -name '*counter_collection.csv' | head -1)" $O/r03_bench_pmc_traffic.json
rm -rf $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE
rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES --output-format csv -d $O/pmc_valu -o pmc -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-agent-mode > /dev/null 2>&1
python tools/job_valu.py "$(find $O/pmc_valu -name '*counter_collection.csv' | head -1)" 10 $O/r03_job_valu_insts.json
rm -rf $O/pmc_valu
python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 > $O/r03_bench_steps20_warmup5.json
