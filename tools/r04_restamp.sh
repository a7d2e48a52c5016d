#!/bin/bash
# Re-collect only the two PMC summaries that bench.py replays (steps 2 and 3 of tools/r04_profile.sh) and the bench line, e.g. after a
# change of the stamp's definition or of a device source that did not touch the profiled kernels.  Run on the GPU box from the repo root.
set -u
O=gpurun_out/r4p; mkdir -p $O
export TMPDIR=/tmp
B="python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-agent-mode --no-pcie-extra"
for ctr in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $ctr --output-format csv -d $O/pmc_$ctr -o pmc -- $B --inflight 1 > /dev/null 2>&1
done
python tools/pmc_traffic.py "$(find $O/pmc_FETCH_SIZE -name '*counter_collection.csv' | head -1)" "$(find $O/pmc_WRITE_SIZE -name '*counter_collection.csv' | head -1)" $O/r04_bench_pmc_traffic.json
rm -rf $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE
rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES --output-format csv -d $O/pmc_valu -o pmc -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-agent-mode --no-pcie-extra > /dev/null 2>&1
python tools/job_valu.py "$(find $O/pmc_valu -name '*counter_collection.csv' | head -1)" 14 $O/r04_job_valu_insts.json
rm -rf $O/pmc_valu
cp $O/r04_bench_pmc_traffic.json $O/r04_job_valu_insts.json profiles/
python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 > $O/r04_bench_steps20_warmup5.json
