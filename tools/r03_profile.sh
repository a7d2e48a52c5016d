#!/bin/bash
# Round-3 profile set (run on the GPU box from the repo root; results under gpurun_out/r3p/, copy the summaries to profiles/).
set -u
O=gpurun_out/r3p; mkdir -p $O
export TMPDIR=/tmp
B="python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-agent-mode"
# 0. the bench line of the driver's command, and the rate at 1..4 segments in flight under both wait policies
python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 > $O/r03_bench_steps20_warmup5.json
for w in poll spin; do for l in 1 2 3 4; do
  python bench.py --steps 8 --warmup 2 --inflight $l --wait $w --no-cpu-baseline --no-agent-mode 2>/dev/null | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.read()); print(json.dumps({'wait':'$w','inflight':$l,'segment_proofs_per_s':round(j['value'],3),'ms_per_proof_per_lane':round(1e3*$l/j['value'],2),'host_cpu_s_per_proof':j['host_cpu_s_per_proof']}))"
done; done > $O/r03_inflight_sweep.jsonl
# 1. per-kernel time: the default command (3 segments in flight) and one segment in flight (the tracer times the kernels: no HIP events
#    of our own around the entry points, they are barrier packets and show up as 10-20 us gaps)
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt3 -o kt -- $B --no-live-profile > $O/bench_kt3.json 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt1 -o kt -- $B --inflight 1 --no-live-profile > $O/bench_kt1.json 2>/dev/null
cp $(find $O/kt3 -name "*kernel_stats.csv" | head -1) $O/r03_bench_kernel_stats_default_cmd.csv
cp $(find $O/kt1 -name "*kernel_stats.csv" | head -1) $O/r03_bench_kernel_stats_inflight1.csv
python tools/latency_gaps.py "$(find $O/kt1 -name "*kernel_trace.csv" | head -1)" $O/r03_latency_gaps.json
rm -rf $O/kt3 $O/kt1
# 2. HBM-side traffic per kernel (separate counter passes, no tracing domains), one segment in flight
for ctr in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $ctr --output-format csv -d $O/pmc_$ctr -o pmc -- $B --inflight 1 > /dev/null 2>&1
done
python tools/pmc_traffic.py "$(find $O/pmc_FETCH_SIZE -name '*counter_collection.csv' | head -1)" "$(find $O/pmc_WRITE_SIZE -name '*counter_collection.csv' | head -1)" $O/r03_bench_pmc_traffic.json
rm -rf $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE
# 3. VALU instructions of the whole job, per segment: steps 2 x 3 lanes + warm-up 3 + 1 isolated probe = 10 segments
rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES --output-format csv -d $O/pmc_valu -o pmc -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-agent-mode > /dev/null 2>&1
python tools/job_valu.py "$(find $O/pmc_valu -name '*counter_collection.csv' | head -1)" 10 $O/r03_job_valu_insts.json
rm -rf $O/pmc_valu
# 4. the helper entry points alone
python tools/helperbench.py > $O/r03_helperbench.jsonl
head -14 $O/r03_bench_kernel_stats_default_cmd.csv | cut -c1-150
cat $O/r03_inflight_sweep.jsonl
# 5. A/B of the device-side FRI challenges, and the gap histogram of a lone proof under both settings
bash tools/ab_inflight.sh dev_draws 0 1 > $O/r03_ab_dev_draws.jsonl
for d in 0 1; do
  BX_TUNABLES=dev_draws=$d rocprofv3 --kernel-trace --output-format csv -d $O/ktd$d -o kt -- $B --inflight 1 --no-live-profile > /dev/null 2>&1
  python tools/latency_gaps.py "$(find $O/ktd$d -name "*kernel_trace.csv" | head -1)" $O/r03_latency_gaps_dev_draws$d.json > /dev/null
  rm -rf $O/ktd$d
done
