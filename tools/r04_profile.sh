#!/bin/bash
# Round-4 profile set (run on the GPU box from the repo root; results under gpurun_out/r4p/, copy the summaries to profiles/).
set -u
O=gpurun_out/r4p; mkdir -p $O
export TMPDIR=/tmp
B="python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-agent-mode --no-pcie-extra"
# 1. per-kernel time: the default command (3 segments in flight) and one segment in flight
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt3 -o kt -- $B --no-live-profile > $O/bench_kt3.json 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt1 -o kt -- $B --inflight 1 --no-live-profile > $O/bench_kt1.json 2>/dev/null
cp $(find $O/kt3 -name "*kernel_stats.csv" | head -1) $O/r04_bench_kernel_stats_default_cmd.csv
cp $(find $O/kt1 -name "*kernel_stats.csv" | head -1) $O/r04_bench_kernel_stats_inflight1.csv
python tools/latency_gaps.py "$(find $O/kt1 -name "*kernel_trace.csv" | head -1)" $O/r04_latency_gaps.json > /dev/null
rm -rf $O/kt3 $O/kt1
# 2. HBM-side traffic per kernel (separate counter passes, no tracing domains), one segment in flight
for ctr in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $ctr --output-format csv -d $O/pmc_$ctr -o pmc -- $B --inflight 1 > /dev/null 2>&1
done
python tools/pmc_traffic.py "$(find $O/pmc_FETCH_SIZE -name '*counter_collection.csv' | head -1)" "$(find $O/pmc_WRITE_SIZE -name '*counter_collection.csv' | head -1)" $O/r04_bench_pmc_traffic.json
rm -rf $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE
# 3. VALU instructions of the whole job, per segment: steps 2 x 3 lanes + warm-up 3 + 4 lone proofs + 1 isolated probe = 14 segments
rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES --output-format csv -d $O/pmc_valu -o pmc -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-agent-mode --no-pcie-extra > /dev/null 2>&1
python tools/job_valu.py "$(find $O/pmc_valu -name '*counter_collection.csv' | head -1)" 14 $O/r04_job_valu_insts.json
rm -rf $O/pmc_valu
# 4. the segment's bytes: 80 MB per segment, staged and uploaded (one slot at a time, and two deep), 1 and 3 in flight; kernel +
#    memory-copy trace of the two-deep run (the H2D copies next to the kernels they overlap)
for l in 3 1; do for m in "" "--segment-bytes 80000000" "--segment-bytes 80000000 --two-deep"; do
  python bench.py --steps 20 --warmup 5 --inflight $l $m --no-cpu-baseline --no-agent-mode --no-pcie-extra 2>/dev/null | tail -1 | \
    python -c "import json,sys; j=json.loads(sys.stdin.read()); print(json.dumps({'inflight':$l,'mode':'$m' or 'resident','segment_proofs_per_s':round(j['value'],3),'host_cpu_s_per_proof':j['host_cpu_s_per_proof']}))"
done; done > $O/r04_segment_bytes.jsonl
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O/ktm -o kt -- $B --segment-bytes 80000000 --two-deep --no-live-profile > /dev/null 2>&1
python tools/copy_overlap.py "$(find $O/ktm -name '*memory_copy_trace.csv' | head -1)" "$(find $O/ktm -name '*kernel_trace.csv' | head -1)" $O/r04_h2d_overlap.json 80000000
head -2 "$(find $O/ktm -name '*memory_copy_trace.csv' | head -1)" > $O/r04_memory_copy_trace_head.csv
rm -rf $O/ktm
# 5. one planned job of 64 segments: proves -> stand-in joins -> resolve -> finalize
for l in 3 1; do python bench.py --job 64 --inflight $l 2>/dev/null | tail -1; done > $O/r04_job64.jsonl
#    ... and the same job as two processes (gloo: two ranks cannot share a GPU under RCCL) sharing the one GPU: subtree per rank, one all_gather
python -m torch.distributed.run --nnodes=1 --nproc-per-node=2 --master-addr 127.0.0.1 --master-port 29641 bench.py --job 64 --gpus 2 --dist-backend gloo --device 0 2>/dev/null | grep '^{' | tail -1 > $O/r04_job64_2ranks_1gpu.json
# 6. the helper entry points alone
python tools/helperbench.py > $O/r04_helperbench.jsonl
# 7. LAST: the bench line of the driver's command, replaying the PMC summaries just collected (so that its `profile_stale` is about the
#    sources, not about the order of this script)
cp $O/r04_bench_pmc_traffic.json $O/r04_job_valu_insts.json profiles/
python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 > $O/r04_bench_steps20_warmup5.json
head -14 $O/r04_bench_kernel_stats_default_cmd.csv | cut -c1-150
cat $O/r04_segment_bytes.jsonl
