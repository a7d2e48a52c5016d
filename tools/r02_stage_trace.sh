#!/bin/bash
# Per-stage device time of bx_prove_segment from roctx ranges (bx_trace_enable level 2: the stream is drained at the end of
# every stage, so a stage's host-side range is its device time).  Run on the GPU box from the repo root.
set -u
O=gpurun_out/r2t; mkdir -p $O
export TMPDIR=/tmp
BX_TRACE=2 rocprofv3 --kernel-trace --marker-trace --stats --output-format csv -d $O/mt -o mt -- \
  python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-agent-mode --inflight 1 > $O/bench_mt.json 2> $O/bench_mt.err
find $O/mt -name "*.csv" | sed 's/^/  /'
for f in $(find $O/mt -name "*stats.csv"); do cp $f $O/r02_stage_trace_$(basename $f | sed 's/^mt_//'); done
M=$(find $O/mt -name "*marker_api_trace.csv" | head -1)
[ -n "$M" ] && python tools/stage_trace.py "$M" $O/r02_stage_times.json
rm -rf $O/mt
ls $O
