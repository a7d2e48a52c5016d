#!/bin/bash
# Round-5 LDE probe (run on the GPU box from the repo root): NTT parity, LDE timing, per-kernel stall counters.
# Results under gpurun_out/r5/; the summaries worth keeping are copied to profiles/ by hand.
set -u
O=gpurun_out/r5; mkdir -p $O
export TMPDIR=/tmp
TAG=${1:-probe}
# 1. parity of every NTT path against the oracle (small sizes, both kernel families, full size, columns per workgroup)
timeout 900 python -m pytest tests/test_hal_gpu.py -x -q -k "ntt or lde or interpolate or evaluate" > $O/${TAG}_ntt_tests.log 2>&1
tail -3 $O/${TAG}_ntt_tests.log
timeout 600 python -m pytest tests/test_fullsize_gpu.py -x -q -k "lde_at_2_20 or seal_bit_exact" >> $O/${TAG}_ntt_tests.log 2>&1
tail -2 $O/${TAG}_ntt_tests.log
# 2. the LDE and the inverse alone
for cols in 64 256; do python tools/ldebench.py --cols $cols --reps 10 --interp --check --tag $TAG; done > $O/${TAG}_ldebench.jsonl 2>&1
for cpw in 4 16; do python tools/ldebench.py --cols 256 --reps 10 --tunables ntt_cols_per_wg=$cpw --tag cpw$cpw; done >> $O/${TAG}_ldebench.jsonl 2>&1
cat $O/${TAG}_ldebench.jsonl
# 3. per-kernel durations
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o kt -- python tools/ldebench.py --cols 256 --reps 10 --interp > /dev/null 2>&1
cp "$(find $O/kt -name '*kernel_stats.csv' | head -1)" $O/${TAG}_lde_kernel_stats.csv; rm -rf $O/kt
head -8 $O/${TAG}_lde_kernel_stats.csv | cut -c1-160
# 4. stall side of the two LDE kernels (separate counter passes; no tracing domains)
rocprofv3 -L > $O/counters_avail.txt 2>&1
P1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU"
P2="SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES"
P3="SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES"
P4="SQ_INST_CYCLES_VMEM SQ_WAVE_CYCLES SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS"
i=0
for P in "$P1" "$P2" "$P3" "$P4"; do i=$((i+1))
  timeout 300 rocprofv3 --pmc $P --output-format csv -d $O/pmc$i -o pmc -- python tools/ldebench.py --cols 64 --reps 3 > $O/pmc$i.log 2>&1 || echo "pmc pass $i failed: $(tail -2 $O/pmc$i.log)"
done
python tools/pmc_stalls.py $O/${TAG}_lde_stalls.json "ntt_passA_fwd12_multi_kernel,ntt_r16_kernel<false, false" $(find $O/pmc1 $O/pmc2 $O/pmc3 $O/pmc4 -name '*counter_collection.csv' 2>/dev/null)
rm -rf $O/pmc1 $O/pmc2 $O/pmc3 $O/pmc4
# 5. the headline, short
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-agent-mode --no-pcie-extra 2>/dev/null | tail -1 > $O/${TAG}_bench_short.json
python - <<PY
import json
j=json.load(open("$O/${TAG}_bench_short.json"))
print("value", j["value"], "roofline", {k: j["roofline"].get(k) for k in ("achieved","frac","avg_ms_per_launch","traffic")}, "single", j.get("single_proof_ms"))
PY
