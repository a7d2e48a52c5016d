#!/bin/bash
# Round-6 profile set (run on the GPU box from the repo root AFTER the final library build; results under gpurun_out/r6p/, the
# summaries are copied to profiles/ at the end).  Every summary is stamped with the SHA-256 of the library's device code
# (boundless_amd.build.device_code_hash), which is what bench.py's `profile_stale` compares.
set -u
O=gpurun_out/r6p; mkdir -p $O
export TMPDIR=/tmp
B="python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-agent-mode --no-pcie-extra --no-plain-hal"
# 1. per-kernel time: the default command (3 segments in flight) and one segment in flight
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt3 -o kt -- $B --no-live-profile > $O/bench_kt3.json 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt1 -o kt -- $B --inflight 1 --no-live-profile > $O/bench_kt1.json 2>/dev/null
cp "$(find $O/kt3 -name '*kernel_stats.csv' | head -1)" $O/r06_bench_kernel_stats_default_cmd.csv
cp "$(find $O/kt1 -name '*kernel_stats.csv' | head -1)" $O/r06_bench_kernel_stats_inflight1.csv
python tools/latency_gaps.py "$(find $O/kt1 -name '*kernel_trace.csv' | head -1)" $O/r06_latency_gaps.json > /dev/null
rm -rf $O/kt3 $O/kt1
# 2. HBM-side traffic per kernel (separate counter passes, no tracing domains), one segment in flight
for ctr in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $ctr --output-format csv -d $O/pmc_$ctr -o pmc -- $B --inflight 1 > /dev/null 2>&1
done
python tools/pmc_traffic.py "$(find $O/pmc_FETCH_SIZE -name '*counter_collection.csv' | head -1)" "$(find $O/pmc_WRITE_SIZE -name '*counter_collection.csv' | head -1)" $O/r06_bench_pmc_traffic.json
rm -rf $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE
# 3. VALU instructions of the whole job, per segment (the tool counts the proofs of the run by their eval_check launches)
rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES --output-format csv -d $O/pmc_valu -o pmc -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-agent-mode --no-pcie-extra --no-plain-hal > /dev/null 2>&1
python tools/job_valu.py "$(find $O/pmc_valu -name '*counter_collection.csv' | head -1)" auto $O/r06_job_valu_insts.json
rm -rf $O/pmc_valu
# 4. the stall side of the two LDE kernels, bench.py with one segment in flight (VERDICT r04 item 2 ii): separate passes of <= 8 SQ counters
P1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU"
P2="SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"
i=0
for P in "$P1" "$P2"; do i=$((i+1))
  timeout 600 rocprofv3 --pmc $P --output-format csv -d $O/pmcs$i -o pmc -- $B --inflight 1 > /dev/null 2>&1 || echo "stall pass $i failed"
done
python tools/pmc_stalls.py $O/r06_lde_stall_counters.json "ntt_passA_fwd12_multi_kernel,ntt_r16_kernel<false, false,hash_rows_kernel" $(find $O/pmcs1 $O/pmcs2 -name '*counter_collection.csv' 2>/dev/null)
rm -rf $O/pmcs1 $O/pmcs2
#    ... and the clock each kernel runs at under the package power limit (GRBM_GUI_ACTIVE over the 8 XCDs / dispatch duration)
rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d $O/clk -o pmc -- $B --inflight 1 > /dev/null 2>&1
python tools/pmc_clock.py "$(find $O/clk -name '*counter_collection.csv' | head -1)" $O/r06_kernel_clocks.json "ntt_,hash_rows,eval_check,hash_fold_deep,witness_derive" > /dev/null
rm -rf $O/clk
# 5. one planned job of 64 segments (stand-in joins: 2^18 synthetic proofs, NOT recursion proofs), and the world-size-8 dry runs on the one GPU
for l in 3 1; do python bench.py --job 64 --inflight $l 2>/dev/null | tail -1; done > $O/r06_job64.jsonl
python -m torch.distributed.run --nnodes=1 --nproc-per-node=8 --master-addr 127.0.0.1 --master-port 29651 bench.py --gpus 8 --dist-backend gloo --device 0 --inflight 1 --steps 3 --warmup 1 2>/dev/null | grep '^{' | tail -1 > $O/r06_ws8_gloo_1gpu.json
python -m torch.distributed.run --nnodes=1 --nproc-per-node=8 --master-addr 127.0.0.1 --master-port 29652 bench.py --gpus 8 --dist-backend gloo --device 0 --inflight 1 --job 64 2>/dev/null | grep '^{' | tail -1 > $O/r06_ws8_gloo_1gpu_job64.json
# 6. the trait-level proof (tests/plain_hal_prover.c) beside bx_prove_segment, one extension at a time, three in flight
python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-agent-mode --no-pcie-extra 2>/dev/null | tail -1 > $O/b_plain.json
O=$O python - <<'PY'
import json, os
O = os.environ["O"]
j = json.load(open(O + "/b_plain.json"))
out = {"what": "single_proof_ms of python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-agent-mode --no-pcie-extra (untimed extra, rank 0): one lone 2^20 / 16-256-64 "
               "proof by bx_prove_segment (min/median/spin_wait) and by tests/plain_hal_prover.c - the plain Hal entry points of SURVEY 8(b2) only, sequenced "
               "outside the library - then with ONE extension entry point swapped in at a time, and three such drivers in flight",
       "device_code_sha": j["replayed_profiles"]["device_code_sha"], "value_proofs_per_s": j["value"], "single_proof_ms": j["single_proof_ms"]}
json.dump(out, open(O + "/r06_plain_hal.json", "w"), indent=1)
PY
BX_TUNABLES=gather_defer=0 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-agent-mode --no-pcie-extra 2>/dev/null | tail -1 | python -c "
import json,sys; j=json.loads(sys.stdin.read()); p=j['single_proof_ms']['plain_hal']; print(json.dumps({'gather_defer': 0, 'plain_hal_min_ms': p['min'], 'calls': p['calls'], 'three_in_flight': p.get('three_in_flight'), 'bx_prove_segment_min_ms': j['single_proof_ms']['min']}))" > $O/r06_plain_hal_gather_defer0.json
# 6b. rocprofv3 per-kernel summary of the trait-level proof, gather_sample queue on / off (7 gather_batch_kernel launches per proof against 5 100 gather_sample_kernel launches)
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt1 -o kt -- python tools/plain_hal_run.py 5 > $O/run_defer1.txt 2>/dev/null
BX_TUNABLES=gather_defer=0 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt0 -o kt -- python tools/plain_hal_run.py 5 > $O/run_defer0.txt 2>/dev/null
cp "$(find $O/kt1 -name '*kernel_stats.csv' | head -1)" $O/r06_plain_hal_kernel_stats.csv
cp "$(find $O/kt0 -name '*kernel_stats.csv' | head -1)" $O/r06_plain_hal_kernel_stats_gather_defer0.csv
rm -rf $O/kt1 $O/kt0
cat $O/run_defer1.txt $O/run_defer0.txt; grep -i "gather" $O/r06_plain_hal_kernel_stats.csv $O/r06_plain_hal_kernel_stats_gather_defer0.csv | cut -c1-220
# inflight sweep
for l in 1 2 3 4; do python bench.py --steps 8 --warmup 3 --inflight $l --no-cpu-baseline --no-agent-mode --no-pcie-extra --no-plain-hal 2>/dev/null | tail -1 | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print(json.dumps({'inflight': $l, 'value': round(j['value'],3), 'ms_per_step': j['ms_per_step'], 'single_proof_ms': j['single_proof_ms']['min']}))"; done > $O/r06_inflight_sweep.jsonl
# 7. LAST: the bench line of the driver's command, replaying the PMC summaries just collected
cp $O/r06_bench_pmc_traffic.json $O/r06_job_valu_insts.json profiles/
python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 > $O/r06_bench_steps20_warmup5.json
head -14 $O/r06_bench_kernel_stats_default_cmd.csv | cut -c1-150
python - <<PY
import json
j = json.load(open("$O/r06_bench_steps20_warmup5.json"))
print("value", j["value"], "stale", j["replayed_profiles"]["profile_stale"], "roofline", {k: j["roofline"].get(k) for k in ("achieved", "frac", "avg_ms_per_launch", "traffic")})
print(json.dumps(j["roofline"].get("valu_view"), indent=1))
PY
# the default command (no flags), as the driver may run it
python bench.py 2>/dev/null | tail -1 > $O/r06_bench_default_cmd.json
cp $O/r06_*.json $O/r06_*.jsonl $O/r06_*.csv profiles/ 2>/dev/null
