#!/usr/bin/env python
"""bench.py — segment-proofs/sec at 2^20 cycles on N MI355X (BASELINE.json metric), one process per GPU.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" = one pass of the hot path over one synthetic 2^20-cycle segment per rank (BASELINE.json configs[1]:
"single 2^20-cycle segment on 1xMI355X via new HIP HAL (NTT+Poseidon2+FRI)"): witness fill -> 3 trace-group commits
(iNTT, zk_shift, 4x LDE, Poseidon2 Merkle) -> check-polynomial commit -> DEEP taps/mix/divide -> FRI (3 rounds) ->
50 queries -> seal, all through include/bx_prover.h.  Segments are independent, so ranks shard them with no
data-path collective ("scaling": "weak"); torch.distributed (RCCL) only provides the barriers and the max-over-ranks.
Inputs are generated on the device (no host buffers cross PCIe in the timed region except the seal and the
Fiat-Shamir digests, exactly as in the reference's prover).

The JSON line also carries
  roofline      the NTT/LDE entry point named by BASELINE's metric: algorithmic bytes / HIP-event time on the HAL
                stream, against the 8 TB/s HBM peak (DESIGN.md §4 explains why this path is VALU-issue-bound);
  kernels       the same for every HAL entry point in the timed region;
  cpu_baseline  the CPU oracle (kind "port": the reference's Rust CPU HAL cannot be built here) timed on this
                box's host cores on a bounded sample.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
VALU_PEAK_GOPS = 39321.6  # 256 CU x 4 SIMD x 16 lanes/clk x 2.4 GHz non-packed VALU lane-ops (measured ~37k, profiles/r01_microbench_valu.jsonl)


def cpu_baseline(po2_sample, widths, po2_full):
    """Time the CPU oracle's prover on a bounded sample and scale linearly in rows to the full segment."""
    from oracle import oracle_lib as ol

    path = None
    try:  # a -march=native build for this box's cores (the prebuilt .so is baseline x86-64)
        path = ol.build(force=True, native=True, out="/tmp/libbx_oracle_native.so")
    except Exception:
        path = None
    L = ol.lib(path) if path else ol.lib()
    cores = L.bxo_get_threads()
    ol.prove_segment(10, 2, 4, 2, 1, L)  # warm
    t0 = time.time()
    ol.prove_segment(po2_sample, *widths, 0xB0D1E550000, L)
    dt = time.time() - t0
    scale = 1 << (po2_full - po2_sample)
    return {
        "value": 1.0 / (dt * scale),
        "unit": "segment-proofs/s",
        "cores": cores,
        "kind": "port",
        "sample": f"one 2^{po2_sample}-cycle synthetic segment (widths {'/'.join(map(str, widths))}) proved by oracle/ "
                  f"(C, OpenMP, {cores} threads) in {dt:.2f}s; value = 1/(t * 2^{po2_full - po2_sample}) (linear in rows)",
        "sample_seconds": round(dt, 3),
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--po2", type=int, default=20)
    ap.add_argument("--widths", type=str, default="16,256,64")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-po2", type=int, default=16)
    args = ap.parse_args()
    widths = tuple(int(x) for x in args.widths.split(","))

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP HAL has no CPU fallback")
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))

    from boundless_amd.prover import HipProverServer, Segment

    server = HipProverServer(device=local_rank, po2=args.po2, widths=widths)
    hal = server.hal

    def barrier():
        torch.cuda.synchronize()
        hal.sync()
        if world > 1:
            dist.barrier()

    step_no = 0

    def one_step():
        nonlocal step_no
        seg = Segment.synthetic(index=step_no * world + rank, po2=args.po2)
        step_no += 1
        return server.prove_segment(seg)

    for _ in range(args.warmup):
        one_step()
    barrier()
    hal.profile_reset()
    hal.profile_enable(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        receipt = one_step()
    barrier()
    elapsed = time.perf_counter() - t0
    hal.profile_enable(False)
    prof = hal.profile_report()

    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        kernels = {}
        for name, r in prof.items():
            ms = r["ms"] / max(r["calls"], 1)
            gbps = r["alg_bytes"] / max(r["calls"], 1) / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
            kernels[name] = {"calls_per_step": r["calls"] / args.steps, "avg_ms": round(ms, 4),
                             "ms_per_step": round(r["ms"] / args.steps, 3), "alg_GBps": round(gbps, 1),
                             "frac_hbm": round(gbps / HBM_PEAK_GBPS, 4)}
        ntt = kernels.get("batch_expand_into_evaluate_ntt", {})
        dom_name = max(kernels, key=lambda k: kernels[k]["ms_per_step"]) if kernels else None
        out = {
            "metric": "segment-proofs/sec @ 2^20 cycles",
            "value": world * args.steps / elapsed,
            "unit": "segment-proofs/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u32 (BabyBear Montgomery)",
            "data": "synthetic",
            "config": {"workload": f"single 2^{args.po2}-cycle synthetic segment per GPU via the HIP HAL (NTT+Poseidon2+FRI), "
                                   f"trace widths code/data/accum = {'/'.join(map(str, widths))}, check 16, 50 queries",
                       "po2": args.po2, "parallelism": f"segments sharded over {world} GPU(s), no collective",
                       "khz_equiv": world * args.steps * (1 << args.po2) / elapsed / 1e3},
            "seal_words": int(receipt.seal.size),
            "roofline": {
                "kernel": "batch_expand_into_evaluate_ntt (ntt_block_kernel + ntt_strided_kernel, 4x LDE)",
                "bound": "hbm",
                "achieved": ntt.get("alg_GBps"),
                "peak": HBM_PEAK_GBPS,
                "unit": "GB/s",
                "frac": ntt.get("frac_hbm"),
                "traffic": None,
                "note": "algorithmic bytes = 4B*(in + out) words per call / HIP-event time on the HAL stream",
            },
            "roofline_dominant": {"kernel": dom_name, **(kernels.get(dom_name, {}) if dom_name else {}),
                                  "note": "Poseidon2 is VALU-issue-bound (no HBM or MFMA roofline applies); see DESIGN.md §4"},
            "kernels": kernels,
        }
        if world == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(min(args.cpu_sample_po2, args.po2), widths, args.po2)
            except Exception as e:  # the baseline is reported, never required for the GPU number
                out["cpu_baseline"] = {"error": str(e)}
        print(json.dumps(out))
    server.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
