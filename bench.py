#!/usr/bin/env python
"""bench.py — segment-proofs/sec at 2^20 cycles on N MI355X (BASELINE.json metric), one process per GPU.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" = one pass of the hot path over one synthetic 2^20-cycle segment per lane and rank (BASELINE.json configs[1]:
"single 2^20-cycle segment on 1xMI355X via new HIP HAL (NTT+Poseidon2+FRI)"): witness generation (free cells, scatter-placed
permuted copies, derived columns) -> code/data commits (iNTT, zk_shift, 4x LDE, Poseidon2 Merkle) -> accumulate
(prefix_products) -> accum commit -> eval_check over the 4N domain -> check commit -> DEEP taps/mix/divide -> FRI (3 rounds) ->
50 queries -> seal, all through include/bx_prover.h.  The circuit is the SYNTHETIC AIR of include/bx_prover.h (not rv32im: the
generated circuit is not in the reference tree), so `value` is the rate of complete STARK proofs of that circuit and is not
comparable to upstream's effective kHz; `config.circuit` says what is and is not included.  Segments are independent, so ranks
shard them with no data-path collective ("scaling": "weak"); torch.distributed (RCCL) only provides the barriers and the
max-over-ranks.  Inputs are generated on the device (no host buffers cross PCIe in the timed region except the seal and the
Fiat-Shamir digests, exactly as in the reference's prover).

By default three segments are in flight per GPU (one prover, stream and host thread each): the others fill the
latency-bound tails (small Merkle layers, Fiat-Shamir round trips) of the first, +15 % throughput; a step is then one
batch of `--inflight` segments per GPU and `value` counts segments.

Other modes (never the default line): `--segment-bytes N [--two-deep]` makes every segment arrive as N host bytes (28-byte header +
payload) that are copied into a pinned staging slot and uploaded on the prover's copy stream inside the timed region (the
reference's 2^20-cycle segment is ~80 MB, executor.rs:45; `--two-deep` = a feeder thread per lane submits segment k+1 while segment
k is proved); `--job K` proves ONE planned job of K segments through the native agent — K Prove tasks, the log-depth tail of Join
tasks (labelled synthetic stand-ins), resolve, finalize — and reports prove-phase rate, join-tail latency and end-to-end seconds
(under torchrun: one process per GPU, every rank joins its own subtree, ONE all_gather of the subtree roots — RCCL under nccl — and
rank 0 joins them: the job's only collective);
`--native-agent` = the one-process multi-GPU design.

The JSON line also carries
  single_proof_ms   wall clock of a lone proof (the reference's agent proves one segment at a time per process), no HIP events around it;
                    .spin_wait = the same with the host thread busy-waiting on its stream (the one-proof-at-a-time latency mode);
                    .plain_hal = the same segment proved from OUTSIDE the library through the plain Hal-trait entry points of SURVEY 8(b2)
                    only (tests/plain_hal_prover.c: what a Rust `impl Hal for HipHal` shim driven by risc0-zkp's prover gets; seal compared with
                    bx_prove_segment's), then with each extension entry point swapped in alone, with three such drivers in flight, and with
                    the ~9 GB of buffers allocated inside every proof as upstream does (`alloc_per_proof`: through the library's bx_alloc pool,
                    and on raw hipMalloc / hipFree);
  pcie_inclusive    untimed extra at N=1: the same workload with 80 MB per segment over PCIe, two deep (`value` stays inputs-resident);
  per_rank, backend one row per rank (rank, device, proofs, seconds) and the backend's world size: whether RCCL saw N ranks, and balance;
  roofline      the NTT/LDE entry point named by BASELINE's metric (`roofline.dominant` = the job's dominant kernel, hash_rows, against
                the VALU issue peak; `roofline.traffic_over_algorithmic` = PMC bytes / algorithmic bytes of the LDE): algorithmic bytes / HIP-event time on the HAL
                stream over the timed region, against the 8 TB/s HBM peak (DESIGN.md §4 explains why this path is
                VALU-issue-bound); measured by an isolated probe (one segment alone) because concurrent streams stretch the
                in-region durations; `roofline_in_region` is the concurrent figure;
                `roofline.valu_view` = the same launch seen from the VALU: instructions per wave (PMC) and their static class mix
                (profiles/r05_lde_isa_mix.json), `frac_weighted_issue` (the class-weighted bound, 2 / 4 cycles per instruction — not
                attainable in a mixed stream) next to `frac_of_mixed_stream_rate` (against the rate a bare loop of the same
                butterflies issues at, profiles/r05_microbench4_operand_kinds.jsonl), and `traffic_GBps` on counter bytes;
  replayed_profiles  which committed PMC summaries the line replays and `profile_stale`: whether they were collected on the DEVICE CODE
                the library that just ran carries (SHA-256 of its .hip_fatbin section, boundless_amd.build.device_code_hash);
  kernels       the same for every HAL entry point in the timed region, from the HIP events of ONE lane per rank (bracketing
                every entry point of every lane costs 0.9 % of the rate; `live_profile` says what was bracketed);
  cpu_baseline  the CPU oracle (kind "port": the reference's Rust CPU HAL cannot be built here) timed on this
                box's host cores: one proof of the metric's own size (2^20, ~35 s), thread count chosen on a small probe within the
                container's CPU quota (16 on the GPU boxes).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

ISO_SEGMENTS = 3  # segments proved one at a time by the isolated probe behind `roofline` / `kernels_isolated`
PROFILE_ROUND = "r06"  # committed PMC summaries this line refers to (profiles/<round>_*.json); falls back to r01's
HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
# VALU issue model used throughout (MI355X_MICROARCH.md: 256 CUs x 4 SIMD-32 units; measured in profiles/r01_microbench2_instr_cost.jsonl):
# a wave64 instruction of the cheap integer class issues in 2 cycles per SIMD, one of the multiply class (v_mul_lo/hi_u32, v_mad_u64_u32,
# v_mad_i64_i32) in 4, so the chip's issue peaks are 1024 SIMDs x 2.4 GHz / 2 and / 4 wave-instructions per second.
VALU_PEAK_GOPS = 39321.6  # = 1024 SIMDs x 2.4 GHz / 4 cycles x 64 lanes: multiply-class lane-ops per second (informational)


_replayed = {}  # profile file -> device-code stamp it was collected on (None when the file predates the stamps)


def _profile(name):
    """profiles/<PROFILE_ROUND>_<name>, or an earlier round's file while this round's has not been collected yet"""
    for rnd in (PROFILE_ROUND, "r05", "r04", "r03", "r02", "r01"):
        p = os.path.join(ROOT, "profiles", f"{rnd}_{name}")
        if os.path.exists(p):
            return p
    raise FileNotFoundError(name)


def _load_profile(name):
    """A committed PMC summary this line replays; remembers which device code it was collected on (boundless_amd.build.device_code_hash)."""
    path = _profile(name)
    j = json.load(open(path))
    _replayed[os.path.basename(path)] = j.get("device_code_sha") if isinstance(j, dict) else None
    return j


def replayed_profiles():
    """Which figures of this line were not measured in this run, and whether the library that ran carries the device code they describe."""
    from boundless_amd.build import csrc_hash, device_code_hash

    cur = device_code_hash()
    return {"device_code_sha": cur, "library_sha": csrc_hash(device_only=False), "files": dict(_replayed),
            "profile_stale": any(v != cur for v in _replayed.values()) if _replayed else False,
            "note": "roofline.traffic, valu_view and roofline_job replay rocprofv3 --pmc passes committed under profiles/ (counters cannot be "
                    "collected inside a timed run); the stamp is the SHA-256 of the library's .hip_fatbin section (the gfx950 code objects): "
                    "profile_stale = at least one replayed file was collected on other DEVICE CODE than the library that just ran carries; a "
                    "host-only edit leaves it unchanged (tests/test_profile_stamp_cpu.py); library_sha = hash of all source text, host side included"}


def relaunch_one_process_per_gpu(n):
    """exec `python -m torch.distributed.run --nnodes=1 --nproc-per-node n --master-addr 127.0.0.1 --master-port <free> bench.py <same args>`"""
    import socket

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.abspath(__file__)] + sys.argv[1:]
    sys.stderr.write(f"bench.py: --gpus {n} without a launcher: re-executing as one process per GPU: {' '.join(cmd[1:])}\n")
    sys.stderr.flush()
    os.execv(sys.executable, cmd)


def rendezvous_only(args):
    """Test hook (tests/test_dist_cpu.py): rendezvous, one all-reduce that counts the ranks, rank 0 prints what the launcher gave it;
    needs no GPU with --dist-backend gloo.  Proves that `--gpus N` reached N processes."""
    import torch

    from boundless_amd.dist import init_distributed, sum_over_ranks

    rank, world, local_rank, dist = init_distributed(args.dist_backend, force=args.force_dist)
    seen = int(round(sum_over_ranks(1, dist)))
    if dist is not None:
        dist.barrier()
    if rank == 0:
        print(json.dumps({"rendezvous_only": True, "n_gpus": world, "ranks_counted": seen, "flag_gpus": args.gpus,
                          "backend": None if dist is None else dist.get_backend()}), flush=True)
    if dist is not None:
        dist.destroy_process_group()
    return 0


def plain_hal_probe(args, sv, product_ms):
    """Untimed extra (rank 0): wall clock of one lone proof sequenced by the test-side driver tests/plain_hal_prover.c over the plain
    entry points of SURVEY.md section 8(b2) — one bx_hash_fold per layer, bx_poly_divide per combo and point, bx_batch_evaluate_any
    per group, separate bx_zk_shift, bx_gather_sample per opened row and path digest — and the same with ONE extension entry point
    swapped in at a time.  The driver has its own host transcript; its seal must equal bx_prove_segment's."""
    import numpy as np

    from boundless_amd.prover import Segment

    tests = os.path.join(ROOT, "tests")
    if tests not in sys.path:
        sys.path.insert(0, tests)
    try:
        import plain_hal
    except Exception as e:  # no gcc on this box: say so instead of failing the line
        return {"error": f"tests/plain_hal_prover.c could not be built: {e}"}
    seg = Segment.synthetic(index=4 * 10**6, po2=args.po2)
    want = sv.prove_segment(seg).seal

    def make(flags, tunables=None):
        """a driver (its own ctx); `tunables` = BX_TUNABLES for that ctx alone (read by bx_init)"""
        old = os.environ.get("BX_TUNABLES")
        if tunables:
            os.environ["BX_TUNABLES"] = (old + "," if old else "") + tunables
        try:
            return plain_hal.PlainHalProver(sv.hal.device, po2=args.po2, widths=tuple(int(x) for x in args.widths.split(",")),
                                            terms=args.terms, degree=args.degree, flags=flags)
        finally:
            if tunables:
                if old is None:
                    del os.environ["BX_TUNABLES"]
                else:
                    os.environ["BX_TUNABLES"] = old

    def timed(flags, tunables=None):
        pp = make(flags, tunables)
        try:
            ts, equal = [], True
            for _ in range(4):
                seal, ms = pp.prove(seg.seed)
                ts.append(ms)
                equal = equal and bool(np.array_equal(seal, want))
            return {"min": round(min(ts[1:]), 3), "median": round(sorted(ts[1:])[1], 3), "calls": int(pp.calls), "seal_equals_bx_prove_segment": equal}
        finally:
            pp.close()

    out = timed(0)
    out["ratio_to_bx_prove_segment"] = round(out["min"] / product_ms, 3) if product_ms else None
    ext = {}
    for flag, name in sorted(plain_hal.EXT_NAMES.items()):
        r = timed(flag)
        r["saves_ms"] = round(out["min"] - r["min"], 3)
        ext[name] = r
    out["with_one_extension"] = ext
    out["with_all_extensions"] = timed(plain_hal.EXT_ALL)
    # throughput of the trait-level path with three provers in flight (three drivers, three ctxs, three host threads — what a Rust
    # process holding three `HipHal` objects could do; the reference's agent holds one, bento/crates/workflow/src/lib.rs:192)
    import threading

    lanes, per_lane = 3, 4

    def in_flight(flags, tunables=None):
        pps = [make(flags, tunables) for _ in range(lanes)]
        try:
            for pp in pps:
                pp.prove(seg.seed)  # warm
            ok = [True] * lanes

            def work(k):
                for j in range(per_lane):
                    seal, _ = pps[k].prove(seg.seed)
                    ok[k] = ok[k] and bool(np.array_equal(seal, want))

            ts = [threading.Thread(target=work, args=(k,)) for k in range(lanes)]
            t0 = time.perf_counter()
            [t.start() for t in ts]
            [t.join() for t in ts]
            dt = time.perf_counter() - t0
            return {"proofs_per_s": round(lanes * per_lane / dt, 3), "proofs": lanes * per_lane, "seals_equal": all(ok)}
        finally:
            for pp in pps:
                pp.close()

    out["three_in_flight"] = in_flight(0)
    # risc0-zkp's prover allocates its buffers inside every proof (hal.alloc_* in commit_group / finalize / fri_prove) and drops them
    # at its end: the same proof with the ~9 GB of big buffers allocated and released inside the timed call — through the library's
    # per-ctx pool (default), and with the pool off (alloc_cache_mb = 0: hipMalloc / stream wait + hipFree, and hipFree drains the
    # whole device, the other lanes' streams included)
    app = plain_hal.ALLOC_PER_PROOF
    out["alloc_per_proof"] = {"pooled": timed(app), "hipMalloc_hipFree": timed(app, "alloc_cache_mb=0"),
                              "three_in_flight_pooled": in_flight(app), "three_in_flight_hipMalloc_hipFree": in_flight(app, "alloc_cache_mb=0")}
    out["note"] = ("one lone proof through the section-8(b2) entry points only, sequenced outside the library with its own host transcript; "
                   "with_one_extension = the same with that extension entry point replacing its plain call sequence; every device buffer but "
                   "`combos` (bx_alloc_zeroed per proof) is allocated up front, which risc0-zkp's prover does not do")
    return out


def _valu_per_wave():
    """VALU instructions per wave, per kernel name: this round's job-level PMC pass, else round 1's opbench pass"""
    try:
        return _load_profile("job_valu_insts.json")["valu_insts_per_wave"]
    except Exception:
        k = json.load(open(os.path.join(ROOT, "profiles", "r01_kernel_valu_counts.json")))["kernels"]
        return {name: v["valu_insts_per_wave"] for name, v in k.items() if "valu_insts_per_wave" in v}


def cpu_baseline(po2_sample, widths, po2_full):
    """Time the CPU oracle's prover (kind "port": the reference's Rust CPU HAL cannot be built here) on this box's host
    cores.  The thread count is picked on a small 2^14 proof (the oracle's parallel regions are short, so more threads is
    not always faster), then ONE proof of the sample size is timed with it.  By default the sample IS the metric's config
    (2^20 cycles, same widths, same circuit): `value` is then measured, not extrapolated; a smaller --cpu-sample-po2 is
    scaled linearly in rows and says so."""
    from oracle import oracle_lib as ol

    path = None
    try:  # a -march=native build for this box's cores (the prebuilt .so is baseline x86-64)
        path = ol.build(force=True, native=True, out="/tmp/libbx_oracle_native.so")
    except Exception:
        path = None
    L = ol.lib(path) if path else ol.lib()
    ol.prove_segment(10, 2, 4, 2, 1, L)  # warm
    ncpu = os.cpu_count() or 1
    quota = None  # the container's CPU quota (cgroup v2 cpu.max = "<max> <period>"): threads beyond it only get throttled
    try:
        mx, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if mx != "max":
            quota = max(1, int(mx) // int(period))
    except (OSError, ValueError):
        quota = None
    usable = min(ncpu, quota) if quota else ncpu
    probe_po2 = min(14, po2_sample)
    best = None
    for threads in sorted({min(usable, 64), min(usable, 32), min(usable, 16)}, reverse=True):
        L.bxo_set_threads(threads)
        t0 = time.time()
        ol.prove_segment(probe_po2, *widths, 0xB0D1E550000, L)
        dt_try = time.time() - t0
        if best is None or dt_try < best[0]:
            best = (dt_try, threads)
    cores = best[1]
    L.bxo_set_threads(cores)
    t0 = time.time()
    ol.prove_segment(po2_sample, *widths, 0xB0D1E550000, L)
    dt = time.time() - t0
    scale = 1 << (po2_full - po2_sample)
    how = "measured at the metric's size" if scale == 1 else f"value = 1/(t * 2^{po2_full - po2_sample}) (scaled linearly in rows)"
    return {
        "value": 1.0 / (dt * scale),
        "unit": "segment-proofs/s",
        "cores": cores,
        "kind": "port",
        "sample": f"one 2^{po2_sample}-cycle synthetic segment (widths {'/'.join(map(str, widths))}, default circuit) proved by oracle/ "
                  f"(C, OpenMP, {cores} threads chosen on a 2^{probe_po2} probe; the box has {ncpu} hardware threads"
                  f"{f' and this container a CPU quota of {quota} (cgroup cpu.max)' if quota else ''}) in {dt:.2f}s; {how}",
        "cpu_quota": quota,
        "sample_seconds": round(dt, 3),
    }


def job_valu_view(segments_per_s_per_gpu):
    """Whole-job VALU issue rate: wave-level VALU instructions per segment (PMC, profiles/r02_job_valu_insts.json, all
    kernels of the default workload) x the measured segment rate of one GPU, against the multiply-class issue peak."""
    try:
        j = _load_profile("job_valu_insts.json")
        rate = j["per_segment"] * segments_per_s_per_gpu
        return {"valu_wave_insts_per_segment": j["per_segment"], "wave_insts_per_s_per_gpu": rate,
                "issue_peak_mul": 1024 * 2.4e9 / 4, "frac_of_mul_class_peak": round(rate / (1024 * 2.4e9 / 4), 3),
                "ns_per_wave_inst_per_simd": round(1024 / rate * 1e9, 3),
                "note": "a multiply-class instruction measures 1.93-1.96 ns per wave and SIMD, a plain add 1.1 ns "
                        "(profiles/r01_microbench3_mad_forms.jsonl): the job average sits at the multiply-class cost"}
    except Exception:
        return None


def agent_mode(args, widths, device, lanes):
    """Segments/s through the native prove agent (include/bx_agent.h) over the in-memory hot store and task db."""
    out = {}
    n = max(2, args.steps) * lanes
    for verify in (True, False):
        out["verify_on" if verify else "verify_off"] = _agent_run(args, widths, [device], lanes, verify, n)
    out["lanes"] = lanes
    out["note"] = ("untimed extra: tasks claimed from the in-memory task db by bx_agent_poll_work; verify_on includes the CPU "
                   "seal verification the reference runs after every prove (prove.rs:53-55); host_cpu_s_per_proof = process CPU time / proofs")
    return out


def _agent_run(args, widths, devices, lanes, verify, segments):
    """`segments` synthetic segments through bx_agent_poll_work on `devices` x `lanes` lanes; returns the measurement."""
    import resource

    from boundless_amd import agent as ag
    from boundless_amd.prover import Segment

    a = ag.Agent(prover=None, device=devices[0], devices=devices if len(devices) > 1 else None, inflight=lanes, widths=widths,
                 poll_time=0.001, verify=verify, terms=args.terms, degree=args.degree)
    try:
        pad = bytes(max(0, args.segment_bytes - 28))  # --segment-bytes: every stored segment carries this payload (uploaded per proof)

        def enqueue(job, n):
            for i in range(n):
                seg = Segment.synthetic(i, po2=args.po2)
                seg.payload = pad
                a.store.set_key_with_expiry(f"job:{job}:segments:{i}", ag.serialize_segment(seg), 600)
                a.taskdb.create_task(job, f"prove-{i}", {"Prove": {"index": i}})

        enqueue("warm", max(1, args.warmup) * lanes * len(devices))
        a.poll_work(max_idle_polls=1)
        base = {d: 0 for d in devices}
        for d, n in a.lane_stats():
            base[d] = base.get(d, 0) + n
        enqueue("timed", segments)
        ru0 = resource.getrusage(resource.RUSAGE_SELF)
        t0 = time.perf_counter()
        done = a.poll_work(max_idle_polls=1)
        dt = time.perf_counter() - t0
        ru1 = resource.getrusage(resource.RUSAGE_SELF)
        if done != segments:
            raise RuntimeError(f"agent completed {done} of {segments} tasks")
        per_dev = {d: -base[d] for d in devices}
        for d, n in a.lane_stats():
            per_dev[d] = per_dev.get(d, 0) + n
        cpu = (ru1.ru_utime - ru0.ru_utime) + (ru1.ru_stime - ru0.ru_stime)
        return {"segment_proofs_per_s": segments / dt, "segments": segments, "seconds": dt, "host_cpu_s_per_proof": round(cpu / segments, 5),
                "cpus_busy_avg": round(cpu / dt, 3), "segments_per_device": {str(d): int(n) for d, n in per_dev.items()}}
    finally:
        a.close()


def native_agent_main(args, widths):
    """`--native-agent`: the second multi-GPU design.  ONE process, no torch.distributed: the native agent
    (include/bx_agent.h, csrc/agent.cpp) runs --inflight prover lanes on each of --gpus devices, every lane claiming from the
    one task db (the reference's request_work queue is the work-stealing queue), each seal CPU-verified as the reference
    does after every prove (prove.rs:53-55).  Prints the same JSON line; `value` = segments/s through the whole feed loop.
    Under torchrun only rank 0 does anything."""
    if int(os.environ.get("RANK", "0")) != 0:
        return
    import torch

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP HAL has no CPU fallback")
    n = args.gpus
    if args.device is None and torch.cuda.device_count() < n:
        raise SystemExit(f"--native-agent --gpus {n}: only {torch.cuda.device_count()} device(s) visible")
    lanes = max(1, args.inflight)
    devices = list(range(n)) if args.device is None else [args.device] * n
    segments = args.steps * lanes * n
    r = _agent_run(args, widths, devices, lanes, True, segments)
    out = {"metric": "segment-proofs/sec @ 2^20 cycles", "value": r["segment_proofs_per_s"], "unit": "segment-proofs/s", "n_gpus": n,
           "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * r["seconds"] / args.steps, "higher_is_better": True,
           "scaling": "weak", "vs_baseline": None, "dtype": "u32 (BabyBear Montgomery)", "data": "synthetic",
           "config": {"workload": f"2^{args.po2}-cycle synthetic segments through the native prove agent (claim -> hot-store GET -> prove on the GPU -> "
                                  f"CPU verify -> SETEX -> UNLINK -> done), trace widths {'/'.join(map(str, widths))}", "po2": args.po2,
                      "segments_proved": segments, "segments_in_flight_per_gpu": lanes, "segment_bytes": max(28, args.segment_bytes),
                      "queue": "one in-memory task db shared by every lane of every device (claim-when-idle)",
                      "parallelism": f"one process, {n} device(s) x {lanes} lanes, no collective, no torch.distributed"},
           "host_cpu_s_per_proof": r["host_cpu_s_per_proof"],
           "host": {"cpus_busy_avg": r["cpus_busy_avg"], "wait_policy": os.environ.get("BX_WAIT", "poll (library default)"),
                    "cpus_allowed": len(os.sched_getaffinity(0)), "includes": "lane threads, finisher threads (CPU verification of every seal), the in-memory stores"},
           "segments_per_device": r["segments_per_device"]}
    print(json.dumps(out))


def job_main(args, widths):
    """`--job K`: BASELINE configs[2]/[3]'s SHAPE — K segment proofs, then the log-depth tail of joins, resolve, finalize — as the
    planner's DAG (bx_plan_job) through the native agent's lanes on --gpus devices, ONE process.  The joins are STAND-INS (one
    synthetic 2^--join-po2 proof seeded by the hash of the two children's seals; the recursion circuit is not available offline), so
    the line is labelled `"join": "synthetic stand-in"`: what it measures is the scheduling shape — how long the K proves keep N
    GPUs busy, and how long the join tail, which cannot, takes."""
    if int(os.environ.get("WORLD_SIZE", "1")) > 1 or (args.force_dist and "MASTER_PORT" in os.environ):
        return job_dist_main(args, widths)
    import torch

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP HAL has no CPU fallback")
    from boundless_amd import agent as ag
    from boundless_amd.prover import Segment

    n = args.gpus
    if args.device is None and torch.cuda.device_count() < n:
        raise SystemExit(f"--job --gpus {n}: only {torch.cuda.device_count()} device(s) visible")
    lanes = max(1, args.inflight)
    devices = list(range(n)) if args.device is None else [args.device] * n
    K = args.job
    a = ag.Agent(prover=None, device=devices[0], devices=devices if len(devices) > 1 else None, inflight=lanes, widths=widths, poll_time=0.001,
                 verify=True, terms=args.terms, degree=args.degree, join_po2=args.join_po2, also_streams="aux", max_shapes=2,
                 lift_po2=args.join_po2 if args.lift else 0)
    try:
        def submit(job, k):
            for i in range(k):
                a.store.set_key_with_expiry(f"job:{job}:segments:{i}", ag.serialize_segment(Segment.synthetic(i, po2=args.po2)), 600)
            return a.taskdb.plan_job(job, k)

        a.prewarm(args.po2)       # every lane creates both buffer sets (segment size and join size) before the clock starts,
        a.prewarm(args.join_po2)  # as a deployment that knows its --segment-po2 does at start-up
        submit("warm", 2 * lanes * n)
        a.poll_work(max_idle_polls=2)
        base = {}
        for d, cnt in a.lane_stats():
            base[d] = base.get(d, 0) + cnt
        if args.jobs > 1:
            def make_aux():  # the deployment's aux agent (compose.yml: `agent -t aux`): one lane that serves the finalize tasks as they become ready
                return ag.Agent(no_prover=True, inflight=1, poll_time=0.001, verify=True, store=a.store, taskdb=a.taskdb, task_stream="aux")
            return batch_of_jobs(args, a, submit, K, n, lanes, widths, make_aux)
        ids = submit("timed", K)
        t0 = time.perf_counter()
        done = a.poll_work(max_idle_polls=2)
        dt = time.perf_counter() - t0
        if done != len(ids) or a.taskdb.job("timed")["state"] != "done":
            raise RuntimeError(f"job: {done} of {len(ids)} tasks done, state {a.taskdb.job('timed')}")
        rows = {t: a.taskdb.task("timed", t) for t in ids}
        proves = [rows[t] for t in ids if t not in ("resolve", "finalize") and rows[t].output is not None and int(t) in _prove_ids(K)]
        joins = [rows[t] for t in ids if t not in ("resolve", "finalize") and int(t) not in _prove_ids(K)]
        t_first = min(r.started_s for r in proves)
        t_proves = max(r.updated_s for r in proves)
        t_end = rows["finalize"].updated_s
        if args.dump:
            import numpy as np

            rollup = ag.deserialize_receipt(a.store.get("receipts/stark/timed.synthetic"))
            os.makedirs(args.dump, exist_ok=True)
            np.savez(os.path.join(args.dump, "rollup.npz"), seal=rollup.seal, po2=rollup.po2)
        per_dev = {d: -c0 for d, c0 in base.items()}
        for d, cnt in a.lane_stats():
            per_dev[d] = per_dev.get(d, 0) + cnt
        out = {"metric": "segment-proofs/sec @ 2^20 cycles", "value": K / (t_proves - t_first), "unit": "segment-proofs/s", "n_gpus": n,
               "steps": 1, "warmup": 1, "ms_per_step": 1e3 * dt, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
               "dtype": "u32 (BabyBear Montgomery)", "data": "synthetic", "join": "synthetic stand-in",
               "lift": ("synthetic stand-in (one more 2^%d-cycle proof per Prove task)" % args.join_po2) if args.lift else "none (the segment's own seal is stored)",
               "config": {"workload": f"one job of {K} 2^{args.po2}-cycle synthetic segments planned by bx_plan_job (executor.rs:566-698): {K} Prove tasks, "
                                      f"{K - 1} stand-in Join tasks (2^{args.join_po2}-cycle synthetic proofs seeded by their children's seals), resolve, finalize; "
                                      f"every seal CPU-verified; trace widths {'/'.join(map(str, widths))}",
                          "po2": args.po2, "join_po2": args.join_po2, "segments_proved": K, "segments_in_flight_per_gpu": lanes,
                          "queue": "one in-memory task db with prerequisites (pending -> ready on update_task_done), every lane of every device claims when idle",
                          "parallelism": f"one process, {n} device(s) x {lanes} lanes, no collective"},
               "job": {"tasks": len(ids), "end_to_end_s": round(t_end - t_first, 4), "prove_phase_s": round(t_proves - t_first, 4),
                       "prove_phase_proofs_per_s": round(K / (t_proves - t_first), 3),
                       "join_tail_s": round(t_end - t_proves, 4), "joins": len(joins),
                       "join_levels": max(1, (K - 1).bit_length()), "joins_started_before_last_prove": sum(1 for r in joins if r.started_s < t_proves),
                       "segments_per_s_end_to_end": round(K / (t_end - t_first), 3), "wall_s_including_claims": round(dt, 4),
                       # the reference's own figure of merit for a job (crates/boundless-cli/src/commands/prover/benchmark.rs:213:
                       # effective_khz = total_cycles / elapsed_secs / 1000) — of THIS job: synthetic circuit, stand-in recursion proofs
                       "effective_khz_synthetic": round(K * (1 << args.po2) / (t_end - t_first) / 1000.0, 1),
                       "note": "value = K / prove phase (first claim to last Prove done); join_tail_s = last Prove done to finalize done: the part of "
                               "the job whose parallelism halves at every level and starves N GPUs"},
               "tasks_per_device": {str(d): int(c) for d, c in per_dev.items()}}
        print(json.dumps(out))
    finally:
        a.close()


def batch_of_jobs(args, a, submit, K, n, lanes, widths, make_aux):
    """`--job K --jobs M`: M planned jobs submitted at once (a broker batch of M orders, each a K-segment execution).  The task db
    hands out the oldest ready task of the OLDEST job (9_request_work.sql:139-141: job-level FIFO), so orders complete one after
    the other instead of all at the end: the line reports when each job's finalize task was done."""
    M = args.jobs
    names = [f"order-{j}" for j in range(M)]
    import threading

    ids = {j: submit(j, K) for j in names}
    total = sum(len(v) for v in ids.values())
    aux = make_aux()
    aux_done = []
    th = threading.Thread(target=lambda: aux_done.append(aux.poll_work(max_idle_polls=None)))
    try:
        t0 = time.perf_counter()
        th.start()
        done = a.poll_work(max_idle_polls=2)
        deadline = time.perf_counter() + 30
        while any(a.taskdb.job(j)["state"] == "running" for j in names) and time.perf_counter() < deadline:
            time.sleep(0.001)  # the last finalize is the aux agent's
        dt = time.perf_counter() - t0
    finally:
        aux.stop()
        th.join()
        aux.close()
    done += sum(aux_done)
    if done != total or any(a.taskdb.job(j)["state"] != "done" for j in names):
        raise RuntimeError(f"batch: {done} of {total} tasks done")
    first = min(a.taskdb.task(j, "0").started_s for j in names)
    rows = []
    for j in names:
        t = [a.taskdb.task(j, x) for x in ids[j]]
        rows.append({"job": j, "first_claim_s": round(min(r.started_s for r in t if r.started_s > 0) - first, 4),
                     "done_s": round(a.taskdb.task(j, "finalize").updated_s - first, 4)})
    end = max(r["done_s"] for r in rows)
    out = {"metric": "segment-proofs/sec @ 2^20 cycles", "value": M * K / end, "unit": "segment-proofs/s", "n_gpus": n, "steps": 1, "warmup": 1,
           "ms_per_step": 1e3 * dt, "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "u32 (BabyBear Montgomery)",
           "data": "synthetic", "join": "synthetic stand-in",
           "config": {"workload": f"a batch of {M} jobs of {K} 2^{args.po2}-cycle synthetic segments each, all submitted before the clock starts: per job {K} Prove "
                                  f"tasks, {K - 1} stand-in Join tasks (2^{args.join_po2}), resolve, finalize; every seal CPU-verified; widths {'/'.join(map(str, widths))}",
                      "po2": args.po2, "join_po2": args.join_po2, "segments_proved": M * K, "segments_in_flight_per_gpu": lanes,
                      "queue": "one in-memory task db; request_work = the oldest ready task of the oldest job (9_request_work.sql:139-141)",
                      "parallelism": f"one process, {n} device(s) x {lanes} lanes + one aux lane (finalize), no collective"},
           "batch": {"jobs": rows, "batch_s": round(end, 4), "mean_job_latency_s": round(sum(r["done_s"] for r in rows) / M, 4),
                     "first_job_done_s": rows[0]["done_s"], "segments_per_s": round(M * K / end, 3),
                     "note": "value = all segments / time until the last job's finalize; with claims in plain creation order (every job's proves "
                             "before any job's joins) every job would finish near batch_s"}}
    print(json.dumps(out))


def job_dist_main(args, widths):
    """`--job K` under torchrun: ONE PROCESS PER GPU.  Rank r proves its K/N segments and joins them to one subtree root on its own GPU;
    the N roots — a receipt each, the only bytes that cross GPUs — are all-gathered (RCCL over xGMI under nccl) and rank 0 joins them,
    resolves and finalizes (boundless_amd/dist.py: distributed_job).  north_star: "RCCL over xGMI only for the final recursion join"."""
    import torch

    from boundless_amd import agent as ag
    from boundless_amd.dist import distributed_job, gather_over_ranks, init_distributed, max_over_ranks
    from boundless_amd.prover import Segment

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP HAL has no CPU fallback")
    rank, world, local_rank, dist = init_distributed(args.dist_backend, force=args.force_dist)
    device = local_rank if args.device is None else args.device
    torch.cuda.set_device(device)
    lanes = max(1, args.inflight)
    a = ag.Agent(prover=None, device=device, inflight=lanes, widths=widths, poll_time=0.001, verify=True, terms=args.terms, degree=args.degree,
                 join_po2=args.join_po2, also_streams="aux", max_shapes=2, lift_po2=args.join_po2 if args.lift else 0)
    try:
        a.prewarm(args.po2)
        a.prewarm(args.join_po2)
        distributed_job(a, 2 * lanes * world, lambda i: Segment.synthetic(10**6 + i, po2=args.po2), rank=rank, world=world, dist=dist, job="warm")
        res = distributed_job(a, args.job, lambda i: Segment.synthetic(i, po2=args.po2), rank=rank, world=world, dist=dist, job="timed")
        rows = gather_over_ranks([rank, device, res["segments"], res["sub_job_s"], res["prove_phase_s"], res["gather_s"], res["end_to_end_s"]], dist)
        prove_phase = max(r[4] for r in rows)
        if rank == 0:
            rollup = res["rollup"]
            rollup.verify_integrity()
            if args.dump:
                import numpy as np

                os.makedirs(args.dump, exist_ok=True)
                np.savez(os.path.join(args.dump, "rollup.npz"), seal=rollup.seal, po2=rollup.po2)
            out = {"metric": "segment-proofs/sec @ 2^20 cycles", "value": args.job / prove_phase, "unit": "segment-proofs/s", "n_gpus": world,
                   "steps": 1, "warmup": 1, "ms_per_step": 1e3 * max(r[6] for r in rows), "higher_is_better": True, "scaling": "strong",
                   "vs_baseline": None, "dtype": "u32 (BabyBear Montgomery)", "data": "synthetic", "join": "synthetic stand-in",
                   "lift": ("synthetic stand-in (one more 2^%d-cycle proof per Prove task)" % args.join_po2) if args.lift else "none (the segment's own seal is stored)",
                   "config": {"workload": f"one job of {args.job} 2^{args.po2}-cycle synthetic segments sharded over {world} one-GPU agents (one process per GPU): "
                                          f"every rank proves {args.job // world} segments and joins them to one subtree root (stand-in joins, 2^{args.join_po2} cycles), "
                                          f"the {world} roots are all-gathered, rank 0 joins them, resolves and finalizes; every seal CPU-verified; "
                                          f"trace widths {'/'.join(map(str, widths))}",
                              "po2": args.po2, "join_po2": args.join_po2, "segments_proved": args.job, "segments_in_flight_per_gpu": lanes,
                              "parallelism": f"{world} processes x {lanes} lanes; ONE collective: all_gather of {world} root receipts "
                                             f"({res['root_receipt_bytes']} bytes each) over torch.distributed/{dist.get_backend()}"},
                   "collective": {"op": "all_gather", "backend": dist.get_backend(), "world_size": dist.get_world_size(),
                                  "bytes_per_rank": res["root_receipt_bytes"], "seconds_max_over_ranks": round(max(r[5] for r in rows), 5),
                                  "note": "includes the wait for the slowest rank's subtree"},
                   "job": {"end_to_end_s": round(max(r[6] for r in rows), 4), "prove_phase_s_max_over_ranks": round(prove_phase, 4),
                           "sub_job_s_max_over_ranks": round(max(r[3] for r in rows), 4), "top_of_tree_s": round(res["top_s"], 4),
                           "top_joins": res["top_joins"], "segments_per_s_end_to_end": round(args.job / max(r[6] for r in rows), 3),
                           "effective_khz_synthetic": round(args.job * (1 << args.po2) / max(r[6] for r in rows) / 1000.0, 1),
                           "rollup_seal_words": int(rollup.seal.size)},
                   "per_rank": [{"rank": int(r[0]), "device": int(r[1]), "segments": int(r[2]), "sub_job_s": round(r[3], 4),
                                 "prove_phase_s": round(r[4], 4)} for r in rows]}
            print(json.dumps(out))
    finally:
        a.close()
        dist.destroy_process_group()


def _prove_ids(k):
    """Task numbers of the Segment tasks in the planner's numbering for k segments (cached)."""
    if k not in _prove_ids.cache:
        from boundless_amd.planner import Planner

        p, ids = Planner(), set()
        for _ in range(k):
            p.enqueue_segment()
        p.finish()
        for i in range(p.task_count()):
            t = p.get_task(i)
            if t.command == "Segment":
                ids.add(t.task_number)
        _prove_ids.cache[k] = ids
    return _prove_ids.cache[k]


_prove_ids.cache = {}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--po2", type=int, default=20)
    ap.add_argument("--widths", type=str, default="16,256,64")
    ap.add_argument("--inflight", type=int, default=3, help="segments proved concurrently per GPU (one prover + stream each); a step = one batch of this many segments per GPU")
    ap.add_argument("--batch", type=int, default=0, help="BASELINE configs[2]: prove this many segments in total, claimed from a shared queue (--steal) instead of --steps per rank")
    ap.add_argument("--steal", action="store_true", help="claim-when-idle ticket queue instead of the static rank split")
    ap.add_argument("--dist-backend", type=str, default=None, help="override the torch.distributed backend (default nccl = RCCL); 'gloo' lets the N>1 path be exercised on a single-GPU box")
    ap.add_argument("--force-dist", action="store_true", help="create the process group even for one rank (under torchrun): exercises the RCCL rendezvous, barrier and all-reduce of the N>1 path on a one-GPU box")
    ap.add_argument("--device", type=int, default=None, help="force the HIP device index for every rank (testing only; default LOCAL_RANK)")
    ap.add_argument("--cpus", type=int, default=0, help="restrict this process (all lane/finisher threads) to the first N allowed CPUs: the host budget of one GPU's share of the box (tools/host_budget.py)")
    ap.add_argument("--wait", choices=("block", "spin", "poll"), default=None, help="how host threads wait for their stream (BX_WAIT; library default: poll = hipEventQuery + usleep)")
    ap.add_argument("--native-agent", action="store_true", help="second N>1 design: ONE process, no torch.distributed; the native agent (include/bx_agent.h) runs "
                    "--inflight lanes on each of --gpus devices, all claiming from one task db; value = segments/s through the whole feed loop")
    ap.add_argument("--inject-child-failure", action="store_true", help="test hook (tests/test_fullsize_gpu.py): the native-agent child that rank 0 spawns at N > 1 is "
                    "given a device that does not exist, so it fails at once; the primary line must still come out and no rank may hang")
    ap.add_argument("--no-native-agent-extra", action="store_true", help="N>1 under torchrun: skip the untimed native-agent run that rank 0 spawns after the timed region")
    ap.add_argument("--no-live-profile", action="store_true", help="do not bracket the entry points with HIP events in the timed region (measures what those events cost: "
                    "an event record is a barrier packet between two kernels; the per-kernel figures then come from the isolated probe only)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-plain-hal", action="store_true", help="skip the untimed `single_proof_ms.plain_hal` extra (one proof through the plain Hal entry points, tests/plain_hal_prover.c)")
    ap.add_argument("--no-agent-mode", action="store_true", help="skip the untimed native-agent (feed loop) measurement")
    ap.add_argument("--cpu-sample-po2", type=int, default=20, help="size of the oracle proof timed for cpu_baseline (default: the metric's 2^20, ~40 s of CPU)")
    ap.add_argument("--terms", type=int, default=0, help="synthetic circuit: product terms per constraint (0 = default)")
    ap.add_argument("--degree", type=int, default=0, help="synthetic circuit: factors per term (0 = default)")
    ap.add_argument("--job", type=int, default=0, help="prove ONE planned job of this many segments (bx_plan_job: proves -> stand-in joins -> resolve -> finalize) through the "
                    "native agent on --gpus devices; reports prove-phase rate, join-tail latency and end-to-end seconds, labelled \"join\": \"synthetic stand-in\"")
    ap.add_argument("--jobs", type=int, default=1, help="--job: submit this many such jobs at once (a broker batch: BASELINE configs[4]'s shape) and report when each one "
                    "finished — the task db serves the oldest job first, as the reference's request_work does (9_request_work.sql:139-141)")
    ap.add_argument("--join-po2", type=int, default=18, help="--job: size of the stand-in join proofs (18 = the reference's recursion proofs)")
    ap.add_argument("--lift", action="store_true", help="--job: every Prove task also runs the stand-in `lift` leg (prove.rs:60-113): a second synthetic proof of 2^--join-po2 "
                    "cycles seeded by the segment seal; the joins consume the lifted receipts")
    ap.add_argument("--segment-bytes", type=int, default=0, help="size of every segment's serialized form: the 28-byte stand-in header + a payload that is uploaded "
                    "(pinned staging slot -> copy stream -> HBM) and handed to witgen like a preflight trace; 0 = header only.  The reference's 2^20-cycle segment is ~80 MB (executor.rs:45)")
    ap.add_argument("--two-deep", action="store_true", help="with --segment-bytes: a feeder thread per lane submits segment k+1 (bx_prover_submit_segment) while segment k is proved")
    ap.add_argument("--no-pcie-extra", action="store_true", help="skip the untimed PCIe-inclusive leg (80 MB segments) of the default N=1 run")
    ap.add_argument("--rendezvous-only", action="store_true", help="test hook: rendezvous + one all-reduce counting the ranks, print {n_gpus, ranks_counted} and exit (no GPU needed with --dist-backend gloo)")
    ap.add_argument("--dump", type=str, default=None, help="directory: every rank writes rank{r}.npz with the segment indices it claimed in the timed region and their seals (parity tests of the N>1 path)")
    args = ap.parse_args()
    widths = tuple(int(x) for x in args.widths.split(","))
    if args.wait:
        os.environ["BX_WAIT"] = args.wait
    if args.cpus:
        os.sched_setaffinity(0, set(sorted(os.sched_getaffinity(0))[: args.cpus]))
    if args.job:
        return job_main(args, widths)
    if args.native_agent:
        return native_agent_main(args, widths)
    # `--gpus N` means N: one process per GPU.  Started without a launcher the line would silently measure ONE GPU under an N = 8 flag
    # (VERDICT r05 weak #7), so plain `python bench.py --gpus N` re-executes itself under torch.distributed.run, and a launcher whose
    # world size disagrees with the flag is an error.
    env_world = os.environ.get("WORLD_SIZE")
    if env_world is None and args.gpus > 1:
        return relaunch_one_process_per_gpu(args.gpus)
    if env_world is not None and int(env_world) != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={env_world} processes; they must agree "
                         f"(python -m torch.distributed.run --nproc-per-node {args.gpus} bench.py --gpus {args.gpus} ...)")
    if args.rendezvous_only:
        return rendezvous_only(args)

    import resource
    import threading

    import torch

    from boundless_amd.dist import SegmentQueue, gather_over_ranks, init_distributed, max_over_ranks, sum_over_ranks

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP HAL has no CPU fallback")
    rank, world, local_rank, dist = init_distributed(args.dist_backend, force=args.force_dist)
    if args.device is not None:
        local_rank = args.device
    torch.cuda.set_device(local_rank)

    from boundless_amd.prover import HipProverServer, Segment

    servers = [HipProverServer(device=local_rank, po2=args.po2, widths=widths, terms=args.terms, degree=args.degree)
               for _ in range(max(1, args.inflight))]
    hal = servers[0].hal

    def barrier():
        torch.cuda.synchronize()
        for sv in servers:
            sv.hal.sync()
        if dist is not None:
            dist.barrier()

    seg_bufs = {}

    def segment_buffer(sv, nbytes):
        """One host buffer per lane holding the serialized segment (built once, outside any timed region: it stands for bytes the
        hot store already handed over)."""
        import numpy as np

        key = (id(sv), nbytes)
        if key not in seg_bufs:
            b = bytearray(nbytes)
            if nbytes > 28:
                b[28:] = np.random.default_rng(nbytes).integers(0, 256, nbytes - 28, dtype=np.uint8).tobytes()
            seg_bufs[key] = b
        return seg_bufs[key]

    def run(total_per_rank, total_global, tag, seg_bytes=None):
        """Prove segments claimed from the queue with `inflight` provers per GPU; returns (proved by this rank, last receipt)."""
        seg_bytes = args.segment_bytes if seg_bytes is None else seg_bytes
        if seg_bytes and seg_bytes < 28:
            raise SystemExit("--segment-bytes must be 0 or at least the 28-byte header")
        q = SegmentQueue(total_global, rank=rank, world=world, dist=dist, mode="steal" if args.steal else "static", name=tag)
        lock = threading.Lock()
        done = [0]
        last = [None]
        claimed = {}
        by_server = {id(sv): 0 for sv in servers}

        def claim(sv):
            with lock:
                if not args.steal and done[0] >= total_per_rank:
                    return None
                idx = q.claim()
                if idx is None:
                    return None
                done[0] += 1
                by_server[id(sv)] += 1
                return idx

        def worker(sv):
            # the segment as the hot store hands it over: header + payload in one host buffer per lane (only the header changes from
            # one synthetic segment to the next; the payload is uploaded every time)
            buf = segment_buffer(sv, seg_bytes) if seg_bytes else None
            if buf is not None and args.two_deep:
                return worker_two_deep(sv, buf)
            while True:
                idx = claim(sv)
                if idx is None:
                    return
                seg = Segment.synthetic(index=idx, po2=args.po2)
                if buf is None:
                    last[0] = sv.prove_segment(seg)
                else:
                    buf[:28] = seg.to_bytes()[:28]
                    last[0] = sv.prove_segment_buffer(buf, index=idx)
                if args.dump and tag == "timed":
                    claimed[idx] = last[0].seal

        def worker_two_deep(sv, buf):
            # segment k+1 is staged and uploaded while segment k is proved — the feeder thread makes the pinned copy and
            # enqueues the upload (bx_prover_submit_segment), the lane thread proves what was submitted (bx_prove_submitted)
            import queue as _q

            handed = _q.Queue()
            free = threading.Semaphore(2)

            def feeder():
                while True:
                    free.acquire()
                    idx = claim(sv)
                    if idx is None:
                        handed.put(None)
                        return
                    buf[:28] = Segment.synthetic(index=idx, po2=args.po2).to_bytes()[:28]
                    sv.submit_segment_buffer(buf)  # returns once the pinned copy is made: buf may be rewritten
                    handed.put(idx)

            ft = threading.Thread(target=feeder)
            ft.start()
            while True:
                idx = handed.get()
                if idx is None:
                    break
                last[0] = sv.prove_submitted(index=idx)
                free.release()
                if args.dump and tag == "timed":
                    claimed[idx] = last[0].seal
            ft.join()

        if len(servers) == 1:
            worker(servers[0])
        else:
            ts = [threading.Thread(target=worker, args=(sv,)) for sv in servers]
            [t.start() for t in ts]
            [t.join() for t in ts]
        if args.dump and tag == "timed":
            import numpy as np

            os.makedirs(args.dump, exist_ok=True)
            order = sorted(claimed)
            np.savez(os.path.join(args.dump, f"rank{rank}.npz"), indices=np.array(order, dtype=np.int64),
                     **{f"seal_{i}": claimed[i] for i in order})
        run.by_server = by_server
        return done[0], last[0]

    per_rank = args.steps * len(servers)
    total_global = args.batch if args.batch else per_rank * world
    if args.segment_bytes:
        for sv in servers:
            segment_buffer(sv, args.segment_bytes)
    run(args.warmup * len(servers), args.warmup * len(servers) * world, "warm")
    barrier()
    # Live per-entry-point HIP events (the durations behind `roofline_in_region` and `kernels`) on ONE lane per rank: an event
    # record is a barrier packet between two kernels, and bracketing every entry point of every lane costs 0.9 % of the rate at any
    # lane count (profiles/r03_ab_live_profile.jsonl).  The profiled lane's segments are what the per-segment figures divide by.
    live = [] if args.no_live_profile else servers[:1]
    for sv in servers:
        sv.hal.profile_reset()
        sv.hal.profile_enable(any(sv is x for x in live))
    ru0 = resource.getrusage(resource.RUSAGE_SELF)
    t0 = time.perf_counter()
    proved, receipt = run(per_rank if not args.batch else total_global, total_global, "timed")
    elapsed_own = time.perf_counter() - t0  # this rank's own work, before it waits for the others
    barrier()
    elapsed = time.perf_counter() - t0
    ru1 = resource.getrusage(resource.RUSAGE_SELF)
    host_cpu_s = (ru1.ru_utime - ru0.ru_utime) + (ru1.ru_stime - ru0.ru_stime)  # every thread of this rank, user + system
    prof = {}
    proved_live = sum(run.by_server[id(sv)] for sv in live)
    for sv in servers:
        sv.hal.profile_enable(False)
    for sv in live:
        for name, r in sv.hal.profile_report().items():
            a0 = prof.setdefault(name, {"calls": 0, "ms": 0.0, "alg_bytes": 0.0})
            for k in a0:
                a0[k] += r[k]
    elapsed = max_over_ranks(elapsed, dist)
    proved_total = int(round(sum_over_ranks(proved, dist)))
    # who did what: one row per rank, so that a SCALE run says by itself whether the backend saw N ranks and whether they were balanced
    per_rank_rows = [{"rank": int(r[0]), "device": int(r[1]), "proofs": int(r[2]), "seconds": round(r[3], 4)}
                     for r in gather_over_ranks([rank, local_rank, proved, elapsed_own], dist)]

    # One proof alone (untimed, rank 0): the reference's agent proves one segment at a time per process.  No HIP events around the
    # entry points here (an event record is a barrier packet), wall clock around bx_prove_segment_bytes.
    single_ms = None
    if rank == 0:
        sv = servers[0]
        ts = []
        for k in range(4):
            t1 = time.perf_counter()
            sv.prove_segment(Segment.synthetic(index=2 * 10**6 + k, po2=args.po2))
            ts.append(1e3 * (time.perf_counter() - t1))
        single_ms = {"min": round(min(ts[1:]), 3), "median": round(sorted(ts[1:])[1], 3), "runs": [round(x, 3) for x in ts[1:]]}
        # the same with the host thread busy-waiting on its stream (BX_WAIT=spin / wait_blocking = 0): the latency mode of a process
        # that proves one segment at a time — its ~10 round trips per proof no longer pay the sleep-poll's wake-up, at the price of
        # one busy core.  Skipped when the run was started with a policy of its own.
        if not os.environ.get("BX_WAIT") and "wait_blocking" not in os.environ.get("BX_TUNABLES", ""):
            try:
                sv.hal.set_tunable("wait_blocking", 0)
                ts = []
                for k in range(4):
                    t1 = time.perf_counter()
                    sv.prove_segment(Segment.synthetic(index=3 * 10**6 + k, po2=args.po2))
                    ts.append(1e3 * (time.perf_counter() - t1))
                single_ms["spin_wait"] = {"min": round(min(ts[1:]), 3), "median": round(sorted(ts[1:])[1], 3), "runs": [round(x, 3) for x in ts[1:]]}
            finally:
                sv.hal.set_tunable("wait_blocking", 2)
    barrier()

    # Isolated probe (untimed, rank 0 only): with several segments in flight the HIP-event durations of the timed region
    # include time-slicing with the other stream, so one extra segment is proved alone to get each entry point's own
    # duration (this is what a rocprofv3 --kernel-trace of `--inflight 1` reports).
    iso = {}
    if rank == 0 and len(servers) > 1:
        sv = servers[0]
        sv.hal.profile_reset()
        sv.hal.profile_enable(True)
        for k in range(ISO_SEGMENTS):  # three proofs, one after the other: 21 LDE calls behind the roofline figure instead of 7
            sv.prove_segment(Segment.synthetic(index=10**6 + k, po2=args.po2))
        sv.hal.profile_enable(False)
        iso = sv.hal.profile_report()
    # The same segment proved from OUTSIDE the library through the plain Hal-trait entry points only (tests/plain_hal_prover.c: what a
    # Rust `impl Hal for HipHal` shim driven by risc0-zkp's own prover gets), then with each extension entry point of bx_hal.h swapped
    # in alone.  After the isolated probe: its stop-and-go load lets the clocks sag, and the LDE probe right
    # behind it measured 0.78 instead of 0.71 ms per call.
    if rank == 0 and single_ms is not None and not args.no_plain_hal:
        single_ms["plain_hal"] = plain_hal_probe(args, servers[0], single_ms["min"])
    barrier()

    if rank == 0:
        kernels = {}
        for name, r in prof.items():
            ms = r["ms"] / max(r["calls"], 1)
            gbps = r["alg_bytes"] / max(r["calls"], 1) / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
            kernels[name] = {"calls_per_step": r["calls"] / max(proved_live, 1), "avg_ms": round(ms, 4),
                             "ms_per_segment": round(r["ms"] / max(proved_live, 1), 3), "alg_GBps": round(gbps, 1),
                             "frac_hbm": round(gbps / HBM_PEAK_GBPS, 4)}
        ntt = kernels.get("batch_expand_into_evaluate_ntt", {})
        iso_k = {}
        for name, r in iso.items():
            ms = r["ms"] / max(r["calls"], 1)
            gbps = r["alg_bytes"] / max(r["calls"], 1) / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
            iso_k[name] = {"avg_ms": round(ms, 4), "ms_per_segment": round(r["ms"] / ISO_SEGMENTS, 3), "alg_GBps": round(gbps, 1),
                           "frac_hbm": round(gbps / HBM_PEAK_GBPS, 4), "calls": r["calls"] // ISO_SEGMENTS,
                           "alg_MB_per_call": round(r["alg_bytes"] / max(r["calls"], 1) / 1e6, 2)}
            # an entry point whose average call moves a few MB in a few tens of microseconds is bounded by launch and ramp latency,
            # not by HBM: its frac_hbm is printed for completeness and means nothing
            if ms < 0.04:
                iso_k[name]["latency_bound"] = True
        # HBM traffic of the LDE's two kernels from the committed PMC passes of this same command (separate
        # rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE runs, tools/pmc_traffic.py); None when the file is absent
        traffic = None
        traffic_file = "(none)"
        try:
            traffic_file = os.path.basename(_profile("bench_pmc_traffic.json"))
            pmc = _load_profile("bench_pmc_traffic.json")["kernels"]
            sel = [v for k, v in pmc.items() if "ntt_r16_kernel<false" in k or "ntt_passA_fwd12_multi_kernel" in k]
            if sel:
                launches = max(v["launches"] for v in sel)
                traffic = sum(v["hbm_bytes_per_launch"] * v["launches"] for v in sel) / launches
        except Exception:
            traffic = None
        def ntt_roofline(k, measured):
            e = k.get("batch_expand_into_evaluate_ntt", {})
            return {"kernel": "batch_expand_into_evaluate_ntt (ntt_r16_kernel pass A + pass B, 4x LDE)", "bound": "hbm",
                    "achieved": e.get("alg_GBps"), "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": e.get("frac_hbm"),
                    "traffic": traffic, "avg_ms_per_launch": e.get("avg_ms"),
                    "achieved_bytes_per_launch": (e.get("alg_GBps", 0) or 0) * 1e9 * (e.get("avg_ms", 0) or 0) * 1e-3,
                    "measured": measured,
                    "traffic_over_algorithmic": (round(traffic / ((e.get("alg_GBps", 0) or 0) * 1e9 * (e.get("avg_ms", 0) or 0) * 1e-3), 3)
                                                 if traffic and e.get("alg_GBps") and e.get("avg_ms") else None),
                    "note": "algorithmic bytes = 4B*(in + out) words per call / HIP-event time on the HAL stream; traffic = "
                            "FETCH_SIZE(x2)+WRITE_SIZE bytes per LDE call (both passes) from profiles/" + traffic_file + " (2.7x = the "
                            "two-pass floor: pass A's output is written and re-read); the path is VALU-issue-bound (DESIGN.md section 4): "
                            "see valu_view, and `dominant` for the job's dominant kernel"}

        # With several segments in flight the per-launch durations of the timed region include time-slicing between the
        # streams, so the kernel's own roofline comes from the isolated probe (same process, right after the timed region,
        # HIP events on the HAL stream; agrees with the rocprofv3 summary of `--inflight 1`); the in-region figure (agrees
        # with the rocprofv3 summary of the default command) is reported beside it.
        def ntt_valu_view(r):
            # VALU-issue view of the same launch.  Instruction COUNTS per wave come from the committed PMC pass (SQ_INSTS_VALU / SQ_WAVES;
            # pass A multi-column: 8192 output elements per wave, pass B: 1024), their split into issue classes from the static mix of
            # the two kernels' gfx950 text (tools/isa_mix.py; pass B is straight-line code, so its static count must equal the PMC's).
            # Two yardsticks: (1) `frac_weighted_issue` = the class-weighted bound VERDICT r04 asked for (2 cycles per cheap, 4 per
            # multiply-class instruction at 2.4 GHz) — NOT reachable: profiles/r05_microbench5_class_mix.jsonl shows that in a MIXED
            # stream every VALU instruction issues at ~1.75 ns whatever its class; (2) `frac_of_mixed_stream_rate` = against the rate a
            # bare loop of the same butterflies reaches on this chip (profiles/r05_microbench4_operand_kinds.jsonl), which is the bound.
            try:
                pmc = _valu_per_wave()
                ia = [v for k, v in pmc.items() if "ntt_passA_fwd12_multi_kernel" in k][0]
                ib = [v for k, v in pmc.items() if "ntt_r16_kernel<false, false, 0, 10, 4" in k][0]
                a, b = ia / 8192.0, ib / 1024.0
                out_elems = r["achieved_bytes_per_launch"] / 4.0 / 1.25
                secs = r["avg_ms_per_launch"] * 1e-3
                rate = out_elems * (a + b) / secs
                view = {"wave_insts_per_output_element": round(a + b, 4), "wave_insts_per_s": rate,
                        "valu_insts_per_wave": {"pass_a_8_columns": ia, "pass_b": ib},
                        "ns_per_wave_inst_per_simd": round(1024.0 / rate * 1e9, 4),
                        "issue_peak_mul": 1024 * 2.4e9 / 4, "issue_peak_cheap": 1024 * 2.4e9 / 2,
                        "frac_of_mul_class_peak": round(rate / (1024 * 2.4e9 / 4), 3)}
                try:
                    mix = _load_profile("lde_isa_mix.json")["kernels"]
                    ma = [v for k, v in mix.items() if "ntt_passA_fwd12_multi_kernel<2>" in k or "ntt_passA_fwd12_multi_kernel<2, false>" in k][0]
                    mb = [v for k, v in mix.items() if "ntt_r16_kernel<false, false, 0, 10, 4" in k][0]
                    fa = ma["mul_class_insts"] / ma["valu_insts"]
                    fb = mb["mul_class_insts"] / mb["valu_insts"]
                    cyc = a * (4 * fa + 2 * (1 - fa)) + b * (4 * fb + 2 * (1 - fb))  # SIMD cycles per output element, per wave-lane group
                    view.update({"static_mix": {"pass_a": {"mul_class_insts": ma["mul_class_insts"], "cheap_insts": ma["cheap_insts"],
                                                           "note": "whole kernel text (prologue + one trip of each column-loop variant)"},
                                                "pass_b": {"mul_class_insts": mb["mul_class_insts"], "cheap_insts": mb["cheap_insts"],
                                                           "static_valu_insts": mb["valu_insts"], "pmc_valu_insts_per_wave": ib}},
                                 "weighted_issue_cycles_per_output_element": round(cyc, 4),
                                 "frac_weighted_issue": round(out_elems * cyc / secs / (1024 * 2.4e9), 3)})
                except Exception:
                    pass
                try:
                    ref_ns = None
                    for ln in open(_profile("microbench4_operand_kinds.jsonl")):
                        d = json.loads(ln)
                        if "waves_per_simd" in d and ref_ns is not None:
                            break  # the first block is 8 waves per SIMD
                        if d.get("seq", "").startswith("non-lazy DIT butterfly") and "literals" in d["seq"]:
                            ref_ns = d["ns_per_wave_butterfly_per_simd"] / 11.0
                    if ref_ns:
                        view.update({"mixed_stream_ns_per_wave_inst": round(ref_ns, 4),
                                     "frac_of_mixed_stream_rate": round(ref_ns / (1024.0 / rate * 1e9), 3)})
                except Exception:
                    pass
                if r.get("traffic"):
                    view["traffic_GBps"] = round(r["traffic"] / secs / 1e9, 1)
                    view["traffic_frac_of_hbm_peak"] = round(r["traffic"] / secs / 1e9 / HBM_PEAK_GBPS, 3)
                r["valu_view"] = view
            except Exception:
                pass
            return r

        roofline_in_region = ntt_roofline(kernels, "timed region, %d segments in flight" % len(servers))
        if len(servers) > 1:
            # HIP-event wall times of ONE lane while the other lanes share the GPU: a kernel's events also span the other lanes' kernels
            # that the hardware ran in between, so these are not kernel durations and GB/s figures derived from them mean nothing
            # (VERDICT r05 weak #10).  The kernel's own figures are `roofline` / `kernels_isolated`.
            for k_ in ("achieved", "frac", "avg_ms_per_launch", "achieved_bytes_per_launch", "traffic_over_algorithmic"):
                roofline_in_region[k_ + "_under_time_slicing"] = roofline_in_region.pop(k_)
            kernels = {name: {"calls_per_step": v["calls_per_step"], "wall_ms_under_time_slicing": v["avg_ms"],
                              "wall_ms_per_segment_under_time_slicing": v["ms_per_segment"]} for name, v in kernels.items()}
        roofline = ntt_valu_view(ntt_roofline(iso_k, "isolated probe: %d extra segments proved one at a time after the timed region" % ISO_SEGMENTS)) if iso_k else roofline_in_region
        # dominance is judged on the isolated durations (in-region ones are stretched by stream sharing)
        dom_src = iso_k if iso_k else kernels
        dom_name = max(dom_src, key=lambda k: dom_src[k]["ms_per_segment"]) if dom_src else None
        # The dominant entry point (hash_rows: Poseidon2 leaf hashing) is VALU-issue-bound: report its instruction rate
        # from the committed PMC count of VALU instructions per permutation against the chip's issue peaks
        # (1024 SIMDs x 2.4 GHz / 2 cycles for the cheap class, / 4 cycles for multiplies, profiles/r01_microbench2_instr_cost.jsonl).
        dominant = {"kernel": dom_name, **(dom_src.get(dom_name, {}) if dom_name else {}),
                    "note": "Poseidon2 is VALU-issue-bound (no HBM or MFMA roofline applies); see DESIGN.md section 4"}
        try:
            src_k = iso_k if iso_k else kernels
            pmc = _valu_per_wave()
            per_perm = [v for k, v in pmc.items() if "hash_fold_kernel" in k][0]
            rows4 = 4 << args.po2
            perms = rows4 * sum((w + 15) // 16 for w in list(widths) + [16])
            hr = src_k.get("hash_rows", {})
            if dom_name == "hash_rows" and hr.get("ms_per_segment"):
                trees_ms = hr["ms_per_segment"]
                # hash_rows calls per segment also include the FRI rounds (64 columns, rows/16): add them
                s_ = 1 << args.po2
                while s_ > 256:
                    perms += (4 * s_ // 16) * 4
                    s_ //= 16
                perm_rate = perms / (trees_ms * 1e-3)
                wave_insts = perm_rate / 64.0 * per_perm
                dominant.update({"valu_insts_per_permutation": round(per_perm), "permutations_per_s": perm_rate,
                                 "wave_insts_per_s": wave_insts, "issue_peak_cheap": 1024 * 2.4e9 / 2, "issue_peak_mul": 1024 * 2.4e9 / 4,
                                 "frac_of_mul_class_peak": round(wave_insts / (1024 * 2.4e9 / 4), 3),
                                 # the same rate counting only the permutation's ALGORITHMIC multiplies (1356 modmul x 3
                                 # multiply-class instructions each): what a reader should hold against "91 % of peak"
                                 "frac_algorithmic": round(perm_rate * 1356 * 3 / 64.0 / (1024 * 2.4e9 / 4), 3),
                                 "measured": "isolated probe" if iso_k else "timed region"})
        except Exception:
            pass
        # The circuit behind prove_segment is the synthetic AIR of include/bx_prover.h, NOT rv32im: the number is the rate of
        # complete STARK proofs of that circuit (witgen + accumulate + eval_check + commits + DEEP + FRI + queries), and is
        # not comparable to upstream's effective kHz.  `share_of_gpu_time` is the circuit stages' part of one proof's GPU time.
        src_c = iso_k if iso_k else kernels
        circ_ops = ("witgen_fill", "scatter", "witgen_derive", "accum_gather", "accum_build", "prefix_products", "accum_store", "eval_check")
        circ_ms = sum(src_c[k]["ms_per_segment"] for k in circ_ops if k in src_c)
        all_ms = sum(v["ms_per_segment"] for v in src_c.values())
        circuit_view = {"kind": "synthetic AIR (include/bx_prover.h), not rv32im: public words bound to the trace, seeded ZK noise rows (last 1994), but no image id / claim; lift is not included",
                        "included": ["witness generation (synthetic) with ZK noise rows", "accumulate (prefix_products)", "eval_check (synthetic constraints / vanishing polynomial)",
                                     "3 trace commits + check commit", "DEEP", "FRI", "50 queries"],
                        "excluded": ["rv32im preflight/witgen/eval_check (generated code, not in the reference tree)", "lift (recursion circuit)",
                                     "CPU verification of the seal (reported in agent_mode)"],
                        "terms": int(receipt.seal[4]), "degree": int(receipt.seal[5]),
                        "stages_ms_per_segment": {k: src_c[k]["ms_per_segment"] for k in circ_ops if k in src_c},
                        "share_of_gpu_time": round(circ_ms / all_ms, 3) if all_ms else None}
        out = {
            "metric": "segment-proofs/sec @ 2^20 cycles",
            "value": proved_total / elapsed,
            "unit": "segment-proofs/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps if not args.batch else 1e3 * elapsed / max(proved_total, 1),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u32 (BabyBear Montgomery)",
            "data": "synthetic",
            "config": {"workload": f"single 2^{args.po2}-cycle synthetic segment per GPU via the HIP HAL (witgen+accum+eval_check+NTT+Poseidon2+FRI), "
                                   f"trace widths code/data/accum = {'/'.join(map(str, widths))}, check 16, 50 queries",
                       "po2": args.po2, "segments_proved": proved_total, "segments_in_flight_per_gpu": len(servers),
                       "queue": ("claim-when-idle ticket queue (c10d store)" if args.steal else
                                 ("static rank split" if world > 1 else "one rank: its lanes take consecutive segment indices")),
                       "parallelism": f"segments sharded over {world} GPU(s), no collective",
                       "rendezvous": (None if dist is None else f"torch.distributed/{dist.get_backend()}"),
                       "circuit": circuit_view},
            "seal_words": int(receipt.seal.size),
            "host_cpu_s_per_proof": round(host_cpu_s / max(proved, 1), 5),
            "host": {"cpu_s_per_proof": round(host_cpu_s / max(proved, 1), 5), "cpu_s_in_timed_region": round(host_cpu_s, 3),
                     "cpus_busy_avg": round(host_cpu_s / elapsed, 3), "wait_policy": os.environ.get("BX_WAIT", "poll (library default)"),
                     "cpus_allowed": len(os.sched_getaffinity(0)),
                     "note": "getrusage(RUSAGE_SELF) user+system over the timed region of rank 0 (all lane threads) / proofs; "
                             "the GPU boxes give a container 16 CPUs for 8 GPUs, i.e. 2 per GPU (profiles/r03_host_budget.json)"},
            "roofline": {**roofline, "dominant": {"kernel": dominant.get("kernel"), "bound": "valu",
                                                  "frac_issue": dominant.get("frac_of_mul_class_peak"), "frac_algorithmic": dominant.get("frac_algorithmic"),
                                                  "valu_insts_per_permutation": dominant.get("valu_insts_per_permutation"),
                                                  "ms_per_segment": dominant.get("ms_per_segment"),
                                                  "note": "the job's dominant kernel (Poseidon2 leaf hashing) is VALU-issue-bound: frac_issue = wave-instructions/s "
                                                          "against 1024 SIMDs x 2.4 GHz / 4 (multiply class), frac_algorithmic counts only the 1356 x 3 "
                                                          "multiply instructions a permutation needs (SURVEY 8d)"}},
            "single_proof_ms": single_ms,
            "per_rank": per_rank_rows,
            "backend": (None if dist is None else {"name": dist.get_backend(), "world_size": dist.get_world_size()}),
            "roofline_in_region": roofline_in_region,
            "roofline_dominant": dominant,
            "roofline_job": job_valu_view(proved_total / elapsed / max(world, 1)),
            "kernels_isolated": iso_k,
            "kernels": kernels,
            "live_profile": {"lanes_with_hip_events": len(live), "lanes": len(servers), "segments_profiled": proved_live,
                             "note": "per-entry-point HIP events bracket ONE lane per rank in the timed region (every lane: -0.9 % on the rate, "
                                     "profiles/r03_ab_live_profile.jsonl); `kernels` and `roofline_in_region` are that lane's WALL times while the "
                                     "other lanes share the GPU (not kernel durations); `kernels_isolated` / `roofline` are one segment alone"},
            "replayed_profiles": replayed_profiles(),
        }
        if world == 1 and not args.no_pcie_extra and not args.segment_bytes and args.po2 >= 18:
            # Untimed extra (never `value`): the same workload with every segment carrying the ~80 MB the reference's 2^20-cycle
            # segment serializes to (executor.rs:45): pinned copy + upload on the copy stream per proof, two deep.
            try:
                nb = 80_000_000
                args.two_deep = True
                for sv in servers:
                    segment_buffer(sv, nb)
                run(len(servers), len(servers), "pcie-warm", seg_bytes=nb)
                barrier()
                t1 = time.perf_counter()
                n_p, _ = run(per_rank, per_rank, "pcie", seg_bytes=nb)
                barrier()
                dt = time.perf_counter() - t1
                ups = [sv.last_upload() for sv in servers]
                out["pcie_inclusive"] = {"segment_bytes": nb, "segment_proofs_per_s": n_p / dt, "ratio_to_value": round(n_p / dt / (proved_total / elapsed), 4),
                                         "upload_ms": round(sum(u[0] for u in ups) / len(ups), 3),
                                         "upload_GBps": round(sum(u[1] / (u[0] * 1e-3) for u in ups if u[0] > 0) / len(ups) / 1e9, 2),
                                         "note": "every proof copies its segment into a pinned staging slot and uploads it on the prover's copy stream "
                                                 "(bx_prover_submit_segment from a feeder thread while the previous segment is proved); `value` above is "
                                                 "measured with inputs resident, this is the PCIe-inclusive rate"}
            except Exception as e:  # reported, never required for the GPU number
                out["pcie_inclusive"] = {"error": f"{type(e).__name__}: {e}"}
        if world == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(min(args.cpu_sample_po2, args.po2), widths, args.po2)
            except Exception as e:  # the baseline is reported, never required for the GPU number
                out["cpu_baseline"] = {"error": str(e)}
    for sv in servers:
        sv.close()
    if world > 1 and not args.no_native_agent_extra:
        # Untimed extra at N > 1: the other multi-GPU design measured by the same command.  Every rank has released its
        # provers; rank 0 runs `bench.py --native-agent` over all N devices in a CHILD process under a timeout (a failure or a
        # hang there cannot take the primary line with it), the other ranks wait at the barrier.
        # The hand-over goes through the c10d store, not through a collective: an RCCL barrier would leave kernels spinning on
        # the other ranks' GPUs while the child measures on them.
        try:  # nothing in here may cost the primary line
            import datetime

            from torch.distributed import distributed_c10d

            store = distributed_c10d._get_default_store()
            torch.cuda.synchronize()
            store.add("bx:released", 1)
            if rank == 0:
                import subprocess

                t_wait = time.time()
                while int(store.add("bx:released", 0)) < world and time.time() - t_wait < 120:
                    time.sleep(0.05)
                cmd = [sys.executable, os.path.abspath(__file__), "--native-agent", "--gpus", str(world), "--steps", str(max(2, args.steps // 2)),
                       "--warmup", "1", "--po2", str(args.po2), "--widths", args.widths, "--inflight", str(args.inflight)]
                if args.inject_child_failure:
                    cmd += ["--device", "99"]
                elif args.device is not None:
                    cmd += ["--device", str(args.device)]
                env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT",
                                                                         "TORCHELASTIC_RUN_ID", "GROUP_RANK", "ROLE_RANK", "LOCAL_WORLD_SIZE")}
                try:
                    r = subprocess.run(cmd, capture_output=True, text=True, timeout=240, env=env)
                    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
                    out["native_agent"] = json.loads(line[-1]) if line else {"error": (r.stderr or r.stdout)[-400:]}
                except Exception as e:  # reported, never required for the primary number
                    out["native_agent"] = {"error": f"{type(e).__name__}: {e}"}
                store.set("bx:native_done", "1")
            else:
                try:
                    store.wait(["bx:native_done"], datetime.timedelta(seconds=420))
                except Exception:  # rank 0 reports; a rank that gives up waiting just leaves
                    pass
        except Exception as e:
            if rank == 0:
                out["native_agent"] = {"error": f"{type(e).__name__}: {e}"}
    if rank == 0:
        if world == 1 and not args.no_agent_mode:
            # Untimed extra (never `value`): the same workload claimed through the native feed loop (bx_agent_poll_work:
            # hot-store GET -> prove -> CPU verify -> SETEX -> UNLINK -> update_task_done) with the same number of lanes.
            try:
                out["agent_mode"] = agent_mode(args, widths, local_rank, max(1, args.inflight))
            except Exception as e:  # reported, never required for the GPU number
                out["agent_mode"] = {"error": str(e)}
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
