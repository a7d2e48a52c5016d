"""GPU parity at the sizes bench.py actually runs (BASELINE.json configs[1]: 2^20 cycles, trace widths 16/256/64).

The per-entry-point tests of test_hal_gpu.py stop at sizes the oracle finishes in a second; the kernels take different
code paths (pass split, tile shapes, grid-stride loops, 32-bit index ranges) at 2^20 -> 2^22 rows x 256 columns, so the
comparisons are repeated here at exactly those shapes: the whole seal word for word, hash_rows / Merkle build,
mix_poly_coeffs and the LDE.  A few tens of seconds of CPU oracle time each.
"""
import numpy as np
import pytest

from oracle import oracle_lib as ol

pytestmark = pytest.mark.gpu
P = ol.P


@pytest.fixture(scope="module")
def hal():
    from boundless_amd.hal import HipHal

    h = HipHal(0)
    yield h
    h.close()


@pytest.fixture(scope="module")
def oracle_mt():
    """The C oracle with enough OpenMP threads for the 2^20-sized cases."""
    import os

    L = ol.lib()
    old = L.bxo_get_threads()
    L.bxo_set_threads(min(os.cpu_count() or 1, 32))
    yield L
    L.bxo_set_threads(old)


def test_seal_bit_exact_vs_oracle_at_the_baseline_config(oracle_mt):
    """BASELINE.json configs[1], literally: one 2^20-cycle segment, widths 16/256/64, seal == the CPU oracle's seal,
    word for word (and the four group roots), then the CPU verifier accepts it (prove.rs:41-55)."""
    from boundless_amd.prover import HipProverServer, Segment

    po2, widths = 20, (16, 256, 64)
    seg = Segment.synthetic(0, po2)
    srv = HipProverServer(0, po2=po2, widths=widths)
    try:
        receipt = srv.prove_segment(seg)
    finally:
        srv.close()
    seal, roots = ol.prove_segment(po2, *widths, seg.seed, oracle_mt)
    assert np.array_equal(receipt.roots, roots), "Merkle roots differ"
    assert receipt.seal.size == seal.size
    bad = np.nonzero(receipt.seal != seal)[0]
    assert bad.size == 0, f"{bad.size} differing seal words, first at {bad[:5]}"
    receipt.verify_integrity()


def test_hash_rows_at_2_22_x_256_vs_oracle(hal, oracle_mt):
    """hash_rows at the data group's LDE shape (2^22 rows x 256 columns, 4 GiB): every 1021st row (4109 rows spread over
    the whole matrix, plus the first and last 64) against the oracle's sponge."""
    rows, cols = 1 << 22, 256
    rng = np.random.default_rng(2022)
    x = ol.random_elems(rng, rows * cols)
    m = hal.copy_from(x)
    out = hal.alloc_digest(rows)
    hal.hash_rows(out, m)
    got = out.view().reshape(rows, 8)
    idx = np.unique(np.concatenate([np.arange(0, rows, 1021), np.arange(64), np.arange(rows - 64, rows)]))
    assert idx.size >= 4096
    sub = np.ascontiguousarray(x.reshape(cols, rows)[:, idx]).reshape(-1)  # column-major idx.size x cols
    ref = np.zeros(8 * idx.size, np.uint32)
    oracle_mt.bxo_hash_rows(ref, sub, idx.size, cols)
    assert np.array_equal(got[idx], ref.reshape(-1, 8))
    # the same rows once more through the slice form (elements of one row are `rows` words apart in the matrix)
    dg = np.zeros(8, np.uint32)
    for r in (0, 1021 * 777, rows - 1):
        oracle_mt.bxo_hash_elem_slice(dg, x[r:], cols, rows)
        assert np.array_equal(got[r], dg)
    m.free()
    out.free()


def test_merkle_build_at_2_22_x_16_vs_oracle(hal, oracle_mt):
    """The code group's tree at full size (2^22 leaves of 16 columns): every node, root included."""
    rows, cols = 1 << 22, 16
    x = ol.random_elems(np.random.default_rng(16), rows * cols)
    nodes = hal.alloc_digest(2 * rows)
    m = hal.copy_from(x)
    hal.merkle_build(nodes, m, rows)
    ref = np.zeros(16 * rows, np.uint32)
    leaves = np.zeros(8 * rows, np.uint32)
    oracle_mt.bxo_hash_rows(leaves, x, rows, cols)
    ref[8 * rows:] = leaves
    size = rows
    while size > 1:
        oracle_mt.bxo_hash_fold(ref, size, size // 2)
        size //= 2
    assert np.array_equal(nodes.view()[8:], ref[8:])
    m.free()
    nodes.free()


def test_mix_poly_coeffs_at_2_20_x_256_vs_oracle(hal, oracle_mt):
    """mix_poly_coeffs at the data group's shape: 256 polynomials of 2^20 coefficients into 3 combos."""
    count, npoly, ncombo = 1 << 20, 256, 3
    rng = np.random.default_rng(20)
    inp = ol.random_elems(rng, npoly * count)
    combos = (np.arange(npoly) % 4 == 0).astype(np.uint32)  # the prover's assignment: every 4th column in combo 1
    combos[-3:] = 2
    mix, start = ol.random_elems(rng, 4), ol.random_elems(rng, 4)
    init = ol.random_elems(rng, ncombo * count * 4)
    out = hal.copy_from(init)
    d_in, d_c = hal.copy_from(inp), hal.copy_from(combos)
    hal.mix_poly_coeffs(out, start, mix, d_in, d_c, npoly, count)
    ref = init.copy()
    oracle_mt.bxo_mix_poly_coeffs(ref, np.ascontiguousarray(start), np.ascontiguousarray(mix), inp, combos, npoly, count)
    assert np.array_equal(out.view(), ref)
    for b in (out, d_in, d_c):
        b.free()


def test_lde_at_2_20_x_64_vs_oracle(hal, oracle_mt):
    """interpolate -> zk_shift -> 4x LDE at 2^20 -> 2^22 on 64 columns (the accum group's shape), every word."""
    n, cols = 1 << 20, 64
    x = ol.random_elems(np.random.default_rng(64), n * cols)
    ref = x.copy()
    io = hal.copy_from(x)
    hal.batch_interpolate_ntt(io, cols)
    hal.zk_shift(io, cols)
    out = hal.alloc(4 * n * cols)
    hal.batch_expand_into_evaluate_ntt(out, io, cols, 2)
    oracle_mt.bxo_batch_interpolate_ntt(ref, cols, n)
    oracle_mt.bxo_zk_shift(ref, cols, n)
    assert np.array_equal(io.view(), ref)
    ref_out = np.zeros(4 * n * cols, np.uint32)
    oracle_mt.bxo_batch_expand_into_evaluate_ntt(ref_out, ref, cols, n, 2)
    assert np.array_equal(out.view(), ref_out)
    io.free()
    out.free()


def _run_bench_two_ranks_on_one_gpu(extra, dump, timeout=900):
    """BASELINE.json configs[2] on a single-GPU box: two ranks (torch.distributed over gloo) share device 0 and claim
    segments from the c10d ticket queue.  Returns (parsed JSON line, {rank: npz})."""
    import json
    import os
    import socket
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "2", "--dist-backend", "gloo", "--device", "0",
           "--no-cpu-baseline", "--no-agent-mode", "--dump", dump] + extra
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=env, cwd=root)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    return json.loads(line), {k: np.load(os.path.join(dump, f"rank{k}.npz")) for k in (0, 1)}


def test_work_stolen_batch_over_two_ranks_is_bit_exact(tmp_path):
    """configs[2] (a batch of independent segments work-stolen across ranks), N>1 path of bench.py on real hardware:
    every segment of the batch is proved exactly once, by whichever rank claimed it, and every seal equals the oracle's."""
    from boundless_amd.prover import Segment

    po2, batch = 12, 8
    out, dumps = _run_bench_two_ranks_on_one_gpu(["--steal", "--batch", str(batch), "--po2", str(po2), "--widths", "4,12,4", "--warmup", "1",
                                                  "--steps", "1", "--inflight", "2"], str(tmp_path))
    assert out["n_gpus"] == 2 and out["config"]["segments_proved"] == batch
    a, b = dumps[0]["indices"].tolist(), dumps[1]["indices"].tolist()
    assert sorted(a + b) == list(range(batch)) and not set(a) & set(b), (a, b)
    for k in (0, 1):
        for i in dumps[k]["indices"].tolist():
            want, _ = ol.prove_segment(po2, 4, 12, 4, Segment.synthetic(i, po2=po2).seed)
            assert np.array_equal(dumps[k][f"seal_{i}"], want), f"segment {i} proved by rank {k}"


def test_default_multi_rank_bench_path_static_split_is_bit_exact(tmp_path):
    """The command the driver's scaling run uses (`bench.py --gpus N --steps K --warmup W` under torch.distributed.run), here with
    two ranks on the one GPU: rank r proves segments r, r + 2, ... (fixed work per rank = weak scaling), the JSON line counts
    the segments of both ranks, and every seal equals the oracle's."""
    from boundless_amd.prover import Segment

    po2, steps, lanes = 12, 2, 2
    out, dumps = _run_bench_two_ranks_on_one_gpu(["--po2", str(po2), "--widths", "4,12,4", "--warmup", "1", "--steps", str(steps),
                                                  "--inflight", str(lanes)], str(tmp_path))
    total = 2 * steps * lanes
    assert out["n_gpus"] == 2 and out["scaling"] == "weak" and out["config"]["segments_proved"] == total
    assert out["steps"] == steps and abs(out["value"] - total / (out["ms_per_step"] * steps / 1e3)) / out["value"] < 1e-6
    a, b = dumps[0]["indices"].tolist(), dumps[1]["indices"].tolist()
    assert a == list(range(0, total, 2)) and b == list(range(1, total, 2))
    for k in (0, 1):
        for i in dumps[k]["indices"].tolist():
            want, _ = ol.prove_segment(po2, 4, 12, 4, Segment.synthetic(i, po2=po2).seed)
            assert np.array_equal(dumps[k][f"seal_{i}"], want), f"segment {i} proved by rank {k}"
    # at N > 1 the same command also measures the other multi-GPU design: rank 0 hands the GPUs (here: two device slots on the one
    # GPU) to a child `bench.py --native-agent`, the other ranks wait on the c10d store, the child's line rides along
    na = out["native_agent"]
    assert "error" not in na, na
    assert na["n_gpus"] == 2 and na["config"]["segments_proved"] == max(2, steps // 2) * lanes * 2
    assert sum(na["segments_per_device"].values()) == na["config"]["segments_proved"] and na["value"] > 0


def _run_bench_ranks(ranks, extra, timeout=900):
    import json
    import os
    import socket
    import subprocess
    import sys
    import time

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={ranks}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", str(ranks), "--dist-backend", "gloo", "--device", "0",
           "--no-cpu-baseline", "--no-agent-mode"] + extra
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    t0 = time.time()
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=env, cwd=root)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    return json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]), time.time() - t0


def test_world_size_8_path_of_the_drivers_command_on_one_gpu():
    """The first real 8-GPU run of `bench.py --gpus 8` must not be the first time HEAD's N = 8 path executes: eight
    torch.distributed ranks (gloo) share the ONE GPU of the test box — rendezvous, static split, barriers, max over ranks, one row per
    rank, and the native-agent child that rank 0 spawns over eight device slots while the other ranks wait on the c10d store.  Then
    the same with that child made to fail at once: the primary line still comes out, the failure is reported in it, nobody hangs.
    Not a scaling measurement (compose.yml:113 runs one agent per GPU; here all eight sit on one)."""
    steps, lanes = 2, 1
    extra = ["--po2", "12", "--widths", "4,12,4", "--warmup", "1", "--steps", str(steps), "--inflight", str(lanes)]
    out, _ = _run_bench_ranks(8, extra)
    assert out["n_gpus"] == 8 and out["scaling"] == "weak" and out["config"]["segments_proved"] == 8 * steps * lanes
    assert out["backend"] == {"name": "gloo", "world_size": 8}
    assert [r["rank"] for r in out["per_rank"]] == list(range(8)) and all(r["proofs"] == steps * lanes for r in out["per_rank"])
    na = out["native_agent"]
    assert "error" not in na, na
    assert na["n_gpus"] == 8 and sum(na["segments_per_device"].values()) == na["config"]["segments_proved"] and na["value"] > 0
    bad, seconds = _run_bench_ranks(8, extra + ["--inject-child-failure"])
    assert bad["n_gpus"] == 8 and bad["value"] > 0 and len(bad["per_rank"]) == 8
    assert "error" in bad["native_agent"], bad["native_agent"]
    assert seconds < 300, f"a failing native-agent child held the ranks for {seconds:.0f}s"


def test_rccl_rendezvous_barrier_and_all_reduce_of_the_multi_rank_path(tmp_path):
    """The driver's N > 1 command uses the default backend (nccl = RCCL), which two ranks cannot share one GPU for.  What a
    one-GPU box can run of it is one rank with the process group forced on: RCCL initialises on the device, the barriers of
    the timed region and the max / sum all-reduces of the timing run on it, the ticket queue goes through the c10d store."""
    import json
    import os
    import socket
    import subprocess
    import sys

    from boundless_amd.prover import Segment

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    po2 = 12
    for extra, want_indices in ((["--steps", "2"], [0, 1, 2, 3]), (["--steps", "1", "--steal", "--batch", "5"], [0, 1, 2, 3, 4])):
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "1", "--force-dist", "--no-cpu-baseline",
               "--no-agent-mode", "--po2", str(po2), "--widths", "4,12,4", "--warmup", "1", "--inflight", "2", "--dump", str(tmp_path)] + extra
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=root)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
        out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
        assert out["config"]["rendezvous"] == "torch.distributed/nccl" and out["n_gpus"] == 1
        d = np.load(os.path.join(str(tmp_path), "rank0.npz"))
        assert sorted(d["indices"].tolist()) == want_indices
        for i in want_indices:
            want, _ = ol.prove_segment(po2, 4, 12, 4, Segment.synthetic(i, po2=po2).seed)
            assert np.array_equal(d[f"seal_{i}"], want)
