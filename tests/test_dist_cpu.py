"""CPU, world_size 2 over gloo: the N>1 path of bench.py (sharding, claim-when-idle queue, max-over-ranks timing)."""
import json
import os
import socket
import subprocess
import sys
import tempfile
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def run_world(script, world=2, timeout=180):
    port = free_port()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), "-"]
    outdir = tempfile.mkdtemp(prefix="bxdist")
    env = dict(os.environ, PYTHONPATH=ROOT, OMP_NUM_THREADS="1", BX_TEST_OUT=outdir)
    # torchrun cannot read a script from stdin: write it to a temp file
    with tempfile.NamedTemporaryFile("w", suffix=".py", delete=False) as f:
        f.write(textwrap.dedent(script))
        path = f.name
    cmd[-1] = path
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=env)
    finally:
        os.unlink(path)
    assert r.returncode == 0, r.stdout + r.stderr
    # one file per rank: concurrent prints to a shared stdout can interleave
    return [json.load(open(os.path.join(outdir, f))) for f in sorted(os.listdir(outdir))]


WORKER = """
import json, time
from boundless_amd.dist import init_distributed, SegmentQueue, timed_region, gather_over_ranks
rank, world, local_rank, dist = init_distributed(backend="gloo")
out = {"rank": rank, "world": world, "backend_world": dist.get_world_size()}
# the per-rank rows of the bench line: every rank ends up with every rank's (rank, device, proofs, seconds)
out["rows"] = gather_over_ranks([rank, 10 + rank, 3 * rank + 1, 0.5 + rank], dist)
for mode in ("static", "steal"):
    q = SegmentQueue(total=23, rank=rank, world=world, dist=dist, mode=mode, name=mode)
    mine = []
    def work():
        while True:
            i = q.claim()
            if i is None:
                break
            mine.append(i)
            time.sleep(0.002 if rank == 0 else 0.02)   # rank 1 is a 10x slower "GPU"
        return len(mine)
    elapsed, n = timed_region(work, dist)
    out[mode] = {"mine": mine, "elapsed": elapsed}
import os
json.dump(out, open(os.path.join(os.environ["BX_TEST_OUT"], f"rank{rank}.json"), "w"))
dist.destroy_process_group()
"""


def test_two_ranks_shard_every_segment_exactly_once():
    res = run_world(WORKER, world=2)
    assert sorted(r["rank"] for r in res) == [0, 1]
    by_rank = {r["rank"]: r for r in res}
    for r in res:  # bench.py's `per_rank` / `backend.world_size`: identical on both ranks, in rank order
        assert r["backend_world"] == 2 and r["rows"] == [[0.0, 10.0, 1.0, 0.5], [1.0, 11.0, 4.0, 1.5]]
    # static: rank r gets r, r+2, ...
    assert by_rank[0]["static"]["mine"] == list(range(0, 23, 2))
    assert by_rank[1]["static"]["mine"] == list(range(1, 23, 2))
    # steal: disjoint cover of 0..22, and the fast rank claimed more
    a, b = by_rank[0]["steal"]["mine"], by_rank[1]["steal"]["mine"]
    assert sorted(a + b) == list(range(23)) and not set(a) & set(b)
    assert len(a) > len(b)
    # max-over-ranks timing: both ranks report the same (maximum) elapsed time
    for mode in ("static", "steal"):
        assert abs(by_rank[0][mode]["elapsed"] - by_rank[1][mode]["elapsed"]) < 1e-9
    # with a 10x slower rank, stealing finishes sooner than the static split
    assert by_rank[0]["steal"]["elapsed"] < by_rank[0]["static"]["elapsed"]


def test_single_process_queue():
    from boundless_amd.dist import SegmentQueue, timed_region

    for mode in ("static", "steal"):
        q = SegmentQueue(5, mode=mode)
        got = []
        while (i := q.claim()) is not None:
            got.append(i)
        assert got == [0, 1, 2, 3, 4]
    elapsed, r = timed_region(lambda: 7)
    assert r == 7 and elapsed >= 0
    from boundless_amd.dist import gather_over_ranks

    assert gather_over_ranks([0, 3, 60, 2.5]) == [[0.0, 3.0, 60.0, 2.5]]


JOB_WORKER = """
import json, os
import numpy as np
from boundless_amd import agent as ag
from boundless_amd.dist import init_distributed, distributed_job
from boundless_amd.prover import Segment, SegmentReceipt

def fake_seal(po2, seed):
    return (np.arange(16, dtype=np.uint32) * np.uint32(2654435761) + np.uint32(seed & 0xFFFFFFFF) + np.uint32(po2)).astype(np.uint32)

class FakeProver:
    def prove_segment(self, seg):
        return SegmentReceipt(seal=fake_seal(seg.po2, seg.seed), index=seg.index, po2=seg.po2)

rank, world, local_rank, dist = init_distributed(backend="gloo")
a = ag.Agent(prover=FakeProver(), verify=False, poll_time=0.002, inflight=2, join_po2=11, also_streams="aux")
res = distributed_job(a, 8, lambda i: Segment.synthetic(i, po2=13), rank=rank, world=world, dist=dist)
out = {k: v for k, v in res.items() if k != "rollup"}
if rank == 0:
    out["rollup_seal"] = res["rollup"].seal.tolist()
    out["rollup_po2"] = res["rollup"].po2
    out["keys_left"] = sorted(a.store.keys())
else:
    out["keys_left"] = sorted(a.store.keys())
a.close()
json.dump(out, open(os.path.join(os.environ["BX_TEST_OUT"], f"rank{rank}.json"), "w"))
dist.destroy_process_group()
"""


def test_a_job_sharded_over_two_ranks_joins_its_subtree_roots_after_one_all_gather():
    """north_star's only collective: segments shard over the ranks (one process per GPU), every rank joins its own subtree, the
    subtree ROOTS are all-gathered (gloo here, RCCL on GPUs) and rank 0 joins them, resolves and finalizes.  With power-of-two shares
    the tree is the single-process planner's, so the rollup seal is the same chain of stand-in joins."""
    import numpy as np

    from boundless_amd import agent as ag
    from boundless_amd.planner import Planner

    def fake_seal(po2, seed):
        return (np.arange(16, dtype=np.uint32) * np.uint32(2654435761) + np.uint32(seed & 0xFFFFFFFF) + np.uint32(po2)).astype(np.uint32)

    res = run_world(JOB_WORKER, world=2)
    by_rank = {r["rank"]: r for r in res}
    assert by_rank[0]["segments"] == by_rank[1]["segments"] == 4 and by_rank[0]["tasks"] == 7  # 4 proves + 3 joins per rank, no finalize
    assert by_rank[0]["top_joins"] == 1 and by_rank[0]["top_tasks"] == 3  # the top join, resolve, finalize (the two leaves were handed over done)
    # the chain of the single-process plan over the same 8 segments
    p, seals, root = Planner(), {}, None
    base = 0xB0D1E550000
    for i in range(8):
        p.enqueue_segment()
    p.finish()
    leaf = 0
    for k in range(p.task_count()):
        t = p.get_task(k)
        if t.command == "Segment":
            seals[t.task_number] = fake_seal(13, base + leaf)
            leaf += 1
    for k in range(p.task_count()):
        t = p.get_task(k)
        if t.command == "Join":
            seals[t.task_number] = fake_seal(11, ag.join_seed(seals[t.depends_on[0]], seals[t.depends_on[1]]))
        elif t.command == "Finalize":
            root = t.depends_on[0]
    assert by_rank[0]["rollup_seal"] == seals[root].tolist() and by_rank[0]["rollup_po2"] == 11
    # what each rank's store is left with: its own subtree root; rank 0 also the top job's root and the rollup
    assert by_rank[1]["keys_left"] == ["job:dj-r1:synthetic_receipts:6"]
    assert sorted(by_rank[0]["keys_left"]) == sorted(["job:dj-r0:synthetic_receipts:6", "job:dj-top:synthetic_receipts:2", "receipts/stark/dj-top.synthetic"])
    assert by_rank[0]["root_receipt_bytes"] == 4 * (3 + 16)


def _bench(argv, env_extra=None, timeout=240):
    import subprocess
    import sys

    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + argv, capture_output=True, text=True, env=env, timeout=timeout)


def test_bench_gpus_n_without_a_launcher_starts_n_ranks():
    """`python bench.py --gpus 2` with no torchrun around it must not measure one GPU under an N = 2 flag: it
    re-executes itself under torch.distributed.run with one process per GPU.  --rendezvous-only + gloo: no GPU needed."""
    r = _bench(["--gpus", "2", "--dist-backend", "gloo", "--rendezvous-only"])
    assert r.returncode == 0, r.stdout + r.stderr
    assert "re-executing as one process per GPU" in r.stderr
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line == {"rendezvous_only": True, "n_gpus": 2, "ranks_counted": 2, "flag_gpus": 2, "backend": "gloo"}
    # N = 1 stays one plain process, no process group
    r = _bench(["--gpus", "1", "--rendezvous-only"])
    assert r.returncode == 0 and "re-executing" not in r.stderr
    assert json.loads(r.stdout.splitlines()[-1]) == {"rendezvous_only": True, "n_gpus": 1, "ranks_counted": 1, "flag_gpus": 1, "backend": None}


def test_bench_refuses_a_launcher_whose_world_size_disagrees_with_gpus():
    r = _bench(["--gpus", "8", "--dist-backend", "gloo", "--rendezvous-only"], env_extra={"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "--gpus 8 but the launcher started WORLD_SIZE=1" in (r.stderr + r.stdout)
