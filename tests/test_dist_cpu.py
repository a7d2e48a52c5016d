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
