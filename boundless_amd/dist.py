"""Multi-GPU sharding of independent segments: one process per GPU, no data-path collective.

Reference model: one agent process per GPU (`CUDA_VISIBLE_DEVICES=$gpu /app/agent -t prove &`, compose.yml:113), all
claiming work from one queue (`taskdb::request_work`, bento/crates/workflow/src/lib.rs:373; `FOR UPDATE SKIP LOCKED`,
bento/crates/taskdb/migrations/9_request_work.sql:126-153).  Segments never exchange data, so the MI355X-native
equivalent needs no RCCL collective on the data path: torch.distributed supplies the rendezvous, barriers and the
max-over-ranks of the timing, and its c10d Store doubles as the atomic ticket counter of the claim-when-idle queue.
"""
import os
import time


def init_distributed(backend=None, force=False):
    """Returns (rank, world, local_rank, dist or None).  Rendezvous from RANK/WORLD_SIZE/MASTER_* (torchrun).

    A single process needs no process group; `force` creates one anyway (under torchrun), so that the RCCL start-up, barrier
    and all-reduce the N > 1 runs depend on can be exercised on a one-GPU box."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world <= 1 and not (force and "MASTER_PORT" in os.environ):
        return rank, 1, local_rank, None
    import torch
    import torch.distributed as dist

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    kwargs = {}
    if backend == "nccl":
        torch.cuda.set_device(local_rank)
        kwargs["device_id"] = torch.device("cuda", local_rank)
    if not dist.is_initialized():
        dist.init_process_group(backend=backend, **kwargs)
    return rank, world, local_rank, dist


class SegmentQueue:
    """Claim-when-idle queue of `total` independent segment indices shared by all ranks.

    mode "static": rank r proves indices r, r + world, r + 2*world, ... (fixed per-GPU work: weak scaling).
    mode "steal" : every claim takes the next ticket from an atomic counter in the c10d Store, so a slow GPU simply
                   claims fewer segments (the `request_work` semantics); each index is handed out exactly once.
    """

    def __init__(self, total, rank=0, world=1, dist=None, mode="static", name="segq"):
        self.total, self.rank, self.world, self.mode = total, rank, world, mode
        self._next_static = rank
        self._store = None
        self._key = f"{name}:next"
        self._local = 0
        if mode == "steal" and dist is not None:
            from torch.distributed import distributed_c10d

            self._store = distributed_c10d._get_default_store()
            if rank == 0:
                self._store.add(self._key, 0)
            dist.barrier()
        elif mode not in ("static", "steal"):
            raise ValueError(f"unknown queue mode {mode}")

    def claim(self):
        """Next segment index for this rank, or None when the queue is drained."""
        if self.mode == "static":
            i = self._next_static
            if i >= self.total:
                return None
            self._next_static += self.world
            return i
        if self._store is None:  # single process
            i = self._local
            self._local += 1
        else:
            i = self._store.add(self._key, 1) - 1
        return i if i < self.total else None


def barrier(dist, sync=None):
    if sync is not None:
        sync()
    if dist is not None:
        dist.barrier()


def timed_region(fn, dist=None, sync=None):
    """barrier + device sync, run fn(), device sync + barrier; returns MAX-over-ranks elapsed seconds."""
    barrier(dist, sync)
    t0 = time.perf_counter()
    result = fn()
    barrier(dist, sync)
    elapsed = time.perf_counter() - t0
    return max_over_ranks(elapsed, dist), result


def max_over_ranks(value, dist=None):
    if dist is None:
        return float(value)
    import torch

    dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
    t = torch.tensor([value], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value, dist=None):
    if dist is None:
        return float(value)
    import torch

    dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
    t = torch.tensor([value], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


def gather_over_ranks(values, dist=None):
    """Every rank contributes a fixed-length list of numbers; returns one list per rank, in rank order (a plain all-gather of a
    small tensor: works on nccl and gloo alike, unlike all_gather_object's pickling path)."""
    if dist is None:
        return [[float(v) for v in values]]
    import torch

    dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
    t = torch.tensor([float(v) for v in values], dtype=torch.float64, device=dev)
    out = [torch.empty_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return [[float(x) for x in o.cpu().tolist()] for o in out]
