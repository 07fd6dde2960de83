"""One-session-per-GPU harness plumbing (replicas only).

The EKF state is monolithic, so multi-GPU means independent sessions, one per rank;
the only collectives are a start barrier and a gather of a small fixed-size result
record per rank, off the hot path.  Backend "nccl" (= RCCL over xGMI) on GPUs,
"gloo" in the CPU tests.
"""
from __future__ import annotations

import os

import numpy as np

RECORD_FIELDS = ("steps", "elapsed_s", "final_n", "pose_x", "pose_y", "pose_theta", "max_abs_err", "seed")


def init(backend: str | None = None):
    """Initialise torch.distributed from the torchrun environment.  Returns
    (dist module or None, rank, local_rank, world)."""
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world <= 1:
        return None, rank, local_rank, world
    import torch
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    kw = {}
    if backend == "nccl":
        kw["device_id"] = torch.device("cuda", local_rank)
    dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return dist, rank, local_rank, world


def _device(dist):
    import torch
    return torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")


def barrier(dist):
    if dist is not None:
        dist.barrier()


def max_over_ranks(dist, x: float) -> float:
    if dist is None:
        return float(x)
    import torch
    t = torch.tensor([float(x)], dtype=torch.float64, device=_device(dist))
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_records(dist, record: dict) -> list[dict]:
    """All-gather one fixed-size result record (RECORD_FIELDS, 64 bytes) per rank."""
    vec = np.array([float(record.get(k, 0.0)) for k in RECORD_FIELDS], dtype=np.float64)
    if dist is None:
        return [dict(zip(RECORD_FIELDS, vec.tolist()))]
    import torch
    t = torch.from_numpy(vec).to(_device(dist))
    outs = [torch.empty_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(outs, t)
    return [dict(zip(RECORD_FIELDS, o.cpu().numpy().tolist())) for o in outs]


def aggregate_updates_per_s(records: list[dict], elapsed_max: float) -> float:
    """Whole-job throughput: the steps all ranks processed / the slowest rank's time."""
    return sum(r["steps"] for r in records) / elapsed_max
