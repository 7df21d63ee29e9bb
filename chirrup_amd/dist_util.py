"""Multi-GPU helpers: the decode path shards by REQUEST (independent replicas, one process per GPU,
no data-path collective -- SURVEY.md section 8e).  torch.distributed (RCCL on GPUs, gloo in the CPU
tests) is only used to bracket a timed region and to agree on its duration."""
import time
from typing import Callable, List, Sequence

import torch
import torch.distributed as dist


def shard_requests(n_requests: int, world: int, rank: int) -> List[int]:
    """Round-robin partition of request indices over ranks (every request on exactly one rank)."""
    return list(range(rank, n_requests, world))


LAST_LOCAL_SECONDS = 0.0      # this rank's own time of the last timed_region (before the max over ranks)


def gather_floats(x: float, device: torch.device) -> List[float]:
    """[x of rank 0, x of rank 1, ...] on every rank ([x] without a process group)."""
    if not dist.is_initialized():
        return [float(x)]
    on_gpu = device.type == "cuda" and dist.get_backend() == "nccl"
    t = torch.tensor([x], dtype=torch.float64, device=device if on_gpu else "cpu")
    out = [torch.empty_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return [float(o.item()) for o in out]


def timed_region(fn: Callable[[], None], steps: int, device: torch.device) -> float:
    """barrier -> sync -> `steps` calls -> sync -> barrier; returns the MAX over ranks of the
    elapsed seconds (the driver's bench contract)."""
    global LAST_LOCAL_SECONDS
    grouped = dist.is_initialized()          # also with a single rank: the same calls run whenever a group exists
    sync = (lambda: torch.cuda.synchronize(device)) if device.type == "cuda" else (lambda: None)
    if grouped:
        dist.barrier()
    sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    sync()
    LAST_LOCAL_SECONDS = time.perf_counter() - t0
    if grouped:
        dist.barrier()
    dt = time.perf_counter() - t0
    if grouped:
        on_gpu = device.type == "cuda" and dist.get_backend() == "nccl"
        t = torch.tensor([dt], dtype=torch.float64, device=device if on_gpu else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    return dt


def whole_job_throughput(units_per_rank_step: Sequence[int] | int, steps: int, seconds: float) -> float:
    """value = units all ranks processed / max-over-ranks time."""
    world = dist.get_world_size() if dist.is_initialized() else 1
    per = units_per_rank_step if isinstance(units_per_rank_step, int) else sum(units_per_rank_step) / max(world, 1)
    return world * per * steps / seconds
