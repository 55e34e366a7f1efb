"""Multi-GPU layer of the hot path: frame sharding + the timing protocol bench.py uses.

Frames are independent units: rank r colours frames r, r+G, r+2G, ... on its own GPU with a
full weight replica.  There is NO data-path collective; torch.distributed (backend "nccl" = RCCL on ROCm,
"gloo" in the CPU tests) is only used for the barrier and the max-over-ranks reduction of the elapsed time.
"""
import os
import time


def env_rank():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def shard_frames(n_frames, rank, world_size):
    """frame indices owned by `rank` (round-robin: n -> GPU n mod G)."""
    if not (0 <= rank < world_size):
        raise ValueError("rank out of range")
    return list(range(rank, n_frames, world_size))


def gather_order(n_frames, world_size):
    """inverse of shard_frames: for each global frame, (rank, local index)."""
    return [(n % world_size, n // world_size) for n in range(n_frames)]


def init_dist(backend, local_rank):
    import torch
    import torch.distributed as dist
    if backend == "nccl":
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    else:
        dist.init_process_group(backend)
    return dist


def timed_steps(step_fn, steps, warmup, sync_fn, dist=None, device=None):
    """W untimed warm-up steps, then EXACTLY `steps` timed steps bracketed by sync_fn (device sync) + barrier
    on both sides; returns the MAX elapsed seconds over ranks."""
    import torch
    for i in range(warmup):
        step_fn(i)

    def fence():
        sync_fn()
        if dist is not None:
            dist.barrier()
        sync_fn()
    fence()
    t0 = time.perf_counter()
    for i in range(steps):
        step_fn(warmup + i)
    fence()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device or "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    return elapsed
