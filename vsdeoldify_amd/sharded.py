"""Multi-GPU layer of the hot path: frame sharding, the clip runner and the timing protocol bench.py uses.

Frames are independent units: rank r colours frames r, r+G, r+2G, ... on its own GPU with a full weight replica;
nothing is exchanged INSIDE the path.  Two users:
  * bench.py --gpus N: every rank colours its own resident batches; torch.distributed (backend "nccl" = RCCL on ROCm, "gloo" in the
    CPU tests) provides only the barrier and the max-over-ranks reduction of the elapsed time (timed_steps);
  * colorize_clip_sharded: ONE clip that lives on rank 0 goes through all GPUs and comes back in frame order — exactly one RCCL scatter of
    the gray frames and one RCCL gather of the coloured ones around the path (device tensors, used in place by libhavc through their
    pointers: DeviceClipFn orders the library's stream against torch's with events, no host synchronisation).
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


# ---- a clip through N GPUs: scatter -> colour -> gather, frame order preserved -----------------------------------------------
def colorize_clip_sharded(frames, colorize_fn, dist=None, rank=0, world_size=1, device="cpu", n_frames=None, frame_shape=None, force_collectives=False):
    """Colour a clip on `world_size` ranks and return it, in frame order, on rank 0 (None on the other ranks).

    frames      rank 0: uint8 tensor / array [n, h, w, 3] (host or device); other ranks: None (pass n_frames / frame_shape, or
                let rank 0 broadcast them)
    colorize_fn maps a uint8 torch tensor [m, h, w, 3] ON `device` to a tensor of the same shape on the same device -- on the
                MI355X this is `DeviceClipFn(colorizer)` below, which hands the tensor's device pointer to libhavc_mi355
                (every entry point takes device pointers): the shard never leaves HBM between the collectives
    Frames are independent: frame i goes to rank i mod G (shard_frames); the only exchange is ONE scatter of
    the gray frames and ONE gather of the coloured frames (RCCL over xGMI with backend "nccl"; "gloo" in the CPU tests).
    Shards are padded to ceil(n / G) frames so that the collectives see equal sizes; padding frames are not coloured.
    force_collectives: run the scatter / gather even with ONE rank (a single-GPU box can then exercise the RCCL leg end to end)."""
    import torch
    if dist is None or (world_size == 1 and not force_collectives):
        t = torch.as_tensor(frames).to(device)
        return colorize_fn(t)
    meta = torch.zeros(4, dtype=torch.int64, device=device)
    if rank == 0:
        t = torch.as_tensor(frames)
        meta = torch.tensor(list(t.shape), dtype=torch.int64, device=device)
    if n_frames is None or frame_shape is None:
        dist.broadcast(meta, src=0)
        n, h, w, c = (int(v) for v in meta.tolist())
    else:
        n, (h, w, c) = n_frames, frame_shape
    m = (n + world_size - 1) // world_size                      # frames per padded shard
    mine = shard_frames(n, rank, world_size)
    shard = torch.empty((m, h, w, c), dtype=torch.uint8, device=device)
    scatter_list = None
    if rank == 0:
        t = t.to(device)
        scatter_list = []
        for r in range(world_size):
            s = torch.zeros((m, h, w, c), dtype=torch.uint8, device=device)
            idx = shard_frames(n, r, world_size)
            if idx:
                s[:len(idx)] = t[idx]
            scatter_list.append(s)
    dist.scatter(shard, scatter_list, src=0)
    out = torch.zeros_like(shard)
    failure = None
    if mine:
        try:
            out[:len(mine)] = colorize_fn(shard[:len(mine)].contiguous())
        except Exception as e:                       # a failing rank still takes part in the collectives below: nobody hangs
            failure = e
    gather_list = [torch.empty_like(out) for _ in range(world_size)] if rank == 0 else None
    dist.gather(out, gather_list, dst=0)
    flag = torch.tensor([1 if failure is not None else 0], dtype=torch.int32, device=device)
    dist.all_reduce(flag, op=dist.ReduceOp.MAX)
    if int(flag.item()):
        raise RuntimeError(f"colorize_clip_sharded: a rank failed to colour its shard (rank {rank}: {failure!r})") from failure
    if rank != 0:
        return None
    result = torch.empty((n, h, w, c), dtype=torch.uint8, device=device)
    for r in range(world_size):
        idx = shard_frames(n, r, world_size)
        if idx:
            result[idx] = gather_list[r][:len(idx)]
    return result


class DeviceClipFn:
    """colorize_fn for colorize_clip_sharded on a GPU rank: wraps anything with `colorize_clip(DeviceImage) -> DeviceImage`
    (HAVCFrameColorizer) or a ClipColorizer.  The torch tensor's storage is used in place through its device pointer."""

    def __init__(self, colorizer):
        self.colorizer = colorizer

    def __call__(self, t):
        import ctypes
        import torch
        from .device import DeviceImage
        assert t.is_cuda and t.dtype == torch.uint8 and t.is_contiguous()
        ctx = self.colorizer.ctx
        # the scatter wrote `t` on torch's stream and the gather will read `out` there; libhavc works on its own (non-blocking) stream:
        # order the two with events in both directions — nothing blocks the host
        mine = torch.cuda.ExternalStream(ctx.stream_ptr(), device=t.device)
        theirs = torch.cuda.current_stream(t.device)
        mine.wait_stream(theirs)
        out = torch.empty_like(t)
        src = DeviceImage(ctx, tuple(t.shape), ctypes.c_void_p(t.data_ptr()))
        if hasattr(self.colorizer, "colorize_device"):               # ClipColorizer: havc_colorize_clip
            n, h, w, _ = t.shape
            self.colorizer.colorize_device(src.ptr, ctypes.c_void_p(out.data_ptr()), n, w, h)
        else:
            res = self.colorizer.colorize_clip(src)
            DeviceImage(ctx, tuple(out.shape), ctypes.c_void_p(out.data_ptr())).copy_from(res)
        theirs.wait_stream(mine)
        t.record_stream(mine)                                        # the caching allocator must not recycle `t` / `out` under the library's work
        out.record_stream(mine)
        return out
