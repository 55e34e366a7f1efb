#!/usr/bin/env python3
"""bench.py — colorized frames/sec/GPU @1080p, DeOldify "stable", render_factor=35 (BASELINE.json configs[1]).

One step = one pass of the whole hot path over one batch of synthetic 1080p frames that are ALREADY resident
in HBM: Spline64 squash to 560x560 -> video U-Net pass -> stable U-Net pass -> YUV merges + Image.blend
-> Spline64 back to 1920x1080 fused with the luma re-attach (havc_colorize_clip, include/havc_mi355.h).
Nothing is skipped or cached inside the timed region; weights are seeded-synthetic (no real weights exist
offline), activations fp16 with fp32 MFMA accumulation.

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

Multi-GPU: frames are independent — every rank colours its own shard on its own GPU with a
full weight replica, no data-path collective; RCCL is used only for the barrier / max-over-ranks timing.
Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

METRIC = "colorized frames/sec/GPU @1080p (DeOldify stable rf=35); CIEDE2000 vs ref"
RENDER_FACTOR, WIDTH, HEIGHT = 35, 1920, 1080
PEAK_F16_TFLOPS = 2500.0            # MI355X dense fp16/bf16 MFMA peak (MI355X_MICROARCH.md: ~2.5 PF dense)
TAG_TAIL_RES = 1


def cpu_baseline_and_parity(sds, frame, gpu_out, threads):
    """Time the CPU oracle (fp32 PyTorch restatement + numpy tail) on ONE frame of the same workload and use
    its output as the parity reference for the GPU result of that frame."""
    import torch
    from oracle import imaging, pipeline
    torch.set_num_threads(threads)
    t0 = time.time()
    ref = pipeline.colorize_frame_fullsize(sds, "stable", frame, RENDER_FACTOR, 0.5)
    dt = time.time() - t0
    de = imaging.delta_e00_images(gpu_out, ref)
    d = np.abs(gpu_out.astype(np.int32) - ref.astype(np.int32))
    parity = {"ciede2000_mean": round(float(de.mean()), 4), "ciede2000_p99": round(float(np.percentile(de, 99)), 4),
              "ciede2000_max": round(float(de.max()), 4), "bytes_within_1lsb": round(float((d <= 1).mean()), 5),
              "bytes_within_2lsb": round(float((d <= 2).mean()), 5), "frames_checked": 1, "against": "oracle (CPU fp32 port)"}
    base = {"value": round(1.0 / dt, 5), "unit": "frames/s", "cores": threads, "kind": "port",
            "sample": f"1 frame of the 1080p clip (2 U-Net passes at 560x560 fp32 + Spline64/YUV tail), {dt:.1f} s"}
    return base, parity


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=16, help="frames per step per GPU")
    ap.add_argument("--clip-frames", type=int, default=16, help="distinct synthetic frames resident per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-threads", type=int, default=32, help="threads for the CPU-oracle baseline leg")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.gpus > 1 and world == 1:
        # started without a launcher: become the launcher (a CHILD process; nothing has touched the GPU yet) and relay its exit code
        import subprocess
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
               "--master-port", os.environ.get("MASTER_PORT", "29533"), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd))

    import torch
    dist = None
    if world > 1:
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))   # nccl == RCCL on ROCm

    from vsdeoldify_amd import _native as nat
    from vsdeoldify_amd.clip import ClipColorizer, synthetic_gray_frame
    from vsdeoldify_amd.synth import synth_state_dict

    sds = {"video": synth_state_dict("wide", 1), "stable": synth_state_dict("wide", 2)}
    cc = ClipColorizer("stable", RENDER_FACTOR, 0.5, device_index=local_rank, state_dicts=sds, max_batch=args.batch)
    ctx = cc.ctx

    # ---- synthetic clip, resident in HBM before the timed region --------------------------------
    n_clip = max(args.clip_frames, args.batch)
    n_clip = (n_clip + args.batch - 1) // args.batch * args.batch
    frames = np.stack([synthetic_gray_frame(rank * n_clip + i, WIDTH, HEIGHT) for i in range(n_clip)])
    fbytes = WIDTH * HEIGHT * 3
    d_src, d_dst = ctx.dev_alloc(frames.nbytes), ctx.dev_alloc(frames.nbytes)
    ctx.dev_upload(d_src, frames)
    n_batches = n_clip // args.batch

    def off(p, frame_idx):
        import ctypes
        return ctypes.c_void_p(p.value + frame_idx * fbytes)

    def step(i):
        f0 = (i % n_batches) * args.batch
        cc.colorize_device(off(d_src, f0), off(d_dst, f0), args.batch, WIDTH, HEIGHT)

    def sync_all():
        ctx.synchronize()
        if dist is not None:
            dist.barrier()
        if torch.cuda.is_available():
            torch.cuda.synchronize()

    for i in range(args.warmup):
        step(i)
    ctx.reset_stats()
    nat.check(ctx.lib.havc_tag_timing_enable(ctx.h, TAG_TAIL_RES, 1), ctx.h)
    sync_all()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(args.warmup + i)
    sync_all()
    elapsed = time.perf_counter() - t0
    import ctypes
    avg_ms, launches = ctypes.c_double(), ctypes.c_int64()
    nat.check(ctx.lib.havc_tag_timing_read(ctx.h, ctypes.byref(avg_ms), ctypes.byref(launches)), ctx.h)
    nat.check(ctx.lib.havc_tag_timing_enable(ctx.h, TAG_TAIL_RES, 0), ctx.h)
    st = ctx.stats()

    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=f"cuda:{local_rank}")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    total_frames = args.steps * args.batch * world
    S = RENDER_FACTOR * 16
    c = 259
    conv_flops = 2.0 * args.batch * S * S * c * c * 9            # algorithmic FLOPs of ONE tail res-conv launch
    achieved = conv_flops / (avg_ms.value * 1e-3) / 1e12 if avg_ms.value > 0 else 0.0
    traffic = None                                   # HBM bytes per launch from PMC passes on this same command
    try:
        pmc = json.load(open(os.path.join(ROOT, "profiles", "r1_tail_conv_pmc.json")))
        if pmc.get("frames_per_launch") == args.batch:
            traffic = pmc["traffic_bytes_per_launch"]
    except Exception:
        pass
    out = {
        "metric": METRIC, "value": round(total_frames / elapsed, 3), "unit": "frames/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f16", "data": "synthetic",
        "config": {"workload": "DeOldify 'stable' generator, render_factor=35, 1080p clip (BASELINE.json configs[1])",
                   "frames_per_step_per_gpu": args.batch, "net_input": f"{S}x{S}", "unet_passes_per_frame": 2,
                   "algorithmic_gflop_per_frame": 2759.32, "weights": "seeded synthetic (wide resnet101 x2)",
                   "parallelism": f"frame-sharded x{world}, weight replica per GPU, no collective"},
        "whole_path_tflops": round(total_frames * 2759.32e9 / elapsed / 1e12 / world, 2),
        "roofline": {"bound": "mfma", "achieved": round(achieved, 2), "peak": PEAK_F16_TFLOPS, "unit": "TFLOP/s",
                     "frac": round(achieved / PEAK_F16_TFLOPS, 4), "traffic": traffic,
                     "kernel": "conv_pipe_kernel<2,4,8,1> (layers.10 res-block 3x3 259->259 @560x560, 2 launches/pass; the 2nd also runs layers.11/12 in its epilogue)",
                     "launches_timed": int(launches.value), "avg_launch_ms": round(avg_ms.value, 4),
                     "flops_per_launch": conv_flops},
        "gpu_ms_per_frame": round(st.total_ms / max(st.frames, 1), 4),
    }
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        got = np.empty((1, HEIGHT, WIDTH, 3), np.uint8)
        step(0)
        ctx.dev_download(got, off(d_dst, 0))
        base, parity = cpu_baseline_and_parity(sds, frames[0], got[0], min(os.cpu_count() or 1, args.cpu_threads))
        out["cpu_baseline"] = base
        out["parity"] = parity
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out))


if __name__ == "__main__":
    main()
