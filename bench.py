#!/usr/bin/env python3
"""bench.py — colorized frames/sec @1080p, DeOldify "stable", render_factor=35 (BASELINE.json configs[1]).

One step = one pass of the whole hot path over one batch of synthetic 1080p frames that are ALREADY resident
in HBM: Spline64 squash to 560x560 -> video U-Net pass -> stable U-Net pass -> YUV merges + Image.blend
-> Spline64 back to 1920x1080 fused with the luma re-attach (havc_colorize_clip, include/havc_mi355.h).
Nothing is skipped or cached inside the timed region; weights are seeded-synthetic (no real weights exist
offline), activations fp16 with fp32 MFMA accumulation.

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W
  python bench.py --config c4          # BASELINE configs[3]: DeOldify + DDColor merge (combine_method=2) at 1080p

`value` is the whole-job rate over the EXACTLY K timed steps (sum over all ranks; `value_per_gpu` = value / n_gpus).
Beside it the line carries, measured in the same process after the timed region (rank 0, N = 1):
  sustained      the same step looped for >= --sustain-seconds (default 30 s): first-second and steady-state rates
  pcie_inclusive host frames in -> host frames out through havc_colorize_clip_host (pinned memory, uploads / passes /
                 downloads of consecutive batches overlapped on three streams)
  batch1         ModelImageRender.get_transformed_image on one 560x560 PIL image per call: the rate a VapourSynth
                 ModifyFrame selector sees (vsslib/vsmodels.py:219-230)
  cpu_baseline / parity   the CPU oracle on one frame of the same clip, and the GPU frame against it

Multi-GPU: frames are independent — every rank colours its own shard on its own GPU with a
full weight replica, no data-path collective; RCCL is used only for the barrier / max-over-ranks timing.
Rank 0 prints ONE JSON line.
"""
import argparse
import ctypes
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
PMC_FILE = os.path.join("profiles", "r2_tail_conv_pmc.json")


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
              "ciede2000_max": round(float(de.max()), 4), "pixels_with_dE_below_1": round(float((de < 1.0).mean()), 5),
              "bytes_within_1lsb": round(float((d <= 1).mean()), 5), "bytes_within_2lsb": round(float((d <= 2).mean()), 5),
              "frames_checked": 1, "against": "oracle (CPU fp32 port)",
              "note": "fp16 MFMA operands; floor / decomposition in profiles/r2_precision_study.txt"}
    base = {"value": round(1.0 / dt, 5), "unit": "frames/s", "cores": threads, "kind": "port",
            "sample": f"1 frame of the 1080p clip (2 U-Net passes at 560x560 fp32 + Spline64/YUV tail), {dt:.1f} s"}
    return base, parity


def off(p, nbytes):
    return ctypes.c_void_p(p.value + nbytes)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=32, help="frames per step per GPU")
    ap.add_argument("--clip-frames", type=int, default=32, help="distinct synthetic frames resident per GPU")
    ap.add_argument("--config", default="c2", choices=["c2", "c3", "c4"],
                    help="c2 = BASELINE configs[1] (headline), c3 = configs[2] (DDColor large, input 512), c4 = configs[3] (DeOldify+DDColor merge)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the sustained / PCIe-inclusive / batch-1 legs")
    ap.add_argument("--sustain-seconds", type=float, default=30.0)
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

    if args.config in ("c3", "c4"):
        return bench_c4(args, rank, local_rank, world, dist, ddcolor_only=args.config == "c3")

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

    def step(i):
        f0 = (i % n_batches) * args.batch
        cc.colorize_device(off(d_src, f0 * fbytes), off(d_dst, f0 * fbytes), args.batch, WIDTH, HEIGHT)

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
    # four tagged convs per step (two passes x two res-block convs); a conv whose operands exceed a 32-bit buffer descriptor runs as
    # several equal launches (32 frames of the 560 x 560 tail = 2 x 16) and every launch has its own event pair
    frames_per_launch = args.steps * args.batch * 4 / max(int(launches.value), 1)
    conv_flops = 2.0 * frames_per_launch * S * S * c * c * 9     # algorithmic FLOPs of ONE tail res-conv launch
    achieved = conv_flops / (avg_ms.value * 1e-3) / 1e12 if avg_ms.value > 0 else 0.0
    # HBM bytes per launch of the dominant kernel: NOT measured in this run (a PMC pass cannot share a run with the timing);
    # taken from the committed result of separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over this same command
    # (tools/pmc_bench.sh), gfx950 correction applied there; null when that file is missing or was taken at another batch.
    traffic, traffic_source = None, None
    try:
        pmc = json.load(open(os.path.join(ROOT, PMC_FILE)))
        if pmc.get("frames_per_launch") == round(frames_per_launch):
            traffic = pmc["traffic_bytes_per_launch"]
            traffic_source = f"{PMC_FILE}: separate rocprofv3 --pmc passes of this command (tools/pmc_bench.sh), not this run"
    except Exception:
        pass
    out = {
        "metric": METRIC, "value": round(total_frames / elapsed, 3), "unit": "frames/s (sum over n_gpus)", "n_gpus": world,
        "value_per_gpu": round(total_frames / elapsed / world, 3),
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f16", "data": "synthetic",
        "config": {"workload": "DeOldify 'stable' generator, render_factor=35, 1080p clip (BASELINE.json configs[1])",
                   "frames_per_step_per_gpu": args.batch, "net_input": f"{S}x{S}", "unet_passes_per_frame": 2,
                   "algorithmic_gflop_per_frame": 2759.32, "weights": "seeded synthetic (wide resnet101 x2)",
                   "conv_tile_autotune": os.environ.get("HAVC_AUTOTUNE", "1") != "0",
                   "parallelism": f"frame-sharded x{world}, weight replica per GPU, no collective"},
        "whole_path_tflops": round(total_frames * 2759.32e9 / elapsed / 1e12 / world, 2),
        "roofline": {"bound": "mfma", "achieved": round(achieved, 2), "peak": PEAK_F16_TFLOPS, "unit": "TFLOP/s",
                     "frac": round(achieved / PEAK_F16_TFLOPS, 4), "traffic": traffic, "traffic_source": traffic_source,
                     "kernel": "conv_pipe_kernel<2,4,8,1> (layers.10 res-block 3x3 259->259 @560x560, 2 launches/pass; the 2nd also runs layers.11/12 in its epilogue)",
                     "launches_timed": int(launches.value), "frames_per_launch": round(frames_per_launch, 2), "avg_launch_ms": round(avg_ms.value, 4),
                     "flops_per_launch": conv_flops},
        "gpu_ms_per_frame": round(st.total_ms / max(st.frames, 1), 4),
    }
    if rank == 0 and world == 1 and not args.no_extras:
        out.update(extras(args, cc, ctx, frames, d_src, d_dst, fbytes, n_batches, sds))
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        got = np.empty((1, HEIGHT, WIDTH, 3), np.uint8)
        step(0)
        ctx.dev_download(got, d_dst)
        base, parity = cpu_baseline_and_parity(sds, frames[0], got[0], min(os.cpu_count() or 1, args.cpu_threads))
        out["cpu_baseline"] = base
        out["parity"] = parity
    if dist is not None:
        try:
            sg = sharded_clip_leg(cc, dist, rank, world, local_rank, torch)
            if rank == 0:
                out["clip_scatter_gather"] = sg
        except Exception as e:                      # never lose the headline line to the optional leg
            if rank == 0:
                out["clip_scatter_gather"] = {"error": f"{type(e).__name__}: {e}"}
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out))


def sharded_clip_leg(cc, dist, rank, world, local_rank, torch):
    """N > 1: ONE clip that lives on rank 0 goes through all GPUs and comes back in frame order (vsdeoldify_amd/sharded.py:
    RCCL scatter of the gray frames, frame i -> rank i mod N, RCCL gather of the coloured frames).  Frame order is verified
    through the luma: vs_recover_clip_luma keeps the source's cv2 Y plane, so Y(out[i]) must equal Y(in[i])."""
    from vsdeoldify_amd import sharded
    from vsdeoldify_amd.clip import synthetic_gray_frame
    n = cc.max_batch * world
    dev = f"cuda:{local_rank}"
    frames = torch.from_numpy(np.stack([synthetic_gray_frame(1000 + i, WIDTH, HEIGHT) for i in range(n)])).to(dev) if rank == 0 else None
    fn = sharded.DeviceClipFn(cc)
    sharded.colorize_clip_sharded(frames, fn, dist, rank, world, dev)                # warm: RCCL communicators, staging
    torch.cuda.synchronize()
    dist.barrier()
    t0 = time.perf_counter()
    res = sharded.colorize_clip_sharded(frames, fn, dist, rank, world, dev)
    torch.cuda.synchronize()
    dist.barrier()
    dt = time.perf_counter() - t0
    if rank != 0:
        return None
    a, b = frames.cpu().numpy(), res.cpu().numpy()

    def luma(x):                                                                     # cv2 RGB2YUV Y (oracle/cvcolor.py constants)
        x = x.astype(np.int64)
        return (x[..., 0] * 4899 + x[..., 1] * 9617 + x[..., 2] * 1868 + 8192) >> 14
    order_ok = all(np.abs(luma(a[i]) - luma(b[i])).max() <= 1 for i in range(n)) and all(
        np.abs(luma(a[i]) - luma(b[(i + 1) % n])).mean() > 1 for i in range(n))
    return {"frames": n, "seconds": round(dt, 4), "value": round(n / dt, 2), "unit": "frames/s", "frame_order_verified": bool(order_ok),
            "how": "rank 0 holds the clip: RCCL scatter (frame i -> rank i mod N) -> havc_colorize_clip on every rank's GPU -> RCCL gather"}


def extras(args, cc, ctx, frames, d_src, d_dst, fbytes, n_batches, sds):
    """the honest-measurement legs: sustained load, PCIe-inclusive, batch-1 per-call"""
    res = {}
    batch = args.batch

    def step(i):
        f0 = (i % n_batches) * batch
        cc.colorize_device(off(d_src, f0 * fbytes), off(d_dst, f0 * fbytes), batch, WIDTH, HEIGHT)
    # ---- sustained: the chip sits at its package power limit; a 1 s window can flatter it ----
    if args.sustain_seconds > 0:
        ctx.synchronize()
        t0 = time.perf_counter()
        marks, i = [], 0
        while True:
            step(i)
            i += 1
            if i % 4 == 0:
                ctx.synchronize()
                now = time.perf_counter() - t0
                marks.append((now, i * batch))
                if now >= args.sustain_seconds:
                    break
        first = next((m for m in marks if m[0] >= 1.0), marks[-1])
        half = next((m for m in marks if m[0] >= marks[-1][0] / 2), marks[-1])
        res["sustained"] = {"seconds": round(marks[-1][0], 2), "frames": marks[-1][1], "value": round(marks[-1][1] / marks[-1][0], 2),
                            "first_second_value": round(first[1] / first[0], 2),
                            "second_half_value": round((marks[-1][1] - half[1]) / max(marks[-1][0] - half[0], 1e-9), 2), "unit": "frames/s"}
    # ---- PCIe-inclusive: pinned host frames in -> pinned host frames out, pipelined ----
    n = len(frames)
    hin, hout = ctx.host_alloc(frames.nbytes), ctx.host_alloc(frames.nbytes)
    try:
        src, dst = hin.reshape(frames.shape), hout.reshape(frames.shape)
        src[...] = frames
        cc.colorize_host(src, out=dst)                      # warm (allocates the staging buffers / streams)
        reps = max(2, int(np.ceil(4 * 64 / n)))
        t0 = time.perf_counter()
        for _ in range(reps):
            cc.colorize_host(src, out=dst)
        dt = time.perf_counter() - t0
        res["pcie_inclusive"] = {"value": round(reps * n / dt, 2), "unit": "frames/s", "frames": reps * n,
                                 "how": "havc_colorize_clip_host: pinned host memory, H2D / passes / D2H of consecutive batches on three streams"}
    finally:
        ctx.host_free(hin)
        ctx.host_free(hout)
    # ---- batch 1, one blocking call per frame: what a ModifyFrame selector gets ----
    from PIL import Image
    from vsdeoldify_amd.render import ModelImageRender
    r1 = ModelImageRender(None, "stable", RENDER_FACTOR, 0.5, device_index=ctx.device_id, state_dicts=sds, max_batch=1)
    S = RENDER_FACTOR * 16
    img = Image.fromarray(np.ascontiguousarray(frames[0][:S, :S]))
    for _ in range(3):
        r1.get_transformed_image(img)
    k = 40
    t0 = time.perf_counter()
    for _ in range(k):
        r1.get_transformed_image(img)
    dt = time.perf_counter() - t0
    res["batch1"] = {"value": round(k / dt, 2), "unit": "frames/s", "ms_per_call": round(dt / k * 1e3, 3),
                     "how": "ModelImageRender('stable', rf=35).get_transformed_image(PIL 560x560), one blocking call per frame (H2D + 2 passes + D2H)"}
    # ---- the same per-frame call from 16 threads (VapourSynth's worker pool) through ONE coalescing render: havc_batcher ----
    import threading
    T, K = 16, 6
    rc = ModelImageRender(None, "stable", RENDER_FACTOR, 0.5, device_index=ctx.device_id, state_dicts=sds, coalesce=T)
    imgs = [Image.fromarray(np.ascontiguousarray(frames[i % len(frames)][:S, :S])) for i in range(T)]

    def worker(t, n):
        for _ in range(n):
            rc.get_transformed_image(imgs[t])
    for n in (1, K):                                           # warm-up round, then the timed one
        ts = [threading.Thread(target=worker, args=(t, n)) for t in range(T)]
        t0 = time.perf_counter()
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        dt = time.perf_counter() - t0
    calls, batches = rc._batcher(S, True).stats()
    res["per_frame_calls_16_threads"] = {"value": round(T * K / dt, 2), "unit": "frames/s", "calls": calls, "batches": batches,
                                         "how": "16 Python threads, each one blocking get_transformed_image(PIL 560x560) per frame, one "
                                                "ModelImageRender(coalesce=16): concurrent calls merged into batches by havc_batcher"}
    return res


def bench_c4(args, rank, local_rank, world, dist, ddcolor_only=False):
    """ddcolor_only (--config c3): BASELINE configs[2], DDColor large at input 512 on a 1080p clip (HAVC_colorizer method=1,
    ddcolor_p=[1,32,..] => input_size = trunc(32/2)*32 = 512, vsmodels.py:302): Spline64 squash -> DDColor -> Spline64 back + luma.
    Otherwise BASELINE configs[3]: HAVC DeOldify + DDColor merge (combine_method=2) on a 1080p clip, defaults of HAVC_colorizer:
    deoldify_p=[0,24,..] (video model, 384x384), ddcolor_p=[1,24,..] (artistic, input 384), mweight=0.4 -- device-resident graph:
    Spline64 squash -> DynamicUnetWide pass + DDColor pass -> Image.blend -> Spline64 back + luma of the source.
    DDColor's parity is UNPINNED (external wheel; oracle/ddcolor.py restates the published architecture)."""
    import torch
    from vsdeoldify_amd.clip import synthetic_gray_frame
    from vsdeoldify_amd.device import DeviceImage
    from vsdeoldify_amd.havc import HAVCFrameColorizer
    from vsdeoldify_amd.synth import synth_ddcolor_state_dict, synth_state_dict
    if ddcolor_only:
        col = HAVCFrameColorizer(method=1, ddcolor_p=(1, 32, 1.0, 0.0, True), device_index=local_rank,
                                 ddcolor_state_dict=synth_ddcolor_state_dict(1), max_batch=args.batch)
    else:
        col = HAVCFrameColorizer(method=2, mweight=0.4, deoldify_p=(0, 24, 1.0, 0.0), ddcolor_p=(1, 24, 1.0, 0.0, True), device_index=local_rank,
                                 state_dicts={"video": synth_state_dict("wide", 1)}, ddcolor_state_dict=synth_ddcolor_state_dict(1),
                                 max_batch=args.batch)
    ctx = col.ctx
    frames = np.stack([synthetic_gray_frame(rank * args.batch + i, WIDTH, HEIGHT) for i in range(args.batch)])
    clip = DeviceImage.from_numpy(ctx, frames)
    keep = []

    def step(i):
        keep[:] = [col.colorize_clip(clip)]

    def sync_all():
        ctx.synchronize()
        if dist is not None:
            dist.barrier()
        if torch.cuda.is_available():
            torch.cuda.synchronize()
    for i in range(max(args.warmup, 1)):
        step(i)
    ctx.reset_stats()
    sync_all()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    sync_all()
    elapsed = time.perf_counter() - t0
    st = ctx.stats()
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=f"cuda:{local_rank}")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    total = args.steps * args.batch * world
    gflop_frame = st.total_flops / max(st.frames / (1 if ddcolor_only else 2), 1) / 1e9 if st.frames else 0.0        # every model counts `frames`
    out = {"metric": "colorized frames/sec/GPU @1080p (DDColor large, input 512)" if ddcolor_only else
                     "colorized frames/sec/GPU @1080p (HAVC DeOldify+DDColor merge, combine_method=2)", "value": round(total / elapsed, 3),
           "unit": "frames/s (sum over n_gpus)", "n_gpus": world, "value_per_gpu": round(total / elapsed / world, 3), "steps": args.steps,
           "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
           "dtype": "f16", "data": "synthetic",
           "config": {"workload": "DDColor modelsize=large, 512 input, 1080p clip (BASELINE.json configs[2])" if ddcolor_only else
                                  "HAVC DeOldify+DDColor merge (combine_method=2) 1080p, frame-sharded (BASELINE.json configs[3])",
                      "frames_per_step_per_gpu": args.batch, "deoldify": None if ddcolor_only else "video, rf=24 (384x384)",
                      "ddcolor": "artistic, input %d (parity UNPINNED)" % (512 if ddcolor_only else 384),
                      "mweight": None if ddcolor_only else 0.4, "algorithmic_gflop_per_frame": round(gflop_frame, 2), "device_resident": True,
                      "parallelism": f"frame-sharded x{world}, weight replica per GPU, no collective"},
           "whole_path_tflops": round(total * gflop_frame * 1e9 / elapsed / 1e12 / world, 2)}
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out))


if __name__ == "__main__":
    main()
