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

`value` is the fast mode (precision="fast", asked for explicitly; the package default is "precise" = the line's `contract` object).  It is the whole-job rate over the EXACTLY K timed steps (sum over all ranks; `value_per_gpu` = value / n_gpus); one step = 64 frames per GPU
(--batch).  Beside it the line carries, measured in the same process after the timed region (rank 0, N = 1):
  precise        the same step with ModelImageRender(precision="precise") (fp32-class arithmetic; measured against the oracle: mean ~1e-3, p99 0.000,
                 >= 99.97 % of the pixels below CIEDE2000 1.0, residual maximum 2.5 - 3.6 = single truncation flips of uint8(x * 255), which two fp32
                 evaluations of the reference differ by as well), its own roofline and its parity against the same oracle frames.  Every parity object
                 carries `meets_contract` (p99 < 1.0 and >= 99 % of the pixels below 1.0) and `every_pixel_below_1` (the literal reading: false
                 for any implementation that is not bit-identical to the reference's summation order)
  other_configs  BASELINE configs[2..4] (c3 / c4 / c5) as short child-process legs run BEFORE this process touches the GPU (--no-other-configs skips)
  sustained      the same step looped for >= --sustain-seconds (default 30 s): first-second and steady-state rates
  pcie_inclusive host frames in -> host frames out through havc_colorize_clip_host (pinned memory, uploads / passes /
                 downloads of consecutive batches overlapped on three streams)
  batch1         ModelImageRender.get_transformed_image on one 560x560 PIL image per call: the rate a VapourSynth
                 ModifyFrame selector sees (vsslib/vsmodels.py:219-230) -- fast mode; `batch1_default_precise` = the same call on a
                 render built without any precision switch (the package default)
  cpu_baseline / parity   the CPU oracle on one frame of the same clip, and the GPU frame against it

Wall time of the default run (`python bench.py`, one MI355X box, round 5): about 4 minutes -- the five child legs (c3, c3 precise, c4, c4 precise, c5) 85 s, the
headline's warm-up + timed steps 10 s, the sustained / PCIe / per-frame-call legs 55 s, the precise leg 15 s, the CPU oracle on 8 frames 45 s.  --no-other-configs,
--no-extras, --no-precise and --no-cpu-baseline each drop their part; the JSON line says which legs ran.

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
PMC_FILES = [os.path.join("profiles", "r6_tail_conv_pmc.json"), os.path.join("profiles", "r5_tail_conv_pmc.json"), os.path.join("profiles", "r4_tail_conv_pmc.json"), os.path.join("profiles", "r3_tail_conv_pmc.json"), os.path.join("profiles", "r2_tail_conv_pmc.json")]   # newest first


PEAK_F32_MATRIX_TFLOPS = 157.3      # MI355X fp32 matrix (xf32-free v_mfma_f32_*_f32) peak: the fair context for an fp32-class figure (MI355X_MICROARCH.md)
PRECISE_MIN_SECONDS = 5.0           # the precise (contract-meeting) leg is timed over at least this long
CHILD_MIN_SECONDS = 2.0             # ... and every c3 / c4 / c5 child leg over at least this long


PARITY_SEEDS = ((1, 2), (11, 12), (21, 22))     # (video, stable) weight seeds: the bench weights first
PARITY_FRAMES = (4, 2, 2)                       # frames of the clip checked per seed pair: 8 frames over 3 weight sets


def _stats(de, d):
    """`meets_contract`: north_star's "within CIEDE2000 < 1.0 of the reference" read as p99 < 1.0 AND >= 99 % of the pixels below 1.0 (the thresholds of
    tests/test_gpu_precise.py); `every_pixel_below_1`: the literal per-pixel reading (ciede2000_max < 1.0)."""
    p99, frac, mx = float(np.percentile(de, 99)), float((de < 1.0).mean()), float(de.max())
    return {"ciede2000_mean": round(float(de.mean()), 4), "ciede2000_p99": round(p99, 4),
            "ciede2000_max": round(mx, 4), "pixels_with_dE_below_1": round(frac, 5),
            "meets_contract": bool(p99 < 1.0 and frac >= 0.99), "every_pixel_below_1": bool(mx < 1.0),
            "bytes_within_1lsb": round(float((d <= 1).mean()), 5), "bytes_within_2lsb": round(float((d <= 2).mean()), 5)}


def cpu_baseline_and_parity(cc_main, frames, threads, device_index, cc_precise=None):
    """Time the CPU oracle (fp32 PyTorch restatement + numpy tail) on frames of the same workload and use its outputs as the parity
    reference for the GPU results of those frames: 8 frames of the clip over 3 seeded weight sets (the bench weights + two more),
    per-frame statistics pooled, the worst frame named.  With cc_precise (the bench weights in precise mode) the SAME reference frames
    also judge the precise path (a second parity object)."""
    import gc
    import torch
    from oracle import imaging, pipeline
    from vsdeoldify_amd.clip import ClipColorizer
    from vsdeoldify_amd.synth import synth_state_dict
    torch.set_num_threads(threads)
    modes = ["fast"] + (["precise"] if cc_precise is not None else [])
    per, des, ds = {m: [] for m in modes}, {m: [] for m in modes}, {m: [] for m in modes}
    cpu_s, cpu_n = 0.0, 0
    for (sv, ss), nfr in zip(PARITY_SEEDS, PARITY_FRAMES):
        sds = {"video": synth_state_dict("wide", sv), "stable": synth_state_dict("wide", ss)}
        idx = [(i * 7) % len(frames) for i in range(nfr)]
        got = {}
        for m in modes:
            if m == "precise" and (sv, ss) != PARITY_SEEDS[0]:
                continue            # (packing a precise generator pair takes a minute of host time: the bench checks the precise path on its own
                                    #  weights; all three weight sets are the -m gpu test's job, tests/test_gpu_precise.py)
            main = cc_main if m == "fast" else cc_precise
            cc = main if ((sv, ss) == PARITY_SEEDS[0] and main is not None) else ClipColorizer("stable", RENDER_FACTOR, 0.5, device_index=device_index,
                                                                                               state_dicts=sds, max_batch=2, precision=m)
            got[m] = np.concatenate([cc.colorize(frames[i:i + 1]) for i in idx])
            if cc is not main:
                for r in (cc.render._video, cc.render._second):
                    r.close()
                del cc
                gc.collect()
        for k, i in enumerate(idx):
            t0 = time.time()
            ref = pipeline.colorize_frame_fullsize(sds, "stable", frames[i], RENDER_FACTOR, 0.5)
            cpu_s += time.time() - t0
            cpu_n += 1
            for m in got:
                de = imaging.delta_e00_images(got[m][k], ref)
                d = np.abs(got[m][k].astype(np.int32) - ref.astype(np.int32))
                des[m].append(de.reshape(-1)); ds[m].append(d.reshape(-1))
                per[m].append({"weights_seed": [sv, ss], "frame": int(i), "mean": round(float(de.mean()), 4), "p99": round(float(np.percentile(de, 99)), 4),
                               "max": round(float(de.max()), 3), "pixels_with_dE_below_1": round(float((de < 1.0).mean()), 5)})
    out = {}
    for m in modes:
        parity = _stats(np.concatenate(des[m]), np.concatenate(ds[m]))
        worst = max(per[m], key=lambda r: r["p99"])
        parity.update({"frames_checked": len(per[m]), "weight_sets": len({tuple(r_["weights_seed"]) for r_ in per[m]}), "worst_frame": worst, "per_frame": per[m], "against": "oracle (CPU fp32 port)",
                       "note": "fp16 MFMA operands; floor / decomposition in profiles/r2_precision_study.txt" if m == "fast" else
                               "hi / lo fp16 pairs, three-segment convs, fp32 epilogues and attention (HAVC_F_PRECISE): what is left is the fp32 summation-order floor"})
        out[m] = parity
    base = {"value": round(cpu_n / cpu_s, 5), "unit": "frames/s", "cores": threads, "host_cpus": os.cpu_count(), "threads_probe": _CPU_THREADS.get("probe"), "kind": "port",
            "sample": f"{cpu_n} frames of the 1080p clip (2 U-Net passes at 560x560 fp32 + Spline64/YUV tail each), {cpu_s:.1f} s"}
    return base, out["fast"], out.get("precise")


def precise_leg(args, sds, device_index, frames, fbytes, batch=None):
    """The fp32-class mode: the same step on the same clip with ModelImageRender(precision="precise") -- hi / lo fp16
    activation pairs, three K segments per conv on the same MFMA kernels (3x the matrix work), fp32 epilogues and attention.  What it measures
    against the oracle is in the leg's `parity` object (p99 0.000, >= 99.97 % of the pixels below 1.0, `meets_contract`; the residual maximum of
    2.5 - 3.6 is a uint8 truncation flip on isolated pixels).  `batch` frames per step (activations are twice as large: 32 frames = 141 GB of
    arenas, which fit once the fast nets' 141 GB are released; 16 otherwise).  Returns (leg dict, ClipColorizer)."""
    from vsdeoldify_amd import _native as nat
    from vsdeoldify_amd.clip import ClipColorizer
    batch = min(batch or 16, args.batch)
    cc = ClipColorizer("stable", RENDER_FACTOR, 0.5, device_index=device_index, state_dicts=sds, max_batch=batch, precision="precise")
    ctx = cc.ctx
    n = len(frames) // batch * batch
    d_src, d_dst = ctx.dev_alloc(n * fbytes), ctx.dev_alloc(n * fbytes)
    try:
        ctx.dev_upload(d_src, frames[:n])

        def step(i):
            f0 = (i % (n // batch)) * batch
            cc.colorize_device(off(d_src, f0 * fbytes), off(d_dst, f0 * fbytes), batch, WIDTH, HEIGHT)
        step(0)
        ctx.synchronize()
        t0 = time.perf_counter()
        step(1)                                                # one more untimed step: its duration sizes the timed region
        ctx.synchronize()
        est = max(time.perf_counter() - t0, 1e-3)
        ctx.reset_stats()
        nat.check(ctx.lib.havc_tag_timing_enable(ctx.h, TAG_TAIL_RES, 1), ctx.h)
        # the contract-meeting figure is timed like the headline: --steps steps, and never less than PRECISE_MIN_SECONDS of work
        steps = max(2, args.steps, int(np.ceil(PRECISE_MIN_SECONDS / est)))
        ctx.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            step(1 + i)
        ctx.synchronize()
        dt = time.perf_counter() - t0
        avg_ms, launches = ctypes.c_double(), ctypes.c_int64()
        nat.check(ctx.lib.havc_tag_timing_read(ctx.h, ctypes.byref(avg_ms), ctypes.byref(launches)), ctx.h)
        nat.check(ctx.lib.havc_tag_timing_enable(ctx.h, TAG_TAIL_RES, 0), ctx.h)
    except Exception:
        for r_ in (cc.render._video, cc.render._second):      # a failed attempt (out of memory at 32 frames per step) must not keep its arenas
            try:
                r_.close()
            except Exception:                                  # noqa: BLE001
                pass
        raise
    finally:
        ctx.dev_free(d_src)
        ctx.dev_free(d_dst)
    S, c = RENDER_FACTOR * 16, 259
    fpl = steps * batch * 4 / max(int(launches.value), 1)
    alg = 2.0 * fpl * S * S * c * c * 9
    ach = alg / (avg_ms.value * 1e-3) / 1e12 if avg_ms.value > 0 else 0.0
    fps = steps * batch / dt
    leg = {"value": round(fps, 2), "unit": "frames/s", "frames_per_step": batch, "steps": steps, "ms_per_step": round(dt / steps * 1e3, 2), "dtype": "f16x2 (hi / lo pairs, fp32 accumulate)",
           "whole_path_tflops": round(fps * 2759.32e9 / 1e12, 2), "whole_path_frac": round(fps * 2759.32e9 / 1e12 / PEAK_F16_TFLOPS, 4),
           "seconds_timed": round(dt, 2),
           "roofline": {"bound": "mfma", "achieved": round(ach, 2), "peak": PEAK_F16_TFLOPS, "unit": "TFLOP/s", "frac": round(ach / PEAK_F16_TFLOPS, 4),
                        "peak_fp32_matrix": PEAK_F32_MATRIX_TFLOPS, "frac_of_fp32_matrix_peak": round(ach / PEAK_F32_MATRIX_TFLOPS, 3),
                        "mfma_executed_frac": round(3 * ach / PEAK_F16_TFLOPS, 4), "launches_timed": int(launches.value), "frames_per_launch": round(fpl, 2),
                        "avg_launch_ms": round(avg_ms.value, 4), "flops_per_launch": alg,
                        "kernel": "conv_pipe_kernel<2,4,8,1> on the three-segment K walk (layers.10 res-block 3x3 259->259 @560x560): `achieved` counts the "
                                  "ALGORITHMIC flops of the conv, the kernel executes 3x that on the MFMA pipe (mfma_executed_frac)"},
           "how": "havc_colorize_clip with nets built by ModelImageRender(precision='precise') / HAVC_PRECISION=precise"}
    return leg, cc


def off(p, nbytes):
    return ctypes.c_void_p(p.value + nbytes)


_CPU_THREADS = {}


def cpu_threads(args):
    """Threads of the CPU-oracle legs.  --cpu-threads N pins them; 0 (default) = what is FASTEST on this host among 32, 64, half and all of
    os.cpu_count() (SURVEY.md section 8d asks for all host cores; on a many-socket box torch's conv with every hardware thread can be slower than
    with 32 -- on the round-5 box, 256 hardware threads: 0.60 s against 0.09 s with 64 for one tail-sized conv -- and the baseline must not be handicapped
    by that, so a tail-sized conv plus 40 encoder-sized convs are timed once per candidate, a few seconds in all).
    The choice and os.cpu_count() are both printed in the cpu_baseline object (`cores`, `host_cpus`, `threads_probe`)."""
    n = os.cpu_count() or 1
    if args.cpu_threads and args.cpu_threads > 0:
        return min(n, args.cpu_threads)
    if "best" not in _CPU_THREADS:
        import torch
        import torch.nn.functional as F
        x, w = torch.randn(1, 264, 280, 280), torch.randn(264, 264, 3, 3)
        xs, ws = torch.randn(1, 256, 35, 35), torch.randn(256, 256, 3, 3)          # the encoder's small layers weigh as much as the tail in the oracle's wall time
        probe = {}
        for th in sorted({min(n, 16), min(n, 32), min(n, 64), max(1, n // 2), n}):
            torch.set_num_threads(th)
            F.conv2d(x, w, None, 1, 1)
            F.conv2d(xs, ws, None, 1, 1)
            t0 = time.time()
            F.conv2d(x, w, None, 1, 1)
            for _ in range(40):
                F.conv2d(xs, ws, None, 1, 1)
            probe[th] = round(time.time() - t0, 4)
        _CPU_THREADS["best"] = min(probe, key=probe.get)
        _CPU_THREADS["probe"] = probe
        print(f"bench: CPU thread probe (seconds for a 259-channel 3x3 conv at 280x280 + 40 256-channel convs at 35x35) {probe} -> {_CPU_THREADS['best']} threads of {n}", file=sys.stderr, flush=True)
    return _CPU_THREADS["best"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=None, help="frames per step per GPU (default: 64 for the headline config and c4, 128 for c3, 32 for c5)")
    ap.add_argument("--clip-frames", type=int, default=None, help="distinct synthetic frames resident per GPU (default: one step's worth)")
    ap.add_argument("--config", default="c2", choices=["c2", "c3", "c4", "c5"],
                    help="c2 = BASELINE configs[1] (headline), c3 = configs[2] (DDColor large, input 512), c4 = configs[3] (DeOldify+DDColor merge), "
                         "c5 = configs[4] (ColorMNet exemplar path, 1 reference frame)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the sustained / PCIe-inclusive / batch-1 legs")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the short c3 / c4 / c5 legs of the default run")
    ap.add_argument("--no-precise", action="store_true", help="skip the precise-mode leg (ModelImageRender(precision='precise'))")
    ap.add_argument("--precision", default="fast", choices=["fast", "precise"],
                    help="c3 / c4: build every model of the graph in this mode (HAVCFrameColorizer(precision=...)); the headline config has its own precise leg")
    ap.add_argument("--sustain-seconds", type=float, default=30.0)
    ap.add_argument("--min-seconds", type=float, default=0.0,
                    help="c3 / c4 / c5: raise --steps (BEFORE the timed region, from one untimed probe step) so that the timed region lasts at least this long")
    ap.add_argument("--cpu-threads", type=int, default=0, help="threads for the CPU-oracle baseline leg (0 = all host cores: os.cpu_count())")
    args = ap.parse_args()
    if args.batch is None:              # 64 frames per step fill the small encoder / decoder layers better than 32 (+2 %, same-box A/B; 105 GB of activations)
        # c3 (DDColor): 64 frames = 65 536 tokens at the 768-channel stage -> 768 tiles of 256 x 256 for pwconv2 = exactly 3 per CU; at 32 frames
        # 384 tiles = 1.5 per CU and the GEMM runs at 75 % (profiles/r4_ddcolor_batch_sweep.txt: 1.09 -> 1.01 ms per frame of GPU ops)
        # round 5, same-box A/B (gpurun_out/r5h): c3 at 128 frames per step 1 266 vs 1 230 frames/s at 64 (+2.9 %), c4 at 64: 879 vs 860 at 32 (+2.2 %),
        # the headline at 96: 359.3 vs 357.3 at 64 (+0.6 %: left at 64); HAVC_TWO_STREAMS=0 at 64 frames: 348.0 (the two streams still buy 2.6 %)
        args.batch = {"c2": 64, "c3": 128, "c4": 64}.get(args.config, 32)
    if args.clip_frames is None:
        args.clip_frames = args.batch

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

    # ---- the other BASELINE configs as short legs of the DEFAULT run (the driver only runs `bench.py --gpus 1`) ----
    # Fresh child processes, one after the other, BEFORE this process touches the GPU; each prints its own JSON line (the same code as
    # `bench.py --config cX`), of which the headline fields are embedded under "other_configs".  --no-extras / --no-other-configs skip them.
    other = None
    if args.config == "c2" and world == 1 and rank == 0 and not args.no_extras and not args.no_other_configs:
        other = other_configs_leg(args)

    import torch
    dist = None
    if world > 1:
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))   # nccl == RCCL on ROCm

    if args.config in ("c3", "c4"):
        return bench_c4(args, rank, local_rank, world, dist, ddcolor_only=args.config == "c3")
    if args.config == "c5":
        return bench_c5(args, rank, local_rank, world, dist)

    from vsdeoldify_amd import _native as nat
    from vsdeoldify_amd.clip import ClipColorizer, synthetic_gray_frame
    from vsdeoldify_amd.synth import synth_state_dict

    sds = {"video": synth_state_dict("wide", 1), "stable": synth_state_dict("wide", 2)}
    # the headline is the opt-in speed mode, asked for explicitly; the package default is "precise" (vsdeoldify_amd/precision.py) = the `contract` leg
    cc = ClipColorizer("stable", RENDER_FACTOR, 0.5, device_index=local_rank, state_dicts=sds, max_batch=args.batch, precision="fast")
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

    multi = None
    if dist is not None:
        mine = torch.tensor([elapsed], dtype=torch.float64, device=f"cuda:{local_rank}")
        every = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)                               # RCCL: one element per rank, proves every rank took part
        t = mine.clone()
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        multi = {"rccl_ranks": int(dist.get_world_size()), "backend": dist.get_backend(), "cuda_device_count": torch.cuda.device_count(),
                 "per_rank_frames_per_s": [round(args.steps * args.batch / float(e.item()), 2) for e in every]}

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
    for pmc_file in PMC_FILES:
        try:
            pmc = json.load(open(os.path.join(ROOT, pmc_file)))
            if pmc.get("frames_per_launch") == round(frames_per_launch):
                traffic = pmc["traffic_bytes_per_launch"]
                traffic_source = f"{pmc_file}: separate rocprofv3 --pmc passes of this command (tools/pmc_bench.sh), not this run"
                break
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
                   "precision": "fast (fp16 MFMA operands, fp32 accumulate): the OPT-IN speed mode (precision='fast' / HAVC_PRECISION=fast) -- the reference is fp32 end "
                                "to end (deoldify/filters.py:45-68) and this mode meets CIEDE2000 < 1.0 in the mean only; the package DEFAULT is 'precise' "
                                "(vsdeoldify_amd/precision.py), which meets the per-pixel contract: its figure is `contract`",
                   "conv_tile_autotune": os.environ.get("HAVC_AUTOTUNE", "1") != "0",
                   "parallelism": f"frame-sharded x{world}, weight replica per GPU, no collective"},
        "whole_path_tflops": round(total_frames * 2759.32e9 / elapsed / 1e12 / world, 2),
        "roofline": {"bound": "mfma", "achieved": round(achieved, 2), "peak": PEAK_F16_TFLOPS, "unit": "TFLOP/s",
                     "frac": round(achieved / PEAK_F16_TFLOPS, 4), "traffic": traffic, "traffic_source": traffic_source,
                     "kernel": "conv_pipe_kernel<2,4,8,1,0,EF> (layers.10 res-block 3x3 259->259 @560x560, 2 launches/pass: EF=1 conv + ReLU, EF=261 conv + ReLU + residual with layers.11/12 in its epilogue; rocprof lists the two instantiations separately, this is their mean)",
                     "launches_timed": int(launches.value), "frames_per_launch": round(frames_per_launch, 2), "avg_launch_ms": round(avg_ms.value, 4),
                     "flops_per_launch": conv_flops},
        "gpu_ms_per_frame": round(st.total_ms / max(st.frames, 1), 4),
    }
    if multi is not None:
        out["multi_gpu"] = multi
    if rank == 0 and world == 1 and not args.no_extras:
        _progress("extras (sustained / PCIe-inclusive / per-frame calls)")
        out.update(extras(args, cc, ctx, frames, d_src, d_dst, fbytes, n_batches, sds))
    cc_precise = None
    if rank == 0 and world == 1 and not args.no_precise:
        # the fast nets (64 frames x 2 generators = 141 GB of activation arenas) are done: release them so that the precise nets can run 32 frames
        # per step as well (the parity leg below builds small fast nets of its own)
        try:
            for r_ in (cc.render._video, cc.render._second):
                r_.close()
            cc = None
        except Exception:                               # noqa: BLE001
            pass
        for pb in ((48, 32, 16) if os.environ.get('HAVC_BENCH_PRECISE_48') else (32, 16)):
            try:
                _progress(f"precise leg ({pb} frames per step)")
                out["precise"], cc_precise = precise_leg(args, sds, local_rank, frames, fbytes, batch=pb)
                break
            except Exception as e:                      # never lose the headline line to the second mode (32 frames may not fit next to other arenas)
                out["precise"] = {"error": f"{type(e).__name__}: {e}"}
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        _progress("CPU oracle + parity")
        base, parity, parity_p = cpu_baseline_and_parity(cc, frames, cpu_threads(args), local_rank, cc_precise)
        out["cpu_baseline"] = base
        out["parity"] = parity
        if parity_p is not None:
            out["precise"]["parity"] = parity_p
    # ---- which number meets north_star's tolerance: a reader of the top-level keys alone sees both ----
    if isinstance(out.get("precise"), dict) and "value" in out["precise"]:
        pp, fp_ = out["precise"].get("parity") or {}, out.get("parity") or {}
        out["contract"] = {"mode": "precise (the package default)", "value": out["precise"]["value"], "unit": "frames/s", "dtype": "f16x2 -> fp32-class (hi / lo fp16 pairs, 22 significand bits, fp32 accumulate)",
                           "meets_contract": pp.get("meets_contract"), "ciede2000_p99": pp.get("ciede2000_p99"), "pixels_with_dE_below_1": pp.get("pixels_with_dE_below_1"),
                           "steps": out["precise"]["steps"], "frames_per_step": out["precise"]["frames_per_step"], "seconds_timed": out["precise"].get("seconds_timed"),
                           "headline_value_meets_contract": fp_.get("meets_contract"),
                           "reading": "`value` (top level) is the fast mode at dtype f16: mean CIEDE2000 < 1.0 but p99 > 1.0 -- OUTSIDE the per-pixel tolerance; "
                                      "`contract.value` is the same step in precise mode, inside it (p99 < 1.0 and >= 99 % of the pixels below 1.0), timed over the same "
                                      "--steps (>= 5 s).  Details: `precise` (roofline, parity per frame)."}
    if other is not None:
        out["other_configs"] = other
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


def _raise_steps(args, step, sync_all, world):
    """--min-seconds: one more UNTIMED step sizes the timed region; K is fixed before it starts and reported in `steps` (single-process legs only: every
    rank of a multi-GPU run must time the same K)"""
    if args.min_seconds <= 0 or world > 1:
        return
    sync_all()
    t0 = time.perf_counter()
    step(0)
    sync_all()
    est = max(time.perf_counter() - t0, 1e-4)
    args.steps = max(args.steps, int(np.ceil(args.min_seconds / est)))


def _progress(msg):
    print(f"bench: [{time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def other_configs_leg(args):
    """c3 / c4 / c5 (BASELINE configs[2..4]) in child processes: {value, ms_per_step, whole_path_tflops, roofline frac, parity} per config"""
    import subprocess
    res = {}

    def child(cfg, extra, steps, warm):
        _progress(f"child leg {cfg} {' '.join(extra)}")
        cmd = [sys.executable, os.path.abspath(__file__), "--config", cfg, "--steps", steps, "--warmup", warm, "--no-extras", "--cpu-threads", str(args.cpu_threads),
               "--min-seconds", str(CHILD_MIN_SECONDS)] + extra
        if args.no_cpu_baseline:
            cmd.append("--no-cpu-baseline")
        t0 = time.time()
        try:
            p = subprocess.run(cmd, capture_output=True, text=True, timeout=420, cwd=ROOT)
            lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
            if p.returncode != 0 or not lines:
                raise RuntimeError(f"rc {p.returncode}: {(p.stderr or p.stdout)[-300:]}")
            o = json.loads(lines[-1])
            par = o.get("parity") or {}
            return {"metric": o["metric"], "value": o["value"], "unit": "frames/s", "ms_per_step": o["ms_per_step"], "dtype": o.get("dtype"),
                    "frames_per_step": o["config"].get("frames_per_step_per_gpu"), "steps": o["steps"], "seconds_timed": round(o["steps"] * o["ms_per_step"] / 1e3, 2),
                    "whole_path_tflops": o.get("whole_path_tflops"), "whole_path_frac": round((o.get("whole_path_tflops") or 0.0) / PEAK_F16_TFLOPS, 4),
                    "roofline": {k: o["roofline"].get(k) for k in ("bound", "scope", "achieved", "peak", "unit", "frac", "traffic", "avg_launch_ms", "frames_per_launch") if k in o["roofline"]},
                    "parity": {k: par.get(k) for k in ("ciede2000_mean", "ciede2000_p99", "ciede2000_max", "pixels_with_dE_below_1", "meets_contract", "frames_checked")} if par else None,
                    "cpu_baseline": (o.get("cpu_baseline") or {}).get("value"), "workload": o["config"]["workload"],
                    "leg_seconds": round(time.time() - t0, 1)}
        except Exception as e:                                  # noqa: BLE001 -- a leg never costs the headline line
            return {"error": f"{type(e).__name__}: {e}", "leg_seconds": round(time.time() - t0, 1)}
    for cfg in ("c3", "c4", "c5"):
        # (c5's first windows carry the exemplar, the growth of the working memory and the first consolidation: its steady state needs a longer warm-up)
        steps, warm = ("8", "4") if cfg == "c5" else ("3", "1")
        res[cfg] = child(cfg, [], steps, warm)
        if cfg in ("c3", "c4") and not args.no_precise:
            # the same graph with every model in precise mode (HAVCFrameColorizer(precision="precise"), round 5): 16 frames per step (pair activations)
            res[cfg]["precise"] = child(cfg, ["--precision", "precise", "--batch", "16"], "2", "1")
        # which figure of this config meets the tolerance (c5's fast path already does)
        src = res[cfg].get("precise") if isinstance(res[cfg].get("precise"), dict) and "value" in res[cfg].get("precise", {}) else res[cfg]
        if "value" in src:
            res[cfg]["contract"] = {"mode": "precise" if src is not res[cfg] else "fast", "value": src["value"], "meets_contract": (src.get("parity") or {}).get("meets_contract"),
                                    "seconds_timed": src.get("seconds_timed")}
    return res


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
    r1 = ModelImageRender(None, "stable", RENDER_FACTOR, 0.5, device_index=ctx.device_id, state_dicts=sds, max_batch=1, precision="fast")
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
    # ---- the same single caller with the low-latency nets (split-K convs for one frame per launch; fp32 summation order differs) ----
    r2 = ModelImageRender(None, "stable", RENDER_FACTOR, 0.5, device_index=ctx.device_id, state_dicts=sds, max_batch=1, low_latency=True, precision="fast")
    for _ in range(3):
        r2.get_transformed_image(img)
    t0 = time.perf_counter()
    for _ in range(k):
        r2.get_transformed_image(img)
    dt = time.perf_counter() - t0
    res["batch1_low_latency"] = {"value": round(k / dt, 2), "unit": "frames/s", "ms_per_call": round(dt / k * 1e3, 3),
                                 "how": "the same call on ModelImageRender(..., low_latency=True) / HAVC_LOW_LATENCY=1: nets for one frame per launch with split-K convs "
                                        "(opt-in: 3 LSB from the batched nets on isolated bytes at this size, tests/test_gpu_deoldify.py)"}
    # ---- the same single caller in the package's DEFAULT mode (precision.py: "precise"): what a drop-in user who sets no switch gets per blocking call ----
    try:
        for r_ in (r1._video, r1._second, r2._video, r2._second):       # (the fast one-frame nets are done: their arenas go first)
            r_.close()
        r3 = ModelImageRender(None, "stable", RENDER_FACTOR, 0.5, device_index=ctx.device_id, state_dicts=sds, max_batch=1)
        for _ in range(3):
            r3.get_transformed_image(img)
        k3 = 20
        t0 = time.perf_counter()
        for _ in range(k3):
            r3.get_transformed_image(img)
        dt = time.perf_counter() - t0
        res["batch1_default_precise"] = {"value": round(k3 / dt, 2), "unit": "frames/s", "ms_per_call": round(dt / k3 * 1e3, 3), "precision": r3._precision,
                                         "how": "ModelImageRender('stable', rf=35) built WITHOUT a precision switch (the package default), one blocking "
                                                "get_transformed_image(PIL 560x560) per frame"}
        for r_ in (r3._video, r3._second):
            r_.close()
    except Exception as e:                                          # noqa: BLE001 -- an extra leg never costs the line
        res["batch1_default_precise"] = {"error": f"{type(e).__name__}: {e}"}
    # ---- the same per-frame call from 16 threads (VapourSynth's worker pool) through ONE coalescing render: havc_batcher ----
    import threading
    T, K = 16, 6
    rc = ModelImageRender(None, "stable", RENDER_FACTOR, 0.5, device_index=ctx.device_id, state_dicts=sds, coalesce=T, precision="fast")
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
    from vsdeoldify_amd.havc import DEF_TWEAK_p
    dd_defaults = dict(ddtweak=(False, False, False), ddtweak_p=(DEF_TWEAK_p, HUE_ADJUST))      # HAVC_colorizer's defaults (__init__.py:2292-2293)
    if ddcolor_only:
        col = HAVCFrameColorizer(method=1, ddcolor_p=(1, 32, 1.0, 0.0, True), device_index=local_rank,
                                 ddcolor_state_dict=synth_ddcolor_state_dict(1), max_batch=args.batch, precision=args.precision, **dd_defaults)
    else:
        col = HAVCFrameColorizer(method=2, mweight=0.4, deoldify_p=(0, 24, 1.0, 0.0), ddcolor_p=(1, 24, 1.0, 0.0, True), device_index=local_rank,
                                 state_dicts={"video": synth_state_dict("wide", 1)}, ddcolor_state_dict=synth_ddcolor_state_dict(1),
                                 max_batch=args.batch, precision=args.precision, **dd_defaults)
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
    from vsdeoldify_amd import _native as nat
    from vsdeoldify_amd.ddcolor_net import TAG_STAGE2_PW1
    tag = TAG_STAGE2_PW1 if ddcolor_only else TAG_TAIL_RES      # the launches each config spends most time in
    for i in range(max(args.warmup, 1)):
        step(i)
    # c4 runs the two models side by side: DDColor lives on a context (HIP stream) of its own, whose work is counted there
    dd_ctx = getattr(getattr(getattr(col, "_ddcolor", None), "rt", None), "ctx", None)
    side = dd_ctx is not None and dd_ctx is not ctx
    ctx.reset_stats()
    if side:
        dd_ctx.reset_stats()
    _raise_steps(args, step, sync_all, world)
    nat.check(ctx.lib.havc_tag_timing_enable(ctx.h, tag, 1), ctx.h)
    sync_all()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    sync_all()
    elapsed = time.perf_counter() - t0
    avg_ms, launches = ctypes.c_double(), ctypes.c_int64()
    nat.check(ctx.lib.havc_tag_timing_read(ctx.h, ctypes.byref(avg_ms), ctypes.byref(launches)), ctx.h)
    nat.check(ctx.lib.havc_tag_timing_enable(ctx.h, tag, 0), ctx.h)
    st = ctx.stats()
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=f"cuda:{local_rank}")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    total = args.steps * args.batch * world
    if ddcolor_only:            # 27 tagged GEMMs per pass: tokens x 768 -> 3072 at 32 x 32 tokens per frame
        per_pass, flops_frame, kname = 27, 2.0 * 32 * 32 * 768 * 3072, "conv_pipe_kernel (ConvNeXt-L stage 2 pwconv1 + GELU, 768 -> 3072 at 32x32 tokens per frame, 27 launches per pass)"
    else:                       # 2 tagged convs per pass: 3x3 259 -> 259 at 384 x 384
        per_pass, flops_frame, kname = 2, 2.0 * 384 * 384 * 259 * 259 * 9, "conv_pipe_kernel<2,4,8,1,0,EF> (DeOldify video layers.10 res-block 3x3 259->259 @384x384, 2 launches per pass: EF=1 and EF=261)"
    fpl = args.steps * args.batch * per_pass / max(int(launches.value), 1)
    achieved = flops_frame * fpl / (avg_ms.value * 1e-3) / 1e12 if avg_ms.value > 0 else 0.0
    roofline = {"bound": "mfma", "achieved": round(achieved, 2), "peak": PEAK_F16_TFLOPS, "unit": "TFLOP/s", "frac": round(achieved / PEAK_F16_TFLOPS, 4),
                "traffic": None, "kernel": kname, "launches_timed": int(launches.value), "frames_per_launch": round(fpl, 2),
                "avg_launch_ms": round(avg_ms.value, 4), "flops_per_launch": flops_frame * fpl}
    # HBM bytes per launch of that kernel from separate --pmc FETCH_SIZE / WRITE_SIZE passes of this very command (tools/pmc_cfg.sh, tools/pmc_cfg_to_json.py),
    # corrected as MI355X_MICROARCH.md prescribes; recorded at its own frames-per-launch and scaled linearly to this run's
    for pf in (os.path.join(ROOT, "profiles", f"r6_{'c3' if ddcolor_only else 'c4'}_pmc.json"), os.path.join(ROOT, "profiles", f"r5_{'c3' if ddcolor_only else 'c4'}_pmc.json")):
        if os.path.isfile(pf) and roofline["traffic"] is None:
            try:
                rec = json.load(open(pf))
                if abs(float(rec.get("frames_per_launch", 0)) - fpl) < 0.5:       # a record taken at another frames-per-launch is not this run's traffic
                    roofline["traffic"] = rec["traffic_bytes_per_launch"]
                    roofline["traffic_source"] = os.path.relpath(pf, ROOT) + " (separate --pmc FETCH_SIZE / WRITE_SIZE passes of this command)"
            except (ValueError, KeyError):
                pass
    tot_flops, tot_frames = st.total_flops, st.frames
    if side:
        st2 = dd_ctx.stats()
        tot_flops, tot_frames = tot_flops + st2.total_flops, tot_frames + st2.frames
        kname += "; timed WHILE the DDColor pass shares the chip (the two models run side by side on two contexts: 5.35 ms = 0.43 alone)"
        roofline["kernel"] = kname
    gflop_frame = tot_flops / max(tot_frames / (1 if ddcolor_only else 2), 1) / 1e9 if tot_frames else 0.0        # every model counts `frames`
    out = {"metric": "colorized frames/sec/GPU @1080p (DDColor large, input 512)" if ddcolor_only else
                     "colorized frames/sec/GPU @1080p (HAVC DeOldify+DDColor merge, combine_method=2)", "value": round(total / elapsed, 3),
           "unit": "frames/s (sum over n_gpus)", "n_gpus": world, "value_per_gpu": round(total / elapsed / world, 3), "steps": args.steps,
           "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
           "dtype": "f16" if args.precision == "fast" else "f16x2 (hi / lo pairs, fp32 accumulate)", "precision": args.precision, "data": "synthetic",
           "config": {"workload": "DDColor modelsize=large, 512 input, 1080p clip (BASELINE.json configs[2])" if ddcolor_only else
                                  "HAVC DeOldify+DDColor merge (combine_method=2) 1080p, frame-sharded (BASELINE.json configs[3])",
                      "frames_per_step_per_gpu": args.batch, "deoldify": None if ddcolor_only else "video, rf=24 (384x384)",
                      "ddcolor": "artistic, input %d (parity UNPINNED)" % (512 if ddcolor_only else 384),
                      "mweight": None if ddcolor_only else 0.4, "algorithmic_gflop_per_frame": round(gflop_frame, 2), "device_resident": True,
                      "parallelism": f"frame-sharded x{world}, weight replica per GPU, no collective"},
           "whole_path_tflops": round(total * gflop_frame * 1e9 / elapsed / 1e12 / world, 2), "roofline": roofline}
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        # the same graph assembled from the oracle pieces on ONE 1080p frame of the clip: timed as the CPU baseline, compared with the GPU frame
        from oracle import ddcolor as D, imaging, pipeline, resample
        threads = cpu_threads(args)
        torch.set_num_threads(threads)
        got = keep[0].frame(0).numpy()
        fr = frames[0]
        t0 = time.time()
        fs = 512 if ddcolor_only else 384
        sq = resample.resize_rgb8(fr, fs, fs)
        from oracle import tweaks
        b_ = tweaks.adjust_hue_range(D.colorize_frame(synth_ddcolor_state_dict(1), sq, input_size=fs), HUE_ADJUST)    # vsmodels.py:365-366
        c_ = b_ if ddcolor_only else pipeline.combine_models(pipeline.model_image_render({"video": synth_state_dict("wide", 1)}, "video", sq, 24, 0, True), b_, 2, 0.4)
        ref = pipeline.post_process(resample.resize_rgb8(c_, WIDTH, HEIGHT), fr)
        dt = time.time() - t0
        de = imaging.delta_e00_images(got, ref)
        d = np.abs(got.astype(np.int32) - ref.astype(np.int32))
        out["cpu_baseline"] = {"value": round(1.0 / dt, 5), "unit": "frames/s", "cores": threads, "host_cpus": os.cpu_count(), "threads_probe": _CPU_THREADS.get("probe"), "kind": "port",
                               "sample": f"1 frame of the 1080p clip through the oracle graph (fp32 torch models + numpy tail), {dt:.1f} s"}
        out["parity"] = dict(_stats(de, d), frames_checked=1,
                             against="oracle graph (CPU fp32); DDColor itself is parity-UNPINNED (external wheel, oracle/ddcolor.py)")
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out))


HUE_ADJUST = "300:360|0.8,0.1"  # HAVC_colorizer's default ddtweak_p[1]: adjust_hue_range on every DDColor frame (vsslib/vsmodels.py:365-366)
C5_H, C5_W = 216, 384          # HAVC_deepex render_speed 'medium' (deepex/__init__.py:58-62); a 16:9 clip needs no borders (vsresize.py:295-316)


def c5_reference_image(gray_small):
    """a colour exemplar for frame 0 (what HAVC_deepex gets from a colorizer or from the user): smooth hue fields over the frame's own luma"""
    h, w, _ = gray_small.shape
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    l = gray_small[..., 0].astype(np.float32)
    col = np.stack([l * 1.05 + 25 * np.sin(yy / 23.0), l * 0.92 + 18 * np.cos(xx / 31.0), l * 0.75 + 20 * np.sin((xx + yy) / 41.0) + 10], -1)
    return np.clip(col, 0, 255).astype(np.uint8)


def bench_c5(args, rank, local_rank, world, dist):
    """BASELINE configs[4]: ColorMNet exemplar path, 1 reference frame, 1080p clip (HAVC_deepex(ex_model=0), __init__.py:1421-1735):
    Spline64 1080p -> 384 x 216 (SmartResizeColorizer) -> ColorMNetRender.colorize_frame, frame after frame (the memory carries state: that
    step is strictly sequential; the key encoder, which depends on the frame alone, runs `lookahead` frames ahead on its own stream)
    -> Spline64 back to 1080p + luma of the source (vs_recover_clip_luma).  The clip stays in HBM; the reference
    image arrives with frame 0 (in the warm-up), the timed steps are steady-state frames incl. memory frames every 5th frame and the
    long-term consolidation.  One step = --batch consecutive frames.  Sequential in time => replicas only across GPUs (one clip per GPU)."""
    import torch
    from vsdeoldify_amd import _native as nat
    from vsdeoldify_amd.clip import synthetic_gray_frame
    from vsdeoldify_amd.colormnet_net import ColorMNetNetwork
    from vsdeoldify_amd.device import DeviceImage
    from vsdeoldify_amd.synth import synth_colormnet_state_dict
    sd = synth_colormnet_state_dict(1)
    torch.cuda.set_device(local_rank)
    net = ColorMNetNetwork(sd, device_index=local_rank)
    ctx = net.ctx
    n_clip = max(args.clip_frames, 8)
    frames = np.stack([synthetic_gray_frame(rank * n_clip + i, WIDTH, HEIGHT) for i in range(n_clip)])
    clip = DeviceImage.from_numpy(ctx, frames)
    from vsdeoldify_amd.colormnet_render import DeepExColorMNet
    dx = DeepExColorMNet(vid_length=10000, render_speed="medium", network=net)          # HAVC_deepex defaults: render_vivid, max_memory_frames = 0
    rnd = dx.render

    def resize(src, sw, sh, out, dw, dh, luma=None):
        nat.check(ctx.lib.havc_spline64_resize(ctx.h, src.ptr, sw, sh, out.ptr, dw, dh, luma.ptr if luma is not None else None), ctx.h)
    small0 = DeviceImage(ctx, (C5_H, C5_W, 3))
    resize(clip.frame(0), WIDTH, HEIGHT, small0, C5_W, C5_H)
    ref_small = c5_reference_image(small0.numpy())
    up = DeviceImage(ctx, (HEIGHT, WIDTH, 3))
    resize(DeviceImage.from_numpy(ctx, ref_small), C5_W, C5_H, up, WIDTH, HEIGHT)
    ref_img = up.numpy()                                            # the exemplar at clip size (what HAVC_deepex receives); squashed again inside
    state = {"t": 0}
    keep = [None]

    def step(_):
        # one step = the next --batch frames of the clip, announced together (DeepExColorMNet.colorize_frames): the key encoder, which does not
        # depend on the memory, runs `lookahead` frames per pass ahead of the frame-by-frame memory step.  DeviceImage in -> DeviceImage out.
        t = state["t"]
        cur = state.pop("next", None) or [clip.frame((t + k) % n_clip) for k in range(args.batch)]
        state["next"] = [clip.frame((t + args.batch + k) % n_clip) for k in range(args.batch)]     # a streaming caller knows what comes next
        keep[0] = dx.colorize_frames(cur, {0: ref_img} if t == 0 else {}, upcoming=state["next"])
        state["t"] = t + args.batch

    def sync_all():
        ctx.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
    for i in range(max(args.warmup, 1)):
        step(i)
    # the launch a frame spends most time in: the 3x3 conv 1536 -> 512 that maps the DINOv2 branch into the 1/8 Fuse block
    plan_net = [v for k, v in net.nets.items() if len(k) == 2][0]                   # the per-frame plan (all slices)
    # the look-ahead plan (key + skip slices, k[3] frames per launch) lives on the helper context whose stream runs the pass concurrently
    look = net._helper if (net.async_lookahead and net._helper is not None) else net
    key_nets = [(k[3], v) for k, v in look.nets.items() if len(k) == 4]
    fpl, tag_net = key_nets[0] if key_nets else (1, plan_net)
    tctx = look.ctx if key_nets else ctx                                          # tag timing is per context: the one that launches the tagged op
    tag_name = "key_encoder.fuse2.encode_enc"
    op = tag_net.plan_ops[tag_net.names.index(tag_name)]
    _raise_steps(args, step, sync_all, world)
    ctx.reset_stats()
    nat.check(tctx.lib.havc_tag_timing_enable(tctx.h, int(op["tag"]), 1), tctx.h)
    sync_all()
    prof = None
    if os.environ.get("HAVC_BENCH_CPROFILE"):                       # tools/c5_host_profile.py: where the HOST spends the timed frames
        import cProfile
        prof = cProfile.Profile()
        prof.enable()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    issued = time.perf_counter() - t0                               # the host has enqueued every frame (the calls only enqueue); the GPU may still be working
    sync_all()
    elapsed = time.perf_counter() - t0
    if prof is not None:
        import io
        import pstats
        prof.disable()
        for key in ("tottime", "cumulative"):
            buf = io.StringIO()
            pstats.Stats(prof, stream=buf).strip_dirs().sort_stats(key).print_stats(40)
            print(buf.getvalue()[:8000], file=sys.stderr)
    avg_ms, launches = ctypes.c_double(), ctypes.c_int64()
    tctx.synchronize()
    nat.check(tctx.lib.havc_tag_timing_read(tctx.h, ctypes.byref(avg_ms), ctypes.byref(launches)), tctx.h)
    nat.check(tctx.lib.havc_tag_timing_enable(tctx.h, int(op["tag"]), 0), tctx.h)
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=f"cuda:{local_rank}")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    total = args.steps * args.batch * world
    sl = plan_net.slices
    gf = {k: sum(int(o["flops"]) for o in plan_net.plan_ops[v[0]:v[0] + v[1]]) * v[2] / 1e9 for k, v in sl.items()}
    # per steady-state frame: key + skip + segment + hidden update + short-term tail every frame, value encoder (+ its hidden update) every 5th
    gflop_frame = gf["key"] + gf["skip"] + gf["segment"] + gf["segment_hidden"] + gf["short"] + (gf["value"] + gf["value_hidden"]) / 5.0
    flops_launch = float(op["flops"]) * fpl
    achieved = flops_launch / (avg_ms.value * 1e-3) / 1e12 if avg_ms.value > 0 else 0.0
    mem = rnd.processor.memory
    out = {"metric": "colorized frames/sec/GPU @1080p (ColorMNet exemplar path, 1 reference frame)", "value": round(total / elapsed, 3),
           "unit": "frames/s (sum over n_gpus)", "n_gpus": world, "value_per_gpu": round(total / elapsed / world, 3), "steps": args.steps,
           "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
           "dtype": "f16", "data": "synthetic",
           "config": {"workload": "ColorMNet exemplar path, 1 reference frame, 1080p clip (BASELINE.json configs[4])", "frames_per_step_per_gpu": args.batch,
                      "network_input": f"{C5_W}x{C5_H} (HAVC_deepex render_speed 'medium'), padded to 448x224 inside the step",
                      "weights": "seeded synthetic (ResNet50 + DINOv2 ViT-S/14 key encoder, ResNet18 value encoder, decoder; DINOv2 branch parity-UNPINNED)",
                      "algorithmic_gflop_per_frame": round(gflop_frame, 2), "mem_every": 5, "device_resident": True,
                      "working_memory_elements": int(mem.work_mem.size), "long_term_elements": int(mem.long_mem.size) if mem.long_mem.engaged() else 0,
                      "key_encoder_lookahead": rnd.lookahead, "read_ahead": bool(getattr(rnd.processor, "reads_ahead", 0)),
                      "host_enqueue_share_of_wall": round(issued / elapsed, 3),      # (includes time BLOCKED in launch calls once the queues are full: ~0.92 on a
                      #  GPU-bound run; tools/c5_host_profile.py shows where the host's own time goes)
                      "parallelism": f"memory step sequential in time (key encoder {rnd.lookahead} frames ahead, concurrently on a second stream): replicas only, one clip per GPU x{world}"},
           "whole_path_tflops": round(total * gflop_frame * 1e9 / elapsed / 1e12 / world, 2),
           "whole_path_frac": round(total * gflop_frame * 1e9 / elapsed / 1e12 / world / PEAK_F16_TFLOPS, 4),
           # `roofline` has the same meaning on every config of this file: the best-utilised TIMED launch against the MFMA peak -- here the
           # look-ahead pass's fuse2.encode_enc at 16 frames per launch.  It is NOT where a ColorMNet frame spends its time: a frame is a latency-bound
           # chain of ~45 - 90 dependent launches of 5 - 40 us on a 14 x 28 grid (2 objects), no launch dominates, and the honest figure for the config
           # is the whole-path rate above (`whole_path_tflops` / `whole_path_frac`); `scope` says which one this object is.
           "roofline": {"bound": "mfma", "scope": "best-utilised timed launch (the frame as a whole is latency-bound: see whole_path_frac)",
                        "achieved": round(achieved, 2), "peak": PEAK_F16_TFLOPS, "unit": "TFLOP/s", "frac": round(achieved / PEAK_F16_TFLOPS, 4), "traffic": None,
                        "kernel": f"conv_pipe_kernel ({tag_name}: 3x3 1536 -> 512 at 28x56, {fpl} frames per launch, look-ahead pass, timed while the memory step "
                                  "shares the chip)",
                        "launches_timed": int(launches.value), "frames_per_launch": fpl, "avg_launch_ms": round(avg_ms.value, 4), "flops_per_launch": flops_launch}}
    if rank == 0 and world == 1 and not args.no_extras:
        # the reference's own call shape: ONE colorize_frame per frame, nothing announced (DeviceImage in -> DeviceImage out: the calls only enqueue)
        dx1 = DeepExColorMNet(vid_length=10000, render_speed="medium", network=net)
        n1 = 4 * args.batch
        for t in range(8):
            dx1.colorize_frame(clip.frame(t % n_clip), ref_img if t == 0 else None)
        sync_all()
        t0 = time.perf_counter()
        for t in range(n1):
            keep[0] = dx1.colorize_frame(clip.frame((8 + t) % n_clip), None)
        sync_all()
        dt = time.perf_counter() - t0
        out["per_frame_calls"] = {"value": round(n1 / dt, 2), "unit": "frames/s", "frames": n1,
                                  "how": "DeepExColorMNet.colorize_frame once per frame, no look-ahead window: the frame's key encoder runs on the look-ahead "
                                         "stream one frame per pass and overlaps the previous frame's memory step"}
        # replicas INSIDE one GPU: R independent clips (scenes), one thread + one context / HIP stream each, packed weights shared.  A single clip
        # is a chain of small launches that leaves most CUs idle; independent clips overlap on the chip.  Not the headline (configs[4] is ONE clip).
        import threading
        R, per = 4, 2 * args.batch
        nets = [net] + [ColorMNetNetwork(None, device_index=local_rank, worker=k, share=net) for k in range(1, R)]
        dxs = [DeepExColorMNet(vid_length=10000, render_speed="medium", network=n_) for n_ in nets]
        clips = [clip] + [DeviceImage.from_numpy(nets[k].ctx, frames) for k in range(1, R)]

        def run(k, count, first):
            with torch.cuda.stream(nets[k].stream):
                dxs[k].colorize_frames([clips[k].frame((t + 3 * k) % n_clip) for t in range(count)], {0: ref_img} if first else {})
            nets[k].ctx.synchronize()
        for k in range(R):                                                   # warm (exemplar, plans, tuning): one replica after the other --
            run(k, 24, True)                                                 # nets are built and tuned from ONE host thread at a time
        ts = [threading.Thread(target=run, args=(k, per, False)) for k in range(R)]          # the timed round: four host threads
        t0 = time.perf_counter()
        for t_ in ts:
            t_.start()
        for t_ in ts:
            t_.join()
        dt = time.perf_counter() - t0
        out["replicas_per_gpu"] = {"clips": R, "frames": R * per, "value": round(R * per / dt, 2), "unit": "frames/s",
                                   "how": "4 independent clips on one GPU, one Python thread + one libhavc context (HIP stream) per clip, shared packed weights"}
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        # CPU oracle over the first frames of the same clip (the exemplar arrives with frame 0): timed, and compared with the GPU's frames
        from oracle import colormnet_clip, imaging, pipeline, resample
        threads = cpu_threads(args)
        torch.set_num_threads(threads)
        K = 4
        dx2 = DeepExColorMNet(vid_length=10000, render_speed="medium", network=net)
        gpu = [o.numpy() for o in dx2.colorize_frames([clip.frame(t) for t in range(K)], {0: ref_img})]      # (the look-ahead path, as timed)
        t0 = time.time()
        smalls = [resample.resize_rgb8(frames[t], C5_W, C5_H) for t in range(K)]
        cols = colormnet_clip.colorize_clip({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, smalls, {0: resample.resize_rgb8(ref_img, C5_W, C5_H)},
                                            vid_length=10000)
        refs = [pipeline.post_process(resample.resize_rgb8(c_, WIDTH, HEIGHT), frames[t]) for t, c_ in enumerate(cols)]
        dt = time.time() - t0
        des = [imaging.delta_e00_images(g_, r_) for g_, r_ in zip(gpu, refs)]
        worst = int(np.argmax([float(np.percentile(d_, 99)) for d_ in des]))
        de = np.concatenate([d_.reshape(-1) for d_ in des])
        out["cpu_baseline"] = {"value": round(K / dt, 5), "unit": "frames/s", "cores": threads, "host_cpus": os.cpu_count(), "threads_probe": _CPU_THREADS.get("probe"), "kind": "port",
                               "sample": f"the first {K} frames of the clip (exemplar with frame 0) through the oracle loop (fp32 torch network + numpy tail), {dt:.1f} s"}
        dd = np.concatenate([np.abs(g_.astype(np.int32) - r_.astype(np.int32)).reshape(-1) for g_, r_ in zip(gpu, refs)])
        out["parity"] = {**_stats(de, dd), "frames_checked": K, "worst_frame": worst, "worst_frame_p99": round(float(np.percentile(des[worst], 99)), 4),
                         "against": "oracle loop (CPU fp32), pinned to the reference's own ColorMNetRender; DINOv2 branch and skimage Lab parity-UNPINNED"}
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out))


if __name__ == "__main__":
    main()
