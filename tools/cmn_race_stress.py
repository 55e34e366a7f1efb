"""Race hunt for the ColorMNet frame loop: the long clip of tests/test_colormnet_net.py
(60 frames, mem_every = 2, a small working memory: consolidations, usage counters, removal of obsolete long-term elements, key
look-ahead of 8 frames, the read of frame t+1 under the decoder of frame t) run N times with the streams of the context and of the
look-ahead context moved against each other by delay kernels of pseudo-random length (havc_debug_stream_jitter), every frame's SHA-1
compared with ONE baseline: the same clip with no jitter, no read-ahead and the look-ahead pass on the step's own stream.

Every comparison is deterministic by construction (inference_core.py:119-230 steps a frame on one stream; the multi-stream schedule
here must give its bytes), so ANY difference is a race (or a machine fault).  On a mismatch the first differing frame, the memory
sizes and the usage counters of both runs are printed, and the run is repeated with READ_AHEAD off / the look-ahead synchronous to
bisect read-ahead vs look-ahead vs consolidation.

Usage (GPU box):  python tools/cmn_race_stress.py [runs=200] [max_us=300] [frames=60]
Exit code 0 = every run identical."""
import hashlib
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np


def make_clip(n_frames):
    g = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "colormnet_net_render.npz"))
    r = np.random.default_rng(4)
    base = g["frames"][0].astype(np.float32)
    frames = [np.stack([np.clip(base + 6 * np.sin(t / 3.0) + r.normal(0, 2, base.shape), 0, 255).astype(np.uint8)] * 3, -1) for t in range(n_frames)]
    return frames, g["refs"][0], int(g["seed"]) if "seed" in g.files else 0


def run_clip(net, frames, ref, lookahead, read_ahead, fast=True):
    """-> (per-frame sha1 list, (work size, long size), usage counters of both stores)"""
    import vsdeoldify_amd.colormnet_fast as cf
    from vsdeoldify_amd.colormnet_render import ColorMNetRender
    from vsdeoldify_amd.device import DeviceImage
    cf.READ_AHEAD, keep = read_ahead, cf.READ_AHEAD
    net.fast = fast
    try:
        rnd = ColorMNetRender(image_size=-1, vid_length=len(frames), encode_mode=1, max_memory_frames=500, reset_on_ref_update=False, network=net,
                              lookahead=lookahead)
        for k, v in (("mem_every", 2), ("max_mid_term_frames", 4), ("min_mid_term_frames", 2), ("num_prototypes", 16), ("max_long_term_elements", 60),
                     ("enable_long_term_count_usage", True)):
            rnd.set_config(k, v)
        dev = [DeviceImage.from_numpy(net.ctx, f) for f in frames]
        refs = [DeviceImage.from_numpy(net.ctx, ref) if t == 0 else None for t in range(len(frames))]
        trace = _install_trace(rnd, len(frames)) if TRACE else None
        outs = [o.numpy() for o in rnd.colorize_batch_frames(dev, refs, False)]
        mem = rnd.processor.memory
        use = [s.get_usage().float().cpu().numpy().ravel() if (s is not None and s.engaged() and s.count_usage) else None for s in (mem.work_mem, mem.long_mem)]
        res = [hashlib.sha1(o.tobytes()).hexdigest() for o in outs], (mem.work_mem.size, mem.long_mem.size), use, outs, getattr(rnd.processor, "reads_ahead", 0)
        if trace is not None:
            LAST_TRACE[0] = trace.cpu().numpy()
        return res
    finally:
        cf.READ_AHEAD = keep
        net.fast = True


# STRESS_TRACE=1 (round 6, after the one mismatch of the closing session): behind every frame's step a few reductions are enqueued ON THE STEP'S STREAM (no host
# synchronisation, so the schedule under test is not serialised): double-precision sums of the decoder's output, of the usage / life counters and of the key / value
# banks.  Compared with the baseline's trace, a mismatching clip then says WHICH quantity left the one-stream schedule first, and at which frame.
TRACE = os.environ.get("STRESS_TRACE", "0") != "0"
TRACE_NAMES = ("prob", "use", "life", "K bank", "V bank", "work n", "long n")
LAST_TRACE = [None]


def _install_trace(rnd, n_frames):
    import torch
    proc = rnd.processor
    log = torch.zeros((n_frames, len(TRACE_NAMES)), dtype=torch.float64, device=rnd.network.device)
    state = {"t": 0}

    def wrap(fn):
        def stepped(*a, **kw):
            out = fn(*a, **kw)
            t = state["t"]
            state["t"] = t + 1
            if t < n_frames and out is not None:
                mem = proc.memory
                b = getattr(mem, "_banks", None)
                log[t, 0] = out.double().sum()
                if b is not None and hasattr(b, "use"):
                    nw, nl = int(mem.work_mem.n), int(mem.long_mem.n if mem.long_mem is not None else 0)
                    lo, hi = b.cap_long - nl, b.cap_long + nw                 # the live columns [long | work] (the banks are torch.empty beyond them)
                    log[t, 1] = b.use[..., lo:hi].double().sum(); log[t, 2] = b.life[..., lo:hi].double().sum()
                    log[t, 3] = b.K[..., lo:hi].double().sum(); log[t, 4] = b.V[..., lo:hi].double().sum()
                    log[t, 5] = float(nw); log[t, 6] = float(nl)
            return out
        return stepped
    for name in ("step_padded", "step_AnyExemplar_padded"):
        if hasattr(proc, name):
            setattr(proc, name, wrap(getattr(proc, name)))
    return log


def describe_trace(got, base):
    if got is None or base is None:
        return
    d = got != base
    if not d.any():
        print("    trace: identical (the difference is not in a traced quantity)")
        return
    for k, name in enumerate(TRACE_NAMES):
        rows = np.nonzero(d[:, k])[0]
        if rows.size:
            t = int(rows[0])
            print(f"    trace: {name:7s} first differs at frame {t}: {got[t, k]!r} vs baseline {base[t, k]!r} (relative {abs(got[t, k] - base[t, k]) / max(abs(base[t, k]), 1e-300):.3g})")


def describe(tag, got, base):
    sha, sizes, use, outs, _ = got
    bsha, bsizes, buse, bouts, _ = base
    first = next((i for i, (a, b) in enumerate(zip(sha, bsha)) if a != b), None)
    print(f"  {tag}: first differing frame {first}; memory sizes {sizes} vs baseline {bsizes}")
    if first is not None:
        d = np.abs(outs[first].astype(int) - bouts[first].astype(int))
        print(f"    frame {first}: max |d| {int(d.max())}, differing bytes {float((d > 0).mean()):.5f}; differing frames {[i for i, (a, b) in enumerate(zip(sha, bsha)) if a != b]}")
    for name, u, bu in (("work", use[0], buse[0]), ("long", use[1], buse[1])):
        if u is not None and bu is not None:
            same = u.shape == bu.shape and np.array_equal(u, bu)
            print(f"    usage counters ({name}): {'identical' if same else 'DIFFER: ' + str(u.shape) + ' vs ' + str(bu.shape)}")


def main():
    runs = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    max_us = int(sys.argv[2]) if len(sys.argv) > 2 else 300
    n_frames = int(sys.argv[3]) if len(sys.argv) > 3 else 60
    # bisecting switches (round 6): frames per look-ahead window, read-ahead on / off, look-ahead pass on its own stream / on the step's stream
    L = int(os.environ.get("STRESS_LOOKAHEAD", "8"))
    RA = os.environ.get("STRESS_READ_AHEAD", "1") != "0"
    ASYNC = os.environ.get("STRESS_ASYNC_LOOKAHEAD", "1") != "0"
    SEED0 = int(os.environ.get("STRESS_SEED0", "1000"))            # first jitter seed (run i uses SEED0 + i)
    from vsdeoldify_amd import _native as nat
    from vsdeoldify_amd.colormnet_net import ColorMNetNetwork
    from vsdeoldify_amd.synth import synth_colormnet_state_dict
    frames, ref, seed = make_clip(n_frames)
    g = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "colormnet_net_modules.npz"))
    net = ColorMNetNetwork(synth_colormnet_state_dict(int(g["seed"])), device_index=0)
    lib = nat.load()
    lib.havc_debug_stream_jitter(0, 1)
    # the baseline: one stream for everything the frame loop does (no read-ahead; the look-ahead pass on the step's own stream), no jitter
    net.async_lookahead = False
    base = run_clip(net, frames, ref, L, False)
    base_trace = LAST_TRACE[0]
    net.async_lookahead = ASYNC
    plain = run_clip(net, frames, ref, L, RA)                        # the product schedule, un-jittered
    print(f"variant: look-ahead window {L}, read-ahead {RA}, look-ahead pass on its own stream {ASYNC}", flush=True)
    print(f"baseline: {n_frames} frames, memory sizes {base[1]}; product schedule un-jittered: {'identical' if plain[0] == base[0] else 'DIFFERENT'}, "
          f"{plain[4]} reads ran ahead", flush=True)
    bad = 0 if plain[0] == base[0] else 1
    if bad:
        describe("un-jittered product schedule", plain, base)
    t0 = time.time()
    for i in range(runs):
        lib.havc_debug_stream_jitter(SEED0 + i, max_us)
        got = run_clip(net, frames, ref, L, RA)
        lib.havc_debug_stream_jitter(0, 1)
        if got[0] != base[0] or got[1] != base[1]:
            bad += 1
            print(f"run {i} (jitter seed {SEED0 + i}): MISMATCH", flush=True)
            describe("jittered", got, base)
            describe_trace(LAST_TRACE[0], base_trace)
            for tag, la, ra in (("same seed, READ_AHEAD off", True, False), ("same seed, look-ahead synchronous", False, True)):
                lib.havc_debug_stream_jitter(SEED0 + i, max_us)
                net.async_lookahead = la
                again = run_clip(net, frames, ref, L, ra)
                net.async_lookahead = ASYNC
                lib.havc_debug_stream_jitter(0, 1)
                print(f"  {tag}: {'identical to the baseline' if again[0] == base[0] else 'MISMATCH'}", flush=True)
        if (i + 1) % 20 == 0:
            print(f"{i + 1} jittered runs, {bad} mismatches, {time.time() - t0:.0f} s", flush=True)
    print(f"cmn_race_stress: {runs} jittered runs (delays up to {max_us} us), {n_frames} frames each, {bad} mismatches")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
