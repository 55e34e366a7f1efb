"""Concurrent set-up stress: N host threads build and first-call {DeOldify, DDColor, ColorMNet} runtimes at the same time, on
fresh contexts, over and over -- what VapourSynth's worker threads do when a script builds several models (vsslib/vsmodels.py:196-233).
Each set-up = new context (streams, scratch warm-up) + weight upload + net creation + tile autotuning + first launches of every kernel; the
results must be byte-identical to a single-threaded baseline.  Weights are PACKED once (host work), everything on the GPU side is fresh per set-up.

  python tools/setup_stress.py [--reps 20] [--threads 4] [--kinds deoldify,ddcolor,colormnet]
  HAVC_SETUP_MUTEX=0 / HAVC_EAGER_SETUP=0 turn the library's protections off (bisecting only: this may hang the GPU).
Run it as a CHILD process under a timeout (tests/test_gpu_setup_stress.py does); exit code 0 = every set-up finished with the right bytes."""
import argparse
import json
import os
import sys
import threading
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("HAVC_TUNE_CACHE", "0")

from vsdeoldify_amd import _native as nat                                  # noqa: E402
from vsdeoldify_amd import render                                          # noqa: E402


def make_frames(n, s, seed):
    r = np.random.default_rng(seed)
    luma = np.clip(128 + 50 * r.standard_normal((n, s, s)), 0, 255).astype(np.uint8)
    return np.ascontiguousarray(np.stack([luma] * 3, -1))


class Kinds:
    """packed models (host side, built once) + one fresh GPU-side set-up per call"""

    def __init__(self, kinds):
        from vsdeoldify_amd.synth import synth_colormnet_state_dict, synth_ddcolor_state_dict, synth_state_dict
        self.kinds = kinds
        if "deoldify" in kinds:
            from vsdeoldify_amd.deoldify_net import DeoldifyGenerator
            self.deo = DeoldifyGenerator(synth_state_dict("deep", 3), "deep")
            self.deo_frames = make_frames(2, 96, 1)
        if "ddcolor" in kinds:
            from vsdeoldify_amd.ddcolor_net import DDColorGenerator
            self.dd = DDColorGenerator(synth_ddcolor_state_dict(1), (3, 3, 27, 3), 9)
            self.dd_frames = make_frames(2, 64, 2)
        if "colormnet" in kinds:
            from vsdeoldify_amd.colormnet_net import ColorMNetPlan
            self.cmn_sd = synth_colormnet_state_dict(1)
            self.cmn = ColorMNetPlan(self.cmn_sd)

    def run(self, kind, key):
        ctx = render.get_context(0, key)
        try:
            if kind == "deoldify":
                rt = render.GeneratorRuntime(ctx, None, "deep", generator=self.deo)
                try:
                    net = rt.net(96, 2)
                    out = np.empty_like(self.deo_frames)
                    nat.check(ctx.lib.havc_deoldify_frames(ctx.h, net.h, None, 0.0, 0, nat.as_ptr(self.deo_frames), nat.as_ptr(out), 2), ctx.h)
                    return out
                finally:
                    rt.close()
            if kind == "ddcolor":
                from vsdeoldify_amd.ddcolor import DDColorRuntime
                rt = DDColorRuntime.__new__(DDColorRuntime)
                rt.ctx, rt.gen, rt.nets = ctx, self.dd, {}
                rt.weights, rt._owns_weights = nat.Weights(ctx, self.dd.blob), True
                try:
                    return rt.colorize(self.dd_frames, 64, max_batch=2)
                finally:
                    for n in rt.nets.values():
                        n.close()
                    rt.weights.close()
            if kind == "colormnet":
                import torch
                from vsdeoldify_amd.colormnet_net import ColorMNetNetwork
                holder = type("H", (), {})()
                holder.plan, holder.weights, holder.ctx = self.cmn, nat.Weights(ctx, self.cmn.blob), ctx
                net = ColorMNetNetwork(None, device_index=0, worker=key, share=holder)
                try:
                    g = torch.Generator().manual_seed(5)
                    frame = torch.randn(1, 3, 112, 224, generator=g).to(net.device)
                    with net.on_stream():
                        key_, shr, sel, f16, f8, f4 = net.encode_key(frame)
                        res = key_.float().cpu().numpy()
                    return res
                finally:
                    net.close()
                    holder.weights.close()
            raise ValueError(kind)
        finally:
            ctx.synchronize()
            render._contexts.pop((0, key), None)
            ctx.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--threads", type=int, default=4)
    ap.add_argument("--kinds", default="deoldify,ddcolor,colormnet")
    args = ap.parse_args()
    kinds = args.kinds.split(",")
    t0 = time.time()
    K = Kinds(kinds)
    base = {k: K.run(k, ("stress-base", k)) for k in kinds}                # single-threaded baseline
    print(f"packed + baseline in {time.time() - t0:.1f} s", flush=True)
    errors, done = [], [0]
    lock = threading.Lock()

    def worker(rep, t, barrier):
        kind = kinds[(t + rep) % len(kinds)]
        try:
            barrier.wait()
            got = K.run(kind, ("stress", rep, t))
            same = np.array_equal(got, base[kind])
            with lock:
                done[0] += 1
                if not same:
                    errors.append(f"rep {rep} thread {t} {kind}: bytes differ from the single-threaded baseline")
        except Exception as e:                                            # noqa: BLE001
            with lock:
                errors.append(f"rep {rep} thread {t} {kind}: {type(e).__name__}: {e}")
    t1 = time.time()
    for rep in range(args.reps):
        barrier = threading.Barrier(args.threads)
        ts = [threading.Thread(target=worker, args=(rep, t, barrier)) for t in range(args.threads)]
        for th in ts:
            th.start()
        for th in ts:
            th.join()
        if errors:
            break
    out = {"setups": done[0], "reps": rep + 1, "threads": args.threads, "kinds": kinds, "seconds": round(time.time() - t1, 1), "errors": errors[:5],
           "setup_mutex": os.environ.get("HAVC_SETUP_MUTEX", "1"), "eager_setup": os.environ.get("HAVC_EAGER_SETUP", "1")}
    print(json.dumps(out), flush=True)
    return 1 if errors else 0


if __name__ == "__main__":
    sys.exit(main())
