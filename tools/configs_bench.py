"""Throughput of the BASELINE.json configurations other than the headline one (which bench.py measures), host buffers in -> host
buffers out on one GPU, seeded synthetic weights:
  C1  eccv16 / siggraph17, 256x256 network input, 1080p frame            (ModelColorization.colorize_frame)
  C3  DDColor large, input 512, 512x512 frame                            (vsddcolor.ddcolor stand-in; parity unpinned)
  C4  HAVC merge method 2 defaults: DeOldify video rf=24 + DDColor rf=24 at 384x384, Image.blend(mweight 0.4 -> weight of b)
Usage: python tools/configs_bench.py [frames]"""
import sys, os, time
import os as _os
_os.environ.setdefault("HAVC_PRECISION", "fast")      # this tool measures the fast (fp16) mode unless told otherwise; the package default is "precise"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from vsdeoldify_amd import mcomb
from vsdeoldify_amd.clip import synthetic_gray_frame
from vsdeoldify_amd.colorization import ModelColorization
from vsdeoldify_amd.ddcolor import DDColorRuntime
from vsdeoldify_amd.render import GeneratorRuntime, get_context
from vsdeoldify_amd import _native as nat
from vsdeoldify_amd.synth import synth_ddcolor_state_dict, synth_state_dict, synth_zhang_state_dict

N = int(sys.argv[1]) if len(sys.argv) > 1 else 16
ctx = get_context(0)


def timed(fn, reps=3):
    fn()
    ts = []
    for _ in range(reps):
        t = time.time(); fn(); ts.append(time.time() - t)
    return min(ts)


# ---- C1: Zhang ----
frames1080 = np.stack([synthetic_gray_frame(i, 1920, 1080) for i in range(4)])
for name in ("eccv16", "siggraph17"):
    mc = ModelColorization(name, use_gpu=True, state_dict=synth_zhang_state_dict(name, 1))
    t = timed(lambda: [mc.colorize_frame(f) for f in frames1080])
    print(f"C1 {name:10s} 1080p frames, net 256x256, one frame per call: {len(frames1080)/t:7.1f} frames/s ({t/len(frames1080)*1e3:.2f} ms/frame)")

# ---- C3: DDColor 512 ----
dd = DDColorRuntime(ctx, synth_ddcolor_state_dict(1))
f512 = np.stack([np.asarray(synthetic_gray_frame(i, 512, 512)) for i in range(N)])
t = timed(lambda: dd.colorize(f512))
print(f"C3 DDColor large, input 512, {N} frames of 512x512 per call (batches of 8): {N/t:7.1f} frames/s ({t/N*1e3:.2f} ms/frame), 499 GFLOP/frame -> {499e-3*N/t:.0f} TFLOP/s")

# ---- C4: DeOldify video rf=24 + DDColor rf=24 @384, method 2 (SimpleMerge) ----
S = 384
rt = GeneratorRuntime(ctx, synth_state_dict("wide", 1), "wide")
f384 = np.stack([np.asarray(synthetic_gray_frame(i, S, S)) for i in range(N)])


def c4():
    net = rt.net(S, 8)
    a = np.empty_like(f384)
    nat.check(ctx.lib.havc_deoldify_frames(ctx.h, net.h, None, 0.0, 1, nat.as_ptr(f384), nat.as_ptr(a), len(f384)), ctx.h)
    b = dd.colorize(f384)
    return [mcomb.simple_merge(x, y, 0.4) for x, y in zip(a, b)]


t = timed(c4)
print(f"C4 merge method 2 @384 (DeOldify video 639 GF + DDColor 281 GF + blend), {N} frames: {N/t:7.1f} frames/s/GPU ({t/N*1e3:.2f} ms/frame)")
