"""Does it pay to run the two generator passes of a batch OUT OF PHASE?  Two contexts, one 'video' DynamicUnetWide pass each in a loop
(560 x 560, 32 frames per pass), started together (lockstep, re-aligned every pass through a barrier) or free-running with thread B
started half a pass late.  Prints generator frames/s for both.  Usage: python tools/stagger_probe.py [batch] [passes]"""
import sys, os, threading, time
import os as _os
_os.environ.setdefault("HAVC_PRECISION", "fast")      # this tool measures the fast (fp16) mode unless told otherwise; the package default is "precise"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from vsdeoldify_amd.device import DeviceImage
from vsdeoldify_amd.render import ModelImageRender
from vsdeoldify_amd.synth import synth_state_dict

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
N = int(sys.argv[2]) if len(sys.argv) > 2 else 12
S = 560
sds = {"video": synth_state_dict("wide", 1)}
rs = [ModelImageRender(None, "video", 35, device_index=0, state_dicts=sds, max_batch=B, worker=k) for k in range(2)]
frames = np.random.default_rng(0).integers(0, 256, (B, S, S, 3), dtype=np.uint8)
clips = [DeviceImage.from_numpy(r.ctx, frames) for r in rs]
for r, c in zip(rs, clips):
    for _ in range(2):
        r.render_square_batch(c, post_process=False)
    r.ctx.synchronize()
t0 = time.perf_counter(); rs[0].render_square_batch(clips[0], post_process=False); rs[0].ctx.synchronize(); one = time.perf_counter() - t0
print(f"one pass alone: {one * 1e3:.1f} ms for {B} frames = {B / one:.1f} generator frames/s")


def run(mode):
    bar = threading.Barrier(2)

    def work(k):
        r, c = rs[k], clips[k]
        bar.wait()
        if mode == "stagger" and k == 1:
            time.sleep(one / 2)
        for i in range(N):
            if mode == "lockstep":
                bar.wait()
            keep = r.render_square_batch(c, post_process=False)
            if mode == "lockstep":
                r.ctx.synchronize()
        r.ctx.synchronize()
    ts = [threading.Thread(target=work, args=(k,)) for k in range(2)]
    t0 = time.perf_counter()
    for t in ts: t.start()
    for t in ts: t.join()
    dt = time.perf_counter() - t0 - (one / 2 if mode == "stagger" else 0)
    print(f"{mode:9s}: {2 * N * B / dt:8.1f} generator frames/s  ({dt / N * 1e3:.1f} ms per pair of passes)", flush=True)

for mode in ("lockstep", "stagger", "free", "lockstep", "stagger"):
    run(mode)
