#!/usr/bin/env python3
"""ColorMNet exemplar path on the GPU: frames/s of ColorMNetRender.colorize_frame at the HAVC_deepex sizes (deepex/__init__.py:58-68) and a
per-slice / per-op profile of the plan.   python tools/colormnet_clip_bench.py [frames] [h] [w]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from PIL import Image  # noqa: E402

from vsdeoldify_amd.colormnet_net import ColorMNetNetwork  # noqa: E402
from vsdeoldify_amd.colormnet_render import ColorMNetRender  # noqa: E402
from vsdeoldify_amd.synth import synth_colormnet_state_dict  # noqa: E402

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 40
h = int(sys.argv[2]) if len(sys.argv) > 2 else 216
w = int(sys.argv[3]) if len(sys.argv) > 3 else 384
t0 = time.time()
net = ColorMNetNetwork(synth_colormnet_state_dict(3))
print(f"pack + upload {time.time() - t0:.1f} s", flush=True)
r = np.random.default_rng(0)
clip = [np.stack([np.clip(128 + 40 * r.standard_normal((h, w)), 0, 255).astype(np.uint8)] * 3, -1) for _ in range(8)]
ref = np.clip(clip[0].astype(np.float32) * [1.1, 0.9, 0.7], 0, 255).astype(np.uint8)
rnd = ColorMNetRender(vid_length=10 ** 6, network=net, reset_on_ref_update=False)
t0 = time.time()
rnd.set_ref_frame(Image.fromarray(ref), False)
rnd.colorize_frame(0, Image.fromarray(clip[0]))
print(f"first frame (plan + autotune) {time.time() - t0:.1f} s")
for rep in range(2):
    t0 = time.time()
    for t in range(frames):
        rnd.set_ref_frame(None)
        rnd.colorize_frame(t + 1, Image.fromarray(clip[t % 8]))
    dt = time.time() - t0
    print(f"{frames} frames {h}x{w}: {dt / frames * 1e3:.2f} ms / frame = {frames / dt:.1f} frames/s; work mem {rnd.processor.memory.work_mem.size}, "
          f"long mem {rnd.processor.memory.long_mem.size if rnd.processor.memory.long_mem.engaged() else 0}")
n = list(net.nets.values())[0]
for buf in n.io.values():
    n.bind(buf, None)                                          # the per-frame tensors are gone: profile on the plan's own buffers
tot = {}
for name, (first, count, batch) in n.slices.items():
    ms = n.profile(batch)[first:first + count]
    fl = sum(int(o["flops"]) for o in n.plan_ops[first:first + count]) * batch
    tot[name] = float(ms.sum())
    print(f"slice {name:15s} batch {batch}: {ms.sum():7.3f} ms, {count:3d} ops, {fl / 1e9:7.1f} GFLOP -> {fl / 1e9 / max(ms.sum(), 1e-6):7.1f} TFLOP/s")
    if os.environ.get("PEROP"):
        for i in np.argsort(-ms)[:12]:
            o = n.plan_ops[first + i]
            print(f"      {ms[i]:7.3f} ms  {n.names[first + i]:60s} type {int(o['type'])} {int(o['flops']) * batch / 1e9:6.2f} GF")
print("sum of slices (a frame that reads the memory and is memorised):", round(sum(tot.values()), 3), "ms")
