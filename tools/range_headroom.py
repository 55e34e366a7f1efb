#!/usr/bin/env python3
"""fp16 head-room of a checkpoint: per-op abs-max of every activation buffer with the range check on (HAVC_RANGE_CHECK, include/havc_mi355.h).
   python tools/range_headroom.py            # the seeded synthetic weights of bench.py: wide 560 (video, stable), DDColor 512, ColorMNet 224x448
With real checkpoints: pass state dicts to the same classes; a run that overflows raises HavcRangeError naming the op."""
import os
import os as _os
_os.environ.setdefault("HAVC_PRECISION", "fast")      # this tool measures the fast (fp16) mode unless told otherwise; the package default is "precise"
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["HAVC_RANGE_CHECK"] = "1"
os.environ.setdefault("HAVC_AUTOTUNE", "0")
from vsdeoldify_amd.clip import synthetic_gray_frame  # noqa: E402
from vsdeoldify_amd.render import ModelImageRender  # noqa: E402
from vsdeoldify_amd.synth import synth_ddcolor_state_dict, synth_state_dict  # noqa: E402
from oracle import resample  # noqa: E402  (harness resize of the bench frame; this tool is not product code)


def table(title, net, top=8):
    amax, bad = net.range_stats()
    order = np.argsort(-amax)
    print(f"== {title}: {len(amax)} ops, non-finite values {int(bad.sum())}, largest |x| {amax.max():.1f} = 2^{np.log2(max(amax.max(), 1e-9)):.1f} "
          f"(fp16 max 65504 = 2^16: head-room {65504 / max(amax.max(), 1e-9):.0f}x)")
    for i in order[:top]:
        print(f"   {amax[i]:10.2f}  {net.names[i] if hasattr(net, 'names') else i}")


frame = synthetic_gray_frame(0, 1920, 1080)
from PIL import Image  # noqa: E402
sq = resample.resize_rgb8(frame, 560, 560)
for name, seed in (("video", 1), ("stable", 2)):
    r = ModelImageRender(None, "video", 35, 0, state_dicts={"video": synth_state_dict("wide", seed)})
    r.get_transformed_image(Image.fromarray(sq))
    table(f"DeOldify wide 560x560, seed {seed} ({name} weights of bench.py)", r._video.net(560, 1))
from vsdeoldify_amd.ddcolor import DDColorRender  # noqa: E402
d = DDColorRender(model=1, input_size=512, state_dict=synth_ddcolor_state_dict(1))
d.colorize_frame(resample.resize_rgb8(frame, 512, 512))
table("DDColor large, input 512", d.rt.net(512, 1))
from vsdeoldify_amd.colormnet_net import ColorMNetNetwork  # noqa: E402
from vsdeoldify_amd.colormnet_render import ColorMNetRender  # noqa: E402
from vsdeoldify_amd.synth import synth_colormnet_state_dict  # noqa: E402
net = ColorMNetNetwork(synth_colormnet_state_dict(1), autotune=False)
rnd = ColorMNetRender(vid_length=100, reset_on_ref_update=False, network=net)
small = resample.resize_rgb8(frame, 384, 216)
ref = np.clip(small.astype(np.float32) * [1.1, 0.9, 0.75], 0, 255).astype(np.uint8)
for t in range(3):
    rnd.set_ref_frame(Image.fromarray(ref) if t == 0 else None, False)
    rnd.colorize_frame(t, Image.fromarray(small))
import torch  # noqa: E402
torch.cuda.synchronize()
print("== ColorMNet 216x384 (3 frames, exemplar with frame 0): every slice ran range-checked without a non-finite value")
