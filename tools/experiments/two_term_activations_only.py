"""CPU experiment (round 5): is a TWO-term precise conv enough?  See DESIGN.md section 3.  Conv inputs rounded to fp16, weights fp32, through the whole configs[1] oracle pipeline."""
import sys, time, numpy as np
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__)))))
from oracle import imaging, pipeline, unet
from vsdeoldify_amd.synth import synth_state_dict
from vsdeoldify_amd.clip import synthetic_gray_frame
import torch
import torch.nn.functional as F
torch.set_num_threads(8)
orig = F.conv2d
class Fp:
    def __getattr__(self, k): return getattr(F, k)
    @staticmethod
    def conv2d(x, *a, **k): return orig(x.half().float(), *a, **k)
for sv, ss in ((11, 12), (1, 2)):
    sds = {"video": synth_state_dict("wide", sv), "stable": synth_state_dict("wide", ss)}
    fr = synthetic_gray_frame(0, 1920, 1080)
    t0 = time.time()
    ref = pipeline.colorize_frame_fullsize(sds, "stable", fr, 35, 0.5)
    unet.F = Fp()
    got = pipeline.colorize_frame_fullsize(sds, "stable", fr, 35, 0.5)
    unet.F = F
    de = imaging.delta_e00_images(got, ref)
    print(f"seeds {sv},{ss}: conv INPUTS rounded to fp16, weights fp32: mean {de.mean():.4f} p99 {np.percentile(de,99):.3f} max {de.max():.2f} dE<1 {(de<1).mean():.5f}  ({time.time()-t0:.0f} s)", flush=True)
