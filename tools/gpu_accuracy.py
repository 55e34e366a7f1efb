"""GPU-vs-oracle accuracy table for the DeOldify generators (raw colour + final image). Run on the GPU box."""
import sys, os, time
import os as _os
_os.environ.setdefault("HAVC_PRECISION", "fast")      # this tool measures the fast (fp16) mode unless told otherwise; the package default is "precise"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import imaging, pipeline
from vsdeoldify_amd import _native as nat
from vsdeoldify_amd.render import GeneratorRuntime, get_context
from vsdeoldify_amd.synth import synth_state_dict
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from tests.test_gpu_deoldify import make_frame

sizes = [int(a) for a in sys.argv[1:]] or [64, 96, 160, 256]
ctx = get_context(0)
for arch, which, seed in (("wide", "video", 1), ("deep", "artistic", 3)):
    sd = synth_state_dict(arch, seed)
    rt = GeneratorRuntime(ctx, sd, arch)
    for S in sizes:
        f = make_frame(S, S)
        net = rt.net(S, 1)
        got = np.empty((1, S, S, 3), np.uint8)
        nat.check(ctx.lib.havc_deoldify_frames(ctx.h, net.h, None, 0.0, 0, nat.as_ptr(np.ascontiguousarray(f[None])), nat.as_ptr(got), 1), ctx.h)
        t = time.time(); ref = pipeline.raw_color_square(sd, arch, f); dt = time.time() - t
        d = np.abs(got[0].astype(int) - ref.astype(int))
        de = imaging.delta_e00_images(got[0], ref)
        fin_g, fin_r = pipeline.post_process(got[0], f), pipeline.post_process(ref, f)
        de2 = imaging.delta_e00_images(fin_g, fin_r)
        print(f"{arch} S={S}: raw LSB max {d.max()} within1 {100*(d<=1).mean():.2f}% within2 {100*(d<=2).mean():.3f}% | "
              f"dE00 raw mean {de.mean():.4f} p99 {np.percentile(de,99):.3f} max {de.max():.3f} | "
              f"final mean {de2.mean():.4f} p99 {np.percentile(de2,99):.3f} p99.9 {np.percentile(de2,99.9):.3f} max {de2.max():.3f} | cpu {dt:.1f}s", flush=True)
    rt.close()
