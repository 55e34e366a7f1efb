"""Development probe of the precise (hi / lo fp16 pair) DeOldify path on the GPU: raw network colour vs the fp32 CPU oracle at small sizes,
the fast path beside it, and the time of a pass.  python tools/precise_dev.py [S ...]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import imaging, pipeline                                   # noqa: E402  (checker only)
from tests.test_gpu_deoldify import make_frame, raw_gpu, summarize     # noqa: E402
from vsdeoldify_amd.render import GeneratorRuntime, get_context        # noqa: E402
from vsdeoldify_amd.synth import synth_state_dict                      # noqa: E402


def main():
    sizes = [int(a) for a in sys.argv[1:]] or [64, 80, 96]
    ctx = get_context(0)
    for arch, seed in (("wide", 1), ("deep", 3)):
        sd = synth_state_dict(arch, seed)
        rts = {p: GeneratorRuntime(ctx, sd, arch, precision=p) for p in ("fast", "precise")}
        for S in sizes:
            frames = np.stack([make_frame(S, 10 + S), make_frame(S, 11 + S)])
            ref = np.stack([pipeline.raw_color_square(sd, arch, f) for f in frames])
            for p, rt in rts.items():
                t0 = time.time()
                got = raw_gpu(ctx, rt, frames)
                dt = time.time() - t0
                de = imaging.delta_e00_images(got, ref)
                d = np.abs(got.astype(int) - ref.astype(int))
                print(f"{arch} S={S} {p:8s}: bytes equal {float((d == 0).mean()):.6f} max |d| {int(d.max())} mean dE00 {de.mean():.5f} "
                      f"p99 {np.percentile(de, 99):.4f} max {de.max():.3f}  ({dt:.2f} s incl. net build)", flush=True)
        for rt in rts.values():
            rt.close()


if __name__ == "__main__":
    main()
