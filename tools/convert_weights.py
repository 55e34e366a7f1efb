#!/usr/bin/env python3
"""Offline weight converter: a DeOldify checkpoint -> the packed device blob of libhavc_mi355.

  python tools/convert_weights.py /path/to/models/ColorizeStable_gen.pth            # writes ColorizeStable_gen.havc next to it
  python tools/convert_weights.py in.pth out.havc --arch deep

The .pth is read exactly like Learner.load does (fastai/basic_train.py:264-286: {'model': sd, 'opt': ...} or a bare state dict);
spectral / weight norm are resolved with the STORED u, v, conv->BN pairs folded, every conv laid out as the fp16
[Npad][tap][Cin/8][8] matrix the implicit-GEMM kernels stream (vsdeoldify_amd/plan.py).  ModelImageRender picks the .havc file up
when it sits next to the .pth and is not older (vsdeoldify_amd/render.py): model start-up drops from seconds of packing to a
file read + one H2D copy.  CPU only: needs neither the GPU nor the HIP library."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("pth")
    ap.add_argument("out", nargs="?")
    ap.add_argument("--arch", choices=["wide", "deep"], help="default: deep for *Artistic*, wide otherwise")
    a = ap.parse_args()
    from vsdeoldify_amd.deoldify_net import DeoldifyGenerator
    from vsdeoldify_amd.render import _load_pth
    arch = a.arch or ("deep" if "artistic" in os.path.basename(a.pth).lower() else "wide")
    out = a.out or os.path.splitext(a.pth)[0] + ".havc"
    t = time.time()
    gen = DeoldifyGenerator(_load_pth(a.pth), arch)
    gen.save(out)
    print(f"{a.pth} ({arch}) -> {out}: {len(gen.blob) / 1e6:.1f} MB packed, {len(gen._pc)} convs, {time.time() - t:.1f} s")


if __name__ == "__main__":
    main()
