#!/usr/bin/env python3
"""One-command validator for a REAL DeOldify checkpoint on the MI355X path (the trained
Colorize{Video,Stable,Artistic}_gen.pth files cannot be fetched in the build sandbox, every number in this repository is on seeded weights).

  python tools/validate_checkpoint.py /path/to/ColorizeVideo_gen.pth [--arch wide|deep] [--render-factor 21] [--frames 4] [--image a.png ...]

Runs on the GPU only (no oracle): (1) the fast path with the fp16 RANGE CHECK on (HAVC_RANGE_CHECK: every op's destination scanned for inf / NaN,
largest |activation| reported: the head-room of the fp16 contract); (2) the PRECISE path (fp32-class arithmetic, the
reference's numerics: deoldify/filters.py:45-68) on the same frames; (3) CIEDE2000 between the two final images (network colour + the reference's
YUV post-process).  Verdict: OK when no activation overflowed and the fast path stays within the stated tolerance of the precise one
(mean < 0.35, p99 < 2.6, >= 84 % of the pixels below 1.0: the bounds the seeded weight sets meet, tests/test_gpu_precise.py FAST_CLIP);
otherwise use ModelImageRender(precision="precise") / HAVC_PRECISION=precise for this checkpoint.  Exit code 0 / 1."""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def delta_e00(lab1, lab2):
    """CIEDE2000 (Sharma, Wu, Dalal 2005) on Lab arrays [..., 3]; product-side copy of the textbook formula (the oracle has its own)"""
    L1, a1, b1 = lab1[..., 0], lab1[..., 1], lab1[..., 2]
    L2, a2, b2 = lab2[..., 0], lab2[..., 1], lab2[..., 2]
    C1, C2 = np.hypot(a1, b1), np.hypot(a2, b2)
    Cm = (C1 + C2) / 2
    G = 0.5 * (1 - np.sqrt(Cm ** 7 / (Cm ** 7 + 25.0 ** 7)))
    a1p, a2p = (1 + G) * a1, (1 + G) * a2
    C1p, C2p = np.hypot(a1p, b1), np.hypot(a2p, b2)
    h1p = np.degrees(np.arctan2(b1, a1p)) % 360
    h2p = np.degrees(np.arctan2(b2, a2p)) % 360
    dLp, dCp = L2 - L1, C2p - C1p
    dh = h2p - h1p
    dh = np.where(C1p * C2p == 0, 0, np.where(dh > 180, dh - 360, np.where(dh < -180, dh + 360, dh)))
    dHp = 2 * np.sqrt(C1p * C2p) * np.sin(np.radians(dh / 2))
    Lm, Cpm = (L1 + L2) / 2, (C1p + C2p) / 2
    hs = h1p + h2p
    hm = np.where(C1p * C2p == 0, hs, np.where(np.abs(h1p - h2p) <= 180, hs / 2, np.where(hs < 360, (hs + 360) / 2, (hs - 360) / 2)))
    T = 1 - 0.17 * np.cos(np.radians(hm - 30)) + 0.24 * np.cos(np.radians(2 * hm)) + 0.32 * np.cos(np.radians(3 * hm + 6)) - 0.20 * np.cos(np.radians(4 * hm - 63))
    Sl = 1 + 0.015 * (Lm - 50) ** 2 / np.sqrt(20 + (Lm - 50) ** 2)
    Sc, Sh = 1 + 0.045 * Cpm, 1 + 0.015 * Cpm * T
    Rt = -2 * np.sqrt(Cpm ** 7 / (Cpm ** 7 + 25.0 ** 7)) * np.sin(np.radians(60 * np.exp(-(((hm - 275) / 25) ** 2))))
    return np.sqrt((dLp / Sl) ** 2 + (dCp / Sc) ** 2 + (dHp / Sh) ** 2 + Rt * (dCp / Sc) * (dHp / Sh))


def rgb_to_lab(rgb):
    x = rgb.astype(np.float64) / 255.0
    x = np.where(x > 0.04045, ((x + 0.055) / 1.055) ** 2.4, x / 12.92)
    m = np.array([[0.412453, 0.357580, 0.180423], [0.212671, 0.715160, 0.072169], [0.019334, 0.119193, 0.950227]])
    xyz = x @ m.T / np.array([0.95047, 1.0, 1.08883])
    f = np.where(xyz > 0.008856, np.cbrt(xyz), 7.787 * xyz + 16.0 / 116.0)
    return np.stack([116 * f[..., 1] - 16, 500 * (f[..., 0] - f[..., 1]), 200 * (f[..., 1] - f[..., 2])], -1)


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("checkpoint", help="a DeOldify generator .pth (Learner.save layout {'model','opt'} or a bare state dict)")
    ap.add_argument("--arch", choices=["wide", "deep"], default=None, help="wide = video / stable (resnet101), deep = artistic (resnet34); default: from the file name")
    ap.add_argument("--render-factor", type=int, default=21)
    ap.add_argument("--frames", type=int, default=4, help="synthetic gray frames when no --image is given")
    ap.add_argument("--image", action="append", default=[], help="gray or colour images to colour instead (any size: squashed like the reference does)")
    args = ap.parse_args()
    os.environ["HAVC_RANGE_CHECK"] = "1"                      # before any context exists: nets are then created zero-filled
    from PIL import Image
    from vsdeoldify_amd import _native as nat
    from vsdeoldify_amd.clip import synthetic_gray_frame
    from vsdeoldify_amd.render import GeneratorRuntime, _load_pth, get_context
    arch = args.arch or ("deep" if "artistic" in os.path.basename(args.checkpoint).lower() else "wide")
    sd = _load_pth(args.checkpoint)
    S = args.render_factor * 16
    if args.image:
        frames = np.stack([np.asarray(Image.open(p).convert("RGB").resize((S, S), Image.BILINEAR)) for p in args.image])
    else:
        frames = np.stack([synthetic_gray_frame(i, S, S) for i in range(args.frames)])
    ctx = get_context(0)
    outs, report = {}, {"checkpoint": args.checkpoint, "arch": arch, "render_size": S, "frames": len(frames)}
    for mode in ("fast", "precise"):
        rt = GeneratorRuntime(ctx, sd, arch, precision=mode)
        try:
            net = rt.net(S, 1)
            out = np.empty_like(frames)
            try:
                for i in range(len(frames)):                  # post_process = 1: the image the reference's filter returns (deoldify/filters.py:100-110)
                    nat.check(ctx.lib.havc_deoldify_frames(ctx.h, net.h, None, 0.0, 1, nat.as_ptr(np.ascontiguousarray(frames[i:i + 1])),
                                                           nat.as_ptr(out[i:i + 1]), 1), ctx.h)
                amax, bad = net.range_stats()
                worst = int(np.argmax(amax))
                report[mode] = {"non_finite": int(bad.sum()), "largest_activation": float(amax.max()), "at_op": net.names[worst],
                                "fp16_headroom": float(65504.0 / max(float(amax.max()), 1e-9))}
            except nat.HavcRangeError as e:
                report[mode] = {"non_finite": -1, "error": str(e)}
            outs[mode] = out
        finally:
            rt.close()
    ok = report["fast"].get("non_finite") == 0 and report["precise"].get("non_finite") == 0
    if ok:
        de = delta_e00(rgb_to_lab(outs["fast"]), rgb_to_lab(outs["precise"]))
        report["fast_vs_precise"] = {"ciede2000_mean": round(float(de.mean()), 4), "ciede2000_p99": round(float(np.percentile(de, 99)), 4),
                                     "ciede2000_max": round(float(de.max()), 3), "pixels_with_dE_below_1": round(float((de < 1.0).mean()), 5)}
        ok = de.mean() < 0.35 and np.percentile(de, 99) < 2.6 and (de < 1.0).mean() >= 0.84
    report["verdict"] = "OK: the fast (fp16) path is within its stated tolerance of the fp32-class path for this checkpoint" if ok else \
        "USE precision='precise' (HAVC_PRECISION=precise) for this checkpoint: the fp16 path overflowed or left its tolerance"
    print(json.dumps(report, indent=1))
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
