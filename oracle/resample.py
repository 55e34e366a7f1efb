"""Spline64 polyphase resampler, numpy restatement — ORACLE / TEST INFRASTRUCTURE ONLY.

zimg's `resize.Spline64` (vsdeoldify/__init__.py:2504,3547) stays in the VapourSynth glue in production and
is OUTSIDE the parity contract.  The bench / clip harness needs *a* resampler on both sides
of the comparison; this is the CPU twin of the one in csrc/havc_runtime.cpp (get_resize_table) +
csrc/colorfilters.hip (resize_h_kernel / resize_v_kernel): same taps, horizontal pass first in float32,
round-half-up to uint8 after the vertical pass, edge replication.
"""
import numpy as np


def spline64(x):
    x = np.abs(x)
    out = np.zeros_like(x)
    m = x < 1
    out[m] = ((49.0 / 41.0 * x[m] - 6387.0 / 2911.0) * x[m] - 3.0 / 2911.0) * x[m] + 1.0
    m = (x >= 1) & (x < 2); t = x[m] - 1
    out[m] = ((-24.0 / 41.0 * t + 4032.0 / 2911.0) * t - 2328.0 / 2911.0) * t
    m = (x >= 2) & (x < 3); t = x[m] - 2
    out[m] = ((6.0 / 41.0 * t - 1008.0 / 2911.0) * t + 582.0 / 2911.0) * t
    m = (x >= 3) & (x < 4); t = x[m] - 3
    out[m] = ((-1.0 / 41.0 * t + 168.0 / 2911.0) * t - 97.0 / 2911.0) * t
    return out


def taps(src, dst):
    scale = dst / src
    fscale = min(scale, 1.0)
    support = 4.0 / fscale
    n = int(np.ceil(2.0 * support)) + 1
    center = (np.arange(dst) + 0.5) / scale - 0.5
    start = np.floor(center - support).astype(np.int64) + 1
    pos = start[:, None] + np.arange(n)[None, :]
    w = spline64((pos - center[:, None]) * fscale)
    w = (w / w.sum(1, keepdims=True)).astype(np.float32)
    return np.clip(pos, 0, src - 1), w


def resize_rgb8_float(img_u8, dw, dh):
    """uint8 [H,W,3] -> float32 [dh,dw,3] (before rounding)."""
    h, w, _ = img_u8.shape
    px, wx = taps(w, dw)
    py, wy = taps(h, dh)
    a = img_u8.astype(np.float32)
    tmp = np.zeros((h, dw, 3), np.float32)
    for t in range(px.shape[1]):                       # same accumulation order as the kernel
        tmp += wx[None, :, t, None] * a[:, px[:, t], :]
    out = np.zeros((dh, dw, 3), np.float32)
    for t in range(py.shape[1]):
        out += wy[:, t, None, None] * tmp[py[:, t], :, :]
    return out


def resize_rgb8(img_u8, dw, dh):
    return np.clip(np.floor(resize_rgb8_float(img_u8, dw, dh) + np.float32(0.5)), 0, 255).astype(np.uint8)
