"""Pillow ImagingResample (8 bits per channel), restated in numpy — ORACLE / TEST INFRASTRUCTURE ONLY.

Reference call sites: `img.resize((S,S), BILINEAR)` (vsdeoldify/deoldify/filters.py:37-41,70-73) and
`Image.resize((256,256), resample=3)` (vsdeoldify/colorization/colorizers/util.py:21-22).
Follows Pillow's src/libImaging/Resample.c: precompute_coeffs (float64), normalize_coeffs_8bpc (22-bit fixed point,
PRECISION_BITS = 32 - 8 - 2), horizontal pass then vertical pass, each rounded and clipped to uint8.
Pinned bit-exactly against Pillow itself (tests/test_oracle_golden.py::test_pil_resize_restatement).
"""
import math

import numpy as np

PRECISION_BITS = 32 - 8 - 2
BILINEAR, BICUBIC = 2, 3


def _bilinear(x):
    x = abs(x)
    return 1.0 - x if x < 1.0 else 0.0


def _bicubic(x, a=-0.5):
    x = abs(x)
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


FILTERS = {BILINEAR: (_bilinear, 1.0), BICUBIC: (_bicubic, 2.0)}


def coeffs(in_size, out_size, resample):
    """-> (xmin[out], xmax[out], k[out][ksize] int32 fixed point)."""
    filt, fsupport = FILTERS[resample]
    scale = filterscale = in_size / out_size
    if filterscale < 1.0:
        filterscale = 1.0
    support = fsupport * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    xmins = np.zeros(out_size, np.int32)
    xmaxs = np.zeros(out_size, np.int32)
    kk = np.zeros((out_size, ksize), np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = int(center - support + 0.5)
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        w = [filt((x + xmin - center + 0.5) * ss) for x in range(xmax)]
        ww = 0.0
        for v in w:
            ww += v
        for x in range(xmax):
            v = w[x] / ww if ww != 0.0 else w[x]
            kk[xx, x] = int(-0.5 + v * (1 << PRECISION_BITS)) if v < 0 else int(0.5 + v * (1 << PRECISION_BITS))
        xmins[xx], xmaxs[xx] = xmin, xmax
    return xmins, xmaxs, kk


def _pass(a, out_size, resample, axis):
    a = np.moveaxis(a, axis, 0).astype(np.int64)
    xmins, xmaxs, kk = coeffs(a.shape[0], out_size, resample)
    out = np.empty((out_size,) + a.shape[1:], np.uint8)
    for xx in range(out_size):
        acc = np.full(a.shape[1:], 1 << (PRECISION_BITS - 1), np.int64)
        for x in range(xmaxs[xx]):
            acc += a[xmins[xx] + x] * int(kk[xx, x])
        out[xx] = np.clip(acc >> PRECISION_BITS, 0, 255).astype(np.uint8)
    return np.moveaxis(out, 0, axis)


def resize(img_u8, out_w, out_h, resample):
    """uint8 [H,W,C] -> uint8 [out_h,out_w,C]; identity copy when the size is unchanged (as Pillow does)."""
    a = np.asarray(img_u8)
    if a.shape[1] != out_w:
        a = _pass(a, out_w, resample, 1)
    if a.shape[0] != out_h:
        a = _pass(a, out_h, resample, 0)
    return a
