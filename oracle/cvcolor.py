"""OpenCV 8-bit colour conversions, restated in numpy — ORACLE / TEST INFRASTRUCTURE ONLY.

PARITY UNPINNED: cv2 is absent from the build container, so these follow OpenCV's published
integer algorithm (modules/imgproc/src/color_yuv.simd.hpp, RGB2YCrCb_i<uchar> /
YCrCb2RGB_i<uchar> with isCrCb=false; yuv_shift = 14, CV_DESCALE(x,n) = (x + (1<<(n-1))) >> n).
Reference call sites: /root/reference/vsdeoldify/deoldify/filters.py:100-110,
/root/reference/vsdeoldify/vsslib/imfilters.py:166-199,312-321.
"""
import numpy as np

COLOR_RGB2YUV, COLOR_YUV2RGB, COLOR_RGB2HSV, COLOR_HSV2RGB = 83, 85, 41, 55
CV_32F, CV_64F = 5, 6

YUV_SHIFT = 14
R2Y, G2Y, B2Y = 4899, 9617, 1868          # 0.299, 0.587, 0.114 in Q14
R2VI, B2UI = 14369, 8061                  # 0.877, 0.492 in Q14
U2BI, U2GI, V2GI, V2RI = 33292, -6472, -9519, 18678   # 2.032, -0.395, -0.581, 1.140 in Q14


def _descale(x):
    return (x + (1 << (YUV_SHIFT - 1))) >> YUV_SHIFT


def rgb2yuv_u8(rgb):
    a = np.asarray(rgb).astype(np.int32)
    r, g, b = a[..., 0], a[..., 1], a[..., 2]
    y = _descale(r * R2Y + g * G2Y + b * B2Y)
    delta = 128 << YUV_SHIFT
    u = _descale((b - y) * B2UI + delta)
    v = _descale((r - y) * R2VI + delta)
    return np.clip(np.stack([y, u, v], -1), 0, 255).astype(np.uint8)


def yuv2rgb_u8(yuv):
    a = np.asarray(yuv).astype(np.int32)
    y, u, v = a[..., 0], a[..., 1] - 128, a[..., 2] - 128
    b = y + _descale(u * U2BI)
    g = y + _descale(u * U2GI + v * V2GI)
    r = y + _descale(v * V2RI)
    return np.clip(np.stack([r, g, b], -1), 0, 255).astype(np.uint8)


def rgb2hsv_u8(rgb):
    """cv2 8-bit RGB->HSV (H in [0,180), S,V in [0,255]); float formula + rounding (unpinned)."""
    a = np.asarray(rgb).astype(np.float32)
    r, g, b = a[..., 0], a[..., 1], a[..., 2]
    v = np.max(a, -1)
    mn = np.min(a, -1)
    d = v - mn
    s = np.where(v > 0, d / np.where(v > 0, v, 1) * 255.0, 0)
    dd = np.where(d > 0, d, 1)
    h = np.where(v == r, (g - b) / dd, np.where(v == g, 2 + (b - r) / dd, 4 + (r - g) / dd)) * 60.0
    h = np.where(d > 0, h, 0)
    h = np.where(h < 0, h + 360, h) / 2.0
    out = np.stack([np.rint(h) % 180, np.rint(s), v], -1)
    return np.clip(out, 0, 255).astype(np.uint8)


def hsv2rgb_u8(hsv):
    a = np.asarray(hsv).astype(np.float32)
    h, s, v = a[..., 0] * 2.0, a[..., 1] / 255.0, a[..., 2] / 255.0
    hh = (h / 60.0) % 6
    i = np.floor(hh).astype(np.int32)
    f = hh - i
    p, q, t = v * (1 - s), v * (1 - s * f), v * (1 - s * (1 - f))
    r = np.choose(i, [v, q, p, p, t, v])
    g = np.choose(i, [t, v, v, q, p, p])
    b = np.choose(i, [p, p, t, v, v, q])
    return np.clip(np.rint(np.stack([r, g, b], -1) * 255.0), 0, 255).astype(np.uint8)


def cvtColor(src, code):
    if code == COLOR_RGB2YUV:
        return rgb2yuv_u8(src)
    if code == COLOR_YUV2RGB:
        return yuv2rgb_u8(src)
    if code == COLOR_RGB2HSV:
        return rgb2hsv_u8(src)
    if code == COLOR_HSV2RGB:
        return hsv2rgb_u8(src)
    raise NotImplementedError(code)


def Laplacian(src, ddepth, ksize=1):
    """3x3 [[0,1,0],[1,-4,1],[0,1,0]] with BORDER_REFLECT_101 (cv2 default aperture 1)."""
    a = np.asarray(src).astype(np.float32 if ddepth == CV_32F else np.float64)
    p = np.pad(a, 1, mode="reflect")
    return p[:-2, 1:-1] + p[2:, 1:-1] + p[1:-1, :-2] + p[1:-1, 2:] - 4 * a
