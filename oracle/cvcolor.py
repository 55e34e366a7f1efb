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


_HSV_SHIFT = 12
_SDIV = np.array([0] + [int(round((255 << _HSV_SHIFT) / i)) for i in range(1, 256)], np.int64)
_HDIV180 = np.array([0] + [int(round((180 << _HSV_SHIFT) / (6.0 * i))) for i in range(1, 256)], np.int64)


def rgb2hsv_u8(rgb):
    """cv2 8-bit RGB->HSV (H in [0,180), S,V in [0,255]): OpenCV's integer path (color_hsv: RGB2HSV_b) -- 12-bit
    reciprocal tables sdiv = round(255<<12 / v), hdiv = round(180<<12 / (6 diff)), h sextant from the max channel,
    descale by (x + 2048) >> 12, h += 180 when negative.  Restated from memory of the published source: unpinned."""
    a = np.asarray(rgb).astype(np.int64)
    r, g, b = a[..., 0], a[..., 1], a[..., 2]
    v = np.maximum(np.maximum(r, g), b)
    vmin = np.minimum(np.minimum(r, g), b)
    diff = v - vmin
    s = (diff * _SDIV[v] + (1 << (_HSV_SHIFT - 1))) >> _HSV_SHIFT
    h = np.where(v == r, g - b, np.where(v == g, b - r + 2 * diff, r - g + 4 * diff))
    h = (h * _HDIV180[diff] + (1 << (_HSV_SHIFT - 1))) >> _HSV_SHIFT
    h = np.where(h < 0, h + 180, h)
    return np.stack([h, s, v], -1).astype(np.uint8)


def hsv2rgb_u8(hsv):
    """cv2 8-bit HSV->RGB (HSV2RGB_b -> HSV2RGB_f, hrange 180), float32 throughout: h = H * (6/180), s = S * (1/255),
    v = V * (1/255) (multiplications by float32 reciprocals), sector = floor(h) (h wrapped into [0, 6)), tab = {v, v(1-s),
    v(1-s h), v(1-s(1-h))}, output saturate_cast<uchar>(x * 255) = round-half-even.  Restated from memory: unpinned."""
    f32 = np.float32
    a = np.asarray(hsv).astype(f32)
    h = a[..., 0] * f32(f32(6.0) / f32(180.0))
    s = a[..., 1] * f32(f32(1.0) / f32(255.0))
    v = a[..., 2] * f32(f32(1.0) / f32(255.0))
    h = np.where(h >= f32(6.0), h - f32(6.0), h)
    i = np.floor(h).astype(np.int32)
    f = (h - i.astype(f32)).astype(f32)
    one = f32(1.0)
    p, q, t = v * (one - s), v * (one - s * f), v * (one - s * (one - f))
    r = np.choose(i, [v, q, p, p, t, v])
    g = np.choose(i, [t, v, v, q, p, p])
    b = np.choose(i, [p, p, t, v, v, q])
    out = np.stack([r, g, b], -1).astype(f32) * f32(255.0)
    gray = (a[..., 1] == 0)[..., None]
    out = np.where(gray, (v * f32(255.0))[..., None], out)
    return np.clip(np.rint(out), 0, 255).astype(np.uint8)


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
