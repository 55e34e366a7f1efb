"""CPU: bound the UNPINNED integer restatements of cv2's colour conversions (oracle/cvcolor.py; cv2 is absent from the build container) by sweeping
ALL inputs against the float definitions OpenCV documents for the same codes.  Every byte of the headline output passes through
cvtColor(RGB2YUV / YUV2RGB) twice (deoldify/filters.py:100-110 at the net size, vsslib/vsfilters.py:863-899 at 1080p); the HSV pair carries
image_chroma_tweak / adjust_hue_range / restore_color_gradient (vsslib/restcolor.py:98-350).  The library's kernels are bit-exact against
oracle/cvcolor.py (tests/test_filters2.py, test_tweaks.py), so a bound on cvcolor-vs-definition is a bound on how far library AND oracle could be
from a real cv2 build: the fixed-point tables can differ from the real-valued formula by rounding only.
Float definitions (OpenCV docs, "Color conversions"):  Y = 0.299 R + 0.587 G + 0.114 B, U = 0.492 (B - Y) + 128, V = 0.877 (R - Y) + 128;
R = Y + 1.140 V', G = Y - 0.395 U' - 0.581 V', B = Y + 2.032 U';  V = max, S = 255 (V - min) / V, H = 30 (G - B) / (V - min) [+60 / +120 per sextant],
H += 180 if negative."""
import numpy as np

from oracle import cvcolor


def _all_triples_by_first(k):
    b, c = np.meshgrid(np.arange(256, dtype=np.int32), np.arange(256, dtype=np.int32), indexing="ij")
    return np.stack([np.full_like(b, k), b, c], -1).reshape(-1, 3)


def _round_clip(x):
    return np.clip(np.floor(x + 0.5), 0, 255).astype(np.int32)


def test_rgb2yuv_and_yuv2rgb_fixed_point_within_one_lsb_of_the_float_definition_over_all_inputs():
    mx_f, ne_f, mx_i, ne_i, n = np.zeros(3, int), np.zeros(3, int), np.zeros(3, int), np.zeros(3, int), 0
    mx_q, ne_q = np.zeros(3, int), np.zeros(3, int)
    rt_mx, rt_ne, rt_n = 0, 0, 0
    for k in range(256):
        t = _all_triples_by_first(k)
        f = t.astype(np.float64)
        # forward
        y = 0.299 * f[:, 0] + 0.587 * f[:, 1] + 0.114 * f[:, 2]
        ref = np.stack([_round_clip(y), _round_clip(0.492 * (f[:, 2] - y) + 128), _round_clip(0.877 * (f[:, 0] - y) + 128)], -1)
        d = np.abs(cvcolor.rgb2yuv_u8(t).astype(np.int32) - ref)
        mx_f = np.maximum(mx_f, d.max(0)); ne_f += (d != 0).sum(0)
        # the same definition evaluated the way an 8-bit pipeline must: U, V from the ROUNDED Y (OpenCV's integer path subtracts the stored Y)
        yq = _round_clip(y).astype(np.float64)
        ref = np.stack([_round_clip(y), _round_clip(0.492 * (f[:, 2] - yq) + 128), _round_clip(0.877 * (f[:, 0] - yq) + 128)], -1)
        d = np.abs(cvcolor.rgb2yuv_u8(t).astype(np.int32) - ref)
        mx_q = np.maximum(mx_q, d.max(0)); ne_q += (d != 0).sum(0)
        # inverse (t read as a YUV triple)
        u, v = f[:, 1] - 128, f[:, 2] - 128
        ref = np.stack([_round_clip(f[:, 0] + 1.140 * v), _round_clip(f[:, 0] - 0.395 * u - 0.581 * v), _round_clip(f[:, 0] + 2.032 * u)], -1)
        d = np.abs(cvcolor.yuv2rgb_u8(t).astype(np.int32) - ref)
        mx_i = np.maximum(mx_i, d.max(0)); ne_i += (d != 0).sum(0)
        # round trip RGB -> YUV -> RGB through the fixed-point pair (what _post_process does to the luma source when U, V come from the same image)
        yuv = cvcolor.rgb2yuv_u8(t)
        ok = ((yuv[:, 1:] > 0) & (yuv[:, 1:] < 255)).all(1)               # saturated U / V (pure reds / blues: 0.877 * 179 + 128 > 255) clip like cv2's
        d = np.abs(cvcolor.yuv2rgb_u8(yuv).astype(np.int32) - t)[ok]
        rt_mx = max(rt_mx, int(d.max())); rt_ne += int((d != 0).sum()); rt_n += int(ok.sum())
        n += len(t)
    print(f"RGB2YUV vs float definition: max |d| Y/U/V {mx_f.tolist()}, fraction != 0 {(ne_f / n).round(6).tolist()}")
    print(f"RGB2YUV vs float definition with U, V from the rounded Y: max |d| {mx_q.tolist()}, fraction != 0 {(ne_q / n).round(6).tolist()}")
    print(f"YUV2RGB vs float definition: max |d| R/G/B {mx_i.tolist()}, fraction != 0 {(ne_i / n).round(6).tolist()}")
    print(f"RGB -> YUV -> RGB round trip (U, V unsaturated: {rt_n / n:.4f} of the inputs): max |d| {rt_mx}, bytes changed {rt_ne / (3 * rt_n):.5f}")
    assert n == 1 << 24 and mx_f.max() <= 1 and mx_i.max() <= 1 and mx_q.max() <= 1
    # Against the real-valued formula U and V differ by one step on 12 - 22 % of the inputs -- that is the rounding of Y BEFORE the subtraction
    # (any 8-bit implementation has it); with Y rounded first only the Q14 constants' own rounding is left: .5-boundary cases.
    assert (ne_f / n).max() < 0.25 and (ne_q / n).max() < 0.01 and (ne_i / n).max() < 0.01
    assert rt_mx <= 2                                                    # inherent to 8-bit YUV (both a cv2 build and the float formula lose the same bits)


def test_rgb2hsv_and_hsv2rgb_within_one_step_of_the_float_definition_over_all_inputs():
    mx, ne, n = np.zeros(3, int), np.zeros(3, int), 0
    for k in range(256):
        t = _all_triples_by_first(k)
        f = t.astype(np.float64)
        r, g, b = f[:, 0], f[:, 1], f[:, 2]
        v = f.max(1)
        diff = v - f.min(1)
        s = np.where(v > 0, 255.0 * diff / np.maximum(v, 1), 0.0)
        dd = np.maximum(diff, 1e-30)
        h = np.where(v == r, (g - b) / dd, np.where(v == g, 2.0 + (b - r) / dd, 4.0 + (r - g) / dd)) * 30.0
        h = np.where(diff == 0, 0.0, h)
        h = np.where(h < 0, h + 180.0, h)
        got = cvcolor.rgb2hsv_u8(t).astype(np.int32)
        dh = np.abs(got[:, 0] - np.floor(h + 0.5).astype(np.int32) % 180)
        dh = np.minimum(dh, 180 - dh)                                    # hue is circular: 179.6 rounds to 180 == 0
        d = np.stack([dh, np.abs(got[:, 1] - _round_clip(s)), np.abs(got[:, 2] - v.astype(np.int32))], -1)
        mx = np.maximum(mx, d.max(0)); ne += (d != 0).sum(0); n += len(t)
    print(f"RGB2HSV vs float definition: max |d| H/S/V {mx.tolist()}, fraction != 0 {(ne / n).round(6).tolist()}")
    assert n == 1 << 24 and mx[0] <= 1 and mx[1] <= 1 and mx[2] == 0
    # inverse: all H in [0, 180), S, V in [0, 255]
    mx2, ne2, n2 = 0, 0, 0
    S, V = np.meshgrid(np.arange(256, dtype=np.float64), np.arange(256, dtype=np.float64), indexing="ij")
    S, V = S.reshape(-1), V.reshape(-1)
    for H in range(180):
        h6 = H / 30.0
        i = int(np.floor(h6)) % 6
        fr = h6 - np.floor(h6)
        s, v = S / 255.0, V / 255.0
        p, q, t_ = v * (1 - s), v * (1 - s * fr), v * (1 - s * (1 - fr))
        rgb = [(v, t_, p), (q, v, p), (p, v, t_), (p, q, v), (t_, p, v), (v, p, q)][i]
        ref = np.stack([np.clip(np.rint(c * 255.0), 0, 255) for c in rgb], -1).astype(np.int32)
        hsv = np.stack([np.full_like(S, H), S, V], -1).astype(np.uint8)
        d = np.abs(cvcolor.hsv2rgb_u8(hsv).astype(np.int32) - ref)
        mx2 = max(mx2, int(d.max())); ne2 += int((d != 0).sum()); n2 += d.size
    print(f"HSV2RGB vs float definition: max |d| {mx2}, bytes != {ne2 / n2:.6f}")
    assert mx2 <= 1 and ne2 / n2 < 0.01
