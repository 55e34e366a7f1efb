"""CPU restatement of the reference's tweak / gray-pixel-restoration filters -- ORACLE ONLY.

Test infrastructure: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
Pinning: every Pillow operation used by the reference on this path (ImageEnhance.*, Image.point, HSV conversion,
Image.blend) is executed through Pillow itself, which is present in this image and on the GPU box; pil_rgb2hsv /
pil_hsv2rgb additionally restate Pillow's C conversion and are pinned exhaustively against Pillow
(tests/test_tweaks.py).  The cv2 calls (cvtColor RGB<->YUV, RGB<->HSV) go through oracle.cvcolor: parity UNPINNED
(cv2 is absent here), as stated in oracle/__init__.py.  Golden vectors produced by executing the reference's own
functions (with that same cv2 stand-in) are in tests/golden/tweaks.npz (tools/gen_golden.py).
"""
import numpy as np
from PIL import Image, ImageEnhance

from . import cvcolor
from .pipeline import chroma_stabilizer, get_image_luma, w_image_luma_merge


# ---- Pillow HSV conversion restated (libImaging/Convert.c rgb2hsv_row / hsv2rgb: float locals, double arithmetic) ----
def pil_rgb2hsv(rgb):
    a = np.asarray(rgb)
    r, g, b = (a[..., i].astype(np.int32) for i in range(3))
    maxc = np.maximum(np.maximum(r, g), b)
    minc = np.minimum(np.minimum(r, g), b)
    gray = maxc == minc
    cr = np.where(gray, 1, maxc - minc).astype(np.float32)
    mx = np.where(maxc == 0, 1, maxc).astype(np.float32)
    s = (cr / mx).astype(np.float32)
    rc = ((maxc - r).astype(np.float32) / cr).astype(np.float32)
    gc = ((maxc - g).astype(np.float32) / cr).astype(np.float32)
    bc = ((maxc - b).astype(np.float32) / cr).astype(np.float32)
    h = np.where(r == maxc, (bc - gc).astype(np.float32),
                 np.where(g == maxc, (2.0 + rc.astype(np.float64) - bc.astype(np.float64)).astype(np.float32),
                          (4.0 + gc.astype(np.float64) - rc.astype(np.float64)).astype(np.float32)))
    h = np.fmod(h.astype(np.float64) / 6.0 + 1.0, 1.0).astype(np.float32)
    uh = np.clip((h.astype(np.float64) * 255.0).astype(np.int32), 0, 255)
    us = np.clip((s.astype(np.float64) * 255.0).astype(np.int32), 0, 255)
    uh = np.where(gray, 0, uh)
    us = np.where(gray, 0, us)
    return np.stack([uh, us, maxc], -1).astype(np.uint8)


def pil_hsv2rgb(hsv):
    a = np.asarray(hsv)
    h, s, v = (a[..., i].astype(np.float32) for i in range(3))
    hd = h.astype(np.float64) * 6.0 / 255.0
    i = np.floor(hd).astype(np.int32)
    f = (hd - i.astype(np.float32).astype(np.float64)).astype(np.float32)
    fs = (s.astype(np.float64) / 255.0).astype(np.float32)
    vd, fsd, fd = v.astype(np.float64), fs.astype(np.float64), f.astype(np.float64)

    def c_round(x):                     # C round(): half away from zero
        return np.where(x >= 0, np.floor(x + 0.5), np.ceil(x - 0.5)).astype(np.int32)
    p = np.clip(c_round(vd * (1.0 - fsd)), 0, 255)
    q = np.clip(c_round(vd * (1.0 - fsd * fd)), 0, 255)
    t = np.clip(c_round(vd * (1.0 - fsd * (1.0 - fd))), 0, 255)
    vi = a[..., 2].astype(np.int32)
    k = i % 6
    r = np.choose(k, [vi, q, p, p, t, vi])
    g = np.choose(k, [t, vi, vi, q, p, p])
    b = np.choose(k, [p, p, t, vi, vi, q])
    gray = a[..., 1] == 0
    out = np.stack([np.where(gray, vi, r), np.where(gray, vi, g), np.where(gray, vi, b)], -1)
    return out.astype(np.uint8)


# ---- hue-range masks (vsslib/restcolor.py:417-470) ----------------------------------------------------------------
_HUE_NAMES = {"red": (0, 30), "orange": (30, 60), "yellow": (60, 90), "yellow-green": (90, 120), "green": (120, 150),
              "blue-green": (150, 180), "cyan": (180, 210), "blue": (210, 240), "blue-violet": (240, 270),
              "violet": (270, 300), "red-violet": (300, 330), "rose": (330, 360)}


def parse_hue_range(tok):
    """restcolor.py:436-470: a colour name or 'min:max' in degrees."""
    if tok in _HUE_NAMES:
        return _HUE_NAMES[tok]
    lo, hi = tok.split(":")
    return float(lo), float(hi)


def hue_ranges(spec):
    return [parse_hue_range(t) for t in spec.split(",")]


def hue_conditions(h_cv, spec):
    """restcolor.py:417-433 (strict inequalities on cv2's H = degrees / 2)."""
    cond = np.zeros(h_cv.shape, bool)
    for lo, hi in hue_ranges(spec):
        cond |= (h_cv > lo * 0.5) & (h_cv < hi * 0.5)
    return cond


def np_adjust_chroma2(np_color, np_gray, hue_range):
    """restcolor.py:353-376: pixels of np_color whose hue is inside the range are replaced by np_gray's."""
    if hue_range in ("none", ""):
        return np_gray
    cond = hue_conditions(cvcolor.rgb2hsv_u8(np_color)[:, :, 0], hue_range)
    return np.where(cond[..., None], np_gray, np_color).astype(np.uint8)


# ---- image_tweak (vsslib/imfilters.py:463-537) -----------------------------------------------------------------------
def gamma_lut(gamma):
    """imfilters.py:507-514."""
    inv = 1.0 / gamma
    return np.array([((i / 255.0) ** inv) * 255 for i in range(256)]).astype("uint8")


def hue_offset(hue_deg):
    """imfilters.py:530: PIL hue is 0..255 for 0..360 degrees."""
    return int((hue_deg / 360.0) * 255)


def image_tweak(img, sat=1.0, cont=1.0, bright=0.0, hue=0.0, gamma=1.0, hue_range="none"):
    """imfilters.py:463-504, executed through Pillow exactly like the reference."""
    pil = Image.fromarray(np.asarray(img))
    img_np = np.asarray(pil)
    if gamma != 1.0:
        # imfilters.py:507-517: the reference hands `uint8_array * 3` (an elementwise product, 256 entries) to Image.point
        # of a 3-band image, which Pillow rejects: gamma != 1 RAISES in the reference.  Kept as the reference behaves.
        pil = pil.point(gamma_lut(gamma) * 3)
    if hue != 0.0:
        h, s, v = pil.convert("HSV").split()
        h_np = (np.array(h, dtype=np.int16) + hue_offset(hue)) % 256
        pil = Image.merge("HSV", (Image.fromarray(h_np.astype("uint8"), mode="L"), s, v)).convert("RGB")
    if bright != 0.0:
        pil = ImageEnhance.Brightness(pil).enhance(1 + bright / 255)
    if cont != 1.0:
        pil = ImageEnhance.Contrast(pil).enhance(cont)
    if sat != 1.0:
        pil = ImageEnhance.Color(pil).enhance(sat)
    if hue_range in ("none", ""):
        return np.asarray(pil)
    return np_adjust_chroma2(img_np, np.asarray(pil), hue_range)


def luma_levels_lut(luma, luma_min=0.0, gamma=1.0, gamma_luma_min=0.0, gamma_alpha=0.0, gamma_min=0.2, i_min=0, i_max=255):
    """The Y -> Y' map of luma_adjusted_levels (imfilters.py:346-364) as a 256-entry table (uint8 wrap-around of
    np.add(y, i_alpha) included)."""
    y = np.arange(256, dtype=np.uint8)
    i_alpha = int(255 * (luma_min - luma)) if luma < luma_min else 0
    y_new = np.add(y, i_alpha).clip(i_min, i_max).astype(np.uint8) if i_alpha > 1 else y
    if gamma != 1 and luma < gamma_luma_min:
        g_new = max(gamma * pow(luma / gamma_luma_min, gamma_alpha), gamma_min) if gamma_alpha != 0 else gamma
        y_new = np.power(y_new / 255, 1 / g_new)
        y_new = np.multiply(y_new, 255).clip(i_min, i_max).astype(np.uint8)
    return y_new


def luma_adjusted_levels(img, luma_min=0.0, gamma=1.0, gamma_luma_min=0.0, gamma_alpha=0.0, gamma_min=0.2, i_min=0, i_max=255):
    """imfilters.py:335-372."""
    yuv = cvcolor.rgb2yuv_u8(np.asarray(img))
    luma = np.mean(yuv[:, :, 0]) / 255
    lut = luma_levels_lut(luma, luma_min, gamma, gamma_luma_min, gamma_alpha, gamma_min, i_min, i_max)
    out = yuv.copy()
    out[:, :, 0] = lut[yuv[:, :, 0]]
    return cvcolor.yuv2rgb_u8(out)


# ---- ConstrainedChromaMerge with the dark-frame red fix (vsslib/mcomb.py:333-361) -----------------------------------
def constrained_chroma_merge(img1, img2, level=0.2, weight=0.5, red_fix=True):
    st = chroma_stabilizer(img1, img2, level, weight)
    if not red_fix:
        return st
    luma = get_image_luma(st, 255)
    if luma > 0.3:
        return st
    if luma > 0.2:
        return w_image_luma_merge(image_tweak(st, sat=0.9, hue_range="280:360,0:30"), st, 0.2, 0.3)
    if luma > 0.1:
        return w_image_luma_merge(image_tweak(st, sat=0.8, hue_range="280:360,0:30"), st, 0.1, 0.2)
    return image_tweak(st, sat=0.7)


def _red_fix(st):
    luma = get_image_luma(st, 255)
    if luma > 0.3:
        return st
    if luma > 0.2:
        return w_image_luma_merge(image_tweak(st, sat=0.9, hue_range="280:360,0:30"), st, 0.2, 0.3)
    if luma > 0.1:
        return w_image_luma_merge(image_tweak(st, sat=0.8, hue_range="280:360,0:30"), st, 0.1, 0.2)
    return image_tweak(st, sat=0.7)


def chroma_bound_adaptive_merge(img1, img2, base_tol=14, max_extra=18, weight=0.5, red_fix=True):
    """ChromaBoundAdaptiveMerge's merge_frame (vsslib/mcomb.py:370-437): adaptive chroma limiter + the same dark-frame red fix."""
    from .pipeline import chroma_stabilizer_adaptive
    st = chroma_stabilizer_adaptive(img1, img2, base_tol, max_extra, weight)
    return _red_fix(st) if red_fix else st


# ---- ChromaRetentionMerge core: restore_color_gradient (vsslib/restcolor.py:98-217) ---------------------------------
def gradient_mask(sat_u8, tht=15, alpha=2.0, algo=0):
    """restcolor.py:137-217."""
    if algo == 0:
        s = sat_u8.clip(0, 255)
        grad = np.where(s < tht, 2.0 * s / alpha - tht, 2.0 * (s - tht) * alpha)
        return (255.0 - tht - grad).clip(0, 255).astype(int)
    s = sat_u8.astype(np.float32)
    tht = int(np.clip(tht, 0, 255))
    if tht == 0:
        return np.zeros_like(sat_u8, dtype=np.uint8)
    if algo == 1:
        max_s = min(2 * tht, 200)
        mask_norm = (1.0 - (np.clip(s, 0, max_s) / max_s)) ** alpha
    else:
        s_rel = np.clip(s / tht, 0, 2)
        mask_norm = np.exp(-alpha * s_rel * np.log(2))
        mask_norm = np.where(s >= 2 * tht, 0.0, mask_norm)
    return (np.clip(mask_norm * 255, 0, 255)).astype(np.uint8)


def restore_color_gradient(img_color, img_gray, sat=1.0, tht=50, weight=0.0, alpha=2.0, return_mask=False, algo=0):
    """restcolor.py:98-134: gray pixels of img_gray (low HSV saturation) take the colours of img_color."""
    np_color, np_gray = np.asarray(img_color), np.asarray(img_gray)
    hsv_color = cvcolor.rgb2hsv_u8(np_color)
    hsv_gray = cvcolor.rgb2hsv_u8(np_gray)
    if sat != 1.0:
        hsv_color[:, :, 1] = hsv_color[:, :, 1] * min(max(sat, 0), 10)          # float -> uint8 cast (wraps above 255)
    np_color_sat = cvcolor.hsv2rgb_u8(hsv_color)
    mask = gradient_mask(hsv_gray[:, :, 1], tht, alpha, algo)
    mask_rgb = np_gray.copy()
    for i in range(3):
        mask_rgb[:, :, i] = mask
    if return_mask:
        return mask_rgb
    mw = (mask_rgb / 255).astype(float)
    res = (np.multiply(np_gray, 1 - mw) + np.multiply(np_color_sat, mw)).clip(0, 255).astype(np.uint8)
    if weight > 0:
        res = (np.multiply(res, 1 - weight) + np.multiply(np_color_sat, weight)).clip(0, 255).astype(np.uint8)
    if weight < 0:
        res = (np.multiply(res, 1 + weight) + np.multiply(np_gray, -weight)).clip(0, 255).astype(np.uint8)
    return res


# ---- image_chroma_tweak (vsslib/imfilters.py:540-548 -> restcolor.py:288-350) and the HAVC_stabilizer frame bodies ------
def np_hue_add(h, hue):
    """nputils.py:330-340 (the float result is cast back to uint8 by the caller's slice assignment)."""
    if hue == 0:
        return h
    hue_half = 0.5 * min(max(int(hue), -360), 360)
    h = h + hue_half
    h = np.where(h > 180, h - 180, h)
    return np.where(h < 0, h + 180, h)


def parse_hue_adjust(hue_adjust):
    """restcolor.py:379-414 -> (hue_range, sat, hue, weight) or None."""
    p = hue_adjust.split("|")
    sat, hue, weight = 1.0, 0, 0
    if len(p) < 1 or len(p) > 2:
        return None
    if len(p) == 1:
        return p[0], sat, hue, weight
    sw = p[1].split(",")

    def isfloat(t):
        try:
            float(t)
            return True
        except ValueError:
            return False
    if len(sw) != 2 or not isfloat(sw[0]) or not isfloat(sw[1]):
        return None
    if sw[0][0] in ("-", "+"):
        hue = int(sw[0])
    else:
        sat = float(sw[0])
    if sat > 10:
        hue, sat = int(sat), 1.0
    return p[0], sat, hue, float(sw[1])


def _wmerge(a, b, w):
    return (np.multiply(a, 1 - w) + np.multiply(b, w)).clip(0, 255).astype(np.uint8)


def np_image_chroma_tweak(img, sat=1, bright=0, hue=0, hue_adjust="none"):
    """restcolor.py:288-350."""
    img = np.asarray(img)
    if sat == 1 and bright == 0 and hue == 0 and hue_adjust == "none":
        return img
    hsv = cvcolor.rgb2hsv_u8(img)
    hsv[:, :, 0] = np_hue_add(hsv[:, :, 0], hue)
    hsv[:, :, 1] = hsv[:, :, 1] * min(max(sat, 0), 10)
    hsv[:, :, 2] = hsv[:, :, 2] * min(max(1 + bright, 0), 10)
    color = cvcolor.hsv2rgb_u8(hsv)
    if hue_adjust in ("none", ""):
        return color
    param = parse_hue_adjust(hue_adjust)
    if param is None:
        return color
    hue_range, sat2, hue2, weight = param
    g = cvcolor.rgb2hsv_u8(color)
    if hue2 != 0:
        g[:, :, 0] = np_hue_add(g[:, :, 0], hue2)
    if sat2 != 1:
        g[:, :, 1] = g[:, :, 1] * min(max(sat2, 0), 10)
    gray_rgb = cvcolor.hsv2rgb_u8(g)
    cond = hue_conditions(hsv[:, :, 0], hue_range)
    restored = np.where(cond[..., None], gray_rgb, img).astype(np.uint8)
    if weight > 0:
        restored = _wmerge(restored, gray_rgb if hue2 == 0 else img, weight)
    if weight < 0:
        restored = _wmerge(restored, img, -weight)
    return restored


def adjust_hue_range(img, hue_adjust="none"):
    """restcolor.py:221-286 (adjust_hue_range -> adjust_chroma): like the adjust stage of np_image_chroma_tweak, but computed from the
    image itself (no identity HSV round trip in front).  vs_sc_ddcolor applies it to EVERY DDColor frame when scene detection is off
    (vsslib/vsmodels.py:365-366 with HAVC_colorizer's default ddtweak_p[1] = "300:360|0.8,0.1", vsfilters.py:435-455)."""
    img = np.asarray(img)
    if hue_adjust in ("none", ""):
        return img
    param = parse_hue_adjust(hue_adjust)
    if param is None:
        return img
    hue_range, sat, hue, weight = param
    if hue_range in ("none", ""):
        return img
    hsv = cvcolor.rgb2hsv_u8(img)
    g = hsv.copy()
    if hue != 0:
        g[:, :, 0] = np_hue_add(g[:, :, 0], hue)
    if sat != 1:
        g[:, :, 1] = g[:, :, 1] * min(max(sat, 0), 10)
    gray_rgb = cvcolor.hsv2rgb_u8(g)
    cond = hue_conditions(hsv[:, :, 0], hue_range)
    restored = np.where(cond[..., None], gray_rgb, img).astype(np.uint8)
    if weight > 0:
        restored = _wmerge(restored, gray_rgb if hue == 0 else img, weight)
    if weight < 0:
        restored = _wmerge(restored, img, -weight)
    return restored


def _luma_merge(img2, img1, lo, hi):
    from .pipeline import image_luma_merge
    return image_luma_merge(img2, img1, lo) if lo == hi else w_image_luma_merge(img2, img1, lo, hi)


def dark_tweak_frame(img, dark_threshold=0.3, dark_amount=0.8, dark_hue_adjust="none"):
    """vs_sc_dark_tweak's merge_frame (vsfilters.py:600-632)."""
    white = min(max(dark_threshold, 0.1), 0.50)
    d_sat = min(max(1.1 - dark_amount, 0.10), 0.80)
    d_bright = -min(max(dark_amount, 0.20), 0.90)
    return _luma_merge(image_tweak(img, bright=d_bright, sat=d_sat, hue_range=dark_hue_adjust), np.asarray(img), 0.1, white)


def chroma_bright_tweak_frame(img, black_threshold=0.3, white_threshold=0.6, dark_sat=0.8, dark_bright=-0.10, chroma_adjust="none"):
    """vs_sc_chroma_bright_tweak's merge_frame (vsfilters.py:525-547)."""
    return _luma_merge(np_image_chroma_tweak(img, bright=dark_bright, sat=dark_sat, hue_adjust=chroma_adjust), np.asarray(img),
                       black_threshold, white_threshold)


def colormap_frame(img, colormap="none"):
    """_vs_sc_colormap's merge_frame (vsfilters.py:575-590)."""
    return np_image_chroma_tweak(img, hue_adjust=colormap)
