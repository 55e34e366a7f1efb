"""Host-side mirror of the per-frame pixel filters the merge path uses (vsdeoldify/vsslib/imfilters.py),
backed by the HIP kernels in csrc/colorfilters.hip.  PIL.Image in / PIL.Image out like the reference;
the *_np variants take/return uint8 HWC arrays -- or `device.DeviceImage`s: when any operand lives in HBM the others are
uploaded once, the result stays in HBM and the call does not block (include/havc_mi355.h, "Pointers").
"""
import numpy as np

from . import _native as nat
from .device import DeviceImage, is_device, operand_ptr as _p
from .render import get_context


def _one(ctx, a):
    """operand + matching output buffer: ndarray -> ndarray, DeviceImage -> DeviceImage"""
    if is_device(a):
        return a, DeviceImage(ctx or a.ctx, a.shape)           # the output lives in the pool of the context that RUNS the filter (device.py)
    a = np.ascontiguousarray(a, dtype=np.uint8)
    if a.ndim != 3 or a.shape[2] != 3:
        raise ValueError("not an RGB image")
    return a, np.empty_like(a)


def _prep(a, b, ctx=None):
    if is_device(a) or is_device(b):
        ctx = ctx or (a.ctx if is_device(a) else b.ctx)
        a = a if is_device(a) else DeviceImage.from_numpy(ctx, a)
        b = b if is_device(b) else DeviceImage.from_numpy(ctx, b)
        if a.shape != b.shape:
            raise ValueError("images do not match")
        return a, b, DeviceImage(ctx, a.shape)
    a = np.ascontiguousarray(a, dtype=np.uint8)
    b = np.ascontiguousarray(b, dtype=np.uint8)
    if a.shape != b.shape or a.ndim != 3 or a.shape[2] != 3:
        raise ValueError("images do not match")          # PIL raises ValueError("images do not match")
    return a, b, np.empty_like(a)


def _wh(a, per_frame=False):
    """(width, height) handed to the C entry point; a DeviceImage stack is ONE tall image for purely per-pixel filters"""
    if a.ndim == 4:
        if per_frame:
            raise ValueError("this filter works on one frame at a time (frame-level statistics / neighbourhoods)")
        return a.shape[2], a.shape[0] * a.shape[1]
    return a.shape[1], a.shape[0]


def blend_np(ctx, a, b, w):
    a, b, out = _prep(a, b)
    nat.check(ctx.lib.havc_blend(ctx.h, _p(a), _p(b), float(w), _p(out), *_wh(a)), ctx.h)
    return out


def chroma_post_process_np(ctx, color, orig):
    a, b, out = _prep(color, orig)
    nat.check(ctx.lib.havc_chroma_post_process(ctx.h, _p(a), _p(b), _p(out), *_wh(a)), ctx.h)
    return out


def chroma_stabilizer_np(ctx, img_stable, img_new, alpha=0.15, weight=1.0):
    a, b, out = _prep(img_stable, img_new)
    nat.check(ctx.lib.havc_chroma_stabilizer(ctx.h, _p(a), _p(b), float(alpha), float(weight), _p(out),
                                             *_wh(a)), ctx.h)
    return out


# ---- PIL-facing functions with the reference's names and argument meaning -------------------------
def image_weighted_merge(img1, img2, weight=0.5, device_index=0):
    """imfilters.py:113-124: weight 0 -> img1, 1 -> img2, else Image.blend(img1, img2, weight)."""
    from PIL import Image
    if weight == 0.0:
        return img1
    if weight == 1.0:
        return img2
    return Image.fromarray(blend_np(get_context(device_index), np.asarray(img1), np.asarray(img2), weight))


def chroma_post_process(img_m, orig, device_index=0):
    """imfilters.py:312-321: chroma (U,V) of img_m on the luma of orig."""
    from PIL import Image
    return Image.fromarray(chroma_post_process_np(get_context(device_index), np.asarray(img_m), np.asarray(orig)))


def chroma_stabilizer(img_stable, img_new, alpha=0.15, weight=1.0, device_index=0):
    """imfilters.py:160-200."""
    from PIL import Image
    return Image.fromarray(chroma_stabilizer_np(get_context(device_index), np.asarray(img_stable), np.asarray(img_new),
                                                alpha, weight))


# ---- merge-method filters of HAVC_merge methods 4, 5, 7 and the temporal limiters --------------------------------
def chroma_stabilizer_adaptive_np(ctx, img_stable, img_new, base_tol=18, max_extra=22, weight=1.0):
    a, b, out = _prep(img_stable, img_new)
    nat.check(ctx.lib.havc_chroma_stabilizer_adaptive(ctx.h, _p(a), _p(b), float(base_tol), float(max_extra),
                                                      float(weight), _p(out), *_wh(a, per_frame=True)), ctx.h)
    return out


def chroma_temporal_limiter_np(ctx, cur, prv, alpha=0.05):
    a, b, out = _prep(cur, prv)
    nat.check(ctx.lib.havc_chroma_temporal_limiter(ctx.h, _p(a), _p(b), float(alpha), _p(out), *_wh(a)), ctx.h)
    return out


def color_temporal_stabilizer_np(ctx, frames, weight_list):
    import ctypes as C
    dev = any(is_device(f) for f in frames)
    frames = [f if is_device(f) else (DeviceImage.from_numpy(ctx, f) if dev else np.ascontiguousarray(f, dtype=np.uint8)) for f in frames]
    if len(frames) != len(weight_list) or not 1 <= len(frames) <= 9 or any(f.shape != frames[0].shape for f in frames):
        raise ValueError("frames / weight_list mismatch (1..9 frames of one size)")
    ptrs = (C.c_void_p * len(frames))(*[(f.ptr.value if is_device(f) else f.ctypes.data) for f in frames])
    w = np.array([float(x) / 100.0 for x in weight_list], np.float64)         # weight_list is in percent (imfilters.py:690)
    out = DeviceImage(ctx, frames[0].shape) if dev else np.empty_like(frames[0])
    nat.check(ctx.lib.havc_color_temporal_stabilizer(ctx.h, C.cast(ptrs, C.c_void_p), nat.as_ptr(w), len(frames), _p(out),
                                                     *_wh(out)), ctx.h)
    return out


def image_luma_np(ctx, img):
    import ctypes as C
    a = img if is_device(img) else np.ascontiguousarray(img, dtype=np.uint8)
    m = C.c_double()
    nat.check(ctx.lib.havc_image_luma(ctx.h, _p(a), *_wh(a, per_frame=True), C.byref(m)), ctx.h)
    return m.value


def luma_merge_np(ctx, img_dark, img_white, mode, tresh=0.0, grad=0.0):
    a, b, out = _prep(img_dark, img_white)
    nat.check(ctx.lib.havc_image_luma_merge(ctx.h, _p(a), _p(b), int(mode), float(tresh), float(grad), _p(out),
                                            *_wh(a)), ctx.h)
    return out


def chroma_stabilizer_adaptive(img_stable, img_new, base_tol=18, max_extra=22, weight=1.0, device_index=0):
    """imfilters.py:202-269."""
    from PIL import Image
    return Image.fromarray(chroma_stabilizer_adaptive_np(get_context(device_index), np.asarray(img_stable), np.asarray(img_new),
                                                         base_tol, max_extra, weight))


def _chroma_temporal_limiter(cur_img, prv_img, alpha=0.05, device_index=0):
    """imfilters.py:638-666."""
    from PIL import Image
    return Image.fromarray(chroma_temporal_limiter_np(get_context(device_index), np.asarray(cur_img), np.asarray(prv_img), alpha))


def _color_temporal_stabilizer(img_f, weight_list=None, device_index=0):
    """imfilters.py:680-705."""
    from PIL import Image
    return Image.fromarray(color_temporal_stabilizer_np(get_context(device_index), [np.asarray(i) for i in img_f], weight_list))


def get_image_luma(img, maxrange=255, device_index=0):
    """imfilters.py:597-601: round(mean(Y) / maxrange, 6)."""
    return round(image_luma_np(get_context(device_index), np.asarray(img)) / maxrange, 6)


def image_luma_merge_np(ctx, img_dark, img_white, luma=0):
    """imfilters.py:66-77 on arrays / DeviceImages"""
    if luma > 0:
        return luma_merge_np(ctx, img_dark, img_white, 0, round(luma * 255))
    return luma_merge_np(ctx, img_dark, img_white, 3)        # np_rgb_to_gray without threshold: mask = luma itself, merged with /255


def w_image_luma_merge_np(ctx, img_dark, img_white, dark_luma=0.3, white_luma=0.9):
    """imfilters.py:80-100 with w_np_rgb_to_gray's threshold / gradient arithmetic (nputils.py:141-183) on the host"""
    if dark_luma >= white_luma:
        return img_dark
    if dark_luma > 0:
        max_white = round(white_luma * 255)
        tresh = min(round(dark_luma * 255), max_white - 10)
        grad = round(1 / (max_white - tresh), 3)
        return luma_merge_np(ctx, img_dark, img_white, 1, tresh, grad)
    return luma_merge_np(ctx, img_dark, img_white, 2)


def image_luma_merge(img_dark, img_white, luma=0, return_mask=False, device_index=0):
    """imfilters.py:66-77 (hard luma mask built from img_white)."""
    from PIL import Image
    if return_mask:
        raise NotImplementedError("return_mask is a debugging aid of the reference; not on the hot path")
    return Image.fromarray(image_luma_merge_np(get_context(device_index), np.asarray(img_dark), np.asarray(img_white), luma))


def w_image_luma_merge(img_dark, img_white, dark_luma=0.3, white_luma=0.9, return_mask=False, device_index=0):
    """imfilters.py:80-100."""
    from PIL import Image
    if dark_luma >= white_luma:
        return img_dark
    if return_mask:
        raise NotImplementedError("return_mask is a debugging aid of the reference; not on the hot path")
    return Image.fromarray(w_image_luma_merge_np(get_context(device_index), np.asarray(img_dark), np.asarray(img_white), dark_luma, white_luma))


# ---- tweaks and gray-pixel restoration ----------------------------------------------------
_HUE_NAMES = {"red": (0, 30), "orange": (30, 60), "yellow": (60, 90), "yellow-green": (90, 120), "green": (120, 150),
              "blue-green": (150, 180), "cyan": (180, 210), "blue": (210, 240), "blue-violet": (240, 270),
              "violet": (270, 300), "red-violet": (300, 330), "rose": (330, 360)}


def parse_hue_ranges(hue_range):
    """restcolor.py:417-470: 'name' or 'min:max' tokens separated by commas -> flat [lo0, hi0, lo1, hi1, ...] in degrees."""
    flat = []
    for tok in hue_range.split(","):
        if tok in _HUE_NAMES:
            lo, hi = _HUE_NAMES[tok]
        else:
            lo, hi = (float(v) for v in tok.split(":"))
        flat += [float(lo), float(hi)]
    return flat


def image_tweak_np(ctx, img, sat=1.0, cont=1.0, bright=0.0, hue=0.0, gamma=1.0, hue_range="none"):
    import ctypes as C
    if gamma != 1.0:
        # imfilters.py:507-517: the reference passes `uint8_array * 3` (256 entries) to Image.point of an RGB image and
        # Pillow raises; the gamma branch of image_tweak has never produced a frame.  Same error, same text.
        raise ValueError("wrong number of lut entries")
    a, out = _one(ctx, img)
    rng = [] if hue_range in ("none", "") else parse_hue_ranges(hue_range)
    arr = (C.c_double * max(len(rng), 1))(*rng)
    hue_off = int((hue / 360.0) * 255) if hue != 0.0 else 0                  # imfilters.py:530
    nat.check(ctx.lib.havc_image_tweak(ctx.h, _p(a), _p(out), *_wh(a, per_frame=(cont != 1.0)), hue_off,
                                       float(1 + bright / 255) if bright != 0.0 else 1.0, float(cont), float(sat), arr,
                                       len(rng) // 2), ctx.h)
    return out


def image_tweak(img, sat=1, cont=1.0, bright=0, hue=0, gamma=1.0, hue_range="none", device_index=0):
    """imfilters.py:463-504 (PIL in / PIL out): hue shift through Pillow's HSV, ImageEnhance Brightness / Contrast / Color,
    optional hue-range mask against the original."""
    from PIL import Image
    return Image.fromarray(image_tweak_np(get_context(device_index), np.asarray(img), sat, cont, bright, hue, gamma, hue_range))


def luma_levels_lut(luma, luma_min=0.0, gamma=1.0, gamma_luma_min=0.0, gamma_alpha=0.0, gamma_min=0.2, i_min=0, i_max=255):
    """The scalar half of luma_adjusted_levels (imfilters.py:346-364) as the reference writes it, applied to all 256 Y
    values (numpy, so the uint8 wrap of np.add and numpy's own pow are the reference's)."""
    y = np.arange(256, dtype=np.uint8)
    i_alpha = int(255 * (luma_min - luma)) if luma < luma_min else 0
    y_new = np.add(y, i_alpha).clip(i_min, i_max).astype(np.uint8) if i_alpha > 1 else y
    if gamma != 1 and luma < gamma_luma_min:
        g_new = max(gamma * pow(luma / gamma_luma_min, gamma_alpha), gamma_min) if gamma_alpha != 0 else gamma
        y_new = np.power(y_new / 255, 1 / g_new)
        y_new = np.multiply(y_new, 255).clip(i_min, i_max).astype(np.uint8)
    return np.ascontiguousarray(y_new, dtype=np.uint8)


def luma_adjusted_levels_np(ctx, img, luma_min=0.0, gamma=1.0, gamma_luma_min=0.0, gamma_alpha=0.0, gamma_min=0.2, i_min=0, i_max=255):
    a, out = _one(ctx, img)
    luma = image_luma_np(ctx, a) / 255
    lut = luma_levels_lut(luma, luma_min, gamma, gamma_luma_min, gamma_alpha, gamma_min, i_min, i_max)
    nat.check(ctx.lib.havc_luma_lut(ctx.h, _p(a), nat.as_ptr(lut), _p(out), *_wh(a)), ctx.h)
    return out


def luma_adjusted_levels(img, luma_min=0, gamma=1.0, gamma_luma_min=0, gamma_alpha=0, gamma_min=0.2, i_min=0, i_max=255,
                         device_index=0):
    """imfilters.py:335-372."""
    from PIL import Image
    return Image.fromarray(luma_adjusted_levels_np(get_context(device_index), np.asarray(img), luma_min, gamma, gamma_luma_min,
                                                   gamma_alpha, gamma_min, i_min, i_max))


def restore_color_gradient_np(ctx, img_color, img_gray, sat=1.0, tht=50, weight=0.0, alpha=2.0, return_mask=False, algo=0):
    a, b, out = _prep(img_color, img_gray)
    nat.check(ctx.lib.havc_restore_color_gradient(ctx.h, _p(a), _p(b), _p(out), *_wh(a),
                                                  float(sat), int(tht), float(weight), float(alpha), int(algo), int(bool(return_mask))), ctx.h)
    return out


def restore_color_gradient(img_color, img_gray, sat=1.0, tht=50, weight=0, alpha=2.0, return_mask=False, algo=0, device_index=0):
    """restcolor.py:98-134."""
    from PIL import Image
    return Image.fromarray(restore_color_gradient_np(get_context(device_index), np.asarray(img_color), np.asarray(img_gray), sat, tht,
                                                     weight, alpha, return_mask, algo))


def parse_hue_adjust(hue_adjust):
    """restcolor.py:379-414: 'ranges|adjust,weight' -> (hue_range, sat, hue, weight) or None."""
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


def image_chroma_tweak_np(ctx, img, sat=1, bright=0, hue=0, hue_adjust="none"):
    import ctypes as C
    if sat == 1 and bright == 0 and hue == 0 and hue_adjust == "none":
        return img if is_device(img) else np.ascontiguousarray(img, dtype=np.uint8)            # restcolor.py:290-291
    a, out = _one(ctx, img)
    param = None if hue_adjust in ("none", "") else parse_hue_adjust(hue_adjust)
    rng = parse_hue_ranges(param[0]) if param else []
    arr = (C.c_double * max(len(rng), 1))(*rng)
    nat.check(ctx.lib.havc_image_chroma_tweak(ctx.h, _p(a), _p(out), *_wh(a), float(sat), float(bright), int(hue),
                                              1 if param else 0, arr, len(rng) // 2, float(param[1]) if param else 1.0,
                                              int(param[2]) if param else 0, float(param[3]) if param else 0.0), ctx.h)
    return out


def adjust_hue_range_np(ctx, img, hue_adjust="none"):
    """adjust_hue_range (vsslib/restcolor.py:221-286) on a u8 frame / clip (ndarray or DeviceImage): the per-frame body of
    vs_sc_adjust_clip_hue (vsfilters.py:435-455)."""
    import ctypes as C
    if hue_adjust in ("none", ""):
        return img
    param = parse_hue_adjust(hue_adjust)
    if param is None or param[0] in ("none", ""):
        return img
    a, out = _one(ctx, img)
    rng = parse_hue_ranges(param[0])
    arr = (C.c_double * max(len(rng), 1))(*rng)
    nat.check(ctx.lib.havc_image_chroma_tweak(ctx.h, _p(a), _p(out), *_wh(a), 1.0, 0.0, 0, 2, arr, len(rng) // 2, float(param[1]), int(param[2]),
                                              float(param[3])), ctx.h)
    return out


def adjust_hue_range(img, hue_adjust="none", device_index=0):
    """restcolor.py:221-233 (PIL in / PIL out)"""
    from PIL import Image
    if hue_adjust in ("none", ""):
        return img
    return Image.fromarray(adjust_hue_range_np(get_context(device_index), np.asarray(img), hue_adjust))


def image_chroma_tweak(img, sat=1, bright=0, hue=0, hue_adjust="none", device_index=0):
    """imfilters.py:540-548."""
    from PIL import Image
    if sat == 1 and bright == 0 and hue == 0 and hue_adjust == "none":
        return img
    return Image.fromarray(image_chroma_tweak_np(get_context(device_index), np.asarray(img), sat, bright, hue, hue_adjust))
