"""Host-side mirror of the per-frame pixel filters the merge path uses (vsdeoldify/vsslib/imfilters.py),
backed by the HIP kernels in csrc/colorfilters.hip.  PIL.Image in / PIL.Image out like the reference;
the *_np variants take/return uint8 HWC arrays.
"""
import numpy as np

from . import _native as nat
from .render import get_context


def _prep(a, b):
    a = np.ascontiguousarray(a, dtype=np.uint8)
    b = np.ascontiguousarray(b, dtype=np.uint8)
    if a.shape != b.shape or a.ndim != 3 or a.shape[2] != 3:
        raise ValueError("images do not match")          # PIL raises ValueError("images do not match")
    return a, b, np.empty_like(a)


def blend_np(ctx, a, b, w):
    a, b, out = _prep(a, b)
    nat.check(ctx.lib.havc_blend(ctx.h, nat.as_ptr(a), nat.as_ptr(b), float(w), nat.as_ptr(out), a.shape[1], a.shape[0]), ctx.h)
    return out


def chroma_post_process_np(ctx, color, orig):
    a, b, out = _prep(color, orig)
    nat.check(ctx.lib.havc_chroma_post_process(ctx.h, nat.as_ptr(a), nat.as_ptr(b), nat.as_ptr(out), a.shape[1], a.shape[0]), ctx.h)
    return out


def chroma_stabilizer_np(ctx, img_stable, img_new, alpha=0.15, weight=1.0):
    a, b, out = _prep(img_stable, img_new)
    nat.check(ctx.lib.havc_chroma_stabilizer(ctx.h, nat.as_ptr(a), nat.as_ptr(b), float(alpha), float(weight), nat.as_ptr(out),
                                             a.shape[1], a.shape[0]), ctx.h)
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
    nat.check(ctx.lib.havc_chroma_stabilizer_adaptive(ctx.h, nat.as_ptr(a), nat.as_ptr(b), float(base_tol), float(max_extra),
                                                      float(weight), nat.as_ptr(out), a.shape[1], a.shape[0]), ctx.h)
    return out


def chroma_temporal_limiter_np(ctx, cur, prv, alpha=0.05):
    a, b, out = _prep(cur, prv)
    nat.check(ctx.lib.havc_chroma_temporal_limiter(ctx.h, nat.as_ptr(a), nat.as_ptr(b), float(alpha), nat.as_ptr(out), a.shape[1], a.shape[0]), ctx.h)
    return out


def color_temporal_stabilizer_np(ctx, frames, weight_list):
    import ctypes as C
    frames = [np.ascontiguousarray(f, dtype=np.uint8) for f in frames]
    if len(frames) != len(weight_list) or not 1 <= len(frames) <= 9 or any(f.shape != frames[0].shape for f in frames):
        raise ValueError("frames / weight_list mismatch (1..9 frames of one size)")
    ptrs = (C.c_void_p * len(frames))(*[f.ctypes.data for f in frames])
    w = np.array([float(x) / 100.0 for x in weight_list], np.float64)         # weight_list is in percent (imfilters.py:690)
    out = np.empty_like(frames[0])
    nat.check(ctx.lib.havc_color_temporal_stabilizer(ctx.h, C.cast(ptrs, C.c_void_p), nat.as_ptr(w), len(frames), nat.as_ptr(out),
                                                     out.shape[1], out.shape[0]), ctx.h)
    return out


def image_luma_np(ctx, img):
    import ctypes as C
    a = np.ascontiguousarray(img, dtype=np.uint8)
    m = C.c_double()
    nat.check(ctx.lib.havc_image_luma(ctx.h, nat.as_ptr(a), a.shape[1], a.shape[0], C.byref(m)), ctx.h)
    return m.value


def luma_merge_np(ctx, img_dark, img_white, mode, tresh=0.0, grad=0.0):
    a, b, out = _prep(img_dark, img_white)
    nat.check(ctx.lib.havc_image_luma_merge(ctx.h, nat.as_ptr(a), nat.as_ptr(b), int(mode), float(tresh), float(grad), nat.as_ptr(out),
                                            a.shape[1], a.shape[0]), ctx.h)
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


def image_luma_merge(img_dark, img_white, luma=0, return_mask=False, device_index=0):
    """imfilters.py:66-77 (hard luma mask built from img_white)."""
    from PIL import Image
    if return_mask:
        raise NotImplementedError("return_mask is a debugging aid of the reference; not on the hot path")
    if luma > 0:
        out = luma_merge_np(get_context(device_index), np.asarray(img_dark), np.asarray(img_white), 0, round(luma * 255))
    else:                                     # np_rgb_to_gray without threshold: mask = luma itself, merged with /255
        out = luma_merge_np(get_context(device_index), np.asarray(img_dark), np.asarray(img_white), 3)
    return Image.fromarray(out)


def w_image_luma_merge(img_dark, img_white, dark_luma=0.3, white_luma=0.9, return_mask=False, device_index=0):
    """imfilters.py:80-100 with w_np_rgb_to_gray's threshold / gradient arithmetic (nputils.py:141-183) on the host."""
    from PIL import Image
    if dark_luma >= white_luma:
        return img_dark
    if return_mask:
        raise NotImplementedError("return_mask is a debugging aid of the reference; not on the hot path")
    ctx = get_context(device_index)
    if dark_luma > 0:
        max_white = round(white_luma * 255)
        tresh = min(round(dark_luma * 255), max_white - 10)
        grad = round(1 / (max_white - tresh), 3)
        out = luma_merge_np(ctx, np.asarray(img_dark), np.asarray(img_white), 1, tresh, grad)
    else:
        out = luma_merge_np(ctx, np.asarray(img_dark), np.asarray(img_white), 2)
    return Image.fromarray(out)
