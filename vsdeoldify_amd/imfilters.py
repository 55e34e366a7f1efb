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
