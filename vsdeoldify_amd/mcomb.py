"""Per-frame bodies of the HAVC model-combination methods (vsdeoldify/vsslib/mcomb.py `merge_frame` selectors), on
uint8 HWC frames, backed by the HIP filters.  The VapourSynth wrappers (ModifyFrame plumbing, scene-change
passthrough) stay in the reference; these are the functions they would call once per frame.

  method 2  SimpleMerge               mcomb.py:206-223   Image.blend(a, b, w)
  method 3  ConstrainedChromaMerge    mcomb.py:333-367   chroma_stabilizer (+ dark-frame red fix)
  method 4  LumaMaskedMerge           mcomb.py:238-271   (w_)image_luma_merge + weighted merge
  method 5  AdaptiveLumaMerge         mcomb.py:289-314   frame-mean luma -> weight -> blend
  method 7  ChromaBoundAdaptiveMerge  mcomb.py:370-437   chroma_stabilizer_adaptive (+ red fix)
  method 6  ChromaRetentionMerge      mcomb.py:450-516   restore_color_gradient (+ blend); the optional Spline64 round
                                                         trip of chroma_resize stays with the caller (zimg)
The dark-frame "red fix" (frame luma <= 0.3, mcomb.py:350-361) = image_tweak (ImageEnhance.Color + hue-range mask) merged
back through w_image_luma_merge.
"""
import numpy as np

from . import imfilters as F
from .device import is_device
from .render import get_context


def _img(x):
    return x if is_device(x) else np.asarray(x)

DEF_STANDARD_DARK, DEF_STANDARD_BRIGHT = 0.22, 0.78        # vsslib/constants.py:28-29


def simple_merge(a, b, weight=0.5, device_index=0):
    if weight == 0.0:
        return _img(a)
    if weight == 1.0:
        return _img(b)
    return F.blend_np(get_context(device_index), a, b, weight)


def _red_fix(ctx, img_stab):
    """mcomb.py:350-361 / 409-420."""
    luma = round(F.image_luma_np(ctx, img_stab) / 255, 6)
    if luma > 0.3:
        return img_stab
    if luma > 0.1:
        dark_luma, white_luma, sat = (0.2, 0.3, 0.9) if luma > 0.2 else (0.1, 0.2, 0.8)        # literals, as the reference writes them
        dark = F.image_tweak_np(ctx, img_stab, sat=sat, hue_range="280:360,0:30")
        max_white = round(white_luma * 255)                                  # w_image_luma_merge (imfilters.py:80-100)
        tresh = min(round(dark_luma * 255), max_white - 10)
        return F.luma_merge_np(ctx, dark, img_stab, 1, tresh, round(1 / (max_white - tresh), 3))
    return F.image_tweak_np(ctx, img_stab, sat=0.7)


def constrained_chroma_merge(a, b, clipb_weight=0.5, chroma_threshold=0.2, red_fix=True, device_index=0):
    ctx = get_context(device_index)
    img_stab = F.chroma_stabilizer_np(ctx, a, b, chroma_threshold, clipb_weight)
    return _red_fix(ctx, img_stab) if red_fix else img_stab


def chroma_bound_adaptive_merge(a, b, red_fix=True, base_tol=14, max_extra=18, clipb_weight=0.5, device_index=0):
    ctx = get_context(device_index)
    img_stab = F.chroma_stabilizer_adaptive_np(ctx, a, b, base_tol, max_extra, clipb_weight)
    return _red_fix(ctx, img_stab) if red_fix else img_stab


def luma_masked_merge(a, b, c=None, luma_mask_limit=0.4, luma_white_limit=0.7, clipm_weight=0.5, device_index=0):
    """a = clipa frame, b = clipb frame, c = de-saturated clipa frame (== a when luma_mask_sat >= 1)."""
    ctx = get_context(device_index)
    c = a if c is None else c
    if luma_mask_limit == luma_white_limit:
        masked = F.image_luma_merge_np(ctx, _img(c), _img(b), luma_mask_limit)
    else:
        masked = F.w_image_luma_merge_np(ctx, _img(c), _img(b), luma_mask_limit, luma_white_limit)
    return simple_merge(a, masked, clipm_weight, device_index) if clipm_weight < 1.0 else masked


def adaptive_luma_merge(a, b, luma_threshold=0.6, alpha=1.0, clipb_weight=0.5, min_weight=0.15, device_index=0):
    ctx = get_context(device_index)
    luma = round(F.image_luma_np(ctx, b) / 255, 6)
    w = max(clipb_weight * pow(luma / luma_threshold, alpha), min_weight) if luma < luma_threshold else clipb_weight
    return F.blend_np(ctx, a, b, w)


def chroma_retention_frame(a, b, sat=0.8, tht=30, mask_weight=0.0, alpha=2.0, return_mask=False, algo=0, device_index=0):
    """The per-frame selector of ChromaRetentionMerge (mcomb.py:450-516 -> vs_sc_recover_gradient_color, vsfilters.py:391-412):
    a = frame to repair (deoldify), b = colour donor (ddcolor).  Frames darker / brighter than the standard luma band get
    weight <= -0.5 and alpha >= 4 (vsfilters.py:403-409).  The clip-level steps around it -- the optional Spline64 round trip
    (chroma_resize) and the closing std.Merge(clip_a, restored, clipb_weight) -- are VapourSynth core filters and stay there."""
    ctx = get_context(device_index)
    alpha = max(min(alpha, 10.0), 1.0)                                       # DEF_MIN/MAX_COLOR_ALPHA, constants.py:80-81
    luma = round(F.image_luma_np(ctx, a) / 255, 6)
    if not (DEF_STANDARD_DARK <= luma <= DEF_STANDARD_BRIGHT):
        mask_weight = min(mask_weight, -0.5)
        alpha = max(alpha, 4.0)
    return F.restore_color_gradient_np(ctx, b, a, sat, tht, mask_weight, alpha, return_mask, algo)
