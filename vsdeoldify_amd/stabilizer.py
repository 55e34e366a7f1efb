"""Per-frame bodies of the HAVC_stabilizer colour filters (`vs_dark_tweak`, `vs_chroma_bright_tweak`,
`vs_colormap`; vsdeoldify/vsslib/vsfilters.py:525-641) on uint8 HWC frames, backed by the HIP filters.  The VapourSynth wrappers
(ModifyFrame plumbing, scene-change passthrough) stay in the reference; these are the functions their selectors call per frame."""
import numpy as np

from . import imfilters as F
from .render import get_context


def _luma_merge(ctx, img2, img1, lo, hi):
    if lo == hi:                                                           # image_luma_merge (imfilters.py:66-77)
        return F.luma_merge_np(ctx, img2, img1, 0, round(lo * 255)) if lo > 0 else F.luma_merge_np(ctx, img2, img1, 3)
    if lo >= hi:                                                           # w_image_luma_merge returns img_dark (imfilters.py:84-85)
        return np.asarray(img2)
    if lo > 0:
        max_white = round(hi * 255)
        tresh = min(round(lo * 255), max_white - 10)
        return F.luma_merge_np(ctx, img2, img1, 1, tresh, round(1 / (max_white - tresh), 3))
    return F.luma_merge_np(ctx, img2, img1, 2)


def dark_tweak_frame(img, dark_threshold=0.3, dark_amount=0.8, dark_hue_adjust="none", device_index=0):
    """vs_sc_dark_tweak (vsfilters.py:600-632): darker, less saturated copy merged in where the luma is low."""
    ctx = get_context(device_index)
    white = min(max(dark_threshold, 0.1), 0.50)
    d_sat = min(max(1.1 - dark_amount, 0.10), 0.80)
    d_bright = -min(max(dark_amount, 0.20), 0.90)
    img = np.asarray(img)
    return _luma_merge(ctx, F.image_tweak_np(ctx, img, sat=d_sat, bright=d_bright, hue_range=dark_hue_adjust), img, 0.1, white)


def chroma_bright_tweak_frame(img, black_threshold=0.3, white_threshold=0.6, dark_sat=0.8, dark_bright=-0.10, chroma_adjust="none",
                              device_index=0):
    """vs_sc_chroma_bright_tweak (vsfilters.py:525-547)."""
    ctx = get_context(device_index)
    img = np.asarray(img)
    return _luma_merge(ctx, F.image_chroma_tweak_np(ctx, img, sat=dark_sat, bright=dark_bright, hue_adjust=chroma_adjust), img,
                       black_threshold, white_threshold)


def colormap_frame(img, colormap="none", device_index=0):
    """_vs_sc_colormap (vsfilters.py:575-590): direct colour mapping through the "chroma adjustment" string."""
    return F.image_chroma_tweak_np(get_context(device_index), np.asarray(img), hue_adjust=colormap)
