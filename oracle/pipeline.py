"""End-to-end CPU restatement of the DeOldify per-frame path — ORACLE / TEST INFRASTRUCTURE ONLY.

  colorizer_filter  <- ColorizerFilter.filter / BaseFilter._model_process / _post_process
                       (vsdeoldify/deoldify/filters.py:23-110)
  model_image_render <- ModelImageRender.get_transformed_image (vsdeoldify/deoldify/visualize.py:118-137)
  chroma_post_process / chroma_stabilizer / image_weighted_merge <- vsdeoldify/vsslib/imfilters.py
"""
import numpy as np
import torch

from . import cvcolor, imaging, unet


def _to_torch_sd(sd):
    return {k: (v if isinstance(v, torch.Tensor) else torch.from_numpy(np.asarray(v))) for k, v in sd.items()}


def raw_color_square(sd, arch, sq_u8):
    """_model_process on an image already at the render size: uint8 [S,S,3] -> uint8 [S,S,3]."""
    x = torch.from_numpy(imaging.model_input(sq_u8))
    with torch.no_grad():
        y = unet.unet_forward(_to_torch_sd(sd), x, arch)
    return imaging.model_output_u8(y[0].numpy())


def post_process(raw_color, orig):
    """ColorizerFilter._post_process (filters.py:100-110) == chroma_post_process (imfilters.py:312-321)."""
    c = cvcolor.rgb2yuv_u8(raw_color)
    o = cvcolor.rgb2yuv_u8(orig)
    o[..., 1:3] = c[..., 1:3]
    return cvcolor.yuv2rgb_u8(o)


chroma_post_process = post_process


def colorizer_filter(sd, arch, img_u8, render_factor, do_post=True):
    from PIL import Image
    S = render_factor * 16
    im = Image.fromarray(img_u8)
    sq = np.asarray(im.resize((S, S), resample=Image.BILINEAR))          # filters.py:37-41
    raw = raw_color_square(sd, arch, sq)
    raw = np.asarray(Image.fromarray(raw).resize(im.size, resample=Image.BILINEAR))   # filters.py:70-73
    return post_process(raw, img_u8) if do_post else raw


def model_image_render(sds, modelname, img_u8, render_factor, video_weight, do_post=True):
    """sds: {'video': sd, 'stable'|'artistic': sd}.  visualize.py:118-137."""
    v = colorizer_filter(sds["video"], "wide", img_u8, render_factor, do_post)
    if modelname == "video":
        return v
    second = colorizer_filter(sds[modelname], "deep" if modelname == "artistic" else "wide", img_u8, render_factor, do_post)
    return imaging.pil_blend(second, v, video_weight)


def chroma_stabilizer(img_stable, img_new, alpha=0.15, weight=1.0):
    """vsslib/imfilters.py:160-200 (numpy float64 products, truncating uint8 casts, cap then floor)."""
    yuv1 = cvcolor.rgb2yuv_u8(img_stable)
    y1, u1, v1 = yuv1[..., 0], yuv1[..., 1], yuv1[..., 2]
    u_up = np.multiply(u1, 1 + alpha).clip(0, 255).astype(np.uint8)
    v_up = np.multiply(v1, 1 + alpha).clip(0, 255).astype(np.uint8)
    u_dn = np.multiply(u1, 1 - alpha).clip(0, 255).astype(np.uint8)
    v_dn = np.multiply(v1, 1 - alpha).clip(0, 255).astype(np.uint8)
    yuv2 = cvcolor.rgb2yuv_u8(img_new)
    out = np.copy(yuv2)

    def clip(a, lo, hi):
        a = np.where(a > hi, hi, a).astype(np.uint8)
        return np.where(a < lo, lo, a).astype(np.uint8)
    out[..., 0] = y1
    out[..., 1] = clip(yuv2[..., 1], u_dn, u_up)
    out[..., 2] = clip(yuv2[..., 2], v_dn, v_up)
    rgb = cvcolor.yuv2rgb_u8(out)
    return imaging.pil_blend(img_stable, rgb, weight) if weight < 1.0 else rgb


def colorize_frame_fullsize(sds, modelname, frame_u8, render_factor, video_weight=0.5):
    """HAVC_colorizer(method=0) per-frame flow on one full-size frame (vsdeoldify/__init__.py:2502-2523):
    Spline64 squash to S x S -> ModelImageRender -> Spline64 back -> chroma_post_process vs the source
    (vs_recover_clip_luma, vsslib/vsfilters.py:863-899).  The resampler is the harness stand-in (resample.py)."""
    from . import resample
    S = render_factor * 16
    h, w, _ = frame_u8.shape
    sq = frame_u8 if (w, h) == (S, S) else resample.resize_rgb8(frame_u8, S, S)
    col = model_image_render(sds, modelname, sq, render_factor, video_weight, True)
    up = col if (w, h) == (S, S) else resample.resize_rgb8(col, w, h)
    return post_process(up, frame_u8)


# ---- merge-method filters (vsslib/imfilters.py, nputils.py) restated ------------------------------------------------
def _np_luma(img):
    a = np.asarray(img)
    return (a[:, :, 0] * 0.299 + a[:, :, 1] * 0.587 + a[:, :, 2] * 0.114).clip(0, 255)          # nputils.py:101-112


def image_luma_merge(img_dark, img_white, luma=0):
    """imfilters.py:66-77 -> np_rgb_to_gray (nputils.py:101-122) + np_image_mask_merge (nputils.py:196-211)."""
    l2 = _np_luma(img_white)
    if luma > 0:
        mask = np.where(l2 > round(luma * 255), 255, 0).astype(np.uint8)
    else:
        mask = l2.astype(np.uint8)
    mw = (mask / 255).astype(float)[..., None]
    m = np.asarray(img_dark) * (1 - mw) + np.asarray(img_white) * mw
    return m.clip(0, 255).astype(np.uint8)


def w_image_luma_merge(img_dark, img_white, dark_luma=0.3, white_luma=0.9):
    """imfilters.py:80-100 -> w_np_rgb_to_gray (nputils.py:141-183) + w_np_image_mask_merge (nputils.py:224-253)."""
    if dark_luma >= white_luma:
        return np.asarray(img_dark)
    l2 = _np_luma(img_white)
    if dark_luma > 0:
        max_white = round(white_luma * 255)
        tresh = min(round(dark_luma * 255), max_white - 10)
        grad = round(1 / (max_white - tresh), 3)
        lg = ((l2 - tresh) * grad).astype(float)
        w = np.where(lg > 1.0, 1.0, lg).astype(np.float32)
        w = np.where(w < 0.0, 0.0, w).astype(np.float32)
    else:
        w = np.divide(l2, 255.0)
    mw = w.astype(float)[..., None]
    m = np.multiply(np.asarray(img_dark), 1 - mw) + np.multiply(np.asarray(img_white), mw)
    return m.clip(0, 255).astype(np.uint8)


def get_image_luma(img, maxrange=255):
    """imfilters.py:597-601."""
    return round(float(np.mean(cvcolor.rgb2yuv_u8(img)[:, :, 0])) / maxrange, 6)


def chroma_temporal_limiter(cur_img, prv_img, alpha=0.05):
    """imfilters.py:638-666 (float64 bounds, cap then floor, truncating uint8 casts)."""
    yuv1 = cvcolor.rgb2yuv_u8(prv_img)
    yuv2 = cvcolor.rgb2yuv_u8(cur_img)
    out = np.copy(yuv2)
    for ch in (1, 2):
        up, dn = np.multiply(yuv1[..., ch], 1 + alpha), np.multiply(yuv1[..., ch], 1 - alpha)
        a = np.where(yuv2[..., ch] > up, up, yuv2[..., ch]).astype(np.uint8)
        out[..., ch] = np.where(a < dn, dn, a).astype(np.uint8)
    return cvcolor.yuv2rgb_u8(out)


def chroma_stabilizer_adaptive(img_stable, img_new, base_tol=18, max_extra=22, weight=1.0):
    """imfilters.py:202-269."""
    yuv1 = cvcolor.rgb2yuv_u8(np.asarray(img_stable).astype(np.uint8))
    y1 = yuv1[:, :, 0].astype(np.float32)
    u1 = yuv1[:, :, 1].astype(np.int16) - 128
    v1 = yuv1[:, :, 2].astype(np.int16) - 128
    yuv2 = cvcolor.rgb2yuv_u8(np.asarray(img_new).astype(np.uint8))
    u2 = yuv2[:, :, 1].astype(np.int16) - 128
    v2 = yuv2[:, :, 2].astype(np.int16) - 128
    texture = np.clip(np.abs(cvcolor.Laplacian(y1, cvcolor.CV_32F)) / 255.0, 0.0, 1.0)
    tol = base_tol + max_extra * texture
    u_m = np.clip(u2, np.clip(u1 - tol, -128, 127), np.clip(u1 + tol, -128, 127))
    v_m = np.clip(v2, np.clip(v1 - tol, -128, 127), np.clip(v1 + tol, -128, 127))
    yuv_out = np.stack([yuv1[:, :, 0], (u_m + 128).astype(np.uint8), (v_m + 128).astype(np.uint8)], axis=2)
    rgb = cvcolor.yuv2rgb_u8(yuv_out)
    return imaging.pil_blend(img_stable, rgb, weight) if weight < 1.0 else rgb


def color_temporal_stabilizer(img_f, weight_list):
    """imfilters.py:680-705."""
    n = len(weight_list)
    nh = round((n - 1) / 2)
    yuv_new = cvcolor.rgb2yuv_u8(np.asarray(img_f[nh]))
    yuv_m = np.multiply(yuv_new, weight_list[nh] / 100.0)
    for i in list(range(0, nh)) + list(range(nh + 1, n)):
        yuv_m += np.multiply(cvcolor.rgb2yuv_u8(np.asarray(img_f[i])), weight_list[i] / 100.0)
    yuv_new[:, :, 1] = yuv_m[:, :, 1]
    yuv_new[:, :, 2] = yuv_m[:, :, 2]
    return cvcolor.yuv2rgb_u8(yuv_new)


# ---- model-combination graph (vsslib/mcomb.py) restated on uint8 frames ---------------------------------------------
def luma_masked_merge(a, b, luma_mask_limit=0.4, luma_white_limit=0.7, clipm_weight=0.5):
    """LumaMaskedMerge.merge_frame with luma_mask_sat >= 1 (clipc == clipa), vsslib/mcomb.py:238-271."""
    if luma_mask_limit == luma_white_limit:
        masked = image_luma_merge(a, b, luma_mask_limit)
    else:
        masked = w_image_luma_merge(a, b, luma_mask_limit, luma_white_limit)
    return imaging.pil_blend(a, masked, clipm_weight) if clipm_weight < 1.0 else masked


def adaptive_luma_merge(a, b, luma_threshold=0.6, alpha=1.0, clipb_weight=0.5, min_weight=0.15):
    """AdaptiveLumaMerge.merge_frame, vsslib/mcomb.py:289-314."""
    luma = get_image_luma(b)
    w = max(clipb_weight * pow(luma / luma_threshold, alpha), min_weight) if luma < luma_threshold else clipb_weight
    return imaging.pil_blend(a, b, w)


def chroma_retention_merge(a, b, sat=0.8, tht=30, clipb_weight=0.9, alpha=2.0, mask_weight=0.0, algo=0):
    """ChromaRetentionMerge with chroma_resize=False (vsslib/mcomb.py:450-516): color_grad_frame (vsslib/vsfilters.py:391-412)
    then vs_simple_merge(clip_a, restored, clipb_weight) = std.Merge, restated as the float32 truncating blend of Image.blend
    (std.Merge of 8-bit clips rounds instead: outside the parity contract like every VapourSynth core filter)."""
    from . import tweaks
    alpha = max(min(alpha, 10.0), 1.0)                                     # DEF_MIN/MAX_COLOR_ALPHA, vsslib/constants.py:80-81
    luma = get_image_luma(a)
    if not (0.22 <= luma <= 0.78):                                         # DEF_STANDARD_DARK / BRIGHT, constants.py:28-29
        mask_weight, alpha = min(mask_weight, -0.5), max(alpha, 4.0)
    restored = tweaks.restore_color_gradient(b, a, sat, tht, mask_weight, alpha, False, algo)
    if clipb_weight == 0.0:
        return a
    if clipb_weight == 1.0:
        return restored
    return imaging.pil_blend(a, restored, clipb_weight)


def combine_models(a, b, method, w, cmc_p=(0.15, True, 20, 24), lmm_p=(0.15, 0.65, 1.0), alm_p=(0.8, 1.0, 0.15), crt_p=(0.8, 30, 2, False, 0, 0)):
    """vs_sc_combine_models (vsslib/mcomb.py:125-192) on one pair of uint8 frames, sat = [1, 1], hue = [0, 0]."""
    from . import tweaks
    red_fix, base_tol, max_extra = (cmc_p[1], cmc_p[2], cmc_p[3]) if len(cmc_p) > 1 else (True, 20, 24)
    if a is None or b is None:
        return a if b is None else b
    if method == 2:
        return imaging.pil_blend(a, b, w)
    if method == 3:
        ccm = tweaks.constrained_chroma_merge(a, b, cmc_p[0], w, red_fix)
        return imaging.pil_blend(ccm, imaging.pil_blend(a, b, min(w, 0.6)), 0.3)
    if method == 4:
        return luma_masked_merge(a, b, lmm_p[0], lmm_p[1], w)
    if method == 5:
        return adaptive_luma_merge(a, b, alm_p[0], alm_p[1], w, alm_p[2])
    if method == 6:
        return chroma_retention_merge(a, b, crt_p[0], crt_p[1], w, crt_p[2], crt_p[4], crt_p[5])
    if method == 7:
        return tweaks.chroma_bound_adaptive_merge(a, b, base_tol, max_extra, w, red_fix)
    raise ValueError("HAVC: only dd_method in (0,6) is supported")
