"""The public HAVC entry points without VapourSynth, executed on the MI355X:

    HAVC_colorizer   vsdeoldify/__init__.py:2290-2523     HAVC_merge       vsdeoldify/__init__.py:2536-2675
    HAVC_ddeoldify   vsdeoldify/__init__.py:3612-3628     ddeoldify        vsdeoldify/__init__.py:3642-3653

Same names, argument lists, defaults, parameter normalisation, frame-size rule, model routing, combine dispatch
(vsslib/mcomb.py:125-192) and error texts; a "clip" is a uint8 array [n, h, w, 3] (or one frame [h, w, 3], or a
`device.DeviceImage` of either shape: then the whole graph runs without leaving HBM).  `HAVCFrameColorizer` is the object
behind `HAVC_colorizer` (models and nets are built once and reused between calls).

What only VapourSynth can do stays there and is REFUSED here instead of being approximated: `vs_tweak` (deoldify / ddcolor
sat / hue other than 1 / 0, `luma_mask_sat` < 1), scene detection (`sc_threshold` > 0, `sc_min_freq` > 0), the parts of the DDColor
pre-tweaks that are VapourSynth filters (`ddtweak`: bright / cont / gamma through vs_tweak, rgb_denoise, retinex), non-RGB24 formats.
Computed here: the hue adjustment vs_sc_ddcolor applies to every DDColor frame (default "300:360|0.8,0.1") and the luma-constrained
pre-tweak with its luma recovery (HAVCFrameColorizer._read_ddtweak).  zimg's Spline64 is replaced by the library's own Spline64 (outside the parity
contract).

    out = HAVC_colorizer(clip, method=2, mweight=0.4, torch_dir=..., ...)          # clip: uint8 [n, 1080, 1920, 3]
    col = HAVCFrameColorizer(method=2, mweight=0.4, package_dir=...); out = col.colorize(frame)
"""
import math
import os

import numpy as np

from . import _native as nat
from .precision import DEFAULT_PRECISION, resolve as resolve_precision
from . import imfilters as F
from . import mcomb
from .device import DeviceImage, is_device
from .render import get_context

DEF_CMC_p = [0.15, True, 20, 24]            # vsslib/constants.py:19-22
DEF_LMM_p = [0.15, 0.65, 1.0]
DEF_ALM_p = [0.8, 1.0, 0.15]
DEF_CRT_p = [0.8, 30, 2, False, 0, 0]
DEF_TWEAK_p = [0.0, 1.0, 2.5, True, 0.3, 0.6, 1.5, 0.5]     # vsslib/constants.py:23
DEF_THT_WHITE, DEF_THT_BLACK = 0.88, 0.12
DEF_STABLE_WEIGHT = DEF_ARTISTIC_WEIGHT = 0.5  # vsslib/constants.py:56-57


class HAVCError(ValueError):
    """what the reference raises as vs.Error / HAVC_LogMessage(EXCEPTION)"""


def _as_clip(x):
    """-> (4-D operand, was_single_frame)"""
    if is_device(x):
        return (x, False) if x.ndim == 4 else (x.reshaped((1,) + x.shape), True)
    a = np.ascontiguousarray(x, dtype=np.uint8)
    if a.ndim == 3:
        a, single = a[None], True
    else:
        single = False
    if a.ndim != 4 or a.shape[3] != 3:
        raise HAVCError("HAVC: only RGB24 clips (uint8 [n, h, w, 3]) are supported")
    return a, single


class HAVCFrameColorizer:
    def __init__(self, method=2, mweight=0.4, deoldify_p=(0, 24, 1.0, 0.0), ddcolor_p=(1, 24, 1.0, 0.0, True), cmc_p=DEF_CMC_p,
                 lmm_p=DEF_LMM_p, alm_p=DEF_ALM_p, crt_p=DEF_CRT_p, cmb_sw=False, device_index=0, package_dir=None,
                 ddcolor_model_dir=None, state_dicts=None, ddcolor_state_dict=None, zhang_state_dict=None, max_batch=1,
                 ddcolor_kwargs=None, ddtweak=(False, False, False), ddtweak_p=(DEF_TWEAK_p, "none"), precision=None):
        """precision: "fast" / "precise" for EVERY model of the graph (DeOldify, DDColor, the Zhang colorizers): the reference runs them all in fp32
        (deoldify/filters.py:45-68, vsslib/vsmodels.py:353-363, colorization/__init__.py:76-95); None reads HAVC_PRECISION, then the package default "precise" (vsdeoldify_amd/precision.py)."""
        try:
            self.precision = resolve_precision(precision)        # explicit > HAVC_PRECISION > "precise" (vsdeoldify_amd/precision.py)
        except ValueError as e:
            raise HAVCError(f"HAVC: {e}") from None
        # ---- __init__.py:2452-2462: method <-> merge weight normalisation ----
        merge_weight = 0.0 if method == 0 else (1.0 if method == 1 else mweight)
        if merge_weight == 0.0:
            method = 0
        elif merge_weight == 1.0:
            method = 1
        if method not in range(0, 8):
            raise HAVCError("HAVC: only dd_method in (0,6) is supported")                        # mcomb.py:192
        self.method, self.merge_weight, self.cmb_sw = method, merge_weight, cmb_sw
        self._read_ddtweak(ddtweak, ddtweak_p)                   # refuses what only VapourSynth can do before anything touches the GPU
        self.deoldify_model, self.deoldify_rf, d_sat, d_hue = deoldify_p[:4]
        self.ddcolor_model, self.ddcolor_rf, c_sat, c_hue = ddcolor_p[:4]
        if device_index > 7:
            raise HAVCError("HAVC_colorizer: wrong device_index, choices are: GPU0...GPU7 (CPU=99 is not supported by this library)")
        if self.ddcolor_rf != 0 and self.ddcolor_rf not in range(10, 65):
            raise HAVCError("HAVC_colorizer: ddcolor render_factor must be between: 10-64")       # __init__.py:2482-2483
        if (d_sat, d_hue) != (1.0, 0.0) and method != 1 or (c_sat, c_hue) != (1.0, 0.0) and method != 0:
            raise NotImplementedError("sat / hue of deoldify_p / ddcolor_p go through vs_tweak (VapourSynth std.Expr + zimg): not in this harness")
        self.cmc_p, self.lmm_p, self.alm_p, self.crt_p = list(cmc_p), list(lmm_p), list(alm_p), list(crt_p)
        if method == 4 and self.lmm_p[2] < 1:
            raise NotImplementedError("luma_mask_sat < 1 uses vs_tweak (VapourSynth): not in this harness")
        self.device_index, self.ctx = device_index, get_context(device_index)
        self.max_batch = max_batch
        self._package_dir, self._dd_dir = package_dir, ddcolor_model_dir
        self._sds, self._dd_sd, self._zh_sd = state_dicts, ddcolor_state_dict, zhang_state_dict
        self._dd_kwargs = dict(ddcolor_kwargs or {})
        self._deoldify = self._ddcolor = self._zhang = None
        self._dd_size = None
        # Methods that run BOTH models on a device clip run them side by side: DDColor on a context (HIP stream) of its own, from a second
        # thread, while DeOldify runs on this one -- two independent chains of launches fill the chip better than one after the other
        # (HAVC_OVERLAP_MODELS=0: one after the other on one stream).  Same bytes either way.
        self.overlap_models = os.environ.get("HAVC_OVERLAP_MODELS", "1") != "0"
        self._pool = None

    def _read_ddtweak(self, flags, tweaks):
        """vs_sc_ddcolor's tweak handling WITHOUT scene detection (vsslib/vsmodels.py:304-344,365-374; scenechange = False because
        sc_threshold = sc_min_freq = 0, __init__.py:2496): what can be computed without VapourSynth is computed, the rest is refused.
          * hue_adjust (ddtweak_p[1], HAVC_colorizer's default "300:360|0.8,0.1"): adjust_hue_range on EVERY DDColor frame — applied;
          * tweaks_enabled with luma_constrained_tweak and neutral bright / cont (the DEF_TWEAK_p defaults): luma_adjusted_levels on every
            frame in front of DDColor, the source's luma put back behind it (vs_recover_clip_luma) — applied;
          * bright / cont / gamma through vs_tweak (std.Expr / std.Levels), rgb_denoise, vs_auto_levels (retinex): VapourSynth filters — refused."""
        flags = list(flags) if isinstance(flags, (list, tuple)) else [flags, False, False]
        self.dd_tweaks_enabled, denoise, retinex = (bool(f) for f in (flags + [False, False])[:3])
        if len(tweaks) == 2:
            t, hue_adjust = list(tweaks[0]), str(tweaks[1]).lower()
        else:
            t, hue_adjust = list(tweaks[:8]), (tweaks[8] if len(tweaks) > 8 else "none")
        self.dd_hue_adjust = hue_adjust
        self.dd_levels = None
        if denoise:
            raise NotImplementedError("ddtweak[1] (rgb_denoise) is a VapourSynth filter chain: not in this harness")
        if self.dd_tweaks_enabled:
            bright, cont, gamma, constrained, luma_min, gamma_luma_min, gamma_alpha, gamma_min = t[:8]
            if retinex:
                raise NotImplementedError("ddtweak[2] (vs_auto_levels / retinex) is a VapourSynth filter chain: not in this harness")
            if not constrained or bright != 0 or cont != 1:
                raise NotImplementedError("without scene detection vs_sc_tweak is vs_tweak (std.Expr / std.Levels): only the luma-constrained tweak "
                                          "with neutral bright / cont (the defaults) is computed here")
            self.dd_levels = (luma_min, gamma, gamma_luma_min, gamma_alpha, gamma_min)

    def _ddcolor_branch(self, sq, input_size, ctx=None):
        """vs_sc_ddcolor (vsmodels.py:290-375) on the squashed clip: [pre-tweak ->] DDColor / Zhang [-> hue adjust] [-> luma of the clip back].
        ctx: the context the branch's filters run on (the DDColor context when the two models run side by side)"""
        ctx = ctx or self.ctx
        src = sq
        if self.dd_levels is not None:                                # sc_constrained_tweak(scenechange=False): luma_adjusted_levels per frame
            frames = [F.luma_adjusted_levels_np(ctx, sq.frame(i) if is_device(sq) else sq[i], *self.dd_levels) for i in range(sq.shape[0])]
            if is_device(sq):
                src = DeviceImage(ctx, sq.shape)
                for i, f in enumerate(frames):
                    src.frame(i).copy_from(f)
            else:
                src = np.stack(frames)
        b = self._ddcolor_clip(src, input_size)
        if self.dd_hue_adjust not in ("none", ""):
            b = F.adjust_hue_range_np(ctx, b if is_device(b) else b.reshape((-1,) + b.shape[2:]), self.dd_hue_adjust)
            if not is_device(b):
                b = b.reshape(sq.shape)
        if self.dd_tweaks_enabled:
            b = F.chroma_post_process_np(ctx, b if is_device(b) else b.reshape((-1,) + b.shape[2:]),
                                         sq if is_device(sq) else sq.reshape((-1,) + sq.shape[2:]))
            if not is_device(b):
                b = b.reshape(sq.shape)
        return b

    # ---- model routing: vsslib/vsmodels.py:196-213 (deoldify), :290-350 (ddcolor / zhang) ----
    def _deoldify_render(self):
        if self._deoldify is None:
            from .render import ModelImageRender
            name, w = {0: ("video", 0), 1: ("stable", DEF_STABLE_WEIGHT), 2: ("artistic", DEF_ARTISTIC_WEIGHT)}.get(self.deoldify_model, ("video", 0))
            self._deoldify = ModelImageRender(self._package_dir, name, self.deoldify_rf, video_weight=w, device_index=self.device_index,
                                              state_dicts=self._sds, max_batch=self.max_batch, precision=self.precision)
        return self._deoldify

    def _ddcolor_model(self, input_size):
        if self._ddcolor is None or self._dd_size != input_size:
            from .ddcolor import DDColorRender
            kw = dict(self._dd_kwargs)
            if self._side_by_side():
                kw.setdefault("worker", ("havc-ddcolor", 0))
            kw.setdefault("precision", self.precision)
            self._ddcolor = DDColorRender(self.ddcolor_model, input_size, self.device_index, state_dict=self._dd_sd, model_dir=self._dd_dir, **kw)
            self._dd_size = input_size
        return self._ddcolor

    def _ddcolor_clip(self, sq, input_size):
        """sq: [n, fs, fs, 3] ndarray or DeviceImage -> same kind"""
        if self.ddcolor_model in (0, 1):
            return self._ddcolor_model(input_size).colorize_frames(sq, max_batch=self.max_batch)
        from .colorization import ModelColorization                                               # vsmodels.py:346-350
        if self._zhang is None:
            self._zhang = ModelColorization("siggraph17" if self.ddcolor_model == 2 else "eccv16", True, self.device_index, state_dict=self._zh_sd,
                                            precision=self.precision)
        return self._zhang.colorize_frames(sq)                                                    # host or device clip: havc_zhang_frames takes both

    def _side_by_side(self):
        return self.overlap_models and self.method not in (0, 1) and self.ddcolor_model in (0, 1)

    def _deoldify_clip(self, sq):
        """ModelImageRender over a clip.  Frames at the model's render size (the usual case: frame_size is derived from the larger
        render factor) run as device batches; any other size goes frame by frame through get_transformed_image, which squashes /
        un-squashes with Pillow BILINEAR on the host exactly where the reference does (deoldify/filters.py:37-41,70-73)."""
        from PIL import Image
        r = self._deoldify_render()
        S = self.deoldify_rf * 16
        if tuple(sq.shape[1:3]) == (S, S):
            return r.render_square_batch(sq)
        host = sq.numpy() if is_device(sq) else sq
        out = np.stack([np.asarray(r.get_transformed_image(Image.fromarray(f))) for f in host])
        return DeviceImage.from_numpy(self.ctx, out) if is_device(sq) else out

    def frame_size(self, width):
        """__init__.py:2490-2502."""
        dd_rf = self.ddcolor_rf or min(max(math.trunc(0.4 * width / 16), 16), 32)
        return dd_rf, min(max(dd_rf, self.deoldify_rf) * 16, width)

    def _spline64(self, img, w, h, luma_from=None):
        return spline64(self.ctx, img, w, h, luma_from)

    # ---- vsslib/mcomb.py:125-192 ----
    def _combine(self, a, b):
        return combine_models(a, b, self.method, self.merge_weight, self.cmc_p, self.lmm_p, self.alm_p, self.crt_p, self.cmb_sw, self.device_index)

    def colorize_clip(self, clip):
        """HAVC_colorizer's graph on a clip: Spline64 squash -> deoldify / ddcolor -> combine -> Spline64 back + luma of the source
        (_clip_chroma_resize, __init__.py:3545-3554).  ndarray in -> ndarray out; DeviceImage in -> DeviceImage out (nothing leaves
        HBM; frames go through the models in batches of max_batch)."""
        clip, single = _as_clip(clip)
        n, h, w, _ = clip.shape
        dd_rf, fs = self.frame_size(w)
        host_in = not is_device(clip)
        dclip = DeviceImage.from_numpy(self.ctx, clip) if host_in else clip
        sq = dclip if (w, h) == (fs, fs) else self._spline64(dclip, fs, fs)
        a = b = None
        dd_size = math.trunc(dd_rf / 2) * 32                                                      # vsmodels.py:302
        if self._side_by_side() and is_device(sq):
            # The two models side by side on two contexts, DDColor driven from a second thread -- from the first clip on: building the models,
            # their nets and the tile autotuning from two host threads at once is serialised INSIDE the library (the set-up mutex of
            # csrc/havc_runtime.cpp; the reference's glue builds models from whichever worker thread asks first, vsslib/vsmodels.py:196-233).
            import concurrent.futures
            if self._pool is None:
                self._pool = concurrent.futures.ThreadPoolExecutor(max_workers=1)
            self.ctx.synchronize()                                                                # the squashed clip is complete: the other context may read it

            def branch():
                bctx = self._ddcolor_model(dd_size).rt.ctx
                try:
                    return self._ddcolor_branch(sq, dd_size, bctx)
                finally:
                    bctx.synchronize()                                                            # (only enqueued: this context must not run ahead of it)
            fut = self._pool.submit(branch)
            err = None
            try:
                a = self._deoldify_clip(sq)
            except BaseException as e:                                                            # noqa: BLE001 -- re-raised below, after the branch has drained
                err = e
            try:
                b = fut.result()          # ALWAYS waited for: the branch's stream reads `sq`, whose buffer returns to this context's pool when we leave
            except BaseException as e:                                                            # noqa: BLE001
                err = err or e
            if err is not None:
                raise err
        else:
            if self.method != 1:
                a = self._deoldify_clip(sq)
            if self.method != 0:
                b = self._ddcolor_branch(sq, dd_size, None)
        col = self._combine(a, b)
        out = self._spline64(col, w, h, luma_from=dclip)
        if host_in:
            out = out.numpy()
        return (out[0] if host_in else out.reshaped(out.shape[1:])) if single else out

    def colorize(self, frame):
        """one uint8 [h, w, 3] frame (or a clip) through the graph"""
        return self.colorize_clip(frame)

    __call__ = colorize_clip


def spline64(ctx, img, w, h, luma_from=None):
    """the harness stand-in of `resize.Spline64` (+ vs_recover_clip_luma when luma_from is given) on frames or clips,
    host or device operands"""
    dev = is_device(img) or is_device(luma_from)
    if dev:
        img = img if is_device(img) else DeviceImage.from_numpy(ctx, img)
        if luma_from is not None and not is_device(luma_from):
            luma_from = DeviceImage.from_numpy(ctx, luma_from)
    else:
        img = np.ascontiguousarray(img, dtype=np.uint8)
        luma_from = None if luma_from is None else np.ascontiguousarray(luma_from, dtype=np.uint8)
    n = img.shape[0] if img.ndim == 4 else 1
    sh, sw = img.shape[-3], img.shape[-2]
    shape = ((n, h, w, 3) if img.ndim == 4 else (h, w, 3))
    if luma_from is not None and tuple(luma_from.shape) != shape:
        raise ValueError("luma_from must have the output shape")
    out = DeviceImage(ctx, shape) if dev else np.empty(shape, np.uint8)
    from .device import operand_ptr
    nat.check(ctx.lib.havc_spline64_resize_n(ctx.h, operand_ptr(img), sw, sh, operand_ptr(out), w, h,
                                             operand_ptr(luma_from) if luma_from is not None else None, n), ctx.h)
    return out


def _per_frame(fn, *clips):
    """apply a frame-level function (frame statistics / neighbourhoods inside) to every frame of 4-D operands"""
    first = clips[0]
    n = first.shape[0]
    if is_device(first):
        out = first.empty_like()
        for i in range(n):
            out.frame(i).copy_from(fn(*[c.frame(i) for c in clips]))
        return out
    return np.stack([np.asarray(fn(*[c[i] for c in clips])) for i in range(n)])


def combine_models(a, b, method, w, cmc_p=DEF_CMC_p, lmm_p=DEF_LMM_p, alm_p=DEF_ALM_p, crt_p=DEF_CRT_p, invert_clips=False, device_index=0):
    """vs_sc_combine_models (vsslib/mcomb.py:125-192) on clips [n, h, w, 3] (ndarray or DeviceImage); sat / hue tweaks refused
    by the callers.  Purely per-pixel methods run on the whole stack in one launch, methods with frame-level decisions (mean
    luma, Laplacian) frame by frame."""
    if invert_clips:
        a, b = b, a
    if a is None or b is None:
        return a if b is None else b
    di = device_index
    big = len(cmc_p) > 1
    red_fix = cmc_p[1] if big else True
    rows = (lambda x: x.as_rows()) if is_device(a) else (lambda x: x.reshape((-1,) + x.shape[2:]))
    unrows = (lambda x: x.reshaped(a.shape)) if is_device(a) else (lambda x: x.reshape(a.shape))
    if method == 2:
        return unrows(mcomb.simple_merge(rows(a), rows(b), w, di))
    if method == 3:
        ccm = _per_frame(lambda x, y: mcomb.constrained_chroma_merge(x, y, w, cmc_p[0], red_fix, di), a, b)
        mm = mcomb.simple_merge(rows(a), rows(b), min(w, 0.6), di)
        return unrows(mcomb.simple_merge(rows(ccm), mm, 0.3, di))
    if method == 4:
        return unrows(mcomb.luma_masked_merge(rows(a), rows(b), None, lmm_p[0], lmm_p[1], w, di))
    if method == 5:
        return _per_frame(lambda x, y: mcomb.adaptive_luma_merge(x, y, alm_p[0], alm_p[1], w, alm_p[2], di), a, b)
    if method == 6:
        if crt_p[3]:
            raise NotImplementedError("ChromaRetentionMerge(chroma_resize=True) is a VapourSynth-level resize round trip")
        restored = _per_frame(lambda x, y: mcomb.chroma_retention_frame(x, y, crt_p[0], crt_p[1], crt_p[4], crt_p[2], False, crt_p[5], di), a, b)
        return unrows(mcomb.simple_merge(rows(a), rows(restored), w, di))   # vs_simple_merge = std.Merge in the reference (VapourSynth core)
    if method == 7:
        return _per_frame(lambda x, y: mcomb.chroma_bound_adaptive_merge(x, y, red_fix, cmc_p[2] if big else 20, cmc_p[3] if big else 24, w, di), a, b)
    raise HAVCError("HAVC: only dd_method in (0,6) is supported")


# ======================================================================================================================
# function-shaped API (argument lists of the reference)
# ======================================================================================================================
_colorizers = {}


def _refuse_vs_only(ddtweak, sc_threshold, sc_min_freq):
    if sc_threshold and sc_threshold > 0 or sc_min_freq and sc_min_freq > 0:
        raise NotImplementedError("scene detection (sc_threshold / sc_min_freq) is VapourSynth glue (SCDetect): not in this harness")


def HAVC_colorizer(clip, method=2, mweight=0.4, deoldify_p=(0, 24, 1.0, 0.0), ddcolor_p=(1, 24, 1.0, 0.0, True), ddtweak=(False, False, False),
                   ddtweak_p=(DEF_TWEAK_p, "300:360|0.8,0.1"), cmc_p=DEF_CMC_p, lmm_p=DEF_LMM_p, alm_p=DEF_ALM_p, crt_p=DEF_CRT_p, cmb_sw=False,
                   sc_threshold=0.0, sc_tht_offset=1, sc_min_freq=0, sc_tht_ssim=0.0, sc_normalize=False, sc_min_int=1, sc_tht_white=DEF_THT_WHITE,
                   sc_tht_black=DEF_THT_BLACK, device_index=0, torch_dir=None, debug_level=0, **harness):
    """vsdeoldify/__init__.py:2290-2298.  `harness` = keyword-only extras of this library: state_dicts / ddcolor_state_dict /
    zhang_state_dict (seeded weights instead of files under torch_dir), ddcolor_model_dir, max_batch, ddcolor_kwargs, precision ("fast" / "precise")."""
    if clip is None or not (is_device(clip) or isinstance(clip, np.ndarray)):
        raise HAVCError("HAVC_colorizer: this is not a clip")                                     # __init__.py:2437-2438
    _refuse_vs_only(ddtweak, sc_threshold, sc_min_freq)
    cmc = list(cmc_p) if isinstance(cmc_p, (list, tuple)) else [cmc_p]
    flags = tuple(ddtweak) if isinstance(ddtweak, (list, tuple)) else (ddtweak, False, False)
    key = (method, mweight, tuple(deoldify_p), tuple(ddcolor_p), tuple(cmc), tuple(lmm_p), tuple(alm_p), tuple(crt_p), cmb_sw, device_index,
           torch_dir, id(harness.get("state_dicts")), id(harness.get("ddcolor_state_dict")), harness.get("max_batch", 1), flags, repr(ddtweak_p),
           harness.get("precision") or os.environ.get("HAVC_PRECISION") or DEFAULT_PRECISION)
    col = _colorizers.get(key)
    if col is None:
        col = HAVCFrameColorizer(method, mweight, deoldify_p, ddcolor_p, cmc, lmm_p, alm_p, crt_p, cmb_sw, device_index, package_dir=torch_dir,
                                 ddtweak=flags, ddtweak_p=ddtweak_p, **harness)
        _colorizers.clear()                     # one live graph at a time: the nets hold GBs of activations
        _colorizers[key] = col
    return col.colorize_clip(clip)


def HAVC_ddeoldify(clip, method=2, mweight=0.4, deoldify_p=(0, 24, 1.0, 0.0), ddcolor_p=(1, 24, 1.0, 0.0, True), ddtweak=False,
                   ddtweak_p=(DEF_TWEAK_p, "300:360|0.8,0.1"), cmc_tresh=0.2, lmm_p=(0.2, 0.8, 1.0), alm_p=(0.8, 1.0, 0.15), cmb_sw=False,
                   sc_threshold=0.0, sc_tht_offset=1, sc_min_freq=0, sc_tht_ssim=0.0, sc_normalize=False, sc_min_int=1, sc_tht_white=DEF_THT_WHITE,
                   sc_tht_black=DEF_THT_BLACK, device_index=0, torch_dir=None, sc_debug=False, **harness):
    """deprecated wrapper, vsdeoldify/__init__.py:3612-3628: the same call with cmc_p = [cmc_tresh] and DEF_CRT_p"""
    import warnings
    warnings.warn("Warning: HAVC_ddeoldify is deprecated and may be removed in the future, please use 'HAVC_colorizer' instead.", DeprecationWarning)
    return HAVC_colorizer(clip, method, mweight, deoldify_p, ddcolor_p, [ddtweak, False, False], ddtweak_p, [cmc_tresh], lmm_p, alm_p, DEF_CRT_p, cmb_sw,
                          sc_threshold, sc_tht_offset, sc_min_freq, sc_tht_ssim, sc_normalize, sc_min_int, sc_tht_white, sc_tht_black, device_index,
                          torch_dir, 1 if sc_debug else 0, **harness)


def ddeoldify(clip, method=2, mweight=0.4, deoldify_p=(0, 24, 1.0, 0.0), ddcolor_p=(1, 24, 1.0, 0.0, True), dotweak=False,
              dotweak_p=(0.0, 1.0, 1.0, False, 0.2, 0.5, 1.5, 0.5), ddtweak=False, ddtweak_p=(DEF_TWEAK_p, "300:360|0.8,0.1"), degrain_strength=0,
              cmc_tresh=0.2, lmm_p=(0.2, 0.8, 1.0), alm_p=(0.8, 1.0, 0.15), cmb_sw=False, device_index=0, torch_dir=None, **harness):
    """deprecated wrapper, vsdeoldify/__init__.py:3642-3653 (dotweak / degrain_strength are accepted and ignored, as there)"""
    import warnings
    warnings.warn("Warning: ddeoldify is deprecated and may be removed in the future, please use 'HAVC_colorizer' instead.", DeprecationWarning)
    return HAVC_colorizer(clip, method, mweight, deoldify_p, ddcolor_p, [ddtweak, False, False], ddtweak_p, [cmc_tresh], lmm_p, alm_p, DEF_CRT_p, cmb_sw,
                          sc_threshold=0, sc_min_freq=0, device_index=device_index, torch_dir=torch_dir, **harness)


def _clip_chroma_resize(clip_hires, clip_lowres, device_index=0):
    """__init__.py:3545-3554: Spline64 to the hi-res size, then vs_recover_clip_luma (luma of clip_hires, chroma of the resized)"""
    hi, _ = _as_clip(clip_hires)
    lo, _ = _as_clip(clip_lowres)
    return spline64(get_context(device_index), lo, hi.shape[2], hi.shape[1], luma_from=hi)


def HAVC_merge(clipa=None, clipb=None, clip_luma=None, weight=0.5, method=2, cmc_p=DEF_CMC_p, lmm_p=DEF_LMM_p, alm_p=DEF_ALM_p, crt_p=DEF_CRT_p,
               device_index=0):
    """vsdeoldify/__init__.py:2536-2675: the HAVC merge methods on two already coloured clips (+ optional luma source)."""
    for name, c in (("clipa", clipa), ("clipb", clipb), ("clip_luma", clip_luma)):
        if c is not None and not (is_device(c) or isinstance(c, np.ndarray)):
            raise HAVCError(f"HAVC_merge: this is not a clip: {name}")                           # __init__.py:2631-2638
    single = (clipa if clipa is not None else clipb).ndim == 3

    def done(x):
        if single and x.ndim == 4:
            return x.reshaped(x.shape[1:]) if is_device(x) else x[0]
        return x
    if method == 0 or weight == 0:                                                                # __init__.py:2640-2645
        return done(_clip_chroma_resize(clip_luma, clipa, device_index)) if clip_luma is not None else clipa
    if method == 1 or weight == 1:                                                                # __init__.py:2647-2652
        return done(_clip_chroma_resize(clip_luma, clipb, device_index)) if clip_luma is not None else clipb
    a, _ = _as_clip(clipa)
    b, _ = _as_clip(clipb)
    if is_device(a) != is_device(b):
        ctx = get_context(device_index)
        a = a if is_device(a) else DeviceImage.from_numpy(ctx, a)
        b = b if is_device(b) else DeviceImage.from_numpy(ctx, b)
    if method == 2:                                                                               # __init__.py:2659-2661
        return done(combine_models(a, b, 2, weight, device_index=device_index))
    if clip_luma is not None:                                                                     # __init__.py:2663-2667
        luma, _ = _as_clip(clip_luma)
        rf = min(max(math.trunc(0.4 * luma.shape[2] / 16), 16), 32)
        fs = min(rf * 16, luma.shape[2])
        ctx = get_context(device_index)
        a, b = spline64(ctx, a, fs, fs), spline64(ctx, b, fs, fs)
    cmc = list(cmc_p) if isinstance(cmc_p, (list, tuple)) else [cmc_p]
    merged = combine_models(a, b, method, weight, cmc, lmm_p, alm_p, crt_p, False, device_index)  # vs_combine_models(sat=[1,1], hue=[0,0])
    if clip_luma is not None:                                                                     # __init__.py:2673-2676
        merged = _clip_chroma_resize(clip_luma, merged, device_index)
    return done(merged)
