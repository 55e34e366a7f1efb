"""`HAVC_colorizer` without VapourSynth: the parameter normalisation, frame-size rule, model routing and
combine-method dispatch of vsdeoldify/__init__.py:2290-2523 and vsslib/mcomb.py:125-192, applied to one uint8 HWC frame at a time
and executed on the MI355X.  This is the harness counterpart of the VapourSynth graph the reference assembles; what only
VapourSynth can do stays there and is refused here instead of being approximated: `vs_tweak` (deoldify / ddcolor sat / hue other than
1 / 0, `luma_mask_sat` < 1), scene detection, the DDColor pre-tweaks.  zimg's Spline64 is replaced by the library's own Spline64
(outside the parity contract).

    col = HAVCFrameColorizer(method=2, mweight=0.4, package_dir=..., ddcolor_model_dir=...)
    out = col.colorize(frame)            # frame: uint8 [H, W, 3], any size; out has the same size
"""
import math

import numpy as np

from . import _native as nat
from . import imfilters as F
from . import mcomb
from .render import get_context

DEF_CMC_p = [0.15, True, 20, 24]            # vsslib/constants.py:19-22
DEF_LMM_p = [0.15, 0.65, 1.0]
DEF_ALM_p = [0.8, 1.0, 0.15]
DEF_CRT_p = [0.8, 30, 2, False, 0, 0]
DEF_STABLE_WEIGHT = DEF_ARTISTIC_WEIGHT = 0.5  # deoldify/constants? vsslib/constants.py:56-57


class HAVCError(ValueError):
    """what the reference raises as vs.Error / HAVC_LogMessage(EXCEPTION)"""


class HAVCFrameColorizer:
    def __init__(self, method=2, mweight=0.4, deoldify_p=(0, 24, 1.0, 0.0), ddcolor_p=(1, 24, 1.0, 0.0, True), cmc_p=DEF_CMC_p,
                 lmm_p=DEF_LMM_p, alm_p=DEF_ALM_p, crt_p=DEF_CRT_p, cmb_sw=False, device_index=0, package_dir=None,
                 ddcolor_model_dir=None, state_dicts=None, ddcolor_state_dict=None, zhang_state_dict=None):
        # ---- __init__.py:2452-2462: method <-> merge weight normalisation ----
        merge_weight = 0.0 if method == 0 else (1.0 if method == 1 else mweight)
        if merge_weight == 0.0:
            method = 0
        elif merge_weight == 1.0:
            method = 1
        if method not in range(0, 8):
            raise HAVCError("HAVC: only dd_method in (0,6) is supported")                        # mcomb.py:192
        self.method, self.merge_weight, self.cmb_sw = method, merge_weight, cmb_sw
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
        self._package_dir, self._dd_dir = package_dir, ddcolor_model_dir
        self._sds, self._dd_sd, self._zh_sd = state_dicts, ddcolor_state_dict, zhang_state_dict
        self._deoldify = self._ddcolor = None
        self._dd_size = None

    # ---- model routing: vsslib/vsmodels.py:196-213 (deoldify), :290-350 (ddcolor / zhang) ----
    def _deoldify_render(self):
        if self._deoldify is None:
            from .render import ModelImageRender
            name, w = {0: ("video", 0), 1: ("stable", DEF_STABLE_WEIGHT), 2: ("artistic", DEF_ARTISTIC_WEIGHT)}.get(self.deoldify_model, ("video", 0))
            self._deoldify = ModelImageRender(self._package_dir, name, self.deoldify_rf, video_weight=w, device_index=self.device_index,
                                              state_dicts=self._sds)
        return self._deoldify

    def _ddcolor_frame(self, sq, input_size):
        if self.ddcolor_model in (0, 1):
            if self._ddcolor is None or self._dd_size != input_size:
                from .ddcolor import DDColorRender
                self._ddcolor = DDColorRender(self.ddcolor_model, input_size, self.device_index, state_dict=self._dd_sd, model_dir=self._dd_dir)
                self._dd_size = input_size
            return self._ddcolor.colorize_frame(sq)
        from .colorization import ModelColorization                                               # vsmodels.py:346-350
        mc = ModelColorization("siggraph17" if self.ddcolor_model == 2 else "eccv16", True, self.device_index, state_dict=self._zh_sd)
        return mc.colorize_frame(sq)

    def frame_size(self, width):
        """__init__.py:2490-2502."""
        dd_rf = self.ddcolor_rf or min(max(math.trunc(0.4 * width / 16), 16), 32)
        return dd_rf, min(max(dd_rf, self.deoldify_rf) * 16, width)

    def _spline64(self, img, w, h, luma_from=None):
        img = np.ascontiguousarray(img, dtype=np.uint8)
        out = np.empty((h, w, 3), np.uint8)
        lf = None if luma_from is None else np.ascontiguousarray(luma_from, dtype=np.uint8)
        nat.check(self.ctx.lib.havc_spline64_resize(self.ctx.h, nat.as_ptr(img), img.shape[1], img.shape[0], nat.as_ptr(out), w, h,
                                                    nat.as_ptr(lf) if lf is not None else None), self.ctx.h)
        return out

    # ---- vsslib/mcomb.py:125-192 ----
    def _combine(self, a, b):
        if self.cmb_sw:
            a, b = b, a
        if a is None or b is None:
            return a if b is None else b
        m, w = self.method, self.merge_weight
        di = self.device_index
        if m == 2:
            return mcomb.simple_merge(a, b, w, di)
        if m == 3:
            ccm = mcomb.constrained_chroma_merge(a, b, w, self.cmc_p[0], self.cmc_p[1] if len(self.cmc_p) > 1 else True, di)
            mm = mcomb.simple_merge(a, b, min(w, 0.6), di)
            return mcomb.simple_merge(ccm, mm, 0.3, di)
        if m == 4:
            return mcomb.luma_masked_merge(a, b, None, self.lmm_p[0], self.lmm_p[1], w, di)
        if m == 5:
            return mcomb.adaptive_luma_merge(a, b, self.alm_p[0], self.alm_p[1], w, self.alm_p[2], di)
        if m == 6:
            if self.crt_p[3]:
                raise NotImplementedError("ChromaRetentionMerge(chroma_resize=True) is a VapourSynth-level resize round trip")
            restored = mcomb.chroma_retention_frame(a, b, self.crt_p[0], self.crt_p[1], self.crt_p[4], self.crt_p[2], False, self.crt_p[5], di)
            return mcomb.simple_merge(a, restored, w, di)             # vs_simple_merge = std.Merge in the reference (VapourSynth core)
        big = len(self.cmc_p) > 1
        return mcomb.chroma_bound_adaptive_merge(a, b, self.cmc_p[1] if big else True, self.cmc_p[2] if big else 20, self.cmc_p[3] if big else 24, w, di)

    def colorize(self, frame):
        """one frame through HAVC_colorizer's graph: squash -> deoldify / ddcolor -> combine -> Spline64 back + luma of the source"""
        from PIL import Image
        frame = np.ascontiguousarray(frame, dtype=np.uint8)
        if frame.ndim != 3 or frame.shape[2] != 3:
            raise HAVCError("HAVC_colorizer: only RGB24 frames")
        h, w = frame.shape[:2]
        dd_rf, fs = self.frame_size(w)
        sq = frame if (w, h) == (fs, fs) else self._spline64(frame, fs, fs)
        a = b = None
        if self.method != 1:
            a = np.asarray(self._deoldify_render().get_transformed_image(Image.fromarray(sq)))
        if self.method != 0:
            b = self._ddcolor_frame(sq, math.trunc(dd_rf / 2) * 32)                               # vsmodels.py:302
        col = self._combine(a, b)
        return self._spline64(col, w, h, luma_from=frame)                                         # _clip_chroma_resize, __init__.py:3545-3554
