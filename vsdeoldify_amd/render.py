"""Host-side mirror of the reference's DeOldify adapters, backed by libhavc_mi355.so.

  ModelImageRender        <- vsdeoldify/deoldify/visualize.py:41-137
  (ColorizerFilter flow)  <- vsdeoldify/deoldify/filters.py:23-124
Same constructor / method names, argument meaning and error behaviour; images are PIL RGB in, PIL RGB
out (new object, input untouched).  Weights are read from
`package_dir/models/Colorize{Video,Stable,Artistic}_gen.pth` exactly like gen_inference_wide/deep
(deoldify/generators.py:12-21,85-95) unless `state_dicts` are injected (tests / bench: seeded
synthetic weights).  There is no CPU path: construction raises if the HIP library or GPU is missing.
"""
import logging
import os
import threading

import numpy as np

from . import _native as nat
from .precision import DEFAULT_PRECISION, resolve as resolve_precision
from .deoldify_net import DeoldifyGenerator

WEIGHTS = {"video": ("ColorizeVideo_gen", "wide"), "stable": ("ColorizeStable_gen", "wide"),
           "artistic": ("ColorizeArtistic_gen", "deep")}
RENDER_BASE = 16                      # ColorizerFilter.render_base, deoldify/filters.py:79

_contexts = {}


def get_context(device_index=0, worker=0):
    """One havc context per (GPU, worker), process-wide, like device.set(DeviceId(n)) (deoldify/_device.py:21-30).
    A context serialises its calls behind one mutex and owns its streams, workspace and activation arenas, so VapourSynth worker
    threads that should colour frames CONCURRENTLY on one GPU take one context each (`worker` = 0, 1, ...).  Packed weights are
    shared between the workers of a device (`_shared_weights`): only the activations (~1.1 GB per frame at 560x560) are per worker."""
    if device_index == 99:
        raise nat.NativeLibraryError("device_index=99 (CPU) is not supported by vsdeoldify_amd: MI355X only")
    key = (device_index, worker)
    if key not in _contexts:
        _contexts[key] = nat.Context(device_index)
    return _contexts[key]


_weights_cache = {}          # (device, arch, fusion flags, id of the state dict / path) -> (generator, device weights)


def _shared_weights(ctx, key, make_generator):
    """pack + upload a model once per device; every worker context of that device builds its nets on the same blob (read-only)"""
    full = (ctx.device_id,) + key
    if full not in _weights_cache:
        gen = make_generator()
        _weights_cache[full] = (gen, nat.Weights(ctx, gen.blob))
    return _weights_cache[full]


def _load_pth(path):
    import torch  # plumbing only: deserialise the checkpoint (fastai/basic_train.py:270-283)
    state = torch.load(path, map_location="cpu", weights_only=False)
    return state["model"] if isinstance(state, dict) and set(state.keys()) == {"model", "opt"} else state


class GeneratorRuntime:
    """Packed weights on one GPU + a cache of nets keyed by (render size, max_batch)."""

    def __init__(self, ctx, state_dict, arch, fuse_final=True, fuse_blur=True, generator=None, share_key=None, precision="fast"):
        self.ctx, self.arch = ctx, arch
        if share_key is not None:              # worker contexts of one device: one packed blob for all of them
            self.gen, self.weights = _shared_weights(ctx, (arch, fuse_final, fuse_blur) + tuple(share_key) + ((precision,) if precision != "fast" else ()),
                                                     lambda: generator or DeoldifyGenerator(state_dict, arch, fuse_final=fuse_final, fuse_blur=fuse_blur,
                                                                                            precision=precision))
            self._owns_weights = False
        else:
            self.gen = generator or DeoldifyGenerator(state_dict, arch, fuse_final=fuse_final, fuse_blur=fuse_blur, precision=precision)
            self.weights = nat.Weights(ctx, self.gen.blob)
            self._owns_weights = True
        self.nets = {}

    def net(self, S, max_batch=1, low_latency=False):
        """low_latency: the split-K plan for nets that run max_batch (<= 2) frames per launch (DeoldifyGenerator.plan)"""
        low_latency = bool(low_latency) and max_batch <= 2 and not self.gen.precise
        key = (S, max_batch, True) if low_latency else (S, max_batch)
        if key not in self.nets:
            ops, bufs, i, o, names = self.gen.plan(S, split_for_frames=max_batch) if low_latency else self.gen.plan(S)
            n = nat.Net(self.ctx, self.weights, ops, bufs, i, o, S, max_batch)
            n.names = names
            if os.environ.get("HAVC_AUTOTUNE", "1") != "0":
                n.autotune(max_batch)          # per-op conv tile configuration by measurement: same bytes, ~1 s once per net
            self.nets[key] = n
        return self.nets[key]

    def close(self):
        for n in self.nets.values():
            n.close()
        self.nets.clear()
        if self._owns_weights:
            self.weights.close()


_batcher_lock = threading.Lock()


class ModelImageRender:
    """Drop-in for vsdeoldify.deoldify.visualize.ModelImageRender."""

    def __init__(self, package_dir=None, modelname="video", render_factor=24, video_weight=0, device_index=0,
                 state_dicts=None, max_batch=1, worker=0, coalesce=0, precision=None, low_latency=None):
        """`precision`: "fast" (default: fp16 activations and MFMA operands, fp32 accumulation: CIEDE2000 against the reference's fp32 path small
        in the mean but p99 1.2 - 2.3 on the final image) or "precise" (fp32-class arithmetic like the reference,
        deoldify/filters.py:45-68: hi / lo fp16 pairs on the same MFMA kernels, 3x the matrix work); None reads HAVC_PRECISION, then the package default "precise" (precision.py).
        `low_latency` (None reads HAVC_LOW_LATENCY, default off): a render that colours ONE frame per call (max_batch <= 2: the reference's call
        shape, vsslib/vsmodels.py:219-230) builds its nets with split-K convs -- a single frame gives most layers 5 - 40 tiles for 256 CUs; bytes
        differ from the batched nets in fp32 summation order only.  Round 5 measured that difference on three weight sets (tests/test_gpu_deoldify.py):
        <= 2 LSB at render factors 6 / 10, but 3 LSB on isolated bytes at the headline size (rf 35, 89.5 % of the bytes equal) -- above the 2-LSB bar
        VERDICT r4 set for making it the default, so it stays opt-in.
        `worker`: index of the per-thread context on this GPU (get_context): renders built with different worker indices run
        concurrently from different threads and share the packed weights.
        `coalesce` = N > 0: ONE render shared by N caller threads (the reference's per-frame call shape under VapourSynth's thread pool):
        their concurrent get_transformed_image calls for frames already at the render size are merged into batches of up to
        max(max_batch, N) frames (havc_batcher); results are those of separate calls."""
        self.package_dir = package_dir
        self._modelname = modelname
        self._video_weight = video_weight
        self._render_factor = render_factor
        self._max_batch = max(max_batch, coalesce)
        self._worker = worker
        self._coalesce = coalesce
        self._batchers = {}
        self._low_latency = (os.environ.get("HAVC_LOW_LATENCY", "0") != "0") if low_latency is None else bool(low_latency)
        self._precision = resolve_precision(precision)           # explicit > HAVC_PRECISION > "precise" (vsdeoldify_amd/precision.py)
        self.ctx = get_context(device_index, worker)
        second = None if modelname == "video" else ("stable" if modelname == "stable" else "artistic")
        self._video = self._runtime("video", state_dicts)
        self._second = self._runtime(second, state_dicts) if second else None

    def _runtime(self, which, state_dicts):
        name, arch = WEIGHTS[which]
        if state_dicts is not None and which in state_dicts:
            sd = state_dicts[which]
            if self._worker or os.environ.get("HAVC_SHARE_WEIGHTS", "0") != "0":
                return GeneratorRuntime(self.ctx, sd, arch, share_key=("sd", id(sd)), precision=self._precision)
        else:
            path = os.path.join(str(self.package_dir), "models", name + ".pth")      # Learner.load path
            packed = os.path.splitext(path)[0] + ".havc"                             # tools/convert_weights.py output, if newer
            if self._precision != "fast":                                            # the packed files hold the fast layout: precise packs from the .pth
                if not os.path.isfile(path) or os.path.getsize(path) == 0:
                    raise FileNotFoundError(f"DeOldify weights not found: {path}")
                return GeneratorRuntime(self.ctx, _load_pth(path), arch, share_key=("file", path, os.path.getmtime(path)), precision=self._precision)
            if os.path.isfile(packed) and (not os.path.isfile(path) or os.path.getmtime(packed) >= os.path.getmtime(path)):
                key = ("file", packed, os.path.getmtime(packed))
                if (self.ctx.device_id, arch, True, True) + key in _weights_cache:
                    return GeneratorRuntime(self.ctx, None, arch, share_key=key)
                return GeneratorRuntime(self.ctx, None, arch, generator=DeoldifyGenerator.load(packed), share_key=key)
            if not os.path.isfile(path) or os.path.getsize(path) == 0:
                raise FileNotFoundError(f"DeOldify weights not found: {path}")
            key = ("file", path, os.path.getmtime(path))
            if (self.ctx.device_id, arch, True, True) + key in _weights_cache:
                return GeneratorRuntime(self.ctx, None, arch, share_key=key)
            return GeneratorRuntime(self.ctx, _load_pth(path), arch, share_key=key)
        return GeneratorRuntime(self.ctx, sd, arch, precision=self._precision)

    # -- raw batched entry (frames already S x S, uint8 [n,S,S,3]) ------------------------------
    def render_square_batch(self, frames, post_process=True):
        """uint8 [n, S, S, 3] (ndarray, or a device.DeviceImage: then nothing leaves HBM and the call does not block)"""
        from .device import DeviceImage, is_device, operand_ptr
        dev = is_device(frames)
        if not dev:
            frames = np.ascontiguousarray(frames, dtype=np.uint8)
        n, S = frames.shape[0], frames.shape[1]
        assert tuple(frames.shape[1:]) == (S, S, 3) and S == self._render_factor * RENDER_BASE
        v = self._video.net(S, self._max_batch, self._low_latency)
        s = self._second.net(S, self._max_batch, self._low_latency) if self._second else None
        out = DeviceImage(self.ctx, frames.shape) if dev else np.empty_like(frames)
        nat.check(self.ctx.lib.havc_deoldify_frames(self.ctx.h, v.h, s.h if s else None, float(self._video_weight),
                                                    1 if post_process else 0, operand_ptr(frames), operand_ptr(out), n),
                  self.ctx.h)
        return out

    def get_transformed_planes(self, planes_in, planes_out=None, post_process=True):
        """One VapourSynth RGB24 frame given as its three planes (2-D uint8 arrays with any row stride, S x S) -> three planes:
        frame_to_image + get_transformed_image + image_to_frame (vsslib/vsmodels.py:214-230, vsslib/vsutils.py:60-110) in one
        call, the plane <-> interleaved shuffles on the GPU."""
        import ctypes as C
        S = self._render_factor * RENDER_BASE
        pl = [np.asarray(p) for p in planes_in]
        if len(pl) != 3 or any(p.shape != (S, S) or p.dtype != np.uint8 or p.strides[1] != 1 for p in pl) or len({p.strides[0] for p in pl}) != 1:
            raise ValueError("planes must be three uint8 [S, S] arrays with unit pixel stride and one common row stride")
        if planes_out is None:
            planes_out = [np.empty((S, S), np.uint8) for _ in range(3)]
        po = [np.asarray(p) for p in planes_out]
        if any(p.shape != (S, S) or p.dtype != np.uint8 or p.strides[1] != 1 or not p.flags.writeable for p in po) or len({p.strides[0] for p in po}) != 1:
            raise ValueError("output planes must be three writable uint8 [S, S] arrays with one common row stride")
        v = self._video.net(S, self._max_batch, self._low_latency)
        s = self._second.net(S, self._max_batch, self._low_latency) if self._second else None
        pin = (C.c_void_p * 3)(*[p.ctypes.data for p in pl])
        pout = (C.c_void_p * 3)(*[p.ctypes.data for p in po])
        nat.check(self.ctx.lib.havc_deoldify_frame_planar(self.ctx.h, v.h, s.h if s else None, float(self._video_weight), 1 if post_process else 0,
                                                          pin, pl[0].strides[0], pout, po[0].strides[0]), self.ctx.h)
        return planes_out

    def get_transformed_image(self, img_orig, post_process=True):
        """PIL RGB in -> PIL RGB out, same size (visualize.py:118-137).  Frames already at the render size
        (the HAVC_colorizer flow, __init__.py:2502-2506) run entirely on the GPU; other sizes use Pillow's own
        BILINEAR stretch on the host exactly where the reference does (filters.py:37-41,70-73)."""
        from PIL import Image
        S = self._render_factor * RENDER_BASE
        img_orig = img_orig.convert("RGB") if img_orig.mode != "RGB" else img_orig
        if img_orig.size == (S, S):
            try:
                if self._coalesce:
                    return Image.fromarray(self._batcher(S, post_process).submit(np.asarray(img_orig)))
                out = self.render_square_batch(np.asarray(img_orig)[None], post_process)[0]
                return Image.fromarray(out)
            except nat.HavcOutOfMemory:
                pass                                 # fall through to the per-model path below, which reproduces the reference's OOM flow
        sq = np.asarray(img_orig.resize((S, S), resample=Image.BILINEAR))
        raw_v, raw_s = self._raw_colors(sq)
        outs = []
        for raw in (raw_v, raw_s):
            if raw is None:
                continue
            col = np.asarray(Image.fromarray(raw).resize(img_orig.size, resample=Image.BILINEAR))
            if post_process:
                from .imfilters import chroma_post_process_np
                col = chroma_post_process_np(self.ctx, col, np.asarray(img_orig))
            outs.append(col)
        if len(outs) == 1:
            return Image.fromarray(outs[0])
        from .imfilters import blend_np
        return Image.fromarray(blend_np(self.ctx, outs[1], outs[0], self._video_weight))

    def _batcher(self, S, post_process):
        key = (S, bool(post_process))
        b = self._batchers.get(key)
        if b is None:
            with _batcher_lock:
                b = self._batchers.get(key)
                if b is None:
                    v = self._video.net(S, self._max_batch, self._low_latency)
                    s = self._second.net(S, self._max_batch, self._low_latency) if self._second else None
                    b = self._batchers[key] = nat.Batcher(self.ctx, v, s, self._video_weight, post_process, callers=self._coalesce,
                                                          wait_us=int(os.environ.get("HAVC_COALESCE_WAIT_US", "300")))
        return b

    def _raw_colors(self, sq):
        S = sq.shape[0]
        outs = []
        for rt in (self._video, self._second):
            if rt is None:
                outs.append(None)
                continue
            try:
                n = rt.net(S, self._max_batch, self._low_latency)
                o = np.empty_like(sq[None])
                nat.check(self.ctx.lib.havc_deoldify_frames(self.ctx.h, n.h, None, 0.0, 0, nat.as_ptr(np.ascontiguousarray(sq[None])),
                                                            nat.as_ptr(o), 1), self.ctx.h)
                outs.append(o[0])
            except nat.HavcOutOfMemory:
                # deoldify/filters.py:55-63: on OOM _model_process warns and returns the (squared, gray) model image, and
                # filter() CONTINUES with it: un-square to the source size, post-process, blend -- the caller still gets an
                # image of img_orig.size
                from PIL import Image
                logging.warning("Warning: render_factor was set too high, and out of memory error resulted. "
                                "Returning original image.")
                outs.append(np.asarray(Image.fromarray(sq).convert("LA").convert("RGB")))
        return outs
