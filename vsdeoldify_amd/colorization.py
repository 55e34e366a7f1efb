"""Host-side mirror of vsdeoldify/colorization/__init__.py (`ModelColorization`), backed by libhavc_mi355.so.

Same singleton semantics (colorization/__init__.py:39-54: one process-wide instance, switching `model` re-initialises),
same `colorize_frame(frame_i: np.ndarray HWC u8) -> np.ndarray HWC u8` (network input fixed at 256x256,
colorization/__init__.py:81).  Weights: the reference downloads them with model_zoo.load_url
(colorizers/eccv16.py:101-107, siggraph17.py:164-170); here they are read from the torch hub checkpoint cache
(`$TORCH_HOME/hub/checkpoints/<file>.pth`) or injected as a state dict (tests / offline).  No CPU path.
"""
import os
import threading

import numpy as np

from . import _native as nat
from .precision import DEFAULT_PRECISION, resolve as resolve_precision
from .render import get_context
from .zhang_net import ZhangGenerator

CHECKPOINTS = {"eccv16": "colorization_release_v2-9b330a0b.pth", "siggraph17": "siggraph17-df00044c.pth"}
NET_SIZE = 256


_batcher_lock = threading.Lock()


class ModelColorization:
    _instance = None
    _initialized = False

    def __new__(cls, *args, **kwargs):
        if cls._instance is None:
            cls._instance = super().__new__(cls)
        return cls._instance

    def __init__(self, model="siggraph17", use_gpu=True, device_index=0, state_dict=None, max_batch=1, coalesce=0, precision=None):
        """coalesce = N > 0: colorize_frame calls made concurrently by N threads are merged into batches (havc_batcher, kind 2).
        precision: "fast" (fp16 activations, fp32 accumulation) or "precise" (fp32-class arithmetic like the reference, which runs the nets in fp32,
        colorization/__init__.py:76-95: hi / lo fp16 pairs, three-segment convs, fp32 softmax / tanh projection); None reads HAVC_PRECISION, then the package default "precise" (precision.py)."""
        if not use_gpu:
            raise nat.NativeLibraryError("vsdeoldify_amd.ModelColorization is MI355X only (use_gpu=False is not supported)")
        precision = resolve_precision(precision)                  # explicit > HAVC_PRECISION > "precise" (vsdeoldify_amd/precision.py)
        if self._initialized and self.colorizer_model == model and state_dict is None and getattr(self, "precision", "fast") == precision:
            return
        if self._initialized:
            self.close()
        self.colorizer_model, self.use_gpu, self.precision = model, use_gpu, precision
        self.ctx = get_context(device_index)
        self._coalesce, self._batchers = coalesce, {}
        self._colorize_init(state_dict, max(max_batch, coalesce))
        self._initialized = True

    def _colorize_init(self, state_dict, max_batch):
        if state_dict is None:
            import torch
            hub = os.path.join(os.environ.get("TORCH_HOME", os.path.expanduser("~/.cache/torch")), "hub", "checkpoints")
            path = os.path.join(hub, CHECKPOINTS["siggraph17" if self.colorizer_model == "siggraph17" else "eccv16"])
            if not os.path.isfile(path):
                raise FileNotFoundError(f"Zhang colorizer weights not found: {path} (the reference fetches them with model_zoo)")
            state_dict = torch.load(path, map_location="cpu")
        self.gen = ZhangGenerator(state_dict, "siggraph17" if self.colorizer_model == "siggraph17" else "eccv16", precision=self.precision)
        self.weights = nat.Weights(self.ctx, self.gen.blob)
        ops, bufs, i, o, names = self.gen.plan(NET_SIZE)
        self.net = nat.Net(self.ctx, self.weights, ops, bufs, i, o, NET_SIZE, max_batch)
        self.net.names = names
        if os.environ.get("HAVC_AUTOTUNE", "1") != "0":
            self.net.autotune(max_batch)

    def close(self):
        for b in getattr(self, "_batchers", {}).values():
            b.close()
        self._batchers = {}
        if getattr(self, "net", None):
            self.net.close()
            self.weights.close()
            self.net = None
        type(self)._initialized = False

    def colorize_frames(self, frames):
        """uint8 [n,H,W,3] -> uint8 [n,H,W,3] (ndarray, or a device.DeviceImage: then nothing leaves HBM and the call only enqueues)."""
        from .device import is_device, operand_ptr
        dev = is_device(frames)
        if not dev:
            frames = np.ascontiguousarray(frames, dtype=np.uint8)
        n, h, w, _ = frames.shape
        from .device import DeviceImage
        out = DeviceImage(self.ctx, frames.shape) if dev else np.empty_like(frames)
        nat.check(self.ctx.lib.havc_zhang_frames(self.ctx.h, self.net.h, operand_ptr(frames), operand_ptr(out), n, w, h), self.ctx.h)
        return out

    def colorize_frame(self, frame_i=None):
        img = np.asarray(frame_i)
        if img.ndim == 2:                                   # load_img_rgb, colorizers/util.py:15-18
            img = np.tile(img[:, :, None], 3)
        if self._coalesce:
            key = img.shape[:2]
            b = self._batchers.get(key)
            if b is None:
                with _batcher_lock:
                    b = self._batchers.get(key)
                    if b is None:
                        b = self._batchers[key] = nat.Batcher(self.ctx, self.net, kind=2, frame_hw=key, callers=self._coalesce,
                                                              wait_us=int(os.environ.get("HAVC_COALESCE_WAIT_US", "300")))
            return b.submit(img)
        return self.colorize_frames(img[None])[0]


def pil_resize_np(ctx, img, size, resample):
    """Pillow Image.resize((w, h), resample) on the GPU, bit-exact (BILINEAR = 2, BICUBIC = 3)."""
    img = np.ascontiguousarray(img, dtype=np.uint8)
    out = np.empty((size[1], size[0], 3), np.uint8)
    nat.check(ctx.lib.havc_pil_resize(ctx.h, nat.as_ptr(img), img.shape[1], img.shape[0], nat.as_ptr(out), size[0], size[1], resample), ctx.h)
    return out
