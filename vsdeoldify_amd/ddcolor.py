"""DDColor on the MI355X behind the call shape of `vsddcolor.ddcolor` as vs-deoldify uses it (vsslib/vsmodels.py:298-363).

PARITY UNPINNED (external wheel, not in the reference tree; oracle/ddcolor.py).  `DDColorRender.colorize_frame` takes / returns
u8 HWC frames; the network runs at input_size = trunc(render_factor / 2) * 32 (vsmodels.py:302) and frames of another size are
squashed / the ab map stretched back inside the library.  The RGBH / RGBS <-> RGB24 casts around the call stay in VapourSynth.  No CPU fallback: everything runs through libhavc_mi355.
"""
import os
import threading

import numpy as np

from . import _native as nat
from .ddcolor_net import DDColorGenerator
from .render import get_context


class DDColorRuntime:
    """Packed DDColor weights on one GPU + a cache of nets keyed by (input size, max_batch)."""

    def __init__(self, ctx, state_dict, depths=(3, 3, 27, 3), dec_layers=9):
        self.ctx = ctx
        self.gen = DDColorGenerator(state_dict, depths, dec_layers)
        self.weights = nat.Weights(ctx, self.gen.blob)
        self.nets = {}

    def net(self, S, max_batch=1):
        key = (S, max_batch)
        if key not in self.nets:
            ops, bufs, i, o, names, consts = self.gen.plan(S)
            n = nat.Net(self.ctx, self.weights, ops, bufs, i, o, S, max_batch)
            n.names, n.plan_ops = names, ops
            for buf, arr, pitch, rows_per_frame in consts:               # constant maps: one copy per frame slot
                a = np.zeros((max_batch, rows_per_frame, pitch), np.float16)
                a[:, :arr.shape[0], :arr.shape[1]] = arr.astype(np.float16)[None]
                n.upload(buf, a)
            if os.environ.get("HAVC_AUTOTUNE", "1") != "0":
                n.autotune(max_batch)
            self.nets[key] = n
        return self.nets[key]

    def colorize(self, frames, input_size=None, max_batch=None):
        """frames: uint8 [N, H, W, 3] -> uint8 [N, H, W, 3]; the network runs at input_size (default: the frame size, which
        must then be square and a multiple of 32)."""
        from .device import is_device, operand_ptr
        dev = is_device(frames)
        if not dev:
            frames = np.ascontiguousarray(frames, dtype=np.uint8)
        assert frames.ndim == 4 and frames.shape[3] == 3
        S = frames.shape[1] if input_size is None else input_size
        assert S % 32 == 0 and (input_size is not None or frames.shape[1] == frames.shape[2])
        n = frames.shape[0]
        net = self.net(S, max_batch or min(n, 8))
        out = frames.empty_like() if dev else np.empty_like(frames)
        nat.check(self.ctx.lib.havc_ddcolor_frames(self.ctx.h, net.h, operand_ptr(frames), operand_ptr(out), n, frames.shape[2],
                                                   frames.shape[1]), self.ctx.h)
        return out

    def colorize_planar_float(self, planes, input_size=None):
        """The call shape of vsddcolor.ddcolor as vs-deoldify uses it (vsslib/vsmodels.py:353-363): ONE frame as float32 (RGBS)
        or float16 (RGBH) planes [3, H, W] in [0, 1] -> planes of the same dtype, not quantised."""
        import ctypes as C
        planes = np.ascontiguousarray(planes)
        if planes.ndim != 3 or planes.shape[0] != 3 or planes.dtype not in (np.float32, np.float16):
            raise ValueError("ddcolor: RGBS / RGBH frame = float32 / float16 array [3, H, W]")
        h, w = planes.shape[1:]
        S = h if input_size is None else input_size
        net = self.net(S, 1)
        out = np.empty_like(planes)
        esz = planes.dtype.itemsize
        pin = (C.c_void_p * 3)(*[planes[p].ctypes.data for p in range(3)])
        pout = (C.c_void_p * 3)(*[out[p].ctypes.data for p in range(3)])
        nat.check(self.ctx.lib.havc_ddcolor_frame_planar_f(self.ctx.h, net.h, pin, w * esz, pout, w * esz, 1 if esz == 2 else 0, w, h), self.ctx.h)
        return out

    def close(self):
        for n in self.nets.values():
            n.close()
        self.nets.clear()
        self.weights.close()


def load_state_dict(path):
    """ddcolor_modelscope.pth / ddcolor_artistic.pth: {'params': state_dict} or a bare state dict (public checkpoint layout)."""
    import torch
    sd = torch.load(path, map_location="cpu", weights_only=False)
    sd = sd.get("params", sd)
    return {k: v.numpy() for k, v in sd.items()}


_batcher_lock = threading.Lock()


class DDColorRender:
    """What `vsddcolor.ddcolor(clip, model, input_size, ...)` does per frame, for frames already at input_size."""

    MODEL_FILES = {0: "ddcolor_modelscope.pth", 1: "ddcolor_artistic.pth"}          # __init__.py:2367-2371

    def __init__(self, model=1, input_size=512, device_index=0, state_dict=None, model_dir=None, depths=(3, 3, 27, 3), dec_layers=9,
                 coalesce=0):
        """coalesce = N > 0: colorize_frame calls made concurrently by N threads (the filter's num_streams / VapourSynth's worker pool)
        are merged into batches of up to N frames (havc_batcher): a DDColor pass is 7.7 ms for one frame and 1.2 ms per frame at 16."""
        if model not in self.MODEL_FILES:
            raise ValueError("ddcolor: model must be 0 (modelscope) or 1 (artistic); 2/3 are siggraph17/eccv16 (ModelColorization)")
        if input_size % 32:
            raise ValueError("ddcolor: input_size must be a multiple of 32")
        if device_index == 99:
            raise ValueError("device_index=99 (CPU) is not supported: this library is MI355X only")
        self.input_size = input_size
        if state_dict is None:
            import os
            if model_dir is None:
                raise ValueError("ddcolor: pass state_dict or model_dir (the vsddcolor models folder)")
            state_dict = load_state_dict(os.path.join(model_dir, self.MODEL_FILES[model]))
        self.rt = DDColorRuntime(get_context(device_index), state_dict, depths, dec_layers)
        self._coalesce, self._batchers = coalesce, {}

    def colorize_frame(self, frame):
        """u8 HWC in -> u8 HWC out, any frame size (the network runs at input_size)."""
        from .device import is_device
        if is_device(frame):
            if frame.ndim != 3:
                raise ValueError("ddcolor: frame must be HWC RGB")
            return self.rt.colorize(frame.reshaped((1,) + frame.shape), self.input_size).reshaped(frame.shape)
        f = np.asarray(frame)
        if f.ndim != 3 or f.shape[2] != 3:
            raise ValueError("ddcolor: frame must be HWC RGB")
        if self._coalesce:
            key = f.shape[:2]
            b = self._batchers.get(key)
            if b is None:
                with _batcher_lock:
                    b = self._batchers.get(key)
                    if b is None:
                        b = self._batchers[key] = nat.Batcher(self.rt.ctx, self.rt.net(self.input_size, self._coalesce), kind=1, frame_hw=key,
                                                              callers=self._coalesce, wait_us=int(os.environ.get("HAVC_COALESCE_WAIT_US", "300")))
            return b.submit(f)
        return self.rt.colorize(f[None], self.input_size)[0]

    def colorize_frames(self, frames, max_batch=None):
        """[N, H, W, 3] u8 (ndarray or DeviceImage) -> same kind"""
        return self.rt.colorize(frames, self.input_size, max_batch)

    def colorize_planar_float(self, planes):
        return self.rt.colorize_planar_float(planes, self.input_size)
