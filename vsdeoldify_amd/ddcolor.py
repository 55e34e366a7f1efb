"""DDColor on the MI355X behind the call shape of `vsddcolor.ddcolor` as vs-deoldify uses it (vsslib/vsmodels.py:298-363).

PARITY UNPINNED (external wheel, not in the reference tree; oracle/ddcolor.py).  `DDColorRender.colorize_frame` takes / returns
u8 HWC frames; the network runs at input_size = trunc(render_factor / 2) * 32 (vsmodels.py:302) and frames of another size are
squashed / the ab map stretched back inside the library.  The RGBH / RGBS <-> RGB24 casts around the call stay in VapourSynth.  No CPU fallback: everything runs through libhavc_mi355.
"""
import os
import threading

import numpy as np

from . import _native as nat
from .precision import DEFAULT_PRECISION, resolve as resolve_precision
from .ddcolor_net import DDColorGenerator
from .render import get_context


class DDColorRuntime:
    """Packed DDColor weights on one GPU + a cache of nets keyed by (input size, max_batch)."""

    def __init__(self, ctx, state_dict, depths=(3, 3, 27, 3), dec_layers=9, share=None, precision="fast"):
        """share: another DDColorRuntime of the same GPU whose packed plan and device weights are reused (read-only): a second context
        (its own HIP stream and activation arena) for frames that run CONCURRENTLY with the first one's (DDColorRender num_streams)."""
        self.ctx = ctx
        if share is not None:
            self.gen, self.weights, self._owns_weights = share.gen, share.weights, False
        else:
            self.gen = DDColorGenerator(state_dict, depths, dec_layers, precision=precision)
            self.weights, self._owns_weights = nat.Weights(ctx, self.gen.blob), True
        self.nets = {}

    def net(self, S, max_batch=1):
        key = (S, max_batch)
        if key not in self.nets:
            ops, bufs, i, o, names, consts = self.gen.plan(S)
            n = nat.Net(self.ctx, self.weights, ops, bufs, i, o, S, max_batch)
            n.names, n.plan_ops = names, ops
            for buf, arr, pitch, rows_per_frame in consts:               # constant maps: one copy per frame slot
                a = np.zeros((max_batch, rows_per_frame, pitch), np.float16)
                if self.gen.precise:                                     # hi / lo pair rows: [hi: pitch / 2 | lo: pitch / 2]
                    hi = arr.astype(np.float16)
                    a[:, :arr.shape[0], :arr.shape[1]] = hi[None]
                    a[:, :arr.shape[0], pitch // 2:pitch // 2 + arr.shape[1]] = ((arr.astype(np.float32) - hi.astype(np.float32)) * np.float32(2048.0)).astype(np.float16)[None]
                else:
                    a[:, :arr.shape[0], :arr.shape[1]] = arr.astype(np.float16)[None]
                n.upload(buf, a)
            if os.environ.get("HAVC_AUTOTUNE", "1") != "0":
                n.autotune(max_batch)
            self.nets[key] = n
        return self.nets[key]

    def colorize(self, frames, input_size=None, max_batch=None, out=None):
        """frames: uint8 [N, H, W, 3] -> uint8 [N, H, W, 3]; the network runs at input_size (default: the frame size, which
        must then be square and a multiple of 32).  out: where to write (same kind and shape as frames)."""
        from .device import is_device, operand_ptr
        dev = is_device(frames)
        if not dev:
            frames = np.ascontiguousarray(frames, dtype=np.uint8)
        assert frames.ndim == 4 and frames.shape[3] == 3
        S = frames.shape[1] if input_size is None else input_size
        assert S % 32 == 0 and (input_size is not None or frames.shape[1] == frames.shape[2])
        n = frames.shape[0]
        net = self.net(S, max_batch or min(n, 8))
        if out is None:
            from .device import DeviceImage
            out = DeviceImage(self.ctx, frames.shape) if dev else np.empty_like(frames)      # this context's pool: its stream orders the reuse
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
        if self._owns_weights:
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
                 coalesce=0, num_streams=None, worker=0, precision=None):
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
            if model_dir is None:
                raise ValueError("ddcolor: pass state_dict or model_dir (the vsddcolor models folder)")
            state_dict = load_state_dict(os.path.join(model_dir, self.MODEL_FILES[model]))
        # worker: which context of the GPU the model lives on (render.get_context): a caller that runs DDColor next to another model (HAVC's
        # DeOldify + DDColor methods) gives it a context of its own, i.e. its own HIP stream
        # precision: "fast" (fp16 activations) / "precise" (fp32-class arithmetic on hi / lo pairs, ddcolor_net.DDColorGenerator); None reads HAVC_PRECISION, then the package default "precise" (precision.py)
        self.precision = resolve_precision(precision)            # explicit > HAVC_PRECISION > "precise" (vsdeoldify_amd/precision.py)
        self.rt = DDColorRuntime(get_context(device_index, worker), state_dict, depths, dec_layers, precision=self.precision)
        self._coalesce, self._batchers = coalesce, {}
        # num_streams (vsddcolor's parameter, vsslib/vsmodels.py:356; the reference's callers leave it at 1): clips of >= 8 frames are cut into
        # that many parts which run concurrently, each on its own context / HIP stream with the same packed weights.  Frames are independent:
        # same bytes.  Default 1: at 32 frames per call one chain already fills the chip better than two chains of 16 (c3: 883 vs 859 frames/s,
        # c4: 655 vs 662; profiles/r3_conv_experiments.txt) -- the knob is for callers that hand over small clips from several places.
        self.num_streams = int(os.environ.get("HAVC_DD_STREAMS", "1")) if num_streams is None else int(num_streams)
        self._device_index, self._workers, self._pool = device_index, [], None

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
        from .device import DeviceImage, is_device
        n, S = frames.shape[0], max(1, self.num_streams)
        if S == 1 or n < 8:
            return self.rt.colorize(frames, self.input_size, max_batch)
        import concurrent.futures
        while len(self._workers) < S - 1:
            k = len(self._workers) + 1
            self._workers.append(DDColorRuntime(get_context(self._device_index, ("ddcolor", k)), None, share=self.rt))
        if self._pool is None:
            self._pool = concurrent.futures.ThreadPoolExecutor(max_workers=S - 1)
        dev = is_device(frames)
        if not dev:
            frames = np.ascontiguousarray(frames, dtype=np.uint8)
        out = DeviceImage(self.rt.ctx, frames.shape) if dev else np.empty_like(frames)
        per = (n + S - 1) // S
        cuts = [(k * per, min(n, (k + 1) * per)) for k in range(S) if k * per < n]
        part_batch = min(per, max_batch) if max_batch else per
        sub = (lambda a, lo, hi: a.frames(lo, hi)) if dev else (lambda a, lo, hi: a[lo:hi])
        if dev:
            self.rt.ctx.synchronize()                                  # the clip was produced on this context's stream; the others do not see it

        def run(rt, lo, hi):
            rt.colorize(sub(frames, lo, hi), self.input_size, part_batch, out=sub(out, lo, hi))
            if dev and rt is not self.rt:
                rt.ctx.synchronize()                                   # device clips are only enqueued: this context's stream must not run ahead of the others' parts
        futs = [self._pool.submit(run, self._workers[k - 1], lo, hi) for k, (lo, hi) in enumerate(cuts) if k]
        run(self.rt, *cuts[0])
        for f in futs:
            f.result()
        return out

    def colorize_planar_float(self, planes):
        return self.rt.colorize_planar_float(planes, self.input_size)
