"""Device-resident clip colorization: the batch / multi-GPU face of the hot path.

`ClipColorizer` keeps both generators of a ModelImageRender on one GPU and colours frames that are
already in HBM through `havc_colorize_clip` (resize -> 2 U-Net passes -> blend -> YUV merge -> resize +
luma re-attach).  `shard_frames` is the whole multi-GPU story: frames are independent, so
frame n goes to rank n mod G and there is no data-path collective.
"""
import numpy as np

from . import _native as nat
from .render import RENDER_BASE, ModelImageRender


def shard_frames(n_frames, rank, world_size):
    """indices of the frames rank `rank` colours (round-robin, like VapourSynth's frame-parallel pulls)."""
    return list(range(rank, n_frames, world_size))


def synthetic_gray_frame(idx, width=1920, height=1080, base_seed=20240229):
    """SURVEY.md §8d synthetic clip: R=G=B luma = clip(128 + 48*lowpass(N(0,1), sigma 24) + 32*ramp + 6*N(0,1))."""
    from scipy.ndimage import gaussian_filter, zoom
    r = np.random.default_rng(base_seed + idx)
    # sigma-24 low-pass noise, synthesised at quarter resolution (sigma 6) and bilinearly enlarged: 16x cheaper
    qh, qw = (height + 3) // 4, (width + 3) // 4
    low = gaussian_filter(r.standard_normal((qh, qw), dtype=np.float32), 6.0)
    low = zoom(low, 4, order=1)[:height, :width]
    low /= max(float(low.std()), 1e-6)
    ramp = np.linspace(-1.0, 1.0, width, dtype=np.float32)[None, :]
    luma = np.clip(128 + 48 * low + 32 * ramp + 6 * r.standard_normal((height, width), dtype=np.float32), 0, 255)
    luma = luma.astype(np.uint8)
    return np.stack([luma, luma, luma], -1)


class ClipColorizer:
    def __init__(self, modelname="stable", render_factor=35, video_weight=0.5, device_index=0, state_dicts=None,
                 package_dir=None, max_batch=8, precision=None):
        self.render = ModelImageRender(package_dir, modelname, render_factor, video_weight, device_index, state_dicts,
                                       max_batch, precision=precision)
        self.ctx = self.render.ctx
        self.S = render_factor * RENDER_BASE
        self.max_batch = max_batch
        self.video = self.render._video.net(self.S, max_batch)
        self.second = self.render._second.net(self.S, max_batch) if self.render._second else None
        self.video_weight = float(video_weight)

    def colorize_device(self, d_src, d_dst, n_frames, width, height):
        nat.check(self.ctx.lib.havc_colorize_clip(self.ctx.h, self.video.h, self.second.h if self.second else None,
                                                  self.video_weight, d_src, d_dst, n_frames, width, height), self.ctx.h)

    def colorize_host(self, frames, out=None):
        """uint8 [n,h,w,3] host array -> coloured host array through havc_colorize_clip_host: uploads, U-Net passes and downloads of
        consecutive batches overlap on three streams.  Pass arrays living in pinned memory (ctx.host_alloc) for true async copies."""
        frames = np.ascontiguousarray(frames, dtype=np.uint8)
        n, h, w, _ = frames.shape
        out = np.empty_like(frames) if out is None else out
        assert out.shape == frames.shape and out.dtype == np.uint8 and out.flags.c_contiguous
        nat.check(self.ctx.lib.havc_colorize_clip_host(self.ctx.h, self.video.h, self.second.h if self.second else None, self.video_weight,
                                                       nat.as_ptr(frames), nat.as_ptr(out), n, w, h), self.ctx.h)
        return out

    def colorize(self, frames):
        """uint8 [n,h,w,3] host array -> coloured uint8 [n,h,w,3] (H2D, device pipeline, D2H)."""
        frames = np.ascontiguousarray(frames, dtype=np.uint8)
        n, h, w, _ = frames.shape
        d_src, d_dst = self.ctx.dev_alloc(frames.nbytes), self.ctx.dev_alloc(frames.nbytes)
        try:
            self.ctx.dev_upload(d_src, frames)
            self.colorize_device(d_src, d_dst, n, w, h)
            out = np.empty_like(frames)
            self.ctx.dev_download(out, d_dst)
            return out
        finally:
            self.ctx.dev_free(d_src)
            self.ctx.dev_free(d_dst)
