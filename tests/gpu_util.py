"""Helpers for the -m gpu parity tests: run single ops / small plans through the C ABI."""
import numpy as np

from vsdeoldify_amd import _native as nat
from vsdeoldify_amd.plan import PlanBuilder, View, WeightPack, pack_conv, pad_to


def nhwc_pad(x_nchw, span=None):
    """float NCHW -> fp16 NHWC with channels zero-padded to `span` (multiple of 8)."""
    n, c, h, w = x_nchw.shape
    span = span or pad_to(c, 8)
    out = np.zeros((n, h, w, span), np.float16)
    out[..., :c] = np.transpose(x_nchw, (0, 2, 3, 1)).astype(np.float16)
    return out


def run_plan(ctx, pack, b, uploads, downloads, batch, cfg=0):
    """uploads: {buf: ndarray}; downloads: {buf: (shape, dtype)} -> {buf: ndarray}."""
    ops, bufs = b.finish()
    ops["reserved"] = cfg
    w = nat.Weights(ctx, pack.blob() or b"\0" * 256)
    net = nat.Net(ctx, w, ops, bufs, 0, 0, 0, batch)
    try:
        for k, v in uploads.items():
            net.upload(k, v)
        net.run_ops(0, len(ops), batch)
        return {k: net.download(k, shp, dt) for k, (shp, dt) in downloads.items()}
    finally:
        net.close()
        w.close()


def conv_op(ctx, x, W, bias=None, scale=None, shift=None, stride=1, pad=0, dil=1, flags=0, res=None, pixshuf=False, cfg=0):
    """x [B,Cin,H,W] float, W [Cout,Cin,kh,kw] -> fp16 NHWC output (float32 NCHW returned) via OP_CONV."""
    B, Cin, H, Wd = x.shape
    pack, b = WeightPack(), PlanBuilder()
    xv = b.tensor(H, Wd, Cin)
    pc = pack_conv(pack, W.astype(np.float32), xv.cmap, xv.span, bias=bias, scale=scale, shift=shift, pixshuf=pixshuf)
    kh, kw = W.shape[2:]
    Ho = (H + 2 * pad - dil * (kh - 1) - 1) // stride + 1
    Wo = (Wd + 2 * pad - dil * (kw - 1) - 1) // stride + 1
    if pixshuf:
        flags |= nat.F_OUT_PIXSHUF
        yv = b.tensor(2 * Ho, 2 * Wo, W.shape[0] // 4)
    else:
        yv = b.tensor(Ho, Wo, W.shape[0])
    ups = {xv.buf: nhwc_pad(x, xv.cpitch)}
    rv = None
    if res is not None:
        rv = b.tensor(Ho, Wo, W.shape[0])
        ups[rv.buf] = nhwc_pad(res, rv.cpitch)
        flags |= nat.F_RESIDUAL
    b.conv("t", pc, xv, yv, stride=stride, pad=pad, dil=dil, flags=flags, res=rv)
    out = run_plan(ctx, pack, b, ups, {yv.buf: ((B, yv.H, yv.W, yv.cpitch), np.float16)}, B, cfg)[yv.buf]
    out = out[..., :yv.span]
    return out.astype(np.float32)[..., :yv.C].transpose(0, 3, 1, 2), out


# ---- precise mode (HAVC_F_PRECISE): tensors are hi / lo fp16 pairs, pixel row = [hi: P | lo: P] ----
def hl_split(x):
    """fp32 array -> (hi, lo) fp16 arrays with hi + lo / 2048 == x up to 2^-22 relative (csrc/conv_common.h split_hl)"""
    x = np.asarray(x, np.float32)
    hi = x.astype(np.float16)
    lo = ((x - hi.astype(np.float32)) * np.float32(2048.0)).astype(np.float16)
    return hi, lo


def hl_pack(x_nchw, cpitch):
    """float NCHW -> precise NHWC buffer image [n, h, w, cpitch] (cpitch = 2 P: hi plane, then lo plane)"""
    n, c, h, w = x_nchw.shape
    P = cpitch // 2
    out = np.zeros((n, h, w, cpitch), np.float16)
    hi, lo = hl_split(np.transpose(x_nchw, (0, 2, 3, 1)))
    out[..., :c] = hi
    out[..., P:P + c] = lo
    return out


def hl_unpack(buf, C):
    """precise NHWC buffer image [n, h, w, 2 P] -> float32 NCHW of the first C channels"""
    P = buf.shape[-1] // 2
    v = buf[..., :C].astype(np.float32) + buf[..., P:P + C].astype(np.float32) / np.float32(2048.0)
    return v.transpose(0, 3, 1, 2)


def conv_op_precise(ctx, x, W, bias=None, scale=None, shift=None, stride=1, pad=0, dil=1, flags=0, res=None, pixshuf=False, cfg=0):
    """conv_op in precise mode: fp32 in, fp32 out (both travel as hi / lo pairs), returns (float32 NCHW, raw buffer image)"""
    B, Cin, H, Wd = x.shape
    pack, b = WeightPack(), PlanBuilder(precise=True)
    xv = b.tensor(H, Wd, Cin)
    pc = pack_conv(pack, W.astype(np.float32), xv.cmap, xv.span, bias=bias, scale=scale, shift=shift, pixshuf=pixshuf, precise=True)
    kh, kw = W.shape[2:]
    Ho = (H + 2 * pad - dil * (kh - 1) - 1) // stride + 1
    Wo = (Wd + 2 * pad - dil * (kw - 1) - 1) // stride + 1
    if pixshuf:
        flags |= nat.F_OUT_PIXSHUF
        yv = b.tensor(2 * Ho, 2 * Wo, W.shape[0] // 4)
    else:
        yv = b.tensor(Ho, Wo, W.shape[0])
    ups = {xv.buf: hl_pack(x, xv.cpitch)}
    rv = None
    if res is not None:
        rv = b.tensor(Ho, Wo, W.shape[0])
        ups[rv.buf] = hl_pack(res, rv.cpitch)
        flags |= nat.F_RESIDUAL
    b.conv("t", pc, xv, yv, stride=stride, pad=pad, dil=dil, flags=flags, res=rv)
    raw = run_plan(ctx, pack, b, ups, {yv.buf: ((B, yv.H, yv.W, yv.cpitch), np.float16)}, B, cfg)[yv.buf]
    return hl_unpack(raw, yv.C), raw


# ---- shared CPU-oracle frames (round 6: the suite evaluated the same 1080p oracle frame in several files, ~5 s each) ----
_ORACLE_FULLSIZE = {}


def oracle_fullsize_stable(seed_video, seed_stable, frame_index, render_factor=35):
    """oracle.pipeline.colorize_frame_fullsize('stable') of frame `frame_index` of the synthetic 1080p clip on the seeded weight pair; cached per process"""
    key = (seed_video, seed_stable, frame_index, render_factor)
    if key not in _ORACLE_FULLSIZE:
        from oracle import pipeline
        from vsdeoldify_amd.clip import synthetic_gray_frame
        from vsdeoldify_amd.synth import synth_state_dict
        sds = {"video": synth_state_dict("wide", seed_video), "stable": synth_state_dict("wide", seed_stable)}
        _ORACLE_FULLSIZE[key] = pipeline.colorize_frame_fullsize(sds, "stable", synthetic_gray_frame(frame_index, 1920, 1080), render_factor, 0.5)
    return _ORACLE_FULLSIZE[key]
