"""-m gpu: whole DeOldify generators and the ModelImageRender drop-in against the CPU oracle.

Floating-point path (fp16 storage, fp32 MFMA accumulation vs the fp32 oracle).  Stated tolerance
(BASELINE.json north_star: CIEDE2000 < 1.0 vs the reference path), justified by profiles/r2_precision_study.txt:
  * raw colour (u8 out of the network, before the YUV merge): >= 99.5 % of bytes within +-1 LSB, >= 99.9 %
    within +-2 LSB, mean CIEDE2000 < 0.25, p99 < 1.0 (measured 0.90-0.96).  The reference TRUNCATES x*255
    (filters.py:65-68), so +-1 LSB flips are inherent to any arithmetic that is not bit-identical to fp32 torch:
    two fp32 CPU evaluations that differ only in summation order already flip 0.01-0.03 % of the pixels by up to
    dE00 2.5 (the noise floor; its p99 is 0);
  * final image of ONE model at the net size (chroma of the colour on the luma of the source): mean CIEDE2000
    < 0.25 (measured 0.05-0.17), p99 < 2.1 (measured 1.6-1.93): a single-LSB U/V flip on a DARK source pixel is
    worth 1-3 dE00 units, so the tail of the distribution is set by the truncation.  The CPU simulation of the HIP
    rounding points (tests/precision_study.py) reproduces these figures and shows that they are intrinsic to fp16
    MFMA inputs: with the whole encoder, the network input and the last layer in fp32 the p99 moves 1.91 -> 1.75;
  * the 1080p output of the HAVC flow (two-model blend + Spline64 up-pass + luma re-attach) is covered by
    tests/test_gpu_fullsize.py: mean 0.11-0.12, p99 1.2, 97 % of the pixels below 1.0.
"""
RAW_TOL = dict(within1=0.995, within2=0.999, mean=0.25, p99=1.0)
FINAL_TOL = dict(mean=0.25, p99=2.1)


def check_final(got, ref):
    de = imaging.delta_e00_images(got, ref)
    assert got.shape == ref.shape and de.mean() < FINAL_TOL["mean"] and np.percentile(de, 99) < FINAL_TOL["p99"], \
        (de.mean(), np.percentile(de, 99), summarize(got, ref))

import numpy as np
import pytest

from oracle import imaging, pipeline
from vsdeoldify_amd import _native as nat
from vsdeoldify_amd.render import GeneratorRuntime, ModelImageRender
from vsdeoldify_amd.synth import synth_state_dict

pytestmark = pytest.mark.gpu


def make_frame(S, seed):
    """structured gray frame (ramp + low-pass noise + grain), R=G=B, like the bench clip."""
    r = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:S, 0:S]
    low = r.standard_normal((S // 8 + 2, S // 8 + 2))
    low = np.kron(low, np.ones((8, 8)))[:S, :S]
    luma = np.clip(128 + 48 * low + 32 * (xx / S - 0.5) * 2 + 6 * r.standard_normal((S, S)), 0, 255).astype(np.uint8)
    return np.stack([luma] * 3, -1)


@pytest.fixture(scope="module")
def sds():
    return {"video": synth_state_dict("wide", 1), "stable": synth_state_dict("wide", 2),
            "artistic": synth_state_dict("deep", 3)}


def raw_gpu(ctx, rt, frames):
    S = frames.shape[1]
    net = rt.net(S, frames.shape[0])
    out = np.empty_like(frames)
    nat.check(ctx.lib.havc_deoldify_frames(ctx.h, net.h, None, 0.0, 0, nat.as_ptr(frames), nat.as_ptr(out), len(frames)), ctx.h)
    return out


def summarize(got, ref):
    d = np.abs(got.astype(np.int32) - ref.astype(np.int32))
    return dict(max=int(d.max()), within1=float((d <= 1).mean()), within2=float((d <= 2).mean()))


@pytest.mark.parametrize("arch,which", [("wide", "video"), ("deep", "artistic")])
@pytest.mark.parametrize("S", [64, 80, 96])     # 80: S/16 odd -> nearest-resize branch (unet.py:200-203)
def test_generator_raw_color(ctx, sds, arch, which, S):
    rt = GeneratorRuntime(ctx, sds[which], arch)
    try:
        frames = np.stack([make_frame(S, 10 + S), make_frame(S, 11 + S)])
        got = raw_gpu(ctx, rt, frames)
        ref = np.stack([pipeline.raw_color_square(sds[which], arch, f) for f in frames])
        s = summarize(got, ref)
        de = imaging.delta_e00_images(got, ref)
        assert (s["within1"] >= RAW_TOL["within1"] and s["within2"] >= RAW_TOL["within2"] and de.mean() < RAW_TOL["mean"]
                and np.percentile(de, 99) < RAW_TOL["p99"]), (s, de.mean(), np.percentile(de, 99))
    finally:
        rt.close()


@pytest.mark.parametrize("modelname", ["video", "stable", "artistic"])
def test_model_image_render_square(ctx, sds, modelname):
    """ModelImageRender.get_transformed_image on a frame already at the render size (HAVC_colorizer flow)."""
    from PIL import Image
    rf = 5
    S = rf * 16
    r = ModelImageRender(None, modelname, rf, 0.5, state_dicts=sds)
    img = make_frame(S, 42)
    got = np.asarray(r.get_transformed_image(Image.fromarray(img)))
    ref = pipeline.model_image_render(sds, modelname, img, rf, 0.5)
    check_final(got, ref)


def test_model_image_render_nonsquare(ctx, sds):
    """non-render-size input: Pillow BILINEAR squash / unsquash on the host like the reference (filters.py:37-41,70-73)."""
    from PIL import Image
    rf = 4
    r = ModelImageRender(None, "stable", rf, 0.5, state_dicts=sds)
    img = np.asarray(Image.fromarray(make_frame(96, 7)).resize((96, 54)))
    got = np.asarray(r.get_transformed_image(Image.fromarray(img)))
    ref = pipeline.model_image_render(sds, "stable", img, rf, 0.5)
    check_final(got, ref)


def test_batch_matches_single(ctx, sds):
    """frames are independent: a batch of 3 must equal three single-frame calls bit for bit."""
    rt = GeneratorRuntime(ctx, sds["video"], "wide")
    try:
        frames = np.stack([make_frame(64, s) for s in (1, 2, 3)])
        assert np.array_equal(raw_gpu(ctx, rt, frames), np.concatenate([raw_gpu(ctx, rt, f[None]) for f in frames]))
    finally:
        rt.close()


def test_fused_final_conv_matches_unfused(ctx, sds):
    """HAVC_F_FUSE_RGB8 (layers.11 + SigmoidRange + u8 inside the res-block conv epilogue) keeps the unfused path's
    rounding points (fp16 r2, fp16 weights, fp32 accumulate); only the summation order of the 259-term dot differs."""
    frames = np.stack([make_frame(96, 5), make_frame(96, 6)])
    outs = []
    for fuse in (True, False):
        rt = GeneratorRuntime(ctx, sds["video"], "wide", fuse_final=fuse)
        try:
            names = rt.net(96, 2).names
            assert ("layers.11.0" in names) != fuse and ("layers.10.layers.1.0+11" in names) == fuse
            outs.append(raw_gpu(ctx, rt, frames))
        finally:
            rt.close()
    d = np.abs(outs[0].astype(np.int32) - outs[1].astype(np.int32))
    assert d.max() <= 1 and (d == 0).mean() > 0.999, (int(d.max()), float((d == 0).mean()))


@pytest.mark.parametrize("S", [272, 320])
def test_fused_shuffle_blur_is_bit_identical(ctx, sds, S):
    """HAVC_F_PS_BLUR (1x1 conv + PixelShuffle + blur in one kernel, 16x16 pixel tiles with a one-pixel halo) must give
    exactly the bytes of the three-pass path: same fp16 rounding of the shuffled tensor, same summation order.
    272: 136 = 9 * 15 + 1 low-res rows (a last tile of one useful row + clamping); 320: 160 rows."""
    frames = np.stack([make_frame(S, 40), make_frame(S, 41)])
    outs = []
    for fuse in (True, False):
        rt = GeneratorRuntime(ctx, sds["video"], "wide", fuse_blur=fuse)
        try:
            names = rt.net(S, 2).names
            assert ("layers.8+blur" in names) == fuse and ("layers.8.blur" in names) != fuse
            outs.append(raw_gpu(ctx, rt, frames))
        finally:
            rt.close()
    assert np.array_equal(outs[0], outs[1])


def test_fused_shuffle_blur_with_padded_channel_counts_deep(ctx):
    """DynamicUnetDeep (artistic): 300 / 336 channels per sub-pixel are padded to 320 / 384 zero weight rows so that the fused
    shuffle + blur epilogue applies; only the real channels are stored (the image channels behind them in the tail tensor survive).
    Exactly the bytes of the three-pass path."""
    from vsdeoldify_amd.synth import synth_state_dict
    sd = synth_state_dict("deep", 3)
    S = 272
    frames = np.stack([make_frame(S, 50), make_frame(S, 51)])
    outs = []
    for fuse in (True, False):
        rt = GeneratorRuntime(ctx, sd, "deep", fuse_blur=fuse)
        try:
            names = rt.net(S, 2).names
            assert ("layers.8+blur" in names) == fuse and ("layers.8.blur" in names) != fuse
            outs.append(raw_gpu(ctx, rt, frames))
        finally:
            rt.close()
    assert np.array_equal(outs[0], outs[1])


@pytest.mark.parametrize("seeds,rf", [((1, 2), 6), ((11, 12), 10), ((21, 22), 10), ((1, 2), 35)])
def test_low_latency_split_k_plan_matches_the_batched_plan(ctx, seeds, rf, monkeypatch):
    """ModelImageRender(low_latency=True): nets for one frame per call with split-K convs (both generators, i.e. both streams of the context).  Same
    arithmetic up to the fp32 summation order of the K parts (which moves some fp16 roundings of the stored activations).  Round 5 measured the
    distance from the batch-independent nets on three seeded weight sets: within 2 LSB everywhere and > 90 % of the bytes equal at render factors 6 / 10
    (93.3 / 91.5 / 92.2 %), but 3 LSB on isolated bytes and 89.5 % equal at the headline size (rf 35) -- which is why the mode stays opt-in (VERDICT r4
    item 6 asked for <= 2 LSB before making it the default).  Both plans meet the tolerance against the oracle."""
    from PIL import Image
    sds = {"video": synth_state_dict("wide", seeds[0]), "stable": synth_state_dict("wide", seeds[1])}
    S = rf * 16
    img = make_frame(S, 9)
    monkeypatch.delenv("HAVC_LOW_LATENCY", raising=False)
    base = ModelImageRender(None, "stable", rf, 0.5, state_dicts=sds)                # the library default: batch-independent nets
    fast = ModelImageRender(None, "stable", rf, 0.5, state_dicts=sds, low_latency=True)
    assert fast._low_latency and not base._low_latency
    try:
        a = np.asarray(base.get_transformed_image(Image.fromarray(img)))
        b = np.asarray(fast.get_transformed_image(Image.fromarray(img)))
        net = fast._video.net(S, 1, True)
        nsplit = sum(1 for o in net.ops if o["type"] == nat.OP_CONV and (int(o["flags"]) >> 16) & 15)
        assert nsplit > 20 and net is not base._video.net(S, 1)
        d = np.abs(a.astype(int) - b.astype(int))
        print(f"low-latency vs batched plan, seeds {seeds} rf {rf}: max |d| {int(d.max())} LSB, bytes equal {float((d == 0).mean()):.4f}")
        lim = (2, 0.90) if rf <= 10 else (3, 0.88)
        assert d.max() <= lim[0] and (d == 0).mean() > lim[1], (int(d.max()), float((d == 0).mean()))
        if rf <= 10:
            check_final(b, pipeline.model_image_render(sds, "stable", img, rf, 0.5))
    finally:
        for r in (base, fast):
            for rt in (r._video, r._second):
                rt.close()
