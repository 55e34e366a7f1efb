"""The public HAVC entry points without VapourSynth: HAVC_colorizer / HAVC_merge / HAVC_ddeoldify / ddeoldify.
CPU: the host rules.  GPU: whole frames and clips against the same graph assembled from the oracle pieces -- the INTEGER stages
(resize, merge methods, luma re-attach) must be bit-exact given the models' outputs; the models themselves are covered by their
own tolerance tests."""
import numpy as np
import pytest

from vsdeoldify_amd import havc

SMALL_DD = dict(depths=(1, 1, 2, 1), dec_layers=3)


def test_parameter_rules_follow_the_reference():
    """__init__.py:2452-2462 (method <-> weight), :2482-2483 (render factor range), :2490-2502 (frame size)"""
    H = havc.HAVCFrameColorizer.__new__(havc.HAVCFrameColorizer)                   # rules only: no GPU context
    for method, mweight, want_m, want_w in ((0, 0.4, 0, 0.0), (1, 0.4, 1, 1.0), (2, 0.0, 0, 0.0), (3, 1.0, 1, 1.0), (5, 0.3, 5, 0.3)):
        mw = 0.0 if method == 0 else (1.0 if method == 1 else mweight)
        m = 0 if mw == 0.0 else (1 if mw == 1.0 else method)
        assert (m, mw) == (want_m, want_w)
    H.ddcolor_rf, H.deoldify_rf = 24, 24
    assert H.frame_size(1920) == (24, 384)
    H.ddcolor_rf, H.deoldify_rf = 0, 24
    assert H.frame_size(1920) == (32, 512) and H.frame_size(720) == (18, 384) and H.frame_size(300) == (16, 300)
    H.ddcolor_rf, H.deoldify_rf = 10, 35
    assert H.frame_size(1920) == (10, 560) and H.frame_size(480) == (10, 480)


def test_vapoursynth_only_features_are_refused_not_approximated():
    f = np.zeros((8, 8, 3), np.uint8)
    with pytest.raises(NotImplementedError):
        havc.HAVC_colorizer(f, ddtweak=[False, True, False])                       # rgb_denoise
    with pytest.raises(NotImplementedError):
        havc.HAVC_colorizer(f, ddtweak=[True, False, True])                        # vs_auto_levels (retinex)
    with pytest.raises(NotImplementedError):
        havc.HAVC_colorizer(f, ddtweak=[True, False, False], ddtweak_p=([10.0, 1.0, 2.5, True, 0.3, 0.6, 1.5, 0.5], "none"))   # bright through vs_tweak
    with pytest.raises(NotImplementedError):
        havc.HAVC_colorizer(f, ddtweak=[True, False, False], ddtweak_p=([0.0, 1.0, 2.5, False, 0.3, 0.6, 1.5, 0.5], "none"))   # gamma through vs_tweak
    with pytest.raises(NotImplementedError):
        havc.HAVC_colorizer(f, sc_threshold=0.1)
    with pytest.raises(havc.HAVCError):
        havc.HAVC_colorizer("not a clip")
    with pytest.raises(havc.HAVCError):
        havc.HAVC_merge(f, "nope")


def test_oracle_combine_matches_reference_fixtures():
    """oracle.pipeline.combine_models (the graph the GPU tests compare against) reproduces the merge-method vectors produced by
    executing the reference's own functions (tests/golden/imfilters*.npz, tweaks.npz)."""
    import os
    from oracle import imaging, pipeline
    from tests.conftest import GOLDEN
    g = np.load(os.path.join(GOLDEN, "imfilters2.npz"))
    a, b = g["a"], g["b"]
    assert np.array_equal(pipeline.luma_masked_merge(a, b, 0.3, 0.9, 1.0), g["w_image_luma_merge_0.3_0.9"])
    assert np.array_equal(pipeline.luma_masked_merge(a, b, 0.55, 0.55, 1.0), g["image_luma_merge_0.55"])
    assert np.array_equal(pipeline.luma_masked_merge(a, b, 0.4, 0.7, 0.5), imaging.pil_blend(a, g["w_image_luma_merge_0.4_0.7"], 0.5))
    t = np.load(os.path.join(GOLDEN, "tweaks.npz"))
    for i in range(4):                # ConstrainedChromaMerge merge_frame at four brightness levels (level 0.2, weight 0.5): method 3's first stage
        a1, a2 = t[f"ccm_in1_{i}"], t[f"ccm_in2_{i}"]
        want = imaging.pil_blend(t[f"ccm_out_{i}"], imaging.pil_blend(a1, a2, 0.5), 0.3)
        assert np.array_equal(pipeline.combine_models(a1, a2, 3, 0.5, cmc_p=[0.2]), want), i
    assert pipeline.combine_models(a, None, 2, 0.4) is a and pipeline.combine_models(None, b, 2, 0.4) is b


@pytest.mark.gpu
def test_errors_match_reference_behaviour(ctx):
    with pytest.raises(havc.HAVCError):
        havc.HAVCFrameColorizer(method=2, ddcolor_p=(1, 70, 1.0, 0.0, True))
    with pytest.raises(havc.HAVCError):
        havc.HAVCFrameColorizer(method=9)
    with pytest.raises(havc.HAVCError):
        havc.HAVCFrameColorizer(method=2, device_index=42)
    with pytest.raises(NotImplementedError):
        havc.HAVCFrameColorizer(method=2, deoldify_p=(0, 24, 0.8, 0.0))


def _frame(seed, h=120, w=200):
    r = np.random.default_rng(seed)
    return np.clip(r.normal(120, 55, (h, w, 1)), 0, 255).astype(np.uint8).repeat(3, -1)


HUE_ADJ = "300:360|0.8,0.1"                     # HAVC_colorizer's default ddtweak_p[1] (__init__.py:2293)


def _weights():
    from vsdeoldify_amd.synth import synth_ddcolor_state_dict, synth_state_dict
    return {"video": synth_state_dict("wide", 1)}, synth_ddcolor_state_dict(1, **SMALL_DD)


@pytest.mark.gpu
@pytest.mark.parametrize("method", [0, 1, 2, 3, 4, 5, 6, 7])
def test_gpu_frame_matches_oracle_graph(ctx, method):
    """(1) end to end vs the all-oracle graph within the models' tolerance; (2) the integer stages bit-exact: the oracle graph fed
    with the GPU's OWN model outputs must reproduce the GPU frame byte for byte."""
    from oracle import ddcolor as D, imaging, pipeline, resample
    sds, dsd = _weights()
    rf, w_merge = 10, 0.4
    frame = _frame(3)
    col = havc.HAVCFrameColorizer(method=method, mweight=w_merge, deoldify_p=(0, rf, 1.0, 0.0), ddcolor_p=(1, rf, 1.0, 0.0, True),
                                  state_dicts=sds, ddcolor_state_dict=dsd, ddcolor_kwargs=SMALL_DD, ddtweak_p=(havc.DEF_TWEAK_p, HUE_ADJ))
    got = col.colorize(frame)
    assert got.shape == frame.shape and got.dtype == np.uint8
    fs = min(rf * 16, frame.shape[1])
    sq = resample.resize_rgb8(frame, fs, fs)

    def graph(a, b):
        c = pipeline.combine_models(a, b, method, w_merge)
        return pipeline.post_process(resample.resize_rgb8(c, frame.shape[1], frame.shape[0]), frame)
    # (2) integer stages, given the GPU's model outputs
    from oracle import tweaks
    a_gpu = col._deoldify_render().render_square_batch(sq[None])[0] if method != 1 else None
    b_gpu = tweaks.adjust_hue_range(col._ddcolor_clip(sq[None], (rf // 2) * 32)[0], HUE_ADJ) if method != 0 else None
    d = np.abs(got.astype(int) - graph(a_gpu, b_gpu).astype(int))
    assert d.max() <= 1 and (d > 0).mean() < 2e-4, (method, int(d.max()), float((d > 0).mean()))      # Spline64 .5-boundary ties only
    # (1) all-oracle graph
    a = pipeline.model_image_render(sds, "video", sq, rf, 0, True) if method != 1 else None
    b = tweaks.adjust_hue_range(D.colorize_frame(dsd, sq, input_size=(rf // 2) * 32, **SMALL_DD), HUE_ADJ) if method != 0 else None
    de = imaging.delta_e00_images(got, graph(a, b))
    assert de.mean() < 0.6, (method, float(de.mean()))


@pytest.mark.gpu
def test_gpu_ddcolor_pre_tweak_matches_the_reference_flow(ctx):
    """ddtweak = [True, False, False] with DEF_TWEAK_p and no scene detection (vsslib/vsmodels.py:333-344,373-374): luma_adjusted_levels on
    every frame in front of DDColor, adjust_hue_range behind it, then the clip's luma back (vs_recover_clip_luma).  Integer stages
    bit-exact given the GPU's own DDColor output of the tweaked frame."""
    from oracle import pipeline, resample, tweaks
    sds, dsd = _weights()
    frame = (_frame(11).astype(np.float32) * 0.35).astype(np.uint8)                # dark: mean luma below luma_min = 0.3 -> the levels change
    rf = 10
    col = havc.HAVCFrameColorizer(method=1, deoldify_p=(0, rf, 1.0, 0.0), ddcolor_p=(1, rf, 1.0, 0.0, True), ddcolor_state_dict=dsd, ddcolor_kwargs=SMALL_DD,
                                  ddtweak=(True, False, False), ddtweak_p=(havc.DEF_TWEAK_p, HUE_ADJ))
    got = col.colorize(frame)
    from vsdeoldify_amd import imfilters as F
    from vsdeoldify_amd.device import DeviceImage
    fs = min(rf * 16, frame.shape[1])
    # the GPU's own squashed frame: a Spline64 .5-tie that rounds the other way moves the frame's mean luma, hence the level shift of EVERY pixel
    sq = col._spline64(DeviceImage.from_numpy(ctx, frame[None]), fs, fs).numpy()[0]
    assert np.abs(sq.astype(int) - resample.resize_rgb8(frame, fs, fs).astype(int)).max() <= 1
    pre = tweaks.luma_adjusted_levels(sq, 0.3, 2.5, 0.6, 1.5, 0.5)
    assert np.abs(pre.astype(int) - sq.astype(int)).max() > 0
    assert np.array_equal(F.luma_adjusted_levels_np(ctx, sq, 0.3, 2.5, 0.6, 1.5, 0.5), pre)          # the pre-tweak itself: bit-exact
    b = col._ddcolor_clip(pre[None], (rf // 2) * 32)[0]
    b = pipeline.post_process(tweaks.adjust_hue_range(b, HUE_ADJ), sq)
    want = pipeline.post_process(resample.resize_rgb8(b, frame.shape[1], frame.shape[0]), frame)
    d = np.abs(got.astype(int) - want.astype(int))
    assert d.max() <= 1 and (d > 0).mean() < 2e-4, (int(d.max()), float((d > 0).mean()))


@pytest.mark.gpu
@pytest.mark.parametrize("method", [2, 3, 5, 7])
def test_clip_and_device_paths_equal_the_frame_path(ctx, method):
    """a 3-frame clip == three single-frame calls; DeviceImage in / out (nothing leaves HBM) == ndarray in / out"""
    from vsdeoldify_amd.device import DeviceImage
    sds, dsd = _weights()
    clip = np.stack([_frame(s, 96, 160) for s in (1, 2, 3)])
    col = havc.HAVCFrameColorizer(method=method, mweight=0.5, deoldify_p=(0, 6, 1.0, 0.0), ddcolor_p=(1, 10, 1.0, 0.0, True),
                                  state_dicts=sds, ddcolor_state_dict=dsd, ddcolor_kwargs=SMALL_DD, max_batch=2)
    whole = col.colorize_clip(clip)
    assert whole.shape == clip.shape
    for i in range(3):
        assert np.array_equal(col.colorize(clip[i]), whole[i]), i
    dev = col.colorize_clip(DeviceImage.from_numpy(ctx, clip))
    assert isinstance(dev, DeviceImage) and np.array_equal(dev.numpy(), whole)


@pytest.mark.gpu
@pytest.mark.parametrize("method", [0, 1, 2, 3, 4, 5, 6, 7])
@pytest.mark.parametrize("with_luma", [False, True])
def test_havc_merge_matches_reference_flow(ctx, method, with_luma):
    """HAVC_merge (__init__.py:2536-2675) on two coloured frames [+ a hi-res luma source]: integer arithmetic end to end ->
    bit-exact vs the oracle graph (Spline64 .5 ties excepted)."""
    import math
    from oracle import pipeline, resample
    r = np.random.default_rng(40 + method)
    base = r.integers(0, 256, (90, 120, 1), dtype=np.uint8).repeat(3, -1).astype(np.int32)
    a = np.clip(base + r.integers(-30, 30, base.shape), 0, 255).astype(np.uint8)
    b = np.clip(base + r.integers(-60, 60, base.shape), 0, 255).astype(np.uint8)
    luma = None
    if with_luma:
        luma = resample.resize_rgb8(np.clip(base, 0, 255).astype(np.uint8), 400, 300)
    w = 0.6
    got = havc.HAVC_merge(a, b, clip_luma=luma, weight=w, method=method)

    def up(x):
        return pipeline.post_process(resample.resize_rgb8(x, luma.shape[1], luma.shape[0]), luma)
    if method in (0, 1):
        want = (a if method == 0 else b) if luma is None else up(a if method == 0 else b)
    elif method == 2:
        want = pipeline.combine_models(a, b, 2, w)                     # method 2 returns before the luma handling (__init__.py:2659-2661)
    else:
        aa, bb = a, b
        if luma is not None:
            fs = min(min(max(math.trunc(0.4 * luma.shape[1] / 16), 16), 32) * 16, luma.shape[1])
            aa, bb = resample.resize_rgb8(a, fs, fs), resample.resize_rgb8(b, fs, fs)
        want = pipeline.combine_models(aa, bb, method, w)
        if luma is not None:
            want = up(want)
    d = np.abs(np.asarray(got).astype(int) - want.astype(int))
    assert got.shape == want.shape and d.max() <= 2 and (d > 0).mean() < 1e-3, (method, with_luma, int(d.max()), float((d > 0).mean()))
    if luma is None and method >= 2:
        assert d.max() == 0                                            # no resampler involved: every byte


@pytest.mark.gpu
def test_legacy_entry_points_forward_like_the_reference(ctx):
    """HAVC_ddeoldify / ddeoldify (__init__.py:3612-3653) = HAVC_colorizer with cmc_p = [cmc_tresh], DEF_CRT_p, ddtweak list"""
    sds, dsd = _weights()
    frame = _frame(9, 80, 120)
    kw = dict(method=3, mweight=0.5, deoldify_p=(0, 5, 1.0, 0.0), ddcolor_p=(1, 10, 1.0, 0.0, True), state_dicts=sds, ddcolor_state_dict=dsd,
              ddcolor_kwargs=SMALL_DD)
    want = havc.HAVC_colorizer(frame, cmc_p=[0.2], lmm_p=(0.2, 0.8, 1.0), alm_p=(0.8, 1.0, 0.15), **kw)
    with pytest.warns(DeprecationWarning):
        assert np.array_equal(havc.HAVC_ddeoldify(frame, cmc_tresh=0.2, **kw), want)
    with pytest.warns(DeprecationWarning):
        assert np.array_equal(havc.ddeoldify(frame, cmc_tresh=0.2, **kw), want)
    with pytest.raises(NotImplementedError):
        havc.HAVC_ddeoldify(frame, ddtweak=True, ddtweak_p=([5.0, 1.0, 2.5, True, 0.3, 0.6, 1.5, 0.5], "none"), **kw)


@pytest.mark.gpu
@pytest.mark.parametrize("method", [2, 5])
def test_two_models_side_by_side_give_the_bytes_of_one_after_the_other(ctx, method):
    """Methods that run DeOldify AND DDColor put DDColor on a context (HIP stream) of its own and run the two from two threads
    (HAVCFrameColorizer.overlap_models).  Outputs live in the pool of the context that produced them and cross contexts only behind a
    synchronize(): repeated clips, device-resident and host, give exactly the bytes of the sequential path, call after call."""
    from vsdeoldify_amd.device import DeviceImage
    sds, dsd = _weights()
    clip = np.stack([_frame(20 + i, 96, 160) for i in range(5)])
    kw = dict(method=method, mweight=0.4, deoldify_p=(0, 6, 1.0, 0.0), ddcolor_p=(1, 12, 1.0, 0.0, True), state_dicts=sds, ddcolor_state_dict=dsd,
              ddcolor_kwargs=SMALL_DD, ddtweak=[True, False, False])
    seq = havc.HAVCFrameColorizer(**kw)
    seq.overlap_models = False
    want = seq.colorize_clip(clip)
    par = havc.HAVCFrameColorizer(**kw)
    assert par.overlap_models and par._side_by_side()
    dclip = DeviceImage.from_numpy(par.ctx, clip)
    for rep in range(6):
        got = par.colorize_clip(dclip if rep % 2 else clip)
        got = got.numpy() if rep % 2 else got
        assert np.array_equal(got, want), (method, rep, int(np.abs(got.astype(int) - want.astype(int)).max()))
    assert par._ddcolor.rt.ctx is not par.ctx
