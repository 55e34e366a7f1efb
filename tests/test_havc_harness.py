"""HAVCFrameColorizer: HAVC_colorizer's parameter normalisation, frame-size rule, routing and combine dispatch
without VapourSynth.  CPU: the host rules.  GPU: whole frames against the same graph assembled from the oracle pieces."""
import numpy as np
import pytest

from vsdeoldify_amd import havc


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


@pytest.mark.gpu
@pytest.mark.parametrize("method", [0, 1, 2, 3, 5, 7])
def test_gpu_frame_matches_oracle_graph(ctx, method):
    from oracle import ddcolor as D, imaging, pipeline, resample, tweaks
    from vsdeoldify_amd.synth import synth_ddcolor_state_dict, synth_state_dict
    small = dict(depths=(1, 1, 2, 1), dec_layers=3)
    sds = {"video": synth_state_dict("wide", 1)}
    dsd = synth_ddcolor_state_dict(1, **small)
    rf, w_merge = 10, 0.4
    r = np.random.default_rng(3)
    frame = np.clip(r.normal(120, 55, (120, 200, 1)), 0, 255).astype(np.uint8).repeat(3, -1)
    col = havc.HAVCFrameColorizer(method=method, mweight=w_merge, deoldify_p=(0, rf, 1.0, 0.0), ddcolor_p=(1, rf, 1.0, 0.0, True),
                                  state_dicts=sds, ddcolor_state_dict=dsd)
    import vsdeoldify_amd.ddcolor as ddmod
    orig = ddmod.DDColorRender.__init__

    def small_init(self, *a, **k):                                       # the reduced DDColor depth of this test
        k.update(small)
        orig(self, *a, **k)
    ddmod.DDColorRender.__init__ = small_init
    try:
        got = col.colorize(frame)
    finally:
        ddmod.DDColorRender.__init__ = orig
    # the same graph from the oracle pieces
    fs = min(rf * 16, frame.shape[1])
    sq = resample.resize_rgb8(frame, fs, fs)
    a = pipeline.model_image_render(sds, "video", sq, rf, 0, True) if method != 1 else None
    b = D.colorize_frame(dsd, sq, input_size=(rf // 2) * 32, **small) if method != 0 else None
    if method == 0:
        c = a
    elif method == 1:
        c = b
    elif method == 2:
        c = imaging.pil_blend(a, b, w_merge)
    elif method == 3:
        ccm = tweaks.constrained_chroma_merge(a, b, 0.15, w_merge, True)
        c = imaging.pil_blend(ccm, imaging.pil_blend(a, b, min(w_merge, 0.6)), 0.3)
    elif method == 5:
        luma = pipeline.get_image_luma(b)
        ww = max(w_merge * pow(luma / 0.8, 1.0), 0.15) if luma < 0.8 else w_merge
        c = imaging.pil_blend(a, b, ww)
    else:
        c = tweaks.chroma_bound_adaptive_merge(a, b, 20, 24, w_merge, True)
    want = pipeline.post_process(resample.resize_rgb8(c, frame.shape[1], frame.shape[0]), frame)
    d = np.abs(got.astype(int) - want.astype(int))
    de = imaging.delta_e00_images(got, want)
    assert got.shape == frame.shape and (d <= 3).mean() > 0.97 and de.mean() < 1.0, (method, float((d <= 3).mean()), float(de.mean()), int(d.max()))
