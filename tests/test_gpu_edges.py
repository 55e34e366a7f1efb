"""Edge cases of the C ABI on the GPU: empty and ragged batches, tiny / odd-sized images, out-of-memory and bad-argument
error behaviour (the reference: parameter errors raise, OOM returns the gray input with a warning, deoldify/filters.py:55-63)."""
import ctypes as C

import numpy as np
import pytest

from oracle import cvcolor, imaging, pipeline, tweaks
from vsdeoldify_amd import _native as nat

pytestmark = pytest.mark.gpu


def _rgb(h, w, seed):
    return np.random.default_rng(seed).integers(0, 256, (h, w, 3), dtype=np.uint8)


@pytest.mark.parametrize("hw", [(1, 1), (1, 7), (5, 3), (17, 31)])
def test_filters_on_tiny_and_odd_images(ctx, hw):
    from vsdeoldify_amd import imfilters as F
    a, b = _rgb(*hw, 1), _rgb(*hw, 2)
    assert np.array_equal(F.blend_np(ctx, a, b, 0.3), imaging.pil_blend(a, b, 0.3))
    assert np.array_equal(F.chroma_post_process_np(ctx, a, b), pipeline.post_process(a, b))
    assert np.array_equal(F.chroma_stabilizer_np(ctx, a, b, 0.2, 0.6), pipeline.chroma_stabilizer(a, b, 0.2, 0.6))
    assert np.array_equal(F.chroma_stabilizer_adaptive_np(ctx, a, b, 18, 22, 1.0), pipeline.chroma_stabilizer_adaptive(a, b, 18, 22, 1.0))
    assert np.array_equal(F.image_tweak_np(ctx, a, sat=0.7, cont=1.2, bright=15, hue=30.0), tweaks.image_tweak(a, sat=0.7, cont=1.2, bright=15, hue=30.0))
    assert np.array_equal(F.image_chroma_tweak_np(ctx, a, sat=0.7, bright=-0.2, hue=20), tweaks.np_image_chroma_tweak(a, sat=0.7, bright=-0.2, hue=20))
    assert np.array_equal(F.restore_color_gradient_np(ctx, a, b, 0.8, 30), tweaks.restore_color_gradient(a, b, 0.8, 30))
    assert abs(F.image_luma_np(ctx, a) - float(np.mean(cvcolor.rgb2yuv_u8(a)[:, :, 0]))) < 1e-9


def test_extreme_pixel_values(ctx):
    from vsdeoldify_amd import imfilters as F
    for v in (0, 255):
        a = np.full((8, 8, 3), v, np.uint8)
        b = 255 - a
        assert np.array_equal(F.blend_np(ctx, a, b, 0.5), imaging.pil_blend(a, b, 0.5))
        assert np.array_equal(F.chroma_stabilizer_np(ctx, a, b, 0.15, 1.0), pipeline.chroma_stabilizer(a, b, 0.15, 1.0))
        assert np.array_equal(F.image_tweak_np(ctx, a, sat=2.0, cont=3.0, bright=200), tweaks.image_tweak(a, sat=2.0, cont=3.0, bright=200))
        assert np.array_equal(F.luma_adjusted_levels_np(ctx, a, luma_min=0.5, gamma=0.6, gamma_luma_min=0.9), tweaks.luma_adjusted_levels(a, luma_min=0.5, gamma=0.6, gamma_luma_min=0.9))


def test_empty_and_ragged_batches(ctx):
    """0 frames is a no-op; 5 frames through a max_batch-2 net = 2 + 2 + 1 and equals frame-by-frame results"""
    from vsdeoldify_amd.render import GeneratorRuntime
    from vsdeoldify_amd.synth import synth_state_dict
    rt = GeneratorRuntime(ctx, synth_state_dict("wide", 1), "wide")
    try:
        net = rt.net(64, 2)
        frames = np.stack([_rgb(64, 64, s)[:, :, :1].repeat(3, -1) for s in range(5)])
        out = np.empty_like(frames)
        nat.check(ctx.lib.havc_deoldify_frames(ctx.h, net.h, None, 0.0, 1, nat.as_ptr(frames), nat.as_ptr(out), 0), ctx.h)      # n = 0
        nat.check(ctx.lib.havc_deoldify_frames(ctx.h, net.h, None, 0.0, 1, nat.as_ptr(frames), nat.as_ptr(out), 5), ctx.h)
        one = np.empty_like(frames[:1])
        for i in range(5):
            nat.check(ctx.lib.havc_deoldify_frames(ctx.h, net.h, None, 0.0, 1, nat.as_ptr(frames[i:i + 1]), nat.as_ptr(one), 1), ctx.h)
            assert np.array_equal(one[0], out[i]), i
    finally:
        rt.close()


def test_bad_arguments_are_refused_not_crashed(ctx):
    lib = ctx.lib
    a = _rgb(4, 4, 0)
    out = np.empty_like(a)
    assert lib.havc_blend(ctx.h, None, nat.as_ptr(a), 0.5, nat.as_ptr(out), 4, 4) == -1                     # HAVC_E_INVALID
    assert lib.havc_blend(ctx.h, nat.as_ptr(a), nat.as_ptr(a), 0.5, nat.as_ptr(out), 0, 4) == -1
    assert lib.havc_image_tweak(ctx.h, nat.as_ptr(a), nat.as_ptr(out), 4, 4, 0, 1.0, 1.0, 1.0, None, 3) == -1   # ranges missing
    assert lib.havc_restore_color_gradient(ctx.h, nat.as_ptr(a), nat.as_ptr(a), nat.as_ptr(out), 4, 4, 1.0, 30, 0.0, 2.0, 7, 0) == -1
    assert lib.havc_pil_resize(ctx.h, nat.as_ptr(a), 4, 4, nat.as_ptr(out), 4, 4, 9) == -1
    msg = lib.havc_last_error(ctx.h)
    assert msg and b"resample" in msg
    with pytest.raises(Exception):
        nat.check(-1, ctx.h)
    # a plan that references a buffer that does not exist is rejected at net creation
    from vsdeoldify_amd.plan import PlanBuilder
    b = PlanBuilder()
    x = b.tensor(8, 8, 8)
    b.affine("bad", x, x, -1, -1, relu=False)
    ops, bufs = b.finish()
    ops["dst"] = 99
    w = nat.Weights(ctx, b"\0" * 256)
    with pytest.raises(Exception):
        nat.Net(ctx, w, ops, bufs, 0, 0, 8, 1)
    w.close()


def test_oom_returns_gray_input_like_the_reference(ctx, monkeypatch):
    """deoldify/filters.py:55-63: on an out-of-memory error _model_process logs a warning and returns the (squared, gray) model
    image -- and filter() CONTINUES with it (un-square to the source size, post-process), so the caller still gets an image of
    the source size."""
    from PIL import Image
    from oracle import pipeline
    from vsdeoldify_amd import render
    from vsdeoldify_amd.synth import synth_state_dict
    mir = render.ModelImageRender("", "video", render_factor=4, state_dicts={"video": synth_state_dict("wide", 1)})
    img = Image.fromarray(_rgb(48, 80, 3))
    assert mir.get_transformed_image(img).size == img.size

    def boom(S, max_batch=1, low_latency=False):
        raise nat.HavcOutOfMemory("simulated HAVC_E_OOM")
    monkeypatch.setattr(mir._video, "net", boom)
    for im in (img, img.resize((64, 64))):                       # non-square source, and a source already at the render size
        got = mir.get_transformed_image(im)
        gray = im.resize((64, 64), resample=Image.BILINEAR).convert("LA").convert("RGB")
        want = pipeline.post_process(np.asarray(gray.resize(im.size, resample=Image.BILINEAR)), np.asarray(im))
        assert got.size == im.size and np.array_equal(np.asarray(got), want)
    with pytest.raises(FileNotFoundError):
        render.ModelImageRender("/nonexistent", "video", render_factor=4)


def test_stream_recreation_is_refused_once_a_handle_is_out():
    """havc_ctx_set_stream_priority / _cus destroy and re-create the context's streams (A/B switches of the ColorMNet look-ahead): legal on a fresh
    context, refused with HAVC_E_INVALID once havc_get_stream has handed a handle out -- a torch ExternalStream wrapped around it would dangle."""
    from vsdeoldify_amd.render import get_context
    c = get_context(0, ("stream-rule-test", 0))
    assert c.lib.havc_ctx_set_stream_priority(c.h, 0) == nat.HAVC_OK           # nothing exported yet
    assert c.stream_ptr() != 0
    assert c.lib.havc_ctx_set_stream_priority(c.h, 0) == nat.HAVC_E_INVALID
    assert c.lib.havc_ctx_set_stream_cus(c.h, 64) == nat.HAVC_E_INVALID
    assert b"handed out" in c.lib.havc_last_error(c.h)
