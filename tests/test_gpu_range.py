"""-m gpu: the fp16 contract's debug switch (include/havc_mi355.h havc_range_check_enable / HAVC_RANGE_CHECK=1; VERDICT r2 #4a).
The reference computes in fp32 end to end (deoldify/filters.py:45-68); this library stores every activation in fp16.  With the switch on,
every op's destination buffer is scanned: a run that left an inf / NaN anywhere fails with HavcRangeError instead of colouring a frame from
garbage, and the per-op abs-max gives the head-room of a checkpoint (tools/range_headroom.py -> profiles/)."""
import numpy as np
import pytest
from PIL import Image

from vsdeoldify_amd import _native as nat
from vsdeoldify_amd.render import ModelImageRender, get_context
from vsdeoldify_amd.synth import synth_state_dict

pytestmark = pytest.mark.gpu
WORKER = 7                                        # a context of its own: the switch must be on before the nets exist


def _img(rf, seed=0):
    r = np.random.default_rng(seed)
    return Image.fromarray(np.clip(128 + 45 * r.standard_normal((rf * 16, rf * 16, 1)), 0, 255).astype(np.uint8).repeat(3, -1))


def test_range_check_passes_on_sane_weights_and_reports_head_room():
    ctx = get_context(0, WORKER)
    ctx.range_check(True)
    try:
        rf = 5
        sd = synth_state_dict("wide", 1)
        r = ModelImageRender(None, "video", rf, 0, state_dicts={"video": sd}, worker=WORKER)
        out = np.asarray(r.get_transformed_image(_img(rf)))
        assert out.shape == (rf * 16, rf * 16, 3)
        net = r._video.net(rf * 16, 1)
        amax, bad = net.range_stats()
        assert bad.sum() == 0 and 0 < amax.max() < 65504 / 8, float(amax.max())      # synthetic weights keep >= 3 bits of head-room
        off = np.asarray(ModelImageRender(None, "video", rf, 0, state_dicts={"video": sd}).get_transformed_image(_img(rf)))
        assert np.array_equal(out, off)                                                # the check observes, it does not change a byte
    finally:
        ctx.range_check(False)


def test_range_check_fires_on_weights_that_overflow_fp16():
    """BatchNorm gains of the decoder x16: activations grow 16x per block and pass 65504 within a few layers -> inf in an fp16 buffer"""
    ctx = get_context(0, WORKER)
    sd = dict(synth_state_dict("wide", 1))
    for k in list(sd):
        if k.startswith(("layers.3.", "layers.4.", "layers.5.", "layers.6.")) and k.endswith((".2.weight", ".2.bias", "shuf.conv.1.weight")):
            sd[k] = (np.asarray(sd[k]) * 16).astype(np.float32)
    rf = 5
    quiet = np.asarray(ModelImageRender(None, "video", rf, 0, state_dicts={"video": sd}).get_transformed_image(_img(rf)))
    assert quiet.shape == (rf * 16, rf * 16, 3)                                        # without the switch: a frame comes back, rc 0, silently wrong
    ctx.range_check(True)
    try:
        r = ModelImageRender(None, "video", rf, 0, state_dicts={"video": sd}, worker=WORKER)
        with pytest.raises(nat.HavcRangeError) as e:
            r.get_transformed_image(_img(rf))
        assert "non-finite" in str(e.value) and "fp16 range" in str(e.value)
        amax, bad = r._video.net(rf * 16, 1).range_stats()
        assert bad.sum() > 0
    finally:
        ctx.range_check(False)


def test_range_check_must_be_enabled_before_the_net_exists():
    ctx = get_context(0, WORKER + 1)
    rf = 4
    r = ModelImageRender(None, "video", rf, 0, state_dicts={"video": synth_state_dict("wide", 1)}, worker=WORKER + 1)
    r.get_transformed_image(_img(rf))
    ctx.range_check(True)
    try:
        with pytest.raises(ValueError):
            r.get_transformed_image(_img(rf))
    finally:
        ctx.range_check(False)
    r.get_transformed_image(_img(rf))
