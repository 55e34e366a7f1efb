"""-m gpu: the frame-level boundary shapes of the reference around the models:
planar VapourSynth frames, the RGBH / RGBS call shape of vsddcolor.ddcolor, device-resident operands, and the pipelined
host-clip entry."""
import numpy as np
import pytest

from vsdeoldify_amd import _native as nat
from vsdeoldify_amd.synth import synth_state_dict

pytestmark = pytest.mark.gpu


def _planes(rgb, stride):
    """three uint8 planes with a row stride, like a VapourSynth RGB24 frame (vsslib/vsutils.py:60-63)"""
    h, w, _ = rgb.shape
    store = np.zeros((3, h, stride), np.uint8)
    store[:, :, :w] = rgb.transpose(2, 0, 1)
    return [store[p, :, :w] for p in range(3)]


def test_planar_interleave_round_trip(ctx):
    import ctypes as C
    r = np.random.default_rng(0)
    for (h, w, stride) in ((48, 64, 64), (37, 53, 64), (1, 1, 32)):
        rgb = r.integers(0, 256, (h, w, 3), dtype=np.uint8)
        pl = _planes(rgb, stride)
        out = np.empty_like(rgb)
        pin = (C.c_void_p * 3)(*[p.ctypes.data for p in pl])
        nat.check(ctx.lib.havc_planar_to_rgb8(ctx.h, pin, stride, nat.as_ptr(out), w, h), ctx.h)
        assert np.array_equal(out, np.dstack(pl))                                      # frame_to_np_array
        back = _planes(np.zeros_like(rgb), stride)
        pout = (C.c_void_p * 3)(*[p.ctypes.data for p in back])
        nat.check(ctx.lib.havc_rgb8_to_planar(ctx.h, nat.as_ptr(rgb), pout, stride, w, h), ctx.h)
        assert all(np.array_equal(back[p], rgb[:, :, p]) for p in range(3))            # np_array_to_frame


@pytest.mark.parametrize("modelname", ["video", "stable"])
def test_planar_frame_entry_equals_the_image_entry(ctx, modelname):
    """vs_sc_deoldify's selector body (frame_to_image -> get_transformed_image -> image_to_frame) on planes with a stride"""
    from PIL import Image
    from tests.test_gpu_deoldify import make_frame
    from vsdeoldify_amd.render import ModelImageRender
    sds = {"video": synth_state_dict("wide", 1), "stable": synth_state_dict("wide", 2)}
    r = ModelImageRender(None, modelname, 4, 0.5, state_dicts=sds)
    img = make_frame(64, 77)
    want = np.asarray(r.get_transformed_image(Image.fromarray(img)))
    got = r.get_transformed_planes(_planes(img, 128), _planes(np.zeros_like(img), 96))
    assert np.array_equal(np.dstack(got), want)
    with pytest.raises(ValueError):
        r.get_transformed_planes(_planes(img, 128)[:2])


def test_ddcolor_rgbs_rgbh_call_shape(ctx):
    """vsddcolor.ddcolor is called with an RGBH / RGBS clip (vsslib/vsmodels.py:353-363): float planes in, float planes out,
    quantised to RGB24 afterwards by zimg.  Same colours as the u8 entry (which truncates x * 255): floor(out * 255) == u8."""
    from vsdeoldify_amd.ddcolor import DDColorRender
    from vsdeoldify_amd.synth import synth_ddcolor_state_dict
    small = dict(depths=(1, 1, 2, 1), dec_layers=3)
    dd = DDColorRender(1, 64, 0, state_dict=synth_ddcolor_state_dict(1, **small), **small)
    r = np.random.default_rng(3)
    frame = np.clip(r.normal(120, 50, (64, 64, 1)), 0, 255).astype(np.uint8).repeat(3, -1)
    u8 = dd.colorize_frame(frame)
    planes32 = (frame.transpose(2, 0, 1).astype(np.float32) / np.float32(255))
    out32 = dd.colorize_planar_float(planes32)
    assert out32.dtype == np.float32 and out32.shape == (3, 64, 64) and out32.min() >= 0 and out32.max() <= 1
    d = np.abs(np.floor(out32.transpose(1, 2, 0).astype(np.float64) * 255.0).astype(int) - u8.astype(int))
    assert d.max() <= 1 and (d > 0).mean() < 2e-3, (int(d.max()), float((d > 0).mean()))
    out16 = dd.colorize_planar_float(planes32.astype(np.float16))
    assert out16.dtype == np.float16 and np.abs(out16.astype(np.float32) - out32).max() <= 1.0 / 1024
    with pytest.raises(ValueError):
        dd.colorize_planar_float(frame.transpose(2, 0, 1))                       # u8 is not the RGBH / RGBS shape


def test_device_operands_match_host_operands(ctx):
    """every filter takes host or device pointers per operand; device results stay in HBM (no copy, no sync) and equal the host path"""
    from vsdeoldify_amd import imfilters as F, mcomb
    from vsdeoldify_amd.device import DeviceImage
    r = np.random.default_rng(8)
    a = r.integers(0, 256, (40, 56, 3), dtype=np.uint8)
    b = r.integers(0, 256, (40, 56, 3), dtype=np.uint8)
    da, db = DeviceImage.from_numpy(ctx, a), DeviceImage.from_numpy(ctx, b)
    cases = [
        lambda x, y: F.blend_np(ctx, x, y, 0.37),
        lambda x, y: F.chroma_post_process_np(ctx, x, y),
        lambda x, y: F.chroma_stabilizer_np(ctx, x, y, 0.2, 0.6),
        lambda x, y: F.chroma_stabilizer_adaptive_np(ctx, x, y, 18, 22, 0.7),
        lambda x, y: F.chroma_temporal_limiter_np(ctx, x, y, 0.1),
        lambda x, y: F.luma_merge_np(ctx, x, y, 1, 60, 0.008),
        lambda x, y: F.image_tweak_np(ctx, x, sat=0.8, cont=1.1, bright=5, hue=20, hue_range="280:360,0:30"),
        lambda x, y: F.image_chroma_tweak_np(ctx, x, sat=0.9, bright=-0.1, hue_adjust="green|0.2,-0.4"),
        lambda x, y: F.luma_adjusted_levels_np(ctx, x, 0.3, 0.8, 0.6, 1.5, 0.5),
        lambda x, y: F.restore_color_gradient_np(ctx, x, y, 0.8, 30, 0.0, 2.0),
        lambda x, y: F.color_temporal_stabilizer_np(ctx, [x, y, x], [30, 40, 30]),
        lambda x, y: mcomb.constrained_chroma_merge(x, y, 0.5, 0.2, True),
        lambda x, y: mcomb.luma_masked_merge(x, y, None, 0.3, 0.7, 0.5),
        lambda x, y: mcomb.adaptive_luma_merge(x, y, 0.8, 1.0, 0.5, 0.15),
        lambda x, y: mcomb.chroma_retention_frame(x, y),
        lambda x, y: mcomb.chroma_bound_adaptive_merge(x, y, True, 20, 24, 0.5),
    ]
    for i, fn in enumerate(cases):
        host = fn(a, b)
        dev = fn(da, db)
        assert isinstance(dev, DeviceImage), i
        assert np.array_equal(dev.numpy(), host), i
        mixed = fn(da, b)                                   # one device operand is enough: the other is uploaded once
        assert np.array_equal(mixed.numpy(), host), i
    assert abs(F.image_luma_np(ctx, da) - F.image_luma_np(ctx, a)) == 0
    # a stack of frames as ONE operand of a purely per-pixel filter
    stack = np.stack([a, b, a])
    dstack_ = DeviceImage.from_numpy(ctx, stack)
    got = F.blend_np(ctx, dstack_, DeviceImage.from_numpy(ctx, stack[::-1].copy()), 0.25).numpy()
    assert np.array_equal(got, np.stack([F.blend_np(ctx, stack[i], stack[2 - i], 0.25) for i in range(3)]))
    with pytest.raises(ValueError):
        F.image_luma_np(ctx, dstack_)                       # frame-level statistic: one frame at a time


@pytest.mark.parametrize("pinned", [True, False])
def test_pipelined_host_clip_equals_resident_clip(ctx, pinned):
    """havc_colorize_clip_host (three streams, double-buffered staging) == havc_colorize_clip on the same frames; 5 frames in
    batches of 2 exercise both staging slots and a ragged last batch."""
    from vsdeoldify_amd.clip import ClipColorizer, synthetic_gray_frame
    sds = {"video": synth_state_dict("wide", 1), "stable": synth_state_dict("wide", 2)}
    cc = ClipColorizer("stable", 6, 0.5, device_index=0, state_dicts=sds, max_batch=2)
    frames = np.stack([synthetic_gray_frame(i, 320, 180) for i in range(5)])
    want = cc.colorize(frames)
    if pinned:
        hin, hout = ctx.host_alloc(frames.nbytes), ctx.host_alloc(frames.nbytes)
        try:
            src = hin.reshape(frames.shape)
            src[...] = frames
            got = cc.colorize_host(src, out=hout.reshape(frames.shape)).copy()
        finally:
            ctx.host_free(hin)
            ctx.host_free(hout)
    else:
        got = cc.colorize_host(frames)
    assert np.array_equal(got, want)
    assert np.array_equal(cc.colorize_host(frames[:1]), want[:1])


def test_worker_contexts_run_concurrently_and_share_weights(ctx):
    """VERDICT r1 #10 / SURVEY §5: VapourSynth calls the selector from several worker threads.  One context per worker (own mutex,
    streams, workspace, activation arena), ONE packed weight blob per device: two threads colour different frames at the same
    time and get exactly what a single thread gets; the second worker adds activations but no second copy of the weights."""
    import threading
    from PIL import Image
    from tests.test_gpu_deoldify import make_frame
    from vsdeoldify_amd import render
    from vsdeoldify_amd.render import ModelImageRender
    sds = {"video": synth_state_dict("wide", 1), "stable": synth_state_dict("wide", 2)}
    frames = [make_frame(64, 200 + i) for i in range(6)]
    ref = ModelImageRender(None, "stable", 4, 0.5, state_dicts=sds)
    want = [np.asarray(ref.get_transformed_image(Image.fromarray(f))) for f in frames]
    workers = [ModelImageRender(None, "stable", 4, 0.5, state_dicts=sds, worker=w) for w in (1, 2)]
    assert workers[0].ctx is not workers[1].ctx and workers[0].ctx is not ctx
    assert workers[0]._video.weights is workers[1]._video.weights                   # shared blob
    w_bytes = len(workers[0]._video.gen.blob) + len(workers[0]._second.gen.blob)
    assert workers[1].ctx.stats().bytes_resident < w_bytes                          # worker 2 holds no weights of its own
    got, errs = {}, []

    def run(w, idxs):
        try:
            for _ in range(3):                                                     # several rounds: overlap is certain
                for i in idxs:
                    got[i] = np.asarray(workers[w].get_transformed_image(Image.fromarray(frames[i])))
        except Exception as e:                                                     # pragma: no cover
            errs.append(e)
    ts = [threading.Thread(target=run, args=(0, [0, 2, 4])), threading.Thread(target=run, args=(1, [1, 3, 5]))]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errs, errs
    assert all(np.array_equal(got[i], want[i]) for i in range(6))


@pytest.mark.parametrize("coalesce", [6, 2])
def test_coalesced_per_frame_calls_match_separate_calls(ctx, coalesce, monkeypatch):
    """The reference's call shape is ONE get_transformed_image per frame from each VapourSynth worker thread.  A render built with
    coalesce=N merges the concurrent calls of N threads into batches (havc_batcher): every caller gets exactly the bytes of a call of
    its own, and fewer batches than calls were run."""
    import threading
    from PIL import Image
    from tests.test_gpu_deoldify import make_frame
    from vsdeoldify_amd.render import ModelImageRender
    sds = {"video": synth_state_dict("wide", 1), "stable": synth_state_dict("wide", 2)}
    monkeypatch.setenv("HAVC_COALESCE_WAIT_US", "50000")     # a leader waits up to 50 ms for its batch to fill: coalescing is certain
    T, K = 6, 4
    start = threading.Barrier(T)
    frames = [make_frame(64, 300 + i) for i in range(T * K)]
    ref = ModelImageRender(None, "stable", 4, 0.5, state_dicts=sds)
    want = [np.asarray(ref.get_transformed_image(Image.fromarray(f))) for f in frames]
    shared = ModelImageRender(None, "stable", 4, 0.5, state_dicts=sds, coalesce=coalesce)      # 2: more caller threads than a batch holds
    got, errs = {}, []

    def run(t):
        try:
            start.wait()
            for k in range(K):
                i = t * K + k
                got[i] = np.asarray(shared.get_transformed_image(Image.fromarray(frames[i])))
        except Exception as e:                                                     # pragma: no cover
            errs.append(e)
    ts = [threading.Thread(target=run, args=(t,)) for t in range(T)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errs, errs
    assert all(np.array_equal(got[i], want[i]) for i in range(T * K))
    calls, batches = shared._batcher(64, True).stats()
    assert calls == T * K and batches < calls, (calls, batches)
    # a lone caller is served after the wait window, alone
    one = np.asarray(shared.get_transformed_image(Image.fromarray(frames[0])))
    assert np.array_equal(one, want[0])
    # frames of another size bypass the batcher (host Pillow stretch, as the reference)
    odd = Image.fromarray(make_frame(64, 1)).resize((80, 48))
    assert shared.get_transformed_image(odd).size == (80, 48)
