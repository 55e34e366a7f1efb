"""-m gpu: BASELINE.json configs[2], [3], [4] at their config sizes against the all-oracle graph on a 1080p frame, and the
weight-FILE paths on the GPU: ModelImageRender(package_dir=...) reading models/Colorize*_gen.pth in both
Learner.save layouts and a converted .havc blob, DDColorRender(model_dir=...), ColorMNetRender(project_dir=...).

Tolerances: fp16 activations / fp32 MFMA accumulation vs the fp32 oracle; DDColor itself is parity-UNPINNED (oracle/ddcolor.py)."""
import os
import subprocess
import sys

import numpy as np
import pytest

from oracle import imaging, pipeline, resample
from vsdeoldify_amd.clip import synthetic_gray_frame
from vsdeoldify_amd.synth import synth_ddcolor_state_dict, synth_state_dict

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _stats(got, ref):
    de = imaging.delta_e00_images(got, ref)
    d = np.abs(got.astype(np.int32) - ref.astype(np.int32))
    return float(de.mean()), float(np.percentile(de, 99)), float((d <= 2).mean())


@pytest.mark.parametrize("config", ["c3", "c4"])
def test_ddcolor_configs_at_full_size_match_the_oracle_graph(ctx, config):
    """c3: HAVC_colorizer(method=1, ddcolor_p=[1,32,..]) -> DDColor large (36 blocks, 9 decoder layers) at input 512 (vsslib/vsmodels.py:302,353-363);
    c4: defaults, method 2: DeOldify video + DDColor at 384 x 384, Image.blend 0.4 (__init__.py:2502-2523, vsslib/mcomb.py:171-172);
    both on a 1080p frame: Spline64 squash -> models -> merge -> Spline64 back + luma of the source."""
    from oracle import ddcolor as D
    from vsdeoldify_amd import havc
    frame = synthetic_gray_frame(3, 1920, 1080)
    dsd, vsd = synth_ddcolor_state_dict(1), {"video": synth_state_dict("wide", 1)}
    from oracle import tweaks
    hue_adj = "300:360|0.8,0.1"                                  # HAVC_colorizer's default ddtweak_p[1] (__init__.py:2293, vsmodels.py:365-366)
    dd = dict(ddtweak_p=(havc.DEF_TWEAK_p, hue_adj))
    if config == "c3":
        col = havc.HAVCFrameColorizer(method=1, ddcolor_p=(1, 32, 1.0, 0.0, True), ddcolor_state_dict=dsd, **dd)
        fs = 512
    else:
        col = havc.HAVCFrameColorizer(method=2, mweight=0.4, deoldify_p=(0, 24, 1.0, 0.0), ddcolor_p=(1, 24, 1.0, 0.0, True), state_dicts=vsd, ddcolor_state_dict=dsd, **dd)
        fs = 384
    got = col.colorize(frame)
    sq = resample.resize_rgb8(frame, fs, fs)
    b = tweaks.adjust_hue_range(D.colorize_frame(dsd, sq, input_size=fs), hue_adj)
    c = b if config == "c3" else pipeline.combine_models(pipeline.model_image_render(vsd, "video", sq, 24, 0, True), b, 2, 0.4)
    ref = pipeline.post_process(resample.resize_rgb8(c, 1920, 1080), frame)
    mean, p99, w2 = _stats(got, ref)
    print(f"{config} @1080p: mean dE00 {mean:.4f} p99 {p99:.3f} bytes within 2 LSB {w2:.5f}")
    # fast path (fp16 activations): measured on MI355X + 15 % (this frame, round 5: c3 mean 0.152 / p99 1.940, c4 0.198 / 2.110); the CONTRACT (p99 < 1.0,
    # >= 99 % of the pixels below 1.0) is met by precision="precise": tests/test_gpu_precise_models.py runs these two graphs in that mode
    lim = {"c3": (0.175, 2.23), "c4": (0.228, 2.43)}[config]
    assert got.shape == frame.shape and mean < lim[0] and p99 < lim[1] and w2 > 0.995, (config, mean, p99, w2)


def test_colormnet_config_at_full_size_matches_the_oracle_loop(ctx):
    """c5: HAVC_deepex(ex_model=0) at render_speed 'medium': 1080p -> 384 x 216 -> ColorMNet (exemplar with frame 0) -> 1080p + luma; three frames"""
    import torch
    from oracle import colormnet_clip
    from vsdeoldify_amd import _native as nat
    from vsdeoldify_amd.colormnet_net import ColorMNetNetwork
    from vsdeoldify_amd.colormnet_render import ColorMNetRender
    from vsdeoldify_amd.synth import synth_colormnet_state_dict
    sd = synth_colormnet_state_dict(1)
    frames = [synthetic_gray_frame(i, 1920, 1080) for i in range(3)]
    smalls = [resample.resize_rgb8(f, 384, 216) for f in frames]
    l = smalls[0][..., 0].astype(np.float32)
    ref_img = np.clip(np.stack([l * 1.05 + 12, l * 0.9, l * 0.7 + 25], -1), 0, 255).astype(np.uint8)
    rnd = ColorMNetRender(vid_length=10000, reset_on_ref_update=False, network=ColorMNetNetwork(sd))
    want = colormnet_clip.colorize_clip({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, smalls, {0: ref_img}, vid_length=10000)
    for t, (f, s) in enumerate(zip(frames, smalls)):
        from PIL import Image
        rnd.set_ref_frame(Image.fromarray(ref_img) if t == 0 else None, False)
        col = np.asarray(rnd.colorize_frame(t, Image.fromarray(s)))
        up = np.empty_like(f)
        nat.check(ctx.lib.havc_spline64_resize(ctx.h, nat.as_ptr(np.ascontiguousarray(col)), 384, 216, nat.as_ptr(up), 1920, 1080, nat.as_ptr(f)), ctx.h)
        ref = pipeline.post_process(resample.resize_rgb8(want[t], 1920, 1080), f)
        mean, p99, w2 = _stats(up, ref)
        print(f"c5 frame {t} @1080p: mean dE00 {mean:.4f} p99 {p99:.3f} bytes within 2 LSB {w2:.5f}")
        assert mean < 0.15 and p99 < 1.5 and w2 > 0.998, (t, mean, p99, w2)


def test_deepex_colormnet_data_path_with_borders(ctx):
    """DeepExColorMNet = HAVC_deepex(ex_model=0)'s data path: a 4:3 clip gets black side borders up to 16:9 (SmartResizeColorizer), Spline64 to
    256 x 144 ('fast'), ColorMNet, Spline64 back, crop, luma of the source; vs the same flow assembled from the oracle pieces"""
    import torch
    from oracle import colormnet_clip
    from vsdeoldify_amd.colormnet_render import DeepExColorMNet
    from vsdeoldify_amd.synth import synth_colormnet_state_dict
    sd = synth_colormnet_state_dict(1)
    clip = np.stack([synthetic_gray_frame(i, 400, 300) for i in range(2)])
    ref = np.clip(clip[0].astype(np.float32) * [1.05, 0.9, 0.75] + [10, 0, 12], 0, 255).astype(np.uint8)
    dx = DeepExColorMNet(vid_length=100, render_speed="fast", render_vivid=False, state_dict=sd)
    assert dx._borders(300, 400) == (0, 67) and dx._borders(1080, 1920) == (0, 0) and dx._borders(800, 1920)[0] > 0
    got = dx.colorize_clip(clip, {0: ref})
    pad = lambda a: np.pad(a, ((0, 0), (67, 67), (0, 0)))
    smalls = [resample.resize_rgb8(pad(f), 256, 144) for f in clip]
    cols = colormnet_clip.colorize_clip({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, smalls, {0: resample.resize_rgb8(pad(ref), 256, 144)}, vid_length=100)
    for t in range(2):
        want = pipeline.post_process(np.ascontiguousarray(resample.resize_rgb8(cols[t], 400 + 134, 300)[:, 67:467]), clip[t])
        mean, p99, w2 = _stats(got[t], want)
        print(f"deepex 4:3 frame {t}: mean dE00 {mean:.4f} p99 {p99:.3f} bytes within 2 LSB {w2:.5f}")
        assert got[t].shape == clip[t].shape and mean < 0.3 and p99 < 2.0, (t, mean, p99)


def test_deepex_lookahead_reads_device_frames_behind_their_producer(ctx):
    """DeepExColorMNet squashes announced frames on the LOOK-AHEAD context's stream.  Frames that are still being written
    on the main context's stream when colorize_frames is called (the output of another model handed straight to deepex: here a Spline64 chain
    enqueued right before the call, on buffers that held other content) must be read behind their producer: same bytes as with host frames."""
    from vsdeoldify_amd.colormnet_net import ColorMNetNetwork
    from vsdeoldify_amd.colormnet_render import DeepExColorMNet
    from vsdeoldify_amd.device import DeviceImage
    from vsdeoldify_amd.havc import spline64
    from vsdeoldify_amd.synth import synth_colormnet_state_dict
    net = ColorMNetNetwork(synth_colormnet_state_dict(1))
    n, H, W = 6, 540, 960
    big = np.stack([synthetic_gray_frame(40 + i, 1920, 1080) for i in range(n)])
    host = [resample.resize_rgb8(f, W, H) for f in big]                                    # what the device chain below produces (checked)
    ref = np.clip(host[0].astype(np.float32) * [1.05, 0.9, 0.75] + [10, 0, 12], 0, 255).astype(np.uint8)
    base = [np.asarray(o) for o in DeepExColorMNet(vid_length=100, render_speed="fast", render_vivid=False, network=net).colorize_frames(host, {0: ref})]
    for rep in range(3):
        dbig = DeviceImage.from_numpy(net.ctx, big)
        junk = [DeviceImage.from_numpy(net.ctx, np.full((H, W, 3), 255 - 40 * rep, np.uint8)) for _ in range(n)]
        del junk                                                                           # their buffers return to the pool, holding stale content
        dx = DeepExColorMNet(vid_length=100, render_speed="fast", render_vivid=False, network=net)
        frames = [spline64(net.ctx, dbig.frame(i), W, H) for i in range(n)]              # only ENQUEUED on the main context's stream
        got = [o.numpy() for o in dx.colorize_frames(frames, {0: ref})]
        for t in range(n):
            assert np.array_equal(frames[t].numpy(), host[t]) or np.abs(frames[t].numpy().astype(int) - host[t]).max() <= 1
            d = np.abs(got[t].astype(int) - base[t].astype(int))
            assert (d <= 2).mean() > 0.999, (rep, t, float((d <= 2).mean()), int(d.max()))


def test_colormnet_at_the_slow_deepex_size(ctx):
    """HAVC_deepex render_speed 'slow': 512 x 288 (deepex/__init__.py:64-65) -> padded to 560 x 336 inside the step: odd 1/16 grid (21 x 35), DINOv2 grid
    24 x 40 interpolated to it, pads on both axes; exemplar + one propagated frame vs the oracle loop"""
    import torch
    from PIL import Image
    from oracle import colormnet_clip
    from vsdeoldify_amd.colormnet_net import ColorMNetNetwork
    from vsdeoldify_amd.colormnet_render import ColorMNetRender
    from vsdeoldify_amd.synth import synth_colormnet_state_dict
    sd = synth_colormnet_state_dict(1)
    smalls = [resample.resize_rgb8(synthetic_gray_frame(i, 1920, 1080), 512, 288) for i in range(2)]
    l = smalls[0][..., 0].astype(np.float32)
    ref_img = np.clip(np.stack([l * 0.95 + 20, l * 0.9, l * 0.8 + 15], -1), 0, 255).astype(np.uint8)
    want = colormnet_clip.colorize_clip({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, smalls, {0: ref_img}, vid_length=100)
    rnd = ColorMNetRender(vid_length=100, reset_on_ref_update=False, network=ColorMNetNetwork(sd))
    for t, s in enumerate(smalls):
        rnd.set_ref_frame(Image.fromarray(ref_img) if t == 0 else None, False)
        got = np.asarray(rnd.colorize_frame(t, Image.fromarray(s)))
        mean, p99, w2 = _stats(got, want[t])
        print(f"slow size frame {t}: mean dE00 {mean:.4f} p99 {p99:.3f} bytes within 2 LSB {w2:.5f}")
        assert got.shape == s.shape and mean < 0.3 and p99 < 2.0, (t, mean, p99)


# ---- weight files (f4) ----
def _save_pth(path, sd, layout):
    import torch
    t = {k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}
    torch.save({"model": t, "opt": {"state": {}}} if layout == "learner" else t, path)


def test_model_image_render_reads_pth_and_havc_files(ctx, tmp_path):
    """Learner.load semantics (fastai/basic_train.py:264-286, deoldify/generators.py:12-21, deoldify/visualize.py:71-100): package_dir/models/
    ColorizeVideo_gen.pth as Learner.save writes it ({'model', 'opt'}), ColorizeStable_gen.pth as a bare state dict, then the stable
    checkpoint converted offline to a .havc blob (tools/convert_weights.py): the three renders colour a frame identically, and as the oracle does."""
    from PIL import Image
    from vsdeoldify_amd.render import ModelImageRender
    sds = {"video": synth_state_dict("wide", 1), "stable": synth_state_dict("wide", 2)}
    models = tmp_path / "models"
    models.mkdir()
    _save_pth(str(models / "ColorizeVideo_gen.pth"), sds["video"], "learner")
    _save_pth(str(models / "ColorizeStable_gen.pth"), sds["stable"], "bare")
    rf = 6
    r = np.random.default_rng(5)
    img = np.clip(128 + 45 * r.standard_normal((rf * 16, rf * 16, 1)), 0, 255).astype(np.uint8).repeat(3, -1)
    from_dicts = np.asarray(ModelImageRender(None, "stable", rf, 0.5, state_dicts=sds).get_transformed_image(Image.fromarray(img)))
    from_pth = np.asarray(ModelImageRender(str(tmp_path), "stable", rf, 0.5).get_transformed_image(Image.fromarray(img)))
    assert np.array_equal(from_pth, from_dicts)
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "convert_weights.py"), str(models / "ColorizeStable_gen.pth")])
    assert (models / "ColorizeStable_gen.havc").is_file()
    from_havc = np.asarray(ModelImageRender(str(tmp_path), "stable", rf, 0.5).get_transformed_image(Image.fromarray(img)))
    assert np.array_equal(from_havc, from_dicts)
    os.remove(models / "ColorizeStable_gen.pth")                              # a deployment that ships only the converted blob
    only_havc = np.asarray(ModelImageRender(str(tmp_path), "stable", rf, 0.5).get_transformed_image(Image.fromarray(img)))
    assert np.array_equal(only_havc, from_dicts)
    ref = pipeline.model_image_render(sds, "stable", img, rf, 0.5)
    de = imaging.delta_e00_images(from_pth, ref)
    assert de.mean() < 0.25 and np.percentile(de, 99) < 2.1
    with pytest.raises(FileNotFoundError):
        ModelImageRender(str(tmp_path / "nowhere"), "video", rf, 0)


def test_ddcolor_render_reads_the_checkpoint_file(ctx, tmp_path):
    """DDColorRender(model_dir=...): ddcolor_artistic.pth in the public layout {'params': state_dict} (vsdeoldify/__init__.py:2367-2371)"""
    import torch
    from vsdeoldify_amd.ddcolor import DDColorRender
    small = dict(depths=(1, 1, 2, 1), dec_layers=2)
    sd = synth_ddcolor_state_dict(2, **small)
    torch.save({"params": {k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}}, str(tmp_path / "ddcolor_artistic.pth"))
    r = np.random.default_rng(6)
    frame = np.clip(128 + 45 * r.standard_normal((96, 96, 1)), 0, 255).astype(np.uint8).repeat(3, -1)
    a = DDColorRender(model=1, input_size=96, state_dict=sd, **small).colorize_frame(frame)
    b = DDColorRender(model=1, input_size=96, model_dir=str(tmp_path), **small).colorize_frame(frame)
    assert np.array_equal(a, b)
    with pytest.raises(FileNotFoundError):
        DDColorRender(model=0, input_size=96, model_dir=str(tmp_path), **small)


def test_colormnet_render_reads_the_checkpoint_file(ctx, tmp_path):
    """ColorMNetRender(project_dir=...): weights/DINOv2FeatureV6_LocalAtten_s2_154000.pth as torch.save(state_dict) (colormnet_render.py:107-108,146-148)"""
    import torch
    from PIL import Image
    from vsdeoldify_amd.colormnet_render import WEIGHTS, ColorMNetRender
    from vsdeoldify_amd.synth import synth_colormnet_state_dict
    sd = synth_colormnet_state_dict(2)
    os.makedirs(tmp_path / "weights")
    torch.save({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, str(tmp_path / WEIGHTS))
    r = np.random.default_rng(7)
    gray = np.clip(128 + 45 * r.standard_normal((100, 150, 1)), 0, 255).astype(np.uint8).repeat(3, -1)
    ref = np.clip(gray.astype(np.float32) * [1.1, 0.9, 0.75], 0, 255).astype(np.uint8)
    outs = []
    for kw in (dict(state_dict=sd), dict(project_dir=str(tmp_path))):
        rnd = ColorMNetRender(vid_length=100, reset_on_ref_update=False, **kw)
        rnd.set_ref_frame(Image.fromarray(ref), False)
        outs.append(np.asarray(rnd.colorize_frame(0, Image.fromarray(gray))))
    assert np.array_equal(outs[0], outs[1]) and outs[0].shape == gray.shape
    with pytest.raises(FileNotFoundError):
        ColorMNetRender(vid_length=10, project_dir=str(tmp_path / "nowhere"))
