"""Zhang et al. colorizers (eccv16 / siggraph17): oracle pinning (CPU) and HIP parity (-m gpu).

Tolerance of the GPU path vs the CPU oracle (fp16 activations, fp32 accumulate; colour maths in fp64 on both sides):
ab maps: max |diff| <= 2.5 (of a +-110 range; the synthetic eccv16 logits are deliberately peaky), mean <= 0.25;  final uint8 image: mean CIEDE2000 < 0.5, p99 < 2.5,
>= 97 % of bytes within +-1 LSB (the reference truncates x*255).  PIL BICUBIC / BILINEAR resize: bit-exact.
"""
import json
import os

import numpy as np
import pytest
import torch

from oracle import imaging, pilresize, zhang
from tests.conftest import GOLDEN
from vsdeoldify_amd.synth import synth_zhang_state_dict, zhang_state_dict_spec


def tsd(sd):
    return {k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}


@pytest.mark.parametrize("model", ["eccv16", "siggraph17"])
def test_zhang_spec_matches_reference(model):
    ref = json.load(open(os.path.join(GOLDEN, f"spec_{model}.json")))["keys"]
    spec = zhang_state_dict_spec(model)
    assert [k for k, _ in ref] == list(spec.keys()) and all(tuple(s) == tuple(spec[k]) for k, s in ref)


@pytest.mark.parametrize("model,S", [("eccv16", 64), ("eccv16", 96), ("siggraph17", 64), ("siggraph17", 96)])
def test_zhang_restatement_matches_reference_forward(model, S):
    """oracle/zhang.py vs ECCVGenerator / SIGGRAPHGenerator executed from the reference tree."""
    g = np.load(os.path.join(GOLDEN, f"zhang_{model}_S{S}.npz"))
    sd = tsd(synth_zhang_state_dict(model, int(g["seed"])))
    with torch.no_grad():
        y = (zhang.eccv16_forward if model == "eccv16" else zhang.siggraph17_forward)(sd, torch.from_numpy(g["x"])).numpy()
    assert np.abs(y - g["y"]).max() < 2e-3, np.abs(y - g["y"]).max()


def test_pil_resize_restatement_bit_exact():
    """oracle/pilresize.py vs Pillow itself (filters.py:37-41,70-73 BILINEAR; colorizers/util.py:21-22 BICUBIC)."""
    from PIL import Image
    r = np.random.default_rng(0)
    for (h, w), (oh, ow) in (((54, 96), (64, 64)), ((64, 64), (54, 96)), ((135, 240), (256, 256)), ((100, 100), (256, 256)),
                             ((256, 256), (256, 256)), ((33, 200), (256, 256))):
        a = r.integers(0, 256, (h, w, 3), dtype=np.uint8)
        for rs in (2, 3):
            ref = np.asarray(Image.fromarray(a).resize((ow, oh), resample=rs))
            assert np.array_equal(pilresize.resize(a, ow, oh, rs), ref), ((h, w), (oh, ow), rs)


def test_lab_round_trip_and_known_values():
    """skimage formulas restated (parity unpinned): white/black/grey known answers and rgb->lab->rgb round trip."""
    lab = zhang.rgb2lab(np.array([[[255, 255, 255], [0, 0, 0], [128, 128, 128]]], np.uint8))
    assert abs(lab[0, 0, 0] - 100.0) < 1e-3 and np.abs(lab[0, 0, 1:]).max() < 1e-2       # D65 white: a,b ~ 0 (skimage: ~5e-3)
    assert abs(lab[0, 1, 0]) < 1e-9 and abs(lab[0, 2, 0] - 53.585) < 1e-2
    r = np.random.default_rng(1)
    rgb = r.integers(0, 256, (32, 32, 3), dtype=np.uint8)
    back = np.rint(zhang.lab2rgb(zhang.rgb2lab(rgb)) * 255).astype(int)
    assert np.abs(back - rgb.astype(int)).max() <= 1
    # the metric's own Lab (oracle/imaging.py) agrees with this one
    assert np.abs(imaging.srgb_to_lab(rgb) - zhang.rgb2lab(rgb)).max() < 1e-2


# ------------------------------------------------------------------------------------------------------------------
gpu = pytest.mark.gpu


def frame(h, w, seed):
    r = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w]
    luma = np.clip(60 + 130 * xx / w + 30 * np.sin(yy / 7.0) + 10 * r.standard_normal((h, w)), 0, 255).astype(np.uint8)
    img = np.stack([luma, luma, luma], -1)
    img[..., 2] = np.clip(img[..., 2].astype(int) + r.integers(-5, 6, (h, w)), 0, 255)
    return img


@gpu
def test_gpu_pil_resize_bit_exact(ctx):
    from PIL import Image
    from vsdeoldify_amd.colorization import pil_resize_np
    r = np.random.default_rng(3)
    for (h, w), (oh, ow) in (((54, 96), (64, 64)), ((64, 64), (54, 96)), ((270, 480), (256, 256)), ((100, 60), (256, 256)),
                             ((256, 256), (256, 256)), ((560, 560), (1080, 1920))):
        a = r.integers(0, 256, (h, w, 3), dtype=np.uint8)
        for rs in (2, 3):
            ref = np.asarray(Image.fromarray(a).resize((ow, oh), resample=rs))
            assert np.array_equal(pil_resize_np(ctx, a, (ow, oh), rs), ref), ((h, w), (oh, ow), rs)


@gpu
@pytest.mark.parametrize("model", ["eccv16", "siggraph17"])
def test_gpu_zhang_ab_map(ctx, model):
    """network only: L (fp32) in -> ab map out, against the oracle forward at 256x256 and 64x64."""
    from vsdeoldify_amd import _native as nat
    from vsdeoldify_amd.zhang_net import ZhangGenerator
    sd = synth_zhang_state_dict(model, 5)
    gen = ZhangGenerator(sd, model)
    w = nat.Weights(ctx, gen.blob)
    for S in (64, 256):
        ops, bufs, i, o, names = gen.plan(S)
        net = nat.Net(ctx, w, ops, bufs, i, o, S, 2)
        imgs = np.stack([frame(S, S, 20 + S), frame(S, S, 21 + S)])
        net.upload(i, imgs)
        net.run_ops(0, len(ops), 2)
        ab = net.download(o, (2, S, S, 2), np.float32)
        net.close()
        for k in range(2):
            l_in = torch.Tensor(zhang.rgb2lab(imgs[k])[:, :, 0])[None, None]
            with torch.no_grad():
                ref = (zhang.eccv16_forward if model == "eccv16" else zhang.siggraph17_forward)(tsd(sd), l_in)[0].numpy().transpose(1, 2, 0)
            d = np.abs(ab[k] - ref)
            assert np.isfinite(ab[k]).all() and d.max() <= 2.5 and d.mean() <= 0.25, (model, S, d.max(), d.mean(), np.abs(ref).max())
    w.close()


@gpu
@pytest.mark.parametrize("model", ["eccv16", "siggraph17"])
@pytest.mark.parametrize("hw", [(256, 256), (135, 240), (300, 200)])
def test_gpu_model_colorization_frame(ctx, model, hw):
    """ModelColorization.colorize_frame drop-in vs oracle/zhang.colorize_frame (colorization/__init__.py:76-95)."""
    from vsdeoldify_amd.colorization import ModelColorization
    sd = synth_zhang_state_dict(model, 7)
    mc = ModelColorization(model, True, state_dict=sd)
    img = frame(hw[0], hw[1], 77)
    got = mc.colorize_frame(img)
    ref = zhang.colorize_frame(tsd(sd), model, img)
    de = imaging.delta_e00_images(got, ref)
    d = np.abs(got.astype(int) - ref.astype(int))
    print(f"zhang {model} {hw} fast: mean dE00 {de.mean():.4f} p99 {np.percentile(de, 99):.3f} bytes within 1 LSB {(d <= 1).mean():.4f} max |d| {d.max()}")
    assert got.shape == img.shape and de.mean() < 0.5 and np.percentile(de, 99) < 2.5 and (d <= 1).mean() >= 0.97, \
        (de.mean(), np.percentile(de, 99), (d <= 1).mean(), d.max())
    mc.close()


@gpu
def test_gpu_model_colorization_coalesced_calls(ctx, monkeypatch):
    """colorize_frame from several threads through ModelColorization(coalesce=N): merged into batches (havc_batcher kind 2), every caller
    gets the bytes of a call of its own."""
    import threading
    from vsdeoldify_amd.colorization import ModelColorization
    monkeypatch.setenv("HAVC_COALESCE_WAIT_US", "50000")     # a leader waits up to 50 ms for its batch to fill: coalescing is certain
    start = threading.Barrier(4)
    sd = synth_zhang_state_dict("eccv16", 9)
    imgs = [frame(120, 160, 300 + i) for i in range(8)]
    mc = ModelColorization("eccv16", True, state_dict=sd)
    want = [mc.colorize_frame(im) for im in imgs]
    mc.close()
    mc = ModelColorization("eccv16", True, state_dict=sd, coalesce=4)
    got, errs = {}, []

    def run(t):
        try:
            start.wait()
            for i in (t, t + 4):
                got[i] = mc.colorize_frame(imgs[i])
        except Exception as e:                                                     # pragma: no cover
            errs.append(e)
    ts = [threading.Thread(target=run, args=(t,)) for t in range(4)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errs, errs
    assert all(np.array_equal(got[i], want[i]) for i in range(8))
    calls, batches = mc._batchers[(120, 160)].stats()
    assert calls == 8 and batches < 8
    mc.close()
