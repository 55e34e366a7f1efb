"""CPU (-m "not gpu"): pin the oracle to the golden vectors produced by EXECUTING the reference
(tools/gen_golden.py) and to Pillow itself; known-answer tests for the CIEDE2000 metric."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import cvcolor, imaging, pipeline, unet
from tests.conftest import GOLDEN
from vsdeoldify_amd.synth import state_dict_spec, synth_state_dict


def tsd(sd):
    return {k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}


@pytest.mark.parametrize("arch", ["wide", "deep"])
def test_state_dict_spec_matches_reference(arch):
    ref = json.load(open(os.path.join(GOLDEN, f"spec_{arch}.json")))["keys"]
    spec = state_dict_spec(arch)
    assert [k for k, _ in ref] == list(spec.keys())
    assert all(tuple(s) == tuple(spec[k]) for k, s in ref)


@pytest.mark.parametrize("arch,S", [("wide", 32), ("wide", 48), ("wide", 80), ("deep", 32), ("deep", 48), ("deep", 80)])
def test_unet_restatement_matches_reference_forward(arch, S):
    """oracle/unet.py vs DynamicUnetWide/Deep executed from the reference tree (incl. the odd-size nearest branch)."""
    g = np.load(os.path.join(GOLDEN, f"unet_{arch}_S{S}.npz"))
    sd = tsd(synth_state_dict(arch, int(g["seed"])))
    with torch.no_grad():
        y = unet.unet_forward(sd, torch.from_numpy(g["x"]), arch).numpy()
    assert np.abs(y - g["y"]).max() < 2e-4, np.abs(y - g["y"]).max()


@pytest.mark.parametrize("arch", ["wide", "deep"])
def test_colorizer_filter_matches_reference(arch):
    """oracle/pipeline.colorizer_filter vs MasterFilter([ColorizerFilter]).filter (deoldify/filters.py)."""
    g = np.load(os.path.join(GOLDEN, f"filter_{arch}.npz"))
    sd, rf = synth_state_dict(arch, int(g["seed"])), int(g["render_factor"])
    for img, pp, raw in ((g["img"], g["out_pp"], g["out_raw"]), (g["sq"], g["sq_pp"], g["sq_raw"])):
        got_raw = pipeline.colorizer_filter(sd, arch, img, rf, do_post=False)
        got_pp = pipeline.colorizer_filter(sd, arch, img, rf, do_post=True)
        # same fp32 CPU math: bit-exact up to a handful of truncation flips from op-order differences
        for got, ref in ((got_raw, raw), (got_pp, pp)):
            d = np.abs(got.astype(int) - ref.astype(int))
            assert d.max() <= 1 and (d > 0).mean() < 2e-3, (d.max(), (d > 0).mean())


@pytest.mark.parametrize("modelname", ["video", "stable", "artistic"])
def test_model_image_render_matches_reference(modelname):
    """oracle/pipeline.model_image_render vs the reference's ModelImageRender.get_transformed_image
    (weights loaded by the reference through Learner.load from .pth files)."""
    g = np.load(os.path.join(GOLDEN, f"render_{modelname}.npz"))
    seeds = json.loads(str(g["seeds"]))
    sds = {"video": synth_state_dict("wide", seeds["video"])}
    if modelname == "stable":
        sds["stable"] = synth_state_dict("wide", seeds["stable"])
    if modelname == "artistic":
        sds["artistic"] = synth_state_dict("deep", seeds["artistic"])
    rf, w = int(g["render_factor"]), float(g["video_weight"])
    for do_post, ref in ((True, g["out"]), (False, g["out_nopp"])):
        got = pipeline.model_image_render(sds, modelname, g["img"], rf, w, do_post)
        d = np.abs(got.astype(int) - ref.astype(int))
        assert d.max() <= 1 and (d > 0).mean() < 2e-3, (d.max(), (d > 0).mean())


def test_imfilters_match_reference():
    """image_weighted_merge / chroma_post_process / chroma_stabilizer (vsslib/imfilters.py), bit-exact."""
    g = np.load(os.path.join(GOLDEN, "imfilters.npz"))
    a, b = g["a"], g["b"]
    for w in (0.2, 0.4, 0.5, 0.8):
        assert np.array_equal(imaging.pil_blend(a, b, w), g[f"merge_{w}"])
    assert np.array_equal(pipeline.chroma_post_process(a, b), g["chroma_post_process"])
    for alpha, wgt in ((0.15, 1.0), (0.2, 0.6), (0.05, 0.5)):
        assert np.array_equal(pipeline.chroma_stabilizer(a, b, alpha, wgt), g[f"chroma_stabilizer_{alpha}_{wgt}"])


def test_pillow_primitives_bit_exact():
    """Image.blend, convert('LA').convert('RGB') restatements vs Pillow itself (present in the image)."""
    from PIL import Image
    r = np.random.default_rng(0)
    a = r.integers(0, 256, (37, 53, 3), dtype=np.uint8)
    b = r.integers(0, 256, (37, 53, 3), dtype=np.uint8)
    for w in (0.0, 0.1, 0.2, 0.3, 0.4, 0.5, 0.6, 0.77, 0.8, 1.0):
        assert np.array_equal(imaging.pil_blend(a, b, w), np.asarray(Image.blend(Image.fromarray(a), Image.fromarray(b), w))), w
    assert np.array_equal(imaging.pil_gray_rgb(a), np.asarray(Image.fromarray(a).convert("LA").convert("RGB")))
    # every (a, b) byte pair for the weights the presets use (havc_utils.py:351-363)
    aa, bb = np.meshgrid(np.arange(256, dtype=np.uint8), np.arange(256, dtype=np.uint8))
    pa, pb = np.repeat(aa[..., None], 3, -1), np.repeat(bb[..., None], 3, -1)
    for w in (0.2, 0.3, 0.4, 0.5, 0.6, 0.7, 0.8):
        assert np.array_equal(imaging.pil_blend(pa, pb, w), np.asarray(Image.blend(Image.fromarray(pa), Image.fromarray(pb), w))), w


def test_yuv_fixed_point_properties():
    """OpenCV BT.601 restatement: gray stays gray, Y of gray == value, round trip within 1 LSB for in-gamut colours."""
    g = np.repeat(np.arange(256, dtype=np.uint8)[:, None, None], 3, -1)
    yuv = cvcolor.rgb2yuv_u8(g)
    assert np.array_equal(yuv[..., 0], g[..., 0]) and (yuv[..., 1:] == 128).all()
    assert np.array_equal(cvcolor.yuv2rgb_u8(yuv), g)
    r = np.random.default_rng(1)
    rgb = r.integers(0, 256, (64, 64, 3), dtype=np.uint8)
    yuv = cvcolor.rgb2yuv_u8(rgb)
    unsat = (yuv[..., 1:] > 0).all(-1) & (yuv[..., 1:] < 255).all(-1)
    back = cvcolor.yuv2rgb_u8(yuv)
    assert np.abs(back.astype(int) - rgb.astype(int))[unsat].max() <= 2
    # documented float formula within 1 LSB
    f = rgb.astype(np.float64)
    y = 0.299 * f[..., 0] + 0.587 * f[..., 1] + 0.114 * f[..., 2]
    u = np.clip(0.492 * (f[..., 2] - y) + 128, 0, 255)
    v = np.clip(0.877 * (f[..., 0] - y) + 128, 0, 255)
    assert np.abs(yuv[..., 0] - y).max() <= 1 and np.abs(yuv[..., 1] - u).max() <= 1.01 and np.abs(yuv[..., 2] - v).max() <= 1.01


# Sharma, Wu, Dalal (2005) CIEDE2000 test data (subset of the 34 published pairs)
SHARMA = [
    ((50.0000, 2.6772, -79.7751), (50.0000, 0.0000, -82.7485), 2.0425),
    ((50.0000, 3.1571, -77.2803), (50.0000, 0.0000, -82.7485), 2.8615),
    ((50.0000, 2.8361, -74.0200), (50.0000, 0.0000, -82.7485), 3.4412),
    ((50.0000, -1.3802, -84.2814), (50.0000, 0.0000, -82.7485), 1.0000),
    ((50.0000, 0.0000, 0.0000), (50.0000, -1.0000, 2.0000), 2.3669),
    ((50.0000, 2.4900, -0.0010), (50.0000, -2.4900, 0.0009), 7.1792),
    ((50.0000, 2.5000, 0.0000), (73.0000, 25.0000, -18.0000), 27.1492),
    ((50.0000, 2.5000, 0.0000), (56.0000, -27.0000, -3.0000), 31.9030),
    ((60.2574, -34.0099, 36.2677), (60.4626, -34.1751, 39.4387), 1.2644),
    ((63.0109, -31.0961, -5.8663), (62.8187, -29.7946, -4.0864), 1.2630),
    ((22.7233, 20.0904, -46.6940), (23.0331, 14.9730, -42.5619), 2.0373),
    ((90.9257, -0.5406, -0.9208), (88.6381, -0.8985, -0.7239), 1.5381),
    ((2.0776, 0.0795, -1.1350), (0.9033, -0.0636, -0.5514), 0.9082),
]


def test_ciede2000_known_answers():
    for l1, l2, want in SHARMA:
        got = float(imaging.ciede2000(np.array(l1), np.array(l2)))
        assert abs(got - want) < 1e-4, (l1, l2, got, want)
        assert abs(float(imaging.ciede2000(np.array(l2), np.array(l1))) - want) < 1e-4
