"""a17 / a19: image_tweak, luma_adjusted_levels, restore_color_gradient, the dark-frame red fix.
CPU: oracle/tweaks.py against vectors produced by executing the reference (tests/golden/tweaks.npz, tools/gen_golden.py)
and Pillow's HSV conversion restated bit for bit.  GPU: the HIP filters through the C ABI against the oracle."""
import numpy as np
import pytest
from PIL import Image

from oracle import tweaks

G = np.load(__file__.rsplit("/", 1)[0] + "/golden/tweaks.npz")


def _cases(key):
    return [(i, eval(str(c))) for i, c in enumerate(G[key])]       # repr()s of plain dicts written by tools/gen_golden.py


def test_oracle_matches_reference_vectors():
    base, col, gray = G["base"], G["col"], G["grayish"]
    for i, c in _cases("tweak_cases"):
        assert np.array_equal(tweaks.image_tweak(base, **c), G[f"tweak_{i}"]), c
    for i, c in _cases("levels_cases"):
        assert np.array_equal(tweaks.luma_adjusted_levels(base, **c), G[f"levels_{i}"]), c
    for i, c in _cases("restore_cases"):
        assert np.array_equal(tweaks.restore_color_gradient(col, gray, **c), G[f"restore_{i}"]), c
    lumas = []
    for i in range(4):
        assert np.array_equal(tweaks.constrained_chroma_merge(G[f"ccm_in1_{i}"], G[f"ccm_in2_{i}"], 0.2, 0.5), G[f"ccm_out_{i}"])
        lumas.append(float(G[f"ccm_luma_{i}"]))
    assert lumas[0] > 0.3 > lumas[1] > 0.2 > lumas[2] > 0.1 > lumas[3]      # every branch of mcomb.py:350-361 is covered


def test_gamma_branch_raises_like_the_reference():
    assert str(G["tweak_gamma_raises"]).startswith("ValueError")
    with pytest.raises(ValueError):
        tweaks.image_tweak(G["base"], gamma=0.8)


def test_pillow_hsv_restatement_bit_exact():
    """every (R, G, B) with R on a coarse grid plus random triples, both directions, against Pillow itself"""
    r = np.arange(256, dtype=np.uint8)
    for R in list(range(0, 256, 15)) + [1, 127, 128, 254, 255]:
        a = np.zeros((256, 256, 3), np.uint8)
        a[..., 0], a[..., 1], a[..., 2] = R, r[:, None], r[None, :]
        assert np.array_equal(tweaks.pil_rgb2hsv(a), np.asarray(Image.fromarray(a).convert("HSV")))
        assert np.array_equal(tweaks.pil_hsv2rgb(a), np.asarray(Image.fromarray(a, mode="HSV").convert("RGB")))


# ------------------------------------------------------------------------------------------------------------------
def _imgs(seed, h=96, w=128):
    r = np.random.default_rng(seed)
    base = r.integers(0, 256, (h, w, 3), dtype=np.uint8)
    col = np.clip(base.astype(int) + r.integers(-60, 61, base.shape), 0, 255).astype(np.uint8)
    gray = np.clip(base.mean(-1, keepdims=True) + r.integers(-25, 26, base.shape), 0, 255).astype(np.uint8)
    return base, col, gray


@pytest.mark.gpu
def test_gpu_image_tweak(ctx):
    from vsdeoldify_amd import imfilters as F
    base, _, _ = _imgs(5)
    cases = [dict(sat=0.8), dict(cont=1.3), dict(bright=40), dict(bright=-60, cont=0.7, sat=1.4), dict(hue=35.0), dict(hue=-120.0, sat=0.5),
             dict(sat=0.9, hue_range="280:360,0:30"), dict(sat=0.3, cont=1.2, hue_range="green,cyan"),
             dict(sat=0.7, bright=10, hue=10, cont=0.9), dict(sat=2.5, cont=2.0, bright=100), dict(hue=180.0), dict(hue=359.0, sat=0.0)]
    for c in cases:
        assert np.array_equal(F.image_tweak_np(ctx, base, **c), tweaks.image_tweak(base, **c)), c
    with pytest.raises(ValueError):
        F.image_tweak_np(ctx, base, gamma=0.8)
    # the golden vectors of the executed reference, straight through the C ABI
    for i, c in _cases("tweak_cases"):
        assert np.array_equal(F.image_tweak_np(ctx, G["base"], **c), G[f"tweak_{i}"]), c


@pytest.mark.gpu
def test_gpu_pillow_hsv_exhaustive_slice(ctx):
    """hue shift by 0 Pillow units is skipped, so shift by 256 * k is not available: use a +1/-1 pair of shifts instead and
    compare each against Pillow on planes that sweep two channels completely."""
    from vsdeoldify_amd import imfilters as F
    r = np.arange(256, dtype=np.uint8)
    for R in (0, 37, 128, 200, 255):
        a = np.zeros((256, 256, 3), np.uint8)
        a[..., 0], a[..., 1], a[..., 2] = R, r[:, None], r[None, :]
        for hue in (2.0, -91.0, 123.0):
            assert np.array_equal(F.image_tweak_np(ctx, a, hue=hue), tweaks.image_tweak(a, hue=hue)), (R, hue)


@pytest.mark.gpu
def test_gpu_luma_adjusted_levels(ctx):
    from vsdeoldify_amd import imfilters as F
    base, _, _ = _imgs(6)
    dark = (base * 0.3).astype(np.uint8)
    for img in (base, dark):
        for c in (dict(luma_min=0.6), dict(luma_min=0.7, gamma=0.7, gamma_luma_min=0.9, gamma_alpha=0.5), dict(gamma=1.4, gamma_luma_min=0.8),
                  dict(luma_min=0.2, gamma=0.5, gamma_luma_min=0.1), dict()):
            assert np.array_equal(F.luma_adjusted_levels_np(ctx, img, **c), tweaks.luma_adjusted_levels(img, **c)), c
    for i, c in _cases("levels_cases"):
        assert np.array_equal(F.luma_adjusted_levels_np(ctx, G["base"], **c), G[f"levels_{i}"]), c


@pytest.mark.gpu
def test_gpu_restore_color_gradient(ctx):
    from vsdeoldify_amd import imfilters as F
    _, col, gray = _imgs(7)
    for c in (dict(), dict(sat=0.8, tht=30, alpha=2.0), dict(sat=1.5, tht=60, weight=0.3, alpha=3.0), dict(sat=0.6, tht=20, weight=-0.4),
              dict(tht=30, return_mask=True), dict(sat=3.0, tht=10, alpha=1.0)):
        assert np.array_equal(F.restore_color_gradient_np(ctx, col, gray, **c), tweaks.restore_color_gradient(col, gray, **c)), c
    # algo 1 / 2 go through powf / exp: a mask value sitting on an integer may land on the other side (<= 1 LSB, rare)
    for c in (dict(tht=40, algo=1), dict(tht=25, alpha=1.5, algo=2), dict(tht=40, algo=1, return_mask=True), dict(tht=25, alpha=1.5, algo=2, return_mask=True)):
        d = np.abs(F.restore_color_gradient_np(ctx, col, gray, **c).astype(int) - tweaks.restore_color_gradient(col, gray, **c))
        assert d.max() <= 1 and (d > 0).mean() < 2e-3, (c, int(d.max()), float((d > 0).mean()))
    for i, c in _cases("restore_cases"):
        if c.get("algo", 0) == 0:
            assert np.array_equal(F.restore_color_gradient_np(ctx, G["col"], G["grayish"], **c), G[f"restore_{i}"]), c


@pytest.mark.gpu
def test_gpu_red_fix_and_chroma_retention(ctx):
    from vsdeoldify_amd import mcomb
    for i in range(4):                                              # the four luma branches, reference-executed vectors
        got = mcomb.constrained_chroma_merge(G[f"ccm_in1_{i}"], G[f"ccm_in2_{i}"], 0.5, 0.2, True)
        assert np.array_equal(got, G[f"ccm_out_{i}"]), i
    base, col, gray = _imgs(8)
    for img, scale in ((gray, 1.0), (gray, 0.3)):                   # inside / outside the standard luma band
        a = (img * scale).astype(np.uint8)
        luma = round(float(np.mean(__import__("oracle.cvcolor", fromlist=["x"]).rgb2yuv_u8(a)[:, :, 0])) / 255, 6)
        w, al = (0.0, 2.0) if 0.22 <= luma <= 0.78 else (-0.5, 4.0)
        ref = tweaks.restore_color_gradient(col, a, 0.8, 30, w, al)
        assert np.array_equal(mcomb.chroma_retention_frame(a, col, 0.8, 30, 0.0, 2.0), ref), (scale, luma)


def test_oracle_chroma_tweak_and_stabilizer_bodies_match_reference_vectors():
    base = G["base"]
    for i, c in _cases("ctweak_cases"):
        assert np.array_equal(tweaks.np_image_chroma_tweak(base, **c), G[f"ctweak_{i}"]), c
    assert np.array_equal(tweaks.chroma_bright_tweak_frame(base, 0.3, 0.6, 0.8, -0.10, "none"), G["bright_tweak_0"])
    assert np.array_equal(tweaks.chroma_bright_tweak_frame(base, 0.4, 0.4, 0.6, -0.25, "red|0.5,0.0"), G["bright_tweak_1"])
    assert np.array_equal(tweaks.dark_tweak_frame(base, 0.3, 0.8, "none"), G["dark_tweak_0"])
    assert np.array_equal(tweaks.dark_tweak_frame(base, 0.45, 0.5, "280:360,0:30"), G["dark_tweak_1"])


@pytest.mark.gpu
def test_gpu_chroma_tweak_and_stabilizer_frames(ctx):
    from vsdeoldify_amd import imfilters as F, stabilizer
    base, _, _ = _imgs(11)
    for i, c in _cases("ctweak_cases"):
        assert np.array_equal(F.image_chroma_tweak_np(ctx, G["base"], **c), G[f"ctweak_{i}"]), c
        assert np.array_equal(F.image_chroma_tweak_np(ctx, base, **c), tweaks.np_image_chroma_tweak(base, **c)), c
    for c in (dict(sat=3.0, bright=0.8), dict(hue=360), dict(hue=-360, sat=0.0), dict(bright=-1.5), dict(hue_adjust="rose,red|-90,0.5"),
              dict(hue_adjust="bogus|x,y")):
        assert np.array_equal(F.image_chroma_tweak_np(ctx, base, **c), tweaks.np_image_chroma_tweak(base, **c)), c
    assert np.array_equal(stabilizer.chroma_bright_tweak_frame(G["base"], 0.3, 0.6, 0.8, -0.10, "none"), G["bright_tweak_0"])
    assert np.array_equal(stabilizer.chroma_bright_tweak_frame(G["base"], 0.4, 0.4, 0.6, -0.25, "red|0.5,0.0"), G["bright_tweak_1"])
    assert np.array_equal(stabilizer.dark_tweak_frame(G["base"], 0.3, 0.8, "none"), G["dark_tweak_0"])
    assert np.array_equal(stabilizer.dark_tweak_frame(G["base"], 0.45, 0.5, "280:360,0:30"), G["dark_tweak_1"])
    assert np.array_equal(stabilizer.colormap_frame(base, "blue|+40,0.2"), tweaks.colormap_frame(base, "blue|+40,0.2"))


def test_oracle_adjust_hue_range_matches_the_executed_reference():
    """oracle.tweaks.adjust_hue_range vs tests/golden/hue_adjust.npz (restcolor.adjust_hue_range executed by tools/gen_golden_hue_adjust.py)"""
    import os
    from oracle import tweaks
    from tests.conftest import GOLDEN
    g = np.load(os.path.join(GOLDEN, "hue_adjust.npz"))
    for i, c in enumerate(g["cases"].tolist()):
        assert np.array_equal(tweaks.adjust_hue_range(g["img"], c), g[f"out_{i}"]), c


@pytest.mark.gpu
def test_gpu_adjust_hue_range_is_bit_exact(ctx):
    """havc_image_chroma_tweak(has_adjust = 2) = adjust_hue_range, frame and DeviceImage clip"""
    import os
    from tests.conftest import GOLDEN
    from vsdeoldify_amd import imfilters as F
    from vsdeoldify_amd.device import DeviceImage
    g = np.load(os.path.join(GOLDEN, "hue_adjust.npz"))
    for i, c in enumerate(g["cases"].tolist()):
        assert np.array_equal(F.adjust_hue_range_np(ctx, g["img"], c), g[f"out_{i}"]), c
    clip = DeviceImage.from_numpy(ctx, np.stack([g["img"], g["img"][::-1]]))
    out = F.adjust_hue_range_np(ctx, clip, "300:360|0.8,0.1").numpy()
    assert np.array_equal(out[0], g["out_0"]) and np.array_equal(out[1], g["out_0"][::-1])
