"""Merge-method / temporal filters of vsslib/imfilters.py: oracle vs the executed reference
(CPU) and HIP vs oracle (-m gpu).  Integer / float64 per-pixel paths: BIT-EXACT."""
import os

import numpy as np
import pytest

from oracle import pipeline
from tests.conftest import GOLDEN

LUMAS = (0, 0.3, 0.55)
WLUMAS = ((0.3, 0.9), (0.4, 0.7), (0.0, 0.8), (0.69, 0.7))
ADAPT = ((18, 22, 1.0), (10, 40, 0.7))


@pytest.fixture(scope="module")
def g():
    return np.load(os.path.join(GOLDEN, "imfilters2.npz"))


def test_oracle_matches_reference_filters(g):
    a, b, c = g["a"], g["b"], g["c"]
    for luma in LUMAS:
        assert np.array_equal(pipeline.image_luma_merge(a, b, luma), g[f"image_luma_merge_{luma}"]), luma
    for dl, wl in WLUMAS:
        assert np.array_equal(pipeline.w_image_luma_merge(a, b, dl, wl), g[f"w_image_luma_merge_{dl}_{wl}"]), (dl, wl)
    assert (pipeline.get_image_luma(a), pipeline.get_image_luma(b)) == tuple(g["get_image_luma"])
    for al in (0.05, 0.2):
        assert np.array_equal(pipeline.chroma_temporal_limiter(a, b, al), g[f"chroma_temporal_limiter_{al}"])
    for bt, me, w in ADAPT:
        assert np.array_equal(pipeline.chroma_stabilizer_adaptive(a, b, bt, me, w), g[f"chroma_stabilizer_adaptive_{bt}_{me}_{w}"])
    assert np.array_equal(pipeline.color_temporal_stabilizer([a, b, c], [25, 50, 25]), g["color_temporal_stabilizer_3"])
    assert np.array_equal(pipeline.color_temporal_stabilizer([a, b, c, b, a], [10, 20, 40, 20, 10]), g["color_temporal_stabilizer_5"])


@pytest.mark.gpu
def test_gpu_filters_bit_exact_vs_golden_and_oracle(ctx, g):
    from PIL import Image
    from vsdeoldify_amd import imfilters as F
    a, b, c = g["a"], g["b"], g["c"]
    pa, pb, pc = Image.fromarray(a), Image.fromarray(b), Image.fromarray(c)
    for luma in LUMAS:
        assert np.array_equal(np.asarray(F.image_luma_merge(pa, pb, luma)), g[f"image_luma_merge_{luma}"]), luma
    for dl, wl in WLUMAS:
        assert np.array_equal(np.asarray(F.w_image_luma_merge(pa, pb, dl, wl)), g[f"w_image_luma_merge_{dl}_{wl}"]), (dl, wl)
    assert (F.get_image_luma(pa), F.get_image_luma(pb)) == tuple(g["get_image_luma"])
    for al in (0.05, 0.2):
        assert np.array_equal(np.asarray(F._chroma_temporal_limiter(pa, pb, al)), g[f"chroma_temporal_limiter_{al}"]), al
    for bt, me, w in ADAPT:
        assert np.array_equal(np.asarray(F.chroma_stabilizer_adaptive(pa, pb, bt, me, w)), g[f"chroma_stabilizer_adaptive_{bt}_{me}_{w}"])
    assert np.array_equal(np.asarray(F._color_temporal_stabilizer([pa, pb, pc], [25, 50, 25])), g["color_temporal_stabilizer_3"])
    assert np.array_equal(np.asarray(F._color_temporal_stabilizer([pa, pb, pc, pb, pa], [10, 20, 40, 20, 10])), g["color_temporal_stabilizer_5"])
    # larger seeded frames against the oracle (1080p-class, ragged width)
    r = np.random.default_rng(5)
    x = r.integers(0, 256, (270, 481, 3), dtype=np.uint8)
    y = np.clip(x.astype(int) + r.integers(-60, 61, x.shape), 0, 255).astype(np.uint8)
    assert np.array_equal(F.luma_merge_np(ctx, x, y, 1, 66, round(1 / (204 - 66), 3)), pipeline.w_image_luma_merge(x, y, 0.26, 0.8))
    assert np.array_equal(F.chroma_stabilizer_adaptive_np(ctx, x, y, 18, 22, 0.6), pipeline.chroma_stabilizer_adaptive(x, y, 18, 22, 0.6))
    assert np.array_equal(F.chroma_temporal_limiter_np(ctx, x, y, 0.1), pipeline.chroma_temporal_limiter(x, y, 0.1))
    assert abs(F.image_luma_np(ctx, x) / 255 - pipeline.get_image_luma(x)) < 1e-6


@pytest.mark.gpu
def test_gpu_merge_methods(ctx, g):
    """per-frame bodies of HAVC_merge methods 2/3/4/5/7 (mcomb.py) against the oracle composition of the same filters."""
    from oracle import imaging
    from vsdeoldify_amd import mcomb
    a, b = g["a"], g["b"]
    assert np.array_equal(mcomb.simple_merge(a, b, 0.4), imaging.pil_blend(a, b, 0.4))
    assert np.array_equal(mcomb.constrained_chroma_merge(a, b, 0.5, 0.2, red_fix=False), pipeline.chroma_stabilizer(a, b, 0.2, 0.5))
    assert np.array_equal(mcomb.chroma_bound_adaptive_merge(a, b, False, 14, 18, 0.5), pipeline.chroma_stabilizer_adaptive(a, b, 14, 18, 0.5))
    masked = pipeline.w_image_luma_merge(a, b, 0.4, 0.7)
    assert np.array_equal(mcomb.luma_masked_merge(a, b, None, 0.4, 0.7, 0.5), imaging.pil_blend(a, masked, 0.5))
    luma = pipeline.get_image_luma(b)
    w = max(0.5 * pow(luma / 0.6, 1.0), 0.15) if luma < 0.6 else 0.5
    assert np.array_equal(mcomb.adaptive_luma_merge(a, b, 0.6, 1.0, 0.5, 0.15), imaging.pil_blend(a, b, w))
    # bright frames pass the red-fix gate unchanged; dark ones take the image_tweak branches (mcomb.py:350-361)
    bright = np.clip(a.astype(int) + 120, 0, 255).astype(np.uint8)
    assert np.array_equal(mcomb.constrained_chroma_merge(bright, bright, 0.5, 0.2, True), pipeline.chroma_stabilizer(bright, bright, 0.2, 0.5))
    from oracle import tweaks
    for div in (2, 3, 8):
        assert np.array_equal(mcomb.constrained_chroma_merge(a // div, b // div, 0.5, 0.2, True),
                              tweaks.constrained_chroma_merge(a // div, b // div, 0.2, 0.5)), div
