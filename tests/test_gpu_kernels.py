"""-m gpu: every HIP kernel against the CPU oracle (torch fp32 functional ops / numpy) on seeded inputs.

Tolerances: activations are stored as fp16 and accumulated in fp32 on MFMA, so a conv output may differ
from the fp32 oracle (fed the SAME fp16-rounded inputs and weights) by fp16 output rounding (2^-11 rel)
plus accumulation-order noise: |diff| <= 2e-3 * max|ref| + 2e-3.
"""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from tests import gpu_util as gu
from vsdeoldify_amd import _native as nat
from vsdeoldify_amd.plan import PlanBuilder, View, WeightPack, pack_conv, pad_to

pytestmark = pytest.mark.gpu


def h16(a):
    return np.asarray(a, np.float32).astype(np.float16).astype(np.float32)


def assert_close(got, ref, what, rtol=2e-3, atol=2e-3):
    ref = np.asarray(ref, np.float32)
    err = np.abs(got - ref).max()
    lim = rtol * np.abs(ref).max() + atol
    assert np.isfinite(got).all(), f"{what}: non-finite output"
    assert err <= lim, f"{what}: max|diff| {err:.4g} > {lim:.4g} (max|ref| {np.abs(ref).max():.4g})"


CONV_CASES = [
    # (B, Cin, Cout, k, stride, pad, dil, H, W)
    (1, 3, 64, 7, 2, 3, 1, 48, 48),        # resnet stem (Cin 3 -> 8 padded)
    (2, 64, 64, 1, 1, 0, 1, 20, 20),       # bottleneck 1x1
    (1, 64, 64, 3, 1, 1, 1, 24, 24),       # 3x3
    (1, 128, 128, 3, 2, 1, 1, 17, 17),     # stride-2 3x3, odd size
    (1, 256, 512, 1, 2, 0, 1, 14, 14),     # downsample 1x1 stride 2
    (1, 320, 256, 3, 1, 1, 1, 24, 24),     # decoder cat conv (layers.7 shape family)
    (1, 264 - 5, 259, 3, 1, 1, 1, 32, 32), # tail res conv: 259 -> 259 (BN tile 272)
    (1, 303, 303, 3, 1, 1, 1, 16, 16),     # deep tail: 303 -> 303 (BN tile 304)
    (3, 96, 40, 3, 1, 1, 1, 9, 11),        # ragged everything, batch 3
    (1, 512, 4096, 3, 1, 1, 1, 6, 6),      # middle conv family: tiny M, wide N
    (1, 32, 32, 3, 1, 2, 2, 16, 16),       # dilation 2 (Zhang eccv16 family)
    (1, 2048, 512, 1, 1, 0, 1, 5, 5),      # long K 1x1
]


@pytest.mark.parametrize("case", CONV_CASES, ids=[str(c) for c in CONV_CASES])
def test_conv_plain(ctx, case):
    B, Cin, Cout, k, s, p, d, H, W = case
    r = np.random.default_rng(hash(case) % 2**31)
    x = h16(r.standard_normal((B, Cin, H, W)))
    Wt = h16(r.standard_normal((Cout, Cin, k, k)) / np.sqrt(Cin * k * k))
    bias = r.standard_normal(Cout).astype(np.float32)
    got, raw = gu.conv_op(ctx, x, Wt, bias=bias, stride=s, pad=p, dil=d)
    ref = F.conv2d(torch.from_numpy(x), torch.from_numpy(Wt), torch.from_numpy(bias), s, p, d).numpy()
    assert_close(got, ref, f"conv {case}")
    assert (raw[..., Cout:] == 0).all(), "pad channels must stay zero"


GLDS_CASES = [  # (cfg, B, Cin, Cout, k, stride, pad, H, W): software-pipelined LDS-DMA kernel (conv_igemm_pipe.hip)
    (60, 1, 320, 256, 3, 1, 1, 24, 24), (60, 2, 64, 512, 1, 1, 0, 19, 21), (60, 1, 128, 256, 3, 2, 1, 33, 33),
    (60, 1, 32, 100, 3, 1, 2, 20, 20), (60, 1, 768, 512, 3, 1, 1, 20, 20), (60, 1, 40, 256, 5, 1, 2, 18, 18),
    (60, 1, 3, 64, 7, 2, 3, 48, 48), (60, 2, 400, 300, 3, 1, 1, 17, 15), (60, 1, 2048, 512, 1, 1, 0, 5, 5),
    (61, 1, 259, 259, 3, 1, 1, 32, 32), (61, 2, 264, 259, 3, 1, 1, 21, 13), (61, 3, 259, 259, 3, 1, 1, 9, 11),
    # smaller tile geometries of the same kernel: 128x128 (70), 128x256 (71), 64x128 (72)
    (70, 1, 320, 256, 3, 1, 1, 24, 24), (70, 2, 64, 128, 1, 1, 0, 19, 21), (70, 1, 128, 256, 3, 2, 1, 33, 33), (70, 1, 3, 64, 7, 2, 3, 48, 48),
    (71, 1, 768, 512, 3, 1, 1, 20, 20), (71, 2, 2048, 512, 1, 1, 0, 5, 5), (71, 1, 40, 300, 3, 1, 1, 18, 18),
    (72, 1, 256, 256, 3, 1, 1, 18, 18), (72, 3, 96, 128, 3, 1, 1, 9, 11), (72, 1, 1024, 256, 1, 1, 0, 35, 35),
    (0, 2, 259, 259, 3, 1, 1, 112, 112), (0, 1, 320, 256, 3, 1, 1, 210, 210),      # heuristic picks the pipelined kernel
]


@pytest.mark.parametrize("case", GLDS_CASES, ids=[str(c) for c in GLDS_CASES])
def test_conv_glds_configs(ctx, case):
    cfg, B, Cin, Cout, k, s, p, H, W = case
    r = np.random.default_rng(hash(case) % 2**31)
    x = h16(r.standard_normal((B, Cin, H, W)))
    Wt = h16(r.standard_normal((Cout, Cin, k, k)) / np.sqrt(Cin * k * k))
    bias = r.standard_normal(Cout).astype(np.float32)
    d = 2 if (k == 3 and p == 2) else 1            # the (.., 3, 1, 2, ..) case is a dilation-2 conv
    res = h16(r.standard_normal((B, Cout, (H + 2 * p - d * (k - 1) - 1) // s + 1, (W + 2 * p - d * (k - 1) - 1) // s + 1)))
    got, raw = gu.conv_op(ctx, x, Wt, bias=bias, stride=s, pad=p, dil=d, flags=nat.F_RELU_PRE, res=res, cfg=cfg)
    ref = F.relu(F.conv2d(torch.from_numpy(x), torch.from_numpy(Wt), torch.from_numpy(bias), s, p, d)) + torch.from_numpy(res)
    assert_close(got, ref.numpy(), f"pipelined conv {case}")
    assert (raw[..., Cout:] == 0).all(), "pad channels must stay zero"


@pytest.mark.parametrize("flags,with_res", [(nat.F_RELU_POST, True), (nat.F_RELU_PRE, True), (nat.F_RELU_PRE | nat.F_AFFINE, False)])
def test_conv_tile_configurations_are_bit_identical(ctx, flags, with_res):
    """The autotuner (havc_net_autotune) and the batch-size dependent heuristic may run an op with ANY tile configuration: all of
    them must produce the same bytes (same packed K order, same MFMA sequence per output, same rounding points in the epilogue),
    so a frame colours identically in every batch size."""
    r = np.random.default_rng(11)
    x = h16(r.standard_normal((2, 256, 20, 23)))
    Wt = h16(r.standard_normal((256, 256, 3, 3)) / 48)
    bias, sc, sh = (r.standard_normal(256).astype(np.float32) for _ in range(3))
    res = h16(r.standard_normal((2, 256, 20, 23))) if with_res else None
    kw = dict(bias=bias, pad=1, flags=flags, res=res)
    if flags & nat.F_AFFINE:
        kw.update(scale=sc, shift=sh)
    outs = {cfg: gu.conv_op(ctx, x, Wt, cfg=cfg, **kw)[1] for cfg in (0, 1, 2, 3, 7, 60, 70, 71, 72, 90, 91, 93, 95, 96, 97, 98)}
    base = outs[0]
    for cfg, o in outs.items():
        assert np.array_equal(o.view(np.uint16), base.view(np.uint16)), f"cfg {cfg} differs from the heuristic's choice"


@pytest.mark.parametrize("cout", [192, 224, 96, 320, 304, 64])
def test_conv_tile_configurations_with_a_ragged_last_column_tile(ctx, cout):
    """channel counts that leave the 256- / 128-wide column tiles partly empty (ConvNeXt pwconv2 at stage 0: 768 -> 192): the tuner
    may still pick those tiles (<= 25 % of the tile idle), so they must produce the heuristic's bytes there too."""
    r = np.random.default_rng(cout)
    x = h16(r.standard_normal((2, 320, 18, 21)))
    Wt = h16(r.standard_normal((cout, 320, 1, 1)) / 18)
    bias, sc, sh = (r.standard_normal(cout).astype(np.float32) for _ in range(3))
    res = h16(r.standard_normal((2, cout, 18, 21)))
    kw = dict(bias=bias, pad=0, flags=nat.F_AFFINE | nat.F_RESIDUAL, res=res, scale=sc, shift=sh)
    cfgs = (0, 1, 60, 96, 97, 91, 71) if cout % 256 >= 192 else (0, 1, 70, 72, 93, 95, 98)       # 320, 304: DynamicUnetDeep widths
    if cout % 64 == 0 and cout <= 192:
        cfgs = cfgs + (99, 92)                                                                       # 64-wide pipelined tiles
    outs = {cfg: gu.conv_op(ctx, x, Wt, cfg=cfg, **kw)[1] for cfg in cfgs}
    for cfg, o in outs.items():
        assert np.array_equal(o.view(np.uint16), outs[0].view(np.uint16)), f"cfg {cfg} differs from the heuristic's choice"


def test_autotune_keeps_the_bytes(ctx):
    """a whole generator before / after havc_net_autotune: identical raw colour"""
    from tests.test_gpu_deoldify import make_frame, raw_gpu
    from vsdeoldify_amd.render import GeneratorRuntime
    from vsdeoldify_amd.synth import synth_state_dict
    import os
    frames = np.stack([make_frame(96, 3), make_frame(96, 4)])
    outs = []
    for tune in ("0", "1"):
        os.environ["HAVC_AUTOTUNE"] = tune
        try:
            rt = GeneratorRuntime(ctx, synth_state_dict("wide", 1), "wide")
            try:
                net = rt.net(96, 2)
                outs.append((raw_gpu(ctx, rt, frames), net.cfgs()))
            finally:
                rt.close()
        finally:
            os.environ.pop("HAVC_AUTOTUNE", None)
    assert np.array_equal(outs[0][0], outs[1][0])
    assert all(c == 0 for c in outs[0][1]) and any(c != 0 for c in outs[1][1])


def test_conv_epilogue_relu_affine_residual(ctx):
    r = np.random.default_rng(1)
    x = h16(r.standard_normal((2, 72, 13, 15)))
    Wt = h16(r.standard_normal((136, 72, 3, 3)) / 25)
    bias, sc, sh = (r.standard_normal(136).astype(np.float32) for _ in range(3))
    res = h16(r.standard_normal((2, 136, 13, 15)))
    xt, wt = torch.from_numpy(x), torch.from_numpy(Wt)
    conv = F.conv2d(xt, wt, torch.from_numpy(bias), 1, 1)
    # conv -> ReLU -> BN (decoder, deoldify/layers.py:39-45)
    got, _ = gu.conv_op(ctx, x, Wt, bias=bias, scale=sc, shift=sh, pad=1, flags=nat.F_RELU_PRE | nat.F_AFFINE)
    ref = F.relu(conv) * torch.from_numpy(sc).view(1, -1, 1, 1) + torch.from_numpy(sh).view(1, -1, 1, 1)
    assert_close(got, ref.numpy(), "relu->affine")
    # bottleneck tail: relu(conv + identity)
    got, _ = gu.conv_op(ctx, x, Wt, bias=bias, pad=1, flags=nat.F_RELU_POST, res=res)
    assert_close(got, F.relu(conv + torch.from_numpy(res)).numpy(), "residual->relu")
    # res_block tail: c + relu(conv + b)   (fastai/layers.py:154-161)
    got, _ = gu.conv_op(ctx, x, Wt, bias=bias, pad=1, flags=nat.F_RELU_PRE, res=res)
    assert_close(got, (F.relu(conv) + torch.from_numpy(res)).numpy(), "relu->residual")


@pytest.mark.parametrize("cout4", [64, 1200])
def test_conv_pixel_shuffle(ctx, cout4):
    r = np.random.default_rng(2)
    x = h16(r.standard_normal((2, 48, 7, 9)))
    Wt = h16(r.standard_normal((cout4, 48, 1, 1)) / 7)
    bias = r.standard_normal(cout4).astype(np.float32)
    got, _ = gu.conv_op(ctx, x, Wt, bias=bias, flags=nat.F_RELU_PRE, pixshuf=True)
    ref = F.pixel_shuffle(F.relu(F.conv2d(torch.from_numpy(x), torch.from_numpy(Wt), torch.from_numpy(bias))), 2)
    assert_close(got, ref.numpy(), "pixel shuffle store")


def test_maxpool_blur_affine(ctx):
    r = np.random.default_rng(3)
    x = h16(r.standard_normal((2, 64, 18, 18)))
    pack, b = WeightPack(), PlanBuilder()
    xv = b.tensor(18, 18, 64)
    mp = b.tensor(9, 9, 64)
    b.maxpool("mp", xv, mp)
    bl = b.tensor(18, 18, 64)
    b.blur_resize("blur", xv, bl)
    bl2 = b.tensor(17, 17, 64)          # nearest 18 -> 17 after the blur (the rf=35 36->35 case)
    b.blur_resize("blur_rs", xv, bl2)
    sc, sh = r.standard_normal(64).astype(np.float32), r.standard_normal(64).astype(np.float32)
    af = b.tensor(18, 18, 64)
    b.affine("aff", xv, af, pack.add(sc), pack.add(sh), relu=True)
    out = gu.run_plan(ctx, pack, b, {xv.buf: gu.nhwc_pad(x)},
                      {v.buf: ((2, v.H, v.W, 64), np.float16) for v in (mp, bl, bl2, af)}, 2)
    xt = torch.from_numpy(x)
    nchw = lambda v: out[v.buf].astype(np.float32).transpose(0, 3, 1, 2)
    assert_close(nchw(mp), F.max_pool2d(xt, 3, 2, 1).numpy(), "maxpool", 0, 0)
    blur = F.avg_pool2d(F.pad(xt, (1, 0, 1, 0), mode="replicate"), 2, stride=1)
    assert_close(nchw(bl), blur.numpy(), "blur", 1e-3, 1e-3)
    assert_close(nchw(bl2), F.interpolate(blur, (17, 17), mode="nearest").numpy(), "blur+nearest", 1e-3, 1e-3)
    ref = F.relu(xt * torch.from_numpy(sc).view(1, -1, 1, 1) + torch.from_numpy(sh).view(1, -1, 1, 1))
    assert_close(nchw(af), ref.numpy(), "affine+relu", 1e-3, 1e-3)


@pytest.mark.parametrize("C,H,W,B", [(512, 6, 6, 1), (512, 13, 11, 2), (768, 9, 9, 1), (512, 24, 23, 2), (768, 17, 19, 1), (256, 10, 10, 1)])
def test_self_attention(ctx, C, H, W, B):
    """fastai SelfAttention (fastai/layers.py:81-96) via two 1x1 convs + the flash kernel."""
    from oracle import unet as ou
    r = np.random.default_rng(C + H)
    d = C // 8
    x = h16(r.standard_normal((B, C, H, W)))
    wq, wk = (h16(r.standard_normal((d, C, 1)) / np.sqrt(C) * 1.5) for _ in range(2))
    wv = h16(r.standard_normal((C, C, 1)) / np.sqrt(C))
    gamma = 0.7
    pack, b = WeightPack(), PlanBuilder()
    xv = b.tensor(H, W, C)
    pc_qk = pack_conv(pack, np.concatenate([wq, wk])[..., None], xv.cmap, xv.span)
    pc_v = pack_conv(pack, wv[..., None], xv.cmap, xv.span)
    qk = b.tensor(H, W, 2 * d)
    b.conv("qk", pc_qk, xv, qk)
    N = H * W
    npitch = pad_to(N, 64)
    vT = b.buf(C * npitch, 2, zero_init=True)
    b.conv("v", pc_v, xv, vT, flags=nat.F_OUT_TRANSPOSED, Co=C, aux0=npitch)
    y = b.tensor(H, W, C)
    b.attention("attn", xv, qk, d, vT, npitch, y, gamma)
    out = gu.run_plan(ctx, pack, b, {xv.buf: gu.nhwc_pad(x, xv.cpitch)}, {y.buf: ((B, H, W, y.cpitch), np.float16)}, B)[y.buf]
    got = out[..., :C].astype(np.float32).transpose(0, 3, 1, 2)
    # oracle: the restated reference layer on a state dict whose spectral fold is the identity (sigma = 1)
    sd = {}
    for nm, w in (("query", wq), ("key", wk), ("value", wv)):
        wm = torch.from_numpy(w)
        v = torch.ones(C) / np.sqrt(C)
        t = wm.reshape(w.shape[0], -1) @ v
        sd[f"a.{nm}.weight_orig"], sd[f"a.{nm}.weight_v"], sd[f"a.{nm}.weight_u"] = wm, v, t / t.dot(t)
    sd["a.gamma"] = torch.tensor([gamma])
    ref = ou.self_attention(sd, "a", torch.from_numpy(x)).numpy()
    assert_close(got, ref, f"attention C={C} N={N}", 4e-3, 4e-3)


def test_prep_rgb8(ctx):
    from oracle import imaging
    r = np.random.default_rng(5)
    img = r.integers(0, 256, (2, 32, 32, 3), dtype=np.uint8)
    b = PlanBuilder()
    inb = b.buf(32 * 32 * 3, 1)
    y0 = b.tensor(32, 32, 3, zero_init=False)
    b.prep_rgb8("prep", inb, 32, y0)
    out = gu.run_plan(ctx, WeightPack(), b, {inb: img}, {y0.buf: ((2, 32, 32, 8), np.float16)}, 2)[y0.buf]
    ref = np.concatenate([imaging.model_input(im) for im in img]).transpose(0, 2, 3, 1)
    assert (out[..., 3:] == 0).all()
    assert np.abs(out[..., :3].astype(np.float32) - ref).max() <= 2e-3   # fp16 rounding of values <= 2.7


def test_color_filters_bit_exact(ctx):
    """blend / chroma_post_process / chroma_stabilizer are integer paths: bit-exact vs the oracle."""
    from oracle import imaging, pipeline
    from vsdeoldify_amd import imfilters
    r = np.random.default_rng(6)
    a = r.integers(0, 256, (61, 83, 3), dtype=np.uint8)
    bimg = r.integers(0, 256, (61, 83, 3), dtype=np.uint8)
    for w in (0.3, 0.4, 0.5, 0.77):
        assert np.array_equal(imfilters.blend_np(ctx, a, bimg, w), imaging.pil_blend(a, bimg, w)), f"blend w={w}"
    assert np.array_equal(imfilters.chroma_post_process_np(ctx, a, bimg), pipeline.chroma_post_process(a, bimg))
    for alpha, wgt in ((0.15, 1.0), (0.2, 0.6), (0.05, 0.5)):
        got = imfilters.chroma_stabilizer_np(ctx, a, bimg, alpha, wgt)
        assert np.array_equal(got, pipeline.chroma_stabilizer(a, bimg, alpha, wgt)), f"chroma_stabilizer {alpha} {wgt}"
    # exhaustive-ish: all (Y,U,V)-relevant corners
    g = np.stack(np.meshgrid(np.arange(0, 256, 5), np.arange(0, 256, 5), np.arange(0, 256, 5)), -1).reshape(-1, 1, 3).astype(np.uint8)
    h = g[::-1].copy()
    assert np.array_equal(imfilters.chroma_post_process_np(ctx, g, h), pipeline.chroma_post_process(g, h))


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(259, 259, 48, 80, 2, True), (64, 256, 33, 47, 3, False), (768, 512, 32, 32, 1, False)])
def test_halo_conv_kernel_identical_to_pipe_kernel(ctx, shape):
    """conv_halo_kernel (cfg 80 / 81: 16x16 tiles, halo staged once per 64-channel group; experimental, not selected by default)
    runs the same stage order and MFMA sequence as conv_pipe_kernel: outputs must be bit-identical, incl. the 259-channel
    remainder segment, partial edge tiles and the residual epilogue."""
    import importlib.util, os
    spec = importlib.util.spec_from_file_location("halo_check", os.path.join(os.path.dirname(__file__), "..", "tools", "halo_check.py"))
    hc = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(hc)
    ref, got = hc.check(shape)
    assert np.isfinite(got).all() and np.array_equal(ref, got)


def test_tuning_cache_round_trip(ctx, tmp_path, monkeypatch):
    """Net.autotune remembers its result across processes (a file per plan / batch / device / library build under HAVC_TUNE_CACHE):
    the second net of the same plan restores the configurations without measuring; a damaged file is ignored; "0" disables."""
    import os
    from vsdeoldify_amd.render import GeneratorRuntime
    from vsdeoldify_amd.synth import synth_state_dict
    monkeypatch.setenv("HAVC_AUTOTUNE", "1")
    monkeypatch.setenv("HAVC_TUNE_CACHE", str(tmp_path))
    sd = synth_state_dict("wide", 1)

    def tuned_cfgs():
        rt = GeneratorRuntime(ctx, sd, "wide")
        try:
            return rt.net(96, 2).cfgs()
        finally:
            rt.close()
    first = tuned_cfgs()
    files = list(tmp_path.iterdir())
    assert len(files) == 1 and any(c != 0 for c in first)
    assert tuned_cfgs() == first                                                   # restored
    files[0].write_bytes(b"\x00" * 7)                                              # damaged: measured again, rewritten
    again = tuned_cfgs()
    assert len(again) == len(first) and files[0].stat().st_size == 4 * len(first)
    monkeypatch.setenv("HAVC_TUNE_CACHE", "0")
    files[0].unlink()
    tuned_cfgs()
    assert not list(tmp_path.iterdir())


def test_conv_fuzz_subset(ctx):
    """a seeded slice of tools/conv_fuzz.py (1 800 random shapes / strides / dilations / epilogues / tile configurations ran clean there):
    values against torch conv2d, and bytes identical across the tile configurations that accept the shape"""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "conv_fuzz.py"), "60", "3"], capture_output=True, text=True, cwd=root, timeout=600)
    assert out.returncode == 0 and out.stdout.strip().endswith("60 cases, 0 problems"), out.stdout[-2000:] + out.stderr[-2000:]


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(392, 256, 256, 3), (513, 1536, 384, 1), (784, 1600, 512, 3)])
def test_conv_split_k_matches_the_unsplit_conv_and_is_tile_independent(ctx, shape):
    """HAVC_F_SPLITK: a small-M conv (one frame of a ColorMNet layer) with its K range cut into parts == the same conv unsplit up to the
    fp32 summation order, and the SAME BYTES whatever tile configuration runs it (the count belongs to the plan, the tile to the tuner)."""
    from tests.gpu_util import conv_op
    M, Cin, Cout, k = shape
    H, W = (14, M // 14) if M % 14 == 0 else (1, M)
    r = np.random.default_rng(M)
    x = (r.standard_normal((1, Cin, H, W)) * 0.5).astype(np.float32)
    w = (r.standard_normal((Cout, Cin, k, k)) / np.sqrt(Cin * k * k)).astype(np.float32)
    bias = r.standard_normal(Cout).astype(np.float32) * 0.1
    res = (r.standard_normal((1, Cout, H, W)) * 0.5).astype(np.float32)
    base, _ = conv_op(ctx, x, w, bias=bias, pad=k // 2, flags=nat.F_RELU_PRE, res=res)
    outs = []
    for cfg in (0, 70, 72, 71):
        got, raw = conv_op(ctx, x, w, bias=bias, pad=k // 2, flags=nat.F_RELU_PRE | nat.F_SPLITK(4), res=res, cfg=cfg)
        outs.append(raw)
        assert np.abs(got - base).max() < 4e-3 * max(1.0, float(np.abs(base).max())), (cfg, float(np.abs(got - base).max()))
    assert all(np.array_equal(outs[0], o) for o in outs[1:])
    t = torch_conv(x, w, bias, k // 2)
    assert np.abs(base - (np.maximum(t, 0).astype(np.float16).astype(np.float32) + res.astype(np.float16).astype(np.float32))).max() < 2e-2


def torch_conv(x, w, bias, pad):
    import torch
    import torch.nn.functional as F
    return F.conv2d(torch.from_numpy(x.astype(np.float16).astype(np.float32)), torch.from_numpy(w.astype(np.float16).astype(np.float32)),
                    torch.from_numpy(bias), padding=pad).numpy()


@pytest.mark.gpu
@pytest.mark.parametrize("cfg", [0, 70, 1])
def test_conv_views_that_do_not_fill_their_frame_run_frame_by_frame(ctx, cfg):
    """A ViT token buffer holds h x w patch rows + one class row per frame; the patch-embedding conv writes an h x w VIEW of it and a later conv
    reads such a view (colormnet_net.py _dino).  With batch > 1 the rows of a view do not continue into the next frame: the runtime launches per
    frame at the buffer's own frame stride.  (As one launch over contiguous rows, frame 1's first patch row landed on frame 0's class row.)"""
    r = np.random.default_rng(8)
    B, h, w, Cin, Cout, P = 3, 5, 7, 24, 64, 4
    T = h * w + 1
    pack, b = WeightPack(), PlanBuilder()
    x = b.tensor(h * P, w * P, Cin)
    from vsdeoldify_amd.plan import pitch_for
    pc_ = pitch_for(pad_to(Cout, 8))
    tok = View(b.buf(T * pc_, 2), 0, pc_, 1, T, Cout, pad_to(Cout, 8))
    W1 = (r.standard_normal((Cout, Cin, P, P)) / np.sqrt(Cin * P * P)).astype(np.float32)
    c1 = pack_conv(pack, W1, x.cmap, x.span, bias=r.standard_normal(Cout).astype(np.float32))
    b.conv("patch_embed", c1, x, View(tok.buf, 0, tok.cpitch, h, w, Cout, tok.span), stride=P)
    patches = View(tok.buf, 0, tok.cpitch, h, w, Cout, tok.span)
    y = b.tensor(h, w, 32)
    W2 = (r.standard_normal((32, Cout, 1, 1)) / np.sqrt(Cout)).astype(np.float32)
    c2 = pack_conv(pack, W2, patches.cmap, patches.span)
    b.conv("head", c2, patches, y)
    xin = (r.standard_normal((B, Cin, h * P, w * P)) * 0.5).astype(np.float32)
    tok0 = np.zeros((B, T, tok.cpitch), np.float16)
    tok0[:, T - 1, :Cout] = (np.arange(Cout) / 8.0 + 1.0).astype(np.float16)          # the class row: must survive
    out = gu.run_plan(ctx, pack, b, {x.buf: gu.nhwc_pad(xin, x.cpitch), tok.buf: tok0},
                      {tok.buf: ((B, T, tok.cpitch), np.float16), y.buf: ((B, h, w, y.cpitch), np.float16)}, B, cfg)
    f16 = lambda a: a.astype(np.float16).astype(np.float32)
    xt, w1 = torch.from_numpy(f16(xin)), torch.from_numpy(f16(W1))
    ref = F.conv2d(xt, w1, None, stride=P).numpy()
    got = out[tok.buf].astype(np.float32)
    bias = got[:, :h * w, :Cout].reshape(B, h, w, Cout).transpose(0, 3, 1, 2) - ref                  # what was added must be ONE bias vector
    assert np.abs(bias - bias[0, :, 0, 0][None, :, None, None]).max() < 2e-2
    assert np.array_equal(got[:, T - 1, :Cout], tok0[:, T - 1, :Cout].astype(np.float32))          # class rows untouched in every frame
    head = F.conv2d(torch.from_numpy(got[:, :h * w, :Cout].reshape(B, h, w, Cout).transpose(0, 3, 1, 2).copy()), torch.from_numpy(f16(W2))).numpy()
    gy = out[y.buf][..., :32].astype(np.float32).transpose(0, 3, 1, 2)
    assert np.abs(gy - head).max() < 2e-2 * max(1.0, float(np.abs(head).max()))


def test_grouped_tile_order_gives_the_same_bytes(ctx):
    """round 6: HAVC_RASTER_GROUP=1 (kernels.h conv_raster: column tiles walked in L2-sized groups inside bands of row tiles, for GEMMs whose weight matrix
    exceeds an XCD's L2 -- measured 2 % slower and left off) changes the block -> tile map only.  ConvNeXt-L's stage-2 pwconv1 shape (768 -> 3072 + GELU) with
    17 row tiles (ragged last band) in a child process with the switch on, against this process (off): identical bytes."""
    import hashlib
    import subprocess
    import sys
    code = (
        "import sys, hashlib, numpy as np; sys.path.insert(0, %r)\n"
        "from tests import gpu_util as gu\n"
        "from vsdeoldify_amd import _native as nat\n"
        "from vsdeoldify_amd.render import get_context\n"
        "r = np.random.default_rng(7)\n"
        "x = (r.standard_normal((1, 768, 68, 64)) * 0.5).astype(np.float16).astype(np.float32)\n"
        "W = (r.standard_normal((3072, 768, 1, 1)) / 28).astype(np.float16).astype(np.float32)\n"
        "b = r.standard_normal(3072).astype(np.float32)\n"
        "out = gu.conv_op(get_context(0), x, W, bias=b, flags=nat.F_GELU, cfg=60)[1]\n"
        "print('SHA', hashlib.sha1(out.tobytes()).hexdigest())\n" % os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    shas = {}
    for g in ("1", "0"):
        p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=dict(os.environ, HAVC_RASTER_GROUP=g, HAVC_TUNE_CACHE="0"))
        lines = [ln for ln in p.stdout.splitlines() if ln.startswith("SHA ")]
        assert p.returncode == 0 and lines, (p.stdout[-300:], p.stderr[-600:])
        shas[g] = lines[-1]
    assert shas["1"] == shas["0"], shas
