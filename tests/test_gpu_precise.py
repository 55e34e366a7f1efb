"""-m gpu: the precise mode (HAVC_F_PRECISE; ModelImageRender(precision="precise") / HAVC_PRECISION=precise) against the fp32 CPU oracle.

The reference computes in fp32 end to end (deoldify/filters.py:45-68, fastai/basic_train.py:352-363) and BASELINE.json's north_star asks for
outputs "within CIEDE2000 < 1.0" of it.  The fast path (fp16 activations) meets that in the mean only (p99 1.2 - 2.3 on the final image
depending on the weights).  The precise path keeps 22 significand bits per activation (hi / lo fp16 pairs), runs every
conv as three K segments on the same MFMA kernels and everything else in fp32.  Stated tolerance, written here:

  * a precise conv / attention / element-wise op vs torch fp32 on the same fp32 inputs: |diff| <= 2e-6 * max|ref| + 1e-6 (fp32-class:
    the plain fp16 path needs 2e-3);
  * raw colour of a generator (u8, before the YUV merge): >= 99.9 % of the bytes EQUAL to the oracle's, no byte off by more than 1
    (uint8(x * 255) truncates: two fp32 evaluations that differ only in summation order already flip 0.01 - 0.03 % of the bytes);
  * final 1080p image of BASELINE configs[1] over 8 frames x 3 seeded weight sets: CIEDE2000 p99 < 1.0 AND >= 99 % of the pixels below 1.0
    (VERDICT r3 item 1), pooled and per frame.
The fast path rides along in the 1080p test on the same frames and reference images, with thresholds that hold on all three weight sets.
"""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import imaging, pipeline
from tests import gpu_util as gu
from tests.test_gpu_deoldify import make_frame, raw_gpu
from vsdeoldify_amd import _native as nat
from vsdeoldify_amd.plan import PlanBuilder, WeightPack
from vsdeoldify_amd.render import GeneratorRuntime, ModelImageRender
from vsdeoldify_amd.synth import synth_state_dict

pytestmark = pytest.mark.gpu

PRECISE_RAW = dict(equal=0.999, max_lsb=1)
PRECISE_CLIP = dict(p99=1.0, frac_lt1=0.99)                 # the contract (BASELINE.json north_star)
# fp16 path, per weight set: measured worst frame on MI355X + 15 % (profiles/r4_pytest_gpu.txt: seeds (1, 2) mean 0.120 / p99 1.251 / below-1.0 0.9675;
# (11, 12) 0.246 / 2.266 / 0.8641; (21, 22) 0.217 / 1.980 / 0.8936; the fraction's slack is 15 % of 1 - fraction).  These prove "no regression", not
# "within the contract": the contract thresholds are PRECISE_CLIP, met by precision="precise" only.
FAST_CLIP_BY_SEED = {(1, 2): dict(mean=0.138, p99=1.44, frac_lt1=0.9626), (11, 12): dict(mean=0.283, p99=2.61, frac_lt1=0.8437),
                     (21, 22): dict(mean=0.250, p99=2.28, frac_lt1=0.8776)}
FAST_CLIP = dict(mean=0.198, p99=2.23, frac_lt1=0.9144)      # pooled over the 8 frames x 3 weight sets: measured 0.1718 / 1.938 / 0.9256, + 15 %


def close32(got, ref, what, rtol=2e-6, atol=1e-6):
    ref = np.asarray(ref, np.float32)
    err = np.abs(got - ref).max()
    lim = rtol * np.abs(ref).max() + atol
    assert np.isfinite(got).all() and err <= lim, f"{what}: max|diff| {err:.4g} > {lim:.4g} (max|ref| {np.abs(ref).max():.4g})"


PCONV = [   # (cfg, B, Cin, Cout, k, stride, pad, H, W)
    (0, 1, 3, 64, 7, 2, 3, 48, 48), (0, 2, 64, 64, 1, 1, 0, 20, 20), (0, 1, 128, 128, 3, 2, 1, 17, 17), (0, 3, 96, 40, 3, 1, 1, 9, 11),
    (0, 1, 259, 259, 3, 1, 1, 32, 32), (61, 2, 264, 259, 3, 1, 1, 21, 13), (60, 1, 320, 256, 3, 1, 1, 24, 24), (70, 2, 64, 128, 1, 1, 0, 19, 21),
    (72, 1, 1024, 256, 1, 1, 0, 35, 35), (96, 1, 512, 256, 3, 1, 1, 18, 18), (99, 1, 64, 64, 3, 1, 1, 30, 30), (1, 1, 256, 256, 3, 1, 1, 12, 12),
    (0, 1, 2048, 512, 1, 1, 0, 5, 5),
]


@pytest.mark.parametrize("case", PCONV, ids=[str(c) for c in PCONV])
def test_precise_conv_matches_torch_fp32(ctx, case):
    """fp32 inputs and weights (NOT pre-rounded to fp16), ReLU + residual epilogue: the three-segment conv is fp32-class"""
    cfg, B, Cin, Cout, k, s, p, H, W = case
    r = np.random.default_rng(hash(case) % 2**31)
    x = r.standard_normal((B, Cin, H, W)).astype(np.float32) * 3
    Wt = (r.standard_normal((Cout, Cin, k, k)) / np.sqrt(Cin * k * k)).astype(np.float32)
    bias = r.standard_normal(Cout).astype(np.float32)
    Ho, Wo = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
    res = r.standard_normal((B, Cout, Ho, Wo)).astype(np.float32)
    got, raw = gu.conv_op_precise(ctx, x, Wt, bias=bias, stride=s, pad=p, flags=nat.F_RELU_PRE, res=res, cfg=cfg)
    ref = F.relu(F.conv2d(torch.from_numpy(x).double(), torch.from_numpy(Wt).double(), torch.from_numpy(bias).double(), s, p)) + torch.from_numpy(res).double()
    close32(got, ref.float().numpy(), f"precise conv {case}")
    P = raw.shape[-1] // 2
    assert (raw[..., Cout:P] == 0).all() and (raw[..., P + Cout:] == 0).all(), "pad channels must stay zero in both planes"


def test_precise_conv_affine_pixshuf_and_large_weights(ctx):
    """ReLU -> affine epilogue, the pixel-shuffle scatter, and a weight pre-scale (|w| >= 31 does not fit 2^11 w_hi in fp16)"""
    r = np.random.default_rng(3)
    x = r.standard_normal((2, 64, 10, 12)).astype(np.float32)
    Wt = (r.standard_normal((128, 64, 1, 1)) / 8).astype(np.float32)
    Wt[5, 7, 0, 0] = 70.0
    sc, sh = r.standard_normal(128).astype(np.float32), r.standard_normal(128).astype(np.float32)
    got, _ = gu.conv_op_precise(ctx, x, Wt, scale=sc, shift=sh, flags=nat.F_RELU_PRE | nat.F_AFFINE)
    ref = F.relu(F.conv2d(torch.from_numpy(x).double(), torch.from_numpy(Wt).double())) * torch.from_numpy(sc).double()[None, :, None, None] + \
        torch.from_numpy(sh).double()[None, :, None, None]
    close32(got, ref.float().numpy(), "precise conv + affine, pre-scaled weights", rtol=4e-6)
    got, _ = gu.conv_op_precise(ctx, x, Wt, flags=nat.F_RELU_PRE, pixshuf=True)
    ref = F.pixel_shuffle(F.relu(F.conv2d(torch.from_numpy(x).double(), torch.from_numpy(Wt).double())), 2)
    close32(got, ref.float().numpy(), "precise conv + pixel shuffle", rtol=4e-6)


@pytest.mark.parametrize("C,hw", [(512, (35, 35)), (256, (9, 13)), (128, (8, 8))])
def test_precise_elementwise_and_attention_ops(ctx, C, hw):
    """max pool, blur (+ nearest resize), BN affine + ReLU and fastai's SelfAttention (fastai/layers.py:81-96) in precise form vs torch fp32.
    N = 1225 / 117 / 64 keys: not multiples of the 64-query / 32-key tiles."""
    H, W = hw
    r = np.random.default_rng(C)
    B, d = 2, C // 8
    x = r.standard_normal((B, C, H, W)).astype(np.float32)
    pack, b = WeightPack(), PlanBuilder(precise=True)
    xv = b.tensor(H, W, C)
    # attention inputs: qk (f | g) and h come from convs in the net; here they are uploaded
    qk, hv, av = b.tensor(H, W, 2 * d), b.tensor(H, W, C), b.tensor(H, W, C)
    f = (r.standard_normal((B, d, H, W)) / 3).astype(np.float32)
    g = (r.standard_normal((B, d, H, W)) / 3).astype(np.float32)
    h = r.standard_normal((B, C, H, W)).astype(np.float32)
    b.attention("attn", xv, qk, d, hv.buf, hv.cpitch, av, 0.37)
    Hm, Wm = (H + 1) // 2, (W + 1) // 2
    mv = b.tensor(Hm, Wm, C)
    b.maxpool("pool", xv, mv)
    bv = b.tensor(H - 1, W - 1, C)                       # blur + nearest resize to a smaller map (the 36 -> 35 case of rf = 35)
    b.blur_resize("blur", xv, bv)
    sc, sh = r.standard_normal(C).astype(np.float32), r.standard_normal(C).astype(np.float32)
    fv = b.tensor(H, W, C)
    b.affine("bn", xv, fv, pack.add(sc), pack.add(sh), relu=True)
    shp = lambda v: ((B, v.H, v.W, v.cpitch), np.float16)
    out = gu.run_plan(ctx, pack, b, {xv.buf: gu.hl_pack(x, xv.cpitch), qk.buf: gu.hl_pack(np.concatenate([f, g], 1), qk.cpitch), hv.buf: gu.hl_pack(h, hv.cpitch)},
                      {v.buf: shp(v) for v in (av, mv, bv, fv)}, B)
    xt = torch.from_numpy(x)
    close32(gu.hl_unpack(out[mv.buf], C), F.max_pool2d(xt, 3, 2, 1).numpy(), "max pool", rtol=0, atol=1e-6)
    blur = F.avg_pool2d(F.pad(xt, (1, 0, 1, 0), mode="replicate"), 2, stride=1)
    close32(gu.hl_unpack(out[bv.buf], C), F.interpolate(blur, (H - 1, W - 1), mode="nearest").numpy(), "blur + nearest resize")
    close32(gu.hl_unpack(out[fv.buf], C), F.relu(xt * torch.from_numpy(sc)[None, :, None, None] + torch.from_numpy(sh)[None, :, None, None]).numpy(), "affine + relu")
    ft, gt, ht = (torch.from_numpy(a).double().reshape(B, a.shape[1], -1) for a in (f, g, h))
    beta = F.softmax(torch.bmm(ft.permute(0, 2, 1), gt), dim=1)
    ref = 0.37 * torch.bmm(ht, beta) + xt.double().reshape(B, C, -1)
    close32(gu.hl_unpack(out[av.buf], C), ref.float().reshape(B, C, H, W).numpy(), "self attention", rtol=4e-6)


@pytest.mark.parametrize("arch,which,seed", [("wide", "video", 1), ("deep", "artistic", 3), ("wide", "stable", 12)])
@pytest.mark.parametrize("S", [64, 80])
def test_precise_generator_raw_color_equals_the_oracle(ctx, arch, which, seed, S):
    sd = synth_state_dict(arch, seed)
    rt = GeneratorRuntime(ctx, sd, arch, precision="precise")
    try:
        frames = np.stack([make_frame(S, 10 + S), make_frame(S, 11 + S)])
        got = raw_gpu(ctx, rt, frames)
        ref = np.stack([pipeline.raw_color_square(sd, arch, f) for f in frames])
        d = np.abs(got.astype(int) - ref.astype(int))
        assert (d == 0).mean() >= PRECISE_RAW["equal"] and d.max() <= PRECISE_RAW["max_lsb"], (float((d == 0).mean()), int(d.max()))
        # batch independence holds in precise mode too
        assert np.array_equal(raw_gpu(ctx, rt, frames[1:]), got[1:])
    finally:
        rt.close()


def test_model_image_render_precision_argument_and_environment(ctx):
    from PIL import Image
    sds = {"video": synth_state_dict("wide", 1), "stable": synth_state_dict("wide", 2)}
    rf, img = 5, make_frame(80, 42)
    ref = pipeline.model_image_render(sds, "stable", img, rf, 0.5)
    outs = {}
    keep = os.environ.get("HAVC_PRECISION")
    for how in ("argument", "environment", "default"):
        if how == "environment":
            os.environ["HAVC_PRECISION"] = "precise"
        elif how == "default":
            os.environ.pop("HAVC_PRECISION", None)             # no argument, no switch: the package default (precision.py) must be the contract-meeting mode
        try:
            r = ModelImageRender(None, "stable", rf, 0.5, state_dicts=sds, precision="precise" if how == "argument" else None)
        finally:
            if keep is None:
                os.environ.pop("HAVC_PRECISION", None)
            else:
                os.environ["HAVC_PRECISION"] = keep
        assert r._video.gen.precise and r._second.gen.precise, how
        outs[how] = np.asarray(r.get_transformed_image(Image.fromarray(img)))
        for rt in (r._video, r._second):
            rt.close()
    assert np.array_equal(outs["argument"], outs["environment"]) and np.array_equal(outs["argument"], outs["default"])
    r = ModelImageRender(None, "stable", rf, 0.5, state_dicts=sds, precision="fast")                 # the opt-in speed mode is still there
    assert not r._video.gen.precise
    de = imaging.delta_e00_images(outs["argument"], ref)
    assert np.percentile(de, 99) < PRECISE_CLIP["p99"] and (de < 1.0).mean() >= PRECISE_CLIP["frac_lt1"], (float(np.percentile(de, 99)), float((de < 1).mean()))
    with pytest.raises(ValueError):
        ModelImageRender(None, "video", rf, 0.5, state_dicts=sds, precision="double")


def test_colorize_clip_1080p_8_frames_3_weight_sets_precise_meets_the_contract(ctx):
    """BASELINE configs[1] (stable, rf = 35, 1080p, two generators + blend + Spline64 + luma): 8 frames x 3 seeded weight sets against
    oracle.pipeline.colorize_frame_fullsize.  precise: CIEDE2000 p99 < 1.0 and >= 99 % of the pixels below 1.0, per frame and pooled;
    fast: the fp16 contract on the same frames, with thresholds that hold for every weight set (not only the bench's)."""
    import gc
    from vsdeoldify_amd.clip import ClipColorizer, synthetic_gray_frame
    seeds, nfr = ((1, 2), (11, 12), (21, 22)), (4, 2, 2)
    pooled = {"fast": [], "precise": []}
    for (sv, ss), n in zip(seeds, nfr):
        sds = {"video": synth_state_dict("wide", sv), "stable": synth_state_dict("wide", ss)}
        idx = [(i * 7) % 32 for i in range(n)]
        frames = np.stack([synthetic_gray_frame(i, 1920, 1080) for i in idx])
        got = {}
        for mode in ("fast", "precise"):
            cc = ClipColorizer("stable", 35, 0.5, device_index=0, state_dicts=sds, max_batch=2, precision=mode)
            got[mode] = np.concatenate([cc.colorize(frames[k:k + 2]) for k in range(0, n, 2)])
            for rt in (cc.render._video, cc.render._second):
                rt.close()
            del cc
            gc.collect()
        for k in range(n):
            ref = gu.oracle_fullsize_stable(sv, ss, idx[k])
            for mode in ("fast", "precise"):
                de = imaging.delta_e00_images(got[mode][k], ref)
                pooled[mode].append(de.reshape(-1))
                p99, frac = float(np.percentile(de, 99)), float((de < 1.0).mean())
                print(f"seeds {sv},{ss} frame {idx[k]} {mode:8s}: mean dE00 {de.mean():.4f} p99 {p99:.3f} max {de.max():.2f} dE<1 {frac:.5f}")
                if mode == "precise":
                    assert p99 < PRECISE_CLIP["p99"] and frac >= PRECISE_CLIP["frac_lt1"], (sv, ss, idx[k], p99, frac)
                else:
                    lim_s = FAST_CLIP_BY_SEED[(sv, ss)]
                    assert de.mean() < lim_s["mean"] and p99 < lim_s["p99"] and frac >= lim_s["frac_lt1"], (sv, ss, idx[k], float(de.mean()), p99, frac)
    for mode, lim in (("precise", PRECISE_CLIP), ("fast", FAST_CLIP)):
        de = np.concatenate(pooled[mode])
        print(f"pooled {mode}: mean {de.mean():.4f} p99 {np.percentile(de, 99):.3f} dE<1 {float((de < 1).mean()):.5f}")
        assert np.percentile(de, 99) < lim["p99"] and (de < 1.0).mean() >= lim["frac_lt1"] and (mode == "precise" or de.mean() < lim["mean"])


@pytest.mark.parametrize("S", [272, 320])
def test_precise_fused_shuffle_blur_is_bit_identical(ctx, S):
    """round 6: HAVC_F_PS_BLUR in precise form (1x1 conv + PixelShuffle + blur on hi / lo pairs in one kernel, two 32-channel passes through the LDS image;
    CustomPixelShuffle_ICNR, vsdeoldify/deoldify/unet.py:24-52) must give exactly the bytes of the conv -> pair tensor -> blur_resize_p chain: the same
    rounding of the shuffled value to its pair, the same fp32 window sum.  Compared on the tail tensor itself (hi and lo planes of the 256 blurred channels,
    every frame) and on the generator's raw colour.  272: a last tile of one useful row + clamping; 320: 160 low-res rows."""
    from vsdeoldify_amd.render import GeneratorRuntime
    sd = synth_state_dict("wide", 1)
    frames = np.stack([make_frame(S, 40), make_frame(S, 41)])
    outs, tails = [], []
    for fuse in (True, False):
        rt = GeneratorRuntime(ctx, sd, "wide", fuse_blur=fuse, precision="precise")
        try:
            net = rt.net(S, 2)
            names = list(net.names)
            assert ("layers.8+blur" in names) == fuse and ("layers.8.blur" in names) != fuse
            outs.append(raw_gpu(ctx, rt, frames))
            ops, _, _, _, _ = rt.gen.plan(S)
            op = ops[names.index("layers.8+blur" if fuse else "layers.8.blur")]
            pitch = int(op["dst_cpitch"])
            t = net.download(int(op["dst"]), (2, S, S, pitch), np.float16)
            tails.append(np.concatenate([t[..., :256], t[..., pitch // 2:pitch // 2 + 256]], -1).copy())     # hi | lo planes of the blurred channels
        finally:
            rt.close()
    assert np.array_equal(tails[0].view(np.uint16), tails[1].view(np.uint16)), float((tails[0] != tails[1]).mean())
    assert np.array_equal(outs[0], outs[1])
