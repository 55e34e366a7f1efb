"""-m gpu: precise mode (HAVC_F_PRECISE) beyond the DeOldify generators (round 5): DDColor, the Zhang colorizers and the HAVC
graphs that contain them.  The reference runs every model in fp32 (vsdeoldify/colorization/__init__.py:76-95, vsdeoldify/vsslib/vsmodels.py:353-363,
vsdeoldify/deoldify/filters.py:45-68); precision="precise" keeps fp32-class arithmetic on hi / lo fp16 pairs (csrc/precise.hip, csrc/precise2.hip).

Tolerances (written where they are used): a precise op vs torch on fp32 inputs: |diff| <= 4e-6 max|ref| + 1e-6 (LayerNorm / attention: 2e-5, the
reductions run in another order than torch's); frames at the config sizes vs the all-oracle graph: the contract of BASELINE.json's north_star,
CIEDE2000 p99 < 1.0 and >= 99 % of the pixels below 1.0.  DDColor itself is parity-UNPINNED (oracle/ddcolor.py restates the published architecture)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import imaging, pipeline, resample
from tests import gpu_util as gu
from vsdeoldify_amd import _native as nat
from vsdeoldify_amd.clip import synthetic_gray_frame
from vsdeoldify_amd.plan import PlanBuilder, View, WeightPack, pack_conv
from vsdeoldify_amd.synth import synth_ddcolor_state_dict, synth_state_dict, synth_zhang_state_dict

pytestmark = pytest.mark.gpu
CONTRACT = dict(p99=1.0, frac_lt1=0.99)


def close32(got, ref, what, rtol=4e-6, atol=1e-6):
    ref = np.asarray(ref, np.float32)
    err = np.abs(got - ref).max()
    lim = rtol * np.abs(ref).max() + atol
    assert np.isfinite(got).all() and err <= lim, f"{what}: max|diff| {err:.4g} > {lim:.4g} (max|ref| {np.abs(ref).max():.4g})"


def _shape(v, B):
    return ((B, v.H, v.W, v.cpitch), np.float16)


@pytest.mark.parametrize("C,hw", [(192, (9, 7)), (768, (5, 6)), (1536, (3, 4)), (256, (1, 112))])
def test_precise_layernorm_dwconv_gelu(ctx, C, hw):
    """channel LayerNorm (with and without the fused ReLU), depthwise 7x7 + bias, and a 1x1 conv with the exact-GELU epilogue vs torch fp32"""
    H, W = hw
    B = 2
    r = np.random.default_rng(C + H)
    x = (r.standard_normal((B, C, H, W)) * 2 + 0.3).astype(np.float32)
    g, be = (1 + 0.2 * r.standard_normal(C)).astype(np.float32), (0.1 * r.standard_normal(C)).astype(np.float32)
    wd, bd = (r.standard_normal((C, 1, 7, 7)) / 7).astype(np.float32), (0.1 * r.standard_normal(C)).astype(np.float32)
    pack, b = WeightPack(), PlanBuilder(precise=True)
    xv = b.tensor(H, W, C)
    y1, y2, y3 = b.tensor(H, W, C), b.tensor(H, W, C), b.tensor(H, W, C)
    go, bo = pack.add(g), pack.add(be)
    b.layernorm("ln", xv, y1, go, bo, 1e-6)
    b.layernorm("ln+relu", xv, y2, go, bo, 1e-6, relu=True)
    wp = np.zeros((49, xv.span), np.float32)
    wp[:, :C] = wd.reshape(C, 49).T
    b.dwconv7("dw", xv, y3, pack.add(wp), pack.add(bd), xv.span)
    Co = 64
    Wt = (r.standard_normal((Co, C, 1, 1)) / np.sqrt(C)).astype(np.float32)
    bias = (0.2 * r.standard_normal(Co)).astype(np.float32)
    y4 = b.tensor(H, W, Co)
    b.conv("pw+gelu", pack_conv(pack, Wt, xv.cmap, xv.span, bias=bias, precise=True), xv, y4, flags=nat.F_GELU)
    out = gu.run_plan(ctx, pack, b, {xv.buf: gu.hl_pack(x, xv.cpitch)}, {v.buf: _shape(v, B) for v in (y1, y2, y3, y4)}, B)
    xt = torch.from_numpy(x).double()
    ln = F.layer_norm(xt.permute(0, 2, 3, 1), (C,), torch.from_numpy(g).double(), torch.from_numpy(be).double(), 1e-6).permute(0, 3, 1, 2)
    close32(gu.hl_unpack(out[y1.buf], C), ln.float().numpy(), "layernorm", rtol=2e-5)
    close32(gu.hl_unpack(out[y2.buf], C), F.relu(ln).float().numpy(), "layernorm + relu", rtol=2e-5)
    dw = F.conv2d(xt, torch.from_numpy(wd).double(), torch.from_numpy(bd).double(), padding=3, groups=C)
    close32(gu.hl_unpack(out[y3.buf], C), dw.float().numpy(), "depthwise 7x7")
    pw = F.gelu(F.conv2d(xt, torch.from_numpy(Wt).double(), torch.from_numpy(bias).double()))
    close32(gu.hl_unpack(out[y4.buf], Co), pw.float().numpy(), "1x1 conv + GELU")


@pytest.mark.parametrize("C,hw,B", [(192, (9, 7), 2), (384, (13, 18), 1), (768, (5, 6), 3), (192, (32, 32), 2)])
def test_precise_fused_dwconv7_layernorm(ctx, C, hw, B):
    """round 6: the ConvNeXt block head (depthwise 7x7 + bias, then LayerNorm over the channels; convnext.py Block: dwconv, norm -- the wheel runs it in fp32,
    vsslib/vsmodels.py:353-363) as ONE kernel on pair tensors (HAVC_OP_DWCONV7_LN with HAVC_F_PRECISE: dwconv7_ln_kernel<..., PREC>) against torch in float64,
    at the tolerance of the two-kernel precise chain (2e-5); ragged patch grids (sizes that are not multiples of the 4 x 4 patch), borders on every side."""
    H, W = hw
    r = np.random.default_rng(C + H + B)
    x = (r.standard_normal((B, C, H, W)) * 2 + 0.3).astype(np.float32)
    g, be = (1 + 0.2 * r.standard_normal(C)).astype(np.float32), (0.1 * r.standard_normal(C)).astype(np.float32)
    wd, bd = (r.standard_normal((C, 1, 7, 7)) / 7).astype(np.float32), (0.1 * r.standard_normal(C)).astype(np.float32)
    pack, b = WeightPack(), PlanBuilder(precise=True)
    xv, y = b.tensor(H, W, C), b.tensor(H, W, C)
    wp = np.zeros((49, xv.span), np.float32)
    wp[:, :C] = wd.reshape(C, 49).T
    b.dwconv7_ln("dwln", xv, y, pack.add(wp), pack.add(bd), xv.span, pack.add(g), pack.add(be), 1e-6)
    out = gu.run_plan(ctx, pack, b, {xv.buf: gu.hl_pack(x, xv.cpitch)}, {y.buf: _shape(y, B)}, B)
    xt = torch.from_numpy(x).double()
    dw = F.conv2d(xt, torch.from_numpy(wd).double(), torch.from_numpy(bd).double(), padding=3, groups=C)
    ref = F.layer_norm(dw.permute(0, 2, 3, 1), (C,), torch.from_numpy(g).double(), torch.from_numpy(be).double(), 1e-6).permute(0, 3, 1, 2)
    close32(gu.hl_unpack(out[y.buf], C), ref.float().numpy(), "fused depthwise 7x7 + LayerNorm (precise)", rtol=2e-5)


@pytest.mark.parametrize("Lk_hw", [(10, 10), (24, 24), (5, 13)])
def test_precise_multihead_attention(ctx, Lk_hw):
    """nn.MultiheadAttention core (8 heads x 32) on pair buffers: 100 queries of a 112-token frame, keys = pixels of a feature map (K at +0, V at +256
    of a 512-channel map), and the queries' self attention (q | k | v slices of one token buffer) -- vs a float64 softmax(Q K^T / sqrt(32)) V"""
    B, E, heads, TOK, Q = 2, 256, 8, 112, 100
    kh, kw = Lk_hw
    r = np.random.default_rng(kh * kw)
    q = r.standard_normal((B, E, 1, TOK)).astype(np.float32)
    kv = r.standard_normal((B, 2 * E, kh, kw)).astype(np.float32)
    qkv = r.standard_normal((B, 3 * E, 1, TOK)).astype(np.float32)
    pack, b = WeightPack(), PlanBuilder(precise=True)
    qv, kvv, o1 = b.tensor(1, TOK, E), b.tensor(kh, kw, 2 * E), b.tensor(1, TOK, E)
    qkvv, o2 = b.tensor(1, TOK, 3 * E), b.tensor(1, TOK, E)
    scale = 1.0 / np.sqrt(32.0)
    b.mha("cross", qv, kvv, 0, E, o1, heads, Q, kh * kw, scale)
    b.mha("self", View(qkvv.buf, 0, qkvv.cpitch, 1, TOK, E, E), qkvv, E, 2 * E, o2, heads, Q, Q, scale)
    out = gu.run_plan(ctx, pack, b, {qv.buf: gu.hl_pack(q, qv.cpitch), kvv.buf: gu.hl_pack(kv, kvv.cpitch), qkvv.buf: gu.hl_pack(qkv, qkvv.cpitch)},
                      {o1.buf: _shape(o1, B), o2.buf: _shape(o2, B)}, B)

    def ref(qm, km, vm):                                  # [B, L, E] each
        qh = torch.from_numpy(qm).double().reshape(B, -1, heads, 32).transpose(1, 2)
        kh_ = torch.from_numpy(km).double().reshape(B, -1, heads, 32).transpose(1, 2)
        vh = torch.from_numpy(vm).double().reshape(B, -1, heads, 32).transpose(1, 2)
        a = torch.softmax(qh @ kh_.transpose(-1, -2) * scale, -1) @ vh
        return a.transpose(1, 2).reshape(B, -1, E).float().numpy()
    qm = q[:, :, 0, :Q].transpose(0, 2, 1)
    kvm = kv.reshape(B, 2 * E, -1).transpose(0, 2, 1)
    got1 = gu.hl_unpack(out[o1.buf], E)[:, :, 0, :Q].transpose(0, 2, 1)
    close32(got1, ref(qm, kvm[..., :E], kvm[..., E:]), "cross attention", rtol=2e-5)
    t = qkv[:, :, 0, :Q].transpose(0, 2, 1)
    got2 = gu.hl_unpack(out[o2.buf], E)[:, :, 0, :Q].transpose(0, 2, 1)
    close32(got2, ref(t[..., :E], t[..., E:2 * E], t[..., 2 * E:]), "self attention", rtol=2e-5)


def test_precise_zhang_ops_and_ddcolor_tail(ctx):
    """proj2 (313-way softmax + 1x1; tanh head), `[::2, ::2]`, and the DDColor tail: fold_queries + PixelShuffle(4) + blur + projection + image term"""
    B, H, W = 2, 6, 10
    r = np.random.default_rng(11)
    pack, b = WeightPack(), PlanBuilder(precise=True)
    x313 = (r.standard_normal((B, 313, H, W)) * 3).astype(np.float32)
    x128 = r.standard_normal((B, 128, H, W)).astype(np.float32)
    v313, v128 = b.tensor(H, W, 313), b.tensor(H, W, 128)
    w1, w2, b2 = (r.standard_normal((2, 313)) / 4).astype(np.float32), (r.standard_normal((2, 128)) / 8).astype(np.float32), r.standard_normal(2).astype(np.float32)
    o1, o2 = b.buf(H * W * 2, 4), b.buf(H * W * 2, 4)
    b.proj2("softmax+1x1", v313, pack.add(w1), -1, 1, 1.0, o1)
    b.proj2("1x1+tanh", v128, pack.add(w2), pack.add(b2), 2, 110.0, o2)
    sub = b.tensor((H + 1) // 2, (W + 1) // 2, 128)
    b.subsample2("sub", v128, sub)
    # DDColor tail: t4 [H][W][16 * 256] -> ab [4H][4W][2]
    E, Qn, TOK = 256, 100, 112
    t4 = (np.maximum(r.standard_normal((B, 16 * E, H, W)), 0) * 0.7).astype(np.float32)
    emb = (r.standard_normal((B, E, 1, TOK)) * 0.5).astype(np.float32)
    img = r.standard_normal((B, 3, 4 * H, 4 * W)).astype(np.float32)
    Wr = (r.standard_normal((2, Qn + 3)) / 10).astype(np.float32)
    rb = r.standard_normal(2).astype(np.float32)
    t4v, embv, imgv, abv = b.tensor(H, W, 16 * E), b.tensor(1, TOK, E), b.tensor(4 * H, 4 * W, 3), b.tensor(4 * H, 4 * W, 2)
    rq = np.zeros((2, 104), np.float32)
    rq[:, :Qn] = Wr[:, :Qn]
    m2 = b.buf(2 * E, 4)
    b.fold_queries("fold", embv, Qn, pack.add(rq), 104, m2)
    b.shuf4_blur_proj("tail", t4v, m2, imgv, pack.add(np.ascontiguousarray(Wr[:, Qn:])), pack.add(rb), abv)
    out = gu.run_plan(ctx, pack, b, {v313.buf: gu.hl_pack(x313, v313.cpitch), v128.buf: gu.hl_pack(x128, v128.cpitch), t4v.buf: gu.hl_pack(t4, t4v.cpitch),
                                     embv.buf: gu.hl_pack(emb, embv.cpitch), imgv.buf: gu.hl_pack(img, imgv.cpitch)},
                      {o1: ((B, H, W, 2), np.float32), o2: ((B, H, W, 2), np.float32), sub.buf: _shape(sub, B), abv.buf: _shape(abv, B)}, B)
    p = torch.softmax(torch.from_numpy(x313).double(), 1)
    close32(out[o1].transpose(0, 3, 1, 2), torch.einsum("oc,bchw->bohw", torch.from_numpy(w1).double(), p).float().numpy(), "softmax + 1x1", rtol=2e-5)
    t = torch.tanh(torch.einsum("oc,bchw->bohw", torch.from_numpy(w2).double(), torch.from_numpy(x128).double()) + torch.from_numpy(b2).double()[None, :, None, None]) * 110.0
    close32(out[o2].transpose(0, 3, 1, 2), t.float().numpy(), "1x1 + tanh", rtol=2e-5)
    assert np.array_equal(gu.hl_unpack(out[sub.buf], 128), gu.hl_unpack(gu.hl_pack(x128, v128.cpitch), 128)[:, :, ::2, ::2]), "[::2, ::2] must copy both planes"
    # reference tail: channels of t4 are ordered (dy*4+dx)*256 + c  ->  pixel_shuffle wants c*16 + (dy*4+dx)
    tt = torch.from_numpy(t4).double().reshape(B, 16, E, H, W).permute(0, 2, 1, 3, 4).reshape(B, 16 * E, H, W)
    feat = F.avg_pool2d(F.pad(F.pixel_shuffle(tt, 4), (1, 0, 1, 0), mode="replicate"), 2, stride=1)
    logits = torch.einsum("bqc,bchw->bqhw", torch.from_numpy(emb[:, :, 0, :Qn]).double().transpose(1, 2), feat)
    ab = F.conv2d(torch.cat([logits, torch.from_numpy(img).double()], 1), torch.from_numpy(Wr).double()[:, :, None, None], torch.from_numpy(rb).double())
    close32(gu.hl_unpack(out[abv.buf], 2), ab.float().numpy(), "DDColor tail (fold + shuffle + blur + projection)", rtol=2e-5)


@pytest.mark.parametrize("model", ["eccv16", "siggraph17"])
def test_precise_zhang_network_and_frame_at_config_size(ctx, model):
    """BASELINE configs[0] (eccv16 at its fixed 256 x 256, colorization/__init__.py:81) and siggraph17: the precise ab map against the fp32 oracle
    forward, then ModelColorization(precision="precise").colorize_frame on a 256 x 256 and a 16:9 frame against oracle/zhang.colorize_frame:
    the contract, p99 < 1.0 and >= 99 % of the pixels below 1.0"""
    from oracle import zhang
    from vsdeoldify_amd.colorization import ModelColorization
    from vsdeoldify_amd.zhang_net import ZhangGenerator
    sd = synth_zhang_state_dict(model, 5)
    tsd = {k: torch.as_tensor(v) for k, v in sd.items()}
    gen = ZhangGenerator(sd, model, precision="precise")
    w = nat.Weights(ctx, gen.blob)
    S = 256
    ops, bufs, i, o, names = gen.plan(S)
    net = nat.Net(ctx, w, ops, bufs, i, o, S, 2)
    imgs = np.stack([synthetic_gray_frame(3, S, S), synthetic_gray_frame(4, S, S)])
    net.upload(i, imgs)
    net.run_ops(0, len(ops), 2)
    ab = net.download(o, (2, S, S, 2), np.float32)
    net.close()
    w.close()
    for k in range(2):
        l_in = torch.Tensor(zhang.rgb2lab(imgs[k])[:, :, 0])[None, None]
        with torch.no_grad():
            ref = (zhang.eccv16_forward if model == "eccv16" else zhang.siggraph17_forward)(tsd, l_in)[0].numpy().transpose(1, 2, 0)
        d = np.abs(ab[k] - ref)
        print(f"{model} precise ab map {k}: max |d| {d.max():.2e} mean {d.mean():.2e} (max |ref| {np.abs(ref).max():.1f})")
        assert np.isfinite(ab[k]).all() and d.max() <= 2e-3 * max(1.0, np.abs(ref).max()), (model, d.max(), np.abs(ref).max())
    mc = ModelColorization(model, True, state_dict=sd, precision="precise")
    try:
        for hw in ((256, 256), (270, 480)):
            img = synthetic_gray_frame(7, hw[1], hw[0])
            got = mc.colorize_frame(img)
            ref = zhang.colorize_frame(tsd, model, img)
            de = imaging.delta_e00_images(got, ref)
            p99, frac = float(np.percentile(de, 99)), float((de < 1.0).mean())
            print(f"{model} precise frame {hw}: mean dE00 {de.mean():.5f} p99 {p99:.4f} below 1.0: {frac:.5f} max {de.max():.2f}")
            assert got.shape == img.shape and p99 < CONTRACT["p99"] and frac >= CONTRACT["frac_lt1"], (model, hw, p99, frac)
    finally:
        mc.close()


def test_precise_ddcolor_stages_match_the_oracle(ctx):
    """small-depth DDColor at 64 x 64: encoder / decoder feature maps and the ab map of the precise plan vs oracle/ddcolor.forward at fp32-class tolerance,
    then frame in -> frame out"""
    from oracle import ddcolor as D
    from oracle import zhang as Z
    from vsdeoldify_amd.ddcolor import DDColorRuntime
    SMALL = dict(depths=(1, 1, 2, 1), dec_layers=3)
    S, B = 64, 2
    sd = synth_ddcolor_state_dict(1, **SMALL)
    rt = DDColorRuntime(ctx, sd, precision="precise", **SMALL)
    try:
        r = np.random.default_rng(5)
        frames = r.integers(0, 256, (B, S, S, 1), dtype=np.uint8).repeat(3, -1)
        frames[1] = r.integers(0, 256, (S, S, 3), dtype=np.uint8)
        out = rt.colorize(frames)
        net = rt.net(S, B)
        gray = np.stack([Z.lab2rgb(np.concatenate([Z.rgb2lab(f)[..., :1], np.zeros((S, S, 2))], -1)) for f in frames]).astype(np.float32)
        with torch.no_grad():
            parts = D.forward(sd, torch.from_numpy(gray).permute(0, 3, 1, 2), return_parts=True, **SMALL)
        names, ops = net.names, net.plan_ops
        for opname, key, C in [("encoder.arch.norm3", "f3", 1536), ("decoder.layers.0.conv", "out0", 512), ("decoder.layers.2.conv", "out2", 256), ("refine_net.0.0", "ab", 2)]:
            op = ops[names.index(opname)]
            raw = net.download(int(op["dst"]), (B, int(op["Ho"]), int(op["Wo"]), int(op["dst_cpitch"])), np.float16)
            P, co = raw.shape[-1] // 2, int(op["dst_coff"])
            got = (raw[..., co:co + C].astype(np.float32) + raw[..., P + co:P + co + C].astype(np.float32) / 2048.0).transpose(0, 3, 1, 2)
            ref = parts[key].numpy()
            err = np.abs(got - ref).max()
            print(f"ddcolor precise {opname}: max |d| {err:.3e} (max |ref| {np.abs(ref).max():.2f})")
            assert err <= 2e-4 * max(1.0, float(np.abs(ref).max())), (opname, float(err), float(np.abs(ref).max()))
        for f, o in zip(frames, out):
            want = D.colorize_frame(sd, f, **SMALL)
            d = np.abs(o.astype(int) - want.astype(int))
            assert (d == 0).mean() > 0.995 and d.max() <= 1, (float((d == 0).mean()), int(d.max()))
    finally:
        rt.close()


@pytest.mark.parametrize("config", ["c3", "c4"])
def test_precise_ddcolor_configs_meet_the_contract_at_full_size(ctx, config):
    """BASELINE configs[2] / [3] through HAVCFrameColorizer(precision="precise") on a 1080p frame against the all-oracle graph (the same graph as
    tests/test_gpu_configs.py, whose fast-path thresholds are p99 < 2.6): the contract -- p99 < 1.0, >= 99 % of the pixels below 1.0"""
    from oracle import ddcolor as D
    from oracle import tweaks
    from vsdeoldify_amd import havc
    frame = synthetic_gray_frame(3, 1920, 1080)
    dsd, vsd = synth_ddcolor_state_dict(1), {"video": synth_state_dict("wide", 1)}
    hue_adj = "300:360|0.8,0.1"
    dd = dict(ddtweak_p=(havc.DEF_TWEAK_p, hue_adj), precision="precise")
    if config == "c3":
        col = havc.HAVCFrameColorizer(method=1, ddcolor_p=(1, 32, 1.0, 0.0, True), ddcolor_state_dict=dsd, **dd)
        fs = 512
    else:
        col = havc.HAVCFrameColorizer(method=2, mweight=0.4, deoldify_p=(0, 24, 1.0, 0.0), ddcolor_p=(1, 24, 1.0, 0.0, True), state_dicts=vsd, ddcolor_state_dict=dsd, **dd)
        fs = 384
    got = col.colorize(frame)
    sq = resample.resize_rgb8(frame, fs, fs)
    bb = tweaks.adjust_hue_range(D.colorize_frame(dsd, sq, input_size=fs), hue_adj)
    c = bb if config == "c3" else pipeline.combine_models(pipeline.model_image_render(vsd, "video", sq, 24, 0, True), bb, 2, 0.4)
    ref = pipeline.post_process(resample.resize_rgb8(c, 1920, 1080), frame)
    de = imaging.delta_e00_images(got, ref)
    p99, frac = float(np.percentile(de, 99)), float((de < 1.0).mean())
    print(f"{config} precise @1080p: mean dE00 {de.mean():.5f} p99 {p99:.4f} below 1.0: {frac:.5f} max {de.max():.2f}")
    assert got.shape == frame.shape and p99 < CONTRACT["p99"] and frac >= CONTRACT["frac_lt1"], (config, p99, frac)


@pytest.mark.parametrize("C,hw", [(512, (35, 35)), (256, (9, 13)), (768, (12, 12))])
def test_precise_attention_on_mfma_and_the_transposed_value_conv(ctx, C, hw):
    """fastai SelfAttention (fastai/layers.py:81-96) with the P . H product on MFMA (three-term splitting, csrc/precise.hip pattn_apply_mfma_kernel): the value
    map comes from a 1x1 conv whose precise epilogue stores it transposed as two planes [2][C][npitch]; vs float64.  N = 1225 / 117 / 144 keys (not
    multiples of the 64-query / 32-key tiles), probabilities from ~1 down to e^-30 (scores scaled up on purpose)."""
    H, W = hw
    B, d, N = 2, C // 8, hw[0] * hw[1]
    r = np.random.default_rng(C + H)
    x = r.standard_normal((B, C, H, W)).astype(np.float32)
    f = (r.standard_normal((B, d, H, W)) * 0.9).astype(np.float32)
    g = (r.standard_normal((B, d, H, W)) * 0.9).astype(np.float32)
    Wv = (r.standard_normal((C, C, 1, 1)) / np.sqrt(C) * 3).astype(np.float32)
    pack, b = WeightPack(), PlanBuilder(precise=True)
    xv, qk, av = b.tensor(H, W, C), b.tensor(H, W, 2 * d), b.tensor(H, W, C)
    npitch = (N + 63) // 64 * 64
    vT = b.buf(2 * C * npitch, 2, zero_init=True)
    b.conv("value", pack_conv(pack, Wv, xv.cmap, xv.span, precise=True), xv, vT, flags=nat.F_OUT_TRANSPOSED, Co=C, aux0=npitch)
    b.attention("attn", xv, qk, d, vT, npitch, av, 0.37, transposed=True)
    out = gu.run_plan(ctx, pack, b, {xv.buf: gu.hl_pack(x, xv.cpitch), qk.buf: gu.hl_pack(np.concatenate([f, g], 1), qk.cpitch)},
                      {av.buf: _shape(av, B), vT: ((B, 2, C, npitch), np.float16)}, B)
    xt = torch.from_numpy(x).double()
    h = F.conv2d(xt, torch.from_numpy(Wv).double()).reshape(B, C, N)
    vt = out[vT]
    got_h = vt[:, 0, :, :N].astype(np.float32) + vt[:, 1, :, :N].astype(np.float32) / 2048.0
    close32(got_h, h.float().numpy(), "transposed value conv")
    assert (vt[:, :, :, N:] == 0).all(), "columns beyond N must stay zero"
    ft, gt = (torch.from_numpy(a).double().reshape(B, d, N) for a in (f, g))
    beta = F.softmax(torch.bmm(ft.permute(0, 2, 1), gt), dim=1)
    ref = 0.37 * torch.bmm(h, beta) + xt.reshape(B, C, N)
    close32(gu.hl_unpack(out[av.buf], C), ref.float().reshape(B, C, H, W).numpy(), "self attention on MFMA", rtol=2e-5)
