#!/usr/bin/env python3
"""Where does the fp16-storage error of the HIP generator come from, and what is the fp32 noise floor?

TEST INFRASTRUCTURE (imports oracle/): a CPU simulation of the HIP path's rounding points on top of the oracle's
functional generator.  Every activation store / weight pack of the HIP plan (vsdeoldify_amd/deoldify_net.py) is a
named rounding point that can be switched between fp32 (identity), fp16 and bf16, so the contribution of each group
to the final-image CIEDE2000 can be measured without a GPU:

  python tests/precision_study.py --S 560 --floor          # fp32 noise floor (summation order / threads / layout)
  python tests/precision_study.py --S 560 --groups         # one group at a time in fp16, everything else fp32
  python tests/precision_study.py --S 560 --config all     # the whole HIP path

Products of two fp16 numbers are exact in fp32, so `conv2d(fp16-rounded x, fp16-rounded w)` in fp32 reproduces the
MFMA arithmetic (fp16 inputs, fp32 accumulate) up to summation order.
"""
import argparse
import os
import sys
import time

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import imaging, pipeline, unet as U          # noqa: E402
from vsdeoldify_amd.synth import synth_state_dict        # noqa: E402

GROUPS = ["input", "enc_w", "enc_act", "enc_stream", "mid_w", "mid_act", "dec_w", "dec_act", "attn", "l8_w", "l8_act", "tail_w", "r1", "r2x",
          "final_w"]


class Q:
    """rounding policy: group -> 'f16' | 'bf16' | None (fp32)"""

    def __init__(self, policy):
        self.p = policy

    def __call__(self, x, group):
        m = self.p.get(group)
        if m == "f16":
            return x.to(torch.float16).to(torch.float32)
        if m == "bf16":
            return x.to(torch.bfloat16).to(torch.float32)
        if m == "f16x2":                       # hi + lo split: ~22 bits
            hi = x.to(torch.float16).to(torch.float32)
            return hi + (x - hi).to(torch.float16).to(torch.float32)
        return x


# ---- Winograd F(2x2, 3x3) for the two 259 -> 259 tail convs (VERDICT r2 "next" 2-iv: parity-gated experiment) ----
# Y = A^T [ (G g G^T) (.) (B^T d B) ] A per 4x4 input tile / 2x2 output tile, 16 element-wise products per tile instead of 36 MACs per
# channel pair (2.25x fewer MFMAs).  On the MFMA path BOTH transformed operands are fp16: the transformed input B^T d B (sums of four
# activations: up to 4x the dynamic range, new rounding point "wino_in") and the transformed weights G g G^T ("wino_w", packed offline).
_BT = torch.tensor([[1., 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]])
_G = torch.tensor([[1., 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]])
_AT = torch.tensor([[1., 1, 1, 0], [0, 1, -1, -1]])


def winograd_conv3x3(x, w, bias, q):
    """x [1,C,H,W] (H, W even), w [K,C,3,3], pad 1 -> [1,K,H,W]; fp32 accumulation over fp16-rounded transformed operands"""
    _, C, H, W = x.shape
    K = w.shape[0]
    U = q(torch.einsum("ai,kcij,bj->kcab", _G, w, _G), "wino_w")                       # [K,C,4,4]
    xp = F.pad(x, (1, 1, 1, 1))
    d = xp.unfold(2, 4, 2).unfold(3, 4, 2)                                              # [1,C,H/2,W/2,4,4]
    V = q(torch.einsum("ai,nctuij,bj->nctuab", _BT, d, _BT), "wino_in")                 # B^T d B, B = _BT^T
    ty, tx = V.shape[2], V.shape[3]
    M = torch.einsum("kcab,ctab->ktab", U, V[0].reshape(C, ty * tx, 4, 4))              # 16 GEMMs, fp32 accumulation
    Y = torch.einsum("ia,ktab,jb->ktij", _AT, M, _AT).reshape(K, ty, tx, 2, 2)
    y = Y.permute(0, 1, 3, 2, 4).reshape(1, K, H, W)
    return y + bias.view(1, -1, 1, 1)


def bn_ss(sd, p):
    s = sd[p + ".weight"] / torch.sqrt(sd[p + ".running_var"] + U.EPS)
    return s, sd[p + ".bias"] - sd[p + ".running_mean"] * s


def forward(sd, x0, q, arch="wide", double_round_res=True):
    """The HIP plan's arithmetic: conv->BN folds, fused epilogues, rounding at every buffer store."""
    deep = arch == "deep"
    kind, nblk = U.RESNET_LAYERS["resnet34" if deep else "resnet101"]
    x0 = q(x0, "input")

    def enc_conv(x, wk, bnk, stride=1, pad=0, relu=True, res=None):
        s, sh = bn_ss(sd, bnk)
        w = q(sd[wk + ".weight"] * s[:, None, None, None], "enc_w")
        y = F.conv2d(x, w, sh, stride, pad)
        if res is not None:
            if double_round_res:
                y = q(y, "enc_act")            # LDS epilogue: fp16 image, then + residual
            y = F.relu(y + res)
        elif relu:
            y = F.relu(y)
        return q(y, "enc_act")

    p = "layers.0"
    x = enc_conv(x0, p + ".0", p + ".1", 2, 3)
    skips = [x]
    x = F.max_pool2d(x, 3, 2, 1)
    # residual stream: stored with the 'enc_stream' policy (fp16 in round 1); every conv that reads it sees q(., 'enc_act')
    for li, n in enumerate(nblk):
        for b in range(n):
            k = f"{p}.{4 + li}.{b}"
            stride = 2 if (li > 0 and b == 0) else 1
            xin = q(x, "enc_act")
            idt = x
            if k + ".downsample.0.weight" in sd:
                s_, sh_ = bn_ss(sd, k + ".downsample.1")
                idt = q(F.conv2d(xin, q(sd[k + ".downsample.0.weight"] * s_[:, None, None, None], "enc_w"), sh_, stride, 0), "enc_stream")
            if kind == "bottleneck":
                o = enc_conv(xin, k + ".conv1", k + ".bn1")
                o = enc_conv(o, k + ".conv2", k + ".bn2", stride, 1)
                wk, bnk, st, pd = k + ".conv3", k + ".bn3", 1, 0
            else:
                o = enc_conv(xin, k + ".conv1", k + ".bn1", stride, 1)
                wk, bnk, st, pd = k + ".conv2", k + ".bn2", 1, 1
            s_, sh_ = bn_ss(sd, bnk)
            y = F.conv2d(o, q(sd[wk + ".weight"] * s_[:, None, None, None], "enc_w"), sh_, st, pd)
            if double_round_res:
                y = q(y, "enc_stream")
            x = q(F.relu(y + idt), "enc_stream")
        if li < 3:
            skips.append(q(x, "enc_act"))
    x = q(x, "enc_act")

    s, sh = bn_ss(sd, "layers.1")
    x = q(F.relu(x * s[None, :, None, None] + sh[None, :, None, None]), "mid_act")

    def dec_conv(x, pk, gw, ga):
        s, sh = bn_ss(sd, pk + ".2")
        y = F.relu(F.conv2d(x, q(U.conv_w(sd, pk + ".0"), gw), sd.get(pk + ".0.bias"), 1, 1))
        return q(y * s[None, :, None, None] + sh[None, :, None, None], ga)

    x = dec_conv(x, "layers.3.0", "mid_w", "mid_act")
    x = dec_conv(x, "layers.3.1", "mid_w", "mid_act")

    def shuffle(x, w, bias, ga):
        y = q(F.relu(F.conv2d(x, w, bias)), ga)                      # shuffled values are rounded, then blurred in fp32
        y = F.pixel_shuffle(y, 2)
        return q(F.avg_pool2d(F.pad(y, (1, 0, 1, 0), mode="replicate"), 2, stride=1), ga)

    def attention(pk, x):
        size = x.size()
        xf = x.view(*size[:2], -1)
        f = q(F.conv1d(xf, q(U.fold_spectral(sd, pk + ".query"), "dec_w")), "attn")
        g = q(F.conv1d(xf, q(U.fold_spectral(sd, pk + ".key"), "dec_w")), "attn")
        h = q(F.conv1d(xf, q(U.fold_spectral(sd, pk + ".value"), "dec_w")), "attn")
        sc = torch.bmm(f.permute(0, 2, 1).contiguous(), g)
        m = sc.max(dim=1, keepdim=True).values
        pe = torch.exp(sc - m)
        l = pe.sum(dim=1, keepdim=True)
        o = torch.bmm(h, q(pe, "attn")) / l
        return q(sd[pk + ".gamma"] * o + xf, "dec_act").view(*size).contiguous()

    for i, skip in enumerate(reversed(skips)):
        pk = f"layers.{4 + i}"
        s, sh = bn_ss(sd, pk + ".shuf.conv.1")
        w = q(U.conv_w(sd, pk + ".shuf.conv.0") * s[:, None, None, None], "dec_w")
        up = shuffle(x, w, sh, "dec_act")
        if skip.shape[-2:] != up.shape[-2:]:
            up = F.interpolate(up, skip.shape[-2:], mode="nearest")
        s, sh = bn_ss(sd, pk + ".bn")
        sk = q(F.relu(skip * s[None, :, None, None] + sh[None, :, None, None]), "dec_act")
        cat = torch.cat([F.relu(up), sk], dim=1)
        if deep:
            x = dec_conv(cat, pk + ".conv1", "dec_w", "dec_act")
            x = dec_conv(x, pk + ".conv2", "dec_w", "dec_act")
            if pk + ".conv2.3.gamma" in sd:
                x = attention(pk + ".conv2.3", x)
        else:
            x = dec_conv(cat, pk + ".conv", "dec_w", "dec_act")
            if pk + ".conv.3.gamma" in sd:
                x = attention(pk + ".conv.3", x)

    x = shuffle(x, q(U.conv_w(sd, "layers.8.conv.0"), "l8_w"), sd["layers.8.conv.0.bias"], "l8_act")
    x = torch.cat([x, x0], dim=1)
    if q.p.get("winograd"):
        r = q(F.relu(winograd_conv3x3(x, U.conv_w(sd, "layers.10.layers.0.0"), sd["layers.10.layers.0.0.bias"], q)), "r1")
        r = F.relu(winograd_conv3x3(r, U.conv_w(sd, "layers.10.layers.1.0"), sd["layers.10.layers.1.0.bias"], q))
    else:
        r = q(F.relu(F.conv2d(x, q(U.conv_w(sd, "layers.10.layers.0.0"), "tail_w"), sd["layers.10.layers.0.0.bias"], 1, 1)), "r1")
        r = F.relu(F.conv2d(r, q(U.conv_w(sd, "layers.10.layers.1.0"), "tail_w"), sd["layers.10.layers.1.0.bias"], 1, 1))
    x = q(x + r, "r2x")
    y = F.conv2d(x, q(U.conv_w(sd, "layers.11.0"), "final_w"), sd["layers.11.0.bias"])
    return torch.sigmoid(y) * 6.0 - 3.0


def frame(S, seed):
    r = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:S, 0:S]
    low = np.kron(r.standard_normal((S // 8 + 2, S // 8 + 2)), np.ones((8, 8)))[:S, :S]
    luma = np.clip(128 + 48 * low + 32 * (xx / S - 0.5) * 2 + 6 * r.standard_normal((S, S)), 0, 255).astype(np.uint8)
    return np.stack([luma] * 3, -1)


def to_images(y, img):
    raw = imaging.model_output_u8(y[0].numpy())
    return raw, pipeline.post_process(raw, img)


def stats(a, b):
    de = imaging.delta_e00_images(a, b)
    d = np.abs(a.astype(np.int32) - b.astype(np.int32))
    return dict(mean=float(de.mean()), p99=float(np.percentile(de, 99)), p999=float(np.percentile(de, 99.9)), max=float(de.max()),
                frac_lt1=float((de < 1.0).mean()), w1=float((d <= 1).mean()), maxlsb=int(d.max()))


def fmt(name, raw, fin):
    return (f"{name:34s} raw: mean {raw['mean']:.4f} p99 {raw['p99']:.3f} max {raw['max']:.2f} w1 {raw['w1']:.5f} maxlsb {raw['maxlsb']}"
            f" | final: mean {fin['mean']:.4f} p99 {fin['p99']:.3f} p99.9 {fin['p999']:.3f} max {fin['max']:.2f} dE<1 {fin['frac_lt1']:.5f}")


def configs():
    return {
        "all": {g: "f16" for g in GROUPS},
        "all_bf16": {g: "bf16" for g in GROUPS},
        # Winograd F(2x2,3x3) tail: everything as "all" + transformed inputs / weights in fp16 (the MFMA operands of that scheme)
        "all_wino": {**{g: "f16" for g in GROUPS}, "winograd": True, "wino_in": "f16", "wino_w": "f16"},
        "wino_exact": {"winograd": True},                                  # fp32 Winograd alone: the algorithm's own (re-association) error
        "only_wino_in": {"winograd": True, "wino_in": "f16"},
        "only_wino_w": {"winograd": True, "wino_w": "f16"},
        "all_single_round": {g: "f16" for g in GROUPS},
        "tail32": {g: "f16" for g in GROUPS if g not in ("r2x", "final_w")},
        "tail32_r1x2": {**{g: "f16" for g in GROUPS if g not in ("r2x", "final_w")}, "r1": "f16x2"},
        "tail_l8_32": {g: "f16" for g in GROUPS if g not in ("r2x", "final_w", "l8_act", "r1")},
        "no_w": {g: "f16" for g in GROUPS if not g.endswith("_w")},
        "only_w": {g: "f16" for g in GROUPS if g.endswith("_w")},
        "enc32": {g: "f16" for g in GROUPS if not g.startswith("enc")},
        "input32": {g: "f16" for g in GROUPS if g != "input"},
        "stream32": {g: "f16" for g in GROUPS if g != "enc_stream"},
        "stream32_in": {g: "f16" for g in GROUPS if g not in ("enc_stream", "input")},
        "stream32_in_fin": {g: "f16" for g in GROUPS if g not in ("enc_stream", "input", "final_w", "r2x")},
        "stream32_encw": {g: "f16" for g in GROUPS if g not in ("enc_stream", "input", "final_w", "r2x", "enc_w")},
        "enc32": {g: "f16" for g in GROUPS if not g.startswith("enc")},
        "enc32_in_fin": {g: "f16" for g in GROUPS if not g.startswith("enc") and g not in ("input", "final_w", "r2x")},
    }


def full1080(args):
    """BASELINE configs[1] on the bench's first synthetic frame: oracle (fp32) vs the simulated HIP arithmetic."""
    from vsdeoldify_amd.clip import synthetic_gray_frame
    sds = {"video": synth_state_dict("wide", 1), "stable": synth_state_dict("wide", 2)}
    img = synthetic_gray_frame(0, 1920, 1080)
    t0 = time.time()
    ref = pipeline.colorize_frame_fullsize(sds, "stable", img, 35, 0.5)
    print(f"# 1080p stable rf=35, oracle {time.time() - t0:.1f} s", flush=True)
    orig = U.unet_forward
    named = configs()
    try:
        for c in args.config:
            if c == "fp32_1thread":
                torch.set_num_threads(1)
                U.unet_forward = orig
            else:
                pol = Q(named[c])
                U.unet_forward = lambda sd, x0, arch="wide", return_presigmoid=False, pol=pol: forward(sd, x0, pol, arch)
            got = pipeline.colorize_frame_fullsize(sds, "stable", img, 35, 0.5)
            torch.set_num_threads(args.threads)
            st = stats(got, ref)
            print(f"1080p {c:22s} mean {st['mean']:.4f} p99 {st['p99']:.3f} p99.9 {st['p999']:.3f} max {st['max']:.2f} dE<1 {st['frac_lt1']:.5f} "
                  f"bytes within 1 LSB {st['w1']:.5f} max LSB {st['maxlsb']}", flush=True)
    finally:
        U.unet_forward = orig


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--S", type=int, default=320)
    ap.add_argument("--arch", default="wide")
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--floor", action="store_true")
    ap.add_argument("--groups", action="store_true")
    ap.add_argument("--config", action="append", default=[])
    ap.add_argument("--threads", type=int, default=8)
    ap.add_argument("--full1080", action="store_true", help="the bench frame: 1080p, stable (video + stable blend), Spline64, luma merge")
    args = ap.parse_args()
    torch.set_num_threads(args.threads)
    if args.full1080:
        return full1080(args)
    sd = pipeline._to_torch_sd(synth_state_dict(args.arch, args.seed))
    img = frame(args.S, 7)
    x0 = torch.from_numpy(imaging.model_input(img))
    print(f"# S={args.S} arch={args.arch} seed={args.seed} threads={args.threads} torch {torch.__version__}")
    with torch.no_grad():
        t0 = time.time()
        y_ref = U.unet_forward(sd, x0, args.arch)
        print(f"# oracle fp32 forward {time.time() - t0:.1f} s")
        raw_ref, fin_ref = to_images(y_ref, img)

        def report(name, y):
            raw, fin = to_images(y, img)
            print(fmt(name, stats(raw, raw_ref), stats(fin, fin_ref)), flush=True)

        if args.floor:
            sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}
            y64 = U.unet_forward(sd64, x0.double(), args.arch).float()
            raw64, fin64 = to_images(y64, img)
            print(fmt("fp32 oracle vs fp64", stats(raw_ref, raw64), stats(fin_ref, fin64)), flush=True)
            torch.set_num_threads(1)
            report("fp32 1 thread vs 8 threads", U.unet_forward(sd, x0, args.arch))
            torch.set_num_threads(args.threads)
            xcl = x0.contiguous(memory_format=torch.channels_last)
            sdcl = {k: (v.contiguous(memory_format=torch.channels_last) if v.dim() == 4 else v) for k, v in sd.items()}
            report("fp32 channels_last", U.unet_forward(sdcl, xcl, args.arch))
            report("fp32 folded-BN restatement", forward(sd, x0, Q({}), args.arch))
            # fp32 with mkldnn off (different conv algorithm / summation order)
            torch.backends.mkldnn.enabled = False
            report("fp32 mkldnn off", U.unet_forward(sd, x0, args.arch))
            torch.backends.mkldnn.enabled = True
        if args.groups:
            for g in GROUPS:
                report(f"only {g} fp16", forward(sd, x0, Q({g: "f16"}), args.arch))
        named = configs()
        for c in args.config:
            report(f"config {c}", forward(sd, x0, Q(named[c]), args.arch, double_round_res=(c != "all_single_round")))


if __name__ == "__main__":
    main()
