"""DeOldify generators (video / stable = DynamicUnetWide on resnet101, artistic = DynamicUnetDeep on
resnet34) as a libhavc_mi355 weight blob + execution plan.

Topology restated from the reference (deoldify/unet.py:94-285, deoldify/layers.py:8-46,
fastai/layers.py:81-220, torchvision resnet via fastai/vision/learner.py:54-63):
this module only decides WHAT runs (op order, shapes, which buffers are concatenated); all arithmetic
is in the HIP kernels.
"""
import numpy as np

from . import _native as nat
from .plan import (TAG_TAIL_RES, PlanBuilder, View, WeightPack, bn_scale_shift, conv_weight, fold_spectral, pack_conv,
                   pad_to, pitch_for, to_np)

RESNET = {"wide": ("bottleneck", [3, 4, 23, 3]), "deep": ("basic", [3, 4, 6, 3])}
import os
FUSE_BLUR_MIN_H = int(os.environ.get("HAVC_FUSE_BLUR_MIN_H", "64"))      # smallest low-res side the fused shuffle + blur conv is used for
Y_RANGE = (-3.0, 3.0)          # SigmoidRange(*y_range), deoldify/generators.py:33,111


def conv_out(n, k, s, p, d=1):
    return (n + 2 * p - d * (k - 1) - 1) // s + 1


def splitk_count(rows, Npad, Kc):
    """split-K count of a conv with `rows` GEMM rows: one frame of a generator leaves the encoder / bottleneck convs at 5 - 40 output tiles
    for 256 CUs with 4 - 72 K-64 stages in series; cutting K puts up to ~256 blocks in flight (every part keeps >= 2 stages)"""
    stages = Kc // 8
    tiles = ((rows + 127) // 128) * ((Npad + 127) // 128)
    if tiles >= 128 or stages < 4:
        return 0
    n = min(15, 256 // tiles, stages // 2)
    return n if n >= 2 else 0


class _SplitBuilder(PlanBuilder):
    """PlanBuilder of the low-latency plans: adds the split-K count of the shape to plain convs"""

    def __init__(self, frames):
        super().__init__()
        self.frames = frames

    def conv(self, name, pc, x, y, stride=1, pad=0, dil=1, flags=0, **kw):
        if not (flags & (nat.F_W_FROM_BUF | nat.F_PS_BLUR | nat.F_FUSE_PROJ | nat.F_FUSE_RGB8 | nat.F_OUT_RGB8)) and pc.Npad % 256 != 16:
            Ho, Wo = conv_out(x.H, pc.kh, stride, pad, dil), conv_out(x.W, pc.kw, stride, pad, dil)
            flags |= nat.F_SPLITK(splitk_count(Ho * Wo * self.frames, pc.Npad, pc.Kc))
        return super().conv(name, pc, x, y, stride=stride, pad=pad, dil=dil, flags=flags, **kw)


class DeoldifyGenerator:
    """Packs a reference state dict once; emits a plan per render size S = render_factor * 16."""

    def __init__(self, state_dict, arch="wide", fuse_final=True, fuse_blur=True, precision="fast"):
        """precision: "fast" = fp16 activations / fp16 MFMA operands / fp32 accumulate; "precise" = fp32-class arithmetic
        like the reference's (deoldify/filters.py:45-68): hi / lo fp16 pairs, three-segment convs, fp32 attention (HAVC_F_PRECISE), the
        same fusions as the fast plan (shuffle + blur since round 6, layers.11 + SigmoidRange + u8 since round 5) with fp32 epilogues."""
        assert arch in RESNET and precision in ("fast", "precise")
        self.precise = precision == "precise"
        if self.precise:
            # round 6: the fused shuffle + blur epilogue exists for pairs too (two passes of 32 channels through the 128 KiB LDS image,
            # conv_pipe_epilogue.inc); HAVC_PRECISE_FUSE_BLUR=0 keeps the two-op chain (A/B runs; same bytes)
            fuse_blur = fuse_blur and os.environ.get("HAVC_PRECISE_FUSE_BLUR", "1") != "0"
            fuse_final = fuse_final and os.environ.get("HAVC_PRECISE_FUSE_FINAL", "1") != "0"      # round 5: layers.11 + SigmoidRange + u8 in the precise epilogue
        self.sd, self.arch, self.fuse_final, self.fuse_blur = to_np(state_dict), arch, fuse_final, fuse_blur
        self.pack, self._pc, self._vec = WeightPack(), {}, {}
        self._frozen = False
        self.plan(64)                      # dry run: packs every tensor
        self.blob = self.pack.blob()
        self._frozen = True

    # ---- offline conversion: packed blob + packing table on disk ---------------------------------
    FORMAT = "havc-deoldify-blob/1"

    def save(self, path):
        """Write the packed model: everything havc_weights_load / plan() need, nothing of the fp32 state dict.  One
        uncompressed .npz: `blob` (the device image, byte for byte), `meta` (JSON: arch, fusion flags, the offset table of
        every packed conv / vector, and the state-dict SHAPES the plan emitter looks at plus the attention gammas)."""
        import json
        from dataclasses import asdict
        meta = {"format": self.FORMAT, "arch": self.arch, "fuse_final": self.fuse_final, "fuse_blur": self.fuse_blur,
                "precision": "precise" if self.precise else "fast",
                "convs": {k: asdict(v) for k, v in self._pc.items()}, "vecs": {k: list(v) for k, v in self._vec.items()},
                "shapes": {k: list(v.shape) for k, v in self.sd.items()},
                "scalars": {k: float(v.reshape(-1)[0]) for k, v in self.sd.items() if k.endswith(".gamma")}}
        with open(path, "wb") as f:
            np.savez(f, blob=np.frombuffer(self.blob, dtype=np.uint8), meta=np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8))

    @classmethod
    def load(cls, path):
        """Inverse of save(): no packing work, the plan emitter runs on a shape-only skeleton of the state dict."""
        import json
        from .plan import PackedConv
        z = np.load(path)
        meta = json.loads(bytes(z["meta"]).decode())
        if meta.get("format") != cls.FORMAT:
            raise ValueError(f"{path}: not a {cls.FORMAT} file")
        g = cls.__new__(cls)
        g.arch, g.fuse_final, g.fuse_blur = meta["arch"], meta["fuse_final"], meta["fuse_blur"]
        g.precise = meta.get("precision", "fast") == "precise"
        g.sd = {k: np.broadcast_to(np.float32(meta["scalars"].get(k, 0.0)), tuple(shp)) for k, shp in meta["shapes"].items()}
        g._pc = {k: PackedConv(**v) for k, v in meta["convs"].items()}
        g._vec = {k: tuple(v) for k, v in meta["vecs"].items()}
        g.pack, g.blob, g._frozen = None, z["blob"].tobytes(), True
        return g

    # ---- cached packing helpers -------------------------------------------------------------
    def _conv(self, key, fn):
        if key not in self._pc:
            assert not self._frozen, key
            self._pc[key] = fn()
        return self._pc[key]

    def _shuf_conv(self, key, make, fuse):
        """pixel-shuffle conv weights in both row orders (plain / HAVC_F_PS_BLUR): which one a render size uses depends on
        the size, and everything is packed once in the constructor's dry run."""
        if not self._frozen:
            pc = self._conv(key, lambda: make(True))
            if self.fuse_blur and self._blur_pad_ok(pc.Cout // 4):
                self._conv(key + "#blur", lambda: make("blur"))
        return self._pc[key + ("#blur" if fuse else "")]

    def _fuse_blur(self, x, up_c, out_hw):
        """HAVC_F_PS_BLUR applies when the blur is not followed by a resize, the channel count tiles by 64 and the 15/16
        tile overlap wastes little (280 -> 19 tiles of 15 = 1.8 %, 140 -> 7 %, 70 -> 14 %: round 5 measured the 70 -> 140 stage too: 1.63 -> 1.08 ms per 64 frames)."""
        return self.fuse_blur and 2 * x.H == out_hw and self._blur_pad_ok(up_c) and x.H >= FUSE_BLUR_MIN_H

    @staticmethod
    def _blur_pad_ok(up_c):
        """channels per sub-pixel are padded to a multiple of 64 with zero weight rows: worth it up to ~15 % padding (300 -> 320, 336 -> 384)"""
        return up_c % 4 == 0 and pad_to(up_c, 64) <= 1.15 * up_c

    def _vecs(self, key, fn):
        if key not in self._vec:
            assert not self._frozen, key
            self._vec[key] = tuple(self.pack.add(np.asarray(v, np.float32)) for v in fn())
        return self._vec[key]

    def _enc_conv(self, wkey, bnkey, x):
        """conv (no bias) -> BN folded into weight/bias (no activation in between in torchvision resnet)."""
        def make():
            W = self.sd[wkey + ".weight"].astype(np.float32)
            s, sh = bn_scale_shift(self.sd, bnkey)
            return pack_conv(self.pack, W * s[:, None, None, None], x.cmap, x.span, bias=sh, precise=self.precise)
        return self._conv(wkey, make)

    def _dec_conv(self, b, p, x, ks=3):
        """custom_conv_layer(norm=Spectral, extra_bn): conv -> ReLU -> BN  (deoldify/layers.py:28-45)."""
        def make():
            s, sh = bn_scale_shift(self.sd, p + ".2")
            return pack_conv(self.pack, conv_weight(self.sd, p + ".0"), x.cmap, x.span, scale=s, shift=sh, precise=self.precise)
        pc = self._conv(p, make)
        y = b.tensor(x.H, x.W, pc.Cout)
        b.conv(p, pc, x, y, pad=ks // 2, flags=nat.F_RELU_PRE | nat.F_AFFINE)
        return y

    def _attention(self, b, p, x):
        """fastai SelfAttention (fastai/layers.py:81-96)."""
        sd, C = self.sd, x.C
        d = C // 8
        pc_qk = self._conv(p + ".qk", lambda: pack_conv(
            self.pack, np.concatenate([fold_spectral(sd, p + ".query"), fold_spectral(sd, p + ".key")])[..., None],
            x.cmap, x.span, precise=self.precise))
        pc_v = self._conv(p + ".value", lambda: pack_conv(self.pack, fold_spectral(sd, p + ".value")[..., None],
                                                          x.cmap, x.span, precise=self.precise))
        qk = b.tensor(x.H, x.W, 2 * d)
        b.conv(p + ".qk", pc_qk, x, qk)
        N = x.H * x.W
        if self.precise and C % 256 == 0 and d % 8 == 0 and d <= 128 and os.environ.get("HAVC_PRECISE_ATTN_MFMA", "1") != "0":
            # round 5: the value conv stores its map TRANSPOSED as two planes [2][C][npitch] (hi, lo) and the attention's P . H product runs on
            # MFMA with the three-term splitting (11.3 -> ~3 ms per 16 frames at 560 x 560); S = f . g and the softmax stay fp32 VALU
            npitch = pad_to(N, 64)
            vT = b.buf(2 * C * npitch, 2, zero_init=True)
            b.conv(p + ".value", pc_v, x, vT, flags=nat.F_OUT_TRANSPOSED, Co=C, aux0=npitch)
            y = b.tensor(x.H, x.W, C)
            b.attention(p, x, qk, d, vT, npitch, y, float(sd[p + ".gamma"].reshape(-1)[0]), transposed=True)
            return y
        if self.precise:                       # fp32 attention kernels read the value map as an ordinary NHWC hi / lo tensor
            hv = b.tensor(x.H, x.W, C)
            b.conv(p + ".value", pc_v, x, hv)
            y = b.tensor(x.H, x.W, C)
            b.attention(p, x, qk, d, hv.buf, hv.cpitch, y, float(sd[p + ".gamma"].reshape(-1)[0]))
            return y
        npitch = pad_to(N, 64)
        vT = b.buf(C * npitch, 2, zero_init=True)
        b.conv(p + ".value", pc_v, x, vT, flags=nat.F_OUT_TRANSPOSED, Co=C, aux0=npitch)
        y = b.tensor(x.H, x.W, C)
        b.attention(p, x, qk, d, vT, npitch, y, float(sd[p + ".gamma"].reshape(-1)[0]))
        return y

    # ---- plan ---------------------------------------------------------------------------------
    def plan(self, S, split_for_frames=0):
        """split_for_frames = n > 0: a LOW-LATENCY plan for nets that run n frames per launch (the reference's per-frame call shape, n = 1): convs
        that give the 256 CUs fewer than 128 output tiles get a split-K count (HAVC_F_SPLITK: a block per (tile, K part), fp32 partial sums
        added in a fixed order).  Results differ from the unsplit plan in fp32 summation order only; the default plan (0) keeps the
        batch-independent bytes."""
        assert S % 16 == 0 and S >= 32, "render size must be render_factor*16"
        sd, deep = self.sd, self.arch == "deep"
        kind, nblk = RESNET[self.arch]
        b = _SplitBuilder(split_for_frames) if (split_for_frames and not self.precise) else PlanBuilder(precise=self.precise)
        pm = b.pm
        in_buf, out_buf = b.buf(S * S * 3, 1), b.buf(S * S * 3, 1)

        c8 = sd["layers.8.conv.0.weight_v"].shape[0] // 4          # channels after the last pixel shuffle
        c8s = pad_to(c8, 8)
        tail_span = c8s + 8
        tail_pitch = pitch_for(tail_span) * pm                     # 264 -> 320: 128-byte aligned pixel rows (precise: hi | lo)
        tail_buf = b.buf(S * S * tail_pitch, 2, zero_init=c8s != c8)
        tail_cmap = np.concatenate([np.arange(c8), c8s + np.arange(3)])
        x0 = b.tensor(S, S, 3, zero_init=False)
        b.prep_rgb8("prep", in_buf, S, x0, View(tail_buf, c8s, tail_pitch, S, S, 3, 8), y1_fill=tail_pitch // pm - tail_span)

        # ---- encoder: torchvision resnet children()[:-2] ----
        e = "layers.0"
        H2 = conv_out(S, 7, 2, 3)
        e2 = b.tensor(H2, H2, 64)
        b.conv(e + ".0", self._enc_conv(e + ".0", e + ".1", x0), x0, e2, stride=2, pad=3, flags=nat.F_RELU_PRE)
        H4 = conv_out(H2, 3, 2, 1)
        x = b.tensor(H4, H4, 64)
        b.maxpool(e + ".3", e2, x)
        skips = [e2]
        for li, n in enumerate(nblk):
            planes = 64 * 2 ** li
            for bi in range(n):
                q = f"{e}.{4 + li}.{bi}"
                stride = 2 if (li > 0 and bi == 0) else 1
                Ho = conv_out(x.H, 3, stride, 1)
                idt = x
                if q + ".downsample.0.weight" in sd:
                    pc = self._enc_conv(q + ".downsample.0", q + ".downsample.1", x)
                    idt = b.tensor(Ho, Ho, pc.Cout)
                    b.conv(q + ".downsample", pc, x, idt, stride=stride)
                if kind == "bottleneck":
                    t1 = b.tensor(x.H, x.W, planes)
                    b.conv(q + ".conv1", self._enc_conv(q + ".conv1", q + ".bn1", x), x, t1, flags=nat.F_RELU_PRE)
                    t2 = b.tensor(Ho, Ho, planes)
                    b.conv(q + ".conv2", self._enc_conv(q + ".conv2", q + ".bn2", t1), t1, t2, stride=stride, pad=1,
                           flags=nat.F_RELU_PRE)
                    y = b.tensor(Ho, Ho, planes * 4)
                    b.conv(q + ".conv3", self._enc_conv(q + ".conv3", q + ".bn3", t2), t2, y,
                           flags=nat.F_RESIDUAL | nat.F_RELU_POST, res=idt)
                else:
                    t1 = b.tensor(Ho, Ho, planes)
                    b.conv(q + ".conv1", self._enc_conv(q + ".conv1", q + ".bn1", x), x, t1, stride=stride, pad=1,
                           flags=nat.F_RELU_PRE)
                    y = b.tensor(Ho, Ho, planes)
                    b.conv(q + ".conv2", self._enc_conv(q + ".conv2", q + ".bn2", t1), t1, y, pad=1,
                           flags=nat.F_RESIDUAL | nat.F_RELU_POST, res=idt)
                x = y
            if li < 3:
                skips.append(x)

        # ---- layers.1 BN, layers.2 ReLU, layers.3 middle_conv (unet.py:236-246) ----
        so, sho = self._vecs("layers.1", lambda: bn_scale_shift(sd, "layers.1"))
        m = b.tensor(x.H, x.W, x.C)
        b.affine("layers.1", x, m, so, sho, relu=True)
        x = self._dec_conv(b, "layers.3.0", m)
        x = self._dec_conv(b, "layers.3.1", x)

        # ---- layers.4-7 UnetBlockWide / UnetBlockDeep (unet.py:55-91,170-205) ----
        for i, skip in enumerate(reversed(skips)):
            p = f"layers.{4 + i}"

            fuse = self._fuse_blur(x, sd[p + ".shuf.conv.0.weight_orig" if p + ".shuf.conv.0.weight_orig" in sd else
                                         p + ".shuf.conv.0.weight"].shape[0] // 4, skip.H)

            def make_shuf(order, p=p, x=x):
                s, sh = bn_scale_shift(sd, p + ".shuf.conv.1")      # conv -> BN, no activation between: fold
                W = conv_weight(sd, p + ".shuf.conv.0")
                return pack_conv(self.pack, W * s[:, None, None, None], x.cmap, x.span, bias=sh, pixshuf=order, precise=self.precise)
            pc = self._shuf_conv(p + ".shuf", make_shuf, fuse)
            up_c = pc.Cout // 4
            ups, sks = pad_to(up_c, 8), pad_to(skip.C, 8)
            cat_span = ups + sks
            cat_pitch = pitch_for(cat_span) * pm
            cat_buf = b.buf(skip.H * skip.W * cat_pitch, 2, zero_init=(ups != up_c or sks != skip.C))
            if fuse:
                # 1x1 conv + BN + ReLU + PixelShuffle(2) + blur in ONE kernel (HAVC_F_PS_BLUR): the shuffled tensor never
                # reaches HBM (saves a write + a read of it: 5.1 GB per 16 frames at 560^2)
                b.conv(p + ".shuf+blur", pc, x, View(cat_buf, 0, cat_pitch, skip.H, skip.W, up_c, ups),
                       flags=nat.F_RELU_PRE | nat.F_OUT_PIXSHUF | nat.F_PS_BLUR)
            else:
                ps = b.tensor(2 * x.H, 2 * x.W, up_c)
                b.conv(p + ".shuf", pc, x, ps, flags=nat.F_RELU_PRE | nat.F_OUT_PIXSHUF)
                # blur (+ nearest resize when the shuffled size != skip size, e.g. 36 -> 35 at rf=35)
                b.blur_resize(p + ".blur", ps, View(cat_buf, 0, cat_pitch, skip.H, skip.W, up_c, ups))
            so, sho = self._vecs(p + ".bn", lambda p=p: bn_scale_shift(sd, p + ".bn"))
            b.affine(p + ".bn", skip, View(cat_buf, ups, cat_pitch, skip.H, skip.W, skip.C, sks), so, sho, relu=True)
            cat = View(cat_buf, 0, cat_pitch, skip.H, skip.W, up_c + skip.C, cat_span,
                       np.concatenate([np.arange(up_c), ups + np.arange(skip.C)]))
            if deep:
                x = self._dec_conv(b, p + ".conv1", cat)
                x = self._dec_conv(b, p + ".conv2", x)
                if p + ".conv2.3.gamma" in sd:
                    x = self._attention(b, p + ".conv2.3", x)
            else:
                x = self._dec_conv(b, p + ".conv", cat)
                if p + ".conv.3.gamma" in sd:
                    x = self._attention(b, p + ".conv.3", x)

        # ---- layers.8 PixelShuffle_ICNR (weight norm, bias, no BN) -> blur -> layers.9 dense merge ----
        fuse8 = self._fuse_blur(x, c8, S)
        pc = self._shuf_conv("layers.8", lambda order, x=x: pack_conv(self.pack, conv_weight(sd, "layers.8.conv.0"), x.cmap, x.span,
                                                                       bias=sd["layers.8.conv.0.bias"], pixshuf=order, precise=self.precise), fuse8)
        assert 2 * x.H == S
        if fuse8:
            b.conv("layers.8+blur", pc, x, View(tail_buf, 0, tail_pitch, S, S, c8, c8s),
                   flags=nat.F_RELU_PRE | nat.F_OUT_PIXSHUF | nat.F_PS_BLUR)
        else:
            ps8 = b.tensor(2 * x.H, 2 * x.W, c8)
            b.conv("layers.8", pc, x, ps8, flags=nat.F_RELU_PRE | nat.F_OUT_PIXSHUF)
            b.blur_resize("layers.8.blur", ps8, View(tail_buf, 0, tail_pitch, S, S, c8, c8s))
        cat = View(tail_buf, 0, tail_pitch, S, S, c8 + 3, tail_span, tail_cmap)

        # ---- layers.10 res_block: 2 x (spectral conv3x3 + bias -> ReLU), + input; layers.11/12 ----
        def res_pc(key):
            return self._conv(key, lambda: pack_conv(self.pack, conv_weight(sd, key), tail_cmap, tail_span,
                                                     bias=sd[key + ".bias"], omap=tail_cmap, ospan=tail_span, precise=self.precise))
        r1 = View(b.buf(S * S * tail_pitch), 0, tail_pitch, S, S, c8 + 3, tail_span, tail_cmap)
        b.conv("layers.10.layers.0.0", res_pc("layers.10.layers.0.0"), cat, r1, pad=1, flags=nat.F_RELU_PRE,
               tag=TAG_TAIL_RES)
        pc2 = res_pc("layers.10.layers.1.0")
        if pc2.Npad == 272 and self.fuse_final:
            # layers.11 (1x1 conv to RGB) + SigmoidRange + denormalise + u8 run in the epilogue of the second res_block
            # conv: r2 (2.65 GB per 16 frames at 560^2) is never written or re-read (HAVC_F_FUSE_RGB8).
            def fused():
                W = conv_weight(sd, "layers.11.0").astype(np.float32)[:, :, 0, 0]            # [3, 259]
                fw = np.zeros((3, pc2.Npad), np.float32)
                fw[:, tail_cmap] = W
                return fw, sd["layers.11.0.bias"].astype(np.float32)
            fw_off, fb_off = self._vecs("layers.11.0#fused", fused)
            oi = b.conv("layers.10.layers.1.0+11", pc2, r1, View(r1.buf, 0, tail_pitch, S, S, c8 + 3, tail_span, tail_cmap),
                        pad=1, flags=nat.F_RELU_PRE | nat.F_RESIDUAL | nat.F_FUSE_RGB8, res=cat, tag=TAG_TAIL_RES,
                        f=(Y_RANGE[0], Y_RANGE[1], 0, 0), aux0=out_buf)
            b.ops[oi]["scale_off"], b.ops[oi]["shift_off"] = fw_off, fb_off
            b.ops[oi]["flops"] += 2 * S * S * 3 * (c8 + 3)
        else:
            r2 = View(b.buf(S * S * tail_pitch), 0, tail_pitch, S, S, c8 + 3, tail_span, tail_cmap)
            b.conv("layers.10.layers.1.0", pc2, r1, r2, pad=1,
                   flags=nat.F_RELU_PRE | nat.F_RESIDUAL, res=cat, tag=TAG_TAIL_RES)
            pc = self._conv("layers.11.0", lambda: pack_conv(self.pack, conv_weight(sd, "layers.11.0"), tail_cmap, tail_span,
                                                             bias=sd["layers.11.0.bias"], precise=self.precise))
            b.conv("layers.11.0", pc, r2, out_buf, flags=nat.F_OUT_RGB8, f=(Y_RANGE[0], Y_RANGE[1], 0, 0), Co=3)
        ops, bufs = b.finish()
        return ops, bufs, in_buf, out_buf, b.names
