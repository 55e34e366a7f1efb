"""DDColor as a plan over the HIP ops: weight folding / packing and op emission for one input size.

PARITY UNPINNED -- the architecture is the published one (piddnad/DDColor; see oracle/ddcolor.py for what is known and what is
not): ConvNeXt-L encoder, three UnetBlockWide-style decoder stages + a x4 pixel shuffle, MultiScaleColorDecoder (100 queries,
9 layers), refine conv.  State-dict keys as in the public checkpoints (vsdeoldify_amd/synth.py:ddcolor_state_dict_spec).

How the graph maps onto the existing kernels:
  * every Linear / 1x1 / 2x2 s2 / 4x4 s4 / 3x3 conv is `conv_pipe_kernel` (tokens are pixels of a 1 x 112 image); GELU, layer
    scale + residual, out-projection + residual are conv epilogues (GELU / AFFINE + RESIDUAL / RESIDUAL);
  * depthwise 7x7, channel LayerNorm, multi-head attention (head dim 32), PixelShuffle(4)+blur: csrc/ddcolor.hip;
  * position encodings never touch the GPU as such: K = (src + pos) Wk + bk = src Wk + (pos Wk + bk), and the bracket is a
    constant map per layer and size, computed here in fp32 and added through the RESIDUAL epilogue (same for the query embedding);
  * einsum(bqc,bchw->bqhw) is a 1x1 conv whose weight rows ARE the colour embeddings (HAVC_F_W_FROM_BUF).
"""
import math

import os

import numpy as np

from . import _native as nat
from .plan import PlanBuilder, View, WeightPack, bn_scale_shift, conv_weight, pack_conv, pad_to, pitch_for, to_np

DEPTHS, DIMS = (3, 3, 27, 3), (192, 384, 768, 1536)
TAG_STAGE2_PW1 = 2             # the 27 pwconv1 GEMMs (768 -> 3072 + GELU) of ConvNeXt stage 2: the launches a DDColor pass spends most time in
HIDDEN, HEADS, QUERIES, TOK = 256, 8, 100, 112            # TOK: tokens per frame incl. 12 pad rows (= Npad of the einsum conv)
MEAN = np.array([0.485, 0.456, 0.406], np.float32)
STD = np.array([0.229, 0.224, 0.225], np.float32)


def position_sine(h, w, num_pos_feats=HIDDEN // 2, temperature=10000.0):
    """PositionEmbeddingSine(normalize=True) -> [h * w, 2 * num_pos_feats] float32 (row-major pixels)."""
    y = np.arange(1, h + 1, dtype=np.float32)[:, None].repeat(w, 1)
    x = np.arange(1, w + 1, dtype=np.float32)[None, :].repeat(h, 0)
    eps, scale = np.float32(1e-6), np.float32(2 * math.pi)
    y = y / (y[-1:, :] + eps) * scale
    x = x / (x[:, -1:] + eps) * scale
    dim_t = np.arange(num_pos_feats, dtype=np.float32)
    dim_t = (np.float32(temperature) ** (2 * np.floor(dim_t / 2) / np.float32(num_pos_feats))).astype(np.float32)
    px, py = x[:, :, None] / dim_t, y[:, :, None] / dim_t
    px = np.stack((np.sin(px[:, :, 0::2]), np.cos(px[:, :, 1::2])), axis=3).reshape(h, w, -1)
    py = np.stack((np.sin(py[:, :, 0::2]), np.cos(py[:, :, 1::2])), axis=3).reshape(h, w, -1)
    return np.concatenate((py, px), axis=2).reshape(h * w, -1).astype(np.float32)


class DDColorGenerator:
    def __init__(self, state_dict, depths=DEPTHS, dec_layers=9, precision="fast"):
        """precision: "fast" = fp16 activations / MFMA operands, fp32 accumulation; "precise" (round 5) = fp32-class arithmetic like the wheel's torch
        graph (vsslib/vsmodels.py:353-363 hands it RGBH / RGBS frames; the network itself computes in fp32): hi / lo fp16 pairs, three-segment convs,
        fp32 depthwise conv / LayerNorm / GELU / attention (HAVC_F_PRECISE; csrc/precise2.hip), the tail folded into ONE fp32 projection kernel."""
        assert precision in ("fast", "precise")
        self.precise = precision == "precise"
        self.sd, self.depths, self.dec_layers = to_np(state_dict), tuple(depths), dec_layers
        self.pack, self._pc, self._vec = WeightPack(), {}, {}
        self._frozen = False
        self.fuse_tail = os.environ.get("HAVC_DD_FUSE_TAIL", "1") != "0" or self.precise      # A/B switch: einsum + refine folded into the last_shuf conv
        # precise: fold the projection into the last_shuf conv's epilogue as the fast plan does (round 6; 0 = round 5's form: the 4096-channel pair tensor is
        # stored and one fp32 kernel shuffles, blurs and projects it)
        self.fuse_proj_p = os.environ.get("HAVC_DD_PRECISE_FUSE_PROJ", "1") != "0"
        self.fuse_dwln = os.environ.get("HAVC_DD_FUSE_DWLN", "1") != "0"      # A/B switch: dwconv + LayerNorm as one kernel (precise: round 6, the widths whose fp32 weights fit the LDS)
        # encoder.norm{0,1,2} feed nothing but the decoder's BatchNorm + ReLU on the skip connection: LayerNorm -> BN -> ReLU as ONE LayerNorm launch with
        # the BN folded into its gamma / beta, written straight into the concat buffer (round 4: the separate affine pass was 1.2 ms per 64 frames)
        self.fuse_skip_norm = os.environ.get("HAVC_DD_FUSE_SKIPNORM", "1") != "0" and self.fuse_tail
        self.plan(64)
        self.blob = self.pack.blob()
        self._frozen = True

    # ---- cached packing ---------------------------------------------------------------------------------------------
    def _conv(self, key, fn):
        if key not in self._pc:
            assert not self._frozen, key
            self._pc[key] = fn()
        return self._pc[key]

    def _vecs(self, key, fn):
        if key not in self._vec:
            assert not self._frozen, key
            self._vec[key] = tuple(self.pack.add(np.asarray(v)) for v in fn())
        return self._vec[key]

    def _lin(self, key, x, W, bias=None, scale=None, shift=None):
        """Linear / conv weights [out, in(, kh, kw)] on view x."""
        W = np.asarray(W, np.float32)
        if W.ndim == 2:
            W = W[:, :, None, None]
        return self._conv(key, lambda: pack_conv(self.pack, W, x.cmap, x.span, bias=bias, scale=scale, shift=shift, precise=self.precise))

    def _ln(self, b, name, key, x, y, eps):
        g, be = self._vecs(key, lambda: (self.sd[key + ".weight"].astype(np.float32), self.sd[key + ".bias"].astype(np.float32)))
        b.layernorm(name, x, y, g, be, eps)
        return y

    # ---- the plan -------------------------------------------------------------------------------------------------------
    def plan(self, S):
        assert S % 32 == 0
        sd, b = self.sd, PlanBuilder(precise=self.precise)
        consts = []                                            # (buffer id, float32 array [rows, channels], row pitch in channels)
        in_buf = b.buf(S * S * 3, 1)
        x0 = b.tensor(S, S, 3)
        if self.precise or self.fuse_tail:
            # the refine conv's image term reads the normalised image from a compact 8-channel tensor of its own (the fused tail never materialises the
            # logits, so the 112-channel coarse buffer would hold nothing else: 16 contiguous bytes per pixel instead of 6 bytes at a 256-byte pitch)
            img_view = b.tensor(S, S, 3)
            coarse_buf, coarse_pitch = img_view.buf, img_view.cpitch
            b.prep_ddcolor("prep", in_buf, S, x0, img_view)
        else:
            coarse_pitch = pitch_for(112)
            coarse_buf = b.buf(S * S * coarse_pitch, 2, zero_init=True)      # channels 0-2 image, 8-107 logits (108-111 pad-token junk)
            b.prep_ddcolor("prep", in_buf, S, x0, View(coarse_buf, 0, coarse_pitch, S, S, 3, 8))

        # ---- ConvNeXt encoder ----
        e = "encoder.arch"
        feats, x = [], x0
        for i in range(4):
            d = f"{e}.downsample_layers.{i}"
            c = DIMS[i]
            if i == 0:
                pc = self._lin(d + ".0", x, sd[d + ".0.weight"], bias=sd[d + ".0.bias"])
                t = b.tensor(x.H // 4, x.W // 4, c)
                b.conv(d + ".0", pc, x, t, stride=4)
                x = self._ln(b, d + ".1", d + ".1", t, b.tensor(t.H, t.W, c), 1e-6)
            else:
                n = self._ln(b, d + ".0", d + ".0", x, b.tensor(x.H, x.W, x.C), 1e-6)
                pc = self._lin(d + ".1", n, sd[d + ".1.weight"], bias=sd[d + ".1.bias"])
                x = b.tensor(n.H // 2, n.W // 2, c)
                b.conv(d + ".1", pc, n, x, stride=2)
            # two activation buffers ping-pong through the blocks of a stage, the scratch tensors are shared
            dbuf, nbuf, hbuf, alt = b.tensor(x.H, x.W, c), b.tensor(x.H, x.W, c), b.tensor(x.H, x.W, 4 * c), b.tensor(x.H, x.W, c)
            for j in range(self.depths[i]):
                p = f"{e}.stages.{i}.{j}"
                wdw, bdw = self._vecs(p + ".dwconv", lambda p=p, x=x: (
                    self._dw_pack(sd[p + ".dwconv.weight"], x.span, np.float32 if self.precise else np.float16), sd[p + ".dwconv.bias"].astype(np.float32)))
                if self.fuse_dwln and c in ((192, 384, 768) if self.precise else (64, 192, 384, 768, 1536)):      # channel counts the fused kernel is instantiated for
                    g, be = self._vecs(p + ".norm", lambda p=p: (sd[p + ".norm.weight"].astype(np.float32), sd[p + ".norm.bias"].astype(np.float32)))
                    b.dwconv7_ln(p + ".dwconv+norm", x, nbuf, wdw, bdw, x.span, g, be, 1e-6)
                else:
                    b.dwconv7(p + ".dwconv", x, dbuf, wdw, bdw, x.span)
                    self._ln(b, p + ".norm", p + ".norm", dbuf, nbuf, 1e-6)
                pc1 = self._lin(p + ".pwconv1", nbuf, sd[p + ".pwconv1.weight"], bias=sd[p + ".pwconv1.bias"])
                b.conv(p + ".pwconv1", pc1, nbuf, hbuf, flags=nat.F_GELU, tag=TAG_STAGE2_PW1 if i == 2 else None)
                pc2 = self._lin(p + ".pwconv2", hbuf, sd[p + ".pwconv2.weight"], bias=sd[p + ".pwconv2.bias"],
                                scale=sd[p + ".gamma"].astype(np.float32), shift=np.zeros(c, np.float32))
                b.conv(p + ".pwconv2", pc2, hbuf, alt, flags=nat.F_AFFINE | nat.F_RESIDUAL, res=x)
                x, alt = alt, x
            if self.fuse_skip_norm and i < 3:
                feats.append(x)                                # the raw stage output: normalised inside the decoder stage that consumes it
            else:
                feats.append(self._ln(b, f"{e}.norm{i}", f"{e}.norm{i}", x, b.tensor(x.H, x.W, c), 1e-6))

        # ---- decoder: three UnetBlockWide stages (deoldify/unet.py:170-205 family) ----
        outs, up = [], feats[3]
        for li in range(3):
            p, skip = f"decoder.layers.{li}", feats[2 - li]

            def make_shuf(p=p, up=up):
                s, sh = bn_scale_shift(sd, p + ".shuf.conv.1")
                return pack_conv(self.pack, conv_weight(sd, p + ".shuf.conv.0") * s[:, None, None, None], up.cmap, up.span, bias=sh, pixshuf=True,
                                 precise=self.precise)
            pc = self._conv(p + ".shuf", make_shuf)
            up_c = pc.Cout // 4
            ps = b.tensor(2 * up.H, 2 * up.W, up_c)
            b.conv(p + ".shuf", pc, up, ps, flags=nat.F_RELU_PRE | nat.F_OUT_PIXSHUF)
            ups, sks = pad_to(up_c, 8), pad_to(skip.C, 8)
            cat_pitch = pitch_for(ups + sks) * b.pm
            cat_buf = b.buf(skip.H * skip.W * cat_pitch, 2, zero_init=False)
            b.blur_resize(p + ".blur", ps, View(cat_buf, 0, cat_pitch, skip.H, skip.W, up_c, ups))
            skip_view = View(cat_buf, ups, cat_pitch, skip.H, skip.W, skip.C, sks)
            if self.fuse_skip_norm:
                nk = f"{e}.norm{2 - li}"

                def make_norm_bn(p=p, nk=nk):
                    s_bn, sh_bn = bn_scale_shift(sd, p + ".bn")
                    return ((sd[nk + ".weight"].astype(np.float32) * s_bn).astype(np.float32),
                            (sd[nk + ".bias"].astype(np.float32) * s_bn + sh_bn).astype(np.float32))
                g, be = self._vecs(nk + "+" + p + ".bn", make_norm_bn)
                b.layernorm(nk + "+bn", skip, skip_view, g, be, 1e-6, relu=True)
            else:
                so, sho = self._vecs(p + ".bn", lambda p=p: bn_scale_shift(sd, p + ".bn"))
                b.affine(p + ".bn", skip, skip_view, so, sho, relu=True)
            cat = View(cat_buf, 0, cat_pitch, skip.H, skip.W, up_c + skip.C, ups + sks, np.concatenate([np.arange(up_c), ups + np.arange(skip.C)]))

            def make_conv(p=p, cat=cat):
                s, sh = bn_scale_shift(sd, p + ".conv.2")
                return pack_conv(self.pack, conv_weight(sd, p + ".conv.0"), cat.cmap, cat.span, scale=s, shift=sh, precise=self.precise)
            pcc = self._conv(p + ".conv", make_conv)
            up = b.tensor(cat.H, cat.W, pcc.Cout)
            b.conv(p + ".conv", pcc, cat, up, pad=1, flags=nat.F_RELU_PRE | nat.F_AFFINE)
            outs.append(up)
        # last_shuf: 1x1 conv (+BN) -> ReLU -> PixelShuffle(4) -> blur; rows re-ordered to (dy*4+dx)*256 + c for the shuffle kernel
        p = "decoder.last_shuf"

        def make_last(up=up):
            s, sh = bn_scale_shift(sd, p + ".conv.1")
            W = conv_weight(sd, p + ".conv.0") * s[:, None, None, None]
            cps = W.shape[0] // 16
            perm = (np.arange(cps)[None, :] * 16 + np.arange(16)[:, None]).reshape(-1)
            return pack_conv(self.pack, W[perm], up.cmap, up.span, bias=sh[perm], precise=self.precise)
        pcl = self._conv(p, make_last)
        last_in = up
        if self.precise and not self.fuse_proj_p:
            t4 = b.tensor(up.H, up.W, pcl.Cout)
            b.conv(p + ".conv", pcl, up, t4, flags=nat.F_RELU_PRE)
        elif not self.fuse_tail:
            t4 = b.tensor(up.H, up.W, pcl.Cout)
            b.conv(p + ".conv", pcl, up, t4, flags=nat.F_RELU_PRE)
            img_feat = b.tensor(S, S, pcl.Cout // 16)
            b.pixshuf4_blur(p + ".shuf+blur", t4, img_feat)

        # ---- colour decoder ----
        d = "decoder.color_decoder"
        E = HIDDEN
        qpos = sd[d + ".query_embed.weight"].astype(np.float32)

        def tok(C):
            return b.tensor(1, TOK, C)

        def const_tokens(arr):                                     # [<= TOK, C] float32 -> constant token buffer
            v = tok(arr.shape[1])
            consts.append((v.buf, arr, v.cpitch, TOK))
            return v
        src = []
        for i, f in enumerate(outs):
            k = f"{d}.input_proj.{i}"
            pc = self._lin(k, f, sd[k + ".weight"], bias=sd[k + ".bias"].astype(np.float32) + sd[d + ".level_embed.weight"][i].astype(np.float32))
            s_i = b.tensor(f.H, f.W, E)
            b.conv(k, pc, f, s_i)
            src.append(s_i)
        tgt = const_tokens(sd[d + ".query_feat.weight"].astype(np.float32))
        scale = 1.0 / math.sqrt(E // HEADS)
        # K / V projections of the cross attentions: they read only the projected feature level (not the query chain), and the layers i, i + 3, i + 6
        # share a level -- ONE GEMM per level with the layers' K | V weights stacked along N (round 4: 9 launches of N = 512 -> 3 of N = 1536, off the
        # sequential query chain); layer i then reads its K / V at channel offset (i // 3) * 2 E.  Same dot products: same bytes.
        kv_level = []
        for lv in range(min(3, self.dec_layers)):
            s_lv = src[lv]
            pos = position_sine(s_lv.H, s_lv.W)
            Ws, cs = [], []
            for i in range(lv, self.dec_layers, 3):
                c = f"{d}.transformer_cross_attention_layers.{i}"
                Wi, bi = sd[c + ".multihead_attn.in_proj_weight"].astype(np.float32), sd[c + ".multihead_attn.in_proj_bias"].astype(np.float32)
                Ws.append(Wi[E:])
                cs.append(np.concatenate([pos @ Wi[E:2 * E].T + bi[E:2 * E], np.broadcast_to(bi[2 * E:], (pos.shape[0], E))], axis=1))
            Wcat, carr = np.concatenate(Ws, 0), np.concatenate(cs, 1)
            ckv = b.tensor(s_lv.H, s_lv.W, carr.shape[1])
            consts.append((ckv.buf, carr, ckv.cpitch, s_lv.H * s_lv.W))
            kv = b.tensor(s_lv.H, s_lv.W, carr.shape[1])
            name = f"{d}.cross_kv.level{lv}"
            b.conv(name, self._lin(name, s_lv, Wcat), s_lv, kv, flags=nat.F_RESIDUAL, res=ckv)
            kv_level.append(kv)
        for i in range(self.dec_layers):
            lv = i % 3
            s_lv = src[lv]
            # cross attention: Q from the queries, K / V from the feature map (position term folded into a constant map)
            c = f"{d}.transformer_cross_attention_layers.{i}"
            Wi, bi = sd[c + ".multihead_attn.in_proj_weight"].astype(np.float32), sd[c + ".multihead_attn.in_proj_bias"].astype(np.float32)
            cq = const_tokens(qpos @ Wi[:E].T + bi[:E])
            q = tok(E)
            b.conv(c + ".q", self._lin(c + ".q", tgt, Wi[:E]), tgt, q, flags=nat.F_RESIDUAL, res=cq)
            a = tok(E)
            b.mha(c + ".attn", q, kv_level[lv], (i // 3) * 2 * E, (i // 3) * 2 * E + E, a, HEADS, QUERIES, s_lv.H * s_lv.W, scale)
            t1 = tok(E)
            b.conv(c + ".out", self._lin(c + ".out", a, sd[c + ".multihead_attn.out_proj.weight"], bias=sd[c + ".multihead_attn.out_proj.bias"]),
                   a, t1, flags=nat.F_RESIDUAL, res=tgt)
            tgt = self._ln(b, c + ".norm", c + ".norm", t1, tok(E), 1e-5)
            # self attention among the queries
            s_ = f"{d}.transformer_self_attention_layers.{i}"
            Wi, bi = sd[s_ + ".self_attn.in_proj_weight"].astype(np.float32), sd[s_ + ".self_attn.in_proj_bias"].astype(np.float32)
            cqkv = const_tokens(np.concatenate([qpos @ Wi[:E].T + bi[:E], qpos @ Wi[E:2 * E].T + bi[E:2 * E],
                                                np.broadcast_to(bi[2 * E:], (QUERIES, E))], axis=1))
            qkv = tok(3 * E)
            b.conv(s_ + ".qkv", self._lin(s_ + ".qkv", tgt, Wi), tgt, qkv, flags=nat.F_RESIDUAL, res=cqkv)
            a = tok(E)
            b.mha(s_ + ".attn", View(qkv.buf, 0, qkv.cpitch, 1, TOK, E, E), qkv, E, 2 * E, a, HEADS, QUERIES, QUERIES, scale)
            t1 = tok(E)
            b.conv(s_ + ".out", self._lin(s_ + ".out", a, sd[s_ + ".self_attn.out_proj.weight"], bias=sd[s_ + ".self_attn.out_proj.bias"]),
                   a, t1, flags=nat.F_RESIDUAL, res=tgt)
            tgt = self._ln(b, s_ + ".norm", s_ + ".norm", t1, tok(E), 1e-5)
            # FFN
            f_ = f"{d}.transformer_ffn_layers.{i}"
            h = tok(2048)
            b.conv(f_ + ".linear1", self._lin(f_ + ".linear1", tgt, sd[f_ + ".linear1.weight"], bias=sd[f_ + ".linear1.bias"]), tgt, h, flags=nat.F_RELU_PRE)
            t1 = tok(E)
            b.conv(f_ + ".linear2", self._lin(f_ + ".linear2", h, sd[f_ + ".linear2.weight"], bias=sd[f_ + ".linear2.bias"]), h, t1,
                   flags=nat.F_RESIDUAL, res=tgt)
            tgt = self._ln(b, f_ + ".norm", f_ + ".norm", t1, tok(E), 1e-5)
        emb = self._ln(b, d + ".decoder_norm", d + ".decoder_norm", tgt, tok(E), 1e-5)
        for k in range(3):
            key = f"{d}.color_embed.layers.{k}"
            nxt = tok(E)
            b.conv(key, self._lin(key, emb, sd[key + ".weight"], bias=sd[key + ".bias"]), emb, nxt, flags=nat.F_RELU_PRE if k < 2 else 0)
            emb = nxt
        ab = b.tensor(S, S, 2)
        if self.fuse_tail:
            # einsum(bqc,bchw->bqhw) and the refine conv are linear per pixel and commute with the shuffle and the blur: fold the colour
            # embeddings and the refine rows into one 2 x 256 matrix per frame, apply it to every 256-channel sub-pixel group in the
            # epilogue of the last_shuf conv (its 4096-channel output, the shuffled / blurred 256-channel map and the 100 logit planes
            # are never stored), then shuffle + blur the 2-channel result and add the image term of the refine conv.
            assert pcl.Cout == 16 * E and pcl.Npad == pcl.Cout

            def make_refine():
                Wr = conv_weight(sd, "refine_net.0.0").astype(np.float32).reshape(2, QUERIES + 3)
                rq = np.zeros((2, 104), np.float32)
                rq[:, :QUERIES] = Wr[:, :QUERIES]
                return rq, np.ascontiguousarray(Wr[:, QUERIES:]), sd["refine_net.0.0.bias"].astype(np.float32)
            rq_off, rimg_off, rb_off = self._vecs("refine_net.0.0/fold", make_refine)
            m2 = b.buf(2 * E, 4)
            b.fold_queries(d + ".fold", emb, QUERIES, rq_off, 104, m2)
            if self.precise and self.fuse_proj_p:
                # the fast plan's form in fp32: the projection in the last_shuf conv's precise epilogue (HAVC_F_FUSE_PROJ with HAVC_F_PRECISE), then the
                # shuffle + blur of the fp32 2-channel map with the image term read from / the ab map written as pairs
                proj = b.buf(last_in.H * last_in.W * 16 * 2, 4)
                b.conv(p + ".conv+proj", pcl, last_in, proj, flags=nat.F_RELU_PRE | nat.F_FUSE_PROJ, proj=(m2, proj))
                b.shuf4_blur_ab("refine_net.0.0", proj, last_in.H, last_in.W, img_view, rimg_off, rb_off, ab,
                                flops=2 * S * S * (E * QUERIES + 2 * (QUERIES + 3)))
                ops, bufs = b.finish()
                return ops, bufs, in_buf, ab.buf, b.names, consts
            if self.precise:
                # the same fold in fp32: shuffle + blur of the 4096-channel pair tensor, the 2 x 256 projection, the image term and the bias in one kernel
                b.shuf4_blur_proj("refine_net.0.0", t4, m2, img_view, rimg_off, rb_off, ab, flops=2 * S * S * (E * QUERIES + 2 * (QUERIES + 3)))
                ops, bufs = b.finish()
                return ops, bufs, in_buf, ab.buf, b.names, consts
            proj = b.buf(last_in.H * last_in.W * 16 * 2, 4)
            b.conv(p + ".conv+proj", pcl, last_in, proj, flags=nat.F_RELU_PRE | nat.F_FUSE_PROJ, proj=(m2, proj))
            b.shuf4_blur_ab("refine_net.0.0", proj, last_in.H, last_in.W, View(coarse_buf, 0, coarse_pitch, S, S, 3, 8), rimg_off, rb_off, ab,
                            flops=2 * S * S * (E * QUERIES + 2 * (QUERIES + 3)))      # algorithmic work of the einsum + refine conv it stands for
        else:
            logits = View(coarse_buf, 8, coarse_pitch, S, S, QUERIES, 104)
            b.conv_dyn(d + ".einsum", img_feat, emb, logits, QUERIES)
            # ---- refine: spectral 1x1 conv on cat[logits, normalised image] ----
            coarse = View(coarse_buf, 0, coarse_pitch, S, S, QUERIES + 3, 112, np.concatenate([8 + np.arange(QUERIES), np.arange(3)]))
            pcr = self._conv("refine_net.0.0", lambda: pack_conv(self.pack, conv_weight(sd, "refine_net.0.0"), coarse.cmap, coarse.span,
                                                                 bias=sd["refine_net.0.0.bias"]))
            b.conv("refine_net.0.0", pcr, coarse, ab)
        ops, bufs = b.finish()
        return ops, bufs, in_buf, ab.buf, b.names, consts

    @staticmethod
    def _dw_pack(W, pitch, dtype=np.float16):
        """[C, 1, 7, 7] -> fp16 (precise: fp32) [49][pitch] (tap-major, channels contiguous)."""
        C = W.shape[0]
        out = np.zeros((49, pitch), dtype)
        out[:, :C] = W.reshape(C, 49).T.astype(dtype)
        return out
