"""ColorMNet network on the MI355X (BASELINE configs[4]): weight packing, plan emission and the four entry points the
per-frame step drives — encode_key, encode_value, segment, short_term_attn — with the reference's argument lists and tensor shapes
(/root/reference/vsdeoldify/colormnet/model/network.py:52-145; driven by inference/inference_core.py, here colormnet_core.InferenceCore).

Topology restated from the reference (this module only decides WHAT runs; all arithmetic is in the HIP kernels):
  KeyEncoder_DINOv2_v6   model/modules.py:158-196  ResNet50 trunk (model/resnet.py:124-166) + Segmentor (:211-247: DINOv2 ViT-S/14 blocks
                         8-11, 1x1 conv + BN + ReLU, bilinear 1/14 -> 1/16) + three Fuse blocks (:370-398) with CrossChannelAttention (:286-331)
  KeyProjection          model/modules.py:213-231  (the three 3x3 convs merged into one GEMM, activations in the layout conversion)
  ValueEncoder           model/modules.py:105-156  ResNet18 on image + object plane + others plane, FeatureFusionBlock (:22-41) with CBAM
                         (model/cbam.py), HiddenReinforcer (:80-103)
  Decoder                model/modules.py:233-271  FeatureFusionBlock, UpsampleBlock x2 (:197-211), pred, HiddenUpdater (:44-78)
  LocalGatedPropagation  model/attention.py:712-860 (the local attention itself: csrc/colormnet.hip) + DWConv2d (model/basic.py:75-94) + Linear
The DINOv2 backbone is torch.hub content, not part of the reference tree: built from the published architecture (oracle/dinov2.py),
PARITY UNPINNED; everything else is pinned to vectors from the executed reference (tests/test_colormnet_net.py).

Execution: ONE plan per padded frame size.  Its ops are grouped in slices that run with different batch counts — image features with one
frame, per-object features (the two ab planes are the "objects": colormnet_render.py:239-241) with one frame per object — and exchange the
reference's fp32 NCHW tensors (keys, values, hidden state, ab planes) with the caller through buffers BOUND to torch device tensors
(havc_net_bind): torch is device memory + bookkeeping, nothing is computed by it.  torch and the library share ONE HIP stream
(torch.cuda.ExternalStream(havc_get_stream)), so no host synchronisation is needed anywhere in a frame.  No CPU fallback.
"""
import math
import os

import numpy as np

from . import _native as nat
from .plan import PlanBuilder, View, WeightPack, bn_scale_shift, pack_conv, pad_to, pitch_for, to_np

HEADS_DINO, DINO_PATCH, DINO_EPS = 6, 14, 1e-6
HEADS_CCA = 8
MAX_DIS = 7


def _conv_out(n, k, s, p):
    return (n + 2 * p - (k - 1) - 1) // s + 1


def chan_attn_splits(P, heads, cc):
    """pixel splits of the channel-attention Gram kernel (mirror of chan_attn_splits in csrc/colormnet_net.hip: sizes its scratch buffers)"""
    tiles = heads * ((cc + 63) // 64) ** 2
    return max(1, min((512 + tiles - 1) // tiles, (P + 127) // 128))


def _dino_pos(pos_embed, h0, w0):
    """dinov2 interpolate_pos_encoding (bicubic, scale factors (h0 + 0.1) / M, (w0 + 0.1) / M, antialias off) -> (cls [D], patches [h0*w0, D]).
    Host-side constant per frame size; torch's own bicubic is the arithmetic the hub model would run."""
    import torch
    import torch.nn.functional as F
    pos = torch.from_numpy(np.asarray(pos_embed, np.float32))
    n = pos.shape[1] - 1
    m = int(math.sqrt(n))
    d = pos.shape[-1]
    cls, patch = pos[0, 0], pos[:, 1:]
    if not (h0 == m and w0 == m):
        grid = patch.reshape(1, m, m, d).permute(0, 3, 1, 2)
        grid = F.interpolate(grid, mode="bicubic", antialias=False, scale_factor=(float(h0 + 0.1) / m, float(w0 + 0.1) / m))
        assert grid.shape[-2:] == (h0, w0)
        patch = grid.permute(0, 2, 3, 1).reshape(1, h0 * w0, d)
    return cls.numpy(), patch[0].numpy()


def splitk_for(rows, Npad, Kc):
    """split-K count of a conv with `rows` GEMM rows (pixels x frames of its slice): a frame of ColorMNet is a chain of launches with 4 - 100
    output tiles for 256 CUs and K up to 14 400 in series; cutting K puts ~256 blocks in flight.  Part of the PLAN (deterministic in the
    shape), so every tile configuration of the autotuner produces the same bytes (include/havc_mi355.h HAVC_F_SPLITK)."""
    stages = Kc // 8
    tiles = ((rows + 127) // 128) * ((Npad + 127) // 128)
    if tiles >= 128 or stages < 24:
        return 0
    n = min(15, 256 // tiles, stages // 4)
    return n if n >= 2 else 0


class _Builder(PlanBuilder):
    """PlanBuilder that knows the batch count of the slice being emitted and adds the split-K count to its convs"""
    slice_batch = 1
    auto_split = True

    def conv(self, name, pc, x, y, stride=1, pad=0, flags=0, **kw):
        if self.auto_split and not (flags & (nat.F_W_FROM_BUF | nat.F_PS_BLUR | nat.F_FUSE_PROJ | nat.F_OUT_RGB8)):
            Ho, Wo = _conv_out(x.H, pc.kh, stride, pad), _conv_out(x.W, pc.kw, stride, pad)
            flags |= nat.F_SPLITK(splitk_for(Ho * Wo * self.slice_batch, pc.Npad, pc.Kc))
        return super().conv(name, pc, x, y, stride=stride, pad=pad, flags=flags, **kw)


class ColorMNetPlan:
    """packs a ColorMNet state dict once; emits the plan for a padded frame size (H, W multiples of 112)"""

    def __init__(self, state_dict):
        self.sd = to_np(state_dict)
        sd = self.sd
        self.key_dim = sd["key_proj.key_proj.weight"].shape[0]
        self.value_dim = sd["value_encoder.fuser.block2.conv2.weight"].shape[0]
        self.hidden_dim = sd["decoder.hidden_update.transform.weight"].shape[0] // 3 if "decoder.hidden_update.transform.weight" in sd else 0
        if self.hidden_dim <= 0:
            raise NotImplementedError("checkpoints without a hidden state (hidden_dim = 0) are not supported")
        self.pack, self._pc, self._vec = WeightPack(), {}, {}
        self._frozen = False
        self.plan(112, 112)
        self.blob = self.pack.blob()
        self._frozen = True

    # ---- cached packing ----
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

    def _w(self, k):
        return self.sd[k].astype(np.float32)

    def _plain(self, key, x, W, bias=None, scale=None, shift=None):
        W = np.asarray(W, np.float32)
        if W.ndim == 2:
            W = W[:, :, None, None]
        return self._conv(key, lambda: pack_conv(self.pack, W, x.cmap, x.span, bias=bias, scale=scale, shift=shift))

    def _bnconv(self, wkey, bnkey, x):
        """conv (no bias) -> BatchNorm folded into weight / bias"""
        def make():
            s, sh = bn_scale_shift(self.sd, bnkey)
            return pack_conv(self.pack, self._w(wkey + ".weight") * s[:, None, None, None], x.cmap, x.span, bias=sh)
        return self._conv(wkey, make)

    def _cv(self, b, p, x, y, pad=0, stride=1, flags=0, res=None):
        """nn.Conv2d with bias, parameters at prefix p"""
        pc = self._plain(p, x, self._w(p + ".weight"), bias=self.sd.get(p + ".bias"))
        b.conv(p, pc, x, y, stride=stride, pad=pad, flags=flags, res=res)
        return y

    def _ln(self, b, p, x, y, eps, relu=False):
        g, be = self._vecs(p, lambda: (self._w(p + ".weight"), self._w(p + ".bias")))
        b.layernorm(p, x, y, g, be, eps, relu)
        return y

    def _dw(self, b, name, keys, x, y, k):
        """depthwise k x k over the concatenation of the parameter sets `keys` (weights [C,1,k,k], optional biases)"""
        def make():
            W = np.concatenate([self._w(q + ".weight") for q in keys], 0)
            C = W.shape[0]
            out = np.zeros((k * k, x.span), np.float16)
            out[:, :C] = W.reshape(C, k * k).T.astype(np.float16)
            if keys[0] + ".bias" in self.sd:
                bias = np.zeros(x.span, np.float32)
                bias[:C] = np.concatenate([self._w(q + ".bias") for q in keys])
                return out, bias
            return (out,)
        offs = self._vecs(name, make)
        b.dwconv(name, x, y, offs[0], offs[1] if len(offs) > 1 else -1, x.span, k)
        return y

    # ---- ResNet trunks ----
    def _trunk(self, b, p, x, kind, names):
        sd = self.sd
        H2, W2 = _conv_out(x.H, 7, 2, 3), _conv_out(x.W, 7, 2, 3)
        e2 = b.tensor(H2, W2, 64)
        # conv1 -> bn1 -> ReLU -> maxpool (the value encoder pools before the ReLU, modules.py:139-142: the same thing, both are monotone)
        b.conv(p + ".conv1", self._bnconv(p + ".conv1", p + ".bn1", x), x, e2, stride=2, pad=3, flags=nat.F_RELU_PRE)
        x = b.tensor(_conv_out(H2, 3, 2, 1), _conv_out(W2, 3, 2, 1), 64)
        b.maxpool(p + ".maxpool", e2, x)
        outs = []
        for li, name in enumerate(names):
            planes, bi = 64 * 2 ** li, 0
            while f"{p}.{name}.{bi}.conv1.weight" in sd:
                q = f"{p}.{name}.{bi}"
                stride = 2 if (li > 0 and bi == 0) else 1
                Ho, Wo = _conv_out(x.H, 3, stride, 1), _conv_out(x.W, 3, stride, 1)
                idt = x
                if q + ".downsample.0.weight" in sd:
                    pc = self._bnconv(q + ".downsample.0", q + ".downsample.1", x)
                    idt = b.tensor(Ho, Wo, pc.Cout)
                    b.conv(q + ".downsample", pc, x, idt, stride=stride)
                if kind == "bottleneck":
                    t1 = b.tensor(x.H, x.W, planes)
                    b.conv(q + ".conv1", self._bnconv(q + ".conv1", q + ".bn1", x), x, t1, flags=nat.F_RELU_PRE)
                    t2 = b.tensor(Ho, Wo, planes)
                    b.conv(q + ".conv2", self._bnconv(q + ".conv2", q + ".bn2", t1), t1, t2, stride=stride, pad=1, flags=nat.F_RELU_PRE)
                    y = b.tensor(Ho, Wo, planes * 4)
                    b.conv(q + ".conv3", self._bnconv(q + ".conv3", q + ".bn3", t2), t2, y, flags=nat.F_RESIDUAL | nat.F_RELU_POST, res=idt)
                else:
                    t1 = b.tensor(Ho, Wo, planes)
                    b.conv(q + ".conv1", self._bnconv(q + ".conv1", q + ".bn1", x), x, t1, stride=stride, pad=1, flags=nat.F_RELU_PRE)
                    y = b.tensor(Ho, Wo, planes)
                    b.conv(q + ".conv2", self._bnconv(q + ".conv2", q + ".bn2", t1), t1, y, pad=1, flags=nat.F_RESIDUAL | nat.F_RELU_POST, res=idt)
                x = y
                bi += 1
            outs.append(x)
        return outs

    # ---- DINOv2 ViT-S/14 -> [h0*w0 (+1 class row)] x 1536 (blocks 8-11 after the final norm) ----
    def _dino(self, b, x0, consts):
        sd, p = self.sd, "key_encoder.network2.backbone"
        D = sd[p + ".cls_token"].shape[-1]
        h0, w0 = x0.H // DINO_PATCH, x0.W // DINO_PATCH
        T = h0 * w0 + 1                                              # the class token is the LAST row here (attention does not care)
        depth = 1 + max(int(k.split(".")[4]) for k in sd if k.startswith(p + ".blocks."))
        cls_pos, patch_pos = _dino_pos(sd[p + ".pos_embed"], h0, w0)

        def tok(C, zero=False):
            pc = pitch_for(pad_to(C, 8))
            return View(b.buf(T * pc, 2, zero), 0, pc, 1, T, C, pad_to(C, 8))
        xt = tok(D)
        posb = tok(D)
        rows = np.zeros((T, D), np.float32)
        rows[:T - 1] = patch_pos
        consts.append((posb.buf, rows, posb.cpitch, T))
        first = np.zeros((T, D), np.float32)
        first[T - 1] = sd[p + ".cls_token"].reshape(-1).astype(np.float32) + cls_pos            # class row: written once, never overwritten
        consts.append((xt.buf, first, xt.cpitch, T))
        pe = self._plain(p + ".patch_embed.proj", x0, self._w(p + ".patch_embed.proj.weight"), bias=self._w(p + ".patch_embed.proj.bias"))
        b.conv(p + ".patch_embed", pe, x0, View(xt.buf, 0, xt.cpitch, h0, w0, D, xt.span), stride=DINO_PATCH, flags=nat.F_RESIDUAL,
               res=View(posb.buf, 0, posb.cpitch, h0, w0, D, posb.span))
        cat_pitch = pitch_for(4 * D)
        cat_buf = b.buf(T * cat_pitch, 2)
        n1, qkv, att, hid = tok(D), tok(3 * D), tok(D), tok(4 * D)
        x, y1, y2 = xt, tok(D), tok(D)                               # the initial token buffer is never written again (its class row is a constant)
        scale = (D // HEADS_DINO) ** -0.5
        taps = (8, 9, 10, 11)                                        # get_intermediate_layers(x, n=[8, 9, 10, 11]) (resnet.py:236)
        assert depth > max(taps), "the reference taps blocks 8-11 of the backbone"
        for i in range(max(taps) + 1):
            q = f"{p}.blocks.{i}"
            self._ln(b, q + ".norm1", x, n1, DINO_EPS)
            self._cv(b, q + ".attn.qkv", n1, qkv)
            b.mha64(q + ".attn", qkv, 0, D, 2 * D, att, HEADS_DINO, T, scale)
            pc = self._plain(q + ".attn.proj", att, self._w(q + ".attn.proj.weight"), bias=self._w(q + ".attn.proj.bias"),
                             scale=self._w(q + ".ls1.gamma"), shift=np.zeros(D, np.float32))
            b.conv(q + ".attn.proj", pc, att, y1, flags=nat.F_AFFINE | nat.F_RESIDUAL, res=x)
            self._ln(b, q + ".norm2", y1, n1, DINO_EPS)
            self._cv(b, q + ".mlp.fc1", n1, hid, flags=nat.F_GELU)
            pc = self._plain(q + ".mlp.fc2", hid, self._w(q + ".mlp.fc2.weight"), bias=self._w(q + ".mlp.fc2.bias"),
                             scale=self._w(q + ".ls2.gamma"), shift=np.zeros(D, np.float32))
            b.conv(q + ".mlp.fc2", pc, hid, y2, flags=nat.F_AFFINE | nat.F_RESIDUAL, res=y1)
            x = y2
            if i in taps:
                g_, be_ = self._vecs(p + ".norm", lambda: (self._w(p + ".norm.weight"), self._w(p + ".norm.bias")))
                b.layernorm(f"{p}.norm.{i}", x, View(cat_buf, taps.index(i) * D, cat_pitch, 1, T, D, D), g_, be_, DINO_EPS)
        return View(cat_buf, 0, cat_pitch, h0, w0, 4 * D, 4 * D)

    # ---- Fuse (resnet.py:370-398) ----
    def _fuse(self, b, p, enc_in, dnc, out):
        sd = self.sd
        dim = dnc.C
        E = b.tensor(dnc.H, dnc.W, dim)
        self._cv(b, p + ".encode_enc", enc_in, E, pad=1)
        n1 = self._ln(b, p + ".norm1", E, b.tensor(dnc.H, dnc.W, dim), 1e-6)
        n2 = self._ln(b, p + ".norm2", dnc, b.tensor(dnc.H, dnc.W, dim), 1e-6)
        c = p + ".crossattn"
        q0 = b.tensor(dnc.H, dnc.W, 2 * dim)
        self._cv(b, c + ".to_q", n1, q0)
        q = self._dw(b, c + ".to_q_dw", [c + ".to_q_dw"], q0, b.tensor(dnc.H, dnc.W, 2 * dim), 3)
        kv0 = b.tensor(dnc.H, dnc.W, 4 * dim)
        pc = self._plain(c + ".to_kv", n2, np.concatenate([self._w(c + ".to_k.weight"), self._w(c + ".to_v.weight")], 0),
                         bias=np.concatenate([self._w(c + ".to_k.bias"), self._w(c + ".to_v.bias")]))
        b.conv(c + ".to_kv", pc, n2, kv0)
        kv = self._dw(b, c + ".to_kv_dw", [c + ".to_k_dw", c + ".to_v_dw"], kv0, b.tensor(dnc.H, dnc.W, 4 * dim), 3)
        k = View(kv.buf, 0, kv.cpitch, kv.H, kv.W, 2 * dim, 2 * dim)
        v = View(kv.buf, 2 * dim, kv.cpitch, kv.H, kv.W, 2 * dim, 2 * dim)
        heads, cc, P = HEADS_CCA, 2 * dim // HEADS_CCA, dnc.H * dnc.W
        S = chan_attn_splits(P, heads, cc)
        wbuf = b.buf(2 * dim * 2 * dim, 2, zero_init=True)
        part_g, part_n = b.buf(heads * S * cc * cc, 4), b.buf(S * 2 * 2 * dim, 4)
        temp, = self._vecs(c + ".temperature", lambda: (self._w(c + ".temperature").reshape(-1),))
        b.chan_attn(c + ".attn", q, k, heads, temp, wbuf, 2 * dim // 8, part_g, part_n)
        o = b.tensor(dnc.H, dnc.W, 2 * dim)
        b.conv_dyn(c + ".attn@v", v, View(wbuf, 0, 2 * dim, 1, 2 * dim, 2 * dim, 2 * dim), o, 2 * dim)
        t = b.tensor(dnc.H, dnc.W, dim)
        self._cv(b, c + ".to_out.0", o, t, flags=nat.F_RESIDUAL, res=E)
        return self._ln(b, p + ".norm3", t, out, 1e-6, relu=True)

    # ---- GroupResBlock (group_modules.py:38-58): x raw, xr = relu(x) ----
    def _resblock(self, b, p, x, xr, out):
        mid = self.sd[p + ".conv1.weight"].shape[0]
        t1 = b.tensor(x.H, x.W, mid)
        self._cv(b, p + ".conv1", xr, t1, pad=1, flags=nat.F_RELU_POST)
        if p + ".downsample.weight" in self.sd:
            t2 = b.tensor(x.H, x.W, mid)
            self._cv(b, p + ".conv2", t1, t2, pad=1)
            self._cv(b, p + ".downsample", x, out, pad=1, flags=nat.F_RESIDUAL, res=t2)
        else:
            self._cv(b, p + ".conv2", t1, out, pad=1, flags=nat.F_RESIDUAL, res=x)
        return out

    def _fusion(self, b, p, cat, catr, out):
        """FeatureFusionBlock (modules.py:22-41) on the concatenated [x, g] (raw and rectified) -> out"""
        mid = self.sd[p + ".block1.conv1.weight"].shape[0]
        g1 = self._resblock(b, p + ".block1", cat, catr, b.tensor(cat.H, cat.W, mid))
        a = p + ".attention"

        def make():
            return (np.concatenate([self._w(a + ".ChannelGate.mlp.1.weight").reshape(-1), self._w(a + ".ChannelGate.mlp.1.bias"),
                                    self._w(a + ".ChannelGate.mlp.3.weight").reshape(-1), self._w(a + ".ChannelGate.mlp.3.bias"),
                                    self._w(a + ".SpatialGate.spatial.conv.weight").reshape(-1), self._w(a + ".SpatialGate.spatial.conv.bias")]),)
        woff, = self._vecs(a, make)
        g2, g2r = b.tensor(cat.H, cat.W, mid), b.tensor(cat.H, cat.W, mid)
        b.cbam(a, g1, g2, woff, b.buf(3 * mid, 4), b.buf(cat.H * cat.W * 2, 4), dual=g2r)       # gate buffer: scale | avg | max
        return self._resblock(b, p + ".block2", g2, g2r, out)

    # ---- the plan ----
    def plan(self, H, W, key_batch=1):
        """key_batch: frames per launch of the "key" slice.  encode_key does not depend on the memory, so the frames of a clip can go through
        the key encoder several at a time (ColorMNetNetwork.prefetch_keys); the split-K counts of that slice are chosen for key_batch frames."""
        assert H % 112 == 0 and W % 112 == 0, "frames are padded to multiples of 112 (inference_core.py:49)"
        sd, b = self.sd, _Builder()
        b.auto_split = os.environ.get("HAVC_CMN_SPLITK", "1") != "0"       # A/B switch (profiling)
        consts, sl, io = [], {}, {}
        CK, CV, HD = self.key_dim, self.value_dim, self.hidden_dim
        h16, w16, h8, w8, h4, w4 = H // 16, W // 16, H // 8, W // 8, H // 4, W // 4
        P16 = h16 * w16

        def fbuf(name, elems):                                      # fp32 buffer bound to a caller tensor at run time
            io[name] = b.buf(elems, 4)
            return io[name]

        def mark(name, first, batch):
            sl[name] = (first, len(b.ops) - first, batch)

        # ================= slice "key": encode_key, one frame (key_batch frames for the look-ahead plan) =================
        s0 = len(b.ops)
        b.slice_batch = key_batch
        x0 = b.tensor(H, W, 3)
        b.planar_in("frame", fbuf("image", 3 * H * W), 3, x0)
        f4, f8, f16 = self._trunk(b, "key_encoder", x0, "bottleneck", ("res2", "layer2", "layer3"))
        dino = self._dino(b, x0, consts)
        n2 = "key_encoder.network2"
        d14 = b.tensor(dino.H, dino.W, dino.C)
        b.conv(n2 + ".conv3", self._bnconv(n2 + ".conv3", n2 + ".bn3", dino), dino, d14, flags=nat.F_RELU_PRE)
        # bilinear to (int(h*14/16), int(w*14/16)) (resnet.py:242-244), then nn.Upsample x2 / x4 of THAT map (modules.py:192-193)
        assert (int(dino.H * 14 / 16), int(dino.W * 14 / 16)) == (h16, w16)
        d16, d8, d4 = b.tensor(h16, w16, dino.C), b.tensor(h8, w8, dino.C), b.tensor(h4, w4, dino.C)
        b.ew(n2 + ".interp", d14, d16, mode=1, ratio=(np.float32(dino.H) / np.float32(h16), np.float32(dino.W) / np.float32(w16)))
        b.ew("key_encoder.upsample2", d16, d8, mode=1, ratio=(0.5, 0.5))
        b.ew("key_encoder.upsample4", d16, d4, mode=1, ratio=(0.25, 0.25))
        g16, g8, g4 = b.tensor(h16, w16, 1024), b.tensor(h8, w8, 512), b.tensor(h4, w4, 256)
        io["g16"], io["g8"], io["g4"] = g16.buf, g8.buf, g4.buf
        self._fuse(b, "key_encoder.fuse1", d16, f16, g16)
        self._fuse(b, "key_encoder.fuse2", d8, f8, g8)
        self._fuse(b, "key_encoder.fuse3", d4, f4, g4)
        # KeyProjection: key | selection | shrinkage as ONE 3x3 conv (modules.py:213-231)
        kp = "key_proj"
        pc = self._plain(kp, g16, np.concatenate([self._w(kp + ".key_proj.weight"), self._w(kp + ".e_proj.weight"), self._w(kp + ".d_proj.weight")], 0),
                         bias=np.concatenate([self._w(kp + ".key_proj.bias"), self._w(kp + ".e_proj.bias"), self._w(kp + ".d_proj.bias")]))
        kpo = b.tensor(h16, w16, 2 * CK + 1)
        b.conv(kp, pc, g16, kpo, pad=1)
        b.planar_out(kp + ".key", kpo, 0, CK, fbuf("key", CK * P16), 0)
        b.planar_out(kp + ".selection", kpo, CK, CK, fbuf("selection", CK * P16), 2)
        b.planar_out(kp + ".shrinkage", kpo, 2 * CK, 1, fbuf("shrinkage", P16), 1)
        mark("key", s0, key_batch)

        # ================= slice "value": encode_value, one frame per object =================
        s0 = len(b.ops)
        b.slice_batch = 2
        v0 = b.tensor(H, W, 5)
        b.planar_in("value_in", fbuf("value_in", 5 * H * W), 5, v0)
        ve = "value_encoder"
        cv_pitch = pitch_for(1024 + 256)
        cv_buf = b.buf(P16 * cv_pitch, 2)
        # the last BasicBlock writes straight into the concat buffer: emit the trunk, then redirect its final conv
        r18 = self._trunk(b, ve, v0, "basic", ("layer1", "layer2", "layer3"))[-1]
        assert (r18.H, r18.W) == (h16, w16)                          # F.interpolate(g, f16.shape[2:]) is the identity (modules.py:147)
        last = b.ops[-1]
        last["dst"], last["dst_coff"], last["dst_cpitch"] = cv_buf, 1024, cv_pitch
        cat = View(cv_buf, 0, cv_pitch, h16, w16, 1280, 1280)
        b.ew(ve + ".distribute", g16, View(cv_buf, 0, cv_pitch, h16, w16, 1024, 1024), src_bcast=True)
        # both halves of the concatenation are outputs of a ReLU (Fuse :396, BasicBlock): relu(cat) == cat
        vh_pitch = pitch_for(CV + HD)
        vh_buf = b.buf(P16 * vh_pitch, 2)
        val = View(vh_buf, 0, vh_pitch, h16, w16, CV, CV)
        self._fusion(b, ve + ".fuser", cat, cat, val)
        b.planar_out(ve + ".value", val, 0, CV, fbuf("value", CV * P16), 0)
        mark("value", s0, 2)
        s0 = len(b.ops)
        hv = View(vh_buf, CV, vh_pitch, h16, w16, HD, HD)
        b.planar_in(ve + ".hidden_in", fbuf("hidden", HD * P16), HD, hv)
        vals = b.tensor(h16, w16, 3 * HD)
        self._cv(b, ve + ".hidden_reinforce.transform", View(vh_buf, 0, vh_pitch, h16, w16, CV + HD, CV + HD), vals, pad=1)
        b.gru(ve + ".hidden_reinforce", vals, io["hidden"], fbuf("hidden_out", HD * P16), HD)
        mark("value_hidden", s0, 2)

        # ================= slice "skip": the decoder's skip convs on the image features, one frame =================
        d = "decoder"
        s0 = len(b.ops)
        b.slice_batch = key_batch                                    # (like the key slice it depends on the frame alone: it runs ahead with it)
        skip8, skip4 = b.tensor(h8, w8, 512), b.tensor(h4, w4, 256)
        io["skip8"], io["skip4"] = skip8.buf, skip4.buf
        self._cv(b, d + ".up_16_8.skip_conv", g8, skip8, pad=1)
        self._cv(b, d + ".up_8_4.skip_conv", g4, skip4, pad=1)
        mark("skip", s0, key_batch)

        # ================= slice "segment": Decoder, one frame per object =================
        s0 = len(b.ops)
        b.slice_batch = 2
        dc_span = 1024 + CV + HD
        dc_pitch = pitch_for(dc_span)
        dc_buf, dcr_buf = b.buf(P16 * dc_pitch, 2), b.buf(P16 * dc_pitch, 2)
        dc = View(dc_buf, 0, dc_pitch, h16, w16, dc_span, dc_span)
        dcr = View(dcr_buf, 0, dc_pitch, h16, w16, dc_span, dc_span)
        # g16 of the frame for both objects, the readout and the hidden state (fp32 planar -> NHWC fp16), each also rectified (the fuser's first
        # ResBlock reads relu(x), its shortcut x): ONE launch (round 5; four before: distribute, readout_in, hidden_in, relu_in)
        b.cmn_decoder_in(d + ".input", g16, fbuf("readout", CV * P16), CV, io["hidden"], HD, dc, dcr_buf)
        G16 = self._fusion(b, d + ".fuser", dc, dcr, b.tensor(h16, w16, 512))
        u8, u8r = b.tensor(h8, w8, 512), b.tensor(h8, w8, 512)
        b.ew(d + ".up_16_8.up+skip", G16, u8, mode=1, ratio=(0.5, 0.5), res=skip8, res_bcast=True, dual=u8r)
        G8 = self._resblock(b, d + ".up_16_8.out_conv", u8, u8r, b.tensor(h8, w8, 256))
        u4, u4r = b.tensor(h4, w4, 256), b.tensor(h4, w4, 256)
        b.ew(d + ".up_8_4.up+skip", G8, u4, mode=1, ratio=(0.5, 0.5), res=skip4, res_bcast=True, dual=u4r)
        g4c_pitch = pitch_for(256 + 8)
        g4c_buf = b.buf(h4 * w4 * g4c_pitch, 2, zero_init=True)
        G4 = self._resblock(b, d + ".up_8_4.out_conv", u4, u4r, View(g4c_buf, 0, g4c_pitch, h4, w4, 256, 256))
        g4r = b.tensor(h4, w4, 256)
        b.ew(d + ".pred.relu", G4, g4r, relu=True)
        logit = View(g4c_buf, 256, g4c_pitch, h4, w4, 1, 8)
        self._cv(b, d + ".pred", g4r, logit, pad=1)
        full = b.tensor(H, W, 1)
        b.ew(d + ".logits_x4", logit, full, mode=1, ratio=(0.25, 0.25))
        b.planar_out(d + ".prob", full, 0, 1, fbuf("prob", H * W), 3)
        mark("segment", s0, 2)
        # ---- HiddenUpdater (modules.py:44-78) ----
        s0 = len(b.ops)
        hu = d + ".hidden_update"
        a16 = b.tensor(h16, w16, 256)
        self._cv(b, hu + ".g16_conv", G16, a16)
        d8 = b.tensor(h16, w16, 256)
        b.ew(hu + ".g8_area", G8, d8, mode=2, factor=2)
        a8 = b.tensor(h16, w16, 256)
        self._cv(b, hu + ".g8_conv", d8, a8, flags=nat.F_RESIDUAL, res=a16)
        g4cat = View(g4c_buf, 0, g4c_pitch, h4, w4, 257, 264)
        d4 = View(b.buf(P16 * g4c_pitch, 2), 0, g4c_pitch, h16, w16, 257, 264)
        b.ew(hu + ".g4_area", g4cat, d4, mode=2, factor=4)
        hu_pitch = pitch_for(256 + HD)
        hu_buf = b.buf(P16 * hu_pitch, 2)
        self._cv(b, hu + ".g4_conv", d4, View(hu_buf, 0, hu_pitch, h16, w16, 256, 256), flags=nat.F_RESIDUAL, res=a8)
        b.ew(hu + ".hidden_in", View(dc_buf, 1024 + CV, dc_pitch, h16, w16, HD, HD), View(hu_buf, 256, hu_pitch, h16, w16, HD, HD))
        vals2 = b.tensor(h16, w16, 3 * HD)
        self._cv(b, hu + ".transform", View(hu_buf, 0, hu_pitch, h16, w16, 256 + HD, 256 + HD), vals2, pad=1)
        b.gru(hu, vals2, io["hidden"], io["hidden_out"], HD)
        mark("segment_hidden", s0, 2)

        # ================= slice "short": depthwise 5x5 + Linear behind the local attention, one frame =================
        s0 = len(b.ops)
        b.slice_batch = 1
        st = "short_term_attn"
        a_in = b.tensor(h16, w16, 2 * CV)
        b.planar_in(st + ".agg_in", fbuf("agg", 2 * CV * P16), 2 * CV, a_in, pixel_major=True)
        a_dw = self._dw(b, st + ".dw_conv", [st + ".dw_conv.conv"], a_in, b.tensor(h16, w16, 2 * CV), 5)
        a_out = b.tensor(h16, w16, 2 * CV)
        pc = self._plain(st + ".projection", a_dw, self._w(st + ".projection.weight"), bias=self._w(st + ".projection.bias"))
        b.conv(st + ".projection", pc, a_dw, a_out)
        b.planar_out(st + ".out", a_out, 0, 2 * CV, fbuf("short", 2 * CV * P16), 0)
        mark("short", s0, 1)

        ops, bufs = b.finish()
        return ops, bufs, b.names, consts, sl, io


class _Feat:
    """multi-scale image features of one encode_key call (NHWC fp16 device tensors the plan wrote through bound buffers); skip8 / skip4: the
    decoder's skip convs of these features when the look-ahead pass already ran them (else None: segment() runs them)"""
    __slots__ = ("g16", "g8", "g4", "shape", "skip8", "skip4")

    def __init__(self, g16, g8, g4, shape, skip8=None, skip4=None):
        self.g16, self.g8, self.g4, self.shape, self.skip8, self.skip4 = g16, g8, g4, shape, skip8, skip4


class _FastBuffers:
    """the per-frame results of one padded frame size, allocated once (colormnet_fast.py)"""

    def __init__(self, network, H, W):
        import torch
        self.net = network._net(H, W)
        self.H, self.W, self.h, self.w = H, W, H // 16, W // 16
        hw, CV, HD, CK = self.h * self.w, network.value_dim, network.hidden_dim, network.key_dim
        ws2 = (2 * MAX_DIS + 1) ** 2
        with network.on_stream():
            new = lambda *shape: torch.empty(shape, dtype=torch.float32, device=network.device)
            self.readout, self.short, self.agg, self.attn = new(2 * CV, hw), new(2 * CV, hw), new(hw, 2 * CV), new(ws2, hw)
            self.readout2 = new(2 * CV, hw)                     # the read of the NEXT frame lands here while the decoder reads `readout` (and vice versa)
            self.prob, self.value_in, self.value = new(2, H, W), new(2, 5, H, W), new(2, CV, hw)
            self.hidden = [new(1, 2, HD, self.h, self.w), new(1, 2, HD, self.h, self.w)]
            self.last_key, self.last_value = new(1, CK, self.h, self.w), new(1, 2, CV, self.h, self.w)

    def other_hidden(self, current):
        """a hidden-state buffer that is not the one being read"""
        return self.hidden[1] if (current is not None and current.data_ptr() == self.hidden[0].data_ptr()) else self.hidden[0]


class _OnStream:
    """run a block on the library's stream; a caller that works on another torch stream is ordered before and after through events (no host sync)"""

    def __init__(self, net):
        self.net = net

    def __enter__(self):
        import torch
        n = self.net
        self.outer = torch.cuda.current_stream(n.device)
        self.foreign = self.outer.cuda_stream != n.stream.cuda_stream
        if self.foreign:
            n.stream.wait_stream(self.outer)
        self.cm = torch.cuda.stream(n.stream)
        self.cm.__enter__()
        return self

    def __exit__(self, *exc):
        self.cm.__exit__(*exc)
        if self.foreign:
            self.outer.wait_stream(self.net.stream)
        return False


class ColorMNetNetwork:
    """the object InferenceCore drives: encode_key / encode_value / segment / short_term_attn on device tensors (torch, fp32, the
    reference's shapes), plus the frame transforms of ColorMNetRender.  One instance per (weights, GPU)."""

    def __init__(self, state_dict, device_index=0, autotune=None, worker=0, share=None):
        """worker: index of the per-thread context on this GPU (render.get_context): networks built with different worker indices run
        CONCURRENTLY from different threads, each on its own HIP stream (independent clips: replicas inside one GPU);
        share: another ColorMNetNetwork of the same GPU whose packed plan and device weights are reused (read-only)."""
        import torch
        from .render import get_context
        if not torch.cuda.is_available():
            raise nat.NativeLibraryError("ColorMNetNetwork: torch sees no GPU (device tensors are the interface of the ColorMNet step)")
        self.ctx = get_context(device_index, worker)
        if share is not None:
            if share.ctx.device_id != self.ctx.device_id:
                raise ValueError("share: a network of the same GPU")
            self.plan, self.weights, self._owns_weights = share.plan, share.weights, False
        else:
            self.plan = ColorMNetPlan(state_dict)
            self.weights, self._owns_weights = nat.Weights(self.ctx, self.plan.blob), True
        self.key_dim, self.value_dim, self.hidden_dim = self.plan.key_dim, self.plan.value_dim, self.plan.hidden_dim
        self.device = torch.device("cuda", device_index)
        self.stream = torch.cuda.ExternalStream(self.ctx.stream_ptr(), device=self.device)
        self.nets = {}
        self._armed, self._helper, self.worker = None, None, worker
        self.async_lookahead = os.environ.get("HAVC_CMN_ASYNC_LOOKAHEAD", "1") != "0"      # look-ahead pass on its own stream (0: on this network's stream)
        self.autotune = (os.environ.get("HAVC_AUTOTUNE", "1") != "0") if autotune is None else autotune
        self.fast = os.environ.get("HAVC_CMN_FAST", "1") != "0"                           # colormnet_fast.py: the frame loop on pre-sized buffers
        self._fastbufs = {}
        sd = self.plan.sd
        ws2 = (2 * MAX_DIS + 1) ** 2
        with self.on_stream():
            self.rel_w = torch.from_numpy(sd["short_term_attn.relative_emb_k.weight"].astype(np.float32).reshape(ws2, -1)).to(self.device)
            self.rel_b = torch.from_numpy(sd["short_term_attn.relative_emb_k.bias"].astype(np.float32)).to(self.device)

    # ---- plumbing ----
    def on_stream(self):
        return _OnStream(self)

    def _net(self, H, W):
        key = (H, W)
        if key not in self.nets:
            ops, bufs, names, consts, sl, io = self.plan.plan(H, W)
            n = nat.Net(self.ctx, self.weights, ops, bufs, 0, 0, H, 2)
            n.names, n.plan_ops, n.slices, n.io = names, ops, sl, io
            for buf, arr, pitch, rows in consts:
                a = np.zeros((2, rows, pitch), np.float16)
                a[:, :arr.shape[0], :arr.shape[1]] = arr.astype(np.float16)[None]
                n.upload(buf, a)
            if self.autotune:
                n.autotune(1)
            self.nets[key] = n
        return self.nets[key]

    def _key_net(self, H, W, B):
        """the look-ahead plan: the same ops with the key slice laid out (and its split-K counts chosen) for B frames per launch"""
        key = (H, W, "key", B)
        if key not in self.nets:
            ops, bufs, names, consts, sl, io = self.plan.plan(H, W, key_batch=B)
            n = nat.Net(self.ctx, self.weights, ops, bufs, 0, 0, H, max(B, 2))
            n.names, n.plan_ops, n.slices, n.io = names, ops, sl, io
            for buf, arr, pitch, rows in consts:
                a = np.zeros((max(B, 2), rows, pitch), np.float16)
                a[:, :arr.shape[0], :arr.shape[1]] = arr.astype(np.float16)[None]
                n.upload(buf, a)
            if self.autotune:
                n.autotune(B)
            self.nets[key] = n
        return self.nets[key]

    def _run(self, net, name, batch=None):
        first, count, b = net.slices[name]
        net.enqueue_ops(first, count, b if batch is None else batch)

    def _new(self, *shape, dtype=None):
        import torch
        return torch.empty(shape, dtype=dtype or torch.float32, device=self.device)

    def _feat_tensor(self, net, name, rows):
        import torch
        pitch = int(net.bufs[net.io[name]]["elems_per_frame"]) // rows
        return torch.empty(rows * pitch + 128, dtype=torch.float16, device=self.device)     # + slack: vector loads may touch the tail

    # ---- look-ahead: encode_key of the next frames of a clip in ONE batched pass (nothing in encode_key depends on the memory) ----
    def _helper_net(self):
        """the network object the look-ahead pass runs on: another context (its own HIP stream and activation arena) of the same GPU, same packed
        weights.  The pass for the NEXT frames then overlaps the frame-by-frame memory step, whose small launches leave most CUs idle."""
        if self._helper is None:
            from .render import get_context
            hctx = get_context(self.ctx.device_id, ("lookahead", self.worker))
            if os.environ.get("HAVC_CMN_LOOKAHEAD_PRIORITY", "0") == "low" and not getattr(hctx, "_low_priority", False):
                # A/B switch, OFF: the look-ahead streams at the lowest stream priority.  The dispatcher then serves the memory step's queue first
                # whenever both have a block ready -- and the batched pass, which the NEXT window cannot start without, starves: c5 1 031 -> 612
                # frames/s (tools/sessions/r5_run14.sh).  (Before any stream handle of hctx is handed out.)
                nat.check(hctx.lib.havc_ctx_set_stream_priority(hctx.h, -1), hctx.h)
                hctx._low_priority = True
            cus = int(os.environ.get("HAVC_CMN_LOOKAHEAD_CUS", "0"))
            if cus > 0 and not getattr(hctx, "_cu_masked", False):
                # the batched pass on `cus` of the 256 CUs: the rest stay free for the memory step's small dependent launches
                nat.check(hctx.lib.havc_ctx_set_stream_cus(hctx.h, cus), hctx.h)
                hctx._cu_masked = True
            self._helper = ColorMNetNetwork(None, device_index=self.ctx.device_id, autotune=self.autotune, worker=("lookahead", self.worker), share=self)
        return self._helper

    def prefetch_keys(self, frames, max_batch=None, _ordered=False):
        """frames: list of [3, H, W] device tensors, padded as InferenceCore pads them (pad_divide_by 112), in the order in which they will be
        stepped.  Returns one entry per frame; the caller (ColorMNetRender) keeps them and hands the frame's entry back through
        `expect_prefetched(entry)` right before the step, whose first encode_key call then takes it instead of computing.
        The pass is only ENQUEUED here, on the helper context's stream, behind everything this network's stream holds so far; an event marks
        its end and the consumer (encode_key) makes this network's stream wait for it."""
        import torch
        if frames is None or len(frames) == 0:
            return []
        B = len(frames)
        H, W = frames[0].shape[-2:]
        hn = self._helper_net() if self.async_lookahead else self
        net = hn._key_net(H, W, max_batch or B)
        h, w = H // 16, W // 16
        with self.on_stream():
            if hn is not self and not _ordered:
                hn.stream.wait_stream(self.stream)                    # whatever this stream still owes the frames first
            with torch.cuda.stream(hn.stream):
                if isinstance(frames, torch.Tensor):                  # a batch the caller assembled in place (prefetch_frames: havc_cmn_frame_in per slot)
                    img = frames
                else:
                    img = torch.stack([f.to(self.device, torch.float32) for f in frames], 0).contiguous()
                key, sel, shr = self._new(B, self.key_dim, h, w), self._new(B, self.key_dim, h, w), self._new(B, 1, h, w)
                big, epf = {}, {}
                feats = ("g16", "g8", "g4", "skip8", "skip4")
                for name in feats:
                    epf[name] = int(net.bufs[net.io[name]]["elems_per_frame"])
                    big[name] = torch.empty(B * epf[name] + 128, dtype=torch.float16, device=self.device)
                for name, t in (("image", img), ("key", key), ("selection", sel), ("shrinkage", shr)) + tuple((n_, big[n_]) for n_ in feats):
                    net.bind(net.io[name], t.data_ptr())
                hn._run(net, "key", B)
                hn._run(net, "skip", B)                                # the decoder's skip convs read only these features: they run ahead too
                done = None
                if hn is not self:
                    done = torch.cuda.Event()
                    done.record(hn.stream)
                    for t in (key, sel, shr) + tuple(big.values()):
                        t.record_stream(self.stream)                   # allocated on the helper's stream, read on this one
            entries = []
            for i in range(B):
                v = [big[n_][i * epf[n_]:(i + 1) * epf[n_] + 128] for n_ in feats]
                entries.append((key[i:i + 1], shr[i:i + 1], sel[i:i + 1], _Feat(v[0], v[1], v[2], (H, W), v[3], v[4]), done, img[i]))
        return entries

    def lookahead_context(self):
        """the libhavc context whose stream runs the look-ahead (callers that prepare frames for prefetch -- the Spline64 squash of
        DeepExColorMNet -- enqueue there, off the memory step's stream); this network's own context when the look-ahead is synchronous"""
        return self._helper_net().ctx if self.async_lookahead else self.ctx

    def lookahead_wait_for_main(self):
        """the look-ahead stream waits (on the GPU) for everything this network's stream holds so far"""
        if self.async_lookahead:
            self._helper_net().stream.wait_stream(self.stream)

    def prefetch_frames(self, frames, max_batch=None):
        """The whole look-ahead of ColorMNetRender for frames that will be stepped next, in order: RGB -> Lab, the L plane repeated and padded as
        InferenceCore does it (pad_divide_by 112), then prefetch_keys -- ALL on the look-ahead stream, so that the memory step's stream is not
        held up by any of it.  frames: u8 [H, W, 3] host arrays or DeviceImages of one size.  -> (Lab tensors, entries); the consumer calls
        wait_prefetched(entry) before it touches the frame's Lab planes."""
        import ctypes as C
        import torch
        from .colormnet_core import DIVIDE_BY, pad_divide_by
        from .device import is_device
        hn = self._helper_net() if self.async_lookahead else self
        if self.fast:
            # no tensor ops: every frame's Lab planes and its padded network input are written by ONE kernel straight into their batch slots
            from .colormnet_fast import frame_pads
            shape0 = tuple(frames[0].shape)
            pad, Hp, Wp = frame_pads(shape0[0], shape0[1])
            with self.on_stream():
                # frames that are not known to be complete may still be in flight on this network's stream: the look-ahead stream waits for its tail
                # (frames uploaded by a blocking call, or squashed on the look-ahead context itself, need no such wait: the pass then overlaps
                # whatever the memory step still has queued)
                ready = all(is_device(f) and (f.complete or getattr(f, "produced_on_lookahead", False)) for f in frames)
                if hn is not self and not ready:
                    hn.stream.wait_stream(self.stream)
                    # wait_stream orders the look-ahead stream behind THIS network's stream only: a frame still being written on a foreign context's
                    # stream (another model's output handed over unsynchronised) is drained on the host, as DeepExColorMNet._announce does
                    for pctx in {id(f.ctx): f.ctx for f in frames if is_device(f) and not f.complete and not getattr(f, "produced_on_lookahead", False)
                                 and f.ctx is not self.ctx and f.ctx is not hn.ctx}.values():
                        pctx.synchronize()
                with torch.cuda.stream(hn.stream):
                    B = len(frames)
                    labs = torch.empty((B, 3, shape0[0], shape0[1]), dtype=torch.float32, device=self.device)
                    img = torch.empty((B, 3, Hp, Wp), dtype=torch.float32, device=self.device)
                    keep = []
                    for i, f in enumerate(frames):
                        if is_device(f):
                            ptr = f.ptr
                        else:
                            a = np.ascontiguousarray(f, dtype=np.uint8)
                            keep.append(a)
                            ptr = nat.as_ptr(a)
                        nat.check(hn.ctx.lib.havc_cmn_frame_in(hn.ctx.h, ptr, C.c_void_p(labs[i].data_ptr()), C.c_void_p(img[i].data_ptr()), shape0[1], shape0[0],
                                                               Wp, Hp, pad[0], pad[2]), hn.ctx.h)
                    if hn is not self:
                        labs.record_stream(self.stream)
                        img.record_stream(self.stream)
                entries = self.prefetch_keys(img, max_batch=max_batch, _ordered=True)
            return [labs[i] for i in range(len(frames))], entries
        with self.on_stream():
            if hn is not self:
                hn.stream.wait_stream(self.stream)                    # frames the caller produced on this stream
            with torch.cuda.stream(hn.stream):
                labs, keep = [], []
                for f in frames:
                    if is_device(f):
                        shape, ptr = f.shape, f.ptr
                    else:
                        a = np.ascontiguousarray(f, dtype=np.uint8)
                        keep.append(a)
                        shape, ptr = a.shape, nat.as_ptr(a)
                    lab = self._new(3, shape[0], shape[1])
                    nat.check(hn.ctx.lib.havc_colormnet_rgb_to_lab(hn.ctx.h, ptr, C.c_void_p(lab.data_ptr()), shape[1], shape[0]), hn.ctx.h)
                    if hn is not self:
                        lab.record_stream(self.stream)
                    labs.append(lab)
                padded = [pad_divide_by(lab[:1].repeat(3, 1, 1), DIVIDE_BY)[0] for lab in labs]
            entries = self.prefetch_keys(padded, max_batch=max_batch, _ordered=True)
        return labs, entries

    def wait_prefetched(self, entry):
        """make this network's stream wait for the look-ahead pass `entry` came from (its Lab planes and squashed frame included)"""
        import torch
        done = entry[4]
        if done is not None:
            with self.on_stream():
                torch.cuda.current_stream(self.device).wait_event(done)

    def expect_prefetched(self, entry):
        """the NEXT encode_key call is for the frame `entry` was computed from (InferenceCore encodes the frame first, then an exemplar)"""
        self._armed = entry

    # ---- network.py:52-85 ----
    def encode_key(self, frame, need_ek=True, need_sk=True):
        import torch
        assert frame.dim() == 4 and frame.shape[0] == 1, "one frame [1, 3, H, W]"
        if self._armed is not None:
            (key, shr, sel, f, done, _img), self._armed = self._armed, None
            if f.shape == tuple(frame.shape[-2:]):
                if done is not None:
                    with self.on_stream():
                        torch.cuda.current_stream(self.device).wait_event(done)     # the look-ahead pass that produced this entry (idempotent)
                return key, (shr if need_sk else None), (sel if need_ek else None), f, f, f
        H, W = frame.shape[-2:]
        net = self._net(H, W)
        h, w = H // 16, W // 16
        with self.on_stream():
            img = frame.to(self.device, torch.float32).contiguous()
            key, sel, shr = self._new(1, self.key_dim, h, w), self._new(1, self.key_dim, h, w), self._new(1, 1, h, w)
            f = _Feat(self._feat_tensor(net, "g16", h * w), self._feat_tensor(net, "g8", 4 * h * w), self._feat_tensor(net, "g4", 16 * h * w), (H, W))
            for name, t in (("image", img), ("key", key), ("selection", sel), ("shrinkage", shr), ("g16", f.g16), ("g8", f.g8), ("g4", f.g4)):
                net.bind(net.io[name], t.data_ptr())
            self._run(net, "key")
            self._keep = (img,)
        return key, (shr if need_sk else None), (sel if need_ek else None), f, f, f

    # ---- network.py:87-101 ----
    def encode_value(self, frame, image_feat_f16, h16, masks, is_deep_update=True):
        import torch
        H, W = frame.shape[-2:]
        net = self._net(H, W)
        h, w = H // 16, W // 16
        if masks.shape[1] != 2:
            raise NotImplementedError("two objects (the a and b planes, colormnet_render.py:239-241)")
        with self.on_stream():
            img = frame.to(self.device, torch.float32)[0]
            m = masks.to(self.device, torch.float32)[0]
            vin = torch.stack([torch.cat([img, m[0:1], m[1:2]], 0), torch.cat([img, m[1:2], m[0:1]], 0)], 0).contiguous()   # image | mask_i | others_i
            value = self._new(1, 2, self.value_dim, h, w)
            net.bind(net.io["value_in"], vin.data_ptr())
            net.bind(net.io["value"], value.data_ptr())
            net.bind(net.io["g16"], image_feat_f16.g16.data_ptr())
            self._run(net, "value")
            if is_deep_update:
                hin = h16.to(self.device, torch.float32).contiguous()
                hout = self._new(1, 2, self.hidden_dim, h, w)
                net.bind(net.io["hidden"], hin.data_ptr())
                net.bind(net.io["hidden_out"], hout.data_ptr())
                self._run(net, "value_hidden")
                h16 = hout
            self._keep = (vin,)
        return value, h16

    # ---- network.py:137-145 ----
    def segment(self, multi_scale_features, memory_readout, hidden_state, selector=None, h_out=True, strip_bg=True):
        import torch
        f = multi_scale_features[0]
        H, W = f.shape
        net = self._net(H, W)
        h, w = H // 16, W // 16
        with self.on_stream():
            ro = memory_readout.to(self.device, torch.float32).contiguous()
            hin = hidden_state.to(self.device, torch.float32).contiguous()
            prob = self._new(1, 2, H, W)
            for name, t in (("g16", f.g16), ("g8", f.g8), ("g4", f.g4), ("readout", ro), ("hidden", hin), ("prob", prob)):
                net.bind(net.io[name], t.data_ptr())
            if f.skip8 is not None:                                   # the look-ahead pass has run the skip convs of this frame
                net.bind(net.io["skip8"], f.skip8.data_ptr())
                net.bind(net.io["skip4"], f.skip4.data_ptr())
            else:
                net.bind(net.io["skip8"], None)
                net.bind(net.io["skip4"], None)
                self._run(net, "skip")
            self._run(net, "segment")
            hidden = None
            if h_out:
                hidden = self._new(1, 2, self.hidden_dim, h, w)
                net.bind(net.io["hidden_out"], hidden.data_ptr())
                self._run(net, "segment_hidden")
            self._keep = (ro, hin)
        return hidden, prob, prob

    # ---- attention.py:783-860 (one head, use_linear=False) ----
    def short_term_attn(self, q, k, v, u, size_2d):
        import torch
        from . import colormnet as K
        h, w = size_2d
        net = self._net(h * 16, w * 16)
        with self.on_stream():
            agg, attn = K.local_attention(q, k, v, self.rel_w, self.rel_b, MAX_DIS, 1, device_index=self.ctx.device_id)    # [h*w, 1, 2 CV]
            out = self._new(1, 2 * self.value_dim, h * w)
            net.bind(net.io["agg"], agg.data_ptr())
            net.bind(net.io["short"], out.data_ptr())
            self._run(net, "short")
            self._keep = (agg,)
        return out.permute(2, 0, 1), attn

    # ---- fast step (colormnet_fast.py): preallocated results, multi-bind / multi-slice calls, no tensor ops between the kernels ----
    def fast_buffers(self, H, W):
        key = (H, W)
        if key not in self._fastbufs:
            self._fastbufs[key] = _FastBuffers(self, H, W)
        return self._fastbufs[key]

    def _bind_run(self, net, binds, slices):
        import ctypes as C
        n = len(binds)
        ids = (C.c_int32 * n)(*[net.io[k] for k, _ in binds])
        ptrs = (C.c_void_p * n)(*[(t.data_ptr() if t is not None else None) for _, t in binds])
        nat.check(self.ctx.lib.havc_net_bind_many(net.h, n, ids, ptrs), self.ctx.h)
        m = len(slices)
        sl = [net.slices[name] for name in slices]
        nat.check(self.ctx.lib.havc_net_enqueue_slices(net.h, m, (C.c_int32 * m)(*[v[0] for v in sl]), (C.c_int32 * m)(*[v[1] for v in sl]),
                                                       (C.c_int32 * m)(*[v[2] for v in sl])), self.ctx.h)

    def short_term_fork(self, B, key, last_key, last_value):
        """LocalGatedPropagation + the `short` slice on the context's second stream (attention.py:783-860); short_term_join adds the result to the readout"""
        import ctypes as C
        net = B.net
        first, count, _ = net.slices["short"]
        p = lambda t: C.c_void_p(t.data_ptr())
        nat.check(self.ctx.lib.havc_cmn_short_term(self.ctx.h, net.h, first, count, net.io["agg"], net.io["short"], p(key), p(last_key), p(last_value),
                                                   p(self.rel_w), p(self.rel_b), p(B.agg), p(B.attn), p(B.short), self.key_dim, 2 * self.value_dim, B.h, B.w,
                                                   MAX_DIS), self.ctx.h)

    def short_term_join(self, B, readout=None):
        import ctypes as C
        readout = B.readout if readout is None else readout
        nat.check(self.ctx.lib.havc_cmn_join_add(self.ctx.h, C.c_void_p(readout.data_ptr()), C.c_void_p(B.short.data_ptr()), readout.numel()), self.ctx.h)

    def segment_fast(self, B, f, hidden_in, hidden_out, readout=None):
        """Decoder (+ HiddenUpdater when hidden_out is given) on the readout (B.readout unless given) -> B.prob (network.py:137-145)"""
        readout = B.readout if readout is None else readout
        binds = [("g16", f.g16), ("g8", f.g8), ("g4", f.g4), ("readout", readout), ("hidden", hidden_in), ("prob", B.prob),
                 ("skip8", f.skip8), ("skip4", f.skip4)]
        slices = ([] if f.skip8 is not None else ["skip"]) + ["segment"]
        if hidden_out is not None:
            binds.append(("hidden_out", hidden_out))
            slices.append("segment_hidden")
        self._bind_run(B.net, binds, slices)

    def encode_value_fast(self, B, image, f16, planes, hidden_in, hidden_out):
        """ValueEncoder (+ HiddenReinforcer when hidden_out is given) -> B.value (network.py:87-101); image [3, H, W] padded, planes [2, H, W]"""
        import ctypes as C
        nat.check(self.ctx.lib.havc_cmn_value_in(self.ctx.h, C.c_void_p(image.data_ptr()), C.c_void_p(planes.data_ptr()), C.c_void_p(B.value_in.data_ptr()),
                                                 B.H * B.W), self.ctx.h)
        binds = [("value_in", B.value_in), ("value", B.value), ("g16", f16.g16)]
        slices = ["value"]
        if hidden_out is not None:
            binds += [("hidden", hidden_in), ("hidden_out", hidden_out)]
            slices.append("value_hidden")
        self._bind_run(B.net, binds, slices)

    def keep_last(self, B, key):
        import ctypes as C
        for dst, src in ((B.last_key, key), (B.last_value, B.value)):
            nat.check(self.ctx.lib.havc_dev_copy(self.ctx.h, C.c_void_p(dst.data_ptr()), C.c_void_p(src.data_ptr()), src.numel() * 4), self.ctx.h)

    def frame_in(self, rgb_u8):
        """u8 [h, w, 3] (host array or DeviceImage) -> (Lab planes [3, h, w], the padded network input [3, H, W], pads): ONE kernel"""
        import ctypes as C
        from .colormnet_fast import frame_pads
        from .device import is_device
        if is_device(rgb_u8):
            shape, ptr, keep = rgb_u8.shape, rgb_u8.ptr, rgb_u8
        else:
            keep = np.ascontiguousarray(rgb_u8, dtype=np.uint8)
            shape, ptr = keep.shape, nat.as_ptr(keep)
        if len(shape) != 3 or shape[2] != 3:
            raise ValueError("RGB image expected")
        pad, Hp, Wp = frame_pads(shape[0], shape[1])
        with self.on_stream():
            lab, img = self._new(3, shape[0], shape[1]), self._new(3, Hp, Wp)
            nat.check(self.ctx.lib.havc_cmn_frame_in(self.ctx.h, ptr, C.c_void_p(lab.data_ptr()), C.c_void_p(img.data_ptr()), shape[1], shape[0], Wp, Hp,
                                                     pad[0], pad[2]), self.ctx.h)
        return lab, img, pad

    def frame_out(self, lab, prob_padded, pad, out=None):
        """L plane of `lab` + the padded ab planes -> u8 [h, w, 3]: a host array (blocks) or `out` (a DeviceImage: only enqueued)"""
        import ctypes as C
        h, w = lab.shape[-2:]
        Hp, Wp = prob_padded.shape[-2:]
        host = out is None
        if host:
            out = np.empty((h, w, 3), np.uint8)
        nat.check(self.ctx.lib.havc_cmn_frame_out(self.ctx.h, C.c_void_p(lab.data_ptr()), C.c_void_p(prob_padded.data_ptr()), nat.as_ptr(out) if host else out.ptr,
                                                  w, h, Wp, Hp, pad[0], pad[2]), self.ctx.h)
        return out

    # ---- ColorMNetRender's frame transforms (colormnet_render.py:285-301, 276-279) ----
    def image_to_lab(self, rgb_u8):
        """u8 [H, W, 3] (host array / PIL image, or a device.DeviceImage) -> normalised Lab [3, H, W] fp32 on the device"""
        import ctypes as C
        from .device import is_device
        if is_device(rgb_u8):
            shape, ptr, keep = rgb_u8.shape, rgb_u8.ptr, rgb_u8
        else:
            keep = np.ascontiguousarray(rgb_u8, dtype=np.uint8)
            shape, ptr = keep.shape, nat.as_ptr(keep)
        if len(shape) != 3 or shape[2] != 3:
            raise ValueError("RGB image expected")
        with self.on_stream():
            lab = self._new(3, shape[0], shape[1])
            nat.check(self.ctx.lib.havc_colormnet_rgb_to_lab(self.ctx.h, ptr, C.c_void_p(lab.data_ptr()), shape[1], shape[0]), self.ctx.h)
        return lab

    def lab_to_image(self, l_plane, ab, out=None):
        """L [1, H, W] + ab [2, H, W] (device, normalised) -> u8 [H, W, 3]: a host array (blocks until it is there), or `out` (a DeviceImage:
        only enqueued)"""
        import ctypes as C
        import torch
        H, W = l_plane.shape[-2:]
        host = out is None
        if host:
            out = np.empty((H, W, 3), np.uint8)
        with self.on_stream():
            lp, abp = l_plane.to(self.device, torch.float32).contiguous(), ab.to(self.device, torch.float32).contiguous()
            nat.check(self.ctx.lib.havc_colormnet_lab_to_rgb(self.ctx.h, C.c_void_p(lp.data_ptr()), C.c_void_p(abp.data_ptr()),
                                                             nat.as_ptr(out) if host else out.ptr, W, H), self.ctx.h)
        return out

    def close(self):
        self._fastbufs.clear()
        if self._helper is not None:
            self._helper.close()
            self._helper = None
        for n in self.nets.values():
            n.close()
        self.nets.clear()
        if self._owns_weights:
            self.weights.close()
