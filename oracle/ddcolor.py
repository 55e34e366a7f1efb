"""DDColor (vsddcolor.ddcolor) — fp32 PyTorch-CPU restatement, ORACLE / TEST INFRASTRUCTURE ONLY.

PARITY UNPINNED.  The algorithm lives in the third-party wheel `vsddcolor` (>=1.0.0, pyproject.toml:29 of the reference;
its README.md:30-34 points at a patched 1.0.1 wheel), which is NOT under /root/reference and not installed here; the reference
holds no test, golden vector or fixture at that boundary.  The only things the reference pins are the call site
(vsdeoldify/vsslib/vsmodels.py:298-363: RGBH/RGBS square clip in, model 0 = ddcolor_modelscope / 1 = ddcolor_artistic,
input_size = trunc(render_factor / 2) * 32, RGB24 cast after).  What follows restates the PUBLISHED architecture of DDColor
(Kang et al., ICCV 2023; piddnad/DDColor basicsr/archs/ddcolor_arch.py and ddcolor_arch_utils/{convnext,unet,transformer,
position_encoding}.py) from the builder's knowledge of that public source:

  encoder   ConvNeXt-L (depths 3/3/27/3, dims 192/384/768/1536): stem conv 4x4 s4 + channel LayerNorm; block =
            depthwise 7x7 -> LayerNorm(C) -> Linear 4C -> GELU -> Linear C -> layer scale gamma -> + x; LayerNorm + conv 2x2 s2
            between stages; an extra channel LayerNorm norm{i} on every stage output feeds the decoder hooks.
  decoder   three UnetBlockWide (same family as deoldify/unet.py:170-205: CustomPixelShuffle_ICNR(blur) on the up path,
            BatchNorm on the skip, ReLU(cat), spectral conv3x3 -> ReLU -> BN) giving 512@1/16, 512@1/8, 256@1/4, then a x4
            CustomPixelShuffle_ICNR to 256 channels at full resolution.
  colour    MultiScaleColorDecoder: 100 learned queries, 9 layers of {cross-attention to the 1/16, 1/8, 1/4 features in turn
            (1x1 input projection + level embedding, sine position encoding on keys), self-attention, FFN 2048}, post-norm,
            8 heads, hidden 256; decoder LayerNorm, 3-layer MLP colour embedding; logits = einsum(bqc, bchw -> bqhw) with the
            256-channel full-resolution feature.
  refine    spectral 1x1 conv (100 + 3 -> 2) on cat[logits, normalised image]  =>  ab.
  wrapper   L of the frame (Lab), gray RGB = Lab(L, 0, 0) -> RGB at input_size, imagenet-normalised, network, ab resized back,
            Lab(L_orig, ab) -> RGB.  Lab arithmetic = oracle/zhang.py's skimage restatement (itself unpinned).
State-dict key names follow that public source (encoder.arch.*, decoder.layers.N.*, decoder.last_shuf.*,
decoder.color_decoder.*, refine_net.0.0.*); vsdeoldify_amd/synth.py:ddcolor_state_dict_spec enumerates them.
The ConvNeXt stage arithmetic is additionally checked against the independent implementation in the `transformers`
package present in this image (tests/test_ddcolor.py): that pins the published encoder block, nothing DDColor-specific.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

from . import unet as U
from . import zhang as Z

DEPTHS, DIMS = (3, 3, 27, 3), (192, 384, 768, 1536)
HIDDEN, HEADS, FFN, QUERIES, DEC_LAYERS, SCALES = 256, 8, 2048, 100, 9, 3
MEAN = torch.tensor([0.485, 0.456, 0.406]).view(1, 3, 1, 1)
STD = torch.tensor([0.229, 0.224, 0.225]).view(1, 3, 1, 1)


def layernorm_cf(x, w, b, eps=1e-6):
    """channels_first LayerNorm of convnext.py (biased variance over C)."""
    u = x.mean(1, keepdim=True)
    s = (x - u).pow(2).mean(1, keepdim=True)
    return w[None, :, None, None] * ((x - u) / torch.sqrt(s + eps)) + b[None, :, None, None]


def convnext_block(sd, p, x):
    c = x.shape[1]
    y = F.conv2d(x, sd[p + ".dwconv.weight"], sd[p + ".dwconv.bias"], padding=3, groups=c)
    y = y.permute(0, 2, 3, 1)
    y = F.layer_norm(y, (c,), sd[p + ".norm.weight"], sd[p + ".norm.bias"], 1e-6)
    y = F.linear(y, sd[p + ".pwconv1.weight"], sd[p + ".pwconv1.bias"])
    y = F.gelu(y)
    y = F.linear(y, sd[p + ".pwconv2.weight"], sd[p + ".pwconv2.bias"])
    y = sd[p + ".gamma"] * y
    return x + y.permute(0, 3, 1, 2)


def encoder(sd, x, depths=DEPTHS):
    """-> [norm0(stage0), ..., norm3(stage3)]"""
    e = "encoder.arch"
    feats = []
    for i in range(4):
        d = f"{e}.downsample_layers.{i}"
        if i == 0:
            x = F.conv2d(x, sd[d + ".0.weight"], sd[d + ".0.bias"], stride=4)
            x = layernorm_cf(x, sd[d + ".1.weight"], sd[d + ".1.bias"])
        else:
            x = layernorm_cf(x, sd[d + ".0.weight"], sd[d + ".0.bias"])
            x = F.conv2d(x, sd[d + ".1.weight"], sd[d + ".1.bias"], stride=2)
        for j in range(depths[i]):
            x = convnext_block(sd, f"{e}.stages.{i}.{j}", x)
        feats.append(layernorm_cf(x, sd[f"{e}.norm{i}.weight"], sd[f"{e}.norm{i}.bias"]))
    return feats


def custom_shuffle(sd, p, x, scale=2):
    x = U.bn(sd, p + ".conv.1", F.conv2d(x, U.conv_w(sd, p + ".conv.0")))
    x = F.pixel_shuffle(F.relu(x), scale)
    return F.avg_pool2d(F.pad(x, (1, 0, 1, 0), mode="replicate"), 2, stride=1)


def unet_block(sd, p, up_in, skip):
    up = custom_shuffle(sd, p + ".shuf", up_in)
    if skip.shape[-2:] != up.shape[-2:]:
        up = F.interpolate(up, skip.shape[-2:], mode="nearest")
    cat = F.relu(torch.cat([up, U.bn(sd, p + ".bn", skip)], dim=1))
    return U.conv_relu_bn(sd, p + ".conv", cat)


def position_sine(b, h, w, num_pos_feats=HIDDEN // 2, temperature=10000.0):
    """PositionEmbeddingSine(normalize=True, scale=2 pi) of DETR / Mask2Former -> [b, 2 * num_pos_feats, h, w]."""
    y_embed = torch.arange(1, h + 1, dtype=torch.float32).view(1, h, 1).expand(b, h, w)
    x_embed = torch.arange(1, w + 1, dtype=torch.float32).view(1, 1, w).expand(b, h, w)
    eps, scale = 1e-6, 2 * math.pi
    y_embed = y_embed / (y_embed[:, -1:, :] + eps) * scale
    x_embed = x_embed / (x_embed[:, :, -1:] + eps) * scale
    dim_t = torch.arange(num_pos_feats, dtype=torch.float32)
    dim_t = temperature ** (2 * torch.div(dim_t, 2, rounding_mode="floor") / num_pos_feats)
    pos_x = x_embed[:, :, :, None] / dim_t
    pos_y = y_embed[:, :, :, None] / dim_t
    pos_x = torch.stack((pos_x[:, :, :, 0::2].sin(), pos_x[:, :, :, 1::2].cos()), dim=4).flatten(3)
    pos_y = torch.stack((pos_y[:, :, :, 0::2].sin(), pos_y[:, :, :, 1::2].cos()), dim=4).flatten(3)
    return torch.cat((pos_y, pos_x), dim=3).permute(0, 3, 1, 2)


def mha(sd, p, query, key, value, heads=HEADS):
    """nn.MultiheadAttention forward (batch_first layout here): [B, Lq, E], [B, Lk, E] -> [B, Lq, E]."""
    e = query.shape[-1]
    w, bias = sd[p + ".in_proj_weight"], sd[p + ".in_proj_bias"]
    q = F.linear(query, w[:e], bias[:e])
    k = F.linear(key, w[e:2 * e], bias[e:2 * e])
    v = F.linear(value, w[2 * e:], bias[2 * e:])
    b, lq, _ = q.shape
    lk, d = k.shape[1], e // heads
    q = q.view(b, lq, heads, d).transpose(1, 2)
    k = k.view(b, lk, heads, d).transpose(1, 2)
    v = v.view(b, lk, heads, d).transpose(1, 2)
    a = torch.softmax(q @ k.transpose(-1, -2) / math.sqrt(d), dim=-1)
    o = (a @ v).transpose(1, 2).reshape(b, lq, e)
    return F.linear(o, sd[p + ".out_proj.weight"], sd[p + ".out_proj.bias"])


def _ln(sd, p, x):
    return F.layer_norm(x, (x.shape[-1],), sd[p + ".weight"], sd[p + ".bias"], 1e-5)


def color_decoder(sd, feats, img_feat, layers=DEC_LAYERS):
    """MultiScaleColorDecoder.forward: feats = [out0 (1/16), out1 (1/8), out2 (1/4)], img_feat [B, 256, H, W] -> [B, 100, H, W]."""
    p = "decoder.color_decoder"
    b = img_feat.shape[0]
    src, pos = [], []
    for i, f in enumerate(feats):
        pr = F.conv2d(f, sd[f"{p}.input_proj.{i}.weight"], sd[f"{p}.input_proj.{i}.bias"])
        pr = pr.flatten(2) + sd[p + ".level_embed.weight"][i][None, :, None]
        src.append(pr.permute(0, 2, 1))                                    # [B, HW, E]
        pos.append(position_sine(b, f.shape[2], f.shape[3]).flatten(2).permute(0, 2, 1))
    qpos = sd[p + ".query_embed.weight"][None].expand(b, -1, -1)
    out = sd[p + ".query_feat.weight"][None].expand(b, -1, -1)
    for i in range(layers):
        lv = i % len(feats)
        c = f"{p}.transformer_cross_attention_layers.{i}"
        out = _ln(sd, c + ".norm", out + mha(sd, c + ".multihead_attn", out + qpos, src[lv] + pos[lv], src[lv]))
        s = f"{p}.transformer_self_attention_layers.{i}"
        qk = out + qpos
        out = _ln(sd, s + ".norm", out + mha(sd, s + ".self_attn", qk, qk, out))
        f = f"{p}.transformer_ffn_layers.{i}"
        t = F.linear(F.relu(F.linear(out, sd[f + ".linear1.weight"], sd[f + ".linear1.bias"])), sd[f + ".linear2.weight"], sd[f + ".linear2.bias"])
        out = _ln(sd, f + ".norm", out + t)
    out = _ln(sd, p + ".decoder_norm", out)
    for k in range(3):
        out = F.linear(out, sd[f"{p}.color_embed.layers.{k}.weight"], sd[f"{p}.color_embed.layers.{k}.bias"])
        if k < 2:
            out = F.relu(out)
    return torch.einsum("bqc,bchw->bqhw", out, img_feat)


def forward(sd, img, depths=DEPTHS, dec_layers=DEC_LAYERS, return_parts=False):
    """DDColor.forward: img = RGB in [0, 1], [B, 3, S, S] (S multiple of 32) -> ab [B, 2, S, S]."""
    sd = {k: torch.as_tensor(v) for k, v in sd.items()}
    x = (img - MEAN) / STD
    f0, f1, f2, f3 = encoder(sd, x, depths)
    out0 = unet_block(sd, "decoder.layers.0", f3, f2)
    out1 = unet_block(sd, "decoder.layers.1", out0, f1)
    out2 = unet_block(sd, "decoder.layers.2", out1, f0)
    out3 = custom_shuffle(sd, "decoder.last_shuf", out2, scale=4)
    logits = color_decoder(sd, [out0, out1, out2], out3, dec_layers)
    coarse = torch.cat([logits, x], dim=1)
    w = U.fold_spectral(sd, "refine_net.0.0")
    ab = F.conv2d(coarse, w, sd["refine_net.0.0.bias"])
    if return_parts:
        return dict(f0=f0, f1=f1, f2=f2, f3=f3, out0=out0, out1=out1, out2=out2, out3=out3, logits=logits, ab=ab)
    return ab


def colorize_frame(sd, frame_u8, depths=DEPTHS, dec_layers=DEC_LAYERS, input_size=None):
    """The inference wrapper (the BUILD's restatement; the reference pins none of it): L of the frame; the frame squashed to
    input_size^2 with Pillow BILINEAR when it has another size; gray RGB from Lab(L_small, 0, 0); network; ab stretched back
    (bilinear, align_corners=False); Lab(L_frame, ab) -> RGB u8, truncating cast of clip(x, 0, 1) * 255 (colorizers/util.py:52-55)."""
    from PIL import Image
    frame_u8 = np.asarray(frame_u8)
    h, w = frame_u8.shape[:2]
    S = h if input_size is None else input_size
    small = frame_u8 if (h, w) == (S, S) else np.asarray(Image.fromarray(frame_u8).resize((S, S), resample=2))
    L = Z.rgb2lab(frame_u8)[..., :1]
    Ls = Z.rgb2lab(small)[..., :1]
    gray = Z.lab2rgb(np.concatenate([Ls, np.zeros_like(Ls), np.zeros_like(Ls)], -1))              # float64 [S, S, 3] in [0, 1]
    x = torch.from_numpy(gray.astype(np.float32)).permute(2, 0, 1)[None]
    with torch.no_grad():
        ab = forward(sd, x, depths, dec_layers)
        if (h, w) != (S, S):
            ab = F.interpolate(ab, size=(h, w), mode="bilinear")
    ab = ab[0].permute(1, 2, 0).numpy().astype(np.float64)
    rgb = Z.lab2rgb(np.concatenate([L, ab], -1))
    return (np.clip(rgb, 0, 1) * 255).astype(np.uint8)
