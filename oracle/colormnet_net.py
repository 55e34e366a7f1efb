"""ColorMNet network (BASELINE configs[4]) — functional fp32 CPU restatement on the RAW reference state dict.
ORACLE / TEST INFRASTRUCTURE ONLY (tests/, __graft_entry__.smoke(), bench.py's cpu_baseline leg).

Citations are to /root/reference/vsdeoldify/colormnet/:
  encode_key      model/network.py:52-85  -> KeyEncoder_DINOv2_v6 (model/modules.py:158-196: ResNet50 trunk model/resnet.py:124-195,
                  Segmentor :211-247, Fuse :370-398, CrossChannelAttention :286-331, LayerNorm2d :249-284) + KeyProjection (modules.py:213-231)
  encode_value    model/network.py:87-101 -> ValueEncoder (modules.py:105-156: ResNet18 with 2 extra input planes, FeatureFusionBlock :22-41,
                  CBAM model/cbam.py:66-77, HiddenReinforcer modules.py:80-103)
  segment         model/network.py:137-145 -> Decoder (modules.py:233-271: FeatureFusionBlock, UpsampleBlock :197-211, HiddenUpdater :44-78,
                  GroupResBlock / up- / down-sample_groups model/group_modules.py:14-68)
  short_term_attn model/attention.py:712-860 (LocalGatedPropagation, configured as network.py:37-45) + DWConv2d model/basic.py:75-94
  frame wrapper   colormnet_render.py:197-301 (colorize_frame / get_image), dataset/range_transform.py:24-47, colormnet_utils.py:185-197

Pinned by tests/golden/colormnet_net_*.npz: outputs of the reference's own modules EXECUTED in the build container on seeded weights
(tools/gen_golden_colormnet_net.py).  Two stand-ins are part of those fixtures' provenance and stay PARITY UNPINNED:
  * the DINOv2 backbone (torch.hub content, not in the reference tree): oracle/dinov2.py, pinned to transformers.Dinov2Model;
  * skimage.color.rgb2lab / lab2rgb (skimage absent): oracle/zhang.py's restatement of the CIE formulas.
"""
import math

import torch
import torch.nn.functional as F

from . import colormnet as mem
from . import dinov2

BN_EPS = 1e-5


def _bn(sd, p, x):
    return F.batch_norm(x, sd[p + ".running_mean"], sd[p + ".running_var"], sd[p + ".weight"], sd[p + ".bias"], False, 0.0, BN_EPS)


def _conv(sd, p, x, stride=1, pad=0, groups=1):
    return F.conv2d(x, sd[p + ".weight"], sd.get(p + ".bias"), stride=stride, padding=pad, groups=groups)


# ---- ResNet trunks (model/resnet.py:50-166; stride on the 3x3 of a Bottleneck) ----
def _bottleneck(sd, p, x, stride):
    out = F.relu(_bn(sd, p + ".bn1", _conv(sd, p + ".conv1", x)))
    out = F.relu(_bn(sd, p + ".bn2", _conv(sd, p + ".conv2", out, stride=stride, pad=1)))
    out = _bn(sd, p + ".bn3", _conv(sd, p + ".conv3", out))
    if p + ".downsample.0.weight" in sd:
        x = _bn(sd, p + ".downsample.1", _conv(sd, p + ".downsample.0", x, stride=stride))
    return F.relu(out + x)


def _basic(sd, p, x, stride):
    out = F.relu(_bn(sd, p + ".bn1", _conv(sd, p + ".conv1", x, stride=stride, pad=1)))
    out = _bn(sd, p + ".bn2", _conv(sd, p + ".conv2", out, pad=1))
    if p + ".downsample.0.weight" in sd:
        x = _bn(sd, p + ".downsample.1", _conv(sd, p + ".downsample.0", x, stride=stride))
    return F.relu(out + x)


def _layer(sd, p, x, block, stride):
    i = 0
    while f"{p}.{i}.conv1.weight" in sd:
        x = block(sd, f"{p}.{i}", x, stride if i == 0 else 1)
        i += 1
    return x


# ---- key encoder ----
def layernorm2d(sd, p, x, eps=1e-6):
    """resnet.py:249-284: over the channels of every pixel, biased variance"""
    mu = x.mean(1, keepdim=True)
    var = (x - mu).pow(2).mean(1, keepdim=True)
    return sd[p + ".weight"].view(1, -1, 1, 1) * ((x - mu) / (var + eps).sqrt()) + sd[p + ".bias"].view(1, -1, 1, 1)


def cross_channel_attention(sd, p, enc, dec, heads=8):
    """resnet.py:286-331: attention ACROSS CHANNELS (c x c per head), queries from `enc`, keys / values from `dec`"""
    b, c, h, w = enc.shape

    def branch(n, x):
        y = _conv(sd, f"{p}.to_{n}", x)
        return _conv(sd, f"{p}.to_{n}_dw", y, pad=1, groups=y.shape[1]).reshape(b, heads, -1, h * w)
    q, k, v = branch("q", enc), branch("k", dec), branch("v", dec)
    q, k = F.normalize(q, dim=-1), F.normalize(k, dim=-1)
    attn = ((q @ k.transpose(-2, -1)) * sd[p + ".temperature"]).softmax(dim=-1)
    out = (attn @ v).reshape(b, -1, h, w)
    return _conv(sd, p + ".to_out.0", out)


def fuse(sd, p, enc, dnc):
    """resnet.py:370-398"""
    enc = _conv(sd, p + ".encode_enc", enc, pad=1)
    res = enc
    out = cross_channel_attention(sd, p + ".crossattn", layernorm2d(sd, p + ".norm1", enc), layernorm2d(sd, p + ".norm2", dnc)) + res
    return F.relu(layernorm2d(sd, p + ".norm3", out))


def segmentor(sd, p, x):
    """resnet.py:211-247: DINOv2 blocks 8-11 -> 1x1 conv + BN + ReLU -> bilinear to (int(h*14/16), int(w*14/16))"""
    bb = {k[len(p) + 10:]: v for k, v in sd.items() if k.startswith(p + ".backbone.")}
    f16 = torch.cat(dinov2.get_intermediate_layers(bb, x, [8, 9, 10, 11]), dim=1)
    f16 = F.relu(_bn(sd, p + ".bn3", _conv(sd, p + ".conv3", f16)))
    new = (int(f16.shape[2] * 14 / 16), int(f16.shape[3] * 14 / 16))
    return F.interpolate(f16, size=new, mode="bilinear", align_corners=False)


def key_encoder(sd, f, p="key_encoder"):
    """modules.py:158-196 -> (g16 [1024], g8 [512], g4 [256])"""
    x = F.relu(_bn(sd, p + ".bn1", _conv(sd, p + ".conv1", f, stride=2, pad=3)))
    x = F.max_pool2d(x, 3, 2, 1)
    f4 = _layer(sd, p + ".res2", x, _bottleneck, 1)
    f8 = _layer(sd, p + ".layer2", f4, _bottleneck, 2)
    f16 = _layer(sd, p + ".layer3", f8, _bottleneck, 2)
    dino = segmentor(sd, p + ".network2", f)
    up = lambda t, s: F.interpolate(t, scale_factor=s, mode="bilinear")        # nn.Upsample(scale_factor, 'bilinear'): align_corners False
    return fuse(sd, p + ".fuse1", dino, f16), fuse(sd, p + ".fuse2", up(dino, 2), f8), fuse(sd, p + ".fuse3", up(dino, 4), f4)


def key_projection(sd, x, need_s, need_e, p="key_proj"):
    """modules.py:213-231"""
    shrinkage = _conv(sd, p + ".d_proj", x, pad=1) ** 2 + 1 if need_s else None
    selection = torch.sigmoid(_conv(sd, p + ".e_proj", x, pad=1)) if need_e else None
    return _conv(sd, p + ".key_proj", x, pad=1), shrinkage, selection


def encode_key(sd, frame, need_sk=True, need_ek=True):
    """network.py:52-85 for a [B,3,H,W] frame"""
    f16, f8, f4 = key_encoder(sd, frame)
    key, shrinkage, selection = key_projection(sd, f16, need_sk, need_ek)
    return key, shrinkage, selection, f16, f8, f4


# ---- group modules (model/group_modules.py) : features [B, objects, C, H, W] ----
def _g(fn, g):
    b, o = g.shape[:2]
    y = fn(g.flatten(0, 1))
    return y.view(b, o, *y.shape[1:])


def group_res_block(sd, p, g):
    out = _g(lambda t: _conv(sd, p + ".conv1", F.relu(t), pad=1), g)
    out = _g(lambda t: _conv(sd, p + ".conv2", F.relu(t), pad=1), out)
    if p + ".downsample.weight" in sd:
        g = _g(lambda t: _conv(sd, p + ".downsample", t, pad=1), g)
    return out + g


def cbam(sd, p, x):
    """model/cbam.py: channel gate (avg + max pooled MLP, sigmoid) then spatial gate (7x7 conv on [max_c, mean_c], sigmoid)"""
    mlp = lambda v: F.linear(F.relu(F.linear(v, sd[p + ".ChannelGate.mlp.1.weight"], sd[p + ".ChannelGate.mlp.1.bias"])),
                             sd[p + ".ChannelGate.mlp.3.weight"], sd[p + ".ChannelGate.mlp.3.bias"])
    att = mlp(x.mean((2, 3))) + mlp(x.amax((2, 3)))
    x = x * torch.sigmoid(att)[:, :, None, None]
    comp = torch.cat([x.max(1, keepdim=True)[0], x.mean(1, keepdim=True)], 1)
    return x * torch.sigmoid(_conv(sd, p + ".SpatialGate.spatial.conv", comp, pad=3))


def feature_fusion(sd, p, x, g):
    """modules.py:22-41"""
    b, o = g.shape[:2]
    g = torch.cat([x.unsqueeze(1).expand(-1, o, -1, -1, -1), g], 2)
    g = group_res_block(sd, p + ".block1", g)
    r = cbam(sd, p + ".attention", g.flatten(0, 1)).view(b, o, *g.shape[2:])
    return group_res_block(sd, p + ".block2", g + r)


def _gru(values, h, hd):
    f = torch.sigmoid(values[:, :, :hd])
    u = torch.sigmoid(values[:, :, hd:2 * hd])
    n = torch.tanh(values[:, :, 2 * hd:])
    return f * h * (1 - u) + u * n


def encode_value(sd, frame, f16, h16, masks, is_deep_update=True, p="value_encoder"):
    """network.py:87-101 + modules.py:125-156.  frame [B,3,H,W], masks [B,objects,H,W] -> (g16 [B,obj,CV,h,w], h16)"""
    o = masks.shape[1]
    if o != 1:
        others = torch.cat([masks[:, [j for j in range(o) if j != i]].sum(1, keepdim=True) for i in range(o)], 1)
    else:
        others = torch.zeros_like(masks)
    g = torch.stack([masks, others], 2)
    g = torch.cat([frame.unsqueeze(1).expand(-1, o, -1, -1, -1), g], 2)
    b = g.shape[0]
    g = g.flatten(0, 1)
    g = _bn(sd, p + ".bn1", _conv(sd, p + ".conv1", g, stride=2, pad=3))
    g = F.relu(F.max_pool2d(g, 3, 2, 1))
    g = _layer(sd, p + ".layer1", g, _basic, 1)
    g = _layer(sd, p + ".layer2", g, _basic, 2)
    g = _layer(sd, p + ".layer3", g, _basic, 2)
    g = F.interpolate(g, f16.shape[2:], mode="bilinear", align_corners=False)
    g = feature_fusion(sd, p + ".fuser", f16, g.view(b, o, *g.shape[1:]))
    if is_deep_update and p + ".hidden_reinforce.transform.weight" in sd:
        hd = h16.shape[2]
        v = _g(lambda t: _conv(sd, p + ".hidden_reinforce.transform", t, pad=1), torch.cat([g, h16], 2))
        h16 = _gru(v, h16, hd)
    return g, h16


def _interp_groups(g, ratio, mode):
    return _g(lambda t: F.interpolate(t, scale_factor=ratio, mode=mode, align_corners=(False if mode == "bilinear" else None)), g)


def upsample_block(sd, p, skip_f, up_g):
    """modules.py:197-211"""
    skip = _conv(sd, p + ".skip_conv", skip_f, pad=1)
    g = _interp_groups(up_g, 2, "bilinear")
    return group_res_block(sd, p + ".out_conv", skip.unsqueeze(1) + g)


def segment(sd, feats, memory_readout, hidden, h_out=True, p="decoder"):
    """network.py:137-145 + modules.py:247-271 -> (hidden or None, prob [B,objects,H,W] = tanh(logits))"""
    f16, f8, f4 = feats
    b, o = memory_readout.shape[:2]
    has_hidden = p + ".hidden_update.transform.weight" in sd
    g16 = feature_fusion(sd, p + ".fuser", f16, torch.cat([memory_readout, hidden], 2) if has_hidden else memory_readout)
    g8 = upsample_block(sd, p + ".up_16_8", f8, g16)
    g4 = upsample_block(sd, p + ".up_8_4", f4, g8)
    logits = _conv(sd, p + ".pred", F.relu(g4.flatten(0, 1)), pad=1)
    if h_out and has_hidden:
        g4 = torch.cat([g4, logits.view(b, o, 1, *logits.shape[-2:])], 2)
        q = p + ".hidden_update"
        g = _g(lambda t: _conv(sd, q + ".g16_conv", t), g16) + _g(lambda t: _conv(sd, q + ".g8_conv", t), _interp_groups(g8, 1 / 2, "area")) + \
            _g(lambda t: _conv(sd, q + ".g4_conv", t), _interp_groups(g4, 1 / 4, "area"))
        hd = hidden.shape[2]
        hidden = _gru(_g(lambda t: _conv(sd, q + ".transform", t, pad=1), torch.cat([g, hidden], 2)), hidden, hd)
    else:
        hidden = None
    logits = F.interpolate(logits, scale_factor=4, mode="bilinear", align_corners=False).view(b, o, *[4 * s for s in logits.shape[-2:]])
    return hidden, torch.tanh(logits)


def short_term_attn(sd, q, k, v, size_2d, p="short_term_attn", max_dis=7):
    """LocalGatedPropagation.forward (attention.py:783-860; one head, use_linear=False): local attention, depthwise 5x5, Linear.
    q, k [n,64,h,w], v [n,CV,h,w] -> ([h*w, n, CV], local_attn)"""
    h, w = size_2d
    ws2 = (2 * max_dis + 1) ** 2
    agg, attn = mem.local_attention(q, k, v, sd[p + ".relative_emb_k.weight"].reshape(ws2, -1), sd[p + ".relative_emb_k.bias"], max_dis, 1)
    n, c = agg.shape[1:]
    x = agg.view(h, w, n, c).permute(2, 3, 0, 1)
    x = F.conv2d(x, sd[p + ".dw_conv.conv.weight"], None, padding=2, groups=c)
    x = x.reshape(n, c, h * w).permute(2, 0, 1)
    return F.linear(x, sd[p + ".projection.weight"], sd[p + ".projection.bias"]), attn


class Network:
    """the four entry points InferenceCore drives (inference/inference_core.py), as an object over a state dict"""

    def __init__(self, sd):
        self.sd = {k: (v if torch.is_tensor(v) else torch.as_tensor(v)) for k, v in sd.items()}
        self.key_dim = self.sd["key_proj.key_proj.weight"].shape[0]
        self.value_dim = self.sd["value_encoder.fuser.block2.conv2.weight"].shape[0]
        self.hidden_dim = self.sd["decoder.hidden_update.transform.weight"].shape[0] // 3 if "decoder.hidden_update.transform.weight" in self.sd else 0
        self.calls = []

    def encode_key(self, frame, need_ek=True, need_sk=True):
        return encode_key(self.sd, frame, need_sk, need_ek)

    def encode_value(self, frame, f16, h16, masks, is_deep_update=True):
        return encode_value(self.sd, frame, f16, h16, masks, is_deep_update)

    def segment(self, feats, memory_readout, hidden, selector=None, h_out=True, strip_bg=True):
        hidden, prob = segment(self.sd, feats, memory_readout, hidden, h_out)
        return hidden, prob, prob

    def short_term_attn(self, q, k, v, u, size_2d):
        return short_term_attn(self.sd, q, k, v, size_2d)


# ---- the frame wrapper (colormnet_render.py:197-301 with image_size = -1, the only value HAVC passes: __init__.py:1700) ----
def frame_to_lab_tensor(rgb_u8):
    """get_image's im_transform: RGB2Lab (float32 of skimage rgb2lab) -> to_mytensor -> Normalize(mean [50,0,0], std [50,110,110])"""
    from . import zhang
    lab = torch.from_numpy(zhang.rgb2lab(rgb_u8).astype("float32")).permute(2, 0, 1)
    return (lab - torch.tensor([50.0, 0, 0]).view(3, 1, 1)) / torch.tensor([50.0, 110, 110]).view(3, 1, 1)


def lab_tensor_to_rgb(l_plane, ab):
    """lab2rgb_transform_PIL (colormnet_utils.py:185-197) + `* 255 -> uint8` (colormnet_render.py:277-279)"""
    from . import zhang
    lab = torch.cat([l_plane, ab], 0)
    lab = (lab - torch.tensor([-1.0, 0, 0]).view(3, 1, 1)) / torch.tensor([1 / 50., 1 / 110., 1 / 110.]).view(3, 1, 1)   # inv_lll2rgb_trans
    rgb = zhang.lab2rgb(lab.permute(1, 2, 0).numpy().astype("float32")).clip(0, 1)
    return (rgb * 255).astype("uint8")
