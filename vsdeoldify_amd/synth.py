"""Seeded synthetic DeOldify state dicts (no real weights exist in the sandbox: SURVEY.md §0.3).

`state_dict_spec(arch)` enumerates the reference's state-dict entries (name -> shape) for the wide
(resnet101, nf_factor 2) and deep (resnet34, nf_factor 1.5) generators; `synth_state_dict` fills them
with a counter-based PRNG keyed by (seed, crc32(name)) so that every consumer (tests, bench, golden
generation) sees identical weights.  Scales are chosen so activations stay O(1) through ~130 layers
and the pre-sigmoid output is not saturated; spectral / weight-norm parameters are generated so that
the eval-mode fold is NON-trivial (sigma != 1, g != |v|) yet yields a He-scaled effective weight;
attention gamma != 0 (the reference's init 0 would hide attention bugs).
"""
import zlib
from collections import OrderedDict

import numpy as np

RESNET = {"wide": ("bottleneck", [3, 4, 23, 3]), "deep": ("basic", [3, 4, 6, 3])}


def _bn(spec, p, c):
    for k in ("weight", "bias", "running_mean", "running_var"):
        spec[f"{p}.{k}"] = (c,)
    spec[f"{p}.num_batches_tracked"] = ()


def _spectral(spec, p, shape, bias=False):
    if bias:
        spec[p + ".bias"] = (shape[0],)
    spec[p + ".weight_orig"] = tuple(shape)
    spec[p + ".weight_u"] = (shape[0],)
    spec[p + ".weight_v"] = (int(np.prod(shape[1:])),)


def state_dict_spec(arch):
    """name -> shape, in the reference's state_dict() order (checked by tests/golden/spec_*.json)."""
    kind, nblk = RESNET[arch]
    deep = arch == "deep"
    spec = OrderedDict()
    e = "layers.0"
    spec[e + ".0.weight"] = (64, 3, 7, 7)
    _bn(spec, e + ".1", 64)
    inpl, exp = 64, (4 if kind == "bottleneck" else 1)
    enc_c = [64]
    for li, n in enumerate(nblk):
        planes = 64 * 2 ** li
        for bi in range(n):
            q = f"{e}.{4 + li}.{bi}"
            stride = 2 if (li > 0 and bi == 0) else 1
            if kind == "bottleneck":
                spec[q + ".conv1.weight"] = (planes, inpl, 1, 1)
                _bn(spec, q + ".bn1", planes)
                spec[q + ".conv2.weight"] = (planes, planes, 3, 3)
                _bn(spec, q + ".bn2", planes)
                spec[q + ".conv3.weight"] = (planes * 4, planes, 1, 1)
                _bn(spec, q + ".bn3", planes * 4)
            else:
                spec[q + ".conv1.weight"] = (planes, inpl, 3, 3)
                _bn(spec, q + ".bn1", planes)
                spec[q + ".conv2.weight"] = (planes, planes, 3, 3)
                _bn(spec, q + ".bn2", planes)
            if bi == 0 and (stride != 1 or inpl != planes * exp):
                spec[q + ".downsample.0.weight"] = (planes * exp, inpl, 1, 1)
                _bn(spec, q + ".downsample.1", planes * exp)
            inpl = planes * exp
        enc_c.append(inpl)
    ni = inpl                                        # 2048 / 512
    _bn(spec, "layers.1", ni)
    _spectral(spec, "layers.3.0.0", (ni * 2, ni, 3, 3)); _bn(spec, "layers.3.0.2", ni * 2)
    _spectral(spec, "layers.3.1.0", (ni, ni * 2, 3, 3)); _bn(spec, "layers.3.1.2", ni)
    skips = [enc_c[3], enc_c[2], enc_c[1], enc_c[0]]     # children 6,5,4,2
    x_c = ni
    for i, sc in enumerate(skips):
        p = f"layers.{4 + i}"
        final = i == 3
        if deep:
            up_out = x_c // 2
            cat = up_out + sc
            nf = int((cat if not final else cat // 2) * 1.5)
        else:
            n_out = 1024 if not final else 512           # nf = 512 * nf_factor(2)
            up_out = nf = n_out // 2
            cat = up_out + sc
        _spectral(spec, p + ".shuf.conv.0", (up_out * 4, x_c, 1, 1)); _bn(spec, p + ".shuf.conv.1", up_out * 4)
        _bn(spec, p + ".bn", sc)
        convs = [("conv1", cat, nf), ("conv2", nf, nf)] if deep else [("conv", cat, nf)]
        for name, ci, co in convs:
            _spectral(spec, f"{p}.{name}.0", (co, ci, 3, 3)); _bn(spec, f"{p}.{name}.2", co)
        if i == 1:                                       # sa = (i == len(sfs_idxs) - 3)
            a = f"{p}.{convs[-1][0]}.3"
            spec[a + ".gamma"] = (1,)
            for nm, co in (("query", nf // 8), ("key", nf // 8), ("value", nf)):
                _spectral(spec, f"{a}.{nm}", (co, nf, 1))
        x_c = nf
    spec["layers.8.conv.0.bias"] = (x_c * 4,)
    spec["layers.8.conv.0.weight_g"] = (x_c * 4, 1, 1, 1)
    spec["layers.8.conv.0.weight_v"] = (x_c * 4, x_c, 1, 1)
    c = x_c + 3
    _spectral(spec, "layers.10.layers.0.0", (c, c, 3, 3), bias=True)
    _spectral(spec, "layers.10.layers.1.0", (c, c, 3, 3), bias=True)
    _spectral(spec, "layers.11.0", (3, c, 1, 1), bias=True)
    return spec


def _rng(seed, name):
    return np.random.default_rng([int(seed), zlib.crc32(name.encode())])


_CACHE = {}


def synth_state_dict(arch, seed=0):
    """Deterministic in (arch, seed); cached per process (callers must treat the arrays as read-only)."""
    key = (arch, int(seed))
    if key not in _CACHE:
        _CACHE[key] = _synth_state_dict(arch, int(seed))
    return _CACHE[key]


def _synth_state_dict(arch, seed=0):
    spec = state_dict_spec(arch)
    sd = OrderedDict()
    for name, shape in spec.items():
        r = _rng(seed, name)
        leaf = name.rsplit(".", 1)[1]
        if leaf == "num_batches_tracked":
            sd[name] = np.array(1000, np.int64)
        elif leaf == "gamma":
            sd[name] = np.array([0.5], np.float32)
        elif leaf == "weight" and len(shape) == 4:                       # encoder conv, He
            fan_in = shape[1] * shape[2] * shape[3]
            sd[name] = (r.standard_normal(shape) * np.sqrt(2.0 / fan_in)).astype(np.float32)
        elif leaf in ("weight_orig", "weight_v") and len(shape) >= 3:    # spectral / weight-norm direction
            fan_in = int(np.prod(shape[1:]))
            final = name.startswith("layers.11.")
            gain = (0.1 if arch == "deep" else 0.25) if final else np.sqrt(2.0)
            wt = (r.standard_normal(shape) * gain / np.sqrt(fan_in)).astype(np.float32)
            if final:                                   # zero-mean rows: the all-positive input offset cancels
                wt -= wt.mean(axis=1, keepdims=True)
            if leaf == "weight_orig":
                sigma0 = np.float32(r.uniform(0.5, 2.0))
                sd[name] = wt * sigma0
                p = name[: -len(".weight_orig")]
                v = _rng(seed, p + ".weight_v").standard_normal(fan_in).astype(np.float32)
                v /= np.linalg.norm(v)
                t = sd[name].reshape(shape[0], -1) @ v
                sd[p + ".weight_v"] = v
                sd[p + ".weight_u"] = (t * (sigma0 / np.dot(t, t))).astype(np.float32)   # u.(W v) = sigma0
            else:                                                        # weight-norm: v = W_t * r_c, g = |W_t|
                rc = r.uniform(0.5, 2.0, size=(shape[0],) + (1,) * (len(shape) - 1)).astype(np.float32)
                sd[name] = wt * rc
                p = name[: -len(".weight_v")]
                sd[p + ".weight_g"] = np.sqrt((wt.reshape(shape[0], -1) ** 2).sum(1)).reshape(
                    (shape[0],) + (1,) * (len(shape) - 1)).astype(np.float32)
        elif leaf in ("weight_u", "weight_v", "weight_g"):
            if name not in sd:
                sd[name] = None                                          # filled by the matching weight_orig/_v
        elif leaf == "bias" and (name[:-5] + ".running_mean") not in spec:   # conv bias
            sd[name] = (r.standard_normal(shape) * 0.05).astype(np.float32)
        else:                                                            # BatchNorm parameters
            p = name.rsplit(".", 1)[0]
            post_relu = p.endswith(".2") and not p.startswith("layers.0.")   # decoder conv -> ReLU -> BN
            last_in_block = p.startswith("layers.0.") and (p.endswith(".bn3") or
                                                            (arch == "deep" and p.endswith(".bn2")))
            if leaf == "weight":
                g = r.uniform(0.8, 1.2, shape)
                sd[name] = (g * (0.15 if last_in_block else 1.0)).astype(np.float32)
            elif leaf == "bias":
                sd[name] = (r.standard_normal(shape) * 0.1).astype(np.float32)
            elif leaf == "running_mean":
                sd[name] = (r.standard_normal(shape) * 0.1 + (0.5 if post_relu else 0.0)).astype(np.float32)
            elif leaf == "running_var":
                sd[name] = (r.uniform(0.7, 1.3, shape) * (0.5 if post_relu else 1.0)).astype(np.float32)
            else:
                raise KeyError(name)
    out = OrderedDict((k, sd[k]) for k in spec)          # reference order
    assert all(v is not None for v in out.values())
    return out


# ---- Zhang et al. colorizers ------------------------------------------------------------------------------------
def _zstack(spec, p, cin, couts, bn=True, first=0, k=3):
    i = first
    for co in couts:
        spec[f"{p}.{i}.weight"] = (co, cin, k, k)
        spec[f"{p}.{i}.bias"] = (co,)
        cin = co
        i += 2
    if bn:
        _bn(spec, f"{p}.{i}", cin)
    return cin


def zhang_state_dict_spec(model):
    """name -> shape in the reference's state_dict() order (eccv16.py:9-85, siggraph17.py:7-126)."""
    spec = OrderedDict()
    if model == "eccv16":
        c = _zstack(spec, "model1", 1, [64, 64])
        c = _zstack(spec, "model2", c, [128, 128])
        c = _zstack(spec, "model3", c, [256, 256, 256])
        for m in ("model4", "model5", "model6", "model7"):
            c = _zstack(spec, m, c, [512, 512, 512])
        spec["model8.0.weight"] = (512, 256, 4, 4); spec["model8.0.bias"] = (256,)
        _zstack(spec, "model8", 256, [256, 256], bn=False, first=2)
        spec["model8.6.weight"] = (313, 256, 1, 1); spec["model8.6.bias"] = (313,)
        spec["model_out.weight"] = (2, 313, 1, 1)
        return spec
    c = _zstack(spec, "model1", 4, [64, 64])
    c = _zstack(spec, "model2", c, [128, 128])
    c = _zstack(spec, "model3", c, [256, 256, 256])
    for m in ("model4", "model5", "model6", "model7"):
        c = _zstack(spec, m, c, [512, 512, 512])
    spec["model8up.0.weight"] = (512, 256, 4, 4); spec["model8up.0.bias"] = (256,)
    _zstack(spec, "model8", 256, [256, 256], first=1)
    spec["model9up.0.weight"] = (256, 128, 4, 4); spec["model9up.0.bias"] = (128,)
    _zstack(spec, "model9", 128, [128], first=1)
    spec["model10up.0.weight"] = (128, 128, 4, 4); spec["model10up.0.bias"] = (128,)
    spec["model10.1.weight"] = (128, 128, 3, 3); spec["model10.1.bias"] = (128,)
    spec["model3short8.0.weight"] = (256, 256, 3, 3); spec["model3short8.0.bias"] = (256,)
    spec["model2short9.0.weight"] = (128, 128, 3, 3); spec["model2short9.0.bias"] = (128,)
    spec["model1short10.0.weight"] = (128, 64, 3, 3); spec["model1short10.0.bias"] = (128,)
    spec["model_class.0.weight"] = (529, 256, 1, 1); spec["model_class.0.bias"] = (529,)
    spec["model_out.0.weight"] = (2, 128, 1, 1); spec["model_out.0.bias"] = (2,)
    return spec


def synth_zhang_state_dict(model, seed=0):
    """seeded weights keeping activations O(1): He-scaled convs, BN stats matched to post-ReLU statistics."""
    key = ("zhang", model, int(seed))
    if key in _CACHE:
        return _CACHE[key]
    sd = OrderedDict()
    for name, shape in zhang_state_dict_spec(model).items():
        r = _rng(seed, "zhang." + model + "." + name)
        leaf = name.rsplit(".", 1)[1]
        if leaf == "num_batches_tracked":
            sd[name] = np.array(100, np.int64)
        elif leaf == "weight" and len(shape) == 4:
            transposed = "up.0" in name or name == "model8.0.weight"
            fan_in = (shape[0] if transposed else shape[1]) * shape[2] * shape[3]
            if transposed:
                fan_in //= 4                                      # each output pixel sees 2x2 of the 4x4 taps
            gain = np.sqrt(2.0)
            if "short" in name:
                gain = 0.7
            w = r.standard_normal(shape) * gain / np.sqrt(fan_in)
            if name == "model8.6.weight":                         # eccv16 logits: peaky 313-way softmax
                w = r.standard_normal(shape) * 4.0 / np.sqrt(fan_in)
            if name == "model_out.weight":                        # eccv16 ab bin centres / 110
                w = r.uniform(-1.0, 1.0, shape)
            if name == "model_out.0.weight":                      # siggraph17: keep tanh out of saturation
                w = r.standard_normal(shape) * 0.15 / np.sqrt(fan_in)
                w -= w.mean(axis=1, keepdims=True)
            sd[name] = w.astype(np.float32)
        elif leaf == "bias" and (name[:-5] + ".running_mean") not in zhang_state_dict_spec(model):
            sd[name] = (r.standard_normal(shape) * 0.05).astype(np.float32)
        elif leaf == "weight":
            sd[name] = r.uniform(0.8, 1.2, shape).astype(np.float32)
        elif leaf == "bias":
            sd[name] = (r.standard_normal(shape) * 0.1).astype(np.float32)
        elif leaf == "running_mean":
            sd[name] = (r.standard_normal(shape) * 0.1 + 0.5).astype(np.float32)          # BN follows a ReLU
        elif leaf == "running_var":
            sd[name] = (r.uniform(0.7, 1.3, shape) * 0.5).astype(np.float32)
        else:
            raise KeyError(name)
    _CACHE[key] = sd
    return sd


# ---- DDColor (public architecture, see oracle/ddcolor.py) ----------------------------------------------
DDCOLOR_DEPTHS, DDCOLOR_DIMS = (3, 3, 27, 3), (192, 384, 768, 1536)


def ddcolor_state_dict_spec(depths=DDCOLOR_DEPTHS, dec_layers=9, queries=100):
    """name -> shape with the key names of the public DDColor checkpoints (encoder.arch.*, decoder.*, refine_net.0.0.*)."""
    dims, hid, ffn = DDCOLOR_DIMS, 256, 2048
    spec = OrderedDict()
    e = "encoder.arch"

    def ln(p, c):
        spec[p + ".weight"], spec[p + ".bias"] = (c,), (c,)

    def lin(p, o, i, *k):
        spec[p + ".weight"], spec[p + ".bias"] = (o, i) + tuple(k), (o,)
    lin(f"{e}.downsample_layers.0.0", dims[0], 3, 4, 4)
    ln(f"{e}.downsample_layers.0.1", dims[0])
    for i in range(1, 4):
        ln(f"{e}.downsample_layers.{i}.0", dims[i - 1])
        lin(f"{e}.downsample_layers.{i}.1", dims[i], dims[i - 1], 2, 2)
    for i in range(4):
        c = dims[i]
        for j in range(depths[i]):
            p = f"{e}.stages.{i}.{j}"
            lin(p + ".dwconv", c, 1, 7, 7)
            ln(p + ".norm", c)
            lin(p + ".pwconv1", 4 * c, c)
            lin(p + ".pwconv2", c, 4 * c)
            spec[p + ".gamma"] = (c,)
    for i in range(4):
        ln(f"{e}.norm{i}", dims[i])
    up_in = dims[3]
    for li, (x_in, n_out) in enumerate(((dims[2], 512), (dims[1], 512), (dims[0], 256))):
        p, up_out = f"decoder.layers.{li}", n_out // 2
        _spectral(spec, p + ".shuf.conv.0", (up_out * 4, up_in, 1, 1)); _bn(spec, p + ".shuf.conv.1", up_out * 4)
        _bn(spec, p + ".bn", x_in)
        _spectral(spec, p + ".conv.0", (n_out, up_out + x_in, 3, 3)); _bn(spec, p + ".conv.2", n_out)
        up_in = n_out
    _spectral(spec, "decoder.last_shuf.conv.0", (256 * 16, 256, 1, 1)); _bn(spec, "decoder.last_shuf.conv.1", 256 * 16)
    d = "decoder.color_decoder"
    spec[d + ".query_feat.weight"] = (queries, hid)
    spec[d + ".query_embed.weight"] = (queries, hid)
    spec[d + ".level_embed.weight"] = (3, hid)
    for i, c in enumerate((512, 512, 256)):
        lin(f"{d}.input_proj.{i}", hid, c, 1, 1)
    for i in range(dec_layers):
        for kind, attn in (("cross", "multihead_attn"), ("self", "self_attn")):
            p = f"{d}.transformer_{kind}_attention_layers.{i}"
            spec[f"{p}.{attn}.in_proj_weight"], spec[f"{p}.{attn}.in_proj_bias"] = (3 * hid, hid), (3 * hid,)
            lin(f"{p}.{attn}.out_proj", hid, hid)
            ln(p + ".norm", hid)
        p = f"{d}.transformer_ffn_layers.{i}"
        lin(p + ".linear1", ffn, hid); lin(p + ".linear2", hid, ffn); ln(p + ".norm", hid)
    ln(d + ".decoder_norm", hid)
    for k in range(3):
        lin(f"{d}.color_embed.layers.{k}", hid, hid)
    _spectral(spec, "refine_net.0.0", (2, queries + 3, 1, 1), bias=True)
    return spec


def synth_ddcolor_state_dict(seed=0, depths=DDCOLOR_DEPTHS, dec_layers=9, queries=100):
    """Seeded synthetic DDColor weights (no checkpoint exists offline): scales keep every stage O(1) and ab inside ~+-40."""
    key = ("ddcolor", int(seed), tuple(depths), dec_layers, queries)
    if key in _CACHE:
        return _CACHE[key]
    spec = ddcolor_state_dict_spec(depths, dec_layers, queries)
    sd = OrderedDict()
    for name, shape in spec.items():
        if name in sd:
            continue
        r = _rng(seed, name)
        leaf = name.rsplit(".", 1)[1]
        parent = name.rsplit(".", 1)[0]
        is_bn = (parent + ".running_mean") in spec
        if leaf == "num_batches_tracked":
            sd[name] = np.array(1000, np.int64)
        elif is_bn:
            post_relu = parent.endswith(".conv.2")
            sd[name] = {"weight": r.uniform(0.8, 1.2, shape), "bias": r.standard_normal(shape) * 0.1,
                        "running_mean": r.standard_normal(shape) * 0.1 + (0.5 if post_relu else 0.0),
                        "running_var": r.uniform(0.7, 1.3, shape) * (0.5 if post_relu else 1.0)}[leaf].astype(np.float32)
        elif leaf == "gamma":
            sd[name] = r.uniform(0.05, 0.3, shape).astype(np.float32)
        elif leaf == "weight_orig":
            fan_in = int(np.prod(shape[1:]))
            gain = 4.0 if name.startswith("refine_net") else np.sqrt(2.0)
            wt = (r.standard_normal(shape) * gain / np.sqrt(fan_in)).astype(np.float32)
            sigma0 = np.float32(r.uniform(0.5, 2.0))
            sd[name] = wt * sigma0
            v = _rng(seed, parent + ".weight_v").standard_normal(fan_in).astype(np.float32)
            v /= np.linalg.norm(v)
            t = sd[name].reshape(shape[0], -1) @ v
            sd[parent + ".weight_v"] = v
            sd[parent + ".weight_u"] = (t * (sigma0 / np.dot(t, t))).astype(np.float32)
        elif leaf in ("weight_u", "weight_v"):
            continue                                                     # filled with weight_orig
        elif leaf == "bias" or leaf == "in_proj_bias":
            is_ln = len(spec.get(parent + ".weight", ())) == 1
            sd[name] = (r.standard_normal(shape) * (0.1 if is_ln else 0.02)).astype(np.float32)
        elif len(shape) == 1:                                            # LayerNorm weight
            sd[name] = r.uniform(0.8, 1.2, shape).astype(np.float32)
        elif "query_" in name or "level_embed" in name:
            sd[name] = r.standard_normal(shape).astype(np.float32)
        else:                                                            # conv / linear weights
            fan_in = int(np.prod(shape[1:]))
            relu_next = any(t in name for t in ("pwconv1", "linear1", "color_embed.layers.0", "color_embed.layers.1"))
            sd[name] = (r.standard_normal(shape) * np.sqrt((2.0 if relu_next else 1.0) / fan_in)).astype(np.float32)
    out = OrderedDict((k, sd[k]) for k in spec)
    _CACHE[key] = out
    return out


# ---- ColorMNet (reference modules colormnet/model/{network,modules,resnet,cbam,attention}.py) -----------------
DINO_DIM, DINO_DEPTH, DINO_GRID = 384, 12, 37


def colormnet_state_dict_spec(key_dim=64, value_dim=512, hidden_dim=64):
    """name -> shape in the order of the reference's ColorMNet.state_dict() (checked against tests/golden/spec_colormnet.json,
    which was dumped from the reference module tree; the DINOv2 backbone keys are those of the hub model, oracle/dinov2.py)."""
    spec = OrderedDict()

    def conv(p, co, ci, k, bias=True):
        spec[p + ".weight"] = (co, ci, k, k)
        if bias:
            spec[p + ".bias"] = (co,)

    def lin(p, co, ci):
        spec[p + ".weight"], spec[p + ".bias"] = (co, ci), (co,)

    def ln(p, c):
        spec[p + ".weight"], spec[p + ".bias"] = (c,), (c,)

    def trunk(p, kind, nblk, extra):
        conv(p + ".conv1", 64, 3 + extra, 7, bias=False)
        _bn(spec, p + ".bn1", 64)
        inpl, exp = 64, (4 if kind == "bottleneck" else 1)
        names = ("res2", "layer2", "layer3") if kind == "bottleneck" else ("layer1", "layer2", "layer3")
        for li, (n, name) in enumerate(zip(nblk, names)):
            planes = 64 * 2 ** li
            for bi in range(n):
                q = f"{p}.{name}.{bi}"
                if kind == "bottleneck":
                    conv(q + ".conv1", planes, inpl, 1, False); _bn(spec, q + ".bn1", planes)
                    conv(q + ".conv2", planes, planes, 3, False); _bn(spec, q + ".bn2", planes)
                    conv(q + ".conv3", planes * 4, planes, 1, False); _bn(spec, q + ".bn3", planes * 4)
                else:
                    conv(q + ".conv1", planes, inpl, 3, False); _bn(spec, q + ".bn1", planes)
                    conv(q + ".conv2", planes, planes, 3, False); _bn(spec, q + ".bn2", planes)
                if bi == 0 and (li > 0 or inpl != planes * exp):
                    conv(q + ".downsample.0", planes * exp, inpl, 1, False); _bn(spec, q + ".downsample.1", planes * exp)
                inpl = planes * exp

    def fusion(p, x_in, g_in, mid, out):
        conv(p + ".block1.downsample", mid, x_in + g_in, 3); conv(p + ".block1.conv1", mid, x_in + g_in, 3); conv(p + ".block1.conv2", mid, mid, 3)
        lin(p + ".attention.ChannelGate.mlp.1", mid // 16, mid); lin(p + ".attention.ChannelGate.mlp.3", mid, mid // 16)
        conv(p + ".attention.SpatialGate.spatial.conv", 1, 2, 7)
        if mid != out:
            conv(p + ".block2.downsample", out, mid, 3)
        conv(p + ".block2.conv1", out, mid, 3); conv(p + ".block2.conv2", out, out, 3)

    k = "key_encoder"
    trunk(k, "bottleneck", (3, 4, 6), 0)
    b = k + ".network2.backbone"
    spec[b + ".cls_token"], spec[b + ".pos_embed"], spec[b + ".mask_token"] = (1, 1, DINO_DIM), (1, 1 + DINO_GRID ** 2, DINO_DIM), (1, DINO_DIM)
    conv(b + ".patch_embed.proj", DINO_DIM, 3, 14)
    for i in range(DINO_DEPTH):
        q = f"{b}.blocks.{i}"
        ln(q + ".norm1", DINO_DIM); lin(q + ".attn.qkv", 3 * DINO_DIM, DINO_DIM); lin(q + ".attn.proj", DINO_DIM, DINO_DIM)
        spec[q + ".ls1.gamma"] = (DINO_DIM,)
        ln(q + ".norm2", DINO_DIM); lin(q + ".mlp.fc1", 4 * DINO_DIM, DINO_DIM); lin(q + ".mlp.fc2", DINO_DIM, 4 * DINO_DIM)
        spec[q + ".ls2.gamma"] = (DINO_DIM,)
    ln(b + ".norm", DINO_DIM)
    conv(k + ".network2.conv3", 4 * DINO_DIM, 4 * DINO_DIM, 1, False); _bn(spec, k + ".network2.bn3", 4 * DINO_DIM)
    for name, dim in (("fuse1", 1024), ("fuse2", 512), ("fuse3", 256)):
        p = f"{k}.{name}"
        conv(p + ".encode_enc", dim, 4 * DINO_DIM, 3)
        ln(p + ".norm1", dim); ln(p + ".norm2", dim)
        spec[p + ".crossattn.temperature"] = (8, 1, 1)
        for t in ("q", "k", "v"):
            conv(f"{p}.crossattn.to_{t}", 2 * dim, dim, 1)
            conv(f"{p}.crossattn.to_{t}_dw", 2 * dim, 1, 3)
        conv(p + ".crossattn.to_out.0", dim, 2 * dim, 1)
        ln(p + ".norm3", dim)
    v = "value_encoder"
    trunk(v, "basic", (2, 2, 2), 2)
    fusion(v + ".fuser", 1024, 256, value_dim, value_dim)
    if hidden_dim > 0:
        conv(v + ".hidden_reinforce.transform", 3 * hidden_dim, value_dim + hidden_dim, 3)
    conv("key_proj.key_proj", key_dim, 1024, 3); conv("key_proj.d_proj", 1, 1024, 3); conv("key_proj.e_proj", key_dim, 1024, 3)
    conv("short_term_attn.relative_emb_k", 225, 64, 1)
    conv("short_term_attn.dw_conv.conv", 2 * value_dim, 1, 5, False)
    lin("short_term_attn.projection", 2 * value_dim, 2 * value_dim)
    d = "decoder"
    fusion(d + ".fuser", 1024, value_dim + hidden_dim, 512, 512)
    if hidden_dim > 0:
        conv(d + ".hidden_update.g16_conv", 256, 512, 1); conv(d + ".hidden_update.g8_conv", 256, 256, 1); conv(d + ".hidden_update.g4_conv", 256, 257, 1)
        conv(d + ".hidden_update.transform", 3 * hidden_dim, 256 + hidden_dim, 3)
    conv(d + ".up_16_8.skip_conv", 512, 512, 3)
    conv(d + ".up_16_8.out_conv.downsample", 256, 512, 3); conv(d + ".up_16_8.out_conv.conv1", 256, 512, 3); conv(d + ".up_16_8.out_conv.conv2", 256, 256, 3)
    conv(d + ".up_8_4.skip_conv", 256, 256, 3)
    conv(d + ".up_8_4.out_conv.conv1", 256, 256, 3); conv(d + ".up_8_4.out_conv.conv2", 256, 256, 3)
    conv(d + ".pred", 1, 256, 3)
    return spec


def synth_colormnet_state_dict(seed=0):
    """Seeded synthetic ColorMNet weights (the checkpoint DINOv2FeatureV6_LocalAtten_s2_154000.pth cannot be fetched offline): every stage
    stays O(1), the memory softmax / channel attention / gates are neither uniform nor saturated, tanh(ab) stays in its linear range."""
    key = ("colormnet", int(seed))
    if key in _CACHE:
        return _CACHE[key]
    spec = colormnet_state_dict_spec()
    sd = OrderedDict()
    for name, shape in spec.items():
        r = _rng(seed, "colormnet." + name)
        parent, leaf = name.rsplit(".", 1)
        is_bn = (parent + ".running_mean") in spec
        n = lambda s: (r.standard_normal(shape) * s).astype(np.float32)
        fan_in = int(np.prod(shape[1:])) if len(shape) > 1 else 1
        if leaf == "num_batches_tracked":
            sd[name] = np.array(1000, np.int64)
        elif is_bn:
            last = parent.endswith((".bn3", ".downsample.1")) and "network2" not in parent or (parent.startswith("value_encoder.layer") and parent.endswith(".bn2"))
            sd[name] = {"weight": r.uniform(0.8, 1.2, shape) * (0.5 if last else 1.0), "bias": r.standard_normal(shape) * 0.1,
                        "running_mean": r.standard_normal(shape) * 0.1, "running_var": r.uniform(0.7, 1.3, shape)}[leaf].astype(np.float32)
        elif leaf == "gamma":                                            # LayerScale
            sd[name] = r.uniform(0.1, 0.5, shape).astype(np.float32)
        elif leaf == "temperature":
            sd[name] = r.uniform(4.0, 16.0, shape).astype(np.float32)
        elif leaf in ("cls_token", "mask_token"):
            sd[name] = n(0.5)
        elif leaf == "pos_embed":
            sd[name] = n(0.2)
        elif leaf == "bias":
            is_norm = len(spec[parent + ".weight"]) == 1
            sd[name] = n(0.1 if is_norm else 0.05)
        elif len(shape) == 1:                                            # LayerNorm / LayerNorm2d weight
            sd[name] = r.uniform(0.8, 1.2, shape).astype(np.float32)
        elif "_dw." in name or "dw_conv" in name:                         # depthwise 3x3 / 5x5
            sd[name] = n(1.0 / np.sqrt(shape[2] * shape[3]) * 1.5)
        elif name == "decoder.pred.weight":
            sd[name] = n(0.2 / np.sqrt(fan_in))
        elif name == "key_proj.key_proj.weight":
            sd[name] = n(0.7 / np.sqrt(fan_in))
        elif name.endswith("relative_emb_k.weight"):
            sd[name] = n(0.1)
        elif name.endswith(".conv2.weight") and (".block" in name or ".out_conv." in name):     # GroupResBlock second conv: damped residual branch
            sd[name] = n(0.5 * np.sqrt(2.0 / fan_in))
        elif any(t in name for t in (".to_q.", ".to_k.", ".to_v.", ".to_out.", ".attn.qkv", ".attn.proj", ".mlp.fc2", "projection", "d_proj", "e_proj",
                                     "transform", "g16_conv", "g8_conv", "g4_conv", "encode_enc", "mlp.3", "spatial.conv", "patch_embed")):
            sd[name] = n(1.0 / np.sqrt(fan_in))
        else:                                                            # convs / linears followed by a ReLU or GELU
            sd[name] = n(np.sqrt(2.0 / fan_in))
    _CACHE[key] = sd
    return sd
