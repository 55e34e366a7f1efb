"""fp32 PyTorch-CPU restatement of the DeOldify generators — ORACLE / TEST INFRASTRUCTURE ONLY.

Functional (no nn.Module graph, no fastai): takes the reference's raw state dict (key names as
produced by DynamicUnetWide/DynamicUnetDeep.state_dict()) and an imagenet-normalised NCHW fp32
tensor, returns the post-SigmoidRange NCHW tensor.  Each step cites the reference line it follows.
Pinned against the executed reference by tests/golden/unet_*.npz (tools/gen_golden.py).

  DynamicUnetWide  /root/reference/vsdeoldify/deoldify/unet.py:208-285   (video / stable, resnet101, nf_factor 2)
  DynamicUnetDeep  /root/reference/vsdeoldify/deoldify/unet.py:94-166    (artistic, resnet34, nf_factor 1.5)
"""
import torch
import torch.nn.functional as F

EPS = 1e-5
RESNET_LAYERS = {"resnet34": ("basic", [3, 4, 6, 3]), "resnet101": ("bottleneck", [3, 4, 23, 3])}


def fold_spectral(sd, p):
    """Eval-mode torch.nn.utils.spectral_norm: W = weight_orig / (u^T W_mat v) with the STORED u, v
    (no power iteration in eval) — custom_conv_layer, /root/reference/vsdeoldify/deoldify/layers.py:35-36."""
    w = sd[p + ".weight_orig"]
    wm = w.reshape(w.shape[0], -1)
    sigma = torch.dot(sd[p + ".weight_u"], torch.mv(wm, sd[p + ".weight_v"]))
    return w / sigma


def fold_weightnorm(sd, p):
    """torch.nn.utils.weight_norm (dim=0): W = g * v / ||v|| per output channel —
    PixelShuffle_ICNR, /root/reference/vsdeoldify/fastai/layers.py:204-208,119."""
    v, g = sd[p + ".weight_v"], sd[p + ".weight_g"]
    n = v.reshape(v.shape[0], -1).norm(dim=1).reshape(-1, *([1] * (v.dim() - 1)))
    return g * v / n


def bn(sd, p, x):
    return F.batch_norm(x, sd[p + ".running_mean"], sd[p + ".running_var"], sd[p + ".weight"], sd[p + ".bias"],
                        False, 0.0, EPS)


def conv_w(sd, p):
    if p + ".weight_orig" in sd:
        return fold_spectral(sd, p)
    if p + ".weight_g" in sd:
        return fold_weightnorm(sd, p)
    return sd[p + ".weight"]


def encoder(sd, x, arch):
    """torchvision resnet children()[:-2] (create_body, fastai/vision/learner.py:54-63); returns the four
    hooked skips [child 2 (relu), 4, 5, 6] (unet.py:229-231) and the layer4 output."""
    kind, nblk = RESNET_LAYERS[arch]
    p = "layers.0"
    x = F.relu(bn(sd, p + ".1", F.conv2d(x, sd[p + ".0.weight"], None, 2, 3)))
    skips = [x]
    x = F.max_pool2d(x, 3, 2, 1)
    for li, n in enumerate(nblk):
        for b in range(n):
            q = f"{p}.{4 + li}.{b}"
            stride = 2 if (li > 0 and b == 0) else 1
            idt = x
            if q + ".downsample.0.weight" in sd:
                idt = bn(sd, q + ".downsample.1", F.conv2d(x, sd[q + ".downsample.0.weight"], None, stride))
            if kind == "bottleneck":
                o = F.relu(bn(sd, q + ".bn1", F.conv2d(x, sd[q + ".conv1.weight"])))
                o = F.relu(bn(sd, q + ".bn2", F.conv2d(o, sd[q + ".conv2.weight"], None, stride, 1)))
                o = bn(sd, q + ".bn3", F.conv2d(o, sd[q + ".conv3.weight"]))
            else:
                o = F.relu(bn(sd, q + ".bn1", F.conv2d(x, sd[q + ".conv1.weight"], None, stride, 1)))
                o = bn(sd, q + ".bn2", F.conv2d(o, sd[q + ".conv2.weight"], None, 1, 1))
            x = F.relu(o + idt)
        if li < 3:
            skips.append(x)
    return skips, x


def conv_relu_bn(sd, p, x, ks=3):
    """custom_conv_layer with norm_type=Spectral, extra_bn=True: conv(no bias) -> ReLU -> BN
    (/root/reference/vsdeoldify/deoldify/layers.py:28-45; NOTE ReLU precedes BN)."""
    x = F.relu(F.conv2d(x, conv_w(sd, p + ".0"), sd.get(p + ".0.bias"), 1, ks // 2))
    return bn(sd, p + ".2", x)


def self_attention(sd, p, x):
    """fastai SelfAttention (/root/reference/vsdeoldify/fastai/layers.py:81-96): softmax over dim=1 (i),
    no 1/sqrt(d), o = gamma * (h @ beta) + x."""
    size = x.size()
    xf = x.view(*size[:2], -1)
    f = F.conv1d(xf, fold_spectral(sd, p + ".query"))
    g = F.conv1d(xf, fold_spectral(sd, p + ".key"))
    h = F.conv1d(xf, fold_spectral(sd, p + ".value"))
    beta = F.softmax(torch.bmm(f.permute(0, 2, 1).contiguous(), g), dim=1)
    o = sd[p + ".gamma"] * torch.bmm(h, beta) + xf
    return o.view(*size).contiguous()


def shuffle_blur(x):
    """relu -> PixelShuffle(2) -> ReplicationPad2d((1,0,1,0)) -> AvgPool2d(2, stride=1).  The blur is
    unconditional: `if self.blur` tests the module (unet.py:50-52; fastai/layers.py:218-220)."""
    x = F.pixel_shuffle(F.relu(x), 2)
    return F.avg_pool2d(F.pad(x, (1, 0, 1, 0), mode="replicate"), 2, stride=1)


def custom_shuffle(sd, p, x):
    """CustomPixelShuffle_ICNR (unet.py:24-52): spectral conv1x1 (no bias, no activ) -> BN -> shuffle_blur."""
    x = bn(sd, p + ".conv.1", F.conv2d(x, conv_w(sd, p + ".conv.0")))
    return shuffle_blur(x)


def unet_block(sd, p, up_in, skip, deep):
    """UnetBlockWide.forward (unet.py:196-205) / UnetBlockDeep.forward (unet.py:84-91)."""
    up = custom_shuffle(sd, p + ".shuf", up_in)
    if skip.shape[-2:] != up.shape[-2:]:
        up = F.interpolate(up, skip.shape[-2:], mode="nearest")
    cat = F.relu(torch.cat([up, bn(sd, p + ".bn", skip)], dim=1))
    if deep:
        x = conv_relu_bn(sd, p + ".conv1", cat)
        x = conv_relu_bn(sd, p + ".conv2", x)
        if p + ".conv2.3.gamma" in sd:
            x = self_attention(sd, p + ".conv2.3", x)
    else:
        x = conv_relu_bn(sd, p + ".conv", cat)
        if p + ".conv.3.gamma" in sd:
            x = self_attention(sd, p + ".conv.3", x)
    return x


def unet_forward(sd, x0, arch="wide", return_presigmoid=False):
    """Whole generator.  arch: 'wide' (resnet101 encoder) or 'deep' (resnet34 encoder)."""
    deep = arch == "deep"
    skips, x = encoder(sd, x0, "resnet34" if deep else "resnet101")
    x = F.relu(bn(sd, "layers.1", x))                       # layers.1, layers.2 (unet.py:246 / 127)
    x = conv_relu_bn(sd, "layers.3.0", x)                   # middle_conv (unet.py:236-244)
    x = conv_relu_bn(sd, "layers.3.1", x)
    for i, skip in enumerate(reversed(skips)):              # sfs_idxs reversed: children 6,5,4,2
        x = unet_block(sd, f"layers.{4 + i}", x, skip, deep)
    # layers.8: fastai PixelShuffle_ICNR, weight-norm conv1x1 + bias, no BN (fastai/layers.py:204-220)
    x = shuffle_blur(F.conv2d(x, conv_w(sd, "layers.8.conv.0"), sd["layers.8.conv.0.bias"]))
    x = torch.cat([x, x0], dim=1)                           # layers.9 MergeLayer(dense=True) (fastai/layers.py:149-152)
    # layers.10 res_block: 2x [spectral conv3x3 + bias -> ReLU], then + input (fastai/layers.py:154-161)
    r = F.relu(F.conv2d(x, conv_w(sd, "layers.10.layers.0.0"), sd["layers.10.layers.0.0.bias"], 1, 1))
    r = F.relu(F.conv2d(r, conv_w(sd, "layers.10.layers.1.0"), sd["layers.10.layers.1.0.bias"], 1, 1))
    x = x + r
    y = F.conv2d(x, conv_w(sd, "layers.11.0"), sd["layers.11.0.bias"])   # layers.11 (unet.py:277-279)
    if return_presigmoid:
        return y
    return torch.sigmoid(y) * 6.0 - 3.0                     # layers.12 SigmoidRange(-3,3) (fastai/layers.py:163-170)
