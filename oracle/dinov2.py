"""DINOv2 ViT-S/14 (the second branch of ColorMNet's key encoder) — CPU restatement.  ORACLE / TEST INFRASTRUCTURE ONLY.

PARITY UNPINNED against the reference: `Segmentor` (/root/reference/vsdeoldify/colormnet/model/resnet.py:211-247) fetches the
backbone at run time with torch.hub.load('facebookresearch/dinov2', 'dinov2_vits14') — that repository is NOT part of the
reference tree and cannot be fetched here.  What the reference pins is only the call
    backbone.get_intermediate_layers(x, n=[8, 9, 10, 11], reshape=True)                      (resnet.py:236)
and the parameter names the checkpoint carries under `key_encoder.network2.backbone.*`.  This file restates the PUBLISHED
architecture (facebookresearch/dinov2 `DinoVisionTransformer`, vit_small, patch 14, no register tokens, LayerScale, eps 1e-6,
interpolate_offset 0.1, bicubic position-embedding interpolation without antialias) from the builder's knowledge of that public
source; it is pinned to the independent implementation in the `transformers` package of this image (Dinov2Model) on the
quantities both define (tests/test_colormnet_net.py::test_dinov2_matches_transformers).

State-dict keys (relative to the backbone): cls_token [1,1,D], pos_embed [1,1+M*M,D], mask_token [1,D] (unused at inference),
patch_embed.proj.{weight [D,3,14,14], bias}, blocks.{i}.{norm1,norm2}.{weight,bias}, blocks.{i}.attn.qkv.{weight [3D,D], bias},
blocks.{i}.attn.proj.{weight,bias}, blocks.{i}.ls1.gamma, blocks.{i}.ls2.gamma, blocks.{i}.mlp.fc1/fc2.{weight,bias}, norm.{weight,bias}.
"""
import math

import torch
import torch.nn.functional as F

PATCH, EPS, OFFSET = 14, 1e-6, 0.1


def interpolate_pos_encoding(pos_embed, h0, w0):
    """pos_embed [1, 1 + M*M, D] -> ([1, D] class position, [h0 * w0, D] patch positions, row-major).
    dinov2 `interpolate_pos_encoding`: bicubic, scale_factor = ((h0 + 0.1) / M, (w0 + 0.1) / M), antialias off."""
    pos = pos_embed.float()
    n = pos.shape[1] - 1
    m = int(math.sqrt(n))
    assert m * m == n
    d = pos.shape[-1]
    cls, patch = pos[:, 0], pos[:, 1:]
    if h0 == m and w0 == m:
        return cls, patch[0]
    grid = patch.reshape(1, m, m, d).permute(0, 3, 1, 2)
    grid = F.interpolate(grid, mode="bicubic", antialias=False, scale_factor=(float(h0 + OFFSET) / m, float(w0 + OFFSET) / m))
    assert grid.shape[-2:] == (h0, w0)
    return cls, grid.permute(0, 2, 3, 1).reshape(h0 * w0, d)


def tokens(sd, x):
    """x [B,3,H,W] (H, W multiples of 14) -> [B, 1 + h0*w0, D]: class token first, then the patches row-major, positions added"""
    b, _, h, w = x.shape
    p = F.conv2d(x, sd["patch_embed.proj.weight"], sd["patch_embed.proj.bias"], stride=PATCH)
    h0, w0 = p.shape[-2:]
    p = p.flatten(2).transpose(1, 2)
    cls_pos, patch_pos = interpolate_pos_encoding(sd["pos_embed"], h0, w0)
    cls = (sd["cls_token"].reshape(1, 1, -1) + cls_pos.reshape(1, 1, -1)).expand(b, -1, -1)
    return torch.cat([cls, p + patch_pos.unsqueeze(0)], 1)


def block(sd, p, x, heads):
    b, n, d = x.shape
    y = F.layer_norm(x, (d,), sd[p + ".norm1.weight"], sd[p + ".norm1.bias"], EPS)
    qkv = F.linear(y, sd[p + ".attn.qkv.weight"], sd[p + ".attn.qkv.bias"]).reshape(b, n, 3, heads, d // heads).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0] * (d // heads) ** -0.5, qkv[1], qkv[2]
    a = torch.softmax(q @ k.transpose(-2, -1), dim=-1)
    y = (a @ v).transpose(1, 2).reshape(b, n, d)
    y = F.linear(y, sd[p + ".attn.proj.weight"], sd[p + ".attn.proj.bias"])
    x = x + sd[p + ".ls1.gamma"] * y
    y = F.layer_norm(x, (d,), sd[p + ".norm2.weight"], sd[p + ".norm2.bias"], EPS)
    y = F.linear(F.gelu(F.linear(y, sd[p + ".mlp.fc1.weight"], sd[p + ".mlp.fc1.bias"])), sd[p + ".mlp.fc2.weight"], sd[p + ".mlp.fc2.bias"])
    return x + sd[p + ".ls2.gamma"] * y


def depth_of(sd):
    return 1 + max(int(k.split(".")[1]) for k in sd if k.startswith("blocks."))


def get_intermediate_layers(sd, x, n, heads=6, reshape=True, norm=True):
    """DinoVisionTransformer.get_intermediate_layers(x, n=[...], reshape=True): outputs of the listed blocks, final norm applied,
    class token dropped, [B, D, H/14, W/14]."""
    b, _, h, w = x.shape
    t = tokens(sd, x)
    outs = []
    for i in range(depth_of(sd)):
        t = block(sd, f"blocks.{i}", t, heads)
        if i in n:
            outs.append(t)
    if norm:
        outs = [F.layer_norm(o, (o.shape[-1],), sd["norm.weight"], sd["norm.bias"], EPS) for o in outs]
    outs = [o[:, 1:] for o in outs]
    if reshape:
        outs = [o.reshape(b, h // PATCH, w // PATCH, -1).permute(0, 3, 1, 2).contiguous() for o in outs]
    return tuple(outs)


class StandIn(torch.nn.Module):
    """An nn.Module with the hub model's parameter names and the one method the reference calls: what tools/gen_golden_colormnet_net.py
    hands to the reference's Segmentor in place of torch.hub.load (recorded in the fixtures' provenance)."""

    def __init__(self, depth=12, dim=384, heads=6, grid=37):
        super().__init__()
        self.heads, self.patch_size = heads, PATCH
        P = torch.nn.Parameter
        self.cls_token, self.pos_embed, self.mask_token = P(torch.zeros(1, 1, dim)), P(torch.zeros(1, 1 + grid * grid, dim)), P(torch.zeros(1, dim))
        self.patch_embed = torch.nn.Module()
        self.patch_embed.proj = torch.nn.Conv2d(3, dim, PATCH, PATCH)
        self.blocks = torch.nn.ModuleList()
        for _ in range(depth):
            blk = torch.nn.Module()
            blk.norm1 = torch.nn.LayerNorm(dim, EPS)                       # attribute order = the hub model's state_dict order
            blk.attn = torch.nn.Module()
            blk.attn.qkv, blk.attn.proj = torch.nn.Linear(dim, 3 * dim), torch.nn.Linear(dim, dim)
            blk.ls1 = torch.nn.Module()
            blk.ls1.gamma = P(torch.ones(dim))
            blk.norm2 = torch.nn.LayerNorm(dim, EPS)
            blk.mlp = torch.nn.Module()
            blk.mlp.fc1, blk.mlp.fc2 = torch.nn.Linear(dim, 4 * dim), torch.nn.Linear(4 * dim, dim)
            blk.ls2 = torch.nn.Module()
            blk.ls2.gamma = P(torch.ones(dim))
            self.blocks.append(blk)
        self.norm = torch.nn.LayerNorm(dim, EPS)

    def get_intermediate_layers(self, x, n=1, reshape=False, return_class_token=False, norm=True):
        assert not return_class_token
        sd = {k: v for k, v in self.state_dict().items()}
        if isinstance(n, int):
            n = list(range(len(self.blocks) - n, len(self.blocks)))
        return get_intermediate_layers(sd, x, list(n), self.heads, reshape, norm)
