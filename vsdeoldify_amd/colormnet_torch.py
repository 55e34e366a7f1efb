"""Adapters that let the REFERENCE's ColorMNet (PyTorch) run on AMD GPUs with the two pieces it cannot run there replaced by
libhavc_mi355:

  LocalGatedPropagation   colormnet/model/attention.py:712-860 as configured in model/network.py:37-45 (one head, use_linear=False):
                          the reference needs the CUDA-only `spatial_correlation_sampler` wheel (enable_corr=True) or, without it,
                          unfolds 225 shifted copies of the keys and scatters the local attention into a dense (HW)^2 matrix.  Here the
                          correlation, the masked softmax over the 15 x 15 window and the aggregation are three HIP kernels
                          (colormnet.local_attention); the depthwise 5 x 5 conv and the output Linear stay torch modules.
                          Same constructor arguments, same parameter names (a reference state_dict loads as is), same results.
  patch_reference_colormnet(network, processor): swap `network.short_term_attn` and `processor.memory` (colormnet_memory.MemoryManager).

This file is integration glue around a user's own torch-ROCm model; the product kernels are in csrc/colormnet.hip.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import colormnet as K
from .colormnet_memory import MemoryManager


class _DWConv2d(nn.Module):
    """colormnet/model/basic.py:75-94 (eval mode: the Dropout2d is inactive)"""

    def __init__(self, indim):
        super().__init__()
        self.conv = nn.Conv2d(indim, indim, 5, dilation=1, padding=2, groups=indim, bias=False)

    def forward(self, x, size_2d):
        h, w = size_2d
        _, bs, c = x.size()
        x = self.conv(x.view(h, w, bs, c).permute(2, 3, 0, 1))
        return x.view(bs, c, h * w).permute(2, 0, 1)


class LocalGatedPropagation(nn.Module):
    def __init__(self, d_qk, d_vu, num_head, dropout=0., max_dis=7, dilation=1, use_linear=True, enable_corr=True, d_att=None, use_dis=False,
                 expand_ratio=2., device_index=0):
        super().__init__()
        if num_head != 1 or use_linear or use_dis:
            raise NotImplementedError("ColorMNet's configuration only: num_head=1, use_linear=False, use_dis=False (model/network.py:37-45)")
        self.expand_d_vu = int(d_vu * expand_ratio)
        self.d_qk, self.d_vu, self.dilation, self.max_dis, self.num_head = d_qk, d_vu, dilation, max_dis, num_head
        self.window_size = 2 * max_dis + 1
        self.d_att = d_qk if d_att is None else d_att
        self.relative_emb_k = nn.Conv2d(self.d_att, self.window_size * self.window_size, kernel_size=1)
        self.dw_conv = _DWConv2d(self.expand_d_vu)
        self.projection = nn.Linear(self.expand_d_vu, d_vu)
        self.device_index = device_index

    @torch.no_grad()
    def forward(self, q, k, v, u, size_2d):
        """q, k [n, d_att, h, w], v [n, d_vu, h, w] -> (output [h*w, n, d_vu], local_attn [n, 1, 225, h*w])"""
        ws2 = self.window_size * self.window_size
        agg, attn = K.local_attention(q, k, v, self.relative_emb_k.weight.reshape(ws2, -1), self.relative_emb_k.bias, self.max_dis, self.dilation,
                                      device_index=self.device_index)
        out = self.projection(self.dw_conv(agg, size_2d))
        return out, attn


def patch_reference_colormnet(network, processor=None, device_index=0):
    """network: the reference's ColorMNet module; processor: its InferenceCore (optional).  Returns the objects it replaced."""
    old = network.short_term_attn
    new = LocalGatedPropagation(d_qk=old.d_qk, d_vu=old.d_vu, num_head=old.num_head, max_dis=old.max_dis, dilation=old.dilation, use_linear=old.use_linear,
                                d_att=old.d_att, use_dis=old.use_dis, expand_ratio=old.expand_d_vu / old.d_vu, device_index=device_index)
    new.load_state_dict({k_: v_ for k_, v_ in old.state_dict().items() if k_.split(".")[0] in ("relative_emb_k", "dw_conv", "projection")})
    new = new.to(next(old.parameters()).device).eval()
    network.short_term_attn = new
    old_mem = None
    if processor is not None:
        import sys
        old_mem = processor.memory
        processor.memory = MemoryManager(processor.config, device_index=device_index)
        mod = sys.modules.get(type(processor).__module__)          # InferenceCore.clear_memory() builds a fresh MemoryManager(config=...)
        if mod is not None and hasattr(mod, "MemoryManager"):
            mod.MemoryManager = MemoryManager
    return old, old_mem
