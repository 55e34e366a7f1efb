"""Import shims that let the reference's ColorMNet package (/root/reference/vsdeoldify/colormnet) run on this CPU-only build container.
Build container only (needs /root/reference); nothing here ships reference code.  On top of tools/refshim.install():

  torch.hub.load('facebookresearch/dinov2', ...)  -> oracle.dinov2.StandIn (the hub repository is not part of the reference tree and
                                                     cannot be fetched: PARITY UNPINNED, see oracle/dinov2.py)
  model_zoo.load_url / load_weights_add_extra_dim -> no download: the trunks keep their constructor initialisation (all weights are
                                                     overwritten by the seeded state dict afterwards)
  spatial_correlation_sampler.SpatialCorrelationSampler -> oracle.colormnet.local_correlation (the CUDA-only wheel; its torch fallback inside
                                                     the reference only runs when hidden_dim == d_att, attention.py:831-833)
  skimage.color.rgb2lab / lab2rgb                 -> oracle.zhang (CIE formulas; skimage absent: PARITY UNPINNED)
  torchvision.transforms                          -> Compose / Normalize / ToTensor / Resize stand-ins (plain tensor arithmetic)
  .cuda() / torch.cuda.mem_get_info               -> identity / "plenty" (the reference hard-codes CUDA placement, colormnet_render.py:146,206,232)
"""
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import refshim  # noqa: E402


def install():
    refshim.install()
    from oracle import colormnet as omem
    from oracle import dinov2, zhang

    vs = sys.modules["vapoursynth"]                               # vsslib/vsutils.py:25-31 reads these at import time
    for i, n in enumerate(("DEBUG", "INFORMATION", "WARNING", "CRITICAL", "FATAL")):
        setattr(vs, "MESSAGE_TYPE_" + n, i)

    # --- torchvision.transforms (range_transform.py, colormnet_render.py) ---
    class Compose:
        def __init__(self, ts):
            self.ts = ts

        def __call__(self, x):
            for t in self.ts:
                x = t(x)
            return x

    class Normalize:
        def __init__(self, mean, std):
            self.mean, self.std = torch.tensor(mean, dtype=torch.float32).view(-1, 1, 1), torch.tensor(std, dtype=torch.float32).view(-1, 1, 1)

        def __call__(self, x):
            return (x - self.mean) / self.std

    def _unused(name):
        def f(*a, **k):
            raise NotImplementedError(name + " is only used with image_size >= 0, which HAVC never passes (__init__.py:1700)")
        return f
    tvt = sys.modules["torchvision.transforms"]
    tvt.Compose, tvt.Normalize = Compose, Normalize
    tvt.ToTensor, tvt.Resize = _unused("ToTensor"), _unused("Resize")
    tvt.InterpolationMode = types.SimpleNamespace(BILINEAR="bilinear", NEAREST="nearest")
    sys.modules["torchvision"].transforms = tvt

    # --- skimage.color ---
    sk = types.ModuleType("skimage")
    skc = types.ModuleType("skimage.color")
    skc.rgb2lab = lambda img: zhang.rgb2lab(np.asarray(img))
    skc.lab2rgb = lambda lab: zhang.lab2rgb(lab)
    sk.color = skc
    sys.modules["skimage"], sys.modules["skimage.color"] = sk, skc

    # --- spatial_correlation_sampler ---
    class SpatialCorrelationSampler(torch.nn.Module):
        def __init__(self, kernel_size=1, patch_size=1, stride=1, padding=0, dilation=1, dilation_patch=1):
            super().__init__()
            assert kernel_size == 1 and stride == 1 and padding == 0 and dilation == 1
            self.max_dis, self.dil = (patch_size - 1) // 2, dilation_patch

        def forward(self, a, b):
            n, c, h, w = a.shape
            ws = 2 * self.max_dis + 1
            return omem.local_correlation(a, b, self.max_dis, self.dil).view(n, ws, ws, h, w)
    scs = types.ModuleType("spatial_correlation_sampler")
    scs.SpatialCorrelationSampler = SpatialCorrelationSampler
    sys.modules["spatial_correlation_sampler"] = scs

    # --- torch.hub / model_zoo / cuda placement ---
    import torch.hub as _hub
    _hub.load = lambda repo, name, **k: dinov2.StandIn()
    from torch.utils import model_zoo
    model_zoo.load_url = lambda url, **k: {}
    torch.nn.Module.cuda = lambda self, *a, **k: self
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.cuda.mem_get_info = lambda *a: (64 << 30, 64 << 30)
    torch.cuda.empty_cache = lambda: None

    ref = refshim.REF_ROOT + "/vsdeoldify"
    for pkg in (".colormnet", ".colormnet.model", ".colormnet.inference", ".colormnet.util", ".colormnet.dataset"):
        m = types.ModuleType("vsdeoldify" + pkg)
        m.__path__ = [ref + pkg.replace(".", "/")]
        sys.modules["vsdeoldify" + pkg] = m
    import importlib
    resnet = importlib.import_module("vsdeoldify.colormnet.model.resnet")
    resnet.load_weights_add_extra_dim = lambda target, source_state, extra_dim=1: None      # nothing to download, nothing to merge
    return resnet


def build_network(config=None):
    """the reference's ColorMNet (model/network.py:19-50) with constructor-initialised weights, eval mode"""
    install()
    from vsdeoldify.colormnet.model.network import ColorMNet
    cfg = {} if config is None else config
    return ColorMNet(cfg).eval()
