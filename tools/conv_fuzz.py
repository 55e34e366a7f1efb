"""Randomised check of the conv kernels against torch conv2d (fp32 on fp16-rounded operands): shapes, strides, dilations, paddings,
epilogue flags and tile configurations drawn at random.   python tools/conv_fuzz.py [n_cases] [seed]     (GPU box)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from tests import gpu_util as gu
from vsdeoldify_amd import _native as nat
from vsdeoldify_amd.render import get_context

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
ctx = get_context(0)
h16 = lambda a: a.astype(np.float16).astype(np.float32)
bad = 0
for case in range(n_cases):
    k = int(rng.choice([1, 1, 3, 3, 3, 5, 7]))
    stride = int(rng.choice([1, 1, 1, 2]))
    dil = int(rng.choice([1, 1, 1, 2])) if k > 1 else 1
    pad = int(rng.choice([0, (k - 1) // 2 * dil, 1])) if k > 1 else 0
    Cin = int(rng.choice([3, 8, 13, 24, 64, 72, 130, 256, 264, 320]))
    Cout = int(rng.choice([8, 16, 24, 64, 96, 128, 192, 256, 272, 304, 320, 512]))
    H, W, B = int(rng.integers(5, 40)), int(rng.integers(5, 40)), int(rng.integers(1, 4))
    if (H + 2 * pad - dil * (k - 1) - 1) // stride + 1 < 1 or (W + 2 * pad - dil * (k - 1) - 1) // stride + 1 < 1:
        continue
    flags = int(rng.choice([0, nat.F_RELU_PRE, nat.F_RELU_POST, nat.F_RELU_PRE | nat.F_AFFINE, nat.F_GELU]))
    with_res = bool(rng.integers(0, 2)) and stride == 1 and not (flags & nat.F_GELU)
    x = h16(rng.standard_normal((B, Cin, H, W)))
    Wt = h16(rng.standard_normal((Cout, Cin, k, k)) / np.sqrt(Cin * k * k))
    bias, sc, sh = (rng.standard_normal(Cout).astype(np.float32) for _ in range(3))
    Ho, Wo = (H + 2 * pad - dil * (k - 1) - 1) // stride + 1, (W + 2 * pad - dil * (k - 1) - 1) // stride + 1
    res = h16(rng.standard_normal((B, Cout, Ho, Wo))) if with_res else None
    kw = dict(bias=bias, stride=stride, pad=pad, dil=dil, flags=flags, res=res)
    if flags & nat.F_AFFINE:
        kw.update(scale=sc, shift=sh)
    y = torch.nn.functional.conv2d(torch.from_numpy(x), torch.from_numpy(Wt), torch.from_numpy(bias), stride=stride, padding=pad, dilation=dil)
    if flags & nat.F_RELU_PRE:
        y = torch.relu(y)
    if flags & nat.F_GELU:
        y = torch.nn.functional.gelu(y)
    if flags & nat.F_AFFINE:
        y = y * torch.from_numpy(sc).view(1, -1, 1, 1) + torch.from_numpy(sh).view(1, -1, 1, 1)
    if with_res:
        y = y.half().float() + torch.from_numpy(res)
    if flags & nat.F_RELU_POST:
        y = torch.relu(y)
    ref = y.numpy()
    cfgs = [0] + [int(c) for c in rng.choice([1, 2, 3, 7, 60, 70, 71, 72, 90, 91, 93, 95, 96, 97, 98, 99, 92], 3, replace=False)]
    base = None
    for cfg in cfgs:
        try:
            got, raw = gu.conv_op(ctx, x, Wt, cfg=cfg, **kw)
        except Exception as e:
            if cfg == 0:
                print(f"case {case}: cfg 0 FAILED {type(e).__name__}: {e} | k{k} s{stride} d{dil} p{pad} {Cin}->{Cout} {H}x{W} b{B} flags {flags:#x} res {with_res}")
                bad += 1
            continue                                             # a forced tile that does not apply to this shape: refused, fine
        err = np.abs(got - ref)
        tol = 4e-3 * max(1.0, float(np.abs(ref).max()))
        if err.max() > tol:
            print(f"case {case} cfg {cfg}: max err {err.max():.4g} (tol {tol:.3g}) | k{k} s{stride} d{dil} p{pad} {Cin}->{Cout} {H}x{W} b{B} flags {flags:#x} res {with_res}")
            bad += 1
        if base is None:
            base = raw
        elif not np.array_equal(raw.view(np.uint16), base.view(np.uint16)):
            print(f"case {case} cfg {cfg}: bytes differ from cfg {cfgs[0]} | k{k} s{stride} d{dil} p{pad} {Cin}->{Cout} {H}x{W} b{B} flags {flags:#x} res {with_res}")
            bad += 1
print(f"{n_cases} cases, {bad} problems")
