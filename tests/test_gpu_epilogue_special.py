"""-m gpu: the conv kernels with their epilogue flags fixed at compile time (csrc/conv_igemm_pipe_ef.hip) against the run-time-flag kernels
(HAVC_EPI_SPECIAL=0): the same main loop and the same arithmetic, so the SAME BYTES for every layer kind and tile geometry that has a
specialised kernel.  The switch is read once per process: each arm runs in a child process."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import sys, numpy as np
sys.path.insert(0, sys.argv[1])
from vsdeoldify_amd import _native as nat
from vsdeoldify_amd.render import get_context
from tests.gpu_util import conv_op
ctx = get_context(0)
r = np.random.default_rng(3)
out = {}
KINDS = {"bias": (0, False), "relu": (nat.F_RELU_PRE, False), "relu_bn": (nat.F_RELU_PRE | nat.F_AFFINE, False), "res": (nat.F_RESIDUAL, True),
         "scale_res": (nat.F_AFFINE | nat.F_RESIDUAL, True), "res_relu": (nat.F_RESIDUAL | nat.F_RELU_POST, True), "post": (nat.F_RELU_POST, False),
         "gelu": (nat.F_GELU, False)}
for cin, cout, k, H, W in ((96, 256, 1, 40, 36), (72, 320, 3, 33, 31)):
    x = r.standard_normal((2, cin, H, W)).astype(np.float32)
    Wt = (r.standard_normal((cout, cin, k, k)) / np.sqrt(cin * k * k)).astype(np.float32)
    bias = r.standard_normal(cout).astype(np.float32)
    sc, sh = (1 + 0.1 * r.standard_normal(cout)).astype(np.float32), (0.1 * r.standard_normal(cout)).astype(np.float32)
    res = r.standard_normal((2, cout, H, W)).astype(np.float32)
    for name, (flags, with_res) in KINDS.items():
        for cfg in (60, 70, 98, 96, 72, 91, 93, 71, 90, 95, 97):
            for with_bias in ((True, False) if name in ("bias", "relu_bn") else (True,)):
                kw = dict(bias=bias if with_bias else None, pad=k // 2, flags=flags, cfg=cfg)
                if flags & nat.F_AFFINE:
                    kw.update(scale=sc, shift=sh)
                if with_res:
                    kw.update(res=res)
                out[f"{name}_{cin}_{cfg}_{int(with_bias)}"] = conv_op(ctx, x, Wt, **kw)[1]
# ReLU + PixelShuffle (Co a multiple of 64: the LDS-transposed path)
x = r.standard_normal((2, 64, 20, 24)).astype(np.float32)
Wt = (r.standard_normal((256, 64, 1, 1)) / 8).astype(np.float32)
pb = r.standard_normal(256).astype(np.float32)
for cfg in (60, 70, 97):
    out[f"ps_{cfg}"] = conv_op(ctx, x, Wt, bias=pb, flags=nat.F_RELU_PRE, pixshuf=True, cfg=cfg)[1]
np.savez(sys.argv[2], **out)
"""


def run_arm(tmp_path, special):
    path = str(tmp_path / f"arm{special}.npz")
    env = dict(os.environ, HAVC_EPI_SPECIAL=str(special), PYTHONPATH=ROOT)
    subprocess.run([sys.executable, "-c", CHILD, ROOT, path], check=True, env=env, cwd=ROOT, timeout=900)
    return np.load(path)


def test_specialised_epilogues_produce_the_bytes_of_the_run_time_flag_kernels(tmp_path):
    a, b = run_arm(tmp_path, 1), run_arm(tmp_path, 0)
    assert set(a.files) == set(b.files) and len(a.files) > 150
    bad = [k for k in a.files if not np.array_equal(a[k], b[k])]
    assert not bad, bad[:10]
    # and the tile geometries agree among themselves (every kind, both arms share one arithmetic)
    for kind in ("gelu_96", "scale_res_72", "relu_bn_96"):
        ref = a[f"{kind}_60_1"]
        for cfg in (70, 98, 96, 72, 91, 93, 71, 90, 95, 97):
            assert np.array_equal(a[f"{kind}_{cfg}_1"], ref), (kind, cfg)


def test_ddcolor_skip_norm_fusion_matches_the_separate_passes(ctx, monkeypatch):
    """encoder.norm{0,1,2} + decoder BatchNorm + ReLU as one LayerNorm launch (HAVC_DD_FUSE_SKIPNORM, default) against LayerNorm, then the affine pass:
    one fp16 rounding fewer on the skip tensors, so frames agree within 1 LSB almost everywhere."""
    from vsdeoldify_amd.ddcolor import DDColorRuntime
    from vsdeoldify_amd.synth import synth_ddcolor_state_dict
    small = dict(depths=(1, 1, 2, 1), dec_layers=3)
    sd = synth_ddcolor_state_dict(1, **small)
    r = np.random.default_rng(2)
    frames = r.integers(0, 256, (2, 64, 64, 1), dtype=np.uint8).repeat(3, -1)
    outs = []
    for fuse in ("1", "0"):
        monkeypatch.setenv("HAVC_DD_FUSE_SKIPNORM", fuse)
        rt = DDColorRuntime(ctx, sd, **small)
        try:
            names = rt.net(64, 2).names
            assert ("encoder.arch.norm0+bn" in names) == (fuse == "1") and ("decoder.layers.2.bn" in names) == (fuse == "0")
            outs.append(rt.colorize(frames))
        finally:
            rt.close()
    d = np.abs(outs[0].astype(int) - outs[1].astype(int))
    assert d.max() <= 3 and (d <= 1).mean() > 0.99, (int(d.max()), float((d <= 1).mean()))
