"""Micro-benchmark of the implicit-GEMM conv kernel on the hot shapes.  Usage (GPU box):
   python tools/conv_bench.py [batch] [reps] [shape-filter]
Each shape is a 1-op plan run through the C ABI; time = HIP events around the op (havc_net_profile)."""
import sys, os
import os as _os
_os.environ.setdefault("HAVC_PRECISION", "fast")      # this tool measures the fast (fp16) mode unless told otherwise; the package default is "precise"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from vsdeoldify_amd import _native as nat
from vsdeoldify_amd.plan import PlanBuilder, WeightPack, pack_conv
from vsdeoldify_amd.render import get_context

SHAPES = {  # name: (Cin, Cout, k, stride, pad, H, W, flags)
    "tail259": (259, 259, 3, 1, 1, 560, 560, nat.F_RELU_PRE),
    "tail256": (256, 256, 3, 1, 1, 560, 560, nat.F_RELU_PRE),
    "tailA320": (320, 256, 3, 1, 1, 560, 560, nat.F_RELU_PRE),      # stagger experiment: no extra column fragment, 640-byte pitch
    "tailB272": (256, 259, 3, 1, 1, 560, 560, nat.F_RELU_PRE),      #   extra column fragment, 512-byte input pitch
    "tailC264": (259, 256, 3, 1, 1, 560, 560, nat.F_RELU_PRE),      #   no extra fragment, K = 38 stages, 640-byte pitch
    "pw1_s0": (192, 768, 1, 1, 0, 128, 128, nat.F_GELU),           # DDColor ConvNeXt MLP (pwconv1 + GELU, pwconv2 + layer scale + residual)
    "pw2_s0": (768, 192, 1, 1, 0, 128, 128, nat.F_AFFINE),
    "pw1_s1": (384, 1536, 1, 1, 0, 64, 64, nat.F_GELU),
    "pw2_s1": (1536, 384, 1, 1, 0, 64, 64, nat.F_AFFINE),
    "pw1_s2": (768, 3072, 1, 1, 0, 32, 32, nat.F_GELU),
    "pw1n_s2": (768, 3072, 1, 1, 0, 32, 32, 0),
    "pw1_s3": (1536, 6144, 1, 1, 0, 16, 16, nat.F_GELU),
    "pw2_s3": (6144, 1536, 1, 1, 0, 16, 16, nat.F_AFFINE),
    "pw2_s2": (3072, 768, 1, 1, 0, 32, 32, nat.F_AFFINE),
    "l7conv": (320, 256, 3, 1, 1, 280, 280, nat.F_RELU_PRE | nat.F_AFFINE),
    "l6conv": (768, 512, 3, 1, 1, 140, 140, nat.F_RELU_PRE | nat.F_AFFINE),
    "l5conv": (1024, 512, 3, 1, 1, 70, 70, nat.F_RELU_PRE | nat.F_AFFINE),
    "l4conv": (1536, 512, 3, 1, 1, 35, 35, nat.F_RELU_PRE | nat.F_AFFINE),
    "mid0": (2048, 4096, 3, 1, 1, 18, 18, nat.F_RELU_PRE | nat.F_AFFINE),
    "mid1": (4096, 2048, 3, 1, 1, 18, 18, nat.F_RELU_PRE | nat.F_AFFINE),
    "l8ps": (256, 1024, 1, 1, 0, 280, 280, nat.F_RELU_PRE | nat.F_OUT_PIXSHUF),
    "l7ps": (512, 1024, 1, 1, 0, 140, 140, nat.F_RELU_PRE | nat.F_OUT_PIXSHUF),
    "l8blur": (256, 1024, 1, 1, 0, 280, 280, nat.F_RELU_PRE | nat.F_OUT_PIXSHUF | nat.F_PS_BLUR),
    "l7blur": (512, 1024, 1, 1, 0, 140, 140, nat.F_RELU_PRE | nat.F_OUT_PIXSHUF | nat.F_PS_BLUR),
    "l8nops": (256, 1024, 1, 1, 0, 280, 280, nat.F_RELU_PRE),
    "l7nops": (512, 1024, 1, 1, 0, 140, 140, nat.F_RELU_PRE),
    "enc3x3_35": (256, 256, 3, 1, 1, 35, 35, nat.F_RELU_PRE),
    "enc1x1_35": (1024, 256, 1, 1, 0, 35, 35, nat.F_RELU_PRE),
    "enc3x3_140": (64, 64, 3, 1, 1, 140, 140, nat.F_RELU_PRE),
    "e3_c3_1x1": (256, 1024, 1, 1, 0, 35, 35, nat.F_RELU_POST),
    "e4_c2_3x3": (512, 512, 3, 1, 1, 18, 18, nat.F_RELU_PRE),
    "e4_c1_1x1": (2048, 512, 1, 1, 0, 18, 18, nat.F_RELU_PRE),
    "e4_c3_1x1": (512, 2048, 1, 1, 0, 18, 18, nat.F_RELU_POST),
    "e2_c2_3x3": (128, 128, 3, 1, 1, 70, 70, nat.F_RELU_PRE),
    "e2_c1_1x1": (512, 128, 1, 1, 0, 70, 70, nat.F_RELU_PRE),
    "e2_c3_1x1": (128, 512, 1, 1, 0, 70, 70, nat.F_RELU_POST),
    "e1_c3_1x1": (64, 256, 1, 1, 0, 140, 140, nat.F_RELU_POST),
    "e1_c1_1x1": (256, 64, 1, 1, 0, 140, 140, nat.F_RELU_PRE),
    "e1_c2_3x3": (64, 64, 3, 1, 1, 140, 140, nat.F_RELU_PRE),
    "stem7x7": (3, 64, 7, 2, 3, 560, 560, nat.F_RELU_PRE),
    "attn_qk": (512, 128, 1, 1, 0, 70, 70, 0),
    "ddkv2": (256, 512, 1, 1, 0, 64, 64, 0),                        # DDColor colour decoder: K / V projection of the 1/8 feature level (+res: the position term)
    "ddkv1": (256, 512, 1, 1, 0, 32, 32, 0),
    "ddkv_l2": (256, 1536, 1, 1, 0, 128, 128, 0),                  # all three layers' K / V of the 1/4 level in one GEMM (cross_kv.level2)
    "ddlast": (256, 4096, 1, 1, 0, 128, 128, nat.F_RELU_PRE),      # last_shuf without the fused projection
    "l4ps": (2048, 2048, 1, 1, 0, 18, 18, nat.F_RELU_PRE | nat.F_OUT_PIXSHUF),
    "l5ps": (512, 2048, 1, 1, 0, 35, 35, nat.F_RELU_PRE | nat.F_OUT_PIXSHUF),
}


def bench(ctx, name, batch, reps, cfg=0):
    with_res = name.endswith("+res")
    Cin, Cout, k, s, p, H, W, flags = SHAPES[name[:-4] if with_res else name]
    if with_res:
        flags |= nat.F_RESIDUAL
    r = np.random.default_rng(0)
    pack, b = WeightPack(), PlanBuilder()
    x = b.tensor(H, W, Cin)
    Wt = (r.standard_normal((Cout, Cin, k, k)) / np.sqrt(Cin * k * k)).astype(np.float32)
    kw = dict(bias=r.standard_normal(Cout).astype(np.float32))
    if flags & nat.F_AFFINE:
        kw.update(scale=np.ones(Cout, np.float32), shift=np.zeros(Cout, np.float32))
    pc = pack_conv(pack, Wt, x.cmap, x.span, pixshuf=("blur" if flags & nat.F_PS_BLUR else bool(flags & nat.F_OUT_PIXSHUF)), **kw)
    Ho = (H + 2 * p - k) // s + 1
    y = b.tensor(2 * Ho, 2 * Ho, Cout // 4) if flags & nat.F_OUT_PIXSHUF else b.tensor(Ho, Ho, Cout)
    res = b.tensor(Ho, Ho, Cout) if with_res else None
    b.conv(name, pc, x, y, stride=s, pad=p, flags=flags, res=res)
    ops, bufs = b.finish()
    ops["reserved"] = cfg
    w = nat.Weights(ctx, pack.blob())
    net = nat.Net(ctx, w, ops, bufs, 0, 0, 0, batch)
    xin = (r.standard_normal((batch, H, W, x.span)) * 0.5).astype(np.float16)
    xin[..., Cin:] = 0
    net.upload(x.buf, xin)
    if with_res:
        net.upload(res.buf, (r.standard_normal((batch, Ho, Ho, res.cpitch)) * 0.5).astype(np.float16))
    for _ in range(2):
        net.profile(batch)
    ms = float(np.median([net.profile(batch)[0] for _ in range(reps)]))
    fl = float(ops["flops"][0]) * batch
    net.close(); w.close()
    return ms, fl / ms / 1e9


if __name__ == "__main__":
    batch = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    filt = sys.argv[3] if len(sys.argv) > 3 else ""
    cfgs = [int(c) for c in sys.argv[4].split(",")] if len(sys.argv) > 4 else [0]
    ctx = get_context(0)
    names = list(SHAPES) + [n + "+res" for n in SHAPES if (n.startswith("e") and "c3" in n) or n.startswith("pw2") or n.startswith("ddkv")]
    filts = [f for f in filt.split(",") if f]
    for name in names:
        if filts and not any(f in name for f in filts):
            continue
        for cfg in cfgs:
            try:
                ms, tf = bench(ctx, name, batch, reps, cfg)
                print(f"{name:12s} cfg={cfg} batch={batch}: {ms:8.3f} ms  {tf:8.1f} TFLOP/s", flush=True)
            except Exception as e:
                print(f"{name:12s} cfg={cfg}: {type(e).__name__}: {e}", flush=True)
