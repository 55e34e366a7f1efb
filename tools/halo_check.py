"""conv_halo_kernel (cfg 80 / 81) against conv_pipe_kernel (cfg 60 / 61) on the same packed weights: outputs must be identical
(same stage order, same MFMA sequence per accumulator).  Usage: python tools/halo_check.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from vsdeoldify_amd import _native as nat
from vsdeoldify_amd.plan import PlanBuilder, WeightPack, pack_conv
from vsdeoldify_amd.render import get_context



def run(Cin, Cout, H, W, batch, cfg, flags, res):
    ctx = get_context(0)
    r = np.random.default_rng(0)
    pack, b = WeightPack(), PlanBuilder()
    x = b.tensor(H, W, Cin)
    Wt = (r.standard_normal((Cout, Cin, 3, 3)) / np.sqrt(Cin * 9)).astype(np.float32)
    pc = pack_conv(pack, Wt, x.cmap, x.span, bias=r.standard_normal(Cout).astype(np.float32))
    y = b.tensor(H, W, Cout)
    rv = b.tensor(H, W, Cout) if res else None
    b.conv("c", pc, x, y, pad=1, flags=flags | (nat.F_RESIDUAL if res else 0), res=rv)
    ops, bufs = b.finish()
    ops["reserved"] = cfg
    w = nat.Weights(ctx, pack.blob())
    net = nat.Net(ctx, w, ops, bufs, 0, 0, 0, batch)
    xin = np.zeros((batch, H, W, x.cpitch), np.float16)
    xin[..., :Cin] = (r.standard_normal((batch, H, W, Cin)) * 0.5).astype(np.float16)
    net.upload(x.buf, xin)
    if res:
        rin = np.zeros((batch, H, W, rv.cpitch), np.float16)
        rin[..., :Cout] = (r.standard_normal((batch, H, W, Cout)) * 0.5).astype(np.float16)
        net.upload(rv.buf, rin)
    net.run_ops(0, 1, batch)
    out = net.download(y.buf, (batch, H, W, y.cpitch), np.float16)[..., :Cout].copy()
    net.close(); w.close()
    return out


SHAPES = ((256, 256, 64, 64, 2, False), (259, 259, 48, 80, 2, True), (320, 256, 40, 56, 1, False), (64, 256, 33, 47, 3, False),
          (259, 259, 35, 35, 1, True), (768, 512, 32, 32, 1, False))


def check(shape):
    Cin, Cout, H, W, batch, res = shape
    extra = 1 if (Cout + 15) // 16 * 16 % 256 == 16 else 0
    ref = run(Cin, Cout, H, W, batch, 60 + extra, nat.F_RELU_PRE, res)
    got = run(Cin, Cout, H, W, batch, 80 + extra, nat.F_RELU_PRE, res)
    return ref, got


if __name__ == "__main__":
    ok = True
    for sh in SHAPES:
        ref, got = check(sh)
        same = np.array_equal(ref, got)
        ok &= same
        d = np.abs(ref.astype(np.float32) - got.astype(np.float32))
        print(f"{sh}: identical={same} max|d|={d.max():.4g} bad={int((d > 0).sum())} finite={np.isfinite(got).all()}")
    print("OK" if ok else "MISMATCH")
    sys.exit(0 if ok else 1)
