"""Micro-benchmark of the ConvNeXt block head (depthwise 7x7 + LayerNorm) at the four DDColor stage shapes (input 512).
   python tools/dwln_bench.py [batch] [reps]     -- fused kernel vs the two-kernel form, HIP events around the ops"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from vsdeoldify_amd import _native as nat
from vsdeoldify_amd.ddcolor_net import DDColorGenerator
from vsdeoldify_amd.plan import PlanBuilder, WeightPack
from vsdeoldify_amd.render import get_context

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 16
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 7
ctx = get_context(0)
r = np.random.default_rng(0)
for C, S in ((192, 128), (384, 64), (768, 32), (1536, 16)):
    for fused in (True, False):
        pack, b = WeightPack(), PlanBuilder()
        x, mid, y = b.tensor(S, S, C), b.tensor(S, S, C), b.tensor(S, S, C)
        Wt = (r.standard_normal((C, 1, 7, 7)) / 7).astype(np.float32)
        offs = [pack.add(a) for a in (DDColorGenerator._dw_pack(Wt, x.span), np.zeros(C, np.float32), np.ones(C, np.float32), np.zeros(C, np.float32))]
        if fused:
            b.dwconv7_ln("dwln", x, y, offs[0], offs[1], x.span, offs[2], offs[3], 1e-6)
        else:
            b.dwconv7("dw", x, mid, offs[0], offs[1], x.span)
            b.layernorm("ln", mid, y, offs[2], offs[3], 1e-6)
        ops, bufs = b.finish()
        w = nat.Weights(ctx, pack.blob())
        net = nat.Net(ctx, w, ops, bufs, 0, 0, 0, batch)
        net.upload(x.buf, r.standard_normal((batch, S, S, x.cpitch)).astype(np.float16))
        for _ in range(2):
            net.profile(batch)
        ms = np.median([net.profile(batch) for _ in range(reps)], axis=0)
        tot = float(np.sum(ms))
        mb = batch * S * S * C * 2 * 2 / 1e6
        print(f"C={C:5d} {S}x{S} batch={batch} {'fused  ' if fused else 'unfused'}: {tot*1e3:8.1f} us   ({mb/tot/1e3:6.2f} TB/s of in+out, "
              f"{batch*S*S*C*49*2/tot/1e9:6.1f} TFLOP/s)", flush=True)
        net.close(); w.close()
