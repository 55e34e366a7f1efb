"""Phase timeline of the pipelined conv kernel (profiling build ABL 40: wave 0 of every block leaves shader-clock stamps in the first output row
of its tile).  python tools/conv_timeline.py [shape] [batch] [cfg]   -- shape from tools/conv_bench.py; cfg 162 = bias-only, 163 = GELU epilogue.
Prints, over all tiles: prologue / main loop / wait at the epilogue barrier / LDS image / store issue / store drain, and per CU the gap between
the end of one block and the start of the next."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from vsdeoldify_amd import _native as nat
from vsdeoldify_amd.plan import PlanBuilder, WeightPack, pack_conv
from vsdeoldify_amd.render import get_context
from conv_bench import SHAPES

name = sys.argv[1] if len(sys.argv) > 1 else "ddkv_l2"
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 16
cfg = int(sys.argv[3]) if len(sys.argv) > 3 else 162
Cin, Cout, k, s, p, H, W, flags = SHAPES[name]
ctx = get_context(0)
r = np.random.default_rng(0)
pack, b = WeightPack(), PlanBuilder()
x = b.tensor(H, W, Cin)
Wt = (r.standard_normal((Cout, Cin, k, k)) / np.sqrt(Cin * k * k)).astype(np.float32)
pc = pack_conv(pack, Wt, x.cmap, x.span, bias=r.standard_normal(Cout).astype(np.float32))
BN_T = 272 if Cout == 259 else 256
y = b.tensor(H, W, Cout)
b.conv(name, pc, x, y, stride=s, pad=p, flags=flags)
ops, bufs = b.finish()
ops["reserved"] = cfg
w = nat.Weights(ctx, pack.blob())
net = nat.Net(ctx, w, ops, bufs, 0, 0, 0, batch)
net.upload(x.buf, (r.standard_normal((batch, H, W, x.span)) * 0.5).astype(np.float16))
for _ in range(3):
    ms = net.profile(batch)[0]
out = net.download(y.buf, (batch * H * W, y.cpitch), np.float16)
M, BM, BN = batch * H * W, 256, 256
rows = out[0:M:BM]                                       # first row of every row tile
recs = []
for nt in range(1 if Cout == 259 else (Cout + BN - 1) // BN):
    st = np.ascontiguousarray(rows[:, nt * BN: nt * BN + 40]).view(np.uint64)      # [tiles, 10]
    recs.append(st)
st = np.concatenate(recs).astype(np.int64)
t = st[:, [0, 8, 9, 1, 2, 3, 4, 5, 6]]                  # stamp order in time: start, index math done, stage-0 DMA issued, prologue barrier, ...
hw = st[:, 7]
cu = ((hw >> 8) & 0xF) | (((hw >> 13) & 0x7) << 4) | (((hw >> 16) & 0x3) << 7)          # CU_ID, SH_ID, SE_ID (XCC not in HW_ID: stamps differ per XCD clock anyway)
ph = np.diff(t, axis=1).astype(np.float64)
names = ["block start -> first DMA issue", "weight pieces, pixel index math, pixel pieces", "wait for stage 0 + barrier", "main loop", "wait at the epilogue barrier", "bias + convert + LDS image", "LDS read + store issue",
         "store drain (vmcnt 0)"]
tot = (t[:, -1] - t[:, 0]).astype(np.float64)
print(f"{name} batch {batch} cfg {cfg}: {ms:.3f} ms, {len(t)} tiles; s_memtime ticks (the shader clock under this load: whole-kernel ticks / kernel time) per block, mean / median")
for i, n in enumerate(names):
    print(f"  {n:48s} {ph[:, i].mean():9.1f} {np.median(ph[:, i]):9.1f}   {100 * ph[:, i].mean() / tot.mean():5.1f} %")
print(f"  {'block total':48s} {tot.mean():9.1f} {np.median(tot):9.1f}")
span = t[:, -1].max() - t[:, 0].min()
print(f"  first start -> last end {span} ticks; sum of block times / 256 CUs = {tot.sum() / 256:.0f} ticks")
