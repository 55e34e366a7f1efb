"""DDColor on the GPU: accuracy against the oracle at a small size and throughput / per-op profile at input_size 512
(BASELINE.json config 3).  Usage: python tools/ddcolor_bench.py [S] [batch] [--acc]"""
import os, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from vsdeoldify_amd.ddcolor import DDColorRuntime
from vsdeoldify_amd.render import get_context
from vsdeoldify_amd.synth import synth_ddcolor_state_dict

S = int(sys.argv[1]) if len(sys.argv) > 1 else 512
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 4
ctx = get_context(0)
t = time.time()
sd = synth_ddcolor_state_dict(1)
print(f"synth {time.time()-t:.1f}s"); t = time.time()
rt = DDColorRuntime(ctx, sd, precision=os.environ.get("PRECISION", "fast"))       # PRECISION=precise: the fp32-class plan
print(f"pack+upload {time.time()-t:.1f}s, {len(rt.gen.blob)/1e6:.0f} MB"); t = time.time()
net = rt.net(S, batch)
print(f"net {time.time()-t:.1f}s, {len(net.plan_ops)} ops, {net.plan_ops['flops'].sum()/1e9:.1f} GFLOP/frame")
r = np.random.default_rng(0)
frames = r.integers(0, 256, (batch, S, S, 1), dtype=np.uint8).repeat(3, -1)
for _ in range(2):
    out = rt.colorize(frames)
ts = []
for _ in range(3):
    t = time.time(); out = rt.colorize(frames); ts.append(time.time() - t)
print(f"colorize {batch} frames host->host: {min(ts)*1e3:.1f} ms -> {batch/min(ts):.1f} fps; chroma spread {np.abs(out[...,0].astype(int)-out[...,2]).mean():.1f}")
for _ in range(2):
    ms = net.profile(batch)
ms = np.mean([net.profile(batch) for _ in range(3)], axis=0)
fl = net.plan_ops["flops"].astype(np.float64) * batch
print(f"GPU ops total {ms.sum():.2f} ms for {batch} frames = {ms.sum()/batch:.2f} ms/frame, {fl.sum()/ms.sum()/1e9:.0f} TFLOP/s")
import collections
grp = collections.OrderedDict()
for i, n in enumerate(net.names):
    g = "encoder stage " + n.split(".")[3] if n.startswith("encoder.arch.stages") else ".".join(n.split(".")[:2]) if not n.startswith("decoder.color_decoder.transformer") else "colour transformer"
    a = grp.setdefault(g, [0.0, 0.0, 0]); a[0] += ms[i]; a[1] += fl[i]; a[2] += 1
PEAK_TF = 2500.0
attributed = False
for g, (tt, f, k) in grp.items():
    tf = f / tt / 1e9 if tt else 0.0
    note = ""
    if tf > PEAK_TF:                                  #  a rate above the MFMA peak is an ATTRIBUTION artefact, not a measurement: the plan books the
        note, attributed = "  [*]", True              # algorithmic FLOPs of the folded einsum + refine projection on this group's op, whose work ran inside last_shuf's epilogue
    print(f"{g:34s} {k:4d} ops {tt:8.3f} ms {100*tt/ms.sum():5.1f}%  {f/1e9:9.1f} GF {tf:7.1f} TF/s{note}")
if attributed:
    print("[*] FLOP attribution, not a rate: the einsum(bqc,bchw) + refine conv are folded into a 2 x 256 projection applied in the epilogue of decoder.last_shuf "
          "(HAVC_F_FUSE_PROJ); their algorithmic FLOPs stay booked on this group's remaining op (the shuffle + blur of the 2-channel map), whose own time is what the ms column shows")
kinds = collections.OrderedDict()
for i, n in enumerate(net.names):
    if n.startswith("encoder.arch.stages.2."):
        a = kinds.setdefault("stage2 " + n.split(".")[-1], [0.0, 0.0, 0]); a[0] += ms[i]; a[1] += fl[i]; a[2] += 1
for g, (tt, f, k) in kinds.items():
    print(f"{g:34s} {k:4d} ops {tt:8.3f} ms avg {1e3*tt/k:7.1f} us  {f/tt/1e9 if tt else 0:7.1f} TF/s  cfgs {sorted(set(c for c, n in zip(net.cfgs(), net.names) if n.startswith('encoder.arch.stages.2.') and n.endswith(g.split()[-1])))}")
order = np.argsort(-ms)
for i in order[:int(os.environ.get('TOP', '14'))]:
    o = net.plan_ops[i]
    print(f"{net.names[i]:58s} {ms[i]:8.3f} ms  t{o['type']} {o['Ci']}->{o['Npad']} k{o['kh']} {o['Hi']}x{o['Wi']}")

if os.environ.get("KINDS"):
    import re
    kk = collections.OrderedDict()
    for i, n in enumerate(net.names):
        key = re.sub(r"\.\d+", ".*", n)
        a = kk.setdefault(key, [0.0, 0]); a[0] += ms[i]; a[1] += 1
    for key, (tt, k) in sorted(kk.items(), key=lambda kv: -kv[1][0]):
        print(f"{key:70s} {k:3d} ops {tt:8.3f} ms  avg {1e3*tt/k:7.1f} us")
