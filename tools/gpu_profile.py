"""Per-op GPU timing of one generator pass (HIP events around every op).  Usage:
   python tools/gpu_profile.py [arch] [S] [batch]   -> table sorted by time + totals."""
import sys, os, time
import os as _os
_os.environ.setdefault("HAVC_PRECISION", "fast")      # this tool measures the fast (fp16) mode unless told otherwise; the package default is "precise"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from vsdeoldify_amd.render import GeneratorRuntime, get_context
from vsdeoldify_amd.synth import synth_state_dict

arch = sys.argv[1] if len(sys.argv) > 1 else "wide"
S = int(sys.argv[2]) if len(sys.argv) > 2 else 560
batch = int(sys.argv[3]) if len(sys.argv) > 3 else 1
ctx = get_context(0)
t = time.time()
rt = GeneratorRuntime(ctx, synth_state_dict(arch, 1), arch, fuse_final=os.environ.get("FUSE_FINAL", "1") == "1",
                      fuse_blur=os.environ.get("FUSE_BLUR", "1") == "1", precision=os.environ.get("PRECISION", "fast"))
print(f"pack+upload {time.time()-t:.1f}s")
net = rt.net(S, batch, os.environ.get("LOW_LATENCY", "0") == "1")
for _ in range(2):
    ms = net.profile(batch)
ms = np.mean([net.profile(batch) for _ in range(3)], axis=0)
ops = net.ops
fl = ops["flops"].astype(np.float64) * batch
cfgs = net.cfgs()
order = np.argsort(-ms)
print(f"{arch} S={S} batch={batch}: total {ms.sum():.3f} ms for {fl.sum()/1e9:.1f} GFLOP -> {fl.sum()/ms.sum()/1e9:.1f} TFLOP/s, {batch/ms.sum()*1e3:.1f} passes/s")
print(f"{'op':40s} {'ms':>8s} {'%':>6s} {'GFLOP':>9s} {'TF/s':>8s}  shape")
for i in order[:int(os.environ.get("TOP", "40"))]:
    o = ops[i]
    print(f"{net.names[i]:40s} {ms[i]:8.3f} {100*ms[i]/ms.sum():6.1f} {fl[i]/1e9:9.2f} {fl[i]/ms[i]/1e9 if ms[i]>0 else 0:8.1f}  "
          f"t{o['type']} {o['Ci']}->{o['Npad']} k{o['kh']} s{o['stride']} {o['Hi']}x{o['Wi']}->{o['Ho']}x{o['Wo']} cfg{cfgs[i]}")
# group totals
import collections
grp = collections.OrderedDict()
def gname(n):
    if n.startswith("layers.0."): return "encoder (" + (".".join(n.split(".")[:3]) if n.split(".")[2] in "4567" else "stem") + ")"
    if n.startswith("layers.10") or n.startswith("layers.11"): return "tail convs"
    if n.startswith("layers.8"): return "layers.8"
    return ".".join(n.split(".")[:2])
for i in range(len(ms)):
    g = gname(net.names[i]); a = grp.setdefault(g, [0.0, 0.0, 0]); a[0] += ms[i]; a[1] += fl[i]; a[2] += 1
print("--- groups ---")
for g, (t, f, n) in grp.items():
    print(f"{g:28s} {n:3d} ops {t:8.3f} ms {100*t/ms.sum():5.1f}%  {f/1e9:9.1f} GF  {f/t/1e9 if t else 0:7.1f} TF/s")
# whole-pass wall time without per-op events
d_in = ctx.dev_alloc(batch * S * S * 3); d_out = ctx.dev_alloc(batch * S * S * 3)
ctx.dev_upload(d_in, np.random.default_rng(0).integers(0, 256, (batch, S, S, 3), dtype=np.uint8))
for _ in range(2): net.run_rgb8_dev(d_in, d_out, batch)
ts = []
for _ in range(5):
    net.run_rgb8_dev(d_in, d_out, batch); ts.append(ctx.stats().last_ms)
print("whole pass (one stream, no per-op events) ms:", ["%.3f" % x for x in ts], f"-> {fl.sum()/np.median(ts)/1e9:.1f} TFLOP/s")
