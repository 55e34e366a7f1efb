"""per-op GPU times of the ColorMNet look-ahead pass (key + skip slices at B frames per launch).  Usage: python tools/cmn_key_profile.py [B]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from vsdeoldify_amd.colormnet_net import ColorMNetNetwork
from vsdeoldify_amd.synth import synth_colormnet_state_dict

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
net = ColorMNetNetwork(synth_colormnet_state_dict(3), device_index=0)
H, W = 224, 448
n = net._key_net(H, W, B)
for _ in range(2):
    n.profile(B)
ms = np.mean([n.profile(B) for _ in range(3)], axis=0)
tot = {}
for sl in ("key", "skip"):
    first, count, _ = n.slices[sl]
    t = ms[first:first + count].sum()
    fl = float(sum(int(o["flops"]) for o in n.plan_ops[first:first + count])) * B
    print(f"slice {sl:5s} batch {B}: {t:7.3f} ms, {count} ops, {fl / 1e9:8.1f} GFLOP -> {fl / t / 1e9:7.1f} TFLOP/s = {t / B * 1e3:6.1f} us per frame")
first, count, _ = n.slices["key"]
order = np.argsort(-ms[first:first + count])[:40]
for i in order:
    o = n.plan_ops[first + i]
    print(f"   {ms[first + i] * 1e3:8.1f} us  {n.names[first + i]:56s} type {int(o['type']):2d} cfg {int(n.cfgs()[first + i]) if hasattr(n, 'cfgs') else -1:3d} splitk {(int(o['flags']) >> 16) & 15:2d}  {float(o['flops']) * B / 1e9:7.2f} GF")
