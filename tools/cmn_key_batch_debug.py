"""debug: the ColorMNet key slice at batch 1 vs batch B (same frame repeated): first op whose output differs.  Usage: python tools/cmn_key_batch_debug.py [B]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from vsdeoldify_amd.colormnet_net import ColorMNetNetwork
from vsdeoldify_amd.synth import synth_colormnet_state_dict

B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
os.environ["HAVC_AUTOTUNE"] = "0"
net = ColorMNetNetwork(synth_colormnet_state_dict(3), device_index=0, autotune=False)
H, W = 112, 224
n1, nb = net._net(H, W), net._key_net(H, W, B)
g = torch.Generator().manual_seed(5)
frames = torch.tanh(torch.randn(1, 1, H, W, generator=g)).repeat(B, 3, 1, 1).numpy().astype(np.float32)
first, count, _ = n1.slices["key"]
fb, cb, bb = nb.slices["key"]
assert (first, count) == (fb, cb) and bb == B
n1.upload(n1.io["image"], frames[:1])
nb.upload(nb.io["image"], frames)
bad = 0
for i in range(count):
    o1, ob = n1.plan_ops[first + i], nb.plan_ops[first + i]
    d = int(o1["dst"])
    if d < 0:
        continue
    n1.run_ops(first, i + 1, 1)
    nb.run_ops(first, i + 1, B)
    eb = int(n1.bufs[d]["elem_bytes"]) if "elem_bytes" in n1.bufs.dtype.names else 2
    epf = int(n1.bufs[d]["elems_per_frame"])
    dt = np.float16 if eb == 2 else np.float32
    a = n1.download(d, (1, epf), dt).astype(np.float32)
    b_ = nb.download(int(ob["dst"]), (B, epf), dt).astype(np.float32)
    a = np.nan_to_num(a, nan=0, posinf=0, neginf=0); b_ = np.nan_to_num(b_, nan=0, posinf=0, neginf=0)
    scale = max(1e-6, float(np.abs(a).max()))
    errs = [float(np.abs(b_[k] - a[0]).max()) / scale for k in range(B)]
    flag = max(errs) > 5e-3
    if flag or i < 3:
        print(f"op {i:3d} {n1.names[first + i]:50s} type {int(o1['type']):2d} flags {int(o1['flags']):#x}/{int(ob['flags']):#x} absmax {scale:9.3f} rel err per frame {['%.4f' % e for e in errs]}", flush=True)
        bad += flag
        if bad >= 6:
            break
print("done, suspicious ops:", bad)
