#!/bin/bash
O=gpurun_out/r3d; mkdir -p $O
cat > /tmp/dbg.py <<'PY'
import sys, os, numpy as np, torch
sys.path.insert(0, os.getcwd())
from PIL import Image
from vsdeoldify_amd.colormnet_net import ColorMNetNetwork
from vsdeoldify_amd.colormnet_render import ColorMNetRender
from vsdeoldify_amd.synth import synth_colormnet_state_dict
net = ColorMNetNetwork(synth_colormnet_state_dict(3), autotune=(sys.argv[3] == "1"))
h, w = int(sys.argv[1]), int(sys.argv[2])
r = np.random.default_rng(0)
clip = [np.stack([np.clip(128 + 40 * r.standard_normal((h, w)), 0, 255).astype(np.uint8)] * 3, -1) for _ in range(8)]
ref = np.clip(clip[0].astype(np.float32) * [1.1, 0.9, 0.7], 0, 255).astype(np.uint8)
rnd = ColorMNetRender(vid_length=10 ** 6, network=net, reset_on_ref_update=False)
rnd.set_ref_frame(Image.fromarray(ref), False)
import vsdeoldify_amd.colormnet_net as M
orig = M.ColorMNetNetwork._run
def run(self, net_, name):
    print("  slice", name, flush=True)
    orig(self, net_, name)
    torch.cuda.synchronize()
M.ColorMNetNetwork._run = run
for t in range(int(sys.argv[4])):
    print("frame", t, flush=True)
    rnd.colorize_frame(t, Image.fromarray(clip[t % 8]))
    torch.cuda.synchronize()
    rnd.set_ref_frame(None)
    print("   mem", rnd.processor.memory.work_mem.size, flush=True)
print("done", flush=True)
PY
timeout 600 python -u /tmp/dbg.py 216 384 0 30 > $O/dbg0.txt 2>&1
tail -8 $O/dbg0.txt
timeout 600 python -u /tmp/dbg.py 216 384 1 3 > $O/dbg1.txt 2>&1
tail -8 $O/dbg1.txt
