"""PCIe-inclusive rate of the clip path: host uint8 frames in -> coloured host frames out (H2D + pipeline + D2H)."""
import sys, os, time
import os as _os
_os.environ.setdefault("HAVC_PRECISION", "fast")      # this tool measures the fast (fp16) mode unless told otherwise; the package default is "precise"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from vsdeoldify_amd.clip import ClipColorizer, synthetic_gray_frame
from vsdeoldify_amd.synth import synth_state_dict
sds = {"video": synth_state_dict("wide", 1), "stable": synth_state_dict("wide", 2)}
cc = ClipColorizer("stable", 35, 0.5, state_dicts=sds, max_batch=16)
frames = np.stack([synthetic_gray_frame(i) for i in range(16)])
cc.colorize(frames)
ts = []
for _ in range(4):
    t = time.perf_counter(); out = cc.colorize(frames); ts.append(time.perf_counter() - t)
print("host->host 16 x 1080p frames: %.1f ms (%.1f fps incl. PCIe both ways, pageable host memory)" % (min(ts) * 1e3, 16 / min(ts)))
