"""A/B of the two split-K forms (run twice: HAVC_SPLITK_FUSED=1 and =0): prints a digest of a low-latency DeOldify frame and of a ColorMNet clip, and the
time per call.  The in-kernel reduction (the last block of a tile adds the parts in the order 0 .. S-1) must give the bytes of the separate reduce launch."""
import hashlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from PIL import Image
from vsdeoldify_amd.render import ModelImageRender
from vsdeoldify_amd.synth import synth_state_dict
from vsdeoldify_amd.clip import synthetic_gray_frame

rf = int(sys.argv[1]) if len(sys.argv) > 1 else 35
sds = {"video": synth_state_dict("wide", 1), "stable": synth_state_dict("wide", 2)}
S = rf * 16
img = Image.fromarray(np.ascontiguousarray(synthetic_gray_frame(0, 1920, 1080)[:S, :S]))
r = ModelImageRender(None, "stable", rf, 0.5, state_dicts=sds, max_batch=1, low_latency=True)
out = np.asarray(r.get_transformed_image(img))
for _ in range(3):
    r.get_transformed_image(img)
t0 = time.perf_counter()
for _ in range(30):
    r.get_transformed_image(img)
dt = (time.perf_counter() - t0) / 30
print(f"HAVC_SPLITK_FUSED={os.environ.get('HAVC_SPLITK_FUSED', '1')} rf {rf}: sha1 {hashlib.sha1(out.tobytes()).hexdigest()} {dt * 1e3:.3f} ms per call = {1 / dt:.1f} frames/s")
