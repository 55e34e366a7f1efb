"""ColorMNet clip on the GPU: is the HOST (Python + ctypes + torch bookkeeping) or the GPU the limit?  Time to ENQUEUE a window of frames vs time until
the GPU has finished them.  Usage: python tools/cmn_host_probe.py [frames]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from vsdeoldify_amd.clip import synthetic_gray_frame
from vsdeoldify_amd.colormnet_net import ColorMNetNetwork
from vsdeoldify_amd.colormnet_render import DeepExColorMNet
from vsdeoldify_amd.device import DeviceImage
from vsdeoldify_amd.synth import synth_colormnet_state_dict

N = int(sys.argv[1]) if len(sys.argv) > 1 else 64
net = ColorMNetNetwork(synth_colormnet_state_dict(1), device_index=0)
frames = np.stack([synthetic_gray_frame(i, 1920, 1080) for i in range(16)])
clip = DeviceImage.from_numpy(net.ctx, frames)
dx = DeepExColorMNet(vid_length=10000, render_speed="medium", network=net)
ref = np.repeat(frames[0][..., :1], 3, -1).copy(); ref[..., 0] = np.clip(ref[..., 0] * 1.2, 0, 255)
fr = lambda t0, n: [clip.frame((t0 + k) % 16) for k in range(n)]
dx.colorize_frames(fr(0, 32), {0: ref})
torch.cuda.synchronize()
for rep in range(3):
    cur = fr(32 + rep * N, N)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out = dx.colorize_frames(cur, {})
    t1 = time.perf_counter()
    torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"{N} frames: enqueue {1e3 * (t1 - t0) / N:.3f} ms per frame (host), finished {1e3 * (t2 - t0) / N:.3f} ms per frame -> {N / (t2 - t0):.1f} frames/s; "
          f"GPU still busy for {1e3 * (t2 - t1):.1f} ms after the last enqueue", flush=True)
if os.environ.get("PROFILE"):
    import cProfile, pstats
    cur = fr(1000, N)
    pr = cProfile.Profile(); pr.enable()
    out = dx.colorize_frames(cur, {})
    pr.disable(); torch.cuda.synchronize()
    st = pstats.Stats(pr); st.sort_stats("cumulative").print_stats(45)
    st.sort_stats("tottime").print_stats(25)
