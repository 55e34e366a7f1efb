import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
print("cpus", os.cpu_count())
x = torch.randn(1, 264, 560, 560); w = torch.randn(264, 264, 3, 3)
for th in (8, 16, 32, 64, 128):
    torch.set_num_threads(th)
    F.conv2d(x, w, None, 1, 1)
    t = time.time(); F.conv2d(x, w, None, 1, 1); dt = time.time() - t
    print(th, "threads: 259-conv %.2fs -> %.1f GFLOP/s" % (dt, 2*560*560*264*264*9/dt/1e9), flush=True)
