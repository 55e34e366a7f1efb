"""Host time of the pieces of FastMemoryManager.match_memory_into, measured with perf_counter_ns around the ctypes call and around the whole method
(cProfile attributes 270 us per call to it while the C function's own stages add up to 10 us).   python tools/c5_host_probe.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from vsdeoldify_amd import colormnet_fast as cf  # noqa: E402
from vsdeoldify_amd import _native as nat  # noqa: E402

lib = nat.load()
raw = lib.havc_memory_read_banked
acc = {"call_ns": 0, "whole_ns": 0, "n": 0, "gc": 0, "each": []}


class Timed:
    def __call__(self, *a):
        t0 = time.perf_counter_ns()
        r = raw(*a)
        dt = time.perf_counter_ns() - t0
        acc["call_ns"] += dt
        acc["each"].append((dt, a[12]))                            # (ns, N)
        return r


orig = cf.FastMemoryManager.match_memory_into


def wrapped(self, *a, **k):
    t0 = time.perf_counter_ns()
    self.ctx.lib.__dict__["havc_memory_read_banked"] = Timed()
    try:
        return orig(self, *a, **k)
    finally:
        acc["whole_ns"] += time.perf_counter_ns() - t0
        acc["n"] += 1


cf.FastMemoryManager.match_memory_into = wrapped
import gc  # noqa: E402
gc.callbacks.append(lambda phase, info: acc.__setitem__("gc", acc["gc"] + (phase == "start")))
sys.argv = ["bench.py", "--config", "c5", "--steps", "6", "--warmup", "3", "--no-cpu-baseline", "--no-extras"]
bench.main()
n = max(acc["n"], 1)
print(f"match_memory_into: {acc['n']} calls, whole method {acc['whole_ns'] / n / 1e3:.1f} us, the ctypes call alone {acc['call_ns'] / n / 1e3:.1f} us; gc runs {acc['gc']}", file=sys.stderr)
e = acc["each"][-160:]
ds = sorted(d for d, _ in e)
print("last 160 calls, us: p10 %.1f p50 %.1f p90 %.1f max %.1f" % tuple(ds[int(len(ds) * q)] / 1e3 for q in (0.1, 0.5, 0.9, 0.999)), file=sys.stderr)
print("sequence (us, N): " + " ".join(f"{d / 1e3:.0f}/{n}" for d, n in e[-60:]), file=sys.stderr)
