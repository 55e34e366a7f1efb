"""Where the HOST spends a ColorMNet frame (bench.py --config c5 is host-bound: the enqueue calls of a frame take ~90 % of its wall time): cProfile over
the TIMED steps of bench_c5 (HAVC_BENCH_CPROFILE), top functions by own time and by cumulative time on stderr.   python tools/c5_host_profile.py [steps]"""
import os
import subprocess
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
steps = sys.argv[1] if len(sys.argv) > 1 else "6"
env = dict(os.environ, HAVC_BENCH_CPROFILE="1")
sys.exit(subprocess.call([sys.executable, os.path.join(root, "bench.py"), "--config", "c5", "--steps", steps, "--warmup", "3", "--no-cpu-baseline", "--no-extras"], env=env))
