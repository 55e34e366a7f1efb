"""-m gpu: the switched-off split-K form (HAVC_SPLITK_FUSED=1: the last block of a tile adds the K parts and runs the epilogue inside the conv kernel,
csrc/conv_pipe_kernel.inc) must keep producing the bytes of the default form (a second launch adds the parts).  The switch is read once per process, so
the A/B runs tools/splitk_ab.py in two child processes and compares the SHA-1 of a low-latency DeOldify frame (60+ split-K convs on both streams)."""
import os
import re
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_in_kernel_split_k_reduction_gives_the_bytes_of_the_reduce_launch():
    shas = {}
    for fused in ("0", "1"):
        env = dict(os.environ, HAVC_SPLITK_FUSED=fused, HAVC_TUNE_CACHE="0")
        p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "splitk_ab.py"), "6"], capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
        assert p.returncode == 0, p.stderr[-500:]
        m = re.search(r"sha1 ([0-9a-f]{40})", p.stdout)
        assert m, p.stdout[-300:]
        shas[fused] = m.group(1)
        print(p.stdout.strip().splitlines()[-1])
    assert shas["0"] == shas["1"], shas
