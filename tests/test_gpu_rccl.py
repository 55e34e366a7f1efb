"""-m gpu: the RCCL leg of the clip runner on ONE GPU: backend "nccl" (= RCCL on ROCm) initialises with a single rank, so a
1-GPU box can run sharded.init_dist's nccl branch, the scatter / gather of colorize_clip_sharded on DEVICE tensors and DeviceClipFn's
event ordering between torch's stream and the library's.  Runs in a child process (its own process group, MASTER_ADDR 127.0.0.1)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import os, sys
import numpy as np
sys.path.insert(0, os.getcwd())
import torch
from vsdeoldify_amd import sharded
from vsdeoldify_amd.clip import ClipColorizer
from vsdeoldify_amd.synth import synth_state_dict
dist = sharded.init_dist("nccl", 0)
assert dist.get_backend() == "nccl" and dist.get_world_size() == 1
t = torch.ones(4, device="cuda:0")
dist.all_reduce(t)                                   # RCCL is alive
assert float(t.sum()) == 4.0
sds = {"video": synth_state_dict("wide", 1)}
cc = ClipColorizer("video", 4, 0, device_index=0, state_dicts=sds, max_batch=2)
r = np.random.default_rng(0)
frames = np.clip(128 + 40 * r.standard_normal((3, 72, 96, 1)), 0, 255).astype(np.uint8).repeat(3, -1)
want = cc.colorize(frames)
fn = sharded.DeviceClipFn(cc)
dev = torch.from_numpy(frames).to("cuda:0")
for rep in range(3):                                 # the same clip through scatter -> havc_colorize_clip (device pointers) -> gather, three times
    got = sharded.colorize_clip_sharded(dev, fn, dist, 0, 1, "cuda:0", force_collectives=True)
    assert got.is_cuda and np.array_equal(got.cpu().numpy(), want), rep
dist.barrier()
dist.destroy_process_group()
print("RCCL-OK")
'''


def test_rccl_single_rank_clip_runner():
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29811", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, "-c", CHILD], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and "RCCL-OK" in p.stdout, (p.stdout[-2000:], p.stderr[-3000:])
