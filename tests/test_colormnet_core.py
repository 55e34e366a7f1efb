"""ColorMNet per-frame step (vsdeoldify_amd/colormnet_core.py InferenceCore) against scenarios recorded by EXECUTING the reference's
InferenceCore on the stub network of tests/colormnet_stub.py (tools/gen_golden_colormnet_core.py): nine frames, a reference image
arriving with frames 0 and 4, memory frames every second frame, synchronous and periodic deep updates, with and without long-term
memory; plus the plain step() entry.  Checked: every returned ab plane, the ORDER and the flags of the network calls, memory sizes."""
import os

import numpy as np
import pytest
import torch

from tests.colormnet_stub import HID, StubNet, clip
from tests.test_colormnet_memory import OracleBackend
from vsdeoldify_amd.colormnet_core import InferenceCore, pad_divide_by, unpad

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "colormnet_core.npz"))
BASE = dict(hidden_dim=HID, top_k=4, enable_long_term=True, enable_long_term_count_usage=True, max_mid_term_frames=4, min_mid_term_frames=2,
            num_prototypes=4, max_long_term_elements=40, mem_every=2, deep_update_every=-1)
SCENARIOS = {"exemplar_sync": dict(BASE), "nosync": dict(BASE, deep_update_every=3),
             "short_only": dict(BASE, enable_long_term=False, enable_long_term_count_usage=False)}


def replay(name, backend, tol):
    cfg = SCENARIOS[name]
    imgs, ex_l, ex_ab = clip()
    net = StubNet()
    proc = InferenceCore(net, cfg, memory_backend=backend)
    proc.set_all_labels([1, 2])
    sizes = []
    for t, img in enumerate(imgs):
        with torch.no_grad():
            if t in (0, 4):
                out = proc.step_AnyExemplar(img, ex_l * (1.0 if t == 0 else -0.5), ex_ab * (1.0 if t == 0 else 0.7), [1, 2], end=False)
            else:
                out = proc.step_AnyExemplar(img, None, None, end=(t == len(imgs) - 1))
        want, sums = GOLD[f"{name}_out_{t}"], GOLD[f"{name}_sum_{t}"]                  # every 4th pixel + the plane sums
        got = out.numpy()
        assert got.shape == (2, 112, 112) and np.abs(got[:, ::4, ::4] - want).max() < tol, (name, t, float(np.abs(got[:, ::4, ::4] - want).max()))
        assert abs(float(got.sum()) - sums[0]) < 2e3 * tol and abs(float(np.abs(got).sum()) - sums[1]) < 2e3 * tol, (name, t)
        sizes.append((proc.memory.work_mem.size, proc.memory.long_mem.size if cfg["enable_long_term"] and proc.memory.long_mem.engaged() else 0))
    assert np.array_equal(np.array(sizes), GOLD[f"{name}_sizes"])
    assert ["/".join(str(x) for x in c) for c in net.calls] == GOLD[f"{name}_calls"].tolist()
    net = StubNet()
    proc = InferenceCore(net, cfg, memory_backend=backend)
    proc.set_all_labels([1, 2])
    for t, img in enumerate(imgs[:5]):
        with torch.no_grad():
            out = proc.step(img, ex_ab if t == 0 else None, [1, 2] if t == 0 else None, end=(t == 4))
        assert np.abs(out.numpy()[:, ::4, ::4] - GOLD[f"{name}_step_{t}"]).max() < tol, (name, "step", t)


@pytest.mark.parametrize("name", list(SCENARIOS))
def test_inference_core_matches_the_executed_reference(name):
    replay(name, OracleBackend(), 2e-5)


def test_pad_divide_by_and_unpad():
    x = torch.arange(3 * 40 * 56, dtype=torch.float32).view(3, 40, 56)
    p, pad = pad_divide_by(x, 112)
    assert p.shape == (3, 112, 112) and pad == (28, 28, 36, 36) and torch.equal(unpad(p, pad), x)
    q, pad = pad_divide_by(torch.zeros(2, 112, 224), 112)
    assert pad == (0, 0, 0, 0) and unpad(q, pad).shape == (2, 112, 224)


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(SCENARIOS))
def test_gpu_inference_core_matches_the_executed_reference(name):
    """the same scenarios with the memory on the MI355X kernels (MemoryManager's default backend)"""
    replay(name, None, 1e-4)
