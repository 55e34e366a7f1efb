"""ColorMNet memory bookkeeping (vsdeoldify_amd/colormnet_memory.py) against a scenario recorded by EXECUTING the reference's
MemoryManager (tools/gen_golden_colormnet_memory.py -> tests/golden/colormnet_memory_manager.npz): 16 frames, each matched against the
memory and then added to it; the working memory is consolidated into long-term prototypes six times and the long-term memory pruned.

CPU: the bookkeeping alone, with the oracle's restatement of the three memory_util functions plugged in as the backend.
GPU: the product path (libhavc_mi355 kernels)."""
import os

import numpy as np
import pytest
import torch

from oracle import colormnet as O
from vsdeoldify_amd.colormnet_memory import MemoryManager

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "colormnet_memory_manager.npz"))
CFG = {k: (bool(v) if k.startswith("enable") else int(v)) for k, v in zip(GOLD["config_keys"].tolist(), GOLD["config_vals"].tolist())}
H, W, CK, CV, OBJ, FRAMES = (int(v) for v in GOLD["dims"])


class OracleBackend:
    """memory_util.get_similarity / do_softmax / readout as restated in oracle/colormnet.py (test infrastructure)"""

    def read_topk_usage(self, mk, ms, qk, qe, mv, top_k, want_usage):
        aff = O.do_softmax(O.get_similarity(mk, None if ms is None else ms.flatten(start_dim=1), qk, qe), top_k)
        return mv @ aff[0], (aff.sum(dim=2) if want_usage else None)

    def dense_readout(self, mk, ms, qk, qe, mv):
        aff = O.do_softmax(O.get_similarity(mk, None if ms is None else ms.flatten(start_dim=1), qk, qe), None)
        return mv @ aff[0]


def replay(long_term, backend=None, device="cpu", tol=2e-5):
    cfg = dict(CFG, enable_long_term=long_term, enable_long_term_count_usage=long_term)
    m = MemoryManager(cfg, backend=backend)
    tag = "lt" if long_term else "st"
    sizes = []
    for t in range(FRAMES):
        f = {k: torch.from_numpy(GOLD[f"in_{k}_{t}"]).to(device) for k in ("key", "shrinkage", "selection", "value")}
        if t > 0:
            got = m.match_memory(f["key"], f["selection"] if long_term else None).cpu().numpy()
            want = GOLD[f"{tag}_readout_{t}"]
            assert got.shape == want.shape == (OBJ, CV, H, W)
            assert np.abs(got - want).max() < tol * max(1.0, float(np.abs(want).max())), (t, float(np.abs(got - want).max()))
        m.add_memory(f["key"], f["shrinkage"], f["value"], [1, 2], selection=f["selection"] if long_term else None)
        sizes.append((m.work_mem.size, m.long_mem.size if long_term else 0))
    assert np.array_equal(np.array(sizes), GOLD[f"{tag}_sizes"])
    if long_term:
        assert np.abs(m.long_mem.key.cpu().numpy() - GOLD["lt_long_key"]).max() == 0          # the same prototypes were chosen
        assert np.abs(m.long_mem.get_usage().cpu().numpy() - GOLD["lt_long_usage"]).max() < 1e-4
    return m


@pytest.mark.parametrize("long_term", [True, False])
def test_memory_manager_bookkeeping_matches_the_executed_reference(long_term):
    m = replay(long_term, backend=OracleBackend())
    assert m.work_mem.num_groups == 1 and m.work_mem.get_v_size(0) == m.work_mem.size
    m.create_hidden_state(2, torch.zeros(1, CK, H, W))
    assert m.get_hidden().shape == (1, 2, CFG["hidden_dim"], H, W)


def test_a_second_object_group_is_refused():
    m = MemoryManager(dict(CFG, enable_long_term=False, enable_long_term_count_usage=False), backend=OracleBackend())
    f = {k: torch.from_numpy(GOLD[f"in_{k}_0"]) for k in ("key", "shrinkage", "value")}
    m.add_memory(f["key"], f["shrinkage"], f["value"], [1, 2])
    with pytest.raises(NotImplementedError):
        m.add_memory(f["key"], f["shrinkage"], f["value"][:, :1], [3])


@pytest.mark.gpu
@pytest.mark.parametrize("long_term", [True, False])
@pytest.mark.parametrize("device", ["cpu", "cuda"])
def test_gpu_memory_manager_matches_the_executed_reference(long_term, device):
    """host tensors (staged by the library) and device tensors (used in place through their pointers)"""
    if device == "cuda" and not torch.cuda.is_available():
        pytest.skip("torch sees no GPU")
    replay(long_term, backend=None, device=device, tol=1e-4)


def test_a_failing_long_term_cleanup_is_swallowed_like_the_reference():
    """memory_manager.py:183-193 wraps remove_obsolete_features + compress_features in try / except: pass.  Two ways to get there:
    num_prototypes larger than the candidate range (torch.topk raises), and a long-term memory of EXACTLY max_size elements
    (kv_memory_store.py:153-154: topk(k=0) then values[-1] -> IndexError): the frame's consolidation is skipped, the clip goes on."""
    cfg = dict(CFG, enable_long_term=True, enable_long_term_count_usage=True, num_prototypes=10 ** 6)
    m = MemoryManager(cfg, backend=OracleBackend())
    for t in range(FRAMES):
        f = {k: torch.from_numpy(GOLD[f"in_{k}_{t}"]) for k in ("key", "shrinkage", "selection", "value")}
        if t > 0:
            m.match_memory(f["key"], f["selection"])
        m.add_memory(f["key"], f["shrinkage"], f["value"], [1, 2], selection=f["selection"])       # never raises
    assert not m.long_mem.engaged() and m.work_mem.size == FRAMES * H * W                          # nothing was ever consolidated
    from vsdeoldify_amd.colormnet_memory import KeyValueMemoryStore
    s = KeyValueMemoryStore(count_usage=True)
    s.add(torch.zeros(1, CK, 5), [torch.zeros(OBJ, CV, 5)], torch.ones(1, 1, 5), None, None)
    with pytest.raises(IndexError):
        s.remove_obsolete_features(5)
    s.remove_obsolete_features(6)                                                                   # below max_size: nothing to do
    assert s.size == 5
