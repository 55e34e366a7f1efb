"""The all-CPU ColorMNet frame loop (ORACLE / TEST INFRASTRUCTURE ONLY): oracle network (oracle/colormnet_net.py) + the per-frame step and
memory bookkeeping with the oracle's memory functions plugged in as the backend.  The step / memory classes are the drop-ins of
vsdeoldify_amd (colormnet_core.InferenceCore, colormnet_memory.MemoryManager: pure host bookkeeping, pinned to the executed reference by
tests/test_colormnet_core.py and tests/test_colormnet_memory.py); with `OracleBackend` nothing touches the GPU library.
Reproduces the reference's own ColorMNetRender.colorize_frame over a clip to <= 1 LSB (tests/test_colormnet_net.py)."""
import numpy as np
import torch

from . import colormnet as mem
from . import colormnet_net as net


class OracleBackend:
    """memory_util.get_similarity / do_softmax / readout as restated in oracle/colormnet.py"""

    def read_topk_usage(self, mk, ms, qk, qe, mv, top_k, want_usage):
        aff = mem.do_softmax(mem.get_similarity(mk, None if ms is None else ms.flatten(start_dim=1), qk, qe), top_k)
        return mv @ aff[0], (aff.sum(dim=2) if want_usage else None)

    def dense_readout(self, mk, ms, qk, qe, mv):
        aff = mem.do_softmax(mem.get_similarity(mk, None if ms is None else ms.flatten(start_dim=1), qk, qe), None)
        return mv @ aff[0]


class OracleNetwork(net.Network):
    """the oracle network with the extra methods ColorMNetRender asks of its network object (frame transforms, stream scope): lets the
    drop-in class's state machine (vsdeoldify_amd/colormnet_render.py) run entirely on the CPU against the executed-reference scenarios"""

    class _Null:
        def __enter__(self):
            return self

        def __exit__(self, *exc):
            return False

    def on_stream(self):
        return self._Null()

    def image_to_lab(self, rgb_u8):
        return net.frame_to_lab_tensor(np.asarray(rgb_u8))

    def lab_to_image(self, l_plane, ab, out=None):
        return net.lab_tensor_to_rgb(l_plane, ab)


def colorize_clip(sd, frames_rgb, refs, config_updates=None, vid_length=None):
    """frames_rgb: list of u8 [H, W, 3]; refs: {frame index: u8 RGB reference image}; -> list of u8 [H, W, 3]
    (colormnet_render.py:197-283 with reset_on_ref_update = False and FirstFrameIsNotExemplar = True: the HAVC_deepex default, method 0)"""
    from vsdeoldify_amd.colormnet_core import InferenceCore
    from vsdeoldify_amd.colormnet_render import default_config
    network = net.Network(sd)
    cfg = default_config(vid_length or len(frames_rgb), min(10000, vid_length or len(frames_rgb)))
    cfg.update(key_dim=network.key_dim, value_dim=network.value_dim, hidden_dim=network.hidden_dim)
    cfg.update(config_updates or {})
    proc = InferenceCore(network, cfg, memory_backend=OracleBackend())
    outs = []
    for t, rgb in enumerate(frames_rgb):
        lab = net.frame_to_lab_tensor(np.asarray(rgb))
        lll = lab[:1].repeat(3, 1, 1)
        ref = refs.get(t)
        with torch.no_grad():
            if ref is not None:
                m = net.frame_to_lab_tensor(np.asarray(ref))
                proc.set_all_labels([1, 2])
                ab = proc.step_AnyExemplar(lll, m[:1].repeat(3, 1, 1), m[1:3], [1, 2], end=False)
            else:
                ab = proc.step_AnyExemplar(lll, None, None, None, end=False)
        outs.append(net.lab_tensor_to_rgb(lll[:1], ab))
    return outs
