"""ColorMNet memory bookkeeping on the MI355X kernels behind the reference's class names and call shapes.

  KeyValueMemoryStore   colormnet/inference/kv_memory_store.py (keys / shrinkage / selection / values + usage counters)
  MemoryManager         colormnet/inference/memory_manager.py: match_memory (:58-150), add_memory (:152-193), compress_features /
                        consolidation (:216-288), hidden-state plumbing (:195-214)

Tensors are torch tensors as the reference's InferenceCore passes them (CPU tensors are staged by the library, CUDA / ROCm tensors of
the ctx's GPU are used in place); concatenation / slicing of the memory banks is plain tensor bookkeeping, the arithmetic runs in
libhavc_mi355: one fused similarity + top-k softmax + readout (+ usage) per frame (havc_memory_read_topk_usage), one similarity +
dense softmax + readout per consolidation (havc_memory_dense_readout).  With these two classes and colormnet.local_attention the
reference's ColorMNet runs on AMD GPUs without the CUDA-only spatial_correlation_sampler wheel and without ever building the
N x HW affinity.  Scope: ONE object group (all objects enter at the first frame: colormnet_render.py always passes [1, 2]); a second
group raises NotImplementedError.  No CPU fallback: `backend` exists for the CPU test of the bookkeeping only.
"""
import ctypes as C

import numpy as np

from . import _native as nat
from .render import get_context


class _HipBackend:
    """the two fused memory operations through the C ABI"""

    def __init__(self, device_index=0):
        self.ctx = get_context(device_index)

    @staticmethod
    def _p(t):
        return None if t is None else C.c_void_p(t.data_ptr())

    def read_topk_usage(self, mk, ms, qk, qe, mv, top_k, want_usage):
        """mk [1,CK,N], ms [1,1,N] / None, qk [1,CK,HW], qe [1,CK,HW] / None, mv [R,N] (all value rows) -> ([R,HW], usage [1,N] / None)"""
        import torch
        ctx = self.ctx
        mk, qk, mv = (x.float().contiguous() for x in (mk, qk, mv))
        ms = None if ms is None else ms.float().contiguous()
        qe = None if qe is None else qe.float().contiguous()
        CK, N, HW, R = mk.shape[1], mk.shape[2], qk.shape[2], mv.shape[0]
        out = torch.empty((R, HW), dtype=torch.float32, device=mk.device)
        usage = torch.empty((1, N), dtype=torch.float32, device=mk.device) if want_usage else None
        from .colormnet import ordered
        with ordered(ctx, mk, ms, qk, qe, mv):                  # the operands were produced asynchronously on torch's stream (cat, float, contiguous)
            nat.check(ctx.lib.havc_memory_read_topk_usage(ctx.h, self._p(mk), self._p(ms), self._p(qk), self._p(qe), self._p(mv), self._p(out),
                                                          self._p(usage), 1, CK, R, N, HW, int(top_k)), ctx.h)
        return out, usage

    def dense_readout(self, mk, ms, qk, qe, mv):
        """candidates mk [1,CK,N], ms, prototypes qk [1,CK,P], qe, rows mv [R,N] -> [R,P] (softmax over the candidates, no top-k)"""
        import torch
        ctx = self.ctx
        mk, qk, mv = (x.float().contiguous() for x in (mk, qk, mv))
        ms = None if ms is None else ms.float().contiguous()
        qe = None if qe is None else qe.float().contiguous()
        CK, N, P, R = mk.shape[1], mk.shape[2], qk.shape[2], mv.shape[0]
        out = torch.empty((R, P), dtype=torch.float32, device=mk.device)
        from .colormnet import ordered
        with ordered(ctx, mk, ms, qk, qe, mv):
            nat.check(ctx.lib.havc_memory_dense_readout(ctx.h, self._p(mk), self._p(ms), self._p(qk), self._p(qe), self._p(mv), self._p(out), 1, CK, R, N, P),
                      ctx.h)
        return out


def _cat(a, b):
    import torch
    return b if a is None else torch.cat([a, b], -1)


class KeyValueMemoryStore:
    """keys [1,CK,N], shrinkage [1,1,N], selection [1,CK,N] or None, values [objects,CV,N] (one object group), optional usage counters"""

    def __init__(self, count_usage):
        self.count_usage = count_usage
        self.k = self.s = self.e = self.v = None
        self.objects = None
        self.use_count = self.life_count = None
        self.version = 0                                       # bumped whenever keys / values change (not by usage updates): MemoryManager caches on it

    def add(self, key, value, shrinkage, selection, objects):
        import torch
        n = key.shape[-1]
        if objects is not None:                                # working memory: value [1 or objects, ...] tensor indexed by object
            objs = [o - 1 for o in objects]
            if self.objects is None:
                self.objects = objs
            elif objs != self.objects:
                raise NotImplementedError("objects entering after the first frame (a second object group) are not supported")
            # (value[objs] with a Python list builds its index tensor on the host and uploads it from pageable memory: the call BLOCKS until the
            #  stream has drained -- 3.3 ms per memory frame, half the host time of a clip, found with tools/cmn_host_probe.py)
            if objs != list(range(value.shape[0])):
                value = torch.stack([value[o] for o in objs], 0)
        else:                                                  # long-term memory: list of per-group tensors
            if len(value) != 1:
                raise NotImplementedError("one object group only")
            value = value[0]
        self.version += 1
        self.k = _cat(self.k, key)
        self.s = _cat(self.s, shrinkage) if shrinkage is not None else self.s
        self.e = _cat(self.e, selection) if selection is not None else self.e
        self.v = _cat(self.v, value)
        if self.count_usage:
            self.use_count = _cat(self.use_count, torch.zeros((1, 1, n), dtype=torch.float32, device=key.device))
            self.life_count = _cat(self.life_count, torch.zeros((1, 1, n), dtype=torch.float32, device=key.device) + 1e-7)

    def update_usage(self, usage):
        if self.count_usage:
            self.use_count = self.use_count + usage.view_as(self.use_count)
            self.life_count = self.life_count + 1

    def _keep(self, pick):
        self.version += 1
        self.k = pick(self.k)
        self.s = None if self.s is None else pick(self.s)
        self.e = None if self.e is None else pick(self.e)
        if self.count_usage:
            self.use_count, self.life_count = pick(self.use_count), pick(self.life_count)

    def sieve_by_range(self, start, end, min_size):
        """drop elements [start, end) (end <= 0 counts from the back, 0 = to the end); values only when at least min_size of them exist"""
        import torch
        n = self.size
        stop = n + end if end < 0 else (n if end == 0 else end)

        def pick(x):
            return torch.cat([x[..., :start], x[..., stop:]], -1)
        self._keep(pick)
        if self.v.shape[-1] >= min_size:
            self.v = pick(self.v)

    def remove_obsolete_features(self, max_size):
        import torch
        if not self.count_usage or self.size < max_size:
            return
        if self.size == max_size:
            # kv_memory_store.py:153-154: torch.topk(usage, k=0) is empty and `values[-1]` raises IndexError; MemoryManager.add_memory swallows
            # it (memory_manager.py:183-193) and with it THIS frame's consolidation.  Same observable behaviour here.
            raise IndexError("index -1 is out of bounds for dimension 0 with size 0")
        usage = self.get_usage().flatten()
        lowest, _ = torch.topk(usage, k=self.size - max_size, largest=False, sorted=True)
        survived = usage > lowest[-1]
        self._keep(lambda x: x[..., survived])
        self.v = self.v[..., survived]

    def get_usage(self):
        if not self.count_usage:
            raise RuntimeError("this store does not count usage")
        return self.use_count / self.life_count

    def get_all_sliced(self, start, end):
        n = self.size
        stop = n + end if end < 0 else (n if end == 0 else end)
        sl = lambda x: None if x is None else x[..., start:stop]
        return sl(self.k), sl(self.s), sl(self.e), sl(self.get_usage())

    def get_v_size(self, gi):
        return self.v.shape[-1]

    def engaged(self):
        return self.k is not None

    @property
    def size(self):
        return 0 if self.k is None else self.k.shape[-1]

    @property
    def num_groups(self):
        return 0 if self.v is None else 1

    key = property(lambda self: self.k)
    value = property(lambda self: [] if self.v is None else [self.v])
    shrinkage = property(lambda self: self.s)
    selection = property(lambda self: self.e)


class MemoryManager:
    """drop-in for colormnet.inference.memory_manager.MemoryManager (same config keys, same method names / arguments / results)"""

    def __init__(self, config, device_index=0, backend=None):
        self.hidden_dim = config["hidden_dim"]
        self.top_k = config["top_k"]
        self.enable_long_term = config["enable_long_term"]
        self.enable_long_term_usage = config["enable_long_term_count_usage"]
        if self.enable_long_term:
            self._read_lt_config(config)
        self.CK = self.CV = self.H = self.W = None
        self.hidden = None
        self.work_mem = KeyValueMemoryStore(count_usage=self.enable_long_term)
        if self.enable_long_term:
            self.long_mem = KeyValueMemoryStore(count_usage=self.enable_long_term_usage)
        self.reset_config = True
        self.backend = backend if backend is not None else _HipBackend(device_index)

    def _read_lt_config(self, config):
        self.max_mt_frames = config["max_mid_term_frames"]
        self.min_mt_frames = config["min_mid_term_frames"]
        self.num_prototypes = config["num_prototypes"]
        self.max_long_elements = config["max_long_term_elements"]

    def update_config(self, config):
        self.reset_config = True
        self.hidden_dim = config["hidden_dim"]
        self.top_k = config["top_k"]
        assert self.enable_long_term == config["enable_long_term"], "cannot update this"
        assert self.enable_long_term_usage == config["enable_long_term_count_usage"], "cannot update this"
        if self.enable_long_term:
            self._read_lt_config(config)

    # ---- per-frame read (memory_manager.py:58-150) ----
    def match_memory(self, query_key, selection):
        """query_key [1,CK,H,W], selection [1,CK,H,W] or None -> [objects, CV, H, W]"""
        import torch
        h, w = query_key.shape[-2:]
        qk = query_key.flatten(start_dim=2)
        qe = selection.flatten(start_dim=2) if selection is not None else None
        use_long = self.enable_long_term and self.long_mem.engaged()
        if use_long:
            # the concatenated [long-term | working] banks change only when a frame is memorised or the memory is consolidated (every
            # mem_every-th frame): between those frames the concatenation (57 MB of values at 14 000 elements) is reused, not rebuilt
            long_size = self.long_mem.size
            stamp = (id(self.long_mem), self.long_mem.version, self.work_mem.version)
            if getattr(self, "_banks", (None,))[0] != stamp:
                self._banks = (stamp, torch.cat([self.long_mem.key, self.work_mem.key], -1),
                               torch.cat([self.long_mem.shrinkage, self.work_mem.shrinkage], -1), torch.cat([self.long_mem.v, self.work_mem.v], -1))
            _, mk, ms, mv = self._banks
        else:
            long_size = 0
            mk, ms, mv = self.work_mem.key, self.work_mem.shrinkage, self.work_mem.v
        objs, cv, n = mv.shape
        out, usage = self.backend.read_topk_usage(mk, ms, qk, qe, mv.reshape(objs * cv, n), self.top_k, want_usage=self.enable_long_term)
        if self.enable_long_term:
            self.work_mem.update_usage(usage[:, long_size:].flatten())
            if use_long and self.enable_long_term_usage:
                self.long_mem.update_usage(usage[:, :long_size].flatten())
        return out.view(objs, self.CV, h, w)

    # ---- per-frame write + long-term clean-up (behaviour of memory_manager.py:152-193; the interface InferenceCore drives) ----
    def _bind_geometry(self, hw):
        """the first memorised frame (or the first one after a reset) fixes the key grid; the long-term bounds are frame counts turned into elements"""
        self.reset_config = False
        self.H, self.W = int(hw[0]), int(hw[1])
        self.HW = self.H * self.W
        if self.enable_long_term:
            self.min_work_elements, self.max_work_elements = (frames * self.HW for frames in (self.min_mt_frames, self.max_mt_frames))

    def _maintain_long_term(self):
        """working memory full -> make room in the long-term store if it is nearly full, then consolidate; a failure leaves the memory as it is"""
        if self.work_mem.size < self.max_work_elements:
            return
        room = self.max_long_elements - self.num_prototypes
        if self.long_mem.size >= room:
            self.long_mem.remove_obsolete_features(room)
        self.compress_features()

    def add_memory(self, key, shrinkage, value, objects, selection=None):
        """key [1, CK, H, W], shrinkage [1, 1, H, W], value [1, objects, CV, H, W], selection [1, CK, H, W] or None; objects: 1-based ids"""
        if self.H is None or self.reset_config:
            self._bind_geometry(key.shape[-2:])
        rows = {name: (t if t is None else t.flatten(start_dim=2)) for name, t in
                (("key", key), ("shrinkage", shrinkage), ("value", value[0]), ("selection", selection))}
        self.CK, self.CV = rows["key"].shape[1], rows["value"].shape[1]
        self.work_mem.add(rows["key"], rows["value"], rows["shrinkage"], rows["selection"], objects)
        if self.enable_long_term:
            try:
                self._maintain_long_term()
            except Exception:              # the reference swallows a failing clean-up and goes on with the clip (memory_manager.py:183-193)
                pass

    def create_hidden_state(self, n, sample_key):
        """make the hidden state cover n objects on sample_key's grid: created as zeros, or grown by zero planes for objects that joined later"""
        import torch.nn.functional as F
        grid = tuple(sample_key.shape[-2:])
        if self.hidden is None:
            self.hidden = sample_key.new_zeros((1, n, self.hidden_dim) + grid, dtype=sample_key.dtype if sample_key.is_floating_point() else None)
        else:
            missing = n - self.hidden.shape[1]
            if missing < 0:                # F.pad with a negative count would silently CROP the objects; the reference's torch.zeros((1, missing, ...)) raises
                raise RuntimeError(f"Trying to create tensor with negative dimension {missing}: hidden state of {self.hidden.shape[1]} objects, {n} asked for")
            if missing:
                self.hidden = F.pad(self.hidden, (0, 0, 0, 0, 0, 0, 0, missing))
        if self.hidden.shape[1] != n:
            raise AssertionError("hidden state / object count mismatch")

    def set_hidden(self, hidden):
        self.hidden = hidden

    def get_hidden(self):
        return self.hidden

    # ---- working memory -> long-term prototypes (memory_manager.py:216-288) ----
    def compress_features(self):
        import torch
        HW = self.HW
        start, end = HW, -self.min_work_elements + HW           # the first frame and the last min_mt_frames - 1 frames stay
        cand_k, cand_s, cand_e, usage = self.work_mem.get_all_sliced(start, end)
        stop = self.work_mem.size + end if end < 0 else self.work_mem.size
        cand_v = self.work_mem.v[..., start:stop]
        # memory_manager.py:240-243 takes torch.topk(usage, k = num_prototypes): on a GPU its choice (and order) among EQUAL usage values is unspecified and may
        # differ from run to run -- and the order of the prototypes is the order of the long-term memory, i.e. the summation order of every later read.  A stable
        # descending sort picks the same elements whenever the values differ and the LOWER index among equals: one answer, every run (round 5).
        top = torch.sort(usage, dim=-1, descending=True, stable=True).indices[..., :self.num_prototypes]
        proto = top.flatten()
        proto_k = cand_k[:, :, proto]
        proto_e = cand_e[:, :, proto] if cand_e is not None else None
        objs, cv, n = cand_v.shape
        rows = torch.cat([cand_v.reshape(objs * cv, n), cand_s.reshape(1, n)], 0)           # values and the shrinkage share the affinity
        got = self.backend.dense_readout(cand_k, cand_s, proto_k, proto_e, rows)
        proto_v = got[:objs * cv].view(objs, cv, -1)
        proto_s = got[objs * cv:].view(1, 1, -1)
        self.work_mem.sieve_by_range(start, end, min_size=self.min_work_elements + HW)
        self.long_mem.add(proto_k, [proto_v], proto_s, selection=None, objects=None)
