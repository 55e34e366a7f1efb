"""ColorMNet frame loop without tensor bookkeeping between the kernels (round 4).

The drop-in classes of colormnet_core.py / colormnet_memory.py follow the reference line by line -- and inherit what its InferenceCore /
MemoryManager / KeyValueMemoryStore do on EVERY frame with torch: F.pad of the frame, repeat / stack / cat to assemble network inputs,
torch.cat of the memory banks, `readout + short`, `use_count + usage`, `life_count + 1`, fresh tensors for every result
(colormnet/inference/inference_core.py:119-230, memory_manager.py:58-246, kv_memory_store.py:36-170).  On the MI355X each of those is a
5 us kernel plus 10 - 30 us of host time, and a frame had ~16 of them next to ~100 library launches: the host, not the GPU, set the frame rate.

Here the same state machine runs on PRE-SIZED device buffers through the fast-step entry points of include/havc_mi355.h:
  * `Banks` / `BankedStore`: working and long-term memory live side by side in ONE buffer per quantity (keys, shrinkage, selection, values, usage
    counters) whose rows have a pitch: long-term elements right-aligned in front of the working memory, so [long | work] is one contiguous
    column range and the read needs no concatenation.  A memorised frame is appended with pitched device copies; the usage update happens
    inside the read (havc_memory_read_banked).  The RARE events -- consolidation of the working memory into prototypes, removal of obsolete
    long-term elements (every ~25th frame), a reference image arriving, a reset -- keep the reference's tensor code (on views of the banks);
  * `FastInferenceCore`: step / step_AnyExemplar with the frame already padded (havc_cmn_frame_in), the short-term attention forked onto the
    context's second stream next to the memory read (havc_cmn_short_term / havc_cmn_join_add), the decoder and the value encoder as
    multi-bind + multi-slice calls on preallocated results.
Same arithmetic, same order of the memory elements, same bytes as the line-by-line classes (tests/test_colormnet_net.py runs the four recorded
scenarios of the reference's own ColorMNetRender through both).  HAVC_CMN_FAST=0 selects the line-by-line classes.  GPU only: no backend hook."""
import ctypes as C
import os

from . import _native as nat
from .colormnet_core import DIVIDE_BY, InferenceCore
from .colormnet_memory import KeyValueMemoryStore, MemoryManager


# the read of frame t+1 (short-term attention + memory read) enqueued under the decoder of frame t on the frames between two memory frames
# (FastInferenceCore._read); HAVC_CMN_READ_AHEAD=0 keeps every read in front of its own decoder (A/B runs; same bytes either way)
READ_AHEAD = os.environ.get("HAVC_CMN_READ_AHEAD", "1") != "0"


def _ptr(t, col=0):
    """device pointer of a tensor, `col` fp32 columns into its rows"""
    return C.c_void_p(t.data_ptr() + 4 * col)


class Banks:
    """one pre-sized buffer per memory quantity: [long-term region (cap_long columns, right-aligned) | working region (cap_work columns)]"""

    def __init__(self, device, CK, objs, CV, cap_long, cap_work, has_selection):
        import torch
        self.cap_long, self.cap_work, self.cap = cap_long, cap_work, cap_long + cap_work
        self.K = torch.empty((1, CK, self.cap), dtype=torch.float32, device=device)
        self.S = torch.empty((1, 1, self.cap), dtype=torch.float32, device=device)
        self.E = torch.empty((1, CK, self.cap), dtype=torch.float32, device=device) if has_selection else None
        self.V = torch.empty((objs, CV, self.cap), dtype=torch.float32, device=device)
        self.use = torch.zeros((1, 1, self.cap), dtype=torch.float32, device=device)
        self.life = torch.full((1, 1, self.cap), 1e-7, dtype=torch.float32, device=device)      # kv_memory_store.py:72: new elements start at 1e-7

    def grow_work(self, need):
        """working memory without a cap (enable_long_term = False): double the region (rare; plain tensor copies)"""
        import torch
        new_work = max(2 * self.cap_work, need)
        for name in ("K", "S", "E", "V", "use", "life"):
            t = getattr(self, name)
            if t is None:
                continue
            n = torch.empty(t.shape[:-1] + (self.cap_long + new_work,), dtype=t.dtype, device=t.device)
            if name == "use":
                n.zero_()
            elif name == "life":
                n.fill_(1e-7)
            n[..., :self.cap] = t
            setattr(self, name, n)
        self.cap_work, self.cap = new_work, self.cap_long + new_work


class BankedStore:
    """KeyValueMemoryStore (kv_memory_store.py) on a region of the banks: same interface, same element order"""

    def __init__(self, banks, ctx, kind, count_usage):
        self.banks, self.ctx, self.kind, self.count_usage = banks, ctx, kind, count_usage
        self.n = 0
        self.objects = None
        self.version = 0
        self.on_grow = None                                    # called after the banks' capacity changed (the manager re-reserves the read's scratch)

    # ---- region ----
    def _range(self):
        b = self.banks
        return (b.cap_long - self.n, b.cap_long) if self.kind == "long" else (b.cap_long, b.cap_long + self.n)

    def _view(self, t):
        if t is None or self.n == 0:
            return None
        a, e = self._range()
        return t[..., a:e]

    k = key = property(lambda self: self._view(self.banks.K))
    s = shrinkage = property(lambda self: self._view(self.banks.S))
    e = selection = property(lambda self: self._view(self.banks.E) if self.kind == "work" else None)
    v = property(lambda self: self._view(self.banks.V))
    value = property(lambda self: [] if self.n == 0 else [self._view(self.banks.V)])
    use_count = property(lambda self: self._view(self.banks.use))
    life_count = property(lambda self: self._view(self.banks.life))
    size = property(lambda self: self.n)
    num_groups = property(lambda self: 1 if self.n else 0)

    def engaged(self):
        return self.n > 0

    def get_v_size(self, gi):
        return self.n

    # ---- per memory frame: append (device copies with a destination pitch, nothing else) ----
    def _put(self, dst, col, src):
        """src [..., n] contiguous -> columns [col, col + n) of every row of dst"""
        rows = src.numel() // src.shape[-1]
        nat.check(self.ctx.lib.havc_dev_copy_2d(self.ctx.h, _ptr(dst, col), dst.shape[-1] * 4, C.c_void_p(src.data_ptr()), src.shape[-1] * 4,
                                                src.shape[-1] * 4, rows), self.ctx.h)

    def add(self, key, value, shrinkage, selection, objects):
        import torch
        b, n = self.banks, key.shape[-1]
        if objects is not None:                                # working memory: value [objects, CV, n]
            objs = [o - 1 for o in objects]
            if self.objects is None:
                self.objects = objs
            elif objs != self.objects:
                raise NotImplementedError("objects entering after the first frame (a second object group) are not supported")
            if objs != list(range(value.shape[0])):
                value = torch.stack([value[o] for o in objs], 0)
        else:                                                  # long-term memory: list of per-group tensors
            if len(value) != 1:
                raise NotImplementedError("one object group only")
            value = value[0]
        self.version += 1
        if self.kind == "work":
            if self.n + n > b.cap_work:
                b.grow_work(self.n + n)
                if self.on_grow is not None:
                    self.on_grow()
            col = b.cap_long + self.n
        else:
            if self.n + n > b.cap_long:
                raise RuntimeError("long-term memory beyond its capacity")      # (remove_obsolete_features runs before every consolidation)
            if self.n:                                         # new prototypes go BEHIND the old ones (torch.cat order): shift the old ones left
                a, e = self._range()
                for t in (b.K, b.S, b.V) + ((b.use, b.life) if self.count_usage else ()):
                    t[..., a - n:e - n] = t[..., a:e].clone()
            col = b.cap_long - n
        c = lambda t: t.float().contiguous()
        self._put(b.K, col, c(key))
        if shrinkage is not None:
            self._put(b.S, col, c(shrinkage))
        if selection is not None and b.E is not None and self.kind == "work":
            self._put(b.E, col, c(selection))
        self._put(b.V, col, c(value))
        if self.kind == "long" and self.count_usage:
            b.use[..., col:col + n] = 0
            b.life[..., col:col + n] = 1e-7
        self.n += n

    def update_usage(self, usage):                             # (the banked read updates the counters in place; kept for interface parity)
        if self.count_usage:
            a, e = self._range()
            self.banks.use[..., a:e] += usage.view(1, 1, -1)
            self.banks.life[..., a:e] += 1

    # ---- rare: consolidation / obsolete-element removal, the reference's tensor code on views, written back in place ----
    def _write_back(self, picked):
        """picked: {bank name: new contents [..., m]} for this region; the freed columns of the usage counters return to their initial values"""
        b = self.banks
        m = next(iter(picked.values())).shape[-1]
        a, e = self._range()
        for name, new in picked.items():
            t = getattr(b, name)
            if self.kind == "work":
                t[..., a:a + m] = new
            else:
                t[..., e - m:e] = new
        if self.kind == "work":
            b.use[..., a + m:e] = 0
            b.life[..., a + m:e] = 1e-7
        self.n = m
        self.version += 1

    def _keep(self, pick, with_values=True):
        names = ["K", "S"] + (["E"] if (self.kind == "work" and self.banks.E is not None) else []) + (["use", "life"] if self.count_usage else [])
        if with_values:
            names.append("V")
        self._write_back({name: pick(self._view(getattr(self.banks, name))).clone() for name in names})

    def sieve_by_range(self, start, end, min_size):
        import torch
        n = self.n
        stop = n + end if end < 0 else (n if end == 0 else end)
        if n < min_size:
            raise NotImplementedError("value banks shorter than the key banks are not supported")       # (never the case: the sieve runs on a full working memory)
        self._keep(lambda x: torch.cat([x[..., :start], x[..., stop:]], -1))

    def remove_obsolete_features(self, max_size):
        import torch
        if not self.count_usage or self.n < max_size:
            return
        if self.n == max_size:
            raise IndexError("index -1 is out of bounds for dimension 0 with size 0")                    # kv_memory_store.py:153-154 (swallowed by add_memory)
        usage = self.get_usage().flatten()
        lowest, _ = torch.topk(usage, k=self.n - max_size, largest=False, sorted=True)
        survived = usage > lowest[-1]
        self._keep(lambda x: x[..., survived])

    def get_usage(self):
        if not self.count_usage:
            raise RuntimeError("this store does not count usage")
        return self.use_count / self.life_count

    def get_all_sliced(self, start, end):
        n = self.n
        stop = n + end if end < 0 else (n if end == 0 else end)
        sl = lambda x: None if x is None else x[..., start:stop]
        return sl(self.k), sl(self.s), sl(self.e), sl(self.get_usage())


class FastMemoryManager(MemoryManager):
    """MemoryManager on banks: add_memory / compress_features are the parent's code (they only use the store interface); the per-frame read
    is one library call on the [long | work] column range"""

    def __init__(self, config, network):
        self.network, self.ctx = network, network.ctx
        self._banks = None
        super().__init__(config, device_index=network.ctx.device_id, backend=object())      # no backend: every read goes through the banks
        self.backend = None
        self.work_mem = self.long_mem = None                   # created with the banks, at the first add_memory (sizes are known then)

    def _make_banks(self, key, value):
        CK, hw = key.shape[1], key.shape[-2] * key.shape[-1]
        objs, CV = value.shape[1], value.shape[2]
        if self.enable_long_term:
            cap_long = self.max_long_elements + self.num_prototypes
            cap_work = (self.max_mt_frames + 2) * hw
        else:
            cap_long, cap_work = 0, 16 * hw
        self._banks = Banks(key.device, CK, objs, CV, cap_long, cap_work, has_selection=self.enable_long_term)
        self.work_mem = BankedStore(self._banks, self.ctx, "work", count_usage=self.enable_long_term)
        self.work_mem.on_grow = self._reserve_read
        if self.enable_long_term:
            self.long_mem = BankedStore(self._banks, self.ctx, "long", count_usage=self.enable_long_term_usage)
        self._hw = hw
        self._reserve_read()

    def _reserve_read(self):
        """the read's scratch (similarity map, top-k lists, usage accumulators) sized for the banks' capacity NOW: growing it with the memory costs a
        device synchronisation per regrowth, in the middle of a clip"""
        nat.check(self.ctx.lib.havc_memory_read_reserve(self.ctx.h, int(self._banks.cap), int(self._hw), int(self.top_k)), self.ctx.h)

    def add_memory(self, key, shrinkage, value, objects, selection=None):
        if self._banks is None:
            self._make_banks(key, value)
        super().add_memory(key, shrinkage, value, objects, selection=selection)

    def compress_features(self):
        """the parent's consolidation, with the dense readout on the library (the candidates are views of the banks: made contiguous here)"""
        from .colormnet_memory import _HipBackend
        self.backend = _HipBackend(self.ctx.device_id)
        self.backend.ctx = self.ctx
        try:
            super().compress_features()
        finally:
            self.backend = None

    def match_memory_into(self, query_key, selection, out):
        """the read of memory_manager.py:58-150 into `out` [objects * CV, H * W] (preallocated); usage counters updated in place"""
        b = self._banks
        n_long = self.long_mem.n if (self.enable_long_term and self.long_mem.engaged()) else 0
        N, col = n_long + self.work_mem.n, b.cap_long - n_long
        CK, HW = query_key.shape[1], query_key.shape[-2] * query_key.shape[-1]
        count = self.enable_long_term
        usage_from = 0 if (n_long and self.enable_long_term_usage) else n_long
        nat.check(self.ctx.lib.havc_memory_read_banked(
            self.ctx.h, _ptr(b.K, col), _ptr(b.S, col), C.c_void_p(query_key.data_ptr()), C.c_void_p(selection.data_ptr()) if selection is not None else None,
            _ptr(b.V, col), C.c_void_p(out.data_ptr()), _ptr(b.use, col) if count else None, _ptr(b.life, col) if count else None, usage_from,
            CK, b.V.shape[0] * b.V.shape[1], N, b.cap, HW, self.top_k), self.ctx.h)
        return out

    def match_memory(self, query_key, selection):
        import torch
        h, w = query_key.shape[-2:]
        out = torch.empty((self._banks.V.shape[0] * self._banks.V.shape[1], h * w), dtype=torch.float32, device=query_key.device)
        return self.match_memory_into(query_key, selection, out).view(self._banks.V.shape[0], self._banks.V.shape[1], h, w)


class FastInferenceCore(InferenceCore):
    """InferenceCore.step / step_AnyExemplar (inference_core.py:45-230) on the fast-step entry points.  The frame arrives PADDED (the render's
    havc_cmn_frame_in wrote the L plane three times into a zero-padded image) together with its pads; results are the padded ab planes."""

    def __init__(self, network, config, device_index=0):
        super().__init__(network, config, device_index=device_index, memory_backend=None)

    def clear_memory(self):
        self.curr_ti = -1
        self.last_mem_ti = 0
        if not self.deep_update_sync:
            self.last_deep_update_ti = -self.deep_update_every
        self.drop_read_ahead()                                 # a read in flight on the second stream belongs to the memory being dropped
        self.memory = FastMemoryManager(self.config, self.network)

    # ---- the two pieces every frame is made of ----
    def hint_next(self, entry):
        """the frame AFTER the one about to be stepped is a plain propagation frame (no exemplar, no processor reset) whose look-ahead entry
        (colormnet_net.prefetch_frames) is `entry`: its read may run under this frame's decoder (READ_AHEAD)"""
        self._next = entry

    def drop_read_ahead(self, keep_hint=False):
        """forget a read enqueued ahead of its frame (the memory is being replaced, or another processor is about to use the context's second
        stream and read scratch): the main stream waits for it, its usage update never happens; the frame it was for is read again when it is stepped"""
        if getattr(self, "_ahead_read", None) is not None:
            nat.check(self.network.ctx.lib.havc_cmn_side_wait(self.network.ctx.h, 0), self.network.ctx.h)
            if getattr(self.network, "_side_owner", None) is self:
                self.network._side_owner = None
        self._ahead_read = None
        if not keep_hint:
            self._next = None

    def _issue_read(self, B, key, selection, readout, with_short_term):
        net = self.network
        if with_short_term:                                    # forked onto the second stream: runs next to the memory read
            net.short_term_fork(B, key, self.last_ti_key, self.last_ti_value)
        self.memory.match_memory_into(key, selection, readout)
        if with_short_term:
            net.short_term_join(B, readout)

    def _read(self, key, selection, feats, normal, with_short_term=True, is_mem=True):
        net = self.network
        f = feats[0]
        B = net.fast_buffers(*f.shape)
        owner = getattr(net, "_side_owner", None)
        if owner is not None and owner is not self:            # two processors stepping alternately on one network: the other one's read-ahead holds the
            owner.drop_read_ahead()                            # second stream and the read's scratch -- it is dropped (and redone by its owner), not raced
        ahead, self._ahead_read = getattr(self, "_ahead_read", None), None
        if ahead is not None:
            net._side_owner = None
        hit = ahead is not None and ahead[0] is key and with_short_term
        if ahead is not None:                                  # (a miss: the caller stepped another frame than the hinted one -- the read is dropped, its
            nat.check(net.ctx.lib.havc_cmn_side_wait(net.ctx.h, 1 if hit else 0), net.ctx.h)      #  usage update never happens)
        if hit:
            readout = ahead[1]
        else:
            readout = B.readout
            self._issue_read(B, key, selection, readout, with_short_term)
        nxt, self._next = getattr(self, "_next", None), None
        ahead_ok = READ_AHEAD and nxt is not None and not is_mem and self.last_ti_key is not None and nxt[3].shape == f.shape
        if ahead_ok:
            # The read of frame t+1 needs t+1's key (look-ahead pass) and the banks / last memory frame, which this frame does not change (it is
            # not a memory frame): it goes onto the second stream, under this frame's decoder (inference_core.py:119-230 reads, then segments).
            # The second stream starts behind the main stream's work UP TO HERE (mark); the decoder is enqueued first, so that the main stream does
            # not sit idle while the host issues the ~12 launches of the read.
            net.wait_prefetched(nxt)
            nat.check(net.ctx.lib.havc_cmn_side_mark(net.ctx.h), net.ctx.h)
        hidden_in = self.memory.get_hidden()
        hidden_out = B.other_hidden(hidden_in) if normal else None
        net.segment_fast(B, f, hidden_in, hidden_out, readout)
        if normal:
            self.memory.set_hidden(hidden_out)
        if ahead_ok:
            other = B.readout2 if readout is B.readout else B.readout
            nat.check(net.ctx.lib.havc_cmn_side_begin(net.ctx.h), net.ctx.h)
            try:
                self._issue_read(B, nxt[0], nxt[2], other, True)
            finally:
                nat.check(net.ctx.lib.havc_cmn_side_end(net.ctx.h), net.ctx.h)
            self._ahead_read = (nxt[0], other)
            net._side_owner = self
            self.reads_ahead = getattr(self, "reads_ahead", 0) + 1
        return B.prob                                          # [2, H, W] padded ab planes

    def _memorise(self, image, key, shrinkage, selection, f16, planes, deep):
        net = self.network
        B = net.fast_buffers(*f16.shape)
        hidden_in = self.memory.get_hidden()
        hidden_out = B.other_hidden(hidden_in) if deep else None
        net.encode_value_fast(B, image, f16, planes, hidden_in, hidden_out)
        h, w = key.shape[-2:]
        self.memory.add_memory(key, shrinkage, B.value.view(1, 2, -1, h, w), self.all_labels, selection=selection if self.enable_long_term else None)
        self.last_mem_ti = self.curr_ti
        net.keep_last(B, key)                                  # the short-term attention of the next frames reads copies (the entry's tensors are recycled)
        self.last_ti_key, self.last_ti_value = B.last_key, B.last_value
        if deep:
            self.memory.set_hidden(hidden_out)
            self.last_deep_update_ti = self.curr_ti

    def _pad_like(self, t, pad):
        """a rare operand (the reference image's planes) padded like the frame"""
        import torch.nn.functional as F
        return F.pad(t, pad) if any(pad) else t

    # ---- inference_core.py:45-117 ----
    def step_padded(self, image, pad, mask=None, valid_labels=None, end=False):
        """image [3, H, W] padded; mask [2, h0, w0] (ab of this very frame, unpadded) or None -> padded ab planes [2, H, W] (None before the first mask)"""
        self.curr_ti += 1
        self.pad = pad
        image = image.unsqueeze(0)
        is_mem, deep, normal = self._schedule(mask is not None, end)
        need_segment = self.curr_ti > 0 and self._labels_differ(valid_labels)
        key, shrinkage, selection, f16, f8, f4 = self.network.encode_key(image, need_ek=(self.enable_long_term or need_segment), need_sk=is_mem)
        if not need_segment:                                   # no _read will consume (or drop) a read enqueued ahead for this frame: nothing may touch the
            self.drop_read_ahead()                             # memory, the second stream or the read's scratch before the main stream has waited for it
        planes = self._read(key, selection, (f16, f8, f4), normal, is_mem=is_mem) if need_segment else None
        self._next = None
        if mask is not None:
            planes = self._pad_like(mask, pad).contiguous()
            self.memory.create_hidden_state(2, key)
        if is_mem:
            self._memorise(image[0], key, shrinkage, selection, f16, planes, deep)
        return planes

    # ---- inference_core.py:119-230 ----
    def step_AnyExemplar_padded(self, image, pad, ref_image=None, msk_ab=None, valid_labels=None, end=False, flag_FirstframeIsExemplar=False):
        """image [3, H, W] padded; ref_image [3, H, W] = the reference image's L plane three times, padded; msk_ab [2, h0, w0] its ab planes"""
        self.curr_ti += 1
        self.pad = pad
        image = image.unsqueeze(0)
        is_mem, deep, normal = self._schedule(msk_ab is not None, end)
        exemplar = msk_ab is not None and not flag_FirstframeIsExemplar
        need_segment = (self.curr_ti > 0 if flag_FirstframeIsExemplar else self.curr_ti >= 0) and self._labels_differ(valid_labels)
        key, shrinkage, selection, f16, f8, f4 = self.network.encode_key(image, need_ek=(self.enable_long_term or need_segment), need_sk=is_mem)
        planes = None
        if exemplar or not need_segment:                       # an exemplar's add_memory below runs BEFORE this frame's _read, and without need_segment no _read runs at
            self.drop_read_ahead(keep_hint=True)               # all: a read enqueued ahead (the caller announced a plain frame) is waited for and forgotten first
        if exemplar:
            need_segment, deep = True, False
            ref = ref_image.unsqueeze(0)
            rkey, rshrink, rsel, rf16, _, _ = self.network.encode_key(ref, need_ek=True, need_sk=is_mem)
            planes = self._pad_like(msk_ab, pad).contiguous()
            self.memory.create_hidden_state(2, key)
            rvalue, _ = self.network.encode_value(ref, rf16, self.memory.get_hidden(), planes.unsqueeze(0), is_deep_update=False)
            try:
                self.memory.add_memory(rkey, rshrink, rvalue, self.all_labels, selection=rsel if self.enable_long_term else None)
                self.last_mem_ti = self.curr_ti
                self.last_ti_key, self.last_ti_value = rkey, rvalue
            except Exception:                                  # inference_core.py:172-180 swallows a failing add here; so does the drop-in
                pass
        if need_segment:
            planes = self._read(key, selection, (f16, f8, f4), normal, with_short_term=not exemplar, is_mem=is_mem)
        self._next = None
        if msk_ab is not None and flag_FirstframeIsExemplar:
            planes = self._pad_like(msk_ab, pad).contiguous()
        if is_mem:
            self._memorise(image[0], key, shrinkage, selection, f16, planes, deep)
        return planes


def frame_pads(h, w):
    """pad_divide_by(.., 112) (util/tensor_util.py:18-36): (left, right, top, bottom) and the padded size"""
    eh, ew = (-h) % DIVIDE_BY, (-w) % DIVIDE_BY
    pad = (ew // 2, ew - ew // 2, eh // 2, eh - eh // 2)
    return pad, h + eh, w + ew
