"""CPU: the host-side schedule of ColorMNet's read-ahead (vsdeoldify_amd/colormnet_fast.py FastInferenceCore._read, round 5) on a recording stand-in
for the network and the library -- no GPU, no kernels: WHICH calls are made, in WHICH order, on WHICH readout buffer.

The reference reads, then segments, every frame (colormnet/inference/inference_core.py:119-230).  The drop-in may enqueue the read of frame t+1 on the
context's second stream before frame t's own decoder has run, because that read depends on t+1's key (look-ahead pass), the banks and the last memory
frame -- none of which a NON-memory frame changes.  What has to hold for that to be the same computation:
  * the read a frame consumes is the one issued for exactly that frame (identity of its key tensor), joined (side_wait) BEFORE its decoder, with its
    usage update applied exactly once (apply = 1) -- or dropped without one (apply = 0) when the caller stepped another frame;
  * no read-ahead is issued on a memory frame (the banks are about to change) or without a hint;
  * the second stream starts behind the main stream's work up to the mark, and the decoder of frame t is enqueued between mark and begin;
  * the decoder of t reads one readout buffer while the read of t+1 fills the other;
  * a second processor stepping on the same network first drops the other one's pending read.
The GPU tests (tests/test_colormnet_net.py) check the bytes; this one pins the protocol."""
import types

import pytest

from vsdeoldify_amd import colormnet_fast as cf


class _Lib:
    def __init__(self, log):
        self.log = log

    def __getattr__(self, name):
        def call(*a):
            self.log.append((name,) + tuple(x for x in a[1:] if isinstance(x, int)))
            return 0
        return call


class _Buf:
    def __init__(self, name):
        self.name = name

    def __repr__(self):
        return self.name


class _Net:
    def __init__(self):
        self.log = []
        self.ctx = types.SimpleNamespace(lib=_Lib(self.log), h=None)
        self.B = types.SimpleNamespace(readout=_Buf("R1"), readout2=_Buf("R2"), prob=_Buf("prob"), other_hidden=lambda cur: _Buf("H'"))

    def fast_buffers(self, *shape):
        return self.B

    def short_term_fork(self, B, key, lk, lv):
        self.log.append(("short_term_fork", key.name))

    def short_term_join(self, B, readout):
        self.log.append(("short_term_join", readout.name))

    def segment_fast(self, B, f, hin, hout, readout):
        self.log.append(("segment", readout.name, hout is not None))

    def wait_prefetched(self, entry):
        self.log.append(("wait_prefetched", entry[0].name))


class _Mem:
    def __init__(self, log):
        self.log = log

    def match_memory_into(self, key, selection, out):
        self.log.append(("memory_read", key.name, out.name))

    def get_hidden(self):
        return _Buf("H")

    def set_hidden(self, h):
        pass


def _core(net):
    c = object.__new__(cf.FastInferenceCore)
    c.network, c.memory = net, _Mem(net.log)
    c.last_ti_key, c.last_ti_value = _Buf("lastK"), _Buf("lastV")
    c._ahead_read = c._next = None
    return c


def _entry(name):
    feat = types.SimpleNamespace(shape=(224, 448))
    return (_Buf(name), None, _Buf(name + ".sel"), feat, None, None)


FEAT = types.SimpleNamespace(shape=(224, 448))


def _step(core, entry, is_mem=False, hint=None, normal=True):
    core.network.log.clear()
    if hint is not None:
        core.hint_next(hint)
    core._read(entry[0], entry[2], (FEAT, FEAT, FEAT), normal, with_short_term=True, is_mem=is_mem)
    return list(core.network.log)


def test_read_ahead_protocol_over_plain_memory_and_unannounced_frames():
    assert cf.READ_AHEAD
    net = _Net()
    core = _core(net)
    e = [_entry(f"k{i}") for i in range(6)]
    # frame 0: nothing pending -> inline read into R1; hinted next plain frame -> mark, decoder, then the section for k1 into R2
    log = _step(core, e[0], hint=e[1])
    assert log == [("short_term_fork", "k0"), ("memory_read", "k0", "R1"), ("short_term_join", "R1"), ("wait_prefetched", "k1"), ("havc_cmn_side_mark",),
                   ("segment", "R1", True), ("havc_cmn_side_begin",), ("short_term_fork", "k1"), ("memory_read", "k1", "R2"), ("short_term_join", "R2"),
                   ("havc_cmn_side_end",)], log
    assert core.reads_ahead == 1 and net._side_owner is core
    # frame 1 (hit): joined with its usage update, decoder on R2, the next section fills R1
    log = _step(core, e[1], hint=e[2])
    assert log[0] == ("havc_cmn_side_wait", 1) and ("memory_read", "k1", "R1") not in log and ("memory_read", "k1", "R2") not in log
    assert log.index(("havc_cmn_side_mark",)) < log.index(("segment", "R2", True)) < log.index(("havc_cmn_side_begin",))
    assert ("memory_read", "k2", "R1") in log and log[-1] == ("havc_cmn_side_end",)
    # frame 2 is a MEMORY frame (hit): consumed, but nothing may run ahead of the banks it is about to change -- even with a hint
    log = _step(core, e[2], is_mem=True, hint=e[3], normal=False)
    assert log == [("havc_cmn_side_wait", 1), ("segment", "R1", False)], log
    assert core._ahead_read is None and net._side_owner is None
    # frame 3: inline again (into R1), no hint -> no section
    log = _step(core, e[3])
    assert log == [("short_term_fork", "k3"), ("memory_read", "k3", "R1"), ("short_term_join", "R1"), ("segment", "R1", True)], log
    # frame 4 hints k5, but the caller steps another frame: the pending read is dropped WITHOUT its usage update and the stepped frame is read inline
    _step(core, e[4], hint=e[5])
    other = _entry("kX")
    log = _step(core, other)
    assert log[:4] == [("havc_cmn_side_wait", 0), ("short_term_fork", "kX"), ("memory_read", "kX", "R1"), ("short_term_join", "R1")], log
    assert ("havc_cmn_side_begin",) not in log


def test_a_second_processor_on_the_network_drops_the_first_ones_read_ahead_and_switching_it_off():
    net = _Net()
    a, b = _core(net), _core(net)
    _step(a, _entry("a0"), hint=_entry("a1"))
    assert net._side_owner is a and a._ahead_read is not None
    log = _step(b, _entry("b0"))
    assert log[0] == ("havc_cmn_side_wait", 0) and a._ahead_read is None and net._side_owner is None       # dropped before b touches stream 2 / the scratch
    assert ("memory_read", "b0", "R1") in log
    log = _step(a, _entry("a1"))                                                                             # a reads its frame again, inline, exactly once
    assert log.count(("memory_read", "a1", "R1")) == 1 and ("havc_cmn_side_wait", 1) not in log
    cf.READ_AHEAD = False
    try:
        log = _step(a, _entry("a2"), hint=_entry("a3"))
        assert ("havc_cmn_side_mark",) not in log and ("havc_cmn_side_begin",) not in log
    finally:
        cf.READ_AHEAD = True


def test_dropping_a_pending_read_on_reset():
    net = _Net()
    core = _core(net)
    _step(core, _entry("k0"), hint=_entry("k1"))
    net.log.clear()
    core.drop_read_ahead()
    assert net.log == [("havc_cmn_side_wait", 0)] and core._ahead_read is None and net._side_owner is None
    core.drop_read_ahead()
    assert net.log == [("havc_cmn_side_wait", 0)]                                                            # nothing pending: no call


def test_a_pending_read_is_dropped_before_a_step_that_has_no_read_or_memorises_first():
    """round 6: a read enqueued ahead used to be waited for only inside the next _read.  A hinted frame that is stepped WITHOUT a _read (nothing to
    segment), or an exemplar frame -- whose reference image is memorised before the frame's own read (inference_core.py:160-180) -- went on to add_memory on the
    main stream while the second stream could still be reading the banks.  Both step functions now drop the pending read first (havc_cmn_side_wait(0))."""
    import torch
    net = _Net()
    net.encode_key = lambda image, need_ek=True, need_sk=True: (_Buf("kN"), None, _Buf("kN.sel"), FEAT, FEAT, FEAT)
    net.encode_value = lambda *a, **k: (_Buf("rv"), None)
    core = _core(net)
    core.curr_ti, core.enable_long_term, core.all_labels = 0, False, [1, 2]
    core._schedule = lambda has_mask, end: (False, False, True)
    core.memory.create_hidden_state = lambda n, k: net.log.append(("create_hidden",))
    core.memory.add_memory = lambda *a, **k: net.log.append(("add_memory",))
    _step(core, _entry("k0"), hint=_entry("k1"))
    assert core._ahead_read is not None
    # (1) the hinted frame arrives with the labels already known: need_segment is False, no _read runs -> the pending section is joined and forgotten first
    core._labels_differ = lambda valid: False
    net.log.clear()
    core.step_padded(torch.zeros(3, 4, 4), (0, 0, 0, 0), mask=None, valid_labels=[1, 2])
    assert net.log[0] == ("havc_cmn_side_wait", 0) and core._ahead_read is None and net._side_owner is None, net.log
    # (2) an exemplar frame with a read pending: the drop comes before the reference image's add_memory
    core._labels_differ = lambda valid: True
    _step(core, _entry("k2"), hint=_entry("k3"))
    assert core._ahead_read is not None
    net.log.clear()
    core.step_AnyExemplar_padded(torch.zeros(3, 4, 4), (0, 0, 0, 0), ref_image=torch.zeros(3, 4, 4), msk_ab=torch.zeros(2, 4, 4), valid_labels=None)
    assert net.log.index(("havc_cmn_side_wait", 0)) < net.log.index(("add_memory",)), net.log
