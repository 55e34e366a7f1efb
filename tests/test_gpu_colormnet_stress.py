"""Race hunt for the ColorMNet frame loop under stream jitter (tools/cmn_race_stress.py holds the clip and the comparison).

The reference steps a frame on ONE stream (colormnet/inference/inference_core.py:119-230); the product schedule uses three (the step's stream, the
context's second stream for the read of frame t+1 / the short-term attention, the look-ahead context's stream for the key encoder).  Whatever the
relative timing of those streams, every frame must carry the bytes of the one-stream schedule: the library's delay kernels
(havc_debug_stream_jitter) move them against each other and every frame's SHA-1 is compared with the one-stream baseline.  Fail-fast: no retry.
The long form is on record in profiles/r6_cmn_race_stress.txt: 85 400 jittered clips, ONE mismatch (cause unknown, ~1.2e-5 per clip: DESIGN.md section 9) -- if this
test ever fails, that event has shown up again: keep the printed description (first differing frame, usage counters)."""
import os
import sys

import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))


@pytest.mark.gpu
def test_gpu_jittered_streams_give_the_bytes_of_the_one_stream_schedule():
    import cmn_race_stress as S
    from tests.test_colormnet_net import gpu_network
    from vsdeoldify_amd import _native as nat
    net = gpu_network()
    lib = nat.load()
    frames, ref, _ = S.make_clip(40)
    lib.havc_debug_stream_jitter(0, 1)
    net.async_lookahead = False
    try:
        base = S.run_clip(net, frames, ref, 8, False)
    finally:
        net.async_lookahead = True
    assert base[1][1] > 0, base[1]                                             # the long-term memory was engaged: consolidations happened
    plain = S.run_clip(net, frames, ref, 8, True)
    assert plain[4] >= 10, plain[4]                                            # reads really ran ahead
    assert plain[0] == base[0] and plain[1] == base[1], S.describe("un-jittered", plain, base)
    for i in range(10):
        lib.havc_debug_stream_jitter(77 + i, 100 + 40 * i)
        try:
            got = S.run_clip(net, frames, ref, 8, True)
        finally:
            lib.havc_debug_stream_jitter(0, 1)
        assert got[0] == base[0] and got[1] == base[1], (i, S.describe(f"jitter seed {77 + i}", got, base))
