"""CPU: the host logic of the precise mode (vsdeoldify_amd/plan.py split_weights / pack_conv(precise=True), PlanBuilder(precise=True),
DeoldifyGenerator(precision="precise")) -- a numpy restatement of what the conv kernel's K walk computes from the packed weights, against float64."""
import numpy as np
import pytest

from tests.gpu_util import hl_split
from vsdeoldify_amd import _native as nat
from vsdeoldify_amd.plan import PlanBuilder, WeightPack, pack_conv, split_weights


def test_split_weights_three_segments_reproduce_the_fp32_product():
    r = np.random.default_rng(0)
    for amp in (0.05, 1.0, 200.0):                     # 200: needs the power-of-two pre-scale (2^11 w_hi must fit fp16)
        w = (r.standard_normal((48, 256)) * amp).astype(np.float32)
        x = (r.standard_normal((20, 256)) * 3).astype(np.float32)
        packed, pscale = split_weights(w)
        assert packed.dtype == np.float16 and packed.shape == (48, 768) and np.isfinite(packed.astype(np.float32)).all()
        K = 256
        xh, xl = hl_split(x)
        # what the MFMA main loop accumulates over the three K segments (fp16 operands, exact products, wide accumulator)
        acc = (xh.astype(np.float64) @ packed[:, :K].astype(np.float64).T + xh.astype(np.float64) @ packed[:, K:2 * K].astype(np.float64).T +
               xl.astype(np.float64) @ packed[:, 2 * K:].astype(np.float64).T)
        got = acc * pscale
        ref = x.astype(np.float64) @ w.astype(np.float64).T
        scale = np.abs(x).astype(np.float64) @ np.abs(w).astype(np.float64).T          # sum of |terms|: the natural error unit
        assert (np.abs(got - ref) / scale).max() < 2.0 ** -21, float((np.abs(got - ref) / scale).max())
        plain = xh.astype(np.float64) @ w.astype(np.float16).astype(np.float64).T
        assert (np.abs(plain - ref) / scale).max() > 2.0 ** -14                         # the fast path's operands, for contrast


def test_precise_pack_and_plan_layout():
    r = np.random.default_rng(1)
    pack, b = WeightPack(), PlanBuilder(precise=True)
    x = b.tensor(12, 10, 259)                          # span 264 -> pitch 320 -> hi | lo = 640
    assert x.cpitch == 640 and x.span == 264 and b.bufs[x.buf][0] == 12 * 10 * 640
    W = r.standard_normal((259, 259, 3, 3)).astype(np.float32) / 48
    fast = pack_conv(WeightPack(), W, x.cmap, x.span)
    pc = pack_conv(pack, W, x.cmap, x.span, precise=True)
    assert pc.Kc == 3 * fast.Kc and pc.Npad == fast.Npad and pc.C8a == fast.C8a and pc.pscale == 2.0 ** -11
    y = b.tensor(12, 10, 259)
    oi = b.conv("c", pc, x, y, pad=1, flags=nat.F_RELU_PRE)
    op = b.ops[oi]
    assert op["flags"] & nat.F_PRECISE and op["Kc"] == pc.Kc and op["f3"] == np.float32(2.0 ** -11) and op["src_cpitch"] == 640
    with pytest.raises(AssertionError):
        b.conv("mixed", fast, x, y, pad=1)             # a fast packing cannot enter a precise plan
    with pytest.raises(AssertionError):
        b.ew("no precise form", x, b.tensor(12, 10, 259))          # ColorMNet's element-wise op: fast path only


def test_precise_generator_plan_is_the_unfused_plan_with_pairs():
    from vsdeoldify_amd.deoldify_net import DeoldifyGenerator
    from vsdeoldify_amd.synth import synth_state_dict
    sd = synth_state_dict("deep", 3)
    g = DeoldifyGenerator(sd, "deep", precision="precise")
    f = DeoldifyGenerator(sd, "deep", fuse_final=False, fuse_blur=False)
    ops, bufs, _, _, names = g.plan(96)
    ops_f, bufs_f, _, _, names_f = f.plan(96)
    assert [n for n in names if ".value" not in n] == [n for n in names_f if ".value" not in n]
    assert int(ops["flops"].sum()) == int(ops_f["flops"].sum())                       # algorithmic FLOPs do not count the three segments
    for o, n in zip(ops, names):
        assert o["flags"] & nat.F_PRECISE, n
        if o["type"] == nat.OP_CONV:
            assert not o["flags"] & (nat.F_PS_BLUR | nat.F_FUSE_RGB8) and o["f3"] > 0, n
            assert bool(o["flags"] & nat.F_OUT_TRANSPOSED) == n.endswith(".value"), n       # round 5: the attention's value map is stored as transposed hi / lo planes
    conv = {n: o for o, n in zip(ops, names) if o["type"] == nat.OP_CONV}
    conv_f = {n: o for o, n in zip(ops_f, names_f) if o["type"] == nat.OP_CONV}
    for n in conv:
        assert conv[n]["Kc"] == 3 * conv_f[n]["Kc"] and conv[n]["src_cpitch"] == 2 * conv_f[n]["src_cpitch"], n


@pytest.mark.parametrize("model", ["eccv16", "siggraph17", "ddcolor"])
def test_precise_plans_of_the_other_models(model):
    """round 5: ZhangGenerator / DDColorGenerator(precision="precise") emit plans whose every op is flagged HAVC_F_PRECISE and has a precise form,
    with pair pitches, three-segment convs and the algorithmic FLOPs of the fast plan (the three K segments are not counted)"""
    from vsdeoldify_amd.plan import PRECISE_OPS
    if model == "ddcolor":
        from vsdeoldify_amd.ddcolor_net import DDColorGenerator
        from vsdeoldify_amd.synth import synth_ddcolor_state_dict
        small = dict(depths=(1, 1, 2, 1), dec_layers=3)
        sd = synth_ddcolor_state_dict(1, **small)
        gp, gf = DDColorGenerator(sd, precision="precise", **small), DDColorGenerator(sd, **small)
        pp, pf = gp.plan(64), gf.plan(64)
        consts = pp[5]
        assert consts and all(pitch % 16 == 0 for _, _, pitch, _ in consts)            # constant token / position maps travel as pairs too
    else:
        from vsdeoldify_amd.synth import synth_zhang_state_dict
        from vsdeoldify_amd.zhang_net import ZhangGenerator
        sd = synth_zhang_state_dict(model, 1)
        gp, gf = ZhangGenerator(sd, model, precision="precise"), ZhangGenerator(sd, model)
        pp, pf = gp.plan(64), gf.plan(64)
    ops, ops_f = pp[0], pf[0]
    assert all(int(o["flags"]) & nat.F_PRECISE for o in ops) and all(int(o["type"]) in PRECISE_OPS for o in ops)
    assert not any(int(o["flags"]) & nat.F_PRECISE for o in ops_f)
    conv, conv_f = ops[ops["type"] == nat.OP_CONV], ops_f[ops_f["type"] == nat.OP_CONV]
    assert len(conv) and all(float(o["f3"]) > 0 for o in conv) and int(conv["Kc"].sum()) == 3 * int(conv_f["Kc"].sum())
    assert int(conv["flops"].sum()) == int(conv_f["flops"].sum())
    assert all(int(o["src_cpitch"]) % 16 == 0 for o in conv)
