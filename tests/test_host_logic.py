"""CPU (-m "not gpu"): host-side logic — C ABI surface, struct layout, weight folding/packing, plan emission,
frame sharding and the world_size-2 timing protocol on gloo."""
import ctypes
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

from tests.conftest import ROOT
from vsdeoldify_amd import _native as nat
from vsdeoldify_amd import plan as P
from vsdeoldify_amd.sharded import gather_order, shard_frames
from vsdeoldify_amd.synth import synth_state_dict

HEADER = os.path.join(ROOT, "include", "havc_mi355.h")


def test_library_loads_and_exports_every_declared_symbol():
    lib = nat.load()
    declared = set(re.findall(r"\b(havc_[a-z0-9_]+)\s*\(", open(HEADER).read()))
    bound = {n for n, _, _ in nat.SYMBOLS}
    assert declared == bound, declared ^ bound
    for name in declared:
        assert hasattr(lib, name)
    assert lib.havc_version().decode().startswith("havc_mi355")


def test_no_device_fails_loudly():
    lib = nat.load()
    if lib.havc_device_count() > 0:
        pytest.skip("a GPU is visible")
    with pytest.raises(nat.NativeLibraryError):
        nat.Context(0)
    from vsdeoldify_amd.render import ModelImageRender
    with pytest.raises(nat.NativeLibraryError):            # no CPU fallback, device_index=99 refused as well
        ModelImageRender(None, "video", 2, 0.0, device_index=99, state_dicts={})


def test_op_struct_layout_matches_c(tmp_path):
    src = tmp_path / "sz.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "havc_mi355.h"\nint main(){printf("%zu %zu %zu %zu %zu %zu %zu\\n",'
                   "sizeof(havc_op),offsetof(havc_op,w_off),offsetof(havc_op,f0),offsetof(havc_op,flops),offsetof(havc_op,tag),"
                   "offsetof(havc_op,out_ox),sizeof(havc_buf));return 0;}\n")
    exe = tmp_path / "sz"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    got = [int(x) for x in subprocess.check_output([str(exe)]).split()]
    d = nat.OP_DTYPE
    assert got == [d.itemsize, d.fields["w_off"][1], d.fields["f0"][1], d.fields["flops"][1], d.fields["tag"][1],
                   d.fields["out_ox"][1], nat.BUF_DTYPE.itemsize]


def test_python_constants_match_the_header():
    """every HAVC_OP_* value, conv epilogue flag HAVC_F_* and element-wise flag HAVC_EW_* of include/havc_mi355.h against the constants of
    vsdeoldify_amd/_native.py the plan emitters write into the op records (a renumbered op would still build, load and run the WRONG kernel)"""
    text = open(HEADER).read()
    ops = {k: int(v) for k, v in re.findall(r"\bHAVC_(OP_[A-Z0-9_]+)\s*=\s*(\d+)", text)}
    assert len(ops) >= 28 and len(set(ops.values())) == len(ops), "op values are unique"
    for name, val in ops.items():
        assert getattr(nat, name) == val, (name, val, getattr(nat, name, None))
    defs = {k: int(v, 0) for k, v in re.findall(r"#define\s+HAVC_((?:F|EW)_[A-Z0-9_]+)\s+(0x[0-9a-fA-F]+|\d+)\b", text)}
    checked = 0
    for name, val in defs.items():
        if hasattr(nat, name):
            assert getattr(nat, name) == val, (name, hex(val), hex(getattr(nat, name)))
            checked += 1
    assert checked >= 15, checked


def test_gelu_epilogue_constants_stay_within_their_stated_error():
    """gelu_erf of csrc/conv_common.h (round 5: Phi as a logistic function of an odd polynomial, tools/fit_gelu.py) restated in numpy float32 from the
    constants IN THE SOURCE FILE, against the exact v Phi(v) (nn.GELU(), the ConvNeXt block / ViT MLP of the reference's models): |error| < 1.3e-5 for
    every fp16-representable |v| <= 16 and exact limits beyond the clamp -- the bound the kernel comment and DESIGN.md section 8 state."""
    from scipy.special import ndtr
    src = open(os.path.join(ROOT, "vsdeoldify_amd", "csrc", "conv_common.h")).read()
    body = src[src.index("__device__ __forceinline__ float gelu_erf(float v)"):]
    body = body[:body.index("\n}\n")]
    c = [np.float32(x) for x in re.findall(r"(-?\d\.\d+e?-?\d*)f \* -L2E", body)]
    assert len(c) == 4 and "fminf(v * v, 36.0f)" in body, (c, body[:200])
    c3, c2, c1, c0 = c                                                       # source order: v^6, v^4, v^2, v^0 coefficients of P
    L2E = np.float32(1.44269504088896341)
    v = np.arange(-16, 16, 2.0 ** -9, dtype=np.float64).astype(np.float16).astype(np.float32)
    v = np.unique(np.concatenate([v, np.float32([-60000, -20, 20, 60000, 0.0])]))
    v2 = np.minimum(v * v, np.float32(36.0))
    p = (np.float32(c3 * -L2E) * v2 + np.float32(c2 * -L2E)).astype(np.float32)
    p = (p * v2 + np.float32(c1 * -L2E)).astype(np.float32)
    p = (p * v2 + np.float32(c0 * -L2E)).astype(np.float32)
    with np.errstate(over="ignore"):
        e = np.exp2((p * v).astype(np.float32)).astype(np.float32)
        got = (v * (np.float32(1.0) / (np.float32(1.0) + e))).astype(np.float32)
    want = v.astype(np.float64) * ndtr(v.astype(np.float64))
    err = np.abs(got - want)
    assert np.isfinite(got).all() and err.max() < 1.3e-5, (float(err.max()), float(v[err.argmax()]))
    assert got[v == 0][0] == 0 and got[v == 60000][0] == 60000 and got[v == -60000][0] == 0


def test_norm_folds_match_oracle():
    from oracle import unet as ou
    sd = synth_state_dict("deep", 4)
    tsd = {k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}
    for p in ("layers.3.0.0", "layers.10.layers.0.0", "layers.11.0"):
        assert np.abs(P.fold_spectral(sd, p) - ou.fold_spectral(tsd, p).numpy()).max() < 1e-6
        assert abs(float(np.dot(sd[p + ".weight_u"], sd[p + ".weight_orig"].reshape(len(sd[p + ".weight_u"]), -1) @ sd[p + ".weight_v"])) - 1) > 1e-2, \
            "synthetic sigma must not be 1 (the fold would be untested)"
    assert np.abs(P.fold_weightnorm(sd, "layers.8.conv.0") - ou.fold_weightnorm(tsd, "layers.8.conv.0").numpy()).max() < 1e-6


def test_pack_conv_layout():
    r = np.random.default_rng(0)
    W = r.standard_normal((5, 11, 3, 3)).astype(np.float32)
    cmap = np.concatenate([np.arange(6), 8 + np.arange(5)])            # two concatenated segments 6 | 5 in a 16-wide span
    pack = P.WeightPack()
    pc = P.pack_conv(pack, W, cmap, 16, bias=np.arange(5, dtype=np.float32))
    blob = pack.blob()
    assert pc.Npad == 16 and pc.Kc == 24 and pc.Ci == 16            # K = 9 taps * 2 chunks = 18 -> padded to 24
    flat = np.frombuffer(blob, np.float16, pc.Npad * pc.Kc * 8, pc.w_off).reshape(16, 24 * 8)
    Wt = flat[:, :9 * 16].reshape(16, 3, 3, 16).astype(np.float32)
    assert np.allclose(Wt[:5][..., cmap], W.transpose(0, 2, 3, 1), atol=2e-3)
    assert (Wt[5:] == 0).all() and (flat[:, 9 * 16:] == 0).all() and (Wt[:5][..., [6, 7, 13, 14, 15]] == 0).all()
    bias = np.frombuffer(blob, np.float32, 16, pc.bias_off)
    assert (bias[:5] == np.arange(5)).all() and (bias[5:] == 0).all()
    # pixel-shuffle row order: packed row q*cps + c <- original row c*4 + q
    W2 = r.standard_normal((32, 8, 1, 1)).astype(np.float32)
    pc2 = P.pack_conv(pack, W2, np.arange(8), 8, pixshuf=True)
    f2 = np.frombuffer(pack.blob(), np.float16, 32 * 8 * 8, pc2.w_off).reshape(32, 64)[:, :8].astype(np.float32)
    for q in range(4):
        for c in range(8):
            assert np.allclose(f2[q * 8 + c], W2[c * 4 + q, :, 0, 0], atol=2e-3)


@pytest.mark.parametrize("arch,gflop", [("wide", {256: 282.49, 384: 639.00, 512: 1144.46, 560: 1379.66}),
                                        ("deep", {560: 1915.05})])
def test_plan_flops_match_survey(arch, gflop):
    """the plan's algorithmic FLOPs per pass reproduce SURVEY.md §8a-T1 (conv + attention bmm, 2*MAC)."""
    from vsdeoldify_amd.deoldify_net import DeoldifyGenerator
    gen = DeoldifyGenerator(synth_state_dict(arch, 0), arch)
    for S, want in gflop.items():
        ops, bufs, i, o, names = gen.plan(S)
        assert abs(ops["flops"].sum() / 1e9 - want) < 0.01, (S, ops["flops"].sum() / 1e9)
        assert len(names) == len(ops) and bufs["elem_bytes"][i] == 1 and bufs["elem_bytes"][o] == 1
        for op in ops:                                            # every buffer id / weight offset in range
            assert -1 <= op["src2"] < len(bufs) and 0 <= op["src"] < len(bufs) and 0 <= op["dst"] < len(bufs)
            for f in ("w_off", "bias_off", "scale_off", "shift_off"):
                assert op[f] == -1 or (0 <= op[f] < len(gen.blob) and op[f] % 16 == 0)
    # odd render size (S/16 odd) keeps the 35 vs 36 mismatch the nearest resize has to absorb
    ops = gen.plan(560)[0]
    blur = [op for op in ops if op["type"] == nat.OP_BLUR_RESIZE]
    assert any(op["Hi"] == 36 and op["Ho"] == 35 for op in blur)


def test_shard_frames_partition():
    for n, g in ((64, 8), (10, 4), (3, 8), (0, 2)):
        shards = [shard_frames(n, r, g) for r in range(g)]
        assert sorted(sum(shards, [])) == list(range(n))
        order = gather_order(n, g)
        assert all(shards[r][i] == k for k, (r, i) in enumerate(order))
    with pytest.raises(ValueError):
        shard_frames(4, 2, 2)


WORKER = r"""
import os, sys, time
sys.path.insert(0, {root!r})
import torch
from vsdeoldify_amd.sharded import env_rank, init_dist, shard_frames, timed_steps
rank, local, world = env_rank()
dist = init_dist("gloo", local)
mine = shard_frames(12, rank, world)
done = []
def step(i):
    time.sleep(0.01 * (rank + 1))           # rank 1 is slower: MAX over ranks must report ITS time
    done.append(mine[i % len(mine)])
el = timed_steps(step, 4, 1, lambda: None, dist)
t = torch.tensor([float(len(done))]); dist.all_reduce(t)
if rank == 0:
    print("RESULT", el, t.item())
dist.destroy_process_group()
"""


def test_two_rank_timing_protocol_gloo(tmp_path):
    """N>1 path on CPU: 2 processes, gloo, barrier-bracketed timing, MAX over ranks, disjoint shards."""
    script = tmp_path / "w.py"
    script.write_text(WORKER.format(root=ROOT))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(2)]
    outs = [p.communicate(timeout=120) for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    line = [l for l in outs[0][0].splitlines() if l.startswith("RESULT")][0].split()
    elapsed, total = float(line[1]), float(line[2])
    assert total == 10.0                       # (1 warm-up + 4 timed) x 2 ranks
    assert 0.075 <= elapsed < 1.0              # >= 4 x 20 ms of the slow rank (max over ranks), not rank 0's 40 ms


def test_hue_string_parsers_mirror_the_reference():
    """vsslib/restcolor.py:379-470 parsing rules restated in vsdeoldify_amd.imfilters (host side of havc_image_tweak / _chroma_tweak)"""
    from vsdeoldify_amd import imfilters as F
    assert F.parse_hue_ranges("280:360,0:30") == [280.0, 360.0, 0.0, 30.0]
    assert F.parse_hue_ranges("red,blue-violet") == [0.0, 30.0, 240.0, 270.0]
    assert F.parse_hue_adjust("red,orange|0.5,0.2") == ("red,orange", 0.5, 0, 0.2)
    assert F.parse_hue_adjust("200:260|-60,0.3") == ("200:260", 1.0, -60, 0.3)          # signed first token = hue shift
    assert F.parse_hue_adjust("green|30,0") == ("green", 1.0, 30, 0.0)                  # > 10 is a hue, not a saturation
    assert F.parse_hue_adjust("cyan") == ("cyan", 1.0, 0, 0)
    assert F.parse_hue_adjust("a|b|c") is None and F.parse_hue_adjust("red|x,y") is None
    # luma_adjusted_levels' table: the uint8 wrap of np.add is part of the reference behaviour
    lut = F.luma_levels_lut(0.2, luma_min=0.6)
    i_alpha = int(255 * (0.6 - 0.2))
    assert lut.dtype == np.uint8 and lut[0] == i_alpha and lut[200] == (200 + i_alpha) % 256 and lut[100] == 100 + i_alpha
    assert np.array_equal(F.luma_levels_lut(0.5), np.arange(256, dtype=np.uint8))


def test_ddcolor_plan_shapes_and_flops():
    """the DDColor plan at input 512: op count, algorithmic FLOPs, einsum conv reading its weights from the token buffer"""
    from vsdeoldify_amd import _native as nat
    from vsdeoldify_amd.ddcolor_net import DDColorGenerator, TOK
    from vsdeoldify_amd.synth import synth_ddcolor_state_dict
    sd = synth_ddcolor_state_dict(1, depths=(1, 1, 1, 1), dec_layers=3)
    g = DDColorGenerator(sd, depths=(1, 1, 1, 1), dec_layers=3)
    ops, bufs, i, o, names, consts = g.plan(128)
    assert names[0] == "prep" and names[-1] == "refine_net.0.0"
    # default tail: einsum + refine folded into the epilogue of the last_shuf conv (no 4096-channel / logits tensors in the plan)
    assert "decoder.color_decoder.einsum" not in names and "decoder.last_shuf.shuf+blur" not in names
    cp = ops[names.index("decoder.last_shuf.conv+proj")]
    assert cp["flags"] & nat.F_FUSE_PROJ and cp["Npad"] == 4096 and cp["src2"] >= 0 and cp["aux0"] >= 0
    assert bufs[int(cp["src2"])]["elem_bytes"] == 4 and bufs[int(cp["src2"])]["elems_per_frame"] == 2 * 256
    assert ops[names.index("decoder.color_decoder.fold")]["type"] == nat.OP_FOLD_QUERIES and ops[-1]["type"] == nat.OP_SHUF4_BLUR_AB
    # op-by-op tail (HAVC_DD_FUSE_TAIL=0): the einsum conv reads its weights from the token buffer; same algorithmic FLOPs either way
    os.environ["HAVC_DD_FUSE_TAIL"] = "0"
    try:
        g0 = DDColorGenerator(sd, depths=(1, 1, 1, 1), dec_layers=3)
    finally:
        os.environ.pop("HAVC_DD_FUSE_TAIL")
    ops0, _, _, _, names0, _ = g0.plan(128)
    e = ops0[names0.index("decoder.color_decoder.einsum")]
    assert e["flags"] & nat.F_W_FROM_BUF and e["Npad"] == TOK and e["Kc"] == 32 and e["src2"] >= 0
    assert abs(float(ops["flops"].sum()) / float(ops0["flops"].sum()) - 1) < 1e-3
    mh = [op for op, n in zip(ops, names) if n.endswith(".attn")]
    assert len(mh) == 6 and all(op["type"] == nat.OP_MHA and op["aux1"] >= 0 for op in mh)
    assert {int(op["Ho"]) for op in mh} == {100, 8 * 8, 16 * 16, 32 * 32}                 # self-attention + the three feature levels
    assert len(consts) == 1 + 3 * 3                                                       # query_feat + per layer: q, kv, qkv maps
    for buf, arr, pitch, rows in consts:
        assert arr.shape[0] <= rows and arr.shape[1] <= pitch and np.isfinite(arr).all()


def _sharded_clip_worker(rank, world, port, n, q):
    import os
    import numpy as np
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from vsdeoldify_amd import sharded
    calls = []

    def stub(t):                                     # a "colorizer" whose output identifies frame content AND the rank that ran it
        calls.append(int(t.shape[0]))
        return (255 - t) // 2 + rank
    frames = None
    if rank == 0:
        r = np.random.default_rng(0)
        frames = torch.from_numpy(r.integers(0, 256, (n, 6, 8, 3), dtype=np.uint8))
    out = sharded.colorize_clip_sharded(frames, stub, dist, rank, world, "cpu")
    if rank == 0:
        want = torch.stack([(255 - frames[i]) // 2 + (i % world) for i in range(n)]) if n else frames
        q.put((bool(torch.equal(out, want)), tuple(out.shape), calls))
    else:
        q.put((out is None, None, calls))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n", [7, 4, 1])
def test_sharded_clip_runner_returns_the_clip_in_frame_order_gloo(n):
    """world_size 2, gloo: ONE scatter + ONE gather, frame i coloured by rank i mod 2, result in frame order on rank 0 (odd clip
    length, even, and fewer frames than ranks)."""
    import torch.multiprocessing as mp
    ctxm = mp.get_context("spawn")
    q = ctxm.Queue()
    port = 29650 + n
    ps = [ctxm.Process(target=_sharded_clip_worker, args=(r, 2, port, n, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = [q.get(timeout=120) for _ in ps]
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(r[0] for r in res), res
    shapes = [r[1] for r in res if r[1] is not None]
    assert shapes == [(n, 6, 8, 3)]
    assert sorted(sum(r[2]) for r in res) == sorted([len(range(0, n, 2)), len(range(1, n, 2))])


def _sharded_fail_worker(rank, world, port, q):
    import os
    import numpy as np
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from vsdeoldify_amd import sharded

    def stub(t):
        if rank == 1:
            raise MemoryError("rank 1 out of memory")
        return t
    frames = torch.from_numpy(np.zeros((4, 6, 8, 3), np.uint8)) if rank == 0 else None
    try:
        sharded.colorize_clip_sharded(frames, stub, dist, rank, world, "cpu")
        q.put("no error")
    except RuntimeError as e:
        q.put("raised: " + str(e)[:60])
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_clip_runner_failure_on_one_rank_raises_everywhere_and_nobody_hangs():
    """a rank whose colorizer throws still joins the gather; the failure is then raised on EVERY rank (world_size 2, gloo)"""
    import torch.multiprocessing as mp
    ctxm = mp.get_context("spawn")
    q = ctxm.Queue()
    ps = [ctxm.Process(target=_sharded_fail_worker, args=(r, 2, 29671, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = [q.get(timeout=120) for _ in ps]
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(r.startswith("raised: colorize_clip_sharded") for r in res), res


def test_default_precision_is_the_contract_meeting_mode(monkeypatch):
    """round 6: explicit argument > HAVC_PRECISION > "precise" (vsdeoldify_amd/precision.py) -- the reference computes in fp32 (deoldify/filters.py:45-68), so a
    drop-in built without any switch must be the mode that meets north_star's CIEDE2000 < 1.0; "fast" is opt-in.  (tests/conftest.py selects "fast" for the
    suite through the process-wide switch.)"""
    from vsdeoldify_amd import precision as P
    monkeypatch.delenv("HAVC_PRECISION", raising=False)
    assert P.DEFAULT_PRECISION == "precise" and P.resolve() == "precise" and P.resolve(None) == "precise"
    assert P.resolve("fast") == "fast"
    monkeypatch.setenv("HAVC_PRECISION", "fast")
    assert P.resolve() == "fast" and P.resolve("precise") == "precise"
    for bad in ("double", "fp32"):
        with pytest.raises(ValueError):
            P.resolve(bad)
    monkeypatch.setenv("HAVC_PRECISION", "half")
    with pytest.raises(ValueError):
        P.resolve()


def test_the_built_library_matches_its_sources():
    """round 6: libhavc_mi355.so is git-ignored and travels prebuilt to the GPU box.  The Makefile writes the SHA-1 of every source it is built from
    into the library (csrc/build_stamp.h -> havc_build_stamp()); tools/build_stamp.py recomputes it from the tree.  A stale binary fails HERE, on the CPU, wherever the
    suite runs -- `python -c "import __graft_entry__ as g; g.build()"` rebuilds it."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import build_stamp
    from vsdeoldify_amd import _native as nat
    lib = nat.load()
    assert lib.havc_build_stamp().decode() == build_stamp.stamp(), "vsdeoldify_amd/lib/libhavc_mi355.so was not built from the sources in this tree: rebuild it (make -C vsdeoldify_amd/csrc)"
