"""ColorMNet memory kernels.  CPU: the oracle against vectors produced by EXECUTING the reference
(tools/gen_golden_colormnet.py).  GPU: the HIP kernels against those vectors and against the oracle on other shapes."""
import os

import numpy as np
import pytest
import torch

from oracle import colormnet as O
from tests.conftest import GOLDEN


def T(a):
    return torch.from_numpy(np.asarray(a))


@pytest.fixture(scope="module")
def gm():
    return np.load(os.path.join(GOLDEN, "colormnet_memory.npz"))


@pytest.fixture(scope="module")
def gl():
    return np.load(os.path.join(GOLDEN, "colormnet_local.npz"))


CASES = [("full", True, True), ("nosel", True, False), ("plain", False, False)]


@pytest.mark.parametrize("tag,has_ms,has_qe", CASES)
def test_oracle_memory_read_matches_reference(gm, tag, has_ms, has_qe):
    ms, qe = (T(gm["ms"]) if has_ms else None), (T(gm["qe"]) if has_qe else None)
    sim = O.get_similarity(T(gm["mk"]), ms, T(gm["qk"]), qe)
    assert torch.equal(sim, T(gm[f"sim_{tag}"]))
    aff = O.do_softmax(sim, int(gm["top_k"]))
    assert torch.equal(aff, T(gm[f"aff_{tag}"])) and int((aff > 0).sum(1).max()) == 30
    assert torch.allclose(O.readout(aff, T(gm["mv"])), T(gm[f"read_{tag}"]), atol=1e-6)
    assert torch.allclose(O.do_softmax(O.get_similarity(T(gm["mk"]), T(gm["ms"]), T(gm["qk"]), T(gm["qe"]))), T(gm["aff_full_notopk"]), atol=1e-7)


def test_oracle_local_attention_matches_reference(gl):
    assert torch.equal(O.local_correlation(T(gl["q"]) / 8.0, T(gl["k"])), T(gl["corr_scaled"]))
    agg, attn = O.local_attention(T(gl["q"]), T(gl["k"]), T(gl["v"]), T(gl["rel_w"]), T(gl["rel_b"]))
    assert torch.allclose(attn, T(gl["attn"]), atol=1e-7) and torch.allclose(agg, T(gl["agg"]), atol=2e-6)
    # every window position outside the image carries no weight; rows sum to one
    assert torch.allclose(attn.sum(2), torch.ones_like(attn.sum(2)), atol=1e-5)


# ---------------------------------------------------------------------------------------------------------------- GPU
@pytest.mark.gpu
@pytest.mark.parametrize("tag,has_ms,has_qe", CASES)
def test_gpu_memory_read_matches_reference_vectors(ctx, gm, tag, has_ms, has_qe):
    from vsdeoldify_amd import colormnet as M
    ms, qe = (gm["ms"] if has_ms else None), (gm["qe"] if has_qe else None)
    sim = M.get_similarity(gm["mk"], ms, gm["qk"], qe)
    assert np.allclose(sim, gm[f"sim_{tag}"], rtol=2e-5, atol=2e-5), float(np.abs(sim - gm[f"sim_{tag}"]).max())
    got = M.match_memory_readout(gm["mk"], ms, gm["qk"], qe, gm["mv"], int(gm["top_k"]))
    ref = gm[f"read_{tag}"]
    assert got.shape == ref.shape and np.allclose(got, ref, rtol=1e-4, atol=1e-4), float(np.abs(got - ref).max())


@pytest.mark.gpu
@pytest.mark.parametrize("B,CK,CV,N,HW,k", [(1, 64, 512, 2000, 600, 30), (2, 32, 40, 77, 130, 30), (1, 64, 8, 20, 50, 30), (1, 16, 16, 300, 64, 5),
                                            (1, 64, 16, 6100, 97, 30), (2, 64, 8, 12345, 70, 30), (1, 64, 8, 17000, 40, 30),      # register / LDS radix select, two-level beyond 16 384
                                            (1, 64, 8, 16384, 40, 30), (1, 64, 8, 16381, 33, 30),     # the LDS select at its limit: N * 4 bytes of dynamic LDS + 16 static (> 64 KiB: opt-in)
                                            (1, 64, 8, 20000, 24, 64)])                                # two-level merge at K = 64, 128 slices: exactly 64 KiB of dynamic LDS
def test_gpu_memory_read_matches_oracle(ctx, B, CK, CV, N, HW, k):
    """bigger / ragged shapes, k > N, several batches; near-ties in the top-k may pick another element of (almost) equal weight:
    the readout is compared, not the index set"""
    from vsdeoldify_amd import colormnet as M
    g = torch.Generator().manual_seed(N + HW)
    mk, qk = torch.randn(B, CK, N, generator=g) * 0.4, torch.randn(B, CK, HW, generator=g) * 0.4
    ms, qe, mv = torch.rand(B, N, generator=g) + 1, torch.rand(B, CK, HW, generator=g), torch.randn(B, CV, N, generator=g)
    ref = O.memory_read(mk, ms, qk, qe, mv, min(k, N))
    got = M.match_memory_readout(mk, ms, qk, qe, mv, k)
    assert isinstance(got, torch.Tensor) and got.shape == ref.shape
    assert torch.allclose(got, ref, rtol=2e-4, atol=2e-4), float((got - ref).abs().max())


@pytest.mark.gpu
def test_gpu_local_attention_matches_reference_vectors(ctx, gl):
    from vsdeoldify_amd import colormnet as M
    corr = M.local_correlation(gl["q"], gl["k"], 7, 1, 1.0 / 8.0)
    assert np.allclose(corr, gl["corr_scaled"], rtol=1e-5, atol=1e-5), float(np.abs(corr - gl["corr_scaled"]).max())
    agg, attn = M.local_attention(gl["q"], gl["k"], gl["v"], gl["rel_w"], gl["rel_b"])
    assert np.allclose(attn, gl["attn"], rtol=1e-4, atol=1e-6), float(np.abs(attn - gl["attn"]).max())
    assert np.allclose(agg, gl["agg"], rtol=1e-4, atol=1e-5), float(np.abs(agg - gl["agg"]).max())


@pytest.mark.gpu
def test_gpu_memory_read_with_ties_at_the_threshold(ctx):
    """a memory that holds the same frames twice (a static shot): every similarity occurs twice, the k-th largest value is tied.  The selection
    takes the values above it and then, in ascending memory index, as many of the tied ones as still fit (wave-per-query radix select); which of
    two identical elements is taken does not change the readout.  k odd: a tie is split.  Also: the same call twice gives the same bytes."""
    from vsdeoldify_amd import colormnet as M
    g = torch.Generator().manual_seed(77)
    B, CK, CV, n, HW, k = 1, 64, 24, 700, 130, 31
    mk1, mv1, ms1 = torch.randn(B, CK, n, generator=g) * 0.4, torch.randn(B, CV, n, generator=g), torch.rand(B, n, generator=g) + 1
    mk, mv, ms = torch.cat([mk1, mk1], -1), torch.cat([mv1, mv1], -1), torch.cat([ms1, ms1], -1)
    qk, qe = torch.randn(B, CK, HW, generator=g) * 0.4, torch.rand(B, CK, HW, generator=g)
    ref = O.memory_read(mk, ms, qk, qe, mv, k)
    got = M.match_memory_readout(mk, ms, qk, qe, mv, k)
    assert torch.allclose(got, ref, rtol=2e-4, atol=2e-4), float((got - ref).abs().max())
    assert torch.equal(got, M.match_memory_readout(mk, ms, qk, qe, mv, k))


@pytest.mark.gpu
@pytest.mark.parametrize("n,C,Cv,h,w,R,dil", [(1, 64, 96, 23, 37, 7, 1), (2, 64, 32, 9, 8, 7, 1), (1, 32, 40, 17, 5, 3, 2), (1, 64, 33, 30, 30, 7, 1)])
def test_gpu_local_attention_matches_oracle(ctx, n, C, Cv, h, w, R, dil):
    """sizes that are not multiples of the 8 x 8 tile, a smaller window, dilation 2, value widths off the 32-channel chunk"""
    from vsdeoldify_amd import colormnet as M
    g = torch.Generator().manual_seed(h * w + Cv)
    q, k, v = torch.randn(n, C, h, w, generator=g), torch.randn(n, C, h, w, generator=g), torch.randn(n, Cv, h, w, generator=g)
    ws = 2 * R + 1
    rel_w, rel_b = torch.randn(ws * ws, C, generator=g) * 0.1, torch.randn(ws * ws, generator=g) * 0.1
    ref_agg, ref_attn = O.local_attention(q, k, v, rel_w, rel_b, R, dil)
    agg, attn = M.local_attention(q, k, v, rel_w, rel_b, R, dil)
    assert torch.allclose(attn, ref_attn, rtol=1e-4, atol=1e-6), float((attn - ref_attn).abs().max())
    assert torch.allclose(agg, ref_agg, rtol=1e-4, atol=1e-5), float((agg - ref_agg).abs().max())
    ref_c = O.local_correlation(q, k, R, dil)
    assert torch.allclose(M.local_correlation(q, k, R, dil), ref_c, rtol=1e-5, atol=1e-5)


def _local_fixture():
    return np.load(os.path.join(os.path.dirname(__file__), "golden", "colormnet_local.npz"))


def test_oracle_short_term_tail_matches_the_executed_reference_module():
    """oracle.colormnet_net.short_term_attn (local attention + DWConv2d + Linear, attention.py:783-860, basic.py:75-94) against the output the
    reference's own LocalGatedPropagation module produced (tests/golden/colormnet_local.npz)"""
    import torch
    from oracle import colormnet_net as ON
    g = _local_fixture()
    sd = {"short_term_attn.relative_emb_k.weight": torch.from_numpy(g["rel_w"]).reshape(225, 64, 1, 1), "short_term_attn.relative_emb_k.bias": torch.from_numpy(g["rel_b"]),
          "short_term_attn.dw_conv.conv.weight": torch.from_numpy(g["dw_w"]), "short_term_attn.projection.weight": torch.from_numpy(g["proj_w"]),
          "short_term_attn.projection.bias": torch.from_numpy(g["proj_b"])}
    q, k, v = (torch.from_numpy(g[n]) for n in ("q", "k", "v"))
    out, attn = ON.short_term_attn(sd, q, k, v, (q.shape[2], q.shape[3]))
    assert np.abs(out.numpy() - g["out"]).max() < 2e-5 and np.abs(attn.numpy() - g["attn"]).max() < 1e-6


@pytest.mark.gpu
def test_gpu_short_term_tail_ops_match_the_executed_reference_module(ctx):
    """the plan ops behind the local attention (planar-in of agg_value, depthwise 5x5, Linear as a 1x1 conv, planar-out) on the reference
    module's recorded agg_value -> its recorded output (fp16 activations: relative tolerance)"""
    from tests.gpu_util import run_plan
    from vsdeoldify_amd.plan import PlanBuilder, WeightPack, pack_conv
    g = _local_fixture()
    agg, want = g["agg"], g["out"]                                 # [h*w, 1, C]
    hw, _, C = agg.shape
    h, w = g["q"].shape[2:]
    pack, b = WeightPack(), PlanBuilder()
    src = b.buf(C * hw, 4)
    a_in, a_dw, a_out = b.tensor(h, w, C), b.tensor(h, w, C), b.tensor(h, w, C)
    b.planar_in("in", src, C, a_in, pixel_major=True)
    wp = np.zeros((25, a_in.span), np.float16)
    wp[:, :C] = g["dw_w"].reshape(C, 25).T
    b.dwconv("dw", a_in, a_dw, pack.add(wp), -1, a_in.span, 5)
    pc = pack_conv(pack, g["proj_w"][:, :, None, None].astype(np.float32), a_dw.cmap, a_dw.span, bias=g["proj_b"].astype(np.float32))
    b.conv("proj", pc, a_dw, a_out)
    dst = b.buf(C * hw, 4)
    b.planar_out("out", a_out, 0, C, dst, 0)
    got = run_plan(ctx, pack, b, {src: agg.reshape(hw, C).astype(np.float32)}, {dst: ((C, hw), np.float32)}, 1)[dst]
    ref = want[:, 0, :].T
    assert np.abs(got - ref).max() < 6e-3 * max(1.0, float(np.abs(ref).max())), float(np.abs(got - ref).max())
