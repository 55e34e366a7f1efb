"""DDColor (a13) — PARITY UNPINNED: vsddcolor is not under /root/reference and the reference holds no vectors at that boundary
(oracle/ddcolor.py header).  CPU: the ConvNeXt pieces of the oracle restatement against the independent implementation shipped in
the `transformers` package (pins the PUBLISHED encoder block, nothing DDColor-specific), plus shape / key bookkeeping.
GPU: the HIP path against the oracle on seeded synthetic weights (self-consistency)."""
import numpy as np
import pytest
import torch

from oracle import ddcolor as D
from vsdeoldify_amd.synth import ddcolor_state_dict_spec, synth_ddcolor_state_dict

SMALL = dict(depths=(1, 1, 2, 1), dec_layers=3)


def _t(sd):
    return {k: torch.as_tensor(v) for k, v in sd.items()}


def test_spec_counts():
    spec = ddcolor_state_dict_spec()
    n = sum(int(np.prod(s)) for s in spec.values())
    assert 220e6 < n < 230e6                                   # ConvNeXt-L encoder (196 M) + decoders
    assert sum(1 for k in spec if k.endswith(".dwconv.weight")) == 36
    assert spec["refine_net.0.0.weight_orig"] == (2, 103, 1, 1)


def test_convnext_pieces_match_transformers():
    tc = pytest.importorskip("transformers.models.convnext.modeling_convnext")
    from transformers import ConvNextConfig
    sd = _t(synth_ddcolor_state_dict(3, **SMALL))
    cfg = ConvNextConfig(num_channels=3, patch_size=4, hidden_sizes=[192, 384, 768, 1536], depths=[1, 1, 2, 1], layer_scale_init_value=1.0)
    torch.manual_seed(0)
    x = torch.randn(2, 3, 64, 64)
    # stem
    emb = tc.ConvNextEmbeddings(cfg).eval()
    with torch.no_grad():
        emb.patch_embeddings.weight.copy_(sd["encoder.arch.downsample_layers.0.0.weight"]); emb.patch_embeddings.bias.copy_(sd["encoder.arch.downsample_layers.0.0.bias"])
        emb.layernorm.weight.copy_(sd["encoder.arch.downsample_layers.0.1.weight"]); emb.layernorm.bias.copy_(sd["encoder.arch.downsample_layers.0.1.bias"])
        ref = emb(x)
        mine = D.layernorm_cf(torch.nn.functional.conv2d(x, sd["encoder.arch.downsample_layers.0.0.weight"], sd["encoder.arch.downsample_layers.0.0.bias"], stride=4),
                              sd["encoder.arch.downsample_layers.0.1.weight"], sd["encoder.arch.downsample_layers.0.1.bias"])
    assert torch.allclose(ref, mine, atol=2e-5), float((ref - mine).abs().max())
    # block
    p = "encoder.arch.stages.0.0"
    layer = tc.ConvNextLayer(cfg, dim=192, drop_path=0.0).eval()
    with torch.no_grad():
        layer.dwconv.weight.copy_(sd[p + ".dwconv.weight"]); layer.dwconv.bias.copy_(sd[p + ".dwconv.bias"])
        layer.layernorm.weight.copy_(sd[p + ".norm.weight"]); layer.layernorm.bias.copy_(sd[p + ".norm.bias"])
        layer.pwconv1.weight.copy_(sd[p + ".pwconv1.weight"]); layer.pwconv1.bias.copy_(sd[p + ".pwconv1.bias"])
        layer.pwconv2.weight.copy_(sd[p + ".pwconv2.weight"]); layer.pwconv2.bias.copy_(sd[p + ".pwconv2.bias"])
        layer.layer_scale_parameter.copy_(sd[p + ".gamma"])
        r2 = layer(ref)
        m2 = D.convnext_block(sd, p, mine)
    assert torch.allclose(r2, m2, atol=5e-5), float((r2 - m2).abs().max())


def test_oracle_attention_matches_torch_multihead_attention():
    """An independent pin for the colour decoder (its source is not in the reference tree): the oracle's multi-head attention
    against torch.nn.MultiheadAttention on random weights."""
    torch.manual_seed(0)
    E, H = 256, 8
    m = torch.nn.MultiheadAttention(E, H, batch_first=True).eval()
    sd = {"a.in_proj_weight": m.in_proj_weight.detach(), "a.in_proj_bias": m.in_proj_bias.detach(),
          "a.out_proj.weight": m.out_proj.weight.detach(), "a.out_proj.bias": m.out_proj.bias.detach()}
    with torch.no_grad():
        m.in_proj_bias.normal_(0, 0.1); m.out_proj.bias.normal_(0, 0.1)
        sd["a.in_proj_bias"], sd["a.out_proj.bias"] = m.in_proj_bias.detach(), m.out_proj.bias.detach()
        q, k, v = torch.randn(2, 100, E), torch.randn(2, 333, E), torch.randn(2, 333, E)
        want = m(q, k, v, need_weights=False)[0]
        got = D.mha(sd, "a", q, k, v, heads=H)
    assert torch.allclose(got, want, atol=2e-5, rtol=1e-5), float((got - want).abs().max())


def test_oracle_forward_shapes_and_determinism():
    sd = synth_ddcolor_state_dict(1, **SMALL)
    x = torch.rand(1, 3, 64, 64, generator=torch.Generator().manual_seed(0))
    with torch.no_grad():
        parts = D.forward(sd, x, return_parts=True, **SMALL)
    assert parts["f3"].shape == (1, 1536, 2, 2) and parts["out2"].shape == (1, 256, 16, 16)
    assert parts["out3"].shape == (1, 256, 64, 64) and parts["logits"].shape == (1, 100, 64, 64) and parts["ab"].shape == (1, 2, 64, 64)
    assert 1.0 < float(parts["ab"].std()) < 30.0               # the synthetic weights give a live, unsaturated colour signal
    frame = np.random.default_rng(0).integers(0, 256, (64, 64, 1), dtype=np.uint8).repeat(3, -1)
    out = D.colorize_frame(sd, frame, **SMALL)
    assert out.shape == (64, 64, 3) and out.dtype == np.uint8 and (out[..., 0] != out[..., 2]).mean() > 0.5


# ---- GPU: HIP path vs the oracle on seeded synthetic weights (self-consistency; parity unpinned) -------------------------
def _views(net_ops, names, name):
    i = names.index(name)
    return net_ops[i]


def _download(net, op, C, batch):
    """NHWC fp16 output view of a plan op -> float32 [B, C, H, W]"""
    pitch, coff, H, W = int(op["dst_cpitch"]), int(op["dst_coff"]), int(op["Ho"]), int(op["Wo"])
    raw = net.download(int(op["dst"]), (batch, H, W, pitch), np.float16)
    return raw[..., coff:coff + C].astype(np.float32).transpose(0, 3, 1, 2)


@pytest.mark.gpu
@pytest.mark.parametrize("fuse_tail", ["1", "0"])
def test_gpu_ddcolor_stages_match_oracle(ctx, fuse_tail, monkeypatch):
    """fuse_tail = 1 (default plan): einsum + refine conv folded into the epilogue of the last_shuf conv (out3 and the logits never exist:
    only ab and the frames are checked at that end); 0: the op-by-op tail with every intermediate checked."""
    from vsdeoldify_amd.ddcolor import DDColorRuntime
    from oracle import zhang as Z
    S, B = 64, 2
    monkeypatch.setenv("HAVC_DD_FUSE_TAIL", fuse_tail)
    sd = synth_ddcolor_state_dict(1, **SMALL)
    rt = DDColorRuntime(ctx, sd, **SMALL)
    try:
        r = np.random.default_rng(5)
        frames = r.integers(0, 256, (B, S, S, 1), dtype=np.uint8).repeat(3, -1)
        frames[1] = r.integers(0, 256, (S, S, 3), dtype=np.uint8)                # a non-gray frame: only its L may matter
        out = rt.colorize(frames)
        net = rt.net(S, B)
        gray = np.stack([Z.lab2rgb(np.concatenate([Z.rgb2lab(f)[..., :1], np.zeros((S, S, 2))], -1)) for f in frames]).astype(np.float32)
        with torch.no_grad():
            parts = D.forward(sd, torch.from_numpy(gray).permute(0, 3, 1, 2), return_parts=True, **SMALL)
        names, ops = net.names, net.plan_ops
        checks = [("encoder.arch.norm0", "f0", 192, 0.03), ("encoder.arch.norm3", "f3", 1536, 0.05), ("decoder.layers.0.conv", "out0", 512, 0.05),
                  ("decoder.layers.2.conv", "out2", 256, 0.05), ("decoder.last_shuf.shuf+blur", "out3", 256, 0.05),
                  ("refine_net.0.0", "ab", 2, 0.25)]
        for opname, key, C, tol in checks:
            if opname not in names:
                assert fuse_tail == "1" and key in ("out3", "f0")     # fused plans: norm0 lives inside the decoder's skip LayerNorm + BN + ReLU
                continue
            got = _download(net, _views(ops, names, opname), C, B)
            ref = parts[key].numpy()
            err = np.abs(got - ref)
            assert err.max() < tol * max(1.0, float(np.abs(ref).max())) and err.mean() < tol * 0.1 * max(1.0, float(np.abs(ref).std())), \
                (opname, float(err.max()), float(err.mean()), float(np.abs(ref).max()))
        # logits live in channels 8..107 of the coarse buffer
        if fuse_tail == "0":
            lop = _views(ops, names, "decoder.color_decoder.einsum")
            raw = net.download(int(lop["dst"]), (B, S, S, int(lop["dst_cpitch"])), np.float16)[..., 8:108].astype(np.float32).transpose(0, 3, 1, 2)
            ref = parts["logits"].numpy()
            assert np.abs(raw - ref).max() < 0.02 * np.abs(ref).max() + 0.1, float(np.abs(raw - ref).max())
        else:
            assert "decoder.last_shuf.conv+proj" in names and "decoder.color_decoder.einsum" not in names
        # end to end: frame in -> frame out against the oracle wrapper
        for f, o in zip(frames, out):
            want = D.colorize_frame(sd, f, **SMALL)
            d = np.abs(o.astype(int) - want.astype(int))
            assert (d <= 2).mean() > 0.99 and d.max() <= 12, (float((d <= 2).mean()), int(d.max()))
    finally:
        rt.close()


@pytest.mark.gpu
def test_gpu_ddcolor_full_depth_end_to_end(ctx):
    """all 36 ConvNeXt blocks and 9 decoder layers at 96x96 (three feature levels of 6x6 / 12x12 / 24x24 keys)"""
    from vsdeoldify_amd.ddcolor import DDColorRender
    sd = synth_ddcolor_state_dict(2)
    S = 96
    r = np.random.default_rng(9)
    frame = np.clip(r.normal(128, 50, (S, S, 1)), 0, 255).astype(np.uint8).repeat(3, -1)
    dd = DDColorRender(model=1, input_size=S, state_dict=sd)
    try:
        got = dd.colorize_frame(frame)
        want = D.colorize_frame(sd, frame)
        d = np.abs(got.astype(int) - want.astype(int))
        assert (d <= 2).mean() > 0.98 and d.max() <= 16, (float((d <= 2).mean()), int(d.max()))
        assert (got[..., 0] != got[..., 2]).mean() > 0.5                      # a coloured frame, not the gray input
        # a frame that is not input_size x input_size: squashed in, ab stretched back (odd render factors / 16:9 sources)
        wide = np.clip(r.normal(120, 60, (80, 144, 1)), 0, 255).astype(np.uint8).repeat(3, -1)
        got2 = dd.colorize_frame(wide)
        want2 = D.colorize_frame(sd, wide, input_size=S)
        d2 = np.abs(got2.astype(int) - want2.astype(int))
        assert got2.shape == wide.shape and (d2 <= 2).mean() > 0.98 and d2.max() <= 16, (float((d2 <= 2).mean()), int(d2.max()))
        with pytest.raises(ValueError):
            dd.colorize_frame(frame[..., 0])
    finally:
        dd.rt.close()
    with pytest.raises(ValueError):
        DDColorRender(model=2, input_size=S, state_dict=sd)


@pytest.mark.gpu
@pytest.mark.parametrize("C,H,W,B", [(192, 13, 10, 3), (384, 8, 8, 3), (768, 7, 9, 3), (1536, 5, 6, 3), (64, 3, 2, 3), (192, 64, 64, 17)])
def test_gpu_dwconv7_layernorm_fused_kernel(ctx, C, H, W, B):
    """ConvNeXt block head (depthwise 7x7 + bias, LayerNorm over channels) as ONE kernel against torch fp32 and against the two-kernel
    form; sizes that are not multiples of the 4 x 4 patch, every LDS weight format (fp32 up to 768 channels, fp16 above), and a launch
    whose 256 persistent blocks each walk more than one group of patches (17 frames of 64 x 64)."""
    from vsdeoldify_amd import _native as nat
    from vsdeoldify_amd.ddcolor_net import DDColorGenerator
    from vsdeoldify_amd.plan import PlanBuilder, WeightPack
    r = np.random.default_rng(C + H)
    Wt = (r.standard_normal((C, 1, 7, 7)) / 7).astype(np.float32)
    bias, gamma, beta = (r.standard_normal(C).astype(np.float32) * s + o for s, o in ((0.1, 0), (0.2, 1), (0.1, 0)))
    x = r.standard_normal((B, H, W, C)).astype(np.float16)
    outs = {}
    for fused in (True, False):
        pack, b = WeightPack(), PlanBuilder()
        xin, mid, y = b.tensor(H, W, C), b.tensor(H, W, C), b.tensor(H, W, C)
        w_off, b_off, g_off, be_off = (pack.add(a) for a in (DDColorGenerator._dw_pack(Wt, xin.span), bias, gamma, beta))
        if fused:
            b.dwconv7_ln("dwln", xin, y, w_off, b_off, xin.span, g_off, be_off, 1e-6)
        else:
            b.dwconv7("dw", xin, mid, w_off, b_off, xin.span)
            b.layernorm("ln", mid, y, g_off, be_off, 1e-6)
        ops, bufs = b.finish()
        wts = nat.Weights(ctx, pack.blob())
        net = nat.Net(ctx, wts, ops, bufs, 0, 0, 0, B)
        try:
            full = np.zeros((B, H, W, xin.cpitch), np.float16)
            full[..., :C] = x
            net.upload(xin.buf, full)
            net.run_ops(0, len(ops), B)
            outs[fused] = net.download(y.buf, (B, H, W, y.cpitch), np.float16)[..., :C].astype(np.float32)
        finally:
            net.close(); wts.close()
    xt = torch.from_numpy(x.astype(np.float32)).permute(0, 3, 1, 2)
    wq = torch.from_numpy(Wt.astype(np.float16).astype(np.float32))
    conv = torch.nn.functional.conv2d(xt, wq, torch.from_numpy(bias), padding=3, groups=C)
    ref = torch.nn.functional.layer_norm(conv.permute(0, 2, 3, 1), (C,), torch.from_numpy(gamma), torch.from_numpy(beta), 1e-6).numpy()
    err = np.abs(outs[True] - ref)
    assert err.max() < 6e-3 and err.mean() < 6e-4, (float(err.max()), float(err.mean()))           # fp16 output rounding: |y| < 8
    err2 = np.abs(outs[False] - ref)
    assert err.mean() <= err2.mean() * 1.05                       # the fused form norms the fp32 conv result: never less accurate


@pytest.mark.gpu
@pytest.mark.parametrize("S", [32, 96])
def test_gpu_ddcolor_frames_do_not_depend_on_the_batch(ctx, S):
    """A frame colours identically alone and inside a batch, also where a 256-pixel GEMM tile of the folded tail spans several frames
    (S = 32: 64 low-res pixels per frame, four frames per tile; S = 96: 576 pixels, tiles straddle frame boundaries) and where the fused
    block head sees 1 x 1 and 3 x 3 feature maps."""
    from vsdeoldify_amd.ddcolor import DDColorRuntime
    sd = synth_ddcolor_state_dict(4, **SMALL)
    rt = DDColorRuntime(ctx, sd, **SMALL)
    try:
        r = np.random.default_rng(S)
        frames = r.integers(0, 256, (5, S, S, 1), dtype=np.uint8).repeat(3, -1)
        together = rt.colorize(frames, max_batch=5)
        alone = np.stack([rt.colorize(frames[i:i + 1], max_batch=1)[0] for i in range(5)])
        assert np.array_equal(together, alone)
        want = np.stack([D.colorize_frame(sd, f, **SMALL) for f in frames[:2]])
        d = np.abs(together[:2].astype(int) - want.astype(int))
        assert (d <= 2).mean() > 0.98 and d.max() <= 16, (float((d <= 2).mean()), int(d.max()))
    finally:
        rt.close()


@pytest.mark.gpu
def test_gpu_ddcolor_coalesced_per_frame_calls(ctx, monkeypatch):
    """colorize_frame from several threads through one DDColorRender(coalesce=N): the calls are merged into batches and every caller
    gets the bytes of a call of its own (a DDColor pass costs 7.7 ms alone and 1.2 ms per frame in a batch of 16)."""
    import threading
    from vsdeoldify_amd.ddcolor import DDColorRender
    monkeypatch.setenv("HAVC_COALESCE_WAIT_US", "50000")     # a leader waits up to 50 ms for its batch to fill: coalescing is certain
    sd = synth_ddcolor_state_dict(5, **SMALL)
    T, K, S = 5, 3, 64
    start = threading.Barrier(T)
    r = np.random.default_rng(3)
    frames = [r.integers(0, 256, (S, S, 1), dtype=np.uint8).repeat(3, -1) for _ in range(T * K)]
    ref = DDColorRender(model=1, input_size=S, state_dict=sd, **SMALL)
    shared = DDColorRender(model=1, input_size=S, state_dict=sd, coalesce=T, **SMALL)
    try:
        want = [ref.colorize_frame(f) for f in frames]
        got, errs = {}, []

        def run(t):
            try:
                start.wait()
                for k in range(K):
                    got[t * K + k] = shared.colorize_frame(frames[t * K + k])
            except Exception as e:                                                 # pragma: no cover
                errs.append(e)
        ts = [threading.Thread(target=run, args=(t,)) for t in range(T)]
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        assert not errs, errs
        assert all(np.array_equal(got[i], want[i]) for i in range(T * K))
        calls, batches = shared._batchers[(S, S)].stats()
        assert calls == T * K and batches < calls, (calls, batches)
    finally:
        for b in shared._batchers.values():
            b.close()
        ref.rt.close(); shared.rt.close()


def test_checkpoint_file_layouts_load(tmp_path):
    """ddcolor_modelscope.pth / ddcolor_artistic.pth as published: {'params': state_dict} (or a bare state dict) of torch tensors under
    the public key names -- what DDColorRender(model_dir=...) reads (the real files cannot be fetched here)."""
    from vsdeoldify_amd.ddcolor import DDColorRender, load_state_dict
    sd = synth_ddcolor_state_dict(6, **SMALL)
    tsd = {k: torch.tensor(v) for k, v in sd.items()}
    for name, obj in ((DDColorRender.MODEL_FILES[0], {"params": tsd}), (DDColorRender.MODEL_FILES[1], tsd)):
        torch.save(obj, tmp_path / name)
        got = load_state_dict(str(tmp_path / name))
        assert set(got) == set(sd) and all(np.array_equal(got[k], sd[k]) for k in sd)
