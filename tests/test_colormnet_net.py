"""ColorMNet network + frame wrapper (BASELINE configs[4]).

CPU (-m "not gpu"): the oracle (oracle/colormnet_net.py, oracle/dinov2.py) against vectors recorded by EXECUTING the reference's
ColorMNet modules and its ColorMNetRender in the build container (tools/gen_golden_colormnet_net.py), the state-dict layout against the
reference's module tree, the DINOv2 restatement against the independent implementation in `transformers`.
GPU (-m gpu): the HIP network (vsdeoldify_amd/colormnet_net.py) against the oracle and the same vectors."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import colormnet_net as O
from oracle import dinov2
from vsdeoldify_amd.synth import colormnet_state_dict_spec, synth_colormnet_state_dict

G = os.path.join(os.path.dirname(__file__), "golden")
MOD = np.load(os.path.join(G, "colormnet_net_modules.npz"))
REN = np.load(os.path.join(G, "colormnet_net_render.npz"))
SEED = int(MOD["seed"])


def tsd(seed=SEED):
    return {k: torch.from_numpy(np.asarray(v)) for k, v in synth_colormnet_state_dict(seed).items()}


def check(name, t, k, tol):
    """fixture = every k-th element + (sum, abs-sum)"""
    a = (t.detach().cpu().numpy() if torch.is_tensor(t) else np.asarray(t)).astype(np.float32)
    assert list(a.shape) == MOD[name + "_shape"].tolist(), (name, a.shape)
    want = MOD[name]
    got = a.reshape(-1)[::k]
    scale = max(1.0, float(np.abs(want).max()))
    err = float(np.abs(got - want).max()) / scale
    assert err < tol, (name, err)
    s = MOD[name + "_sum"]
    assert abs(float(a.astype(np.float64).sum()) - s[0]) < tol * a.size and abs(float(np.abs(a).astype(np.float64).sum()) - s[1]) < tol * a.size, name
    return err


def stride_of(name):
    n = int(np.prod(MOD[name + "_shape"]))
    return (n + len(MOD[name]) - 1) // len(MOD[name])


def test_state_dict_layout_is_the_reference_module_tree():
    ref = json.load(open(os.path.join(G, "spec_colormnet.json")))["keys"]
    assert [(k, list(v)) for k, v in colormnet_state_dict_spec().items()] == [(k, list(v)) for k, v in ref]
    sd = synth_colormnet_state_dict(0)
    assert list(sd) == [k for k, _ in ref] and all(tuple(np.shape(sd[k])) == tuple(s) for k, s in ref)


def test_dinov2_matches_transformers():
    """oracle/dinov2.py against transformers.Dinov2Model on seeded weights: every hidden state of a 3-block ViT-S/14 at the pretraining
    grid (no position interpolation: the two libraries need not agree on that kludge) and the final norm"""
    from transformers import Dinov2Config, Dinov2Model
    depth, grid = 3, 4
    cfg = Dinov2Config(hidden_size=384, num_hidden_layers=depth, num_attention_heads=6, image_size=14 * grid, patch_size=14, layer_norm_eps=1e-6,
                       hidden_act="gelu", mlp_ratio=4)
    m = Dinov2Model(cfg).eval()
    g = torch.Generator().manual_seed(5)
    with torch.no_grad():
        for p in m.parameters():
            p.copy_(torch.randn(p.shape, generator=g) * (0.05 if p.dim() > 1 else 0.3) + (1.0 if p.dim() == 1 and p.shape[0] == 384 and False else 0.0))
    hf = m.state_dict()
    sd = {"cls_token": hf["embeddings.cls_token"], "pos_embed": hf["embeddings.position_embeddings"], "mask_token": hf["embeddings.mask_token"],
          "patch_embed.proj.weight": hf["embeddings.patch_embeddings.projection.weight"], "patch_embed.proj.bias": hf["embeddings.patch_embeddings.projection.bias"],
          "norm.weight": hf["layernorm.weight"], "norm.bias": hf["layernorm.bias"]}
    for i in range(depth):
        a, b = f"encoder.layer.{i}", f"blocks.{i}"
        for n_ in ("norm1", "norm2"):
            sd[f"{b}.{n_}.weight"], sd[f"{b}.{n_}.bias"] = hf[f"{a}.{n_}.weight"], hf[f"{a}.{n_}.bias"]
        att = f"{a}.attention.attention"
        sd[b + ".attn.qkv.weight"] = torch.cat([hf[f"{att}.{t}.weight"] for t in ("query", "key", "value")], 0)
        sd[b + ".attn.qkv.bias"] = torch.cat([hf[f"{att}.{t}.bias"] for t in ("query", "key", "value")], 0)
        sd[b + ".attn.proj.weight"], sd[b + ".attn.proj.bias"] = hf[f"{a}.attention.output.dense.weight"], hf[f"{a}.attention.output.dense.bias"]
        sd[b + ".ls1.gamma"], sd[b + ".ls2.gamma"] = hf[f"{a}.layer_scale1.lambda1"], hf[f"{a}.layer_scale2.lambda1"]
        for f_ in ("fc1", "fc2"):
            sd[f"{b}.mlp.{f_}.weight"], sd[f"{b}.mlp.{f_}.bias"] = hf[f"{a}.mlp.{f_}.weight"], hf[f"{a}.mlp.{f_}.bias"]
    x = torch.randn(1, 3, 14 * grid, 14 * grid, generator=g)
    with torch.no_grad():
        want = m(pixel_values=x, output_hidden_states=True)
        t = dinov2.tokens(sd, x)
        assert (t - want.hidden_states[0]).abs().max() < 1e-5
        for i in range(depth):
            t = dinov2.block(sd, f"blocks.{i}", t, 6)
            assert (t - want.hidden_states[i + 1]).abs().max() < 2e-4, i
        outs = dinov2.get_intermediate_layers(sd, x, [depth - 1])
        last = want.last_hidden_state[:, 1:].reshape(1, grid, grid, 384).permute(0, 3, 1, 2)
        assert (outs[0] - last).abs().max() < 2e-4


def test_dinov2_position_interpolation_is_the_hub_kludge():
    """bicubic with scale factors (h0 + 0.1) / M, (w0 + 0.1) / M: the output grid is exactly h0 x w0 and the class position is untouched"""
    pos = torch.randn(1, 1 + 37 * 37, 8, generator=torch.Generator().manual_seed(1))
    cls, patch = dinov2.interpolate_pos_encoding(pos, 8, 16)
    assert patch.shape == (128, 8) and torch.equal(cls, pos[:, 0])
    cls, same = dinov2.interpolate_pos_encoding(pos, 37, 37)
    assert torch.equal(same, pos[0, 1:])


def _frame():
    return torch.from_numpy(MOD["frame"]).view(1, 1, *MOD["frame"].shape).repeat(1, 3, 1, 1)


def test_oracle_network_matches_the_executed_reference():
    sd = tsd()
    with torch.no_grad():
        frame = _frame()
        key, shr, sel, f16, f8, f4 = O.encode_key(sd, frame)
        for n_, t in (("key", key), ("shrinkage", shr), ("selection", sel), ("f16", f16), ("f8", f8), ("f4", f4)):
            check(n_, t, stride_of(n_), 2e-4)
        check("dino16", O.segmentor(sd, "key_encoder.network2", frame), stride_of("dino16"), 2e-4)
        masks = torch.from_numpy(MOD["masks"]).unsqueeze(0)
        h0 = torch.from_numpy(MOD["hidden0"]).unsqueeze(0)
        for deep in (1, 0):
            val, h1 = O.encode_value(sd, frame, f16, h0, masks, is_deep_update=bool(deep))
            check(f"value_deep{deep}", val, stride_of(f"value_deep{deep}"), 2e-4)
            check(f"hidden_deep{deep}", h1, 1, 2e-4)
        readout = torch.from_numpy(MOD["readout"].astype(np.float32)).unsqueeze(0)
        for h_out in (1, 0):
            hid, prob = O.segment(sd, (f16, f8, f4), readout, h0, h_out=bool(h_out))
            check(f"prob_hout{h_out}", prob, stride_of(f"prob_hout{h_out}"), 2e-4)
            if h_out:
                check("hidden_seg", hid, 1, 2e-4)
            else:
                assert hid is None
        short, _ = O.short_term_attn(sd, key, torch.from_numpy(MOD["k2"]).unsqueeze(0), val.flatten(1, 2), key.shape[-2:])
        check("short", short, stride_of("short"), 2e-4)


SCENARIOS = {"keep": (dict(reset_on_ref_update=False, max_memory_frames=0), False), "vivid": (dict(reset_on_ref_update=True, max_memory_frames=0), False),
             "propagate": (dict(reset_on_ref_update=False, max_memory_frames=0), True), "capped": (dict(reset_on_ref_update=False, max_memory_frames=3), False)}


def run_render_scenario(name, network, memory_backend=None):
    """drive the drop-in ColorMNetRender exactly as tools/gen_golden_colormnet_net.py drove the reference's class; returns the 9 coloured frames"""
    from PIL import Image
    from vsdeoldify_amd.colormnet_render import ColorMNetRender
    kw, propagate = SCENARIOS[name]
    frames, refs = REN["frames"], REN["refs"]
    rnd = ColorMNetRender(image_size=-1, vid_length=len(frames), enable_resize=False, encode_mode=1, propagate=propagate, network=network,
                          memory_backend=memory_backend, **kw)
    rnd.set_config("mem_every", int(REN["mem_every"]))
    outs = []
    for t, fr in enumerate(frames):
        rnd.set_ref_frame(Image.fromarray(refs[0]) if t == 0 else (Image.fromarray(refs[1]) if t == 4 else None), propagate)
        outs.append(np.asarray(rnd.colorize_frame(ti=t, frame_i=Image.fromarray(np.stack([fr] * 3, -1)))))
    return np.stack(outs)


def want_of(name):
    return REN["outs" if name == "keep" else "outs_" + name]


def test_oracle_frame_loop_matches_the_reference_render_class():
    """the reference's own ColorMNetRender.colorize_frame over 9 frames (exemplars with frames 0 and 4) vs oracle network + drop-in
    InferenceCore / MemoryManager: u8 frames, identical up to float noise at the rounding boundary"""
    from oracle import colormnet_clip
    frames, refs, want = REN["frames"], REN["refs"], REN["outs"]
    got = np.stack(colormnet_clip.colorize_clip(tsd(), [np.stack([f] * 3, -1) for f in frames], {0: refs[0], 4: refs[1]}, {"mem_every": int(REN["mem_every"])}))
    d = np.abs(got.astype(np.int32) - want.astype(np.int32))
    assert got.shape == want.shape and d.max() <= 1 and (d > 0).mean() < 2e-3, (int(d.max()), float((d > 0).mean()))


@pytest.mark.parametrize("name", list(SCENARIOS))
def test_render_state_machine_matches_the_reference_class_on_cpu(name):
    """The DROP-IN ColorMNetRender (vsdeoldify_amd/colormnet_render.py: reference counters, the memory reset when a new reference arrives
    (render_vivid) or max_memory_frames is reached, FirstFrameIsNotExemplar / step vs step_AnyExemplar) with the oracle network plugged in,
    against the frames the reference's own class produced in the same four scenarios: <= 1 LSB."""
    from oracle import colormnet_clip
    got = run_render_scenario(name, colormnet_clip.OracleNetwork(tsd()), colormnet_clip.OracleBackend())
    want = want_of(name)
    d = np.abs(got.astype(np.int32) - want.astype(np.int32))
    assert got.shape == want.shape and d.max() <= 1 and (d > 0).mean() < 2e-3, (name, int(d.max()), float((d > 0).mean()))


class _LookaheadOracleNetwork:
    """the oracle network + the look-ahead protocol of ColorMNetNetwork (prefetch_keys / expect_prefetched / drop_prefetched), with the keys
    computed by the oracle's own encode_key: exercises the FIFO bookkeeping of the drop-in render on the CPU"""

    def __new__(cls, sd):
        import collections
        from oracle import colormnet_clip

        class Net(colormnet_clip.OracleNetwork):
            def __init__(self, sd_):
                super().__init__(sd_)
                self._armed, self.served, self.batches = None, 0, []

            def prefetch_keys(self, frames, max_batch=None):
                self.batches.append(len(frames))
                return [super(Net, self).encode_key(f.unsqueeze(0)) for f in frames]

            def expect_prefetched(self, entry):
                self._armed = entry

            def encode_key(self, frame, need_ek=True, need_sk=True):
                if self._armed is not None:
                    (key, shr, sel, f16, f8, f4), self._armed = self._armed, None
                    self.served += 1
                    return key, (shr if need_sk else None), (sel if need_ek else None), f16, f8, f4
                return super().encode_key(frame, need_ek=need_ek, need_sk=need_sk)
        return Net(sd)


@pytest.mark.parametrize("name", ["keep", "vivid", "capped"])
def test_render_lookahead_bookkeeping_on_cpu(name):
    """colorize_batch_frames with a look-ahead window of 4 over the 9-frame fixture clip: every frame's key comes out of the prefetch FIFO
    exactly once (exemplar images are encoded on the spot), windows are 4 + 4 + 1 frames, nothing is left behind, and the frames equal the
    reference's recorded ones (<= 1 LSB) — memory resets inside a window ("vivid" at frame 4, "capped" every 3 frames) included."""
    from PIL import Image
    from oracle import colormnet_clip
    from vsdeoldify_amd.colormnet_render import ColorMNetRender
    kw, propagate = SCENARIOS[name]
    frames, refs, want = REN["frames"], REN["refs"], want_of(name)
    network = _LookaheadOracleNetwork(tsd())
    rnd = ColorMNetRender(image_size=-1, vid_length=len(frames), enable_resize=False, encode_mode=1, propagate=propagate, network=network,
                          memory_backend=colormnet_clip.OracleBackend(), lookahead=4, **kw)
    rnd.set_config("mem_every", int(REN["mem_every"]))
    imgs = [Image.fromarray(np.stack([fr] * 3, -1)) for fr in frames]
    ref_list = [Image.fromarray(refs[0]) if t == 0 else (Image.fromarray(refs[1]) if t == 4 else None) for t in range(len(frames))]
    got = np.stack([np.asarray(o) for o in rnd.colorize_batch_frames(imgs, ref_list, propagate)])
    assert network.batches == [4, 4] and network.served == 8 and network._armed is None and not rnd._ahead      # the last window is one frame: no prefetch
    d = np.abs(got.astype(np.int32) - want.astype(np.int32))
    assert d.max() <= 1 and (d > 0).mean() < 2e-3, (name, int(d.max()), float((d > 0).mean()))


def test_render_lookahead_forgets_frames_the_caller_skips():
    """prefetch() announces frames; a caller that then passes OTHER frames gets them computed on the spot and the stale look-ahead is dropped
    on both sides (render FIFO and network FIFO stay in step); frames before the first reference pass through and consume their entry."""
    from PIL import Image
    from oracle import colormnet_clip
    from vsdeoldify_amd.colormnet_render import ColorMNetRender
    frames, refs = REN["frames"], REN["refs"]
    network = _LookaheadOracleNetwork(tsd())
    rnd = ColorMNetRender(image_size=-1, vid_length=6, network=network, memory_backend=colormnet_clip.OracleBackend(), lookahead=4,
                          reset_on_ref_update=False)
    imgs = [Image.fromarray(np.stack([fr] * 3, -1)) for fr in frames[:6]]
    rnd.prefetch(imgs[:3])
    assert len(rnd._ahead) == 3
    rnd.set_ref_frame(None)
    assert rnd.colorize_frame(0, imgs[0]) is imgs[0]                     # no reference yet: passthrough, its entry is consumed
    assert len(rnd._ahead) == 2 and network._armed is None
    rnd.set_ref_frame(Image.fromarray(refs[0]))
    out = rnd.colorize_frame(1, imgs[4])                                 # not the announced frame: look-ahead dropped, frame computed
    assert not rnd._ahead and network.served == 0 and np.asarray(out).shape == np.asarray(imgs[4]).shape
    rnd.prefetch(imgs[2:4])
    rnd.set_ref_frame(None)
    rnd.colorize_frame(2, imgs[2])
    rnd.colorize_frame(3, imgs[3])
    assert network.served == 2 and network._armed is None and not rnd._ahead


# =====================================================================================================================================
# GPU: the new plan ops one by one (tests/gpu_util.run_plan: upload fp16 NHWC, run, download) against plain torch fp32 on the CPU
# =====================================================================================================================================
import torch.nn.functional as F  # noqa: E402

from tests.gpu_util import nhwc_pad, run_plan  # noqa: E402


def _rand(*shape, seed=0, scale=1.0):
    return (torch.randn(*shape, generator=torch.Generator().manual_seed(seed)) * scale).numpy().astype(np.float32)


def _f16(a):
    return a.astype(np.float16).astype(np.float32)


def _nchw(out_nhwc, C):
    return out_nhwc[..., :C].astype(np.float32).transpose(0, 3, 1, 2)


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["bilinear_up2", "bilinear_frac", "area4", "copy_relu_dual"])
def test_ew_op(ctx, mode):
    from vsdeoldify_amd.plan import PlanBuilder, WeightPack
    B, C, H, W = 2, 24, 7, 14
    x = _f16(_rand(B, C, H, W, seed=1))
    res1 = _f16(_rand(1, C, 2 * H, 2 * W, seed=2))
    b = PlanBuilder()
    xv = b.tensor(H, W, C)
    ups = {xv.buf: nhwc_pad(x, xv.cpitch)}
    if mode == "bilinear_up2":                                   # upsample_groups(g) + skip (broadcast over the objects), with the rectified copy
        yv, y2, rv = b.tensor(2 * H, 2 * W, C), b.tensor(2 * H, 2 * W, C), b.tensor(2 * H, 2 * W, C)
        ups[rv.buf] = np.concatenate([nhwc_pad(res1, rv.cpitch), np.zeros_like(nhwc_pad(res1, rv.cpitch))], 0)
        b.ew("t", xv, yv, mode=1, ratio=(0.5, 0.5), res=rv, res_bcast=True, dual=y2)
        want = F.interpolate(torch.from_numpy(x), scale_factor=2, mode="bilinear", align_corners=False).numpy() + res1
        out = run_plan(ctx, WeightPack(), b, ups, {yv.buf: ((B, 2 * H, 2 * W, yv.cpitch), np.float16), y2.buf: ((B, 2 * H, 2 * W, y2.cpitch), np.float16)}, B)
        assert np.abs(_nchw(out[yv.buf], C) - want).max() < 4e-3
        assert np.abs(_nchw(out[y2.buf], C) - np.maximum(want, 0)).max() < 4e-3
    elif mode == "bilinear_frac":                                # the Segmentor's 1/14 -> 1/16 grid (size given: ratio = in / out)
        x8 = _f16(_rand(1, C, 8, 16, seed=3))
        xv2 = b.tensor(8, 16, C)
        yv = b.tensor(7, 14, C)
        b.ew("t", xv2, yv, mode=1, ratio=(np.float32(8) / np.float32(7), np.float32(16) / np.float32(14)))
        want = F.interpolate(torch.from_numpy(x8), size=(7, 14), mode="bilinear", align_corners=False).numpy()
        out = run_plan(ctx, WeightPack(), b, {xv2.buf: nhwc_pad(x8, xv2.cpitch)}, {yv.buf: ((1, 7, 14, yv.cpitch), np.float16)}, 1)
        assert np.abs(_nchw(out[yv.buf], C) - want).max() < 3e-3
    elif mode == "area4":                                        # downsample_groups(g, 1/4, 'area')
        x4 = _f16(_rand(B, C, 8, 12, seed=4))
        xv2, yv = b.tensor(8, 12, C), b.tensor(2, 3, C)
        b.ew("t", xv2, yv, mode=2, factor=4)
        want = F.interpolate(torch.from_numpy(x4), scale_factor=0.25, mode="area").numpy()
        out = run_plan(ctx, WeightPack(), b, {xv2.buf: nhwc_pad(x4, xv2.cpitch)}, {yv.buf: ((B, 2, 3, yv.cpitch), np.float16)}, B)
        assert np.abs(_nchw(out[yv.buf], C) - want).max() < 2e-3
    else:                                                        # the broadcast copy of the image features into both objects' concat buffers
        yv, y2 = b.tensor(H, W, C), b.tensor(H, W, C)
        b.ew("t", xv, yv, src_bcast=True, dual=y2)
        out = run_plan(ctx, WeightPack(), b, ups, {yv.buf: ((B, H, W, yv.cpitch), np.float16), y2.buf: ((B, H, W, y2.cpitch), np.float16)}, B)
        for f in range(B):
            assert np.array_equal(_nchw(out[yv.buf], C)[f], x[0]) and np.array_equal(_nchw(out[y2.buf], C)[f], np.maximum(x[0], 0))


@pytest.mark.gpu
@pytest.mark.parametrize("k", [3, 5])
def test_dwconv_op(ctx, k):
    from vsdeoldify_amd.plan import PlanBuilder, WeightPack
    B, C, H, W = 2, 40, 9, 11
    x, w, bias = _f16(_rand(B, C, H, W, seed=5)), _f16(_rand(C, 1, k, k, seed=6, scale=0.3)), _rand(C, seed=7, scale=0.1)
    pack, b = WeightPack(), PlanBuilder()
    xv, yv = b.tensor(H, W, C), b.tensor(H, W, C)
    wp = np.zeros((k * k, xv.span), np.float16)
    wp[:, :C] = w.reshape(C, k * k).T
    bp = np.zeros(xv.span, np.float32)
    bp[:C] = bias
    b.dwconv("t", xv, yv, pack.add(wp), pack.add(bp) if k == 3 else -1, xv.span, k)
    want = F.conv2d(torch.from_numpy(x), torch.from_numpy(w), torch.from_numpy(bias) if k == 3 else None, padding=k // 2, groups=C).numpy()
    out = run_plan(ctx, pack, b, {xv.buf: nhwc_pad(x, xv.cpitch)}, {yv.buf: ((B, H, W, yv.cpitch), np.float16)}, B)
    assert np.abs(_nchw(out[yv.buf], C) - want).max() < 5e-3


@pytest.mark.gpu
@pytest.mark.parametrize("dim,P", [(32, (7, 14)), (128, (5, 9))])
def test_cross_channel_attention_ops(ctx, dim, P):
    """chan_attn + the W_FROM_BUF conv = softmax(normalize(q) normalize(k)^T * temperature) @ v per head (resnet.py:310-331)"""
    from oracle import colormnet_net as ON
    from vsdeoldify_amd.plan import PlanBuilder, View, WeightPack
    H, W = P
    heads, C2 = 8, 2 * dim
    q, k, v = (_f16(_rand(1, C2, H, W, seed=s)) for s in (8, 9, 10))
    temp = np.linspace(3.0, 12.0, heads).astype(np.float32)
    pack, b = WeightPack(), PlanBuilder()
    qv, kv = b.tensor(H, W, C2), b.tensor(H, W, 2 * C2)
    cc = C2 // heads
    from vsdeoldify_amd.colormnet_net import chan_attn_splits
    S = chan_attn_splits(H * W, heads, cc)
    wbuf = b.buf(C2 * C2, 2, zero_init=True)
    b.chan_attn("attn", qv, View(kv.buf, 0, kv.cpitch, H, W, C2, C2), heads, pack.add(temp), wbuf, C2 // 8, b.buf(heads * S * cc * cc, 4), b.buf(S * 2 * C2, 4))
    if C2 % 64 == 0:
        ov = b.tensor(H, W, C2)
        b.conv_dyn("attn@v", View(kv.buf, C2, kv.cpitch, H, W, C2, C2), View(wbuf, 0, C2, 1, C2, C2, C2), ov, C2)
    kvn = np.concatenate([k, v], 1)
    dl = {wbuf: ((C2, C2), np.float16)}
    if C2 % 64 == 0:
        dl[ov.buf] = ((1, H, W, ov.cpitch), np.float16)
    out = run_plan(ctx, pack, b, {qv.buf: nhwc_pad(q, qv.cpitch), kv.buf: nhwc_pad(kvn, kv.cpitch)}, dl, 1)
    tq, tk, tv = (torch.from_numpy(t).reshape(1, heads, cc, H * W) for t in (q, k, v))
    attn = ((F.normalize(tq, dim=-1) @ F.normalize(tk, dim=-1).transpose(-2, -1)) * torch.from_numpy(temp).view(heads, 1, 1)).softmax(-1)
    Wm = out[wbuf].astype(np.float32)
    for h in range(heads):
        blk = Wm[h * cc:(h + 1) * cc, h * cc:(h + 1) * cc]
        assert np.abs(blk - attn[0, h].numpy()).max() < 2e-3, h
        Wm[h * cc:(h + 1) * cc, h * cc:(h + 1) * cc] = 0
    assert np.abs(Wm).max() == 0                                # off-diagonal blocks stay zero
    if C2 % 64 == 0:
        want = (attn @ tv).reshape(1, C2, H, W).numpy()
        assert np.abs(_nchw(out[ov.buf], C2) - want).max() < 8e-3


@pytest.mark.gpu
@pytest.mark.parametrize("T", [50, 513])
def test_mha64_op(ctx, T):
    """6 heads x 64, flash form, against softmax(q k^T / 8) v"""
    from vsdeoldify_amd.plan import PlanBuilder, View, WeightPack
    B, heads, D = 2, 6, 384
    qkv = _f16(_rand(B, T, 3 * D, seed=11, scale=0.8))
    b = PlanBuilder()
    qv = View(b.buf(T * 3 * D, 2), 0, 3 * D, 1, T, 3 * D, 3 * D)
    yv = View(b.buf(T * D, 2), 0, D, 1, T, D, D)
    b.mha64("t", qv, 0, D, 2 * D, yv, heads, T, 0.125)
    out = run_plan(ctx, WeightPack(), b, {qv.buf: qkv.astype(np.float16)}, {yv.buf: ((B, T, D), np.float16)}, B)[yv.buf].astype(np.float32)
    t = torch.from_numpy(qkv).reshape(B, T, 3, heads, 64).permute(2, 0, 3, 1, 4)
    want = (torch.softmax(t[0] * 0.125 @ t[1].transpose(-2, -1), -1) @ t[2]).transpose(1, 2).reshape(B, T, D).numpy()
    assert np.abs(out - want).max() < 6e-3, float(np.abs(out - want).max())


@pytest.mark.gpu
def test_cbam_gru_planar_ops(ctx):
    from oracle import colormnet_net as ON
    from vsdeoldify_amd.plan import PlanBuilder, WeightPack
    B, C, H, W, HD = 2, 64, 6, 9, 16
    x = _f16(_rand(B, C, H, W, seed=12))
    sd = {"a.ChannelGate.mlp.1.weight": _rand(C // 16, C, seed=13, scale=0.3), "a.ChannelGate.mlp.1.bias": _rand(C // 16, seed=14, scale=0.1),
          "a.ChannelGate.mlp.3.weight": _rand(C, C // 16, seed=15, scale=0.5), "a.ChannelGate.mlp.3.bias": _rand(C, seed=16, scale=0.1),
          "a.SpatialGate.spatial.conv.weight": _rand(1, 2, 7, 7, seed=17, scale=0.2), "a.SpatialGate.spatial.conv.bias": _rand(1, seed=18, scale=0.1)}
    pack, b = WeightPack(), PlanBuilder()
    xv, yv, y2 = b.tensor(H, W, C), b.tensor(H, W, C), b.tensor(H, W, C)
    woff = pack.add(np.concatenate([sd[k].reshape(-1) for k in sd]))
    b.cbam("cbam", xv, yv, woff, b.buf(3 * C, 4), b.buf(H * W * 2, 4), dual=y2)
    # GRU on fp32 planar hidden + planar in / out round trip with the KeyProjection activations
    vals = _f16(_rand(B, 3 * HD, H, W, seed=19))
    hid = _rand(B, HD, H, W, seed=20, scale=0.5)
    vv = b.tensor(H, W, 3 * HD)
    hb, ho = b.buf(HD * H * W, 4), b.buf(HD * H * W, 4)
    b.gru("gru", vv, hb, ho, HD)
    pin = b.buf(5 * H * W, 4)
    pv = b.tensor(H, W, 5)
    b.planar_in("pin", pin, 5, pv)
    outs = [b.buf(5 * H * W, 4) for _ in range(4)]
    for act, ob in enumerate(outs):
        b.planar_out(f"pout{act}", pv, 0, 5, ob, act)
    p5 = _rand(B, 5, H, W, seed=21)
    dl = {yv.buf: ((B, H, W, yv.cpitch), np.float16), y2.buf: ((B, H, W, y2.cpitch), np.float16), ho: ((B, HD, H, W), np.float32)}
    dl.update({ob: ((B, 5, H, W), np.float32) for ob in outs})
    out = run_plan(ctx, pack, b, {xv.buf: nhwc_pad(x, xv.cpitch), vv.buf: nhwc_pad(vals, vv.cpitch), hb: hid, pin: p5}, dl, B)
    tsd_ = {k: torch.from_numpy(v) for k, v in sd.items()}
    want = torch.from_numpy(x) + ON.cbam(tsd_, "a", torch.from_numpy(x))
    assert np.abs(_nchw(out[yv.buf], C) - want.numpy()).max() < 6e-3
    assert np.abs(_nchw(out[y2.buf], C) - np.maximum(want.numpy(), 0)).max() < 6e-3
    g = ON._gru(torch.from_numpy(vals).unsqueeze(0), torch.from_numpy(hid).unsqueeze(0), HD)[0].numpy()
    assert np.abs(out[ho] - g).max() < 2e-4
    h5 = _f16(p5)
    for act, fn in enumerate((lambda t: t, lambda t: t * t + 1, lambda t: 1 / (1 + np.exp(-t)), np.tanh)):
        assert np.abs(out[outs[act]] - fn(h5)).max() < 1e-5, act


@pytest.mark.gpu
def test_decoder_input_op(ctx):
    """HAVC_OP_CMN_DECODER_IN (round 5): [g16 of the frame for every object | readout | hidden] and its rectified twin in one launch -- exact
    (fp16 copies and fp32 -> fp16 roundings), including the bytes around the written channel range"""
    from vsdeoldify_amd.plan import PlanBuilder, WeightPack, View, pitch_for
    B, H, W, Cg, CV, HD = 2, 5, 7, 32, 24, 8
    P = H * W
    g = _f16(_rand(1, Cg, H, W, seed=31))
    ro, hid = _rand(B, CV, H, W, seed=32, scale=2.0), _rand(B, HD, H, W, seed=33)
    pack, b = WeightPack(), PlanBuilder()
    pack.add(np.zeros(8, np.float32))
    gv = b.tensor(H, W, Cg)
    span = Cg + CV + HD
    pitch = pitch_for(span + 8)
    dc_buf, dcr_buf = b.buf(P * pitch, 2, zero_init=True), b.buf(P * pitch, 2, zero_init=True)
    rb, hb = b.buf(CV * P, 4), b.buf(HD * P, 4)
    b.cmn_decoder_in("dec_in", gv, rb, CV, hb, HD, View(dc_buf, 0, pitch, H, W, span, span), dcr_buf)
    gin = np.concatenate([nhwc_pad(g, gv.cpitch)] * B, 0)                       # (the op reads frame 0 only)
    gin[1:] = 77
    out = run_plan(ctx, pack, b, {gv.buf: gin, rb: ro, hb: hid}, {dc_buf: ((B, H, W, pitch), np.float16), dcr_buf: ((B, H, W, pitch), np.float16)}, B)
    want = np.concatenate([np.broadcast_to(g, (B, Cg, H, W)), _f16(ro), _f16(hid)], 1).transpose(0, 2, 3, 1).astype(np.float16)
    assert np.array_equal(out[dc_buf][..., :span], want)
    assert np.array_equal(out[dcr_buf][..., :span], np.maximum(want, np.float16(0)))
    assert (out[dc_buf][..., span:] == 0).all() and (out[dcr_buf][..., span:] == 0).all()


# =====================================================================================================================================
# GPU: the network and the frame wrapper
# =====================================================================================================================================
_NET = {}


def gpu_network(seed=SEED):
    from vsdeoldify_amd.colormnet_net import ColorMNetNetwork
    if seed not in _NET:
        _NET[seed] = ColorMNetNetwork(synth_colormnet_state_dict(seed), device_index=0)
    return _NET[seed]


def rel(a, b):
    a, b = np.asarray(a, np.float32), np.asarray(b, np.float32)
    return float(np.abs(a - b).max()) / max(1.0, float(np.abs(b).max())), float(np.sqrt(((a - b) ** 2).mean()) / (np.sqrt((b ** 2).mean()) + 1e-12))


def feat_nchw(t, rows_hw, C):
    """a feature tensor the plan wrote (NHWC fp16, pitched) -> float32 [C, h, w]"""
    h, w = rows_hw
    a = t.cpu().numpy()
    pitch = (a.size - 128) // (h * w)
    return a[:h * w * pitch].reshape(h, w, pitch)[..., :C].astype(np.float32).transpose(2, 0, 1)


@pytest.mark.gpu
def test_gpu_network_modules_match_the_oracle_and_the_executed_reference():
    """encode_key / encode_value / segment / short_term_attn on the MI355X (fp16 activations, fp32 accumulation) against the fp32 oracle on
    the same inputs, and against the executed-reference vectors directly.  Tolerances: relative to the tensor's largest magnitude."""
    sd, net = tsd(), gpu_network()
    dev = net.device
    with torch.no_grad():
        frame = _frame()
        key, shr, sel, f16, f8, f4 = O.encode_key(sd, frame)
        gk, gs, ge, gf, _, _ = net.encode_key(frame.to(dev), need_ek=True, need_sk=True)
        h, w = key.shape[-2:]
        torch.cuda.synchronize()
        report = {}
        for n_, want, got in (("f16", f16[0], feat_nchw(gf.g16, (h, w), 1024)), ("f8", f8[0], feat_nchw(gf.g8, (2 * h, 2 * w), 512)),
                              ("f4", f4[0], feat_nchw(gf.g4, (4 * h, 4 * w), 256)), ("key", key, gk.cpu()), ("shrinkage", shr, gs.cpu()),
                              ("selection", sel, ge.cpu())):
            report[n_] = rel(got, want.numpy())
            assert report[n_][0] < 0.03 and report[n_][1] < 0.01, (n_, report[n_])
        check("key", gk.cpu(), stride_of("key"), 0.03)               # the executed-reference vectors, no oracle in between
        check("selection", ge.cpu(), stride_of("selection"), 0.03)
        masks = torch.from_numpy(MOD["masks"]).unsqueeze(0)
        h0 = torch.from_numpy(MOD["hidden0"]).unsqueeze(0)
        # value / decoder: feed the GPU its own features (what the product does) and compare with the oracle on the oracle's features
        for deep in (True, False):
            val, h1 = O.encode_value(sd, frame, f16, h0, masks, is_deep_update=deep)
            gv, gh = net.encode_value(frame.to(dev), gf, h0.to(dev), masks.to(dev), is_deep_update=deep)
            r = rel(gv.cpu().numpy(), val.numpy())
            assert r[0] < 0.03 and r[1] < 0.01, ("value", deep, r)
            assert rel(gh.cpu().numpy(), h1.numpy())[0] < 0.02
        check("value_deep1", gv.cpu(), stride_of("value_deep1"), 0.03)
        readout = torch.from_numpy(MOD["readout"].astype(np.float32)).unsqueeze(0)
        for h_out in (True, False):
            hid, prob = O.segment(sd, (f16, f8, f4), readout, h0, h_out=h_out)
            ghid, gprob, _ = net.segment((gf, gf, gf), readout.to(dev), h0.to(dev), h_out=h_out, strip_bg=False)
            d = np.abs(gprob.cpu().numpy() - prob.numpy())
            assert d.max() < 0.02 and d.mean() < 0.002, ("prob", h_out, float(d.max()), float(d.mean()))      # ab = 110 * prob: < 2.2 / 0.22 units
            if h_out:
                assert rel(ghid.cpu().numpy(), hid.numpy())[0] < 0.02
            else:
                assert ghid is None
        check("prob_hout1", gprob.cpu() if h_out else gprob.cpu(), stride_of("prob_hout1"), 0.02)
        k2 = torch.from_numpy(MOD["k2"]).unsqueeze(0)
        short, _ = O.short_term_attn(sd, key, k2, val.flatten(1, 2), key.shape[-2:])
        gshort, _ = net.short_term_attn(key.to(dev), k2.to(dev), val.flatten(1, 2).to(dev), None, key.shape[-2:])
        r = rel(gshort.cpu().numpy(), short.numpy())
        assert r[0] < 0.02 and r[1] < 0.005, ("short", r)
        print("relative errors (max / rms):", {k: tuple(round(v, 5) for v in r_) for k, r_ in report.items()})


@pytest.mark.gpu
def test_gpu_lab_transforms_match_the_oracle():
    net = gpu_network()
    rgb = REN["refs"][0]
    lab = net.image_to_lab(rgb)
    want = O.frame_to_lab_tensor(rgb)
    assert (lab.cpu() - want).abs().max() < 2e-6
    ab = torch.tanh(torch.randn(2, *rgb.shape[:2], generator=torch.Generator().manual_seed(3)) * 0.4)
    got = net.lab_to_image(lab[:1], ab.to(net.device))
    ref = O.lab_tensor_to_rgb(want[:1], ab)
    d = np.abs(got.astype(np.int32) - ref.astype(np.int32))
    assert d.max() <= 1 and (d > 0).mean() < 1e-3


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(SCENARIOS))
def test_gpu_render_matches_the_reference_render_class(name):
    """ColorMNetRender (drop-in) on the MI355X over the 9-frame clip of the fixture — exemplars with frames 0 and 4, memory frames every second
    frame — in the four scenarios recorded from the reference's own ColorMNetRender (keep the memory / render_vivid reset / frame_propagate /
    max_memory_frames reset).  fp16 activations vs fp32: CIEDE2000 statistics per frame; the memory read is a top-k softmax, so a handful of
    pixels may pick another memory element (tolerated through the p99 / max split, as for the DeOldify path)."""
    from oracle import imaging
    got, want = run_render_scenario(name, gpu_network()), want_of(name)
    worst = (0.0, 0.0)
    for t in range(len(want)):
        de = imaging.delta_e00_images(got[t], want[t])
        worst = max(worst, (float(de.mean()), float(np.percentile(de, 99))))
        # measured worst frame per scenario on MI355X (round 5: keep 0.211 / 1.129, vivid 0.153 / 0.919, propagate 0.122 / 0.621, capped 0.198 / 1.028) + 15 %
        assert got[t].shape == want[t].shape and de.mean() < 0.245 and np.percentile(de, 99) < 1.30, (name, t, float(de.mean()), float(np.percentile(de, 99)), float(de.max()))
    print(f"{name}: worst frame mean dE00 %.3f, p99 %.3f" % worst)


@pytest.mark.gpu
def test_gpu_render_first_frames_without_a_reference_pass_through():
    """colormnet_render.py:242-247: nothing to propagate from until the first reference image arrives"""
    from PIL import Image
    from vsdeoldify_amd.colormnet_render import ColorMNetRender
    rnd = ColorMNetRender(vid_length=4, network=gpu_network())
    img = Image.fromarray(np.stack([REN["frames"][0]] * 3, -1))
    rnd.set_ref_frame(None)
    assert rnd.colorize_frame(0, img) is img
    with pytest.raises(NotImplementedError):
        ColorMNetRender(image_size=256, vid_length=4, network=gpu_network())


@pytest.mark.gpu
def test_gpu_key_lookahead_equals_encode_key_frame_by_frame():
    """prefetch_keys (the key encoder on several frames per pass, split-K counts chosen for the batch) against encode_key one frame at a time:
    the same arithmetic up to the fp32 summation order of the split-K parts; the FIFO hands the entries out in order and only when armed."""
    net = gpu_network()
    dev = net.device
    g = torch.Generator().manual_seed(5)
    frames = [torch.tanh(torch.randn(1, 1, 112, 224, generator=g)).repeat(1, 3, 1, 1) for _ in range(5)]
    single = [net.encode_key(f.to(dev)) for f in frames]
    entries = net.prefetch_keys([f[0].to(dev) for f in frames], max_batch=8)           # 5 of 8: a ragged last batch
    assert len(entries) == 5
    h, w = 7, 14
    for i, f in enumerate(frames):
        if i == 2:                                                            # not armed: computed on the spot
            k_plain = net.encode_key(f.to(dev))[0]
            assert rel(k_plain.cpu().numpy(), single[i][0].cpu().numpy())[0] == 0.0
        net.expect_prefetched(entries[i])
        gk, gs, ge, gf, _, _ = net.encode_key(f.to(dev))
        assert net._armed is None                                             # one shot
        torch.cuda.synchronize()
        sk, ss, se, sf, _, _ = single[i]
        for n_, a, b_ in (("key", gk, sk), ("shrinkage", gs, ss), ("selection", ge, se)):
            r = rel(a.cpu().numpy(), b_.cpu().numpy())
            assert r[0] < 4e-3 and r[1] < 1e-3, (i, n_, r)
        for n_, rows, C in (("g16", (h, w), 1024), ("g8", (2 * h, 2 * w), 512), ("g4", (4 * h, 4 * w), 256)):
            r = rel(feat_nchw(getattr(gf, n_), rows, C), feat_nchw(getattr(sf, n_), rows, C))
            assert r[0] < 4e-3 and r[1] < 1e-3, (i, n_, r)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["keep", "capped"])
def test_gpu_colorize_batch_frames_with_lookahead_matches_the_reference_render_class(name):
    """ColorMNetRender.colorize_batch_frames (colormnet_render.py:186-195) knows its frames up front: the key encoder runs 4 frames per pass
    ahead of the sequential step.  Same fixtures and tolerances as the frame-by-frame test (exemplars with frames 0 and 4; "capped" resets the
    memory every 3 frames, i.e. inside a look-ahead window), and within 2 LSB of the frame-by-frame drop-in for 99.9 % of the bytes."""
    from PIL import Image
    from oracle import imaging
    from vsdeoldify_amd.colormnet_render import ColorMNetRender
    kw, propagate = SCENARIOS[name]
    frames, refs, want = REN["frames"], REN["refs"], want_of(name)
    rnd = ColorMNetRender(image_size=-1, vid_length=len(frames), enable_resize=False, encode_mode=1, propagate=propagate, network=gpu_network(),
                          lookahead=4, **kw)
    rnd.set_config("mem_every", int(REN["mem_every"]))
    imgs = [Image.fromarray(np.stack([fr] * 3, -1)) for fr in frames]
    ref_list = [Image.fromarray(refs[0]) if t == 0 else (Image.fromarray(refs[1]) if t == 4 else None) for t in range(len(frames))]
    got = np.stack([np.asarray(o) for o in rnd.colorize_batch_frames(imgs, ref_list, propagate)])
    assert not rnd._ahead and rnd.network._armed is None
    for t in range(len(want)):
        de = imaging.delta_e00_images(got[t], want[t])
        assert de.mean() < 0.3 and np.percentile(de, 99) < 1.6, (name, t, float(de.mean()), float(np.percentile(de, 99)))     # (the look-ahead pass's split-K order moves a few pixels)
    plain = run_render_scenario(name, gpu_network())
    d = np.abs(got.astype(np.int32) - plain.astype(np.int32))
    assert (d <= 2).mean() > 0.999, (int(d.max()), float((d <= 2).mean()))


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(SCENARIOS))
def test_gpu_fast_step_equals_the_line_by_line_classes(name):
    """colormnet_fast.py (banked memory, padded frames, forked short-term attention, preallocated results) against colormnet_core /
    colormnet_memory (the reference's tensor bookkeeping line by line) on the four recorded scenarios: the same kernels on the same numbers in
    the same order -- identical frames."""
    net = gpu_network()
    assert net.fast
    fast = run_render_scenario(name, net)
    net.fast = False
    try:
        plain = run_render_scenario(name, net)
    finally:
        net.fast = True
    d = np.abs(fast.astype(np.int32) - plain.astype(np.int32))
    assert d.max() == 0, (name, int(d.max()), float((d > 0).mean()))


@pytest.mark.gpu
def test_gpu_fast_step_long_clip_with_consolidation_and_lookahead():
    """Plain assertions, no retry (round 6).  Round 5 wrapped this body in one retry after an unexplained 1-in-40 failure; round 6 hunted it with
    stream jitter (tests/test_gpu_colormnet_stress.py, tools/cmn_race_stress.py, profiles/r6_cmn_race_stress.txt) and closed the one lifetime hole the
    audit found (a read enqueued ahead that no _read consumed: colormnet_fast.FastInferenceCore.step_*).  Every comparison here is deterministic by
    construction: a failure is a race or a machine fault and must fail the test."""
    _long_clip_body()


def _long_clip_body():
    """60 frames with mem_every = 2 and a small working memory: several consolidations into long-term prototypes, usage counters, the removal of
    obsolete long-term elements -- through both implementations, frame by frame and with the key look-ahead; DeviceImage in / out."""
    from vsdeoldify_amd.colormnet_render import ColorMNetRender
    from vsdeoldify_amd.device import DeviceImage
    net = gpu_network()
    r = np.random.default_rng(4)
    base = REN["frames"][0].astype(np.float32)
    frames = [np.stack([np.clip(base + 6 * np.sin(t / 3.0) + r.normal(0, 2, base.shape), 0, 255).astype(np.uint8)] * 3, -1) for t in range(60)]
    ref = REN["refs"][0]

    def run(fast, lookahead):
        net.fast = fast
        try:
            rnd = ColorMNetRender(image_size=-1, vid_length=len(frames), encode_mode=1, max_memory_frames=500, reset_on_ref_update=False, network=net,
                                  lookahead=lookahead)
            rnd.set_config("mem_every", 2)
            rnd.set_config("max_mid_term_frames", 4)
            rnd.set_config("min_mid_term_frames", 2)
            rnd.set_config("num_prototypes", 16)
            rnd.set_config("max_long_term_elements", 60)
            rnd.set_config("enable_long_term_count_usage", True)
            dev = [DeviceImage.from_numpy(net.ctx, f) for f in frames]
            refs = [DeviceImage.from_numpy(net.ctx, ref) if t == 0 else None for t in range(len(frames))]
            outs = rnd.colorize_batch_frames(dev, refs, False)
            mem = rnd.processor.memory
            return np.stack([o.numpy() for o in outs]), (mem.work_mem.size, mem.long_mem.size)
        finally:
            net.fast = True
    a, sa = run(True, 1)
    b, sb = run(False, 1)
    assert sa == sb and sa[1] > 0, (sa, sb)                                   # the long-term memory was engaged, both ended with the same sizes
    assert np.array_equal(a, b), (int(np.abs(a.astype(int) - b.astype(int)).max()), float((a != b).mean()))
    c, sc = run(True, 8)                                                      # + key look-ahead (split-K counts of the batched pass: fp32 summation order)
    d = np.abs(a.astype(int) - c.astype(int))
    print(f"long clip: memory sizes fast {sa} plain {sb} look-ahead {sc}; look-ahead vs frame by frame: max |d| {int(d.max())}, within 2 LSB {float((d <= 2).mean()):.5f}")
    assert sc == sa and (d <= 2).mean() > 0.999, (sc, int(d.max()), float((d <= 2).mean()))
    # round 5: with the look-ahead the READ of frame t+1 runs under the decoder of frame t between memory frames (colormnet_fast.READ_AHEAD):
    # the same kernels on the same numbers, on another stream -- identical frames and memory sizes with it switched off
    import vsdeoldify_amd.colormnet_fast as cf
    assert cf.READ_AHEAD
    cf.READ_AHEAD = False
    try:
        c0, sc0 = run(True, 8)
    finally:
        cf.READ_AHEAD = True
    assert sc0 == sc and np.array_equal(c, c0), (sc0, sc, int(np.abs(c.astype(int) - c0.astype(int)).max()), float((c != c0).mean()))


@pytest.mark.gpu
def test_gpu_read_ahead_runs_and_a_caller_leaving_the_announced_order_drops_it():
    """The read enqueued ahead of its frame must (1) really happen on plain frames between memory frames, (2) leave NO trace (usage counters) when the
    caller then steps another frame than the announced one: frames and memory identical to the same call sequence with READ_AHEAD off."""
    import vsdeoldify_amd.colormnet_fast as cf
    from vsdeoldify_amd.colormnet_render import ColorMNetRender
    from vsdeoldify_amd.device import DeviceImage
    net = gpu_network()
    r = np.random.default_rng(9)
    base = REN["frames"][0].astype(np.float32)
    frames = [np.stack([np.clip(base + 5 * np.cos(t / 2.0) + r.normal(0, 2, base.shape), 0, 255).astype(np.uint8)] * 3, -1) for t in range(14)]
    ref = REN["refs"][0]

    def run():
        rnd = ColorMNetRender(image_size=-1, vid_length=100, encode_mode=1, max_memory_frames=500, reset_on_ref_update=False, network=net, lookahead=8)
        rnd.set_config("mem_every", 3)
        rnd.set_config("enable_long_term_count_usage", True)
        dev = [DeviceImage.from_numpy(net.ctx, f) for f in frames]
        outs = []
        rnd.prefetch(dev[:8])
        for t in range(6):                                                    # frames 0 .. 5 in the announced order, 6 and 7 announced as plain too
            rnd.set_ref_frame(DeviceImage.from_numpy(net.ctx, ref) if t == 0 else None, False)
            rnd.next_is_plain = True
            outs.append(rnd.colorize_frame(t, dev[t]).numpy())
        hits = getattr(rnd.processor, "reads_ahead", 0)
        pending = getattr(rnd.processor, "_ahead_read", None) is not None
        for t in (10, 11, 12):                                                # ... but the caller jumps: frame 6's read, if it ran ahead, is dropped
            rnd.set_ref_frame(None, False)
            outs.append(rnd.colorize_frame(t, dev[t]).numpy())
        import torch
        torch.cuda.synchronize()
        mem = rnd.processor.memory
        use = mem.work_mem.get_usage().float().cpu().numpy() if mem.work_mem.count_usage else None
        return np.stack(outs), hits, pending, use

    a, hits, pending, use_a = run()
    assert hits >= 3, hits                                                    # mem_every = 3: two of three frames are followed by a read-ahead
    cf.READ_AHEAD = False
    try:
        b, hits_b, pending_b, use_b = run()
    finally:
        cf.READ_AHEAD = True
    assert hits_b == 0 and not pending_b
    assert np.array_equal(a, b), (int(np.abs(a.astype(int) - b.astype(int)).max()), float((a != b).mean()))
    if use_a is not None:
        assert np.array_equal(use_a, use_b)
    print(f"read-ahead: {hits} reads ran ahead, one pending at the jump: {pending}")
