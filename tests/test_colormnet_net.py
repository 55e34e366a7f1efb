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


def oracle_clip(sd, frames, refs, mem_every, backend, network=None, device="cpu"):
    """the all-oracle frame loop: oracle network + the drop-in InferenceCore / MemoryManager on the oracle's memory functions"""
    from vsdeoldify_amd.colormnet_render import default_config
    from vsdeoldify_amd.colormnet_core import InferenceCore
    net = network or O.Network(sd)
    cfg = default_config(len(frames), 0)
    cfg.update(mem_every=mem_every, key_dim=64, value_dim=512, hidden_dim=64)
    proc = InferenceCore(net, cfg, memory_backend=backend)
    outs = []
    for t, fr in enumerate(frames):
        rgb = np.stack([fr] * 3, -1)
        lab = O.frame_to_lab_tensor(rgb)
        lll = lab[:1].repeat(3, 1, 1).to(device)
        ref = refs.get(t)
        with torch.no_grad():
            if ref is not None:
                m = O.frame_to_lab_tensor(ref).to(device)
                proc.set_all_labels([1, 2])
                ab = proc.step_AnyExemplar(lll, m[:1].repeat(3, 1, 1), m[1:3], [1, 2], end=False)
            else:
                ab = proc.step_AnyExemplar(lll, None, None, None, end=False)
        outs.append(O.lab_tensor_to_rgb(lll[:1].cpu(), ab.cpu()))
    return np.stack(outs)


def test_oracle_frame_loop_matches_the_reference_render_class():
    """the reference's own ColorMNetRender.colorize_frame over 9 frames (exemplars with frames 0 and 4) vs oracle network + drop-in
    InferenceCore / MemoryManager: u8 frames, identical up to float noise at the rounding boundary"""
    from tests.test_colormnet_memory import OracleBackend
    frames, refs, want = REN["frames"], REN["refs"], REN["outs"]
    got = oracle_clip(tsd(), list(frames), {0: refs[0], 4: refs[1]}, int(REN["mem_every"]), OracleBackend())
    d = np.abs(got.astype(np.int32) - want.astype(np.int32))
    assert got.shape == want.shape and d.max() <= 1 and (d > 0).mean() < 2e-3, (int(d.max()), float((d > 0).mean()))
