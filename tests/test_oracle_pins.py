"""CPU (-m "not gpu"): INDEPENDENT pins for the parts of the oracle whose reference-side library is absent from the build
container ("oracle sub-parts are self-referential").

  * encoder: torchvision is absent, so the golden fixtures were generated with oracle/resnet.py standing in for
    torchvision.models.resnet101/34 (tools/refshim.py).  Here both oracle/resnet.py (the stand-in) and oracle/unet.encoder
    (the functional restatement the GPU tests compare against) are checked against the `transformers` package's ResNet --
    an independent implementation of the same published architecture (v1.5: stride on the 3x3 conv,
    downsample_in_bottleneck=False), full resnet101 / resnet34 depth, seeded weights mapped key by key.
  * .pth loading: vsdeoldify_amd.render._load_pth against files written in both layouts Learner.save / torch.save produce
    (fastai/basic_train.py:264-286: {'model','opt'} or a bare state dict).
"""
import numpy as np
import pytest
import torch

from oracle import resnet as oresnet, unet
from vsdeoldify_amd.synth import synth_state_dict

ARCH = {"wide": ("resnet101", "bottleneck", [3, 4, 23, 3], [256, 512, 1024, 2048]),
        "deep": ("resnet34", "basic", [3, 4, 6, 3], [64, 128, 256, 512])}


def _hf_resnet(arch, sd):
    from transformers import ResNetConfig, ResNetModel
    name, kind, depths, hidden = ARCH[arch]
    cfg = ResNetConfig(num_channels=3, embedding_size=64, hidden_sizes=hidden, depths=depths, layer_type=kind, hidden_act="relu",
                       downsample_in_first_stage=False, downsample_in_bottleneck=False)
    m = ResNetModel(cfg).eval()
    tgt = m.state_dict()
    new = {}

    def put(dst, src):
        for k in ("weight", "bias", "running_mean", "running_var", "num_batches_tracked"):
            if src + "." + k in sd:
                new[dst + "." + k] = torch.as_tensor(np.asarray(sd[src + "." + k]))
    new["embedder.embedder.convolution.weight"] = torch.as_tensor(sd["layers.0.0.weight"])
    put("embedder.embedder.normalization", "layers.0.1")
    nconv = 3 if kind == "bottleneck" else 2
    for li, n in enumerate(depths):
        for bi in range(n):
            src, dst = f"layers.0.{4 + li}.{bi}", f"encoder.stages.{li}.layers.{bi}"
            for k in range(nconv):
                new[f"{dst}.layer.{k}.convolution.weight"] = torch.as_tensor(sd[f"{src}.conv{k + 1}.weight"])
                put(f"{dst}.layer.{k}.normalization", f"{src}.bn{k + 1}")
            if f"{src}.downsample.0.weight" in sd:
                new[f"{dst}.shortcut.convolution.weight"] = torch.as_tensor(sd[f"{src}.downsample.0.weight"])
                put(f"{dst}.shortcut.normalization", f"{src}.downsample.1")
    assert set(new) == set(tgt), (sorted(set(tgt) - set(new))[:5], sorted(set(new) - set(tgt))[:5])
    m.load_state_dict(new, strict=True)
    return m


@pytest.mark.parametrize("arch", ["wide", "deep"])
def test_encoder_matches_transformers_resnet(arch):
    pytest.importorskip("transformers")
    sd = synth_state_dict(arch, 5)
    tsd = {k: torch.as_tensor(np.asarray(v)) for k, v in sd.items() if k.startswith("layers.0.")}
    x = torch.from_numpy(np.random.default_rng(0).standard_normal((1, 3, 80, 96)).astype(np.float32))
    hf = _hf_resnet(arch, tsd)
    stem = {}
    hf.embedder.embedder.register_forward_hook(lambda m, i, o: stem.__setitem__("y", o))
    with torch.no_grad():
        out = hf(x, output_hidden_states=True)
        skips, top = unet.encoder(tsd, x, ARCH[arch][0])
    want = [stem["y"]] + list(out.hidden_states[1:])          # relu(bn(conv1)), layer1..layer4
    got = skips + [top]
    assert len(want) == len(got) == 5
    for g, w in zip(got, want):
        assert g.shape == w.shape
        assert float((g - w).abs().max()) <= 2e-4 * max(1.0, float(w.abs().max())), float((g - w).abs().max())
    # the stand-in that replaced torchvision when the golden fixtures were generated (tools/refshim.py)
    name = ARCH[arch][0]
    m = getattr(oresnet, name)().eval()
    body = torch.nn.Sequential(*list(m.children())[:-2])                                     # create_body cut=-2
    body.load_state_dict({k[len("layers.0."):]: v for k, v in tsd.items()}, strict=True)
    with torch.no_grad():
        y = body(x)
    assert float((y - want[-1]).abs().max()) <= 2e-4 * max(1.0, float(want[-1].abs().max()))


@pytest.mark.parametrize("layout", ["model_opt", "bare"])
def test_load_pth_both_layouts(tmp_path, layout):
    """render._load_pth (the Learner.load path, fastai/basic_train.py:270-283): {'model': sd, 'opt': ...} and a bare dict."""
    from vsdeoldify_amd import render
    sd = {k: torch.as_tensor(np.asarray(v)) for k, v in synth_state_dict("deep", 9).items()}
    f = tmp_path / "ColorizeArtistic_gen.pth"
    torch.save({"model": sd, "opt": {"state": {}, "param_groups": []}} if layout == "model_opt" else sd, f)
    got = render._load_pth(str(f))
    assert list(got.keys()) == list(sd.keys())
    assert all(torch.equal(got[k], sd[k]) for k in sd)
    # and through the packer: identical blobs from the file and from the in-memory dict
    from vsdeoldify_amd.deoldify_net import DeoldifyGenerator
    a, b = DeoldifyGenerator(got, "deep"), DeoldifyGenerator(synth_state_dict("deep", 9), "deep")
    assert a.blob == b.blob


def test_offline_converter_round_trip(tmp_path):
    """tools/convert_weights.py: .pth -> .havc (packed blob + offset table); loading it yields the same blob and the same plans,
    and ModelImageRender's loader prefers it when it is not older than the .pth."""
    import subprocess
    import sys
    import os
    from vsdeoldify_amd.deoldify_net import DeoldifyGenerator
    sd = {k: torch.as_tensor(np.asarray(v)) for k, v in synth_state_dict("deep", 4).items()}
    pth = tmp_path / "ColorizeArtistic_gen.pth"
    torch.save({"model": sd, "opt": {}}, pth)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.check_call([sys.executable, os.path.join(root, "tools", "convert_weights.py"), str(pth)])
    havc = tmp_path / "ColorizeArtistic_gen.havc"
    assert havc.is_file()
    a, b = DeoldifyGenerator(synth_state_dict("deep", 4), "deep"), DeoldifyGenerator.load(str(havc))
    assert a.blob == b.blob and b.arch == "deep"
    for S in (64, 80, 112):
        (oa, ba, ia, outa, na), (ob, bb, ib, outb, nb) = a.plan(S), b.plan(S)
        assert na == nb and (ia, outa) == (ib, outb) and oa.tobytes() == ob.tobytes() and ba.tobytes() == bb.tobytes()
