"""Zhang et al. colorizers (eccv16 / siggraph17), fp32 PyTorch-CPU restatement + pre/post — ORACLE ONLY.

  eccv16_forward      <- ECCVGenerator.forward      (vsdeoldify/colorization/colorizers/eccv16.py:87-98)
  siggraph17_forward  <- SIGGRAPHGenerator.forward  (vsdeoldify/colorization/colorizers/siggraph17.py:128-161);
                         the reference runs model9up..model_out twice and keeps the second result (pure waste):
                         computed once here.
  colorize_frame      <- ModelColorization.colorize_frame (vsdeoldify/colorization/__init__.py:76-95) with
                         preprocess_img / postprocess_tens (colorizers/util.py:25-55).
Network forwards are PINNED by tests/golden/zhang_*.npz (reference modules executed with seeded weights).
rgb2lab / lab2rgb restate scikit-image 0.2x (skimage/color/colorconv.py: rgb2xyz, xyz2lab, lab2xyz, xyz2rgb; D65, 2 deg
observer, float64) — PARITY UNPINNED (skimage absent from the build container).
The PIL BICUBIC resize is Pillow's own (present) in the oracle; oracle/pilresize.py restates it bit-exactly for the HIP kernel.
"""
import numpy as np
import torch
import torch.nn.functional as F

L_CENT, L_NORM, AB_NORM = 50.0, 100.0, 110.0     # base_color.py:9-11


def _seq(sd, p, x, spec):
    """spec: list of ('c', idx, stride, pad, dil) | ('r',) | ('l', slope) | ('b', idx) | ('t', idx) following nn.Sequential order."""
    for op in spec:
        if op[0] == "c":
            _, i, s, pd, d = op
            x = F.conv2d(x, sd[f"{p}.{i}.weight"], sd.get(f"{p}.{i}.bias"), s, pd, d)
        elif op[0] == "t":
            x = F.conv_transpose2d(x, sd[f"{p}.{op[1]}.weight"], sd[f"{p}.{op[1]}.bias"], 2, 1)
        elif op[0] == "r":
            x = F.relu(x)
        elif op[0] == "l":
            x = F.leaky_relu(x, op[1])
        elif op[0] == "b":
            i = op[1]
            x = F.batch_norm(x, sd[f"{p}.{i}.running_mean"], sd[f"{p}.{i}.running_var"], sd[f"{p}.{i}.weight"],
                             sd[f"{p}.{i}.bias"], False, 0.0, 1e-5)
    return x


def _block(n_conv, first_idx=0, strides=None, pad=1, dil=1, bn=True):
    spec, i = [], first_idx
    for k in range(n_conv):
        spec += [("c", i, (strides or [1] * n_conv)[k], pad, dil), ("r",)]
        i += 2
    if bn:
        spec.append(("b", i))
    return spec


def eccv16_forward(sd, input_l):
    x = (input_l - L_CENT) / L_NORM
    x = _seq(sd, "model1", x, _block(2, strides=[1, 2]))
    x = _seq(sd, "model2", x, _block(2, strides=[1, 2]))
    x = _seq(sd, "model3", x, _block(3, strides=[1, 1, 2]))
    x = _seq(sd, "model4", x, _block(3))
    x = _seq(sd, "model5", x, _block(3, pad=2, dil=2))
    x = _seq(sd, "model6", x, _block(3, pad=2, dil=2))
    x = _seq(sd, "model7", x, _block(3))
    x = _seq(sd, "model8", x, [("t", 0), ("r",), ("c", 2, 1, 1, 1), ("r",), ("c", 4, 1, 1, 1), ("r",), ("c", 6, 1, 0, 1)])
    out_reg = F.conv2d(F.softmax(x, dim=1), sd["model_out.weight"])
    return F.interpolate(out_reg, scale_factor=4, mode="bilinear") * AB_NORM


def siggraph17_forward(sd, input_a):
    z = input_a * 0
    x = torch.cat(((input_a - L_CENT) / L_NORM, z / AB_NORM, z / AB_NORM, z), dim=1)
    c1 = _seq(sd, "model1", x, _block(2))
    c2 = _seq(sd, "model2", c1[:, :, ::2, ::2], _block(2))
    c3 = _seq(sd, "model3", c2[:, :, ::2, ::2], _block(3))
    c4 = _seq(sd, "model4", c3[:, :, ::2, ::2], _block(3))
    c5 = _seq(sd, "model5", c4, _block(3, pad=2, dil=2))
    c6 = _seq(sd, "model6", c5, _block(3, pad=2, dil=2))
    c7 = _seq(sd, "model7", c6, _block(3))
    c8u = _seq(sd, "model8up", c7, [("t", 0)]) + _seq(sd, "model3short8", c3, [("c", 0, 1, 1, 1)])
    c8 = _seq(sd, "model8", c8u, [("r",), ("c", 1, 1, 1, 1), ("r",), ("c", 3, 1, 1, 1), ("r",), ("b", 5)])
    c9u = _seq(sd, "model9up", c8, [("t", 0)]) + _seq(sd, "model2short9", c2, [("c", 0, 1, 1, 1)])
    c9 = _seq(sd, "model9", c9u, [("r",), ("c", 1, 1, 1, 1), ("r",), ("b", 3)])
    c10u = _seq(sd, "model10up", c9, [("t", 0)]) + _seq(sd, "model1short10", c1, [("c", 0, 1, 1, 1)])
    c10 = _seq(sd, "model10", c10u, [("r",), ("c", 1, 1, 1, 1), ("l", 0.2)])
    return torch.tanh(F.conv2d(c10, sd["model_out.0.weight"], sd["model_out.0.bias"])) * AB_NORM


# ---- scikit-image colour conversions (float64) ------------------------------------------------------------
XYZ_FROM_RGB = np.array([[0.412453, 0.357580, 0.180423], [0.212671, 0.715160, 0.072169], [0.019334, 0.119193, 0.950227]])
RGB_FROM_XYZ = np.linalg.inv(XYZ_FROM_RGB)
D65 = np.array([0.95047, 1.0, 1.08883])


def rgb2lab(rgb_u8):
    arr = np.asarray(rgb_u8).astype(np.float64) / 255.0
    mask = arr > 0.04045
    arr = np.where(mask, np.power((arr + 0.055) / 1.055, 2.4), arr / 12.92)
    xyz = arr @ XYZ_FROM_RGB.T
    xyz = xyz / D65
    mask = xyz > 0.008856
    xyz = np.where(mask, np.cbrt(xyz), 7.787 * xyz + 16.0 / 116.0)
    x, y, z = xyz[..., 0], xyz[..., 1], xyz[..., 2]
    return np.stack([116.0 * y - 16.0, 500.0 * (x - y), 200.0 * (y - z)], -1)


def lab2rgb(lab):
    lab = np.asarray(lab, np.float64)
    L, a, b = lab[..., 0], lab[..., 1], lab[..., 2]
    y = (L + 16.0) / 116.0
    x = a / 500.0 + y
    z = y - b / 200.0
    z = np.where(z < 0, 0.0, z)                       # skimage clips negative z (warning filtered at __init__.py:74)
    out = np.stack([x, y, z], -1)
    mask = out > 0.2068966
    out = np.where(mask, np.power(out, 3.0), (out - 16.0 / 116.0) / 7.787)
    out = out * D65
    arr = out @ RGB_FROM_XYZ.T
    mask = arr > 0.0031308
    arr = np.where(mask, 1.055 * np.power(np.where(mask, arr, 1.0), 1 / 2.4) - 0.055, arr * 12.92)
    return np.clip(arr, 0, 1)


def colorize_frame(sd, model, frame_u8):
    """ModelColorization.colorize_frame: uint8 HWC in -> uint8 HWC out (network input fixed at 256x256)."""
    from PIL import Image
    img = np.asarray(frame_u8)
    if img.ndim == 2:
        img = np.tile(img[:, :, None], 3)
    rs = np.asarray(Image.fromarray(img).resize((256, 256), resample=3))          # util.py:21-22 BICUBIC
    l_orig = torch.Tensor(rgb2lab(img)[:, :, 0])[None, None]
    l_rs = torch.Tensor(rgb2lab(rs)[:, :, 0])[None, None]
    with torch.no_grad():
        ab = (eccv16_forward if model == "eccv16" else siggraph17_forward)(sd, l_rs)
        if l_orig.shape[2:] != ab.shape[2:]:
            ab = F.interpolate(ab, size=l_orig.shape[2:], mode="bilinear")
    lab = torch.cat((l_orig, ab), dim=1).numpy()[0].transpose(1, 2, 0)
    return np.uint8(np.clip(lab2rgb(lab) * 255, 0, 255))
