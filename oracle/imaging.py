"""numpy restatements of the Pillow / fastai image arithmetic on the DeOldify path — ORACLE ONLY.

Each function cites the reference line it follows; all of them are pinned bit-exactly against Pillow
itself (tests/test_oracle_imaging.py; Pillow is present in the image) and against golden vectors
produced by executing the reference's ColorizerFilter (tests/golden/filter_*.npz).
"""
import numpy as np

IMAGENET_MEAN = np.array([0.485, 0.456, 0.406], np.float32)   # fastai/vision/data.py:79
IMAGENET_STD = np.array([0.229, 0.224, 0.225], np.float32)


def pil_gray_rgb(rgb):
    """img.convert('LA').convert('RGB') (deoldify/filters.py:92-93):
    L = (19595 R + 38470 G + 7471 B + 0x8000) >> 16, replicated to 3 channels."""
    a = np.asarray(rgb).astype(np.uint32)
    L = ((19595 * a[..., 0] + 38470 * a[..., 1] + 7471 * a[..., 2] + 0x8000) >> 16).astype(np.uint8)
    return np.stack([L, L, L], -1)


def pil_blend(a, b, w):
    """PIL.Image.blend(a, b, w) (deoldify/visualize.py:129; vsslib/imfilters.py:122):
    (UINT8)((int)a + w * ((int)b - (int)a)) evaluated in float32, truncating cast."""
    a = np.asarray(a).astype(np.float32)
    b = np.asarray(b).astype(np.float32)
    return (a + np.float32(w) * (b - a)).astype(np.uint8)


def model_input(rgb_u8):
    """BaseFilter._model_process pre-processing (deoldify/filters.py:45-53): gray-replicate, pil2tensor
    float32 CHW, /255, (x-mean)/std.  Returns float32 [1,3,H,W]."""
    g = pil_gray_rgb(rgb_u8).astype(np.float32)
    x = g.transpose(2, 0, 1) / np.float32(255)
    x = (x - IMAGENET_MEAN[:, None, None]) / IMAGENET_STD[:, None, None]
    return x[None].astype(np.float32)


def model_output_u8(y):
    """pred_batch(reconstruct=True) + image2np(out*255).astype(uint8)
    (fastai/basic_train.py:358-363, fastai/vision/data.py:60-62,300, deoldify/filters.py:65-68):
    clamp(y*std+mean, 0, 1) then TRUNCATING cast of x*255.  y: float32 [3,H,W] -> uint8 [H,W,3]."""
    y = np.asarray(y, np.float32)
    d = y * IMAGENET_STD[:, None, None] + IMAGENET_MEAN[:, None, None]
    d = np.clip(d, np.float32(0), np.float32(1))
    return (d.transpose(1, 2, 0) * np.float32(255)).astype(np.uint8)


# ---- CIEDE2000 (Sharma, Wu, Dalal 2005) on sRGB uint8 images: the parity metric of BASELINE.json ----
def srgb_to_lab(rgb_u8):
    c = np.asarray(rgb_u8).astype(np.float64) / 255.0
    lin = np.where(c > 0.04045, ((c + 0.055) / 1.055) ** 2.4, c / 12.92)
    M = np.array([[0.412453, 0.357580, 0.180423], [0.212671, 0.715160, 0.072169], [0.019334, 0.119193, 0.950227]])
    xyz = lin @ M.T / np.array([0.95047, 1.0, 1.08883])
    f = np.where(xyz > 0.008856, np.cbrt(xyz), 7.787 * xyz + 16.0 / 116.0)
    return np.stack([116.0 * f[..., 1] - 16.0, 500.0 * (f[..., 0] - f[..., 1]), 200.0 * (f[..., 1] - f[..., 2])], -1)


def ciede2000(lab1, lab2):
    L1, a1, b1 = lab1[..., 0], lab1[..., 1], lab1[..., 2]
    L2, a2, b2 = lab2[..., 0], lab2[..., 1], lab2[..., 2]
    C1, C2 = np.hypot(a1, b1), np.hypot(a2, b2)
    Cb = (C1 + C2) / 2
    G = 0.5 * (1 - np.sqrt(Cb ** 7 / (Cb ** 7 + 25.0 ** 7)))
    a1p, a2p = (1 + G) * a1, (1 + G) * a2
    C1p, C2p = np.hypot(a1p, b1), np.hypot(a2p, b2)
    h1p = np.degrees(np.arctan2(b1, a1p)) % 360
    h2p = np.degrees(np.arctan2(b2, a2p)) % 360
    dLp, dCp = L2 - L1, C2p - C1p
    dh = h2p - h1p
    dh = np.where(C1p * C2p == 0, 0, np.where(dh > 180, dh - 360, np.where(dh < -180, dh + 360, dh)))
    dHp = 2 * np.sqrt(C1p * C2p) * np.sin(np.radians(dh / 2))
    Lbp, Cbp = (L1 + L2) / 2, (C1p + C2p) / 2
    hs = h1p + h2p
    hbp = np.where(C1p * C2p == 0, hs, np.where(np.abs(h1p - h2p) <= 180, hs / 2,
                                                 np.where(hs < 360, (hs + 360) / 2, (hs - 360) / 2)))
    T = (1 - 0.17 * np.cos(np.radians(hbp - 30)) + 0.24 * np.cos(np.radians(2 * hbp)) +
         0.32 * np.cos(np.radians(3 * hbp + 6)) - 0.20 * np.cos(np.radians(4 * hbp - 63)))
    dth = 30 * np.exp(-(((hbp - 275) / 25) ** 2))
    Rc = 2 * np.sqrt(Cbp ** 7 / (Cbp ** 7 + 25.0 ** 7))
    Sl = 1 + 0.015 * (Lbp - 50) ** 2 / np.sqrt(20 + (Lbp - 50) ** 2)
    Sc, Sh = 1 + 0.045 * Cbp, 1 + 0.015 * Cbp * T
    Rt = -np.sin(np.radians(2 * dth)) * Rc
    return np.sqrt((dLp / Sl) ** 2 + (dCp / Sc) ** 2 + (dHp / Sh) ** 2 + Rt * (dCp / Sc) * (dHp / Sh))


def delta_e00_images(rgb_a, rgb_b):
    """per-pixel CIEDE2000 between two uint8 sRGB images."""
    return ciede2000(srgb_to_lab(rgb_a), srgb_to_lab(rgb_b))
