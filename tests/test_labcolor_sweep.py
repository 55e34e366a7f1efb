"""CPU: bound the UNPINNED restatement of scikit-image's rgb2lab / lab2rgb (oracle/zhang.py; skimage is absent from the build container) by sweeping
ALL 2^24 sRGB inputs against the float64 DEFINITION of the same transform (the cv2 analogue is tests/test_cvcolor_sweep.py).
Every byte of a Zhang (colorizers/util.py:25-55), DDColor and ColorMNet (colormnet_utils.py:185-197) frame passes through this pair: L of the source
in, Lab -> RGB out.  The library's kernels (csrc/zhang.hip) are checked against oracle/zhang.py on the GPU (tests/test_zhang.py), so a bound on
oracle-vs-definition is a bound on how far library AND oracle could be from a real skimage build: skimage hard-codes ROUNDED constants (the
0.412453 ... matrix, 0.008856, 7.787, the D65 white 0.95047 / 1.08883), the definition below derives them from first principles.

Definition (IEC 61966-2-1, CIE 15): sRGB EOTF; RGB -> XYZ matrix from the primaries (0.64, 0.33), (0.30, 0.60), (0.15, 0.06) and the white point
D65 = (0.3127, 0.3290); f(t) = cbrt(t) if t > (6/29)^3 else t / (3 (6/29)^2) + 4/29; L = 116 f(Y/Yn) - 16, a = 500 (f(X/Xn) - f(Y/Yn)), b = 200 (f(Y/Yn) - f(Z/Zn))."""
import numpy as np

from oracle import zhang


def _definition_matrix():
    xy = np.array([[0.64, 0.33], [0.30, 0.60], [0.15, 0.06]])
    P = np.stack([xy[:, 0] / xy[:, 1], np.ones(3), (1 - xy[:, 0] - xy[:, 1]) / xy[:, 1]])          # columns: XYZ of the primaries at Y = 1
    wx, wy = 0.3127, 0.3290
    white = np.array([wx / wy, 1.0, (1 - wx - wy) / wy])
    S = np.linalg.solve(P, white)
    return P * S[None, :], white


def _lab_definition(rgb_u8):
    M, white = _definition_matrix()
    c = rgb_u8.astype(np.float64) / 255.0
    lin = np.where(c > 0.04045, ((c + 0.055) / 1.055) ** 2.4, c / 12.92)
    t = (lin @ M.T) / white
    d = 6.0 / 29.0
    f = np.where(t > d ** 3, np.cbrt(t), t / (3 * d * d) + 4.0 / 29.0)
    return np.stack([116.0 * f[:, 1] - 16.0, 500.0 * (f[:, 0] - f[:, 1]), 200.0 * (f[:, 1] - f[:, 2])], -1)


def _all_triples_by_first(k):
    b, c = np.meshgrid(np.arange(256, dtype=np.uint8), np.arange(256, dtype=np.uint8), indexing="ij")
    return np.stack([np.full_like(b, k), b, c], -1).reshape(-1, 3)


def test_rgb2lab_restatement_against_the_cie_definition_and_round_trip_over_all_inputs():
    mx = np.zeros(3)
    rt_round_bad = rt_trunc_bad = n = 0
    worst_gray = 0.0
    for k in range(256):
        t = _all_triples_by_first(k)
        lab = zhang.rgb2lab(t)
        mx = np.maximum(mx, np.abs(lab - _lab_definition(t)).max(0))
        back = zhang.lab2rgb(lab)                                         # float64 in [0, 1]
        rt_round_bad += int((np.floor(back * 255.0 + 0.5).astype(np.int32) != t).sum())
        rt_trunc_bad += int(((back * 255.0).astype(np.int32) != t).sum())     # the reference's cast: np.uint8(np.clip(x * 255, 0, 255)) TRUNCATES
        g = t[(t[:, 1] == k) & (t[:, 2] == k)]
        if len(g):
            worst_gray = max(worst_gray, float(np.abs(zhang.rgb2lab(g)[:, 1:]).max()))
        n += len(t)
    print(f"rgb2lab (skimage's rounded constants) vs the CIE / IEC definition over all {n} inputs: max |dL| {mx[0]:.4f}, |da| {mx[1]:.4f}, |db| {mx[2]:.4f}")
    print(f"rgb -> lab -> rgb: bytes changed with round-to-nearest {rt_round_bad / (3 * n):.2e}, with the reference's truncating cast {rt_trunc_bad / (3 * n):.4f}")
    print(f"largest |a|, |b| of a gray input: {worst_gray:.2e}")
    assert n == 1 << 24
    # the rounded constants move L by < 0.01 and a / b by < 0.05 units (CIEDE2000 << 0.1): a skimage build with other constants of the same precision
    # cannot be further away than that
    assert mx[0] < 0.01 and mx[1] < 0.05 and mx[2] < 0.05, mx
    assert rt_round_bad == 0                                              # the pair is an exact inverse at 8 bits
    # with the truncating cast a float64 round trip lands just below the integer on a fraction of the bytes: the reason an unchanged-colour frame
    # is NOT the identity in the reference either; both sides of every comparison share it
    assert rt_trunc_bad / (3 * n) < 0.6                                  # measured 0.506: x * 255 sits a few ulp below the integer on half the bytes
    assert worst_gray < 1e-2                                              # skimage matrix rows do not sum exactly to the white point: gray has |ab| up to 5e-3, not 0


def test_lab2rgb_clips_negative_z_and_out_of_gamut_like_skimage():
    lab = np.array([[50.0, 0.0, 300.0], [100.0, 120.0, -120.0], [0.0, 0.0, 0.0], [100.0, 0.0, 0.0]])
    rgb = zhang.lab2rgb(lab)
    assert rgb.min() >= 0.0 and rgb.max() <= 1.0 and np.isfinite(rgb).all()
    assert np.allclose(rgb[2], 0.0, atol=1e-9) and np.allclose(rgb[3], 1.0, atol=2e-3)
