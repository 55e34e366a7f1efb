"""The coefficients of gelu_erf (csrc/conv_common.h): Phi(v) = 1 / (1 + exp(-v P(min(v^2, 36)))), P of degree 3 in v^2, fitted by iteratively
re-weighted least squares to the minimax error of v Phi(v) over |v| <= 12.  Prints the coefficients and the maximum absolute error (1.17e-5)."""
import numpy as np
from scipy.optimize import least_squares
from scipy.special import ndtr

CLAMP = 36.0


def approx(c, v):
    v2 = np.minimum(v * v, CLAMP)
    u = c[-1]
    for k in c[-2::-1]:
        u = u * v2 + k
    return v / (1 + np.exp(-u * v))


def main():
    v = np.linspace(-12, 12, 48001)
    exact = v * ndtr(v)
    c = np.array([1.5957, 0.0713, 0.0, 0.0])
    w = np.ones_like(v)
    for _ in range(80):
        c = least_squares(lambda c: (approx(c, v) - exact) * w * 1e4, c, method="lm").x
        e = np.abs(approx(c, v) - exact)
        w = w * (1 + 2 * e / e.max())
        w /= w.mean()
    e = np.abs(approx(c, v) - exact)
    print("coefficients (v^0, v^2, v^4, v^6 of P):", ", ".join(f"{x:.8e}" for x in c))
    print(f"max |v Phi(v) - approximation| over [-12, 12]: {e.max():.3e} at v = {v[e.argmax()]:.3f}")
    c32 = c.astype(np.float32).astype(np.float64)
    v32 = np.linspace(-12, 12, 200001).astype(np.float32).astype(np.float64)
    print(f"with fp32 coefficients: {np.abs(approx(c32, v32) - v32 * ndtr(v32)).max():.3e}")


if __name__ == "__main__":
    main()
