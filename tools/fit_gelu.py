"""Fits the polynomial used by gelu_erf_f (csrc/common.h): Phi(-a) = 2^Q(a), a = |x|, Q of degree 5, minimising the
maximum absolute error of gelu(x) = 0.5 x + |x| (0.5 - 2^Q(|x|)) against x * Phi(x) (scipy.special.ndtr, float64), then
reports the error of the float32 evaluation over |x| <= 1e5.  Run: python tools/fit_gelu.py"""
import numpy as np
from scipy.optimize import least_squares
from scipy.special import log_ndtr, ndtr

AMAX, DEG = 6.0, 5


def approx(c, x):
    a = np.abs(x)
    return a * (0.5 - np.exp2(np.polyval(c, np.minimum(a, AMAX)))) + 0.5 * x


def main():
    xx = np.linspace(-AMAX - 0.5, AMAX + 0.5, 8001)
    res = lambda c: approx(c, xx) - xx * ndtr(xx)
    aa = np.linspace(0, AMAX, 2000)
    c = np.polyfit(aa, log_ndtr(-aa) / np.log(2), DEG, w=np.exp(log_ndtr(-aa)) * aa + 1e-3)
    w = np.ones_like(xx)
    for _ in range(80):  # iteratively re-weighted least squares -> (near) minimax
        c = least_squares(lambda c: w * res(c), c, xtol=1e-15, ftol=1e-15).x
        e = np.abs(res(c))
        w = w * (1 + 3 * e / e.max())
        w /= w.mean()
    print("coefficients (highest degree first):", [float(k) for k in c])
    a = np.concatenate([np.linspace(0, 12, 100001), np.logspace(1, 5, 2000)]).astype(np.float32)
    x = np.concatenate([-a, a])
    q = np.float32(c[0])
    for k in c[1:]:
        q = q * np.abs(x) + np.float32(k)
    with np.errstate(all="ignore"):
        t = np.exp2(q).astype(np.float32)
    out = np.abs(x) * (np.float32(0.5) - t) + np.float32(0.5) * x
    ref = x.astype(np.float64) * ndtr(x.astype(np.float64))
    err = np.abs(out - ref)
    print(f"float32 evaluation, |x| <= 1e5 (no clamp): max abs err {err.max():.3e} at x = {x[err.argmax()]:.4f}; all finite: {np.isfinite(out).all()}")


if __name__ == "__main__":
    main()
