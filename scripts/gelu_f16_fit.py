"""Coefficients and error table of the packed-f16 GELU of the decoder's forward kernel (ggd_mlp.hip: gelu_h2x4).

    gelu(x) = max(x, 0) - u * Q(u),   u = min(|x|, 4),   Q(u) = 1 - Phi(u) ~ q(v),  v = u / 2 - 1 in [-1, 1]

q: degree-N polynomial, Chebyshev interpolant of Q on [0, 4] converted to the monomial basis in v (sum |c_k| ~ 1.1: a
Horner chain in f16 loses nothing to cancellation, unlike Phi(x) = 0.5 + x P(x^2) whose coefficients alternate up to
13 on the same range).  The script evaluates the chain exactly as the kernel does (every fma rounded once to f16) on
EVERY f16 value of [-16, 16] and prints the error against float64.
"""
import sys
import numpy as np
from numpy.polynomial import chebyshev as Ch
from scipy.special import erf

h = np.float16
U = 4.0


def fma(a, b, c):
    return (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(h)


def gelu(x):
    return 0.5 * x * (1 + erf(x / np.sqrt(2)))


def Q(u):
    return 0.5 * (1 - erf(u / np.sqrt(2)))


def coeffs(n):
    return Ch.cheb2poly(Ch.chebinterpolate(lambda v: Q((v + 1) * U / 2), n))


def gelu_h2(x, mono):
    C = lambda v: np.full_like(x, h(v))
    u = np.minimum(np.abs(x), h(U))
    v = fma(u, C(2 / U), C(-1))
    ch = mono.astype(h)
    q = np.full_like(x, ch[-1])
    for k in range(len(ch) - 2, -1, -1):
        q = fma(q, v, C(ch[k]))
    return fma(-u, q, np.maximum(x, h(0)))


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 6   # the kernel's degree
    mono = coeffs(n)
    allh = np.arange(0, 0x7c00, dtype=np.uint16).view(h)
    allh = allh[np.abs(allh) <= 16]
    x = np.concatenate([allh, -allh]).astype(h)
    xf = x.astype(np.float64)
    err = np.abs(gelu_h2(x, mono).astype(np.float64) - gelu(xf))
    rnd = np.abs(gelu(xf).astype(h).astype(np.float64) - gelu(xf))     # what rounding the exact value to f16 costs
    print(f"degree {n}: coefficients of q(v), c0 first (rounded to f16 by the kernel):")
    print("  " + ", ".join(f"{c:.9e}" for c in mono))
    print("  as f16: " + ", ".join(f"{float(h(c))!r}" for c in mono))
    for lo, hi in ((-16, -4), (-4, -2), (-2, -0.5), (-0.5, 0.5), (0.5, 2), (2, 4), (4, 16)):
        m = (xf >= lo) & (xf < hi)
        print(f"  x in [{lo:5}, {hi:4}): max |err| {err[m].max():.2e}   (f16 rounding of the exact value alone: {rnd[m].max():.2e};"
              f" a bf16 result: {np.max(np.abs(gelu(xf[m])) * 2.0 ** -9):.2e})")
    m = (np.abs(xf) <= 0.5) & (np.abs(xf) > 1e-4)
    print(f"  relative error for 1e-4 < |x| <= 0.5: {np.max(err[m] / np.abs(gelu(xf[m]))):.2e}")


if __name__ == "__main__":
    main()
