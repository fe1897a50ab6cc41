"""Coefficients and error table of the packed-f16 GELU of the decoder's forward kernel (ggd_mlp.hip: gelu_h2x4).

The hidden layers' weights and biases are halved in the forward image, so the accumulators hold y = z / 2 and

    gelu(z) = z / 2 + |z| (1/2 - Q(|z|)) = y + a s(v),   a = |y|,  v = min(a, 2) - 1 in [-1, 1],  s(v) ~ 1 - 2 Q(2 (v + 1))

with Q = 1 - Phi.  s is a degree-N polynomial in the monomial basis of v (sum |s_k| ~ 2.4, all partial sums O(1): a Horner
chain in f16 loses nothing to cancellation, unlike Phi(x) = 0.5 + x P(x^2) whose coefficients alternate up to 13 on the
same range), fitted minimax (iteratively reweighted least squares on Chebyshev nodes, error weighted by |z| as it enters the
result) under the constraint s(1) = 1: beyond |z| = 4 the result is z (z > 0) or 0 (z < 0) up to the rounding of the
coefficients, whatever |z|.  The script evaluates the chain exactly as the kernel does (every fma rounded once to f16) on
EVERY f16 value of [-64, 64] and prints the error against float64.
"""
import sys
import numpy as np
from scipy.special import erf

h = np.float16


def fma(a, b, c):
    return (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(h)


def gelu(x):
    return 0.5 * x * (1 + erf(x / np.sqrt(2)))


def Q(u):
    return 0.5 * (1 - erf(u / np.sqrt(2)))


def fit(n, iters=60):
    """s_0 .. s_n of s(v) = 1 - 2 q(v), q(v) = sum_{k >= 1} c_k (v^k - 1) ~ Q(2 (v + 1)) (so q(1) = 0)."""
    m = 4000
    v = np.cos(np.pi * (np.arange(m) + 0.5) / m)
    u = (v + 1) * 2
    A = np.stack([v ** k - 1 for k in range(1, n + 1)], 1)
    wgt = np.maximum(u, 0.25)           # the error of q enters gelu multiplied by |z|
    w = np.ones(m)
    for _ in range(iters):
        c, *_ = np.linalg.lstsq(A * (w * wgt)[:, None], Q(u) * w * wgt, rcond=None)
        e = np.abs(A @ c - Q(u)) * wgt
        w = w * (1 + 3 * e / e.max())
        w /= w.mean()
    q = np.concatenate([[-c.sum()], c])
    return np.concatenate([[1 - 2 * q[0]], -2 * q[1:]])


def gelu_kernel(y, s):
    """The kernel's instruction sequence on y = z / 2 (f16)."""
    C = lambda val: np.full_like(y, h(val))
    a = np.abs(y)
    v = (np.minimum(a, h(2)).astype(np.float64) - 1).astype(h)      # v_pk_min_f16, v_pk_add_f16 (exact)
    sh = s.astype(h)
    p = fma(v, C(sh[-1]), C(sh[-2]))
    for k in range(len(sh) - 3, -1, -1):
        p = fma(p, v, C(sh[k]))
    return fma(a, p, y)


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 6   # the kernel's degree
    s = fit(n)
    allh = np.arange(0, 0x7c00, dtype=np.uint16).view(h)
    allh = allh[np.abs(allh) <= 32]
    y = np.concatenate([allh, -allh]).astype(h)
    z = 2.0 * y.astype(np.float64)
    err = np.abs(gelu_kernel(y, s).astype(np.float64) - gelu(z))
    rnd = np.abs(gelu(z).astype(h).astype(np.float64) - gelu(z))     # what rounding the exact value to f16 costs
    print(f"degree {n}: s_0 .. s_{n} (rounded to f16 by the kernel), sum = {s.sum():.9f}:")
    print("  " + ", ".join(f"{c:.9e}" for c in s))
    for lo, hi in ((-64, -4), (-4, -2), (-2, -0.5), (-0.5, 0.5), (0.5, 2), (2, 4), (4, 64)):
        m = (z >= lo) & (z < hi)
        print(f"  z in [{lo:5}, {hi:4}): max |err| {err[m].max():.2e}   (f16 rounding of the exact value alone: {rnd[m].max():.2e};"
              f" a bf16 result: {np.max(np.abs(gelu(z[m])) * 2.0 ** -9):.2e})")
    m = np.abs(z) < 8
    print(f"  mean |err| over the f16 values of |z| < 8: {err[m].mean():.2e}; relative error for 1e-4 < |z| <= 0.5: "
          f"{np.max(err[(np.abs(z) <= 0.5) & (np.abs(z) > 1e-4)] / np.abs(gelu(z[(np.abs(z) <= 0.5) & (np.abs(z) > 1e-4)]))):.2e}")


if __name__ == "__main__":
    main()
