#!/usr/bin/env python3
"""Near-minimax polynomial coefficients for the device exp() and cos() kernels.

Remez exchange (in mpmath, 40 digits) on the REMAINDER of a few exact leading Taylor terms, so
that the leading coefficients stay the inline constants 1, 1, 1/2 (VOP3 takes no FP64 literal:
every other coefficient costs an SGPR pair).

  exp(r)  = 1 + r + r^2/2 + r^3 g(r),      |r| <= ln2/2      (exp_nonpos / exp_any / exp_k)
  2^f     = 1 + f g(f),                    |f| <= 1/2        (exp2_nonpos / exp2_any)
  cos(w)  = 1 - s/2 + s^2 g(s), s = w^2,   |w| <= pi/2       (cos_2pi_x3)

Prints the coefficients (highest power first, as the Horner chains use them) and the maximum
error of the float64 Horner evaluation against mpmath on a dense grid.
usage: tools/minimax_fit.py
"""
import mpmath as mp
import numpy as np

mp.mp.dps = 40


def remez(h, basis, nb, a, b, weight, iters=40):
    """minimise max |weight(x) (h(x) - sum_j c_j basis(j, x))| over [a, b] (nb coefficients)."""
    n = nb + 1
    xs = [mp.mpf(a + b) / 2 + mp.mpf(b - a) / 2 * mp.cos(mp.pi * (n - 1 - i) / (n - 1))
          for i in range(n)]
    xs = [x if abs(x) > mp.mpf(b - a) * 1e-6 else mp.mpf(b - a) * 1e-3 for x in xs]
    coef = None
    for _ in range(iters):
        A = mp.matrix(n, n)
        rhs = mp.matrix(n, 1)
        for i, x in enumerate(xs):
            w = weight(x)
            for j in range(nb):
                A[i, j] = w * basis(j, x)
            A[i, nb] = (-1) ** i
            rhs[i] = w * h(x)
        sol = mp.lu_solve(A, rhs)
        coef = [sol[j] for j in range(nb)]
        err = lambda x: weight(x) * (h(x) - sum(c * basis(j, x) for j, c in enumerate(coef)))
        grid = [mp.mpf(a) + mp.mpf(b - a) * k / 6000 for k in range(6001)]
        vals = [err(x) for x in grid]
        # extrema of the error between its sign changes (zeros of even order do not split)
        ext, k0 = [], 0
        for k in range(1, len(grid) + 1):
            if k == len(grid) or vals[k] * vals[k0] < 0:
                kk = max(range(k0, k), key=lambda t: abs(vals[t]))
                ext.append(grid[kk])
                k0 = k
        if len(ext) != n:
            # keep the n largest alternating extrema if the scan found more; give up if fewer
            if len(ext) < n:
                break
            ext = sorted(sorted(ext, key=lambda x: -abs(err(x)))[:n])
        moved = max(abs(x - y) for x, y in zip(ext, xs))
        xs = ext
        if moved < mp.mpf(b - a) / 6000:
            break
    return [float(c) for c in coef]


def horner(coefs_high_first, x):
    p = np.full_like(x, coefs_high_first[0])
    for c in coefs_high_first[1:]:
        p = p * x + c
    return p


def fit_exp(deg):
    a = float(mp.log(2) / 2) * 1.0000001
    h = lambda r: mp.exp(r) - 1 - r - r * r / 2
    c = remez(h, lambda j, r: r ** (j + 3), deg - 2, -a, a, lambda r: mp.exp(-r))
    full = c[::-1] + [0.5, 1.0, 1.0]
    x = np.linspace(-a, a, 200001)
    ref = np.array([float(mp.exp(mp.mpf(float(v)))) for v in x[::50]])
    got = horner(full, x[::50])
    return full, float(np.max(np.abs(got / ref - 1.0)))


def fit_exp2(deg):
    """2^f = 1 + f (c1 + c2 f + ...), |f| <= 1/2 (the Gaussians of the burst factor are
    evaluated in base 2: their exponent constants carry the log2(e))."""
    a = 0.5 * 1.0000001
    h = lambda f: mp.mpf(2) ** f - 1
    c = remez(h, lambda j, f: f ** (j + 1), deg, -a, a, lambda f: mp.mpf(2) ** (-f))
    full = c[::-1] + [1.0]
    x = np.linspace(-a, a, 200001)[::50]
    ref = np.array([float(mp.mpf(2) ** mp.mpf(float(v))) for v in x])
    got = horner(full, x)
    return full, float(np.max(np.abs(got / ref - 1.0)))


def fit_cos(terms):
    """`terms` coefficients of g(s) -> polynomial in w of degree 2 * (terms + 1)."""
    smax = float((mp.pi / 2) ** 2) * 1.0000001
    h = lambda s: mp.cos(mp.sqrt(s)) - 1 + s / 2
    c = remez(h, lambda j, s: s ** (j + 2), terms, 0.0, smax, lambda s: mp.mpf(1))
    full = c[::-1] + [-0.5, 1.0]
    x = np.linspace(-np.pi / 2, np.pi / 2, 4001)
    ref = np.array([float(mp.cos(mp.mpf(float(v)))) for v in x])
    got = horner(full, x * x)
    return full, float(np.max(np.abs(got - ref)))


if __name__ == "__main__":
    for deg in (8, 9, 10):
        c, e = fit_exp(deg)
        print("exp degree %d: max rel err %.2e" % (deg, e))
        print("   ", ", ".join("%.17e" % v for v in c))
    for deg in (9, 10, 11):
        c, e = fit_exp2(deg)
        print("exp2 degree %d: max rel err %.2e" % (deg, e))
        print("   ", ", ".join("%.17e" % v for v in c))
    for terms in (5, 6, 7):
        c, e = fit_cos(terms)
        print("cos degree %d in w: max abs err %.2e" % (2 * (terms + 1), e))
        print("   ", ", ".join("%.17e" % v for v in c))
