"""Coefficients of exp2_fast's polynomial (vag_device.h): 2^f = 1 + f q(f) on [-1/2, 1/2] with q of degree N - 1 from a Remez
exchange in 60-digit arithmetic (mpmath), rounded to double, and the error of the ROUNDED Horner form evaluated in double.
The Taylor form needs degree 12 for 1.7e-16; the minimax form reaches the same with degree 10 (two v_fma_f64 less per call)."""
import sys

import mpmath as mp
import numpy as np

mp.mp.dps = 60


def g(x):  # (2^x - 1) / x
    x = mp.mpf(x)
    if x == 0:
        return mp.log(2)
    return (mp.power(2, x) - 1) / x


def remez(n, a=-0.5, b=0.5, iters=30):
    """minimax polynomial of degree n for g = (2^x - 1) / x"""
    a, b = mp.mpf(a), mp.mpf(b)
    w = lambda x: mp.mpf(1)  # unweighted in q: the relative error of 2^x is then <= |x| dq / 2^x <= 0.71 dq
    xs = [(a + b) / 2 + (b - a) / 2 * mp.cos(mp.pi * (n + 1 - i) / (n + 1)) for i in range(n + 2)]
    for _ in range(iters):
        A = mp.matrix(n + 2, n + 2)
        rhs = mp.matrix(n + 2, 1)
        for i, x in enumerate(xs):
            for j in range(n + 1):
                A[i, j] = x ** j
            wx = w(x)
            A[i, n + 1] = (-1) ** i / wx
            rhs[i] = g(x)
        sol = mp.lu_solve(A, rhs)
        c = [sol[j] for j in range(n + 1)]
        err = lambda x: (mp.polyval(c[::-1], x) - g(x)) * w(x)
        # new extrema: scan finely, pick one extremum per sign-run
        grid = [a + (b - a) * mp.mpf(i) / 4000 for i in range(4001)]
        vals = [err(x) for x in grid]
        ext = []
        i = 0
        while i < len(grid):
            j = i
            s = mp.sign(vals[i])
            best = i
            while j < len(grid) and (mp.sign(vals[j]) == s or vals[j] == 0):
                if abs(vals[j]) > abs(vals[best]):
                    best = j
                j += 1
            ext.append(grid[best])
            i = j
        if len(ext) != n + 2:
            break
        xs = ext
    return c, max(abs(v) for v in vals)


for n in (int(sys.argv[1]),) if len(sys.argv) > 1 else (9, 10):
    c, e = remez(n)
    cd = [float(x) for x in c]
    # double-precision Horner of 1 + f * q(f), against the exact value
    f = np.linspace(-0.5, 0.5, 200001)
    p = np.full_like(f, cd[-1])
    for k in range(n - 1, -1, -1):
        p = p * f + cd[k]
    val = p * f + 1.0
    exact = np.array([float(mp.power(2, mp.mpf(x))) for x in f[::200]])
    rel = np.max(np.abs(val[::200] - exact) / exact)
    print(f"degree of q = {n} (2^f of degree {n + 1}): minimax rel err {mp.nstr(e, 3)}, double Horner max rel err {rel:.2e}")
    print("  coefficients q0..q%d:" % n, ", ".join(repr(x) for x in cd))
