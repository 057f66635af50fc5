#!/usr/bin/env python3
"""Near-minimax polynomial for atan on [0, 1]: atan(a) = a + a s p(s), s = a^2.  p interpolates
(atan(sqrt s) / sqrt s - 1) / s at the Chebyshev nodes of [0, 1] in 60-digit arithmetic, coefficients rounded to double, error of the ROUNDED polynomial measured against atan in high precision.
Prints the coefficients for flow_common.h (ft_atan).   python3 tools/minimax_atan.py [n_coefficients ...]"""
import sys
import mpmath as mp
mp.mp.dps = 60


def f(s):
    if s == 0:
        return -mp.mpf(1) / 3
    r = mp.sqrt(s)
    return (mp.atan(r) / r - 1) / s


def fit(nodes, m):
    n = len(nodes)
    A = mp.matrix(n, n); b = mp.matrix(n, 1)
    for i, x in enumerate(nodes):
        for j in range(m + 1):
            A[i, j] = x ** j
        b[i] = f(x)
    return mp.lu_solve(A, b)


for nc in (int(a) for a in sys.argv[1:] or ['20']):
    m = nc - 1
    nodes = [(1 + mp.cos(mp.pi * (2 * k + 1) / (2 * (m + 1)))) / 2 for k in range(m + 1)]
    q = fit(nodes, m)
    cd = [float(q[j]) for j in range(m + 1)]

    def atan_p(a, cd=cd):
        s = a * a
        p = mp.mpf(cd[-1])
        for c in reversed(cd[:-1]):
            p = p * s + mp.mpf(c)
        return a + a * s * p
    N = 4000
    err = max(abs(atan_p(mp.mpf(k) / N) / mp.atan(mp.mpf(k) / N) - 1) for k in range(1, N + 1))
    print(f'{nc} coefficients: max relative error of the double-rounded polynomial (exact arithmetic) on (0, 1] = {mp.nstr(err, 4)}')
    print('  coefficients of p, highest first:')
    for c in reversed(cd):
        print(f'    {c!r},')
