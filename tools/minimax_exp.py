#!/usr/bin/env python3
"""Near-minimax polynomial for exp(r) on |r| <= ln2/2 with the two lowest coefficients pinned to 1 (so that
exp(r) = 1 + r + r^2 q(r) keeps full relative accuracy near 0): Chebyshev-node interpolation of
(exp(r) - 1 - r) / r^2 in 60-digit arithmetic, coefficients rounded to double, error of the ROUNDED polynomial
evaluated in high precision.  Prints C arrays for flow_common.h."""
import sys
import mpmath as mp
mp.mp.dps = 60
h = mp.log(2) / 2
for deg in (int(a) for a in sys.argv[1:] or ['11']):
    m = deg - 2                                 # degree of q
    nodes = [h * mp.cos(mp.pi * (2 * k + 1) / (2 * (m + 1))) for k in range(m + 1)]
    f = lambda r: (mp.exp(r) - 1 - r) / r ** 2 if r != 0 else mp.mpf(1) / 2
    A = mp.matrix(m + 1, m + 1); b = mp.matrix(m + 1, 1)
    for i, x in enumerate(nodes):
        for j in range(m + 1):
            A[i, j] = x ** j
        b[i] = f(x)
    q = mp.lu_solve(A, b)
    coef = [mp.mpf(1), mp.mpf(1)] + [q[j] for j in range(m + 1)]
    cd = [float(c) for c in coef]
    def P(r):
        s = mp.mpf(cd[-1])
        for c in reversed(cd[:-1]):
            s = s * r + mp.mpf(c)
        return s
    err = max(abs(P(h * (mp.mpf(k) / 2000)) / mp.exp(h * (mp.mpf(k) / 2000)) - 1) for k in range(-2000, 2001))
    print(f'degree {deg}: max relative error of the double-rounded polynomial (exact arithmetic) = {mp.nstr(err, 4)}')
    print('  coefficients, highest first:')
    for c in reversed(cd):
        print(f'    {c!r},')
