"""Fits the polynomial used by gelu_erf_f (csrc/ca_common.h): Phi(x) - 1/2 ~= xc * P(xc^2), xc = clamp(x, +-X).
   python tools/fit_gelu.py [X] [degree]     (prints the fp32 coefficients, lowest order first, and the max |gelu| error)"""
import sys
import numpy as np
from numpy.polynomial import chebyshev as Ch, polynomial as P
from scipy.special import erf

X = float(sys.argv[1]) if len(sys.argv) > 1 else 4.5
n = int(sys.argv[2]) if len(sys.argv) > 2 else 9
N = 8000
t = np.cos(np.pi * (np.arange(N) + 0.5) / N)
u = (t + 1) / 2 * X * X
x = np.sqrt(u)
target = 0.5 * erf(x / np.sqrt(2))
h = np.where(x > 1e-9, target / np.maximum(x, 1e-9), 1 / np.sqrt(2 * np.pi))
V = Ch.chebvander(t, n)
w = x.copy()
coef, *_ = np.linalg.lstsq(V * w[:, None], h * w, rcond=None)
for _ in range(80):  # push the least-squares fit towards equal ripple
    err = (V @ coef) * x - target
    w = w * (1 + 3 * np.abs(err) / np.abs(err).max())
    coef, *_ = np.linalg.lstsq(V * w[:, None], h * w, rcond=None)
pt = Ch.cheb2poly(coef)  # Chebyshev in t -> monomials in u (t = 2u/X^2 - 1)
pu, powk, lin = np.zeros(1), np.ones(1), np.array([-1.0, 2.0 / (X * X)])
for c in pt:
    pu = P.polyadd(pu, c * powk)
    powk = P.polymul(powk, lin)
xs = np.linspace(-8, 8, 800001)
xc = np.clip(xs, -X, X).astype(np.float32)
c32 = pu.astype(np.float32)
acc = np.full_like(xc, c32[-1])
for c in c32[-2::-1]:
    acc = acc * (xc * xc) + c
gelu = xs.astype(np.float32) * (np.float32(0.5) + xc * acc)
true = xs * 0.5 * (1 + erf(xs / np.sqrt(2)))
print("max |gelu error| on [-8, 8]: %.2e" % np.abs(gelu - true).max())
print(", ".join("%.9ef" % c for c in c32))
