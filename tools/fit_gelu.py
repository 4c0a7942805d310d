"""Fit of the GELU approximation in csrc/common.h: gelu(x) = x * sigmoid(p(x)), p odd of degree 5, equal-ripple-ish
(iteratively re-weighted least squares) against the exact x * Phi(x) on [-8, 8].  Prints the coefficients, their
-log2(e) multiples (the kernel evaluates e^(-p) as exp2) and the maximum absolute error on [-20, 20] with the clamp."""
import numpy as np
from scipy.optimize import least_squares
from scipy.special import ndtr

x = np.linspace(-8, 8, 16001)
gel = x * ndtr(x)


def model(c, x):
    x2 = x * x
    return x / (1 + np.exp(-x * (c[0] + x2 * (c[1] + x2 * c[2]))))


c, w = np.array([1.5957691, 0.0713548, 0.0]), np.ones_like(x)
for _ in range(60):
    c = least_squares(lambda c: (model(c, x) - gel) * w, c, xtol=1e-15, ftol=1e-15).x
    e = np.abs(model(c, x) - gel)
    w = w * (1 + 3 * e / e.max())
print("p(x) = x (c0 + c1 x^2 + c2 x^4):", c, " max |err| on [-8, 8]:", e.max())
print("-log2(e) * c:", -1.4426950408889634 * c)
xx = np.linspace(-20, 20, 40001)
xc = np.clip(xx, -8, 8)
print("max |err| on [-20, 20] with |x| clamped to 8 inside p:",
      np.abs(xx / (1 + np.exp(-xc * (c[0] + xc ** 2 * (c[1] + xc ** 2 * c[2])))) - xx * ndtr(xx)).max())
