"""Coefficients of `binom_tail_quantile` (csrc/m3d_binom.h): z(p) - w as a degree-12 polynomial in
u = (2 / w - (rl + rh)) / (rh - rl), w = sqrt(-2 ln p), for p in [2^-64, 1/2] (z = upper-tail normal quantile).
Chebyshev least squares on a dense grid, converted to the monomial basis; prints the float32 Horner error.
The quantile is only the STARTING POINT of an exact search, so nothing here affects results."""
import numpy as np
from numpy.polynomial import chebyshev as C
from scipy.special import ndtri

lnp = -np.linspace(np.log(2.0), 64 * np.log(2.0), 400001)
w = np.sqrt(-2 * lnp)
z = -ndtri(np.exp(lnp))
r = 1 / w
rl, rh = r.min(), r.max()
u = (2 * r - (rl + rh)) / (rh - rl)
cheb = C.chebfit(u, z - w, 12)
mono = C.cheb2poly(cheb)
a, b = 2 / (rh - rl), -(rl + rh) / (rh - rl)
print(f"u = {a!r} / w + {b!r}")
print("monomial coefficients, highest first:")
print(", ".join(f"{c:.9g}f" for c in mono[::-1]))
w32 = w.astype(np.float32)
u32 = np.float32(a) / w32 + np.float32(b)
acc = np.full_like(u32, np.float32(mono[-1]))
for c in mono[-2::-1]:
    acc = acc * u32 + np.float32(c)
print("max |error| float64 fit:", np.abs(C.chebval(u, cheb) - (z - w)).max(),
      " float32 Horner:", np.abs((acc + w32).astype(np.float64) - z).max())
