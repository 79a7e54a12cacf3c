#!/usr/bin/env python3
"""Derive the fp32 polynomial coefficients of the c2d canonical math functions.

TEST/SPEC INFRASTRUCTURE (not shipped in the product path): this script only
documents where the constants in oracle/c2d_oracle.c and csrc/c2d_math.hpp come
from.  It fits, in float64 on Chebyshev nodes with relative-error weighting,

  log1p(f)            = f + f^2 * L(f)         f in [-1/3, 1/3]   (degree-8 L)
  sin(r)              = r + r^3 * S(r^2)       r in [-pi/4, pi/4] (degree-3 S)
  cos(r)              = 1 + r^2 * C(r^2)       r in [-pi/4, pi/4] (degree-3 C... see below)
  sin(pi/2 * x)       = x * P(x^2)             x in [0, 1]        (degree-5 P)
  cos(pi/2 * x)       = Q(x^2)                 x in [0, 1]        (degree-5 Q)

rounds every coefficient to fp32 and reports the max error of the *fp32 Horner
evaluation* against float64 libm, in ulps of the result.
"""
import numpy as np

def cheb_nodes(a, b, n):
    k = np.arange(n)
    x = np.cos(np.pi * (k + 0.5) / n)
    return 0.5 * (a + b) + 0.5 * (b - a) * x

def fit(fun, a, b, deg, weight=None, n=4000):
    x = cheb_nodes(a, b, n)
    y = fun(x)
    w = np.ones_like(x) if weight is None else weight(x)
    V = np.vander(x, deg + 1, increasing=True)
    c, *_ = np.linalg.lstsq(V * w[:, None], y * w, rcond=None)
    return c

def hexf(c):
    return [float(np.float32(v)).hex() for v in c]

def show(name, c):
    c32 = np.asarray(c, dtype=np.float32)
    print(name, "=", ", ".join("%sf /* %.9g */" % (float(v).hex(), v) for v in c32))
    return c32

def horner32(c32, x32):
    # fp32 fma Horner, emulated in float64 with a rounding to fp32 after each fma
    acc = np.full_like(x32, c32[-1], dtype=np.float32)
    for k in range(len(c32) - 2, -1, -1):
        acc = (acc.astype(np.float64) * x32.astype(np.float64) + np.float64(c32[k])).astype(np.float32)
    return acc

def fma32(a, b, c):
    return (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(np.float32)

def ulps(got32, ref64):
    ref32 = ref64.astype(np.float32)
    u = np.spacing(np.abs(ref32)).astype(np.float64)
    return np.max(np.abs(got32.astype(np.float64) - ref64) / u)

rng = np.random.default_rng(1)

# ---- log1p(f) = f + f^2 L(f)
L = fit(lambda f: np.where(f == 0, -0.5, (np.log1p(f) - f) / np.where(f == 0, 1, f * f)), -1/3, 1/3, 8)
L32 = show("LOG_L", L)
f = rng.uniform(-1/3, 1/3, 2_000_000).astype(np.float32)
q = horner32(L32, f)
r = fma32(f * f, q, f)   # note: f*f rounded to fp32 first
r = fma32((f * f).astype(np.float32), q, f)
print("  log1p max ulp:", ulps(r, np.log1p(f.astype(np.float64))))

# ---- sin/cos on [-pi/4, pi/4]
S = fit(lambda r: np.where(r == 0, -1/6, (np.sin(r) - r) / np.where(r == 0, 1, r**3)), 1e-9, (np.pi/4)**1, 3,
        n=4000)
# fit in z = r^2
def fit_even(fun_z, zmax, deg):
    z = cheb_nodes(0.0, zmax, 4000)
    V = np.vander(z, deg + 1, increasing=True)
    c, *_ = np.linalg.lstsq(V, fun_z(z), rcond=None)
    return c
zmax = (np.pi / 4) ** 2 * 1.02
S = fit_even(lambda z: (np.sin(np.sqrt(z)) - np.sqrt(z)) / (np.sqrt(z) * z), zmax, 3)
C = fit_even(lambda z: (np.cos(np.sqrt(z)) - 1.0) / z, zmax, 4)
S32 = show("SIN_S", S)
C32 = show("COS_C", C)
r = rng.uniform(-np.pi/4, np.pi/4, 2_000_000).astype(np.float32)
z = (r * r).astype(np.float32)
sp = horner32(S32, z)
sn = fma32((z * r).astype(np.float32), sp, r)
cp = horner32(C32, z)
cs = fma32(z, cp, np.float32(1.0) + np.zeros_like(z))
print("  sin max ulp:", ulps(sn, np.sin(r.astype(np.float64))))
print("  cos max ulp:", ulps(cs, np.cos(r.astype(np.float64))))

# ---- sin(pi/2 x), cos(pi/2 x) on x in [0, 1] after folding to [0, 1/2]:
# the device folds x>1/2 to 1-x and swaps, so the fit range is [0, 1/2].
zmax = 0.25 * 1.02
P = fit_even(lambda z: np.sin(np.pi/2 * np.sqrt(z)) / np.sqrt(z), zmax, 4)
Q = fit_even(lambda z: np.cos(np.pi/2 * np.sqrt(z)), zmax, 4)
P32 = show("SINPI2_P", P)
Q32 = show("COSPI2_Q", Q)
x = rng.uniform(1e-9, 0.5, 2_000_000).astype(np.float32)
z = (x * x).astype(np.float32)
sn = (horner32(P32, z).astype(np.float32) * x).astype(np.float32)
cs = horner32(Q32, z)
print("  sinpi2 max ulp:", ulps(sn, np.sin(np.pi/2 * x.astype(np.float64))))
print("  cospi2 max ulp:", ulps(cs, np.cos(np.pi/2 * x.astype(np.float64))))

# Cody-Waite split of pi/2 (three fp32 terms, the first two with trailing zero bits)
def split(v, bits):
    import math
    m, e = math.frexp(v)
    m = math.floor(m * (1 << bits)) / (1 << bits)
    return math.ldexp(m, e)
p = np.pi / 2
h = split(p, 12); m = split(p - h, 12); l = float(np.float32(p - h - m))
print("PIO2_HI = %sf  PIO2_MID = %sf  PIO2_LO = %sf" % (float(np.float32(h)).hex(), float(np.float32(m)).hex(), float(l).hex()))
print("TWO_OVER_PI = %sf" % float(np.float32(2/np.pi)).hex())
print("LN2 = %sf" % float(np.float32(np.log(2))).hex())
import math
print("LOG40_DOUBLE (log(1.0/(double)0.025f)) = %s" % (math.log(1.0 / float(np.float32(0.025)))).hex())
