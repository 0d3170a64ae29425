"""Generate fp64 polynomial coefficients for the device sin/cos kernels
(drake_ddp_amd/csrc/fastmath.hpp).  Near-minimax via Chebyshev-node
interpolation in 60-digit arithmetic, coefficients rounded to double, and the
resulting double-precision evaluation checked against mpmath.

  sin(r) = r + r^3 * S(r^2),   cos(r) = 1 - r^2/2 + r^4 * C(r^2),   |r| <= pi/2 (+ margin)
"""
import mpmath as mp
import numpy as np

mp.mp.dps = 60
R = mp.pi / 2 * mp.mpf("1.001")
SMAX = R * R


def cheb_fit(f, deg, a, b):
    n = deg + 1
    nodes = [(a + b) / 2 + (b - a) / 2 * mp.cos(mp.pi * (2 * k + 1) / (2 * n)) for k in range(n)]
    V = mp.matrix(n, n)
    y = mp.matrix(n, 1)
    for i, s in enumerate(nodes):
        for j in range(n):
            V[i, j] = s ** j
        y[i] = f(s)
    c = mp.lu_solve(V, y)
    return [c[j] for j in range(n)]


def S(s):
    if s == 0:
        return -mp.mpf(1) / 6
    r = mp.sqrt(s)
    return (mp.sin(r) - r) / (r * s)


def Cf(s):
    if s == 0:
        return mp.mpf(1) / 24
    r = mp.sqrt(s)
    return (mp.cos(r) - 1 + s / 2) / (s * s)


def horner(c, s):
    acc = np.full_like(s, c[-1])
    for k in range(len(c) - 2, -1, -1):
        acc = acc * s + c[k]      # numpy: no fma, slightly pessimistic
    return acc


for name, f, deg in (("S", S, 10), ("C", Cf, 10)):
    c = cheb_fit(f, deg, mp.mpf(0), SMAX)
    cd = [float(x) for x in c]
    rs = np.linspace(-float(R), float(R), 20001)
    s = rs * rs
    if name == "S":
        approx = rs + rs * s * horner(cd, s)
        exact = np.array([float(mp.sin(mp.mpf(float(r)))) for r in rs])
    else:
        approx = 1.0 - 0.5 * s + s * s * horner(cd, s)
        exact = np.array([float(mp.cos(mp.mpf(float(r)))) for r in rs])
    err = np.abs(approx - exact)
    ulp = np.spacing(np.maximum(np.abs(exact), 1e-300))
    print(f"// {name}: degree {deg} in r^2; max abs err {err.max():.3e}; max err in ulps {np.max(err/ulp):.2f} (|r|>1e-3: {np.max((err/ulp)[np.abs(rs)>1e-3]):.2f})")
    print("static constexpr double k%s[%d] = {" % (name, len(cd)))
    for x in cd:
        print(f"    {x.hex()},  // {x!r}")
    print("};")

# Cody-Waite split of pi: P1 has 33 significant bits (n*P1 exact for |n| < 2^20), P2 next 33, P3 the rest
def split(x, bits):
    m, e = mp.frexp(x)
    q = mp.floor(m * 2 ** bits) / 2 ** bits
    return mp.ldexp(q, e)

p1 = split(mp.pi, 33); r1 = mp.pi - p1
p2 = split(r1, 33); r2 = r1 - p2
p3 = float(r2)
print("static constexpr double kPi1 = %s;  // %r" % (float(p1).hex(), float(p1)))
print("static constexpr double kPi2 = %s;  // %r" % (float(p2).hex(), float(p2)))
print("static constexpr double kPi3 = %s;  // %r" % (p3.hex(), p3))
print("static constexpr double kInvPi = %s;  // %r" % (float(1 / mp.pi).hex(), float(1 / mp.pi)))
assert float(p1) == p1 and float(p2) == p2
