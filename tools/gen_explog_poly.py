"""fp64 polynomial coefficients for fast_exp / fast_log1p01 (drake_ddp_amd/csrc/fastmath.hpp).
exp(r) = 1 + r + r^2 P(r), |r| <= ln2/2;  log1p(f) = 2s + s w R(w), s = f/(2+f), w = s^2 <= 0.0296."""
import mpmath as mp
import numpy as np
mp.mp.dps = 60

def cheb_fit(f, deg, a, b):
    n = deg + 1
    nodes = [(a + b) / 2 + (b - a) / 2 * mp.cos(mp.pi * (2 * k + 1) / (2 * n)) for k in range(n)]
    V = mp.matrix(n, n); y = mp.matrix(n, 1)
    for i, s in enumerate(nodes):
        for j in range(n): V[i, j] = s ** j
        y[i] = f(s)
    c = mp.lu_solve(V, y)
    return [float(c[j]) for j in range(n)]

H = mp.log(2) / 2 * mp.mpf("1.01")
P = lambda r: (mp.mpf(1) / 2 + r / 6 + r * r / 24) if abs(r) < mp.mpf('1e-15') else (mp.exp(r) - 1 - r) / (r * r)
cp = cheb_fit(P, 10, -H, H)
rs = np.linspace(-float(H), float(H), 20001)
acc = np.full_like(rs, cp[-1])
for k in range(len(cp) - 2, -1, -1): acc = acc * rs + cp[k]
approx = 1.0 + (rs + rs * rs * acc)
exact = np.array([float(mp.exp(mp.mpf(float(r)))) for r in rs])
print("// exp: max rel err %.3e (%.2f ulp)" % (np.max(np.abs(approx - exact) / exact), np.max(np.abs(approx - exact) / np.spacing(exact))))
print("constexpr double kE[%d] = {%s};" % (len(cp), ", ".join(x.hex() for x in cp)))

WMAX = mp.mpf("0.0300")
def R(w):
    if w < mp.mpf('1e-30'): return mp.mpf(2) / 3 + 2 * w / 5
    s = mp.sqrt(w)
    return (2 * mp.atanh(s) / s - 2) / w
cr = cheb_fit(R, 7, mp.mpf(0), WMAX)
fs = np.linspace(-0.2929, 0.4143, 20001)
s = fs / (2.0 + fs); w = s * s
acc = np.full_like(w, cr[-1])
for k in range(len(cr) - 2, -1, -1): acc = acc * w + cr[k]
approx = 2.0 * s + s * w * acc
exact = np.array([float(mp.log1p(mp.mpf(float(f)))) for f in fs])
m = np.abs(exact) > 1e-300
print("// log1p: max rel err %.3e (%.2f ulp)" % (np.max(np.abs(approx - exact)[m] / np.abs(exact[m])), np.max(np.abs(approx - exact)[m] / np.spacing(np.abs(exact[m])))))
print("constexpr double kL[%d] = {%s};" % (len(cr), ", ".join(x.hex() for x in cr)))
l2 = mp.log(2)
hi = float(mp.floor(l2 * 2 ** 32) / 2 ** 32)
print("constexpr double kLn2Hi = %s, kLn2Lo = %s, kLog2e = %s, kLn2 = %s;" % (hi.hex(), float(l2 - hi).hex(), float(1 / l2).hex(), float(l2).hex()))
