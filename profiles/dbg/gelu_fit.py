import numpy as np
from scipy.special import erfc, erf
# GELU(v) = max(v,0) - |v| * h(x),  x = |v|/sqrt(2),  h = 0.5 erfc(x) = 2^(-g(x)),  g(x) = 1 - log2(erfc(x))
# fit g on [0, XM] by polynomial in x (g(0) = 1 exactly pinned), weighted
def gexact(x):
    return 1.0 - np.log2(erfc(x))
def fit(deg, XM, n=4000, wpow=0.0):
    x = np.cos(np.pi * (np.arange(n) + 0.5) / n) * 0.5 * XM + 0.5 * XM   # chebyshev nodes
    y = gexact(x) - 1.0
    # polynomial without constant: y = x*(c1 + c2 x + ...)
    A = np.stack([x ** k for k in range(1, deg + 1)], 1)
    w = 1.0 / (1.0 + x) ** wpow
    # iteratively reweighted to approximate minimax of weighted error
    ww = np.ones_like(x)
    for it in range(60):
        c, *_ = np.linalg.lstsq(A * (w * ww)[:, None], y * w * ww, rcond=None)
        e = np.abs(A @ c - y) * w
        ww = ww * (1 + 2.0 * e / e.max())
        ww /= ww.mean()
    return c
def gelu_exact(v):
    return 0.5 * v * (1 + erf(v / np.sqrt(2.0)))
def gelu_poly(v, c):
    v = v.astype(np.float32)
    x = (np.abs(v) * np.float32(0.70710678118654752440)).astype(np.float32)
    p = np.float32(c[-1])
    for k in range(len(c) - 2, -1, -1):
        p = (p * x + np.float32(c[k])).astype(np.float32)
    p = (p * x + np.float32(1.0)).astype(np.float32)    # g(x)
    h = np.exp2(-p).astype(np.float32)
    return (np.maximum(v, 0) - np.abs(v) * h).astype(np.float32)
def gelu_as(v):   # current: A&S 7.1.26
    v = v.astype(np.float32)
    x = np.abs(v) * np.float32(0.70710678118654752440)
    t = (1 / (np.float32(0.3275911) * x + 1)).astype(np.float32)
    pl = np.float32(1.061405429) * t + np.float32(-1.453152027)
    pl = pl * t + np.float32(1.421413741); pl = pl * t + np.float32(-0.284496736); pl = pl * t + np.float32(0.254829592)
    e = 1 - pl * t * np.exp2(x * x * np.float32(-1.44269504088896340736))
    return (0.5 * v + 0.5 * np.abs(v) * e).astype(np.float32)
v = np.linspace(-12, 12, 2000001)
ge = gelu_exact(v)
def report(name, g):
    err = np.abs(g - ge)
    ulp = np.maximum(np.abs(ge), 1e-30) * 2.0 ** -8     # half-ulp-ish bf16 spacing scale
    print("%-22s max abs %.3e   max abs/(bf16 ulp of result) %.3f  (at v=%.3f)  max rel for |v|<6 %.3e" % (name, err.max(), (err / ulp).max(), v[np.argmax(err / ulp)], (err / np.maximum(np.abs(ge), 1e-30))[np.abs(v) < 6].max()))
report("A&S 7.1.26 (now)", gelu_as(v))
for deg in (5, 6, 7, 8):
    for XM in (4.5, 6.0):
        for wp in (0.0, 1.0, 2.0):
            c = fit(deg, XM, wpow=wp)
            report("deg %d XM %.1f w %.0f" % (deg, XM, wp), gelu_poly(v, c))
print()
np.set_printoptions(precision=10)
for deg, XM, wp in ((6, 4.5, 2.0), (6, 4.5, 1.0), (7, 4.5, 1.0), (6, 5.0, 2.0), (7, 5.0, 2.0)):
    c = fit(deg, XM, wpow=wp)
    xs = np.array([4.5, 5, 6, 8, 10, 15, 30, 100.0])
    g = 1 + sum(c[k] * xs ** (k + 1) for k in range(deg))
    print(deg, XM, wp, "coeffs", c, "\n   g at", xs, "=", g, " exact g(4.5, 5, 6) =", gexact(np.array([4.5, 5, 6.0])))
    vv = v[np.abs(v) < 5]
    gg = gelu_poly(vv, c); ee = gelu_exact(vv)
    print("   |v|<5: max abs %.3e, max err / bf16 half-ulp of result %.4f" % (np.abs(gg - ee).max(), (np.abs(gg - ee) / (np.abs(ee) * 2.0 ** -9 + 1e-30)).max()))
c = fit(7, 4.5, wpow=1.0)
print("C7 =", ", ".join("%.9ef" % x for x in c))
