import numpy as np
from scipy.special import erf
from scipy.optimize import least_squares
x = np.linspace(-9, 9, 200001)
g = 0.5 * x * (1 + erf(x / np.sqrt(2)))
def model(c, x):
    x2 = x * x
    p = c[0]
    for k in c[1:]:
        pass
    # p(x) = x*(c0 + c1 x2 + c2 x2^2 + ...)
    acc = np.zeros_like(x)
    for k in reversed(c):
        acc = acc * x2 + k
    u = x * acc
    u = np.clip(u, -80, 80)
    return x / (1 + np.exp(-u))
for deg in (2, 3, 4):
    c0 = np.array([1.5957691, 0.0713548] + [0.0] * (deg - 2))
    def res(c):
        return (model(c, x) - g)
    # minimax via iteratively reweighted LS
    w = np.ones_like(x)
    c = c0
    for it in range(60):
        r = least_squares(lambda c: w * res(c), c, method="lm")
        c = r.x
        e = np.abs(res(c))
        w = w * (1 + 4 * e / e.max()) ; w /= w.mean()
    e = np.abs(res(c))
    rel = e / np.maximum(np.abs(g), 1e-30)
    print(deg, c, "max abs err %.3e at x=%.3f" % (e.max(), x[e.argmax()]), "max rel err (|x|<6) %.3e" % rel[np.abs(x) < 6].max())
