"""fit of R(a) = -log2 Phi(-a) on [0, 6.5] (degree 8, weighted minimax by iterated reweighting) for csrc/common.h's GELU,
and its fp32 accuracy against float64 (also prints the previous Abramowitz-Stegun form for comparison)"""
import numpy as np
from scipy.special import log_ndtr, ndtr
from numpy.polynomial import chebyshev as Ch
A = 6.5; deg = 8
def R(a): return -log_ndtr(-a)/np.log(2.0)
xs = np.cos(np.linspace(0, np.pi, 8001)); a = (xs+1)*A/2
h = np.exp(log_ndtr(-a))
W = h*np.log(2)*np.maximum(a, 1.0)
wt = W.copy()
for it in range(200):
    c = Ch.chebfit(xs, R(a), deg, w=wt)
    err = (Ch.chebval(xs,c)-R(a))*W
    wt = wt*(1+ 0.5*np.abs(err)/np.abs(err).max())
p = Ch.cheb2poly(c)
P = np.polynomial.Polynomial(p)(np.polynomial.Polynomial([-1, 2/A]))
coef = P.coef
print("fit err", np.abs(err).max())
c32 = coef.astype(np.float32)
print(", ".join(f"{v:.9e}f" for v in c32))
# fp32 evaluation with fma emulated in float64 then rounded
x = np.concatenate([np.linspace(-12,12,2000001), np.random.default_rng(0).normal(size=1000000)*2]).astype(np.float32)
ax = np.minimum(np.abs(x), np.float32(A)).astype(np.float32)
r = np.full_like(ax, c32[-1])
for k in range(deg-1,-1,-1):
    r = (r.astype(np.float64)*ax.astype(np.float64) + np.float64(c32[k])).astype(np.float32)
e = np.exp2(-r.astype(np.float64)).astype(np.float32)
relu = (0.5*x.astype(np.float64)+0.5*np.abs(x).astype(np.float64)).astype(np.float32)
g = (relu.astype(np.float64) - ax.astype(np.float64)*e.astype(np.float64)).astype(np.float32)
x64 = x.astype(np.float64)
gref = x64*ndtr(x64)
print("gelu abs err", np.abs(g-gref).max(), "rel err (|x|>1e-3)", (np.abs(g-gref)/np.maximum(np.abs(gref),1e-3)).max())
cdf = np.where(x>=0, (1-e.astype(np.float64)), e.astype(np.float64)).astype(np.float32)
print("Phi abs err", np.abs(cdf-ndtr(x64)).max())
phi = np.exp2((x64*x64*(-0.5*np.log2(np.e)) + np.log2(1/np.sqrt(2*np.pi))).astype(np.float32).astype(np.float64)).astype(np.float32)
der = (x64*phi + cdf).astype(np.float32)
dref = ndtr(x64) + x64*np.exp(-0.5*x64*x64)/np.sqrt(2*np.pi)
print("gelu' abs err", np.abs(der-dref).max())
# old A&S form for comparison
z = np.abs(x64)*0.7071067811865476; t = 1/(1+0.3275911*z)
poly = t*(0.254829592+t*(-0.284496736+t*(1.421413741+t*(-1.453152027+t*1.061405429))))
half = 0.5*poly*np.exp(-z*z); cdf0 = np.where(x64>=0, 1-half, half)
print("old: gelu abs err", np.abs(x64*cdf0-gref).max(), "Phi", np.abs(cdf0-ndtr(x64)).max())
