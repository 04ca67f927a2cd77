"""Fit of the GELU polynomial of csrc/block_stream.hip (gelu_op): Phi(x) ~ clamp01(1/2 + u q(u^2 - shift)), u = x / 4, minimax on the
error of x Phi(x) over |x| <= 12, evaluated in emulated fp16 (one rounding per fma).  Prints the candidates; the kernel uses degree 4,
shift 0.5, fit range |u| <= 0.9.   python scripts/fit_gelu.py"""
import numpy as np
from scipy.optimize import minimize
from scipy.special import erf
def phi(x): return 0.5*(1+erf(x/np.sqrt(2)))
def gelu(x): return x*phi(x)
f16=lambda a: np.asarray(a,dtype=np.float32).astype(np.float16).astype(np.float32)
def fma16(a,b,c): return f16(np.float32(a)*np.float32(b)+np.float32(c))   # one rounding
def eval16(c, x, shift):
    # u = x/4 in fp16 (converted from the fp32 accumulator), s = u*u - shift, q = Horner(c, s), P = sat01(0.5 + u q), out = u*P  (= gelu/4)
    u=f16(x/4)
    s=fma16(u,u,-shift)
    q=fma16(s,f16(c[0]),f16(c[1]))
    for k in range(2,len(c)): q=fma16(q,s,f16(c[k]))
    P=np.clip(fma16(u,q,0.5),0,1)
    return f16(u*P)*4
def evalf(c,x,shift):
    u=x/4; s=u*u-shift
    q=np.polyval(c,s)
    return x*np.clip(0.5+u*q,0,1)
xs=np.linspace(-12,12,48001)
w=np.exp(-xs**2/(2*1.5**2))
def fit(deg,shift,umax):
    # least squares start on |x| <= 4*umax
    m=np.abs(xs)<=4*umax
    u=xs[m]/4; s=u*u-shift
    A=np.stack([u*s**k for k in range(deg,-1,-1)],1)*xs[m][:,None]
    b=(phi(xs[m])-0.5)*xs[m]
    c=np.linalg.lstsq(A,b,rcond=None)[0]
    def obj(c):
        e=evalf(c,xs,shift)-gelu(xs)
        return np.max(np.abs(e))
    r=minimize(obj,c,method='Nelder-Mead',options=dict(maxiter=40000,xatol=1e-9,fatol=1e-12))
    return r.x
for deg in (3,4,5):
    for shift in (0.0,0.5):
        for umax in (0.9,1.0,1.1):
            c=fit(deg,shift,umax)
            e=evalf(c,xs,shift)-gelu(xs)
            e16=eval16(c,xs,shift)-gelu(xs)
            print(f"deg {deg} shift {shift} umax {umax}: fp64 max {np.abs(e).max():.2e}  fp16 max {np.abs(e16).max():.2e} rms(N(0,1.5)) {np.sqrt((e16**2*w).sum()/w.sum()):.2e}  lead {c[0]:+.3f}")
print()
best=None
for umax in (0.8,0.85,0.9,0.95):
    c=fit(4,0.5,umax)
    e16=eval16(c,xs,0.5)-gelu(xs)
    print(umax, np.abs(e16).max(), np.sqrt((e16**2*w).sum()/w.sum()), [float(np.float16(v)) for v in c], list(c))
