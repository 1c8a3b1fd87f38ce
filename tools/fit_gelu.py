import numpy as np
from scipy.special import erfc, log_ndtr
from numpy.polynomial import chebyshev as C
UMAX=6.0
def target(u): return -log_ndtr(-u)/np.log(2.0)    # Q(u) = -log2 Phi(-u)
u=np.cos(np.pi*(np.arange(4000)+0.5)/4000)*UMAX/2+UMAX/2
f=np.exp(log_ndtr(-u))
for deg in (5,6,7,8,9):
    # weighted LSQ with iterative reweighting toward minimax of gelu abs error = u*f*ln2*dQ
    w=u*f+1e-9
    wt=np.ones_like(u)
    for it in range(60):
        A=np.vander(u,deg+1,increasing=True)*(w*wt)[:,None]
        b=target(u)*(w*wt)
        coef,*_=np.linalg.lstsq(A,b,rcond=None)
        err=np.abs((np.vander(u,deg+1,increasing=True)@coef-target(u))*w)
        wt=wt*(1+ (err/err.max())**2*2); wt/=wt.mean()
    # evaluate in float32 Horner, compute gelu error vs exact on dense grid incl. negative
    v=np.linspace(-8,8,400001)
    uu=np.minimum(np.abs(v),UMAX).astype(np.float32)
    q=np.float32(coef[-1])*np.ones_like(uu)
    for c_ in coef[-2::-1]:
        q=q*uu+np.float32(c_)
    fa=np.exp2(-q.astype(np.float64))
    phi=np.where(v>=0,1-fa,fa)
    g=v*phi
    ge=v*0.5*erfc(-v/np.sqrt(2))
    print(deg,"max abs gelu err %.3e"%np.abs(g-ge).max(),"at v=%.3f"%v[np.abs(g-ge).argmax()], "coef",["%.9g"%c_ for c_ in coef])
