"""NumPy prototype (round 4): CG on the Lyapunov equation Yh R + R Yh = C of the eigen-free NT route against CG on the
equivalent, better conditioned (Yh/s + s Zh) R + R (Yh/s + s Zh) = C/s + s Zh C Zh, s^2 = tr(Yh)/tr(Zh); spectra of K with
cond 10..1e5 (log-uniform, low-rank, two clusters).  Prints the CG steps to 1e-12 and the error against the exact solution."""
import numpy as np
def cg_lyap(M, C, tol=1e-12, maxit=400):
    R = np.zeros_like(C); r = C.copy(); p = r.copy(); rr = np.sum(r*r); rr0 = rr; k=0
    while k < maxit and rr > tol**2*rr0:
        T = M@p; Ap = T+T.T
        a = rr/np.sum(p*Ap); R += a*p; r -= a*Ap; rn = np.sum(r*r); p = r + (rn/rr)*p; rr = rn; k+=1
    return R, k
rng=np.random.default_rng(0)
n=300
for condK, kind in [(10,'log'),(100,'log'),(1e3,'log'),(1e4,'log'),(1e3,'lowrank'),(1e3,'two')]:
    Q,_=np.linalg.qr(rng.standard_normal((n,n)))
    if kind=='log': lam=np.logspace(0,-np.log10(condK),n)
    elif kind=='lowrank': lam=np.concatenate([np.ones(5), np.full(n-5,1/condK)*np.exp(rng.uniform(0,2,n-5))])
    else: lam=np.concatenate([np.ones(n//2)*np.exp(rng.uniform(-0.3,0,n//2)), np.full(n-n//2,1/condK)*np.exp(rng.uniform(0,0.3,n-n//2))])
    K=(Q*lam)@Q.T; c=np.linalg.norm(K,1)
    y=np.sqrt(lam/c); Y=(Q*y)@Q.T; Z=(Q*(1/y))@Q.T
    C=rng.standard_normal((n,n)); C=C+C.T
    R0,k0=cg_lyap(Y,C)
    s=np.sqrt(np.trace(Y)/np.trace(Z))
    M=Y/s+s*Z; C2=C/s+s*(Z@C@Z); C2=(C2+C2.T)/2
    R1,k1=cg_lyap(M,C2)
    # exact
    Rex=Q@((Q.T@C@Q)/(y[:,None]+y[None,:]))@Q.T
    print(kind,condK,'orig its',k0,'err %.1e'%(np.linalg.norm(R0-Rex)/np.linalg.norm(Rex)),'| combined its',k1,'(+2 products) err %.1e'%(np.linalg.norm(R1-Rex)/np.linalg.norm(Rex)), 's',s, 'kproxy %.2f'%(np.trace(Y)*np.trace(Z)/n**2))
print("---- optimal s = sqrt(ymin ymax) vs trace s")
for condK, kind in [(100,'log'),(1e3,'log'),(1e3,'lowrank'),(1e3,'two'),(1e5,'lowrank')]:
    Q,_=np.linalg.qr(rng.standard_normal((n,n)))
    if kind=='log': lam=np.logspace(0,-np.log10(condK),n)
    elif kind=='lowrank': lam=np.concatenate([np.ones(5), np.full(n-5,1/condK)*np.exp(rng.uniform(0,2,n-5))])
    else: lam=np.concatenate([np.ones(n//2)*np.exp(rng.uniform(-0.3,0,n//2)), np.full(n-n//2,1/condK)*np.exp(rng.uniform(0,0.3,n-n//2))])
    K=(Q*lam)@Q.T; c=np.linalg.norm(K,1)
    y=np.sqrt(lam/c); Y=(Q*y)@Q.T; Z=(Q*(1/y))@Q.T
    C=rng.standard_normal((n,n)); C=C+C.T
    out=[]
    for s in (np.sqrt(np.trace(Y)/np.trace(Z)), np.sqrt(y.min()*y.max())):
        M=Y/s+s*Z; C2=C/s+s*(Z@C@Z); C2=(C2+C2.T)/2
        R1,k1=cg_lyap(M,C2); out.append(k1)
    R0,k0=cg_lyap(Y,C)
    print(kind,condK,'orig',k0,'trace-s',out[0],'opt-s',out[1])
