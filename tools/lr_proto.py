"""Prototype of the low-rank H-step terms (hstep_lr.h): pivoted Cholesky of the SE kernel with its tangent along
ln omega, the Woodbury form of tr(A^-1) and of d log det A / d ln omega, and their even / odd folded version --
against 40-digit arithmetic (mpmath).  CPU only; `python tools/lr_proto.py`."""
import os
import numpy as np, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle.vlgp_oracle import gp_objective
import mpmath as mp
mp.mp.dps = 40

def pchol_tangent(T, omega, tol, rmax=None):
    """pivoted Cholesky of K0=exp(-omega D^2) with tangent wrt ln omega"""
    t=np.arange(T,dtype=float)
    D2=(t[:,None]-t[None,:])**2
    K=np.exp(-omega*D2); dK=-omega*D2*K
    d=np.ones(T); dd=np.zeros(T)
    G=np.zeros((T,T)); Gd=np.zeros((T,T)); r=0; piv=[]
    while r<T and d.max()>tol and (rmax is None or r<rmax):
        p=int(d.argmax()); piv.append(p)
        g=np.sqrt(d[p]); gdot=0.5*dd[p]/g
        col=K[:,p]-G[:,:r]@G[p,:r]
        cold=dK[:,p]-Gd[:,:r]@G[p,:r]-G[:,:r]@Gd[p,:r]
        G[:,r]=col/g
        Gd[:,r]=(cold-G[:,r]*gdot)/g
        d=d-G[:,r]**2; dd=dd-2*G[:,r]*Gd[:,r]
        d[piv]=0; dd[piv]=0
        r+=1
    return G[:,:r],Gd[:,:r],piv

def seg_terms_dense_mp(t, sigmasq, omega, eps, w):
    """ground truth tr(A^-1) and cs=sum_jk s_j s_k dK_jk Ainv_jk in mpmath"""
    T=len(t)
    K=mp.matrix(T,T); dK=mp.matrix(T,T)
    for i in range(T):
        for j in range(T):
            d2=mp.mpf(t[i]-t[j])**2
            k=mp.mpf(sigmasq)*mp.exp(-mp.mpf(omega)*d2)
            K[i,j]=k+(mp.mpf(eps) if i==j else 0); dK[i,j]=-k*d2*mp.mpf(omega)
    s=[mp.sqrt(mp.mpf(x)) for x in w]
    A=mp.matrix(T,T)
    for i in range(T):
        for j in range(T):
            A[i,j]=s[i]*K[i,j]*s[j]+(1 if i==j else 0)
    Ai=A**-1
    tA=sum(Ai[i,i] for i in range(T))
    cs=sum(s[i]*s[j]*dK[i,j]*Ai[i,j] for i in range(T) for j in range(T))
    return float(tA),float(cs)

def seg_terms_dense(t,sigmasq,omega,eps,w):
    T=len(t); D2=(t[:,None]-t[None,:])**2
    Ks=sigmasq*np.exp(-omega*D2); dK=-omega*D2*Ks; K=Ks+eps*np.eye(T)
    s=np.sqrt(w); A=np.eye(T)+s[:,None]*K*s[None,:]
    Ai=np.linalg.inv(A)
    return np.trace(Ai), np.sum(s[:,None]*s[None,:]*dK*Ai)

def seg_terms_lr(U,Ud,eps,w):
    d=1/(1+eps*w); wt=w*d
    B0=U.T@(wt[:,None]*U); B0p=U.T@((wt*d)[:,None]*U)
    Md=Ud.T@(wt[:,None]*U); Md=Md+Md.T
    M=np.eye(U.shape[1])+B0
    Mi=np.linalg.inv(M)
    tA=d.sum()-np.sum(Mi*B0p)
    cs=np.sum(Mi*Md)
    return tA,cs

if __name__=="__main__":
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    g=np.load(os.path.join(ROOT, 'tests/golden/hstep.npz'))
    mu=g['mu']; w=g['w']; print(mu.shape, w.shape, g['logp'])
    c3=np.load('/root/repo/tests/golden/vem_c3.npz')
    W=c3['seg_w']; print('c3 seg_w', W.shape, W.min(), W.max(), np.median(W))
    T=50; t=np.arange(T,dtype=float); eps=1e-4
    rng=np.random.default_rng(0)
    for omega in [1e-3,4.4e-3,1e-2,2.7e-2,5e-2]:
        for tol in [1e-10,1e-12,1e-14]:
            G,Gd,piv=pchol_tangent(T,omega,tol)
            errs=[]
            for si in range(3):
                for l in range(W.shape[2]):
                    ww=W[si*7,:,l]
                    tA0,cs0=seg_terms_dense_mp(t,1.0,omega,eps,ww)
                    tA1,cs1=seg_terms_lr(G,Gd,eps,ww)
                    tA2,cs2=seg_terms_dense(t,1.0,omega,eps,ww)
                    errs.append((abs(tA1-tA0)/abs(tA0),abs(cs1-cs0)/max(abs(cs0),1e-300),abs(tA2-tA0)/abs(tA0),abs(cs2-cs0)/abs(cs0)))
            e=np.array(errs).max(0)
            print("omega %.1e tol %.0e rank %d  lr: tA %.1e cs %.1e | dense f64: tA %.1e cs %.1e"%(omega,tol,G.shape[1],*e))


# ---- even / odd folded version (what hstep_lr.h implements) ----
def pchol_tan(K, dK, tol, rmax):
    n=K.shape[0]; d=np.diag(K).copy(); dd=np.diag(dK).copy()
    G=np.zeros((n,rmax)); Gd=np.zeros((n,rmax)); r=0; done=np.zeros(n,bool)
    while r<rmax and r<n:
        dm=np.where(done,-1,d); p=int(dm.argmax())
        if dm[p]<=tol: break
        g=np.sqrt(d[p]); gdot=0.5*dd[p]/g
        col=K[:,p]-G[:,:r]@G[p,:r]
        cold=dK[:,p]-Gd[:,:r]@G[p,:r]-G[:,:r]@Gd[p,:r]
        G[:,r]=col/g; Gd[:,r]=(cold-G[:,r]*gdot)/g
        d=d-G[:,r]**2; dd=dd-2*G[:,r]*Gd[:,r]; done[p]=True; r+=1
    return G[:,:r],Gd[:,:r]

def tables(T, dt, sigmasq, omega, tol, rmax=64):
    """folded even/odd low-rank factor of sigmasq*exp(-omega D^2) and its d/dln(omega)"""
    h=T//2; nt=(T+1)//2; odd=T%2
    t=np.arange(T)*dt
    kf=lambda a,b: np.exp(-omega*(a-b)**2)
    dkf=lambda a,b: -omega*(a-b)**2*np.exp(-omega*(a-b)**2)
    ti=t[:nt,None]; tj=t[None,:nt]; tjr=t[None,T-1-np.arange(nt)]
    # even: e_tau=(d_tau+d_tau')/sqrt2 (tau<h), e_m=d_m
    Ke=kf(ti,tj)+kf(ti,tjr); dKe=dkf(ti,tj)+dkf(ti,tjr)
    if odd:
        m=nt-1
        Ke[m,:]=kf(t[m],t[:nt])*np.sqrt(2); Ke[:,m]=Ke[m,:]; Ke[m,m]=1.0
        dKe[m,:]=dkf(t[m],t[:nt])*np.sqrt(2); dKe[:,m]=dKe[m,:]; dKe[m,m]=0.0
    Ko=(kf(ti,tj)-kf(ti,tjr))[:h,:h]; dKo=(dkf(ti,tj)-dkf(ti,tjr))[:h,:h]
    Ge,Gde=pchol_tan(Ke,dKe,tol,rmax); Go,Gdo=pchol_tan(Ko,dKo,tol,rmax)
    re,ro=Ge.shape[1],Go.shape[1]
    s=np.sqrt(sigmasq)
    U=np.zeros((nt,re+ro)); Ud=np.zeros((nt,re+ro))
    U[:,:re]=s*Ge; Ud[:,:re]=s*Gde; U[:h,re:]=s*Go; Ud[:h,re:]=s*Gdo
    return U,Ud,re,ro

def sweep_inv(M):
    A=M.copy(); r=A.shape[0]
    for k in range(r):
        d=1.0/A[k,k]; col=A[:,k].copy()     # = pivot row by symmetry
        f=col*d; f[k]=1-d
        A=A-np.outer(f,col)
        A[:,k]=f; A[k,k]=-d
        # row k: a_kj - (1-d) a_kj = d a_kj OK; but A[k,:] col k fixed above
    return -A

def seg_terms_eo(U,Ud,re,ro,T,eps,w):
    h=T//2; nt=(T+1)//2
    d=1/(1+eps*w); wt=w*d; wd=wt*d
    def fold(x):
        ap=np.zeros(nt); am=np.zeros(nt)
        ap[:h]=0.5*(x[:h]+x[::-1][:h]); am[:h]=0.5*(x[:h]-x[::-1][:h])
        if T%2: ap[h]=x[h]
        return ap,am
    ap,am=fold(wt); bp,bm=fold(wd)
    r=re+ro; par=np.arange(r)>=re
    same=(par[:,None]==par[None,:])
    def build(A,B,wp,wm):
        return np.where(same,A.T@(wp[:,None]*B),A.T@(wm[:,None]*B))
    B0=build(U,U,ap,am); B0p=build(U,U,bp,bm)
    Md=build(Ud,U,ap,am); Md=Md+Md.T
    Mi=sweep_inv(np.eye(r)+B0)
    return d.sum()-np.sum(Mi*B0p), np.sum(Mi*Md)

if __name__=="__main__":
    c3=np.load(os.path.join(ROOT, 'tests/golden/vem_c3.npz')); W=c3['seg_w']
    rng=np.random.default_rng(1)
    for T in (50,49,24,33,64):
        t=np.arange(T,dtype=float)
        for omega in [1e-3,8e-3,2e-2]:
            for tol in [1e-12,1e-13,1e-14]:
                U,Ud,re,ro=tables(T,1.0,0.8,omega,tol)
                errs=[]
                for si in range(2):
                    ww=np.resize(W[si*7,:,si],T)*rng.uniform(0.5,20,T)
                    tA0,cs0=seg_terms_dense_mp(t,0.8,omega,1e-4,ww)
                    tA1,cs1=seg_terms_eo(U,Ud,re,ro,T,1e-4,ww)
                    errs.append((abs(tA1-tA0)/abs(tA0),abs(cs1-cs0)/abs(cs0)))
                e=np.array(errs).max(0)
                print("T %d omega %.0e tol %.0e ranks %d+%d: tA %.1e cs %.1e"%(T,omega,tol,re,ro,*e))
