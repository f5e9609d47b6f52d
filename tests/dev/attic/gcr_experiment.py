"""Round 4: GCR(m) / FCG(m) on the exact (east-edge quirk) operator against CG on the symmetric one + BiCGStab, scipy,
the V(1,1) pairwise hierarchy of kcycle_proto.py.  python tests/dev/attic/gcr_experiment.py x 700 900  (profiles/r04_k5.md)"""
import sys, time, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT); os.chdir(ROOT)
import numpy as np, scipy.sparse as sp, scipy.sparse.linalg as spl
from tests.dev.attic.amg_experiment import build_system
from tests.dev.attic.sa_experiment import pairwise_aggregates
import tests.dev.attic.boxmg_experiment as bx
src=open(os.path.join(ROOT, 'tests/dev/attic/kcycle_proto.py')).read()
exec("def hierarchy"+src.split("def hierarchy")[1].split("for passes in")[0])
def setup_q(cond, dirn, quirk):
    Cs, fixed, val = build_system(cond, dirn, quirk=quirk)
    free=~fixed; F=sp.diags(free.astype(float))
    Cff=(F@Cs@F).tocsr()
    deg=np.asarray(Cs.sum(1)).ravel(); deg[fixed]=1.0
    rhs=np.asarray(F@Cs@(val*fixed)).ravel()
    A=(sp.diags(deg)-Cff).tocsr()
    return A, rhs
def fcg(A,b,M,maxit,tol=1e-15,m=1):
    x=np.zeros_like(b); r=b.copy(); P=[];Q=[]; b2=np.linalg.norm(b)
    for it in range(1,maxit+1):
        z=M(r); p=z.copy()
        for pj,qj in zip(P,Q): p-= (z@qj)/(pj@qj)*pj
        q=A@p; a=(p@r)/(p@q); x+=a*p; r-=a*q
        P.append(p);Q.append(q); P=P[-m:];Q=Q[-m:]
        if np.linalg.norm(r)<=tol*b2: break
    return x,it,np.linalg.norm(b-A@x)/b2
def gcr(A,b,M,maxit,tol=1e-15,m=1):
    # ORTHOMIN(m): residual-minimising, q's mutually orthogonal
    x=np.zeros_like(b); r=b.copy(); P=[];Q=[];N=[]; b2=np.linalg.norm(b)
    for it in range(1,maxit+1):
        p=M(r); q=A@p
        for pj,qj,nj in zip(P,Q,N):
            c=(q@qj)/nj; p-=c*pj; q-=c*qj
        n=q@q; a=(r@q)/n; x+=a*p; r-=a*q
        P.append(p);Q.append(q);N.append(n); P=P[-m:];Q=Q[-m:];N=N[-m:]
        if np.linalg.norm(r)<=tol*b2: break
    return x,it,np.linalg.norm(b-A@x)/b2
def bicgstab(A,b,M,maxit,tol=1e-15,x0=None):
    x=np.zeros_like(b) if x0 is None else x0.copy(); r=b-A@x; rh=r.copy(); b2=np.linalg.norm(b)
    rho=al=om=1.0; v=np.zeros_like(b); p=np.zeros_like(b)
    for it in range(1,maxit+1):
        rho1=rh@r; be=(rho1/rho)*(al/om); rho=rho1
        p=r+be*(p-om*v); ph=M(p); v=A@ph; al=rho/(rh@v); s=r-al*v
        sh=M(s); t=A@sh; om=(t@s)/(t@t); x+=al*ph+om*sh; r=s-om*t
        if np.linalg.norm(r)<=tol*b2: break
    return x,it,np.linalg.norm(b-A@x)/b2
cond=bx.load(sys.argv[1],sys.argv[2:])
As,bs=setup_q(cond,0.,False); Aq,bq=setup_q(cond,0.,True)
lv=hierarchy(As,1)
M=make_cycle(lv,nu=1,kfrom=99)
t=time.time(); xs,it,rr=fcg(As,bs,M,600,m=1); print('sym FCG1 its',it,'relres %.1e'%rr,flush=True)
x2,it2,rr2=bicgstab(Aq,bq,M,300,x0=xs); print('  + BiCGStab from it: its',it2,'(V-cycles %d)'%(2*it2),'relres %.1e'%rr2,'=> total V-cycles',it+2*it2,flush=True)
for m in (1,2,3,4,5,6,8,16,32,1000):
    x,it,rr=gcr(Aq,bq,M,600,m=m); print('quirk GCR(%d) its'%m,it,'relres %.1e'%rr,flush=True)
M2=make_cycle(lv,nu=2,kfrom=99)
for m in (1,4):
    x,it,rr=gcr(Aq,bq,M2,600,m=m); print('nu2 quirk GCR(%d) its'%m,it,'relres %.1e'%rr,flush=True)
