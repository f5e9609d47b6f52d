import sys, time; sys.path.insert(0,'.')
import numpy as np, scipy.sparse as sp, scipy.sparse.linalg as spl
from tests.dev.attic.sa_experiment import pairwise_aggregates, setup
from tests.dev.attic.boxmg_experiment import load
from tests.dev.attic.amg_experiment5 import fpcg
which=sys.argv[1]
cond=load(which, sys.argv[2:])
A,rhs,fixed,val=setup(cond,0.)
def hierarchy(A, passes, theta=0.03, min_n=300):
    levels=[]
    while A.shape[0]>min_n and len(levels)<40:
        n=A.shape[0]
        agg,nc=pairwise_aggregates(A,theta,passes)
        if nc>0.85*n: break
        P=sp.csr_matrix((np.ones(n),(np.arange(n),agg)),shape=(n,nc))
        levels.append((A,P)); A=(P.T@A@P).tocsr()
    levels.append((A,None)); return levels
def make_cycle(levels, nu=1, omega=0.7, kfrom=0, kto=99, inner=2):
    dinv=[1.0/A.diagonal() for A,_ in levels]
    lu=spl.splu(levels[-1][0].tocsc())
    def smooth(l,x,b):
        A=levels[l][0]
        for _ in range(nu): x=x+omega*dinv[l]*(b-A@x)
        return x
    def cyc(l,b):
        A,P=levels[l]
        if P is None: return lu.solve(b)
        x=smooth(l,np.zeros(b.shape),b)
        rc=P.T@(b-A@x)
        if kfrom<=l+1<=kto and levels[l+1][1] is not None:
            Ac=levels[l+1][0]; ec=np.zeros_like(rc); r=rc.copy(); pold=None
            for k in range(inner):
                z=cyc(l+1,r)
                p=z if pold is None else z-((z@Apold)/(pold@Apold))*pold
                Ap=Ac@p; a=(p@r)/(p@Ap); ec+=a*p; r-=a*Ap; pold,Apold=p,Ap
                pass
        else: ec=cyc(l+1,rc)
        x=x+P@ec
        return smooth(l,x,b)
    return lambda b: cyc(0,b)
for passes in (1,2,3):
    t=time.time(); lv=hierarchy(A,passes); ts=time.time()-t
    nn=[a.shape[0] for a,_ in lv]; nnz=[a.nnz for a,_ in lv]
    print('passes',passes,'levels',nn,'complexity %.2f'%(sum(nnz)/nnz[0]),'setup %.1f'%ts,flush=True)
    for name,kw in (('V(2,2)',dict(nu=2)),('K(1,1) all',dict(nu=1)) ,('K(2,2) all',dict(nu=2))):
        if name.startswith('V'): kw=dict(kw,kfrom=99)
        if passes==1 and name.startswith('K'): continue
        M=make_cycle(lv,**kw); t=time.time(); x,it,rr=fpcg(A,rhs,M,400,tol=1e-15)
        print('   ',name,'its',it,'relres %.1e'%rr,'%.1fs'%(time.time()-t),flush=True)
