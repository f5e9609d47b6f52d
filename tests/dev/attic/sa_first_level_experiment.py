"""Scratch experiment 9 (CPU, scipy; round 4): smoothed prolongation on the FIRST level only (aggregates: 3 x 3 raster blocks split
into their strongly connected parts "b3", greedy "g", pairwise "p"), unsmoothed pairwise aggregation below; diagnostics of the
level-1 operator and the two-grid method with an exact level-1 solve.  usage: python tests/dev/attic/sa_first_level_experiment.py c1|g10 [b3,g,p]"""
import sys, time; sys.path.insert(0,'.')
import numpy as np, scipy.sparse as sp, scipy.sparse.linalg as spl, scipy.sparse.csgraph as csg
from tests.dev.attic.sa_experiment import pairwise_aggregates, greedy_aggregates, strong_mask, setup
import tests.dev.attic.boxmg_experiment as bx
from tests.dev.attic.amg_experiment5 import fpcg
def make_cycle(levels, nu=1, omega=0.7):
    dinv = [1.0 / A.diagonal() for A, _ in levels]
    lu = spl.splu(levels[-1][0].tocsc())

    def cyc(l, b):
        A, P = levels[l]
        if P is None:
            return lu.solve(b)
        x = np.zeros(b.shape)
        for _ in range(nu):
            x = x + omega * dinv[l] * (b - A @ x)
        x = x + P @ cyc(l + 1, P.T @ (b - A @ x))
        for _ in range(nu):
            x = x + omega * dinv[l] * (b - A @ x)
        return x
    return lambda b: cyc(0, b)
which=sys.argv[1]
cond=bx.load(which, [a for a in sys.argv[2:] if a[0].isdigit()])
R,C=cond.shape
A0,rhs,fixed,val=setup(cond,0.)
def block_aggregates(A, shape, bs, theta):
    R,C=shape; n=R*C
    i,j,w,strong,d=strong_mask(A,theta)
    r_i,c_i=np.divmod(i,C); r_j,c_j=np.divmod(j,C)
    same=(r_i//bs==r_j//bs)&(c_i//bs==c_j//bs)
    keep=strong&same
    G=sp.csr_matrix((np.ones(keep.sum()),(i[keep],j[keep])),shape=(n,n))
    nc,lab=csg.connected_components(G,directed=False)
    return lab,nc
def hierarchy(A, first, theta0=0.02, theta_p=0.03, omega=2./3, min_n=300, smooth_second=False):
    levels=[]
    while A.shape[0]>min_n and len(levels)<40:
        n=A.shape[0]
        if len(levels)==0 and first[0]=='b':
            agg,nc=block_aggregates(A,(R,C),int(first[1:]),theta0); th=theta0; smooth=True
        elif len(levels)==0 and first=='g':
            i,j,w,strong,d=strong_mask(A,theta0); agg,nc=greedy_aggregates(n,i,j,strong); th=theta0; smooth=True
        else:
            agg,nc=pairwise_aggregates(A,theta_p,1); th=theta_p/8.0; smooth=False
        if nc>0.85*n: break
        T=sp.csr_matrix((np.ones(n),(np.arange(n),agg)),shape=(n,nc))
        if smooth:
            i,j,w,strong,d=strong_mask(A,th)
            weak_sum=np.bincount(i[~strong],weights=w[~strong],minlength=n)
            dF=d-weak_sum
            AF=sp.csr_matrix((np.r_[-w[strong],dF],(np.r_[i[strong],np.arange(n)],np.r_[j[strong],np.arange(n)])),shape=(n,n))
            scale=np.where(dF>1e-300,omega/np.where(dF>1e-300,dF,1.0),0.0)
            P=(T-sp.diags(scale)@(AF@T)).tocsr()
        else: P=T
        levels.append((A,P)); A=(P.T@A@P).tocsr()
    levels.append((A,None)); return levels
for first in (sys.argv[2].split(',') if len(sys.argv)>2 and not sys.argv[2][0].isdigit() else ('p','g','b2','b3','b4')):
    t=time.time()
    lv=hierarchy(A0,first) if first!='p' else hierarchy(A0,'x')
    ts=time.time()-t
    nnz=[a.nnz for a,_ in lv]
    M=make_cycle(lv); t=time.time(); x,it,rr=fpcg(A0,rhs,M,500,tol=1e-15)
    print(which,first,'n',[a.shape[0] for a,_ in lv][:5],'nnz/row',[round(a.nnz/a.shape[0],1) for a,_ in lv][:4],'P nnz/row %.1f'%(lv[0][1].nnz/lv[0][1].shape[0]),'complexity %.2f'%(sum(nnz)/nnz[0]),'its',it,'%.0fs'%(time.time()-t),flush=True)
print('--- diagnostics', flush=True)
lv=hierarchy(A0,'b3')
A1=lv[1][0]; d1=A1.diagonal(); off=A1-sp.diags(d1)
rowabs=np.asarray(abs(off).sum(1)).ravel(); pos=np.asarray(off.maximum(0).sum(1)).ravel()
print('level1 n',A1.shape[0],'min diag',d1.min(),'rows with diag < sum|off|:',int((d1<rowabs*(1-1e-12)).sum()),'max ratio sum|off|/diag',float((rowabs/d1).max()),'rows with positive off-diag sum > 1e-3 diag:',int((pos>1e-3*d1).sum()))
# two-grid with exact coarse solve
lu=spl.splu(A1.tocsc()); A=A0; P=lv[0][1]; dinv=1/A.diagonal()
def M2(b):
    x=0.7*dinv*b; x=x+P@lu.solve(P.T@(b-A@x)); x=x+0.7*dinv*(b-A@x); return x
x,it,rr=fpcg(A0,rhs,M2,300,tol=1e-15); print('two-grid (exact level 1) its',it,flush=True)
# l1-Jacobi on deeper levels
def make_cycle_l1(levels, nu=1):
    dinv=[1.0/np.maximum(A.diagonal(), np.asarray(abs(A).sum(1)).ravel()/2) for A,_ in levels]
    lu=spl.splu(levels[-1][0].tocsc())
    def cyc(l,b):
        A,P=levels[l]
        if P is None: return lu.solve(b)
        om=0.7
        x=om*dinv[l]*b
        x=x+P@cyc(l+1,P.T@(b-A@x))
        x=x+om*dinv[l]*(b-A@x)
        return x
    return lambda b: cyc(0,b)
x,it,rr=fpcg(A0,rhs,make_cycle_l1(lv),500,tol=1e-15); print('V(1,1) with max(diag, l1/2) scaling its',it,flush=True)
