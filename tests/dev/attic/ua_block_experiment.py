"""Scratch experiment 10 (CPU, scipy; round 4): the first level aggregated by 2 x 2 raster blocks split into the connected components
of their strong links (unsmoothed), pairwise aggregation below -- the variant that shipped (amg.hip: k_block_agg).
usage: python tests/dev/attic/ua_block_experiment.py c1|g10|g11"""
import sys, time; sys.path.insert(0,'.')
import numpy as np, scipy.sparse as sp, scipy.sparse.linalg as spl, scipy.sparse.csgraph as csg
from tests.dev.attic.sa_experiment import pairwise_aggregates, strong_mask, setup
import tests.dev.attic.boxmg_experiment as bx
from tests.dev.attic.amg_experiment5 import fpcg
which=sys.argv[1]
cond=bx.load(which, [])
R,C=cond.shape
A0,rhs,fixed,val=setup(cond,0.)
def make_cycle(levels, nu=1, omega=0.7):
    dinv=[1.0/A.diagonal() for A,_ in levels]
    lu=spl.splu(levels[-1][0].tocsc())
    def cyc(l,b):
        A,P=levels[l]
        if P is None: return lu.solve(b)
        x=np.zeros(b.shape)
        for _ in range(nu): x=x+omega*dinv[l]*(b-A@x)
        x=x+P@cyc(l+1,P.T@(b-A@x))
        for _ in range(nu): x=x+omega*dinv[l]*(b-A@x)
        return x
    return lambda b: cyc(0,b)
def block_aggregates(A, shape, bs, theta):
    R,C=shape; n=R*C
    i,j,w,strong,d=strong_mask(A,theta)
    r_i,c_i=np.divmod(i,C); r_j,c_j=np.divmod(j,C)
    keep=strong&(r_i//bs[0]==r_j//bs[0])&(c_i//bs[1]==c_j//bs[1])
    G=sp.csr_matrix((np.ones(keep.sum()),(i[keep],j[keep])),shape=(n,n))
    nc,lab=csg.connected_components(G,directed=False)
    return lab,nc
def hierarchy(A, first, theta0=0.03/8, theta_p=0.03, min_n=300, stall=0.85):
    levels=[]
    while A.shape[0]>min_n and len(levels)<40:
        n=A.shape[0]
        if len(levels)==0 and first is not None: agg,nc=block_aggregates(A,(R,C),first,theta0)
        else: agg,nc=pairwise_aggregates(A,theta_p,1)
        if nc>stall*n: break
        P=sp.csr_matrix((np.ones(n),(np.arange(n),agg)),shape=(n,nc))
        levels.append((A,P)); A=(P.T@A@P).tocsr()
    levels.append((A,None)); return levels
for first,th in ((None,0),((2,2),0.03/8),((2,2),0.03),((2,2),0.1),((2,1),0.03)):
    t=time.time(); lv=hierarchy(A0,first,theta0=th); ts=time.time()-t
    nnz=[a.nnz for a,_ in lv]
    M=make_cycle(lv); t=time.time(); x,it,rr=fpcg(A0,rhs,M,600,tol=1e-15)
    print(which,first,th,'n',[a.shape[0] for a,_ in lv][:4],'nnz/row',[round(a.nnz/a.shape[0],1) for a,_ in lv][:3],'complexity %.2f'%(sum(nnz)/nnz[0]),'its',it,flush=True)
