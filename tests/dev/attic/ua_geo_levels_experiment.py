"""Scratch experiment 11 (CPU, scipy; round 4): the same split-block aggregation on more than one level (level k groups nodes by the
2^(k+1) block of their home cell): only the first level pays.  usage: python tests/dev/attic/ua_geo_levels_experiment.py c1|g10"""
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
def geo_hierarchy(A, geo_levels, theta=0.03, theta_p=0.03, min_n=300, stall=0.85, shift_step=1):
    levels=[]; n=A.shape[0]
    hr,hc=np.divmod(np.arange(n),C)          # home cell of every node
    while A.shape[0]>min_n and len(levels)<40:
        n=A.shape[0]; k=len(levels)
        if k<geo_levels:
            sh=(k+1)*shift_step
            i,j,w,strong,d=strong_mask(A,theta)
            keep=strong&((hr[i]>>sh)==(hr[j]>>sh))&((hc[i]>>sh)==(hc[j]>>sh))
            G=sp.csr_matrix((np.ones(keep.sum()),(i[keep],j[keep])),shape=(n,n))
            nc,agg=csg.connected_components(G,directed=False)
        else:
            agg,nc=pairwise_aggregates(A,theta_p,1)
        if nc>stall*n: break
        P=sp.csr_matrix((np.ones(n),(np.arange(n),agg)),shape=(n,nc))
        # home of an aggregate: its first member
        first=np.full(nc,n,dtype=np.int64); np.minimum.at(first,agg,np.arange(n))
        hr,hc=hr[first],hc[first]
        levels.append((A,P)); A=(P.T@A@P).tocsr()
    levels.append((A,None)); return levels
for geo in (0,1,2,3,5,20):
    t=time.time(); lv=geo_hierarchy(A0,geo); ts=time.time()-t
    nnz=[a.nnz for a,_ in lv]
    M=make_cycle(lv); x,it,rr=fpcg(A0,rhs,M,600,tol=1e-15)
    print(which,'geo levels',geo,'n',[a.shape[0] for a,_ in lv][:7],'levels',len(lv),'nnz/row',[round(a.nnz/a.shape[0],1) for a,_ in lv][:4],'complexity %.2f'%(sum(nnz)/nnz[0]),'its',it,flush=True)
