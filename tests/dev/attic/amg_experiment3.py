"""Scratch experiment 3: unstructured pairwise (strongest-neighbour) aggregation
AMG + PCG on the C1 potential system, Jacobi-type smoothing only (what a GPU
implementation would use)."""
import sys, time
sys.path.insert(0, '.')
import numpy as np, scipy.sparse as sp, scipy.sparse.linalg as spl
from oracle import ssrs_oracle as orc
from tests.dev.attic.amg_experiment2 import setup

def pairwise(A, passes=2):
    """returns aggregate id per node after `passes` rounds of strongest-neighbour matching"""
    n = A.shape[0]
    agg = np.arange(n)
    Acur = A
    for _ in range(passes):
        m = Acur.shape[0]
        S = -(Acur - sp.diags(Acur.diagonal())).tocsr()      # positive couplings
        S.eliminate_zeros()
        match = np.full(m, -1)
        for rnd in range(12):
            free = match < 0
            Sf = sp.diags(free.astype(float)) @ S @ sp.diags(free.astype(float))
            Sf = Sf.tocsr(); Sf.eliminate_zeros()
            best = np.full(m, -1)
            rows_with = np.diff(Sf.indptr) > 0
            # argmax per row
            data, indices, indptr = Sf.data, Sf.indices, Sf.indptr
            rowid = np.repeat(np.arange(m), np.diff(indptr))
            order = np.lexsort((-data, rowid))
            first = np.r_[True, rowid[order][1:] != rowid[order][:-1]]
            best[rowid[order][first]] = indices[order][first]
            i = np.where(rows_with & free)[0]
            mutual = i[(best[best[i]] == i)]
            match[mutual] = best[mutual]
        # build aggregates: pairs share id = min(i, match)
        cid = np.where(match >= 0, np.minimum(np.arange(m), match), np.arange(m))
        uniq, inv = np.unique(cid, return_inverse=True)
        P = sp.csr_matrix((np.ones(m), (np.arange(m), inv)), shape=(m, uniq.size))
        Acur = (P.T @ Acur @ P).tocsr()
        agg = inv[agg]
    return agg, Acur

def hierarchy(A, min_n=500, max_levels=25):
    levels = []
    while A.shape[0] > min_n and len(levels) < max_levels:
        agg, Ac = pairwise(A, 2)
        n = A.shape[0]
        P = sp.csr_matrix((np.ones(n), (np.arange(n), agg)), shape=(n, Ac.shape[0]))
        levels.append((A, P))
        if Ac.shape[0] > 0.9 * n: 
            A = Ac; break
        A = Ac
    levels.append((A, None))
    return levels

def make_cycle(levels, nu=2, omega=0.7, gamma=1, kcycle=False):
    dinv = [1.0 / A.diagonal() for A, _ in levels]
    l1 = [1.0 / np.asarray(abs(A).sum(1)).ravel() for A, _ in levels]
    lu = spl.splu(levels[-1][0].tocsc())
    def smooth(l, x, b):
        A = levels[l][0]
        for _ in range(nu):
            x += omega * dinv[l] * (b - A @ x)
        return x
    def cyc(l, b):
        A, P = levels[l]
        if P is None: return lu.solve(b)
        x = smooth(l, np.zeros(b.shape), b)
        rc = P.T @ (b - A @ x)
        if kcycle and l + 1 < len(levels) - 1:
            # 2 steps of flexible CG on the coarse level preconditioned by the cycle
            Ac = levels[l + 1][0]
            ec = np.zeros_like(rc); r = rc.copy(); pold = None
            for k in range(2):
                z = cyc(l + 1, r)
                if pold is None: p = z
                else:
                    beta = -(z @ Apold) / (pold @ Apold); p = z + beta * pold
                Ap = Ac @ p; alpha = (p @ r) / (p @ Ap)
                ec += alpha * p; r -= alpha * Ap; pold, Apold = p, Ap
        else:
            ec = np.zeros_like(rc)
            for _ in range(gamma):
                ec += cyc(l + 1, rc - (levels[l + 1][0] @ ec if ec.any() else 0))
        x += P @ ec
        return smooth(l, x, b)
    return lambda b: cyc(0, b)

def fpcg(A, b, M, maxit, ref, shape, fixed, val, tol=1e-13):
    x = np.zeros_like(b); r = b.copy(); pold = None; b2 = np.linalg.norm(b)
    for it in range(1, maxit + 1):
        z = M(r)
        if pold is None: p = z
        else: p = z - ((z @ Apold) / (pold @ Apold)) * pold
        Ap = A @ p; a = (p @ r) / (p @ Ap); x += a * p; r -= a * Ap; pold, Apold = p, Ap
        if it % 10 == 0:
            xx = np.where(fixed, val, x).reshape(shape)
            print(f'   it {it:4d} relres {np.linalg.norm(r)/b2:.2e} maxerr {np.abs(xx-ref).max():.4f} meanerr {np.abs(xx-ref).mean():.5f}', flush=True)
        if np.linalg.norm(r) <= tol * b2: break
    return x, it

if __name__ == '__main__':
    g = np.load('tests/golden/g8_c1.npz')
    cond = orc.get_above_threshold_speed(g['orograph_f32'], 0.75); ref = g['potential'].astype(float)
    R, C = cond.shape
    A, rhs, fixed, val = setup(cond, 0.)
    t = time.time(); lv = hierarchy(A); print('levels', [a.shape[0] for a, _ in lv], 'setup', round(time.time() - t, 1))
    for kc in (False, True):
        M = make_cycle(lv, nu=2, omega=0.7, kcycle=kc)
        t = time.time(); x, it = fpcg(A, rhs, M, 100 if not kc else 60, ref, (R, C), fixed, val)
        print('kcycle', kc, 'its', it, 'time', round(time.time() - t, 1))
