"""Scratch experiment 8 (CPU, scipy; round 4): SMOOTHED aggregation on strength-aware aggregates for the
potential system (candidate i): P = (I - w D_F^-1 A_F) P_tent, A_F = A with the weak
links lumped into the diagonal, Galerkin coarse operators.  Aggregates: pairwise matching passes (what the
GPU hierarchy builds) or greedy root + strong neighbours.

usage: python tests/dev/attic/sa_experiment.py c1 | g10 | g11 | synth ROWS COLS [RES] | speckle F"""
import sys, time
sys.path.insert(0, '.')
import numpy as np, scipy.sparse as sp, scipy.sparse.linalg as spl
from tests.dev.attic.amg_experiment2 import setup
from tests.dev.attic.amg_experiment4 import hash32
from tests.dev.attic.boxmg_experiment import load, pcg


def strong_mask(A, theta):
    """COO of the off-diagonal entries with a flag: a_ij >= theta * sqrt(a_ii a_jj)."""
    d = A.diagonal()
    S = A.tocoo()
    off = S.row != S.col
    i, j, w = S.row[off], S.col[off], -S.data[off]
    strong = w >= theta * np.sqrt(d[i] * d[j])
    return i, j, w, strong, d


def greedy_aggregates(n, i, j, strong):
    """Vanek-style: roots whose strong neighbourhood is free, then leftovers join a neighbour's aggregate."""
    G = sp.csr_matrix((np.ones(strong.sum()), (i[strong], j[strong])), shape=(n, n))
    indptr, indices = G.indptr, G.indices
    agg = np.full(n, -1)
    na = 0
    order = np.argsort(hash32(np.arange(n), np.arange(n)[::-1]))        # a fixed pseudo-random order
    for v in order:
        if agg[v] >= 0:
            continue
        nb = indices[indptr[v]:indptr[v + 1]]
        if nb.size and np.all(agg[nb] < 0):
            agg[v] = na; agg[nb] = na; na += 1
    for v in np.where(agg < 0)[0]:
        nb = indices[indptr[v]:indptr[v + 1]]
        nb = nb[agg[nb] >= 0]
        if nb.size:
            agg[v] = -2 - agg[nb[0]]
    m = agg <= -2
    agg[m] = -2 - agg[m]
    for v in np.where(agg < 0)[0]:
        agg[v] = na; na += 1
    return agg, na


def pairwise_aggregates(A, theta, passes):
    from tests.dev.attic.amg_experiment6 import match_pass
    n = A.shape[0]; agg = np.arange(n); Ac = A
    dfine = A.diagonal()
    for p in range(passes):
        inv, nc = match_pass(Ac, theta=theta * 8, symmetric=True)
        P1 = sp.csr_matrix((np.ones(Ac.shape[0]), (np.arange(Ac.shape[0]), inv)), shape=(Ac.shape[0], nc))
        Ac = (P1.T @ Ac @ P1).tocsr(); agg = inv[agg]
        # strength of later passes against the summed FINE diagonals (round-1 fix: a collapsed floating cluster
        # must not look weakly anchored and pair with a dead neighbour)
        dsum = np.bincount(agg, weights=dfine, minlength=nc)
        Ac = Ac + sp.diags(dsum - Ac.diagonal())
    return agg, Ac.shape[0]


def hierarchy(A, theta=0.02, omega=2. / 3, min_n=500, agg_kind='greedy', passes=2, smooth=True, max_levels=30):
    levels = []
    while A.shape[0] > min_n and len(levels) < max_levels:
        n = A.shape[0]
        i, j, w, strong, d = strong_mask(A, theta)
        if agg_kind == 'greedy':
            agg, nc = greedy_aggregates(n, i, j, strong)
        else:
            agg, nc = pairwise_aggregates(A, theta, passes)
        if nc > 0.8 * n:
            break
        T = sp.csr_matrix((np.ones(n), (np.arange(n), agg)), shape=(n, nc))
        if smooth:
            weak_sum = np.bincount(i[~strong], weights=w[~strong], minlength=n)      # weak links lumped: rows keep their sums
            dF = d - weak_sum
            AF = sp.csr_matrix((np.r_[-w[strong], dF], (np.r_[i[strong], np.arange(n)], np.r_[j[strong], np.arange(n)])), shape=(n, n))
            P = (T - sp.diags(omega / dF) @ (AF @ T)).tocsr()
        else:
            P = T
        levels.append((A, P))
        A = (P.T @ A @ P).tocsr()
    levels.append((A, None))
    return levels


def make_cycle(levels, nu=2, omega=0.7):
    dinv = [1.0 / A.diagonal() for A, _ in levels]
    lu = spl.splu(levels[-1][0].tocsc())

    def cyc(l, b):
        A, P = levels[l]
        if P is None:
            return lu.solve(b)
        x = np.zeros(b.shape)
        for _ in range(nu):
            x += omega * dinv[l] * (b - A @ x)
        x += P @ cyc(l + 1, P.T @ (b - A @ x))
        for _ in range(nu):
            x += omega * dinv[l] * (b - A @ x)
        return x
    return lambda b: cyc(0, b)


def run(cond, label, variants):
    A, rhs, fixed, val = setup(cond, 0.)
    print(f'{label}: unknowns {A.shape[0]}, zero-conductivity cells {float((cond <= 0).mean()):.2f}', flush=True)
    for name, hkw, ckw in variants:
        t = time.time(); lv = hierarchy(A, **hkw); ts = time.time() - t
        nnz = [a.nnz for a, _ in lv]
        M = make_cycle(lv, **ckw)
        t = time.time(); x, it, hist = pcg(A, rhs, M, 400)
        k8 = next((k + 1 for k, h in enumerate(hist) if h <= 1e-8), None)
        print(f'   {name}: levels {[a.shape[0] for a, _ in lv]} op complexity {sum(nnz) / nnz[0]:.2f} '
              f'(P nnz/row {lv[0][1].nnz / lv[0][1].shape[0]:.1f}) setup {ts:.1f}s | {it} its to {hist[-1]:.1e} ({k8} to 1e-8) {time.time() - t:.1f}s', flush=True)


if __name__ == '__main__':
    which = sys.argv[1] if len(sys.argv) > 1 else 'c1'
    if which == 'speckle':
        rng = np.random.default_rng(0)
        cond = np.abs(rng.normal(.8, .6, (200, 240))); cond[rng.random(cond.shape) < float(sys.argv[2])] = 0
    else:
        cond = load(which, sys.argv[2:])
    run(cond, which, [
        ('UA pairwise x1 V(2,2)', dict(agg_kind='pairwise', passes=1, smooth=False), dict(nu=2)),
        ('SA greedy th .02 V(1,1)', dict(agg_kind='greedy', theta=0.02), dict(nu=1)),
        ('SA greedy th .02 V(2,2)', dict(agg_kind='greedy', theta=0.02), dict(nu=2)),
        ('SA greedy th .05 V(2,2)', dict(agg_kind='greedy', theta=0.05), dict(nu=2)),
        ('SA pairwise x2 th .03 V(2,2)', dict(agg_kind='pairwise', passes=2, theta=0.03), dict(nu=2)),
    ])
