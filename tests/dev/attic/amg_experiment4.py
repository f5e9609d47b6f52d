"""Scratch experiment 4: handshake matching with hashed tie-breaks (what the GPU
kernels would do), full hierarchy down to ~1000 nodes, V-cycle + Jacobi, PCG."""
import sys, time
sys.path.insert(0, '.')
import numpy as np, scipy.sparse as sp, scipy.sparse.linalg as spl
from oracle import ssrs_oracle as orc
from tests.dev.attic.amg_experiment2 import setup
from tests.dev.attic.amg_experiment3 import fpcg, make_cycle

def hash32(a, b):
    x = (a.astype(np.uint64) * np.uint64(0x9E3779B97F4A7C15) ^ b.astype(np.uint64) * np.uint64(0xC2B2AE3D27D4EB4F)) & np.uint64(0xFFFFFFFF)
    return x.astype(np.float64) / 4294967296.0

def match_pass(A, rounds=8, theta=0.25):
    m = A.shape[0]
    S = -(A - sp.diags(A.diagonal())).tocoo()
    keep = S.data > 0
    i, j, w = S.row[keep], S.col[keep], S.data[keep]
    # relative strength: only neighbours >= 0.25 * row max are candidates
    rmax = np.zeros(m); np.maximum.at(rmax, i, w)
    strong = w >= theta * rmax[i]
    i, j, w = i[strong], j[strong], w[strong]
    lo, hi = np.minimum(i, j), np.maximum(i, j)
    pri = w * (1.0 + 1e-3 * hash32(lo, hi))      # symmetric priority with tie-break
    match = np.full(m, -1)
    for _ in range(rounds):
        ok = (match[i] < 0) & (match[j] < 0)
        if not ok.any(): break
        ii, jj, pp = i[ok], j[ok], pri[ok]
        order = np.lexsort((-pp, ii))
        first = np.r_[True, ii[order][1:] != ii[order][:-1]]
        prop = np.full(m, -1); prop[ii[order][first]] = jj[order][first]
        cand = np.where(prop >= 0)[0]
        mutual = cand[prop[prop[cand]] == cand]
        match[mutual] = prop[mutual]
    cid = np.where(match >= 0, np.minimum(np.arange(m), match), np.arange(m))
    uniq, inv = np.unique(cid, return_inverse=True)
    return inv, uniq.size

def hierarchy(A, min_n=1000, passes=2, theta=0.25):
    levels = []
    while A.shape[0] > min_n and len(levels) < 40:
        n = A.shape[0]; agg = np.arange(n); Ac = A
        for _ in range(passes):
            inv, nc = match_pass(Ac, theta=theta)
            P1 = sp.csr_matrix((np.ones(Ac.shape[0]), (np.arange(Ac.shape[0]), inv)), shape=(Ac.shape[0], nc))
            Ac = (P1.T @ Ac @ P1).tocsr(); agg = inv[agg]
        if Ac.shape[0] > 0.85 * n:
            if theta > 0: theta = 0.0; continue
            break
        P = sp.csr_matrix((np.ones(n), (np.arange(n), agg)), shape=(n, Ac.shape[0]))
        levels.append((A, P)); A = Ac
    levels.append((A, None))
    return levels

if __name__ == '__main__':
    g = np.load('tests/golden/g8_c1.npz')
    cond = orc.get_above_threshold_speed(g['orograph_f32'], 0.75); ref = g['potential'].astype(float)
    R, C = cond.shape
    A, rhs, fixed, val = setup(cond, 0.)
    for passes, theta in ((1, 0.25),):
        t = time.time(); lv = hierarchy(A, passes=passes, theta=theta)
        print('passes', passes, 'theta', theta, 'levels', [a.shape[0] for a, _ in lv], 'setup', round(time.time() - t, 1))
        for nu, om in ((2, 0.7),):
            M = make_cycle(lv, nu=nu, omega=om, kcycle=False)
            t = time.time(); x, it = fpcg(A, rhs, M, 300, ref, (R, C), fixed, val, tol=1e-12)
            print('  V-cycle nu', nu, 'omega', om, 'its', it, 'time', round(time.time() - t, 1))
