"""Scratch experiment 6 (CPU, scipy): aggregation criteria for the two-phase (live / dead)
conductance rasters: one-sided vs symmetric strength of connection, aggregate size,
over-correction, K-cycle."""
import sys, time
sys.path.insert(0, '.')
import numpy as np, scipy.sparse as sp, scipy.sparse.linalg as spl
from oracle import ssrs_oracle as orc
from tests.dev.attic.amg_experiment2 import setup
from tests.dev.attic.amg_experiment4 import hash32
from tests.dev.attic.amg_experiment5 import fpcg


def match_pass(A, rounds=8, theta=0.25, symmetric=False, permissive=0):
    m = A.shape[0]
    d = A.diagonal()
    S = -(A - sp.diags(d)).tocoo()
    keep = S.data > 0
    i, j, w = S.row[keep], S.col[keep], S.data[keep]
    if symmetric:
        strong = w >= theta * np.sqrt(d[i] * d[j]) / 8.0      # a_ij >= theta * sqrt(a_ii a_jj) / degree
    else:
        rmax = np.zeros(m); np.maximum.at(rmax, i, w)
        strong = w >= theta * rmax[i]
    lo, hi = np.minimum(i, j), np.maximum(i, j)
    pri_all = w * (1.0 + 1e-3 * hash32(lo, hi))
    match = np.full(m, -1)
    for rnd in range(rounds + permissive):
        sel = strong if rnd < rounds else np.ones_like(strong)
        ok = sel & (match[i] < 0) & (match[j] < 0)
        if not ok.any():
            continue
        ii, jj, pp = i[ok], j[ok], pri_all[ok]
        order = np.lexsort((-pp, ii))
        first = np.r_[True, ii[order][1:] != ii[order][:-1]]
        prop = np.full(m, -1); prop[ii[order][first]] = jj[order][first]
        cand = np.where(prop >= 0)[0]
        mutual = cand[prop[prop[cand]] == cand]
        match[mutual] = prop[mutual]
    cid = np.where(match >= 0, np.minimum(np.arange(m), match), np.arange(m))
    uniq, inv = np.unique(cid, return_inverse=True)
    return inv, uniq.size


def hierarchy(A, min_n=1000, passes=2, **kw):
    levels = []
    while A.shape[0] > min_n and len(levels) < 40:
        n = A.shape[0]; agg = np.arange(n); Ac = A
        for _ in range(passes):
            inv, nc = match_pass(Ac, **kw)
            P1 = sp.csr_matrix((np.ones(Ac.shape[0]), (np.arange(Ac.shape[0]), inv)), shape=(Ac.shape[0], nc))
            Ac = (P1.T @ Ac @ P1).tocsr(); agg = inv[agg]
        if Ac.shape[0] > 0.9 * n:
            if not kw.get('permissive'):
                kw = dict(kw, permissive=4); continue      # stalled: allow weak pairs from here on
            break
        P = sp.csr_matrix((np.ones(n), (np.arange(n), agg)), shape=(n, Ac.shape[0]))
        levels.append((A, P)); A = Ac
    levels.append((A, None))
    return levels


def make_cycle(levels, nu=2, omega=0.7, alpha=1.0, kcycle=0):
    dinv = [1.0 / A.diagonal() for A, _ in levels]
    last = levels[-1][0]
    lu = spl.splu(last.tocsc()) if last.shape[0] < 200000 else None

    def smooth(l, x, b):
        A = levels[l][0]
        for _ in range(nu):
            x = x + omega * dinv[l] * (b - A @ x)
        return x

    def cyc(l, b):
        A, P = levels[l]
        if P is None:
            if lu is not None:
                return lu.solve(b)
            x = np.zeros_like(b)
            for _ in range(30):
                x = x + omega * dinv[l] * (b - A @ x)
            return x
        x = smooth(l, np.zeros(b.shape), b)
        rc = P.T @ (b - A @ x)
        if kcycle and 0 < l + 1 <= kcycle and levels[l + 1][1] is not None:
            Ac = levels[l + 1][0]
            ec = np.zeros_like(rc); r = rc.copy(); pold = None
            for k in range(2):
                z = cyc(l + 1, r)
                p = z if pold is None else z - ((z @ Apold) / (pold @ Apold)) * pold
                Ap = Ac @ p; a = (p @ r) / (p @ Ap)
                ec += a * p; r -= a * Ap; pold, Apold = p, Ap
        else:
            ec = alpha * cyc(l + 1, rc)
        x = x + P @ ec
        return smooth(l, x, b)
    return lambda b: cyc(0, b)


if __name__ == '__main__':
    which = sys.argv[1] if len(sys.argv) > 1 else 'c1'
    if which == 'c1':
        g = np.load('tests/golden/g8_c1.npz')
        cond = orc.get_above_threshold_speed(g['orograph_f32'], 0.75)
    else:
        from ssrs_amd.synthetic import synthetic_dem
        rows, cols = int(sys.argv[2]), int(sys.argv[3])
        z = synthetic_dem((rows, cols), 10.)
        oro = orc.compute_orographic_updraft(10., 270., orc.compute_slope_degrees(z, 10.),
                                             orc.compute_aspect_degrees(z, 10.)).astype(np.float32)
        cond = orc.get_above_threshold_speed(oro, 0.75)
    A, rhs, fixed, val = setup(cond, 0.)
    print('unknowns', A.shape[0], flush=True)
    configs = [
        ('symmetric th .25 p2, permissive after stall', dict(passes=2, theta=0.25, symmetric=True)),
        ('symmetric th .25 p1, permissive after stall', dict(passes=1, theta=0.25, symmetric=True)),
        ('symmetric th .10 p2, permissive after stall', dict(passes=2, theta=0.10, symmetric=True)),
    ]
    for name, kw in configs:
        t = time.time(); lv = hierarchy(A, **kw)
        print(name, 'levels', [a.shape[0] for a, _ in lv], 'setup', round(time.time() - t, 1), flush=True)
        for ckw in (dict(), dict(alpha=1.4), dict(kcycle=2), dict(kcycle=3), dict(kcycle=5)):
            M = make_cycle(lv, **ckw)
            t = time.time(); x, it, rr = fpcg(A, rhs, M, 500)
            print('   ', ckw, 'its', it, 'relres', f'{rr:.1e}', 'time', round(time.time() - t, 1), flush=True)
