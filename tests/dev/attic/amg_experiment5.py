"""Scratch experiment 5 (CPU, scipy): what would cut the iteration count of the shipped
pairwise-aggregation V-cycle on the C1 potential system?  Over-correction of the coarse
grid correction, Chebyshev / l1-Jacobi / Gauss-Seidel smoothing, aggregate size."""
import sys, time
sys.path.insert(0, '.')
import numpy as np, scipy.sparse as sp, scipy.sparse.linalg as spl
from oracle import ssrs_oracle as orc
from tests.dev.attic.amg_experiment2 import setup
from tests.dev.attic.amg_experiment4 import hierarchy


def make_cycle(levels, nu=2, omega=0.7, alpha=1.0, smoother='jacobi', cheb_deg=2):
    dinv = [1.0 / A.diagonal() for A, _ in levels]
    l1 = [1.0 / np.asarray(abs(A).sum(1)).ravel() for A, _ in levels]
    lu = spl.splu(levels[-1][0].tocsc())
    lam = []
    if smoother == 'cheb':
        for (A, _), d in zip(levels, dinv):
            v = np.random.default_rng(0).random(A.shape[0])
            for _ in range(15):
                v = d * (A @ v); v /= np.linalg.norm(v)
            lam.append(1.1 * (v @ (d * (A @ v))))
    tri = []
    if smoother == 'gs':
        for A, _ in levels:
            tri.append((sp.tril(A).tocsr(), sp.triu(A).tocsr()))

    def smooth(l, x, b, post):
        A = levels[l][0]
        if smoother == 'jacobi':
            for _ in range(nu):
                x = x + omega * dinv[l] * (b - A @ x)
        elif smoother == 'l1':
            for _ in range(nu):
                x = x + l1[l] * (b - A @ x)
        elif smoother == 'cheb':
            lmax, lmin = lam[l], lam[l] / 8.0
            theta, delta = 0.5 * (lmax + lmin), 0.5 * (lmax - lmin)
            sigma = theta / delta; rho = 1.0 / sigma
            r = b - A @ x; d = dinv[l] * r / theta; x = x + d
            for _ in range(cheb_deg - 1):
                rho_new = 1.0 / (2.0 * sigma - rho)
                r = b - A @ x
                d = rho_new * rho * d + 2.0 * rho_new / delta * (dinv[l] * r)
                x = x + d; rho = rho_new
        elif smoother == 'gs':
            L, U = tri[l]
            for _ in range(nu):
                if not post:
                    x = x + spl.spsolve_triangular(L, b - A @ x, lower=True)
                else:
                    x = x + spl.spsolve_triangular(U, b - A @ x, lower=False)
        return x

    def cyc(l, b):
        A, P = levels[l]
        if P is None:
            return lu.solve(b)
        x = smooth(l, np.zeros(b.shape), b, False)
        ec = cyc(l + 1, P.T @ (b - A @ x))
        x = x + alpha * (P @ ec)
        return smooth(l, x, b, True)
    return lambda b: cyc(0, b)


def fpcg(A, b, M, maxit, tol=1e-8):
    x = np.zeros_like(b); r = b.copy(); pold = None; b2 = np.linalg.norm(b)
    for it in range(1, maxit + 1):
        z = M(r)
        p = z if pold is None else z - ((z @ Apold) / (pold @ Apold)) * pold
        Ap = A @ p; a = (p @ r) / (p @ Ap); x += a * p; r -= a * Ap; pold, Apold = p, Ap
        if np.linalg.norm(r) <= tol * b2:
            break
    return x, it, np.linalg.norm(r) / b2


if __name__ == '__main__':
    which = sys.argv[1] if len(sys.argv) > 1 else 'c1'
    if which == 'c1':
        g = np.load('tests/golden/g8_c1.npz')
        cond = orc.get_above_threshold_speed(g['orograph_f32'], 0.75)
    else:                                   # synthetic-DEM raster of the given shape at 10 m
        from ssrs_amd.synthetic import synthetic_dem
        rows, cols = int(sys.argv[2]), int(sys.argv[3])
        z = synthetic_dem((rows, cols), 10.)
        oro = orc.compute_orographic_updraft(10., 270., orc.compute_slope_degrees(z, 10.),
                                             orc.compute_aspect_degrees(z, 10.)).astype(np.float32)
        cond = orc.get_above_threshold_speed(oro, 0.75)
    A, rhs, fixed, val = setup(cond, 0.)
    print('unknowns', A.shape[0], 'dead fraction', float((cond <= 0).mean()))
    for passes in (2,):
        t = time.time(); lv = hierarchy(A, passes=passes, theta=0.25)
        print('passes', passes, 'levels', [a.shape[0] for a, _ in lv], 'setup', round(time.time() - t, 1), flush=True)
        variants = [dict(), dict(alpha=1.4), dict(alpha=1.8), dict(smoother='l1'),
                    dict(smoother='cheb', cheb_deg=2), dict(smoother='cheb', cheb_deg=3),
                    dict(smoother='cheb', cheb_deg=3, alpha=1.5), dict(smoother='gs', nu=1), dict(smoother='gs', nu=1, alpha=1.5)]
        for kw in variants:
            M = make_cycle(lv, **kw)
            t = time.time(); x, it, rr = fpcg(A, rhs, M, 600)
            print('  ', kw, 'its', it, 'relres', f'{rr:.1e}', 'time', round(time.time() - t, 1), flush=True)
