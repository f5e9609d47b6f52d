"""Scratch experiment 2: free-cell formulation (Dirichlet cells become ground
links), CG with an aggregation V-cycle; uniform vs real conductivity."""
import sys, time
sys.path.insert(0, '.')
import numpy as np, scipy.sparse as sp, scipy.sparse.linalg as spl
from oracle import ssrs_oracle as orc
from tests.dev.attic.amg_experiment import build_system, gs_colors

def setup(cond, dirn):
    R, C = cond.shape; n = R * C
    Cs, fixed, val = build_system(cond, dirn, quirk=False)
    free = ~fixed
    F = sp.diags(free.astype(float))
    Cff = (F @ Cs @ F).tocsr()
    ground = np.asarray((F @ Cs @ sp.diags(fixed.astype(float))).sum(1)).ravel()
    rhs = np.asarray(F @ Cs @ (val * fixed)).ravel()
    deg = np.asarray(Cff.sum(1)).ravel() + ground
    deg[fixed] = 1.0                      # inactive cells: identity
    A = (sp.diags(deg) - Cff).tocsr()
    return A, rhs, fixed, val

def hierarchy(A, shape, active, min_cells=200):
    levels = []; R, C = shape
    while R * C > min_cells:
        R2, C2 = (R + 1) // 2, (C + 1) // 2
        r, c = np.divmod(np.arange(R * C), C)
        agg = (r // 2) * C2 + (c // 2)
        w = active.astype(float)                       # inactive cells do not interpolate
        P = sp.csr_matrix((w, (np.arange(R * C), agg)), shape=(R * C, R2 * C2))
        levels.append((A, P, (R, C)))
        Ac = (P.T @ A @ P).tolil()
        act2 = np.asarray(P.T @ w).ravel() > 0
        d = Ac.diagonal(); d[~act2] = 1.0; Ac.setdiag(d)
        A = Ac.tocsr(); active = act2; R, C = R2, C2
    levels.append((A, None, (R, C)))
    return levels

def make_vcycle(levels, nu=1, gamma=1):
    cols = [gs_colors(s) for _, _, s in levels]
    diags = [A.diagonal() for A, _, _ in levels]
    lu = spl.splu(levels[-1][0].tocsc())
    def smooth(l, x, b, rev):
        A = levels[l][0]; d = diags[l]
        for cs in (cols[l][::-1] if rev else cols[l]):
            x[cs] += (b[cs] - A[cs] @ x) / d[cs]
    def cyc(l, b):
        A, P, _ = levels[l]
        if P is None: return lu.solve(b)
        x = np.zeros(b.shape)
        for _ in range(nu): smooth(l, x, b, False)
        for _ in range(gamma if l > 0 else 1):
            r = b - A @ x
            x += P @ cyc(l + 1, P.T @ r)
        for _ in range(nu): smooth(l, x, b, True)
        return x
    return lambda b: cyc(0, b)

def pcg(A, b, M, tol, maxit, ref=None, shape=None, fixed=None, val=None):
    x = np.zeros_like(b); r = b.copy(); z = M(r); p = z.copy(); rz = r @ z
    b2 = np.linalg.norm(b)
    for it in range(1, maxit + 1):
        Ap = A @ p; a = rz / (p @ Ap); x += a * p; r -= a * Ap
        if it % 10 == 0 or np.linalg.norm(r) <= tol * b2:
            xx = np.where(fixed, val, x).reshape(shape)
            print(f'   it {it:4d} relres {np.linalg.norm(r)/b2:.2e} maxerr {np.abs(xx-ref).max():.4f}', flush=True)
        if np.linalg.norm(r) <= tol * b2: break
        z = M(r); rz2 = r @ z; p = z + (rz2 / rz) * p; rz = rz2
    return x, it

if __name__ == '__main__':
    g = np.load('tests/golden/g8_c1.npz')
    cond = orc.get_above_threshold_speed(g['orograph_f32'], 0.75); ref = g['potential'].astype(float)
    R, C = cond.shape
    for name, cnd, rf in [('uniform', np.ones_like(cond), np.broadcast_to(1000 * (1 - np.arange(R)[:, None] / (R - 1.)), (R, C))),
                          ('real', cond, ref)]:
        A, rhs, fixed, val = setup(cnd, 0.)
        t = time.time(); lv = hierarchy(A, (R, C), ~fixed)
        print(name, 'levels', len(lv), 'setup', round(time.time() - t, 2))
        for gamma in (1, 2):
            M = make_vcycle(lv, nu=1, gamma=gamma)
            t = time.time()
            x, it = pcg(A, rhs, M, 1e-12, 150 if name == 'real' else 60, rf, (R, C), fixed, val)
            print(f' gamma {gamma}: its {it} time {time.time()-t:.1f}')
