"""Scratch experiment (not product): does a structured 2x2-aggregation multigrid
preconditioner make a Krylov solve of the SSRS potential system converge on the
C1 raster?  scipy on the CPU, to decide the GPU solver design."""
import sys, time
sys.path.insert(0, '.')
import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spl
from oracle import ssrs_oracle as orc

def build_system(cond, dirn, quirk=True):
    R, C = cond.shape
    n = R * C
    idx = np.arange(n).reshape(R, C)
    bn, be = orc.get_boundary_nodes(dirn, (R, C))
    fixed = np.zeros((R, C), bool); val = np.zeros((R, C))
    fixed[bn % R, bn // R] = True; val[bn % R, bn // R] = be
    rows, cols, w = [], [], []
    for dr, dc in [(-1,-1),(-1,0),(-1,1),(0,-1),(0,1),(1,-1),(1,0),(1,1)]:
        r0, r1 = max(0, -dr), min(R, R - dr); c0, c1 = max(0, -dc), min(C, C - dc)
        a = cond[r0:r1, c0:c1]; b = cond[r0+dr:r1+dr, c0+dc:c1+dc]
        with np.errstate(divide='ignore'):
            hm = np.where((a != 0) & (b != 0), 2. / (1. / a + 1. / b), 1e-8)
        diag = (dr != 0 and dc != 0)
        fac = np.full(hm.shape, np.float64(np.float32(np.sqrt(2.))) if diag else 1.0)
        if quirk and dr == -1:     # east-edge interior rows: S gets sqrt2, SW gets 1
            rr = np.arange(r0, r1)[:, None]; cc = np.arange(c0, c1)[None, :]
            q = (cc == C - 1) & (rr > 0) & (rr < R - 1)
            fac = np.where(q, np.float64(np.float32(np.sqrt(2.))) if not diag else 1.0, fac)
        rows.append(idx[r0:r1, c0:c1].ravel()); cols.append(idx[r0+dr:r1+dr, c0+dc:c1+dc].ravel())
        w.append((hm / fac).ravel())
    Cm = sp.csr_matrix((np.concatenate(w), (np.concatenate(rows), np.concatenate(cols))), shape=(n, n))
    return Cm, fixed.ravel(), val.ravel()

def hierarchy(L, shape, min_size=16):
    """L: SPD matrix on the full raster numbering restricted to free cells handled by caller.
    Here: levels on structured grid with 2x2 aggregation of *all* cells (fixed cells have unit rows)."""
    levels = []
    A = L; R, C = shape
    while R * C > min_size * min_size and len(levels) < 14:
        R2, C2 = (R + 1) // 2, (C + 1) // 2
        r, c = np.divmod(np.arange(R * C), C)
        agg = (r // 2) * C2 + (c // 2)
        P = sp.csr_matrix((np.ones(R * C), (np.arange(R * C), agg)), shape=(R * C, R2 * C2))
        levels.append((A, P, (R, C)))
        A = (P.T @ A @ P).tocsr()
        R, C = R2, C2
    levels.append((A, None, (R, C)))
    return levels

def gs_colors(shape):
    R, C = shape
    r, c = np.divmod(np.arange(R * C), C)
    return [np.where((r % 2 == a) & (c % 2 == b))[0] for a in (0, 1) for b in (0, 1)]

def make_vcycle(levels, nu=1, omega_c=1.0):
    cols = [gs_colors(s) for _, _, s in levels]
    diags = [A.diagonal() for A, _, _ in levels]
    coarse_lu = spl.splu(levels[-1][0].tocsc())
    def smooth(l, x, b, reverse=False):
        A = levels[l][0]; d = diags[l]
        order = cols[l][::-1] if reverse else cols[l]
        for cset in order:
            r = b[cset] - A[cset] @ x
            x[cset] += r / d[cset]
        return x
    def cyc(l, b):
        A, P, _ = levels[l]
        if P is None:
            return coarse_lu.solve(b)
        x = np.zeros(b.shape, dtype=np.float64)
        for _ in range(nu): smooth(l, x, b)
        r = b - A @ x
        ec = cyc(l + 1, P.T @ r)
        x += omega_c * (P @ ec)
        for _ in range(nu): smooth(l, x, b, reverse=True)
        return x
    return lambda b: cyc(0, b)

if __name__ == '__main__':
    which = sys.argv[1] if len(sys.argv) > 1 else 'c1'
    if which == 'c1':
        g = np.load('tests/golden/g8_c1.npz')
        cond = orc.get_above_threshold_speed(g['orograph_f32'], 0.75); ref = g['potential']
    else:
        g = np.load('tests/golden/g5_potential.npz'); cond = g['updraft']; ref = g['pot_d0']
    R, C = cond.shape
    Cm, fixed, val = build_system(cond, 0., quirk=True)
    Cs, _, _ = build_system(cond, 0., quirk=False)
    free = ~fixed
    D = np.asarray(Cm.sum(1)).ravel(); Ds = np.asarray(Cs.sum(1)).ravel()
    # exact (quirk) operator in symmetric-like form  L = D - C  on free rows, identity on fixed rows
    def assemble(Cmat, Dv):
        Lm = sp.diags(Dv) - Cmat
        F = sp.diags(free.astype(float)); X = sp.diags(fixed.astype(float))
        return (F @ Lm @ F + X).tocsr(), (F @ Lm @ X)
    L, Lfb = assemble(Cm, D)
    Ls, _ = assemble(Cs, Ds)
    b = np.asarray(-(Lfb @ val), dtype=np.float64); b[fixed] = val[fixed]
    print('n', R * C, 'nnz', L.nnz, 'asym', abs(L - L.T).max())
    t = time.time(); levels = hierarchy(Ls, (R, C)); print('levels', [s for _, _, s in levels], time.time() - t)
    for oc in (1.0, 1.5):
        M = make_vcycle(levels, nu=1, omega_c=oc)
        it = [0]
        def cb(xk): it[0] += 1
        t = time.time()
        x, info = spl.gmres(L, b, M=spl.LinearOperator(L.shape, matvec=M), rtol=1e-13, restart=60, maxiter=10, callback=cb, callback_type='pr_norm')
        res = np.linalg.norm(b - L @ x) / np.linalg.norm(b)
        err = np.abs(x.reshape(R, C) - ref)
        print(f'omega_c {oc}: gmres its {it[0]} info {info} relres {res:.2e} maxerr {err.max():.4f} meanerr {err.mean():.5f} time {time.time()-t:.1f}')
