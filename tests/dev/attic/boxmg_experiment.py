"""Scratch experiment 7 (CPU, scipy; round 4): black-box multigrid (Dendy) for the potential system --
standard 2 x 2 coarsening of the raster, OPERATOR-DEPENDENT interpolation from the 9-point stencil,
Galerkin coarse operators (which stay 9-point stencils: every level can be matrix-free like level 0).
Question: how many PCG iterations to 1e-15 on the two-phase (live / dead, 1e-8 links) rasters, against
the ~400 of the shipped pairwise aggregation?

usage: python tests/dev/attic/boxmg_experiment.py c1 | g10 | g11 | synth ROWS COLS [RES] | soak N"""
import sys, time
sys.path.insert(0, '.')
import numpy as np, scipy.sparse as sp, scipy.sparse.linalg as spl
from oracle import ssrs_oracle as orc
from tests.dev.attic.amg_experiment2 import setup


def stencil(A, shape):
    """Positive off-diagonal weights of a symmetric 9-point matrix as rasters: E, N, NE, NW of each cell."""
    R, C = shape
    n = R * C

    def off(k):
        d = np.zeros(n)
        v = -A.diagonal(k)
        d[:v.size] = v
        return d.reshape(R, C)
    wE, wN, wNE, wNW = off(1), off(C), off(C + 1), off(C - 1)
    wE[:, -1] = 0; wN[-1, :] = 0; wNE[-1, :] = 0; wNE[:, -1] = 0; wNW[-1, :] = 0; wNW[:, 0] = 0
    return A.diagonal().reshape(R, C), wE, wN, wNE, wNW


def shift(a, dr, dc):
    """b[r, c] = a[r + dr, c + dc], zero outside."""
    R, C = a.shape
    b = np.zeros_like(a)
    r0, r1 = max(0, -dr), min(R, R - dr)
    c0, c1 = max(0, -dc), min(C, C - dc)
    b[r0:r1, c0:c1] = a[r0 + dr:r1 + dr, c0 + dc:c1 + dc]
    return b


def interpolation(A, shape, mode='dendy'):
    R, C = shape
    Rc, Cc = (R + 1) // 2, (C + 1) // 2
    d, wE, wN, wNE, wNW = stencil(A, shape)
    wW, wS, wSW, wSE = shift(wE, 0, -1), shift(wN, -1, 0), shift(wNE, -1, -1), shift(wNW, -1, 1)
    idx = np.arange(R * C).reshape(R, C)
    cidx = lambda r, c: (r // 2) * Cc + (c // 2)
    rows, cols, vals = [], [], []
    rr, cc = np.meshgrid(np.arange(R), np.arange(C), indexing='ij')
    er, ec = rr % 2 == 0, cc % 2 == 0
    # coarse points
    m = er & ec
    rows.append(idx[m]); cols.append(cidx(rr[m], cc[m])); vals.append(np.ones(m.sum()))
    # horizontal edge points (coarse row, fine column): from W and E
    aW, aE = wW + wNW + wSW, wE + wNE + wSE
    den_h = d - wN - wS
    pW = np.where(den_h > 0, aW / np.where(den_h > 0, den_h, 1), 0)
    pE = np.where(den_h > 0, aE / np.where(den_h > 0, den_h, 1), 0)
    # vertical edge points (fine row, coarse column): from S and N
    aS, aN = wS + wSW + wSE, wN + wNE + wNW
    den_v = d - wW - wE
    pS = np.where(den_v > 0, aS / np.where(den_v > 0, den_v, 1), 0)
    pN = np.where(den_v > 0, aN / np.where(den_v > 0, den_v, 1), 0)
    m = er & ~ec
    ok = m & (cc + 1 < C)
    rows.append(idx[m]); cols.append(cidx(rr[m], cc[m] - 1)); vals.append(pW[m])
    rows.append(idx[ok]); cols.append(cidx(rr[ok], cc[ok] + 1)); vals.append(pE[ok])
    m = ~er & ec
    ok = m & (rr + 1 < R)
    rows.append(idx[m]); cols.append(cidx(rr[m] - 1, cc[m])); vals.append(pS[m])
    rows.append(idx[ok]); cols.append(cidx(rr[ok] + 1, cc[ok])); vals.append(pN[ok])
    # cell centres (fine row, fine column): the stencil equation with the edge neighbours interpolated
    m = ~er & ~ec
    pW_n, pE_n = shift(pW, 1, 0), shift(pE, 1, 0)        # of the N neighbour (a horizontal edge point)
    pW_s, pE_s = shift(pW, -1, 0), shift(pE, -1, 0)
    pS_w, pN_w = shift(pS, 0, -1), shift(pN, 0, -1)      # of the W neighbour (a vertical edge point)
    pS_e, pN_e = shift(pS, 0, 1), shift(pN, 0, 1)
    dd = np.where(d > 0, d, 1)
    cSW = (wSW + wS * pW_s + wW * pS_w) / dd
    cSE = (wSE + wS * pE_s + wE * pS_e) / dd
    cNW = (wNW + wN * pW_n + wW * pN_w) / dd
    cNE = (wNE + wN * pE_n + wE * pN_e) / dd
    for coef, dr, dc in ((cSW, -1, -1), (cSE, -1, 1), (cNW, 1, -1), (cNE, 1, 1)):
        ok = m & (rr + dr < R) & (cc + dc < C)
        rows.append(idx[ok]); cols.append(cidx(rr[ok] + dr, cc[ok] + dc)); vals.append(coef[ok])
    P = sp.csr_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(R * C, Rc * Cc))
    return P, (Rc, Cc)


def hierarchy(A, shape, min_cells=400):
    levels = []
    while shape[0] * shape[1] > min_cells and min(shape) > 4:
        P, cshape = interpolation(A, shape)
        levels.append((A, P, shape))
        A = (P.T @ A @ P).tocsr()
        A.eliminate_zeros()
        shape = cshape
    levels.append((A, None, shape))
    return levels


def colours(shape):
    R, C = shape
    r, c = np.divmod(np.arange(R * C), C)
    return [np.where((r % 2 == a) & (c % 2 == b))[0] for a, b in ((0, 0), (1, 1), (0, 1), (1, 0))]


def make_cycle(levels, nu=1, smoother='gs4', omega=0.8, gamma=1):
    cols = [colours(s) for _, _, s in levels]
    dinv = [1.0 / A.diagonal() for A, _, _ in levels]
    lu = spl.splu(levels[-1][0].tocsc())
    rows_of = [[A[cs] for cs in cl] for (A, _, _), cl in zip(levels, cols)] if smoother == 'gs4' else None

    def smooth(l, x, b, rev):
        A = levels[l][0]
        if smoother == 'gs4':
            order = range(3, -1, -1) if rev else range(4)
            for _ in range(nu):
                for k in order:
                    cs = cols[l][k]
                    x[cs] += (b[cs] - rows_of[l][k] @ x) * dinv[l][cs]
        else:
            for _ in range(nu):
                x += omega * dinv[l] * (b - A @ x)
        return x

    def cyc(l, b):
        A, P, _ = levels[l]
        if P is None:
            return lu.solve(b)
        x = smooth(l, np.zeros(b.shape), b, False)
        for _ in range(gamma if l > 0 else 1):
            x += P @ cyc(l + 1, P.T @ (b - A @ x))
        return smooth(l, x, b, True)
    return lambda b: cyc(0, b)


def pcg(A, b, M, maxit, tol=1e-15):
    x = np.zeros_like(b); r = b.copy(); z = M(r); p = z.copy(); rz = r @ z; b2 = np.linalg.norm(b)
    hist = []
    for it in range(1, maxit + 1):
        Ap = A @ p; a = rz / (p @ Ap); x += a * p; r -= a * Ap
        rel = np.linalg.norm(r) / b2
        hist.append(rel)
        if rel <= tol:
            break
        z = M(r); rz2 = r @ z; p = z + (rz2 / rz) * p; rz = rz2
    return x, it, hist


def load(which, argv):
    if which == 'c1':
        g = np.load('tests/golden/g8_c1.npz')
        return orc.get_above_threshold_speed(g['orograph_f32'], 0.75)
    if which in ('g10', 'g11'):
        sys.path.insert(0, 'tests')
        from conftest import load_g10
        g = load_g10('g10_10m.npz' if which == 'g10' else 'g11_wander.npz')
        return orc.get_above_threshold_speed(g['orograph_f32'], 0.75)
    from ssrs_amd.synthetic import synthetic_dem
    rows, cols = int(argv[0]), int(argv[1])
    res = float(argv[2]) if len(argv) > 2 else 10.
    z = synthetic_dem((rows, cols), res)
    oro = orc.compute_orographic_updraft(10., 270., orc.compute_slope_degrees(z, res),
                                         orc.compute_aspect_degrees(z, res)).astype(np.float32)
    return orc.get_above_threshold_speed(oro, 0.75)


def run(cond, dirn=0., label='', configs=None, tol=1e-15):
    A, rhs, fixed, val = setup(cond, dirn)
    t = time.time(); lv = hierarchy(A, cond.shape)
    nnz = [a.nnz for a, _, _ in lv]
    print(f'{label} unknowns {A.shape[0]} dead {float((cond <= 0).mean()):.2f} levels {len(lv)} '
          f'operator complexity {sum(nnz) / nnz[0]:.2f} max row nnz {[int(np.diff(a.indptr).max()) for a, _, _ in lv[:4]]} setup {time.time() - t:.1f}s', flush=True)
    out = {}
    for name, kw in (configs or [('V(1,1) gs4', dict(nu=1)), ('V(2,2) jacobi .8', dict(nu=2, smoother='jacobi'))]):
        M = make_cycle(lv, **kw)
        t = time.time(); x, it, hist = pcg(A, rhs, M, 300, tol)
        k8 = next((i + 1 for i, h in enumerate(hist) if h <= 1e-8), None)
        print(f'    {name}: {it} its to {hist[-1]:.1e} ({k8} to 1e-8), {time.time() - t:.1f}s', flush=True)
        out[name] = it
    return out


if __name__ == '__main__':
    which = sys.argv[1] if len(sys.argv) > 1 else 'c1'
    if which == 'soak':
        master = np.random.default_rng(4242)
        worst = 0
        for k in range(int(sys.argv[2])):
            seed = int(master.integers(0, 2**31)); rng = np.random.default_rng(seed)
            rows, cols = int(rng.integers(6, 90)), int(rng.integers(6, 110))
            dirn = float(rng.choice([0., 45., 90., 135., 180., 225., 270., 315., -45., rng.uniform(0, 360)]))
            cond = np.abs(rng.normal(0.8, 0.6, (rows, cols))) * 10.0 ** rng.uniform(-3, 1)
            dead = rng.choice([0.0, 0.2, 0.5, 0.7])
            cond[rng.random((rows, cols)) < dead] = 0.0
            if rng.random() < 0.3:
                r0, c0 = int(rng.integers(0, rows - 3)), int(rng.integers(0, cols - 3))
                cond[r0:r0 + rows // 3, c0:c0 + cols // 3] = 0.0
            o = run(cond, dirn, f'soak {seed} {rows}x{cols} dirn {dirn:.0f} dead {dead}', [('V(1,1) gs4', dict(nu=1))])
            worst = max(worst, max(o.values()))
        print('worst iterations', worst)
    else:
        run(load(which, sys.argv[2:]), 0., which)
