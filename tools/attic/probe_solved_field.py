"""Is the multi-million-step 'wandering' at 10 m the reference's behaviour or solver error?
Solve the C2 potential at several tolerances, count the strict interior minima / maxima of the
f32 field (the exact solution is discrete-harmonic: none) and step a batch through each field.
usage: probe_solved_field.py [rows cols [ntracks [cap]]]"""
import os, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import torch.nn.functional as F
from ssrs_amd import layers, movmodel
from ssrs_amd.potential import solve_potential
from ssrs_amd.synthetic import synthetic_dem

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
cols = int(sys.argv[2]) if len(sys.argv) > 2 else 6000
n = int(sys.argv[3]) if len(sys.argv) > 3 else 100000
cap = int(sys.argv[4]) if len(sys.argv) > 4 else 60000
tols = [float(x) for x in os.environ.get('TOLS', '1e-8,1e-12,1e-15').split(',')]
shape = (rows, cols)
res = 10.
dem = torch.from_numpy(synthetic_dem(shape, res)).cuda()
_, upd = layers.updraft_from_dem(dem, res, 10., 270., threshold=0.75)
print(f'{rows}x{cols} @10 m, dead fraction {float((upd <= 0).double().mean()):.3f}', flush=True)
width = (cols * res / 1000., rows * res / 1000.)
np.random.seed(30)
band = (5, 55, 1, 2) if rows == 5000 else (1., width[0] - 1., 0.1, 0.3)
r, c = movmodel.get_starting_indices(n, band, 'random', width, res)
starts = np.stack([r, c], 1)


def extrema(pot):
    p = pot.double()
    inner = p[2:-2, 2:-2]
    lo = torch.ones_like(inner, dtype=torch.bool)
    hi = torch.ones_like(inner, dtype=torch.bool)
    for dr in (-1, 0, 1):
        for dc in (-1, 0, 1):
            if dr == 0 and dc == 0:
                continue
            nb = p[2 + dr:rows - 2 + dr, 2 + dc:cols - 2 + dc]
            lo &= inner < nb
            hi &= inner > nb
    return int(lo.sum().item()), int(hi.sum().item())


prev = None
for tol in tols:
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        torch.cuda.synchronize(); t = time.time()
        pot, st = solve_potential(upd, 0., rel_tol=tol, max_iterations=3000, return_stats=True)
        torch.cuda.synchronize(); dt = time.time() - t
    mn, mx = extrema(pot)
    line = (f'rel_tol {tol:g}: {st["iterations"]} iterations, |r|/|b| {st["residual"]:.2e}, {dt:.2f} s; '
            f'strict interior minima {mn}, maxima {mx}')
    if prev is not None:
        line += f'; max |pot - previous| {float((pot - prev).abs().max().item()):.3e}'
    print(line, flush=True)
    prev = pot.clone()
    table = movmodel.build_transition_table(upd, pot, ring=True)
    torch.cuda.synchronize(); t = time.time()
    out = movmodel.simulate_tracks(0., starts, shape, 1, 1., upd, pot, seed=30, table=table, profile=True, max_moves=cap)
    torch.cuda.synchronize(); dt = time.time() - t
    L = out.lengths.cpu().numpy() - 1
    print(f'   {n} tracks: steps mean {L.mean():.0f} median {np.median(L):.0f} p95 {np.percentile(L, 95):.0f} max {L.max()} '
          f'(cap {cap}), at cap {np.mean(L >= cap):.4f}; {dt * 1e3:.1f} ms wall, {out.stats["total_steps"] / dt / 1e9:.2f} G steps/s, '
          f'{n / dt:.3e} tracks/s; launches {out.stats["launches"]} (window {out.stats["window_launches"]}, tiles {out.stats["tile_launches"]})', flush=True)
    del table
