"""How many distinct cells does a track visit within one launch (512 steps) once it is
wandering in the solved C2 field?"""
import os, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from ssrs_amd import layers, movmodel
from ssrs_amd.potential import solve_potential
from ssrs_amd.synthetic import synthetic_dem
shape = (5000, 6000)
dem = torch.from_numpy(synthetic_dem(shape, 10.)).cuda()
_, upd = layers.updraft_from_dem(dem, 10., 10., 270., threshold=0.75)
with warnings.catch_warnings():
    warnings.simplefilter('ignore')
    pot, st = solve_potential(upd, 0., max_iterations=3000, return_stats=True)   # library default rel_tol
np.random.seed(30)
r, c = movmodel.get_starting_indices(100000, (5, 55, 1, 2), 'random', (60., 50.), 10.)
starts = np.stack([r, c], 1)[:1500]
out = movmodel.simulate_tracks(0., starts, shape, 1, 1., upd, pot, seed=30, use_table=True, want_tracks=True, max_moves=40000)
tr = out.tracks()
uniq = {64: [], 512: [], 4096: []}
for t in tr:
    if len(t) < 30000:
        continue
    cells = t[20000:, 0].astype(np.int64) * 6000 + t[20000:, 1]
    for w in uniq:
        for s in range(0, len(cells) - w, w):
            uniq[w].append(len(np.unique(cells[s:s + w])))
for w, u in uniq.items():
    u = np.array(u)
    print(f'window {w}: distinct cells mean {u.mean():.1f} median {np.median(u):.0f} p10 {np.percentile(u, 10):.0f} p90 {np.percentile(u, 90):.0f}  (n={u.size})')
