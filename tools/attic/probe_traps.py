"""Where do the tracks that run into the step cap spend their time at C2?  Solved field
(default tolerance), 20k tracks capped at 60000 steps; for a few capped tracks: bounding box of
the second half of the trajectory, live fraction and f32 potential levels inside it."""
import os, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from ssrs_amd import layers, movmodel
from ssrs_amd.potential import solve_potential
from ssrs_amd.synthetic import synthetic_dem
shape = (5000, 6000); res = 10.
dem = torch.from_numpy(synthetic_dem(shape, res)).cuda()
_, upd = layers.updraft_from_dem(dem, res, 10., 270., threshold=0.75)
with warnings.catch_warnings():
    warnings.simplefilter('ignore')
    pot, st = solve_potential(upd, 0., return_stats=True)
print('solve', st, flush=True)
n, cap = 20000, 60000
np.random.seed(30)
r, c = movmodel.get_starting_indices(n, (5, 55, 1, 2), 'random', (60., 50.), res)
starts = np.stack([r, c], 1)
out = movmodel.simulate_tracks(0., starts, shape, 1, 1., upd, pot, seed=30, use_table=True, max_moves=cap)
L = out.lengths.cpu().numpy() - 1
ends = out.ends.cpu().numpy()
capped = np.nonzero(L >= cap)[0]
print(f'{n} tracks: at cap {capped.size / n:.3f}; end rows of capped tracks: percentiles {np.percentile(ends[capped, 0], [0, 10, 50, 90, 100])}', flush=True)
hist_rows = np.bincount(ends[capped, 0] // 250, minlength=20)
print('capped tracks by end-row band of 250 rows:', hist_rows.tolist())
P = pot.cpu().numpy(); U = upd.cpu().numpy()
sub = capped[:: max(1, capped.size // 6)][:6]
res_t = movmodel.simulate_tracks(0., starts[sub], shape, 1, 1., upd, pot, seed=30, use_table=True, max_moves=cap,
                                 want_tracks=True, want_hist=False)
# NB: track ids restart at 0 for the subset, so these are other streams from the same start cells
for tr in res_t.tracks():
    if len(tr) <= cap:
        print('  (this replay left the raster after', len(tr) - 1, 'steps)')
        continue
    half = tr[len(tr) // 2:]
    r0, r1, c0, c1 = half[:, 0].min(), half[:, 0].max(), half[:, 1].min(), half[:, 1].max()
    box = P[r0:r1 + 1, c0:c1 + 1]; live = U[r0:r1 + 1, c0:c1 + 1] > 0
    vis = np.zeros(shape, bool); vis[half[:, 0], half[:, 1]] = True
    pv = P[vis]
    ulp = np.spacing(np.float32(np.median(pv)))
    print(f'  second half of the track stays in rows {r0}..{r1}, cols {c0}..{c1} ({vis.sum()} distinct cells); live fraction of those cells '
          f'{(U[vis] > 0).mean():.2f}; potential there: median {np.median(pv):.4f}, max - min = {float(pv.max()) - float(pv.min()):.3e} = '
          f'{(float(pv.max()) - float(pv.min())) / ulp:.1f} f32 ulp, {np.unique(pv).size} distinct values', flush=True)
    # the potential just north of the box: is the way north uphill?
    north = P[min(r1 + 1, shape[0] - 1):min(r1 + 6, shape[0]), c0:c1 + 1]
    print(f'     five rows north of the box: min {north.min():.4f} max {north.max():.4f} (box min {box.min():.4f})')
