"""C2 workload: histogram-only call vs the two-pass trajectory call (Simulator default)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from ssrs_amd import layers, movmodel
from ssrs_amd.synthetic import synthetic_dem, ramp_potential
rows, cols, res = 5000, 6000, 10.
n = 100000
dem = torch.from_numpy(synthetic_dem((rows, cols), res)).cuda()
np.random.seed(30)
r, c = movmodel.get_starting_indices(n, (5, 55, 1, 2), 'random', (60., 50.), res)
starts = torch.from_numpy(np.stack([r, c], 1).astype(np.int32)).cuda()
pot = torch.from_numpy(ramp_potential((rows, cols))).cuda()
_, upd = layers.updraft_from_dem(dem, res, 10., 270., threshold=0.75)
for name, kw in (('hist only (ring)', dict()), ('two-pass trajectories', dict(want_tracks=True))):
    for rep in range(2):
        torch.cuda.synchronize(); t = time.time()
        out = movmodel.simulate_tracks(0., starts, (rows, cols), 1, 1., upd, pot, seed=30, use_table=True, **kw)
        torch.cuda.synchronize(); dt = time.time() - t
    print(name, round(dt * 1e3, 2), 'ms', 'traj MB', 0 if out.traj is None else out.traj.numel() * 2 / 1e6, flush=True)
