"""Solved potential on a 2000 x 2400 synthetic raster, 20k tracks: how often does a step
take the exact sequence?"""
import os, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from ssrs_amd import layers, movmodel
from ssrs_amd.potential import solve_potential
from ssrs_amd.synthetic import synthetic_dem
shape = (2000, 2400)
dem = torch.from_numpy(synthetic_dem(shape, 10.)).cuda()
_, upd = layers.updraft_from_dem(dem, 10., 10., 270., threshold=0.75)
with warnings.catch_warnings():
    warnings.simplefilter('ignore')
    pot, st = solve_potential(upd, 0., max_iterations=3000, return_stats=True)
print('solve', st['iterations'], flush=True)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
np.random.seed(30)
r, c = movmodel.get_starting_indices(n, (2, 22, 1, 2), 'random', (24., 20.), 10.)
starts = np.stack([r, c], 1)
for kw in (dict(scattered=False), dict(scattered=True), dict(ring=False)):
    torch.cuda.synchronize(); t = time.time()
    out = movmodel.simulate_tracks(0., starts, shape, 1, 1., upd, pot, seed=30, use_table=True, profile=True, **kw)
    torch.cuda.synchronize(); dt = time.time() - t
    L = out.lengths.cpu().numpy()
    print(kw, 'steps', out.stats['total_steps'], 'mean', L.mean(), 'max', L.max(), 'launches', out.stats['launches'],
          'kernel s', out.stats['kernel_ms'] / 1e3, 'wall', round(dt, 2), 'steps/s', out.stats['total_steps'] / dt / 1e9, flush=True)
