"""A/B of the double pairwise aggregation experiment (SSRS_AMG_DOUBLE, profiles/r01_notes.md): which field
property breaks it?  python tools/attic/probe_double.py"""
import os, sys, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from ssrs_amd import layers
from ssrs_amd.potential import solve_potential
from ssrs_amd.synthetic import synthetic_dem
shape = (500, 600)
rng = np.random.default_rng(3)
dem = torch.from_numpy(synthetic_dem(shape, 10.)).cuda()
_, upd = layers.updraft_from_dem(dem, 10., 10., 270., threshold=0.75)
terrain = upd.cpu().numpy()
speckle = np.abs(rng.normal(0.8, 0.6, shape))
blocks = speckle.copy()
for _ in range(12):
    r0, c0 = rng.integers(0, shape[0] - 80), rng.integers(0, shape[1] - 80)
    blocks[r0:r0 + 80, c0:c0 + 80] = 0.0
cases = {'terrain (51 % dead patches)': terrain,
         'terrain, dead cells -> 1e-3': np.where(terrain > 0, terrain, 1e-3),
         'terrain live pattern, live values all 1': (terrain > 0).astype(np.float64),
         'speckle + 12 dead blocks of 80 x 80': blocks,
         'speckle with terrain dead mask': np.where(terrain > 0, speckle, 0.0)}
for name, cond in cases.items():
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        pot, st = solve_potential(cond, 0., rel_tol=1e-8, max_iterations=800, return_stats=True)
    print(f'{name:42s} its {st["iterations"]:4d} res {st["residual"]:.1e} levels {st["amg_levels"]}', flush=True)
