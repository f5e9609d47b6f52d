"""K5 at C2: set-up time, solve time, iterations and the bytes of workspace really used."""
import os, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from ssrs_amd import layers
from ssrs_amd.potential import solve_potential
from ssrs_amd.synthetic import synthetic_dem
for shape in ((500, 600), (1000, 1200), (2000, 2400), (5000, 6000)):
    res = 100. if shape == (500, 600) else 10.
    dem = torch.from_numpy(synthetic_dem(shape, res)).cuda()
    _, upd = layers.updraft_from_dem(dem, res, 10., 270., threshold=0.75)
    for tol in (1e-8, 1e-15):
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            torch.cuda.synchronize(); t = time.time()
            pot, st = solve_potential(upd, 0., rel_tol=tol, return_stats=True)
            torch.cuda.synchronize(); dt = time.time() - t
        print(f'{shape[0]}x{shape[1]} rel_tol {tol:g}: {st["iterations"]} iterations, set-up {st["setup_ms"] / 1e3:.3f} s, iterations {st["kernel_ms"] / 1e3:.3f} s, '
              f'wall {dt:.3f} s, |r|/|b| {st["residual"]:.1e}; workspace used {st["workspace_used"] / 1e9:.2f} GB of {st["workspace_bytes"] / 1e9:.2f} GB reserved '
              f'({st["workspace_used"] / (shape[0] * shape[1]):.0f} B per cell), {st["amg_levels"]} levels', flush=True)
