"""Solve the C2 potential once (for rocprofv3): 5000 x 6000 synthetic raster at 10 m."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch, warnings
from ssrs_amd import layers
from ssrs_amd.potential import solve_potential
from ssrs_amd.synthetic import synthetic_dem
shape = tuple(int(v) for v in sys.argv[1].split('x')) if len(sys.argv) > 1 else (5000, 6000)
dem = torch.from_numpy(synthetic_dem(shape, 10.)).cuda()
_, upd = layers.updraft_from_dem(dem, 10., 10., 270., threshold=0.75)
torch.cuda.synchronize(); t = time.time()
with warnings.catch_warnings():
    warnings.simplefilter('ignore')
    pot, st = solve_potential(upd, 0., rel_tol=float(os.environ.get('SOLVE_TOL', '1e-15')), max_iterations=2000, return_stats=True)
torch.cuda.synchronize()
print(shape, st, 'wall', round(time.time() - t, 2), 's', 'peak GB', round(torch.cuda.max_memory_allocated() / 1e9, 1), flush=True)
