import sys, time; sys.path.insert(0,'.')
import numpy as np, torch
from ssrs_amd.potential import solve_potential
from ssrs_amd import layers
from ssrs_amd.synthetic import synthetic_dem
rows, cols = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (5000, 6000)
res = 50000. / rows
dem = torch.from_numpy(synthetic_dem((rows, cols), res)).cuda()
_, upd = layers.updraft_from_dem(dem, res, 10., 270., threshold=0.75)
print('zero fraction', float((upd == 0).float().mean()), flush=True)
t = time.time()
pot, st = solve_potential(upd, 0., rel_tol=float(sys.argv[3]) if len(sys.argv) > 3 else 1e-8, max_iterations=int(sys.argv[4]) if len(sys.argv) > 4 else 600, return_stats=True, extra_sweeps=int(sys.argv[5]) if len(sys.argv) > 5 else 0, cycle=sys.argv[6] if len(sys.argv) > 6 else 'K', strong_rounds=int(sys.argv[7]) if len(sys.argv) > 7 else 0)
torch.cuda.synchronize()
print(rows, cols, st, 'wall', round(time.time() - t, 2), 'mem GB', torch.cuda.max_memory_allocated() / 1e9, flush=True)
print('pot range', float(pot.min()), float(pot.max()), 'nan', bool(torch.isnan(pot).any()))
