"""C2 potential solve with K-cycle depths: iterations and time (python tools/attic/probe_kcycle.py 0 1 2 3)."""
import os, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from ssrs_amd import layers
from ssrs_amd.potential import solve_potential
from ssrs_amd.synthetic import synthetic_dem
shape = (5000, 6000)
dem = torch.from_numpy(synthetic_dem(shape, 10.)).cuda()
_, upd = layers.updraft_from_dem(dem, 10., 10., 270., threshold=0.75)
for kd in [int(v) for v in sys.argv[1:]] or [0, 1, 2]:
    torch.cuda.synchronize(); t = time.time()
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        pot, st = solve_potential(upd, 0., rel_tol=1e-8, max_iterations=600, return_stats=True, kdepth=kd, cycle='K' if kd else 'V')
    torch.cuda.synchronize()
    print('kdepth', kd, st, 'wall', round(time.time() - t, 2), flush=True)
