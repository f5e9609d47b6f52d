"""K5 on the field of one synthetic wind snapshot of configs[4] (5000 x 6000 @10 m) under SSRS_AMG_NU variants.
usage: python tools/dev/probe_k5_snapshot.py SNAPSHOT [nu ...]"""
import os, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from ssrs_amd import layers
from ssrs_amd.potential import solve_potential
from ssrs_amd.synthetic import synthetic_dem, wind_lattice
s = int(sys.argv[1])
nus = sys.argv[2:] or ['1,1', '2,2', '1,2', '2,1']
shape = (5000, 6000)
dem = torch.from_numpy(synthetic_dem(shape, 10.)).cuda()
x, y, ws, wd = wind_lattice((60., 50.), 2.0, phase=2 * np.pi * s / 256)
oro, _ = layers.updraft_from_dem_lattice(dem, 10., x, y, ws, wd)
upd = layers.get_above_threshold_speed(oro, 0.75)
del dem, oro
ref = None
for nu in nus:
    os.environ['SSRS_AMG_NU'] = nu
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        pot, st = solve_potential(upd, 0., return_stats=True)
    ref = pot if ref is None else ref
    print(f'snapshot {s} nu {nu}: its {st["iterations"]} conv {st["converged"]} res {st["residual"]:.1e} solve {st["kernel_ms"] / 1e3:.2f}s '
          f'max|d| vs first {float((pot - ref).abs().max()):.2e}', flush=True)
