import sys, time; sys.path.insert(0,'.')
import numpy as np
from ssrs_amd.potential import solve_potential
from ssrs_amd import layers
g=np.load('tests/golden/g5_potential.npz')
for dirn in (0., 180., -45., 90., 30.):
    pot, st = solve_potential(g['updraft'], dirn, rel_tol=1e-12, return_stats=True)
    ref = g[f'pot_d{int(dirn % 360)}']
    print(dirn, st, 'maxabs', np.abs(pot-ref).max(), flush=True)
g8=np.load('tests/golden/g8_c1.npz')
upd = layers.get_above_threshold_speed(g8['orograph_f32'], 0.75)
for tol in (1e-8, 1e-11):
    t=time.time(); pot, st = solve_potential(upd, 0., rel_tol=tol, return_stats=True)
    print('C1', tol, st, 'maxabs', np.abs(pot-g8['potential']).max(), 'mean', np.abs(pot-g8['potential']).mean(), round(time.time()-t,2), flush=True)
