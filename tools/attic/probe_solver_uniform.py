import sys; sys.path.insert(0,'.')
import numpy as np, torch
from ssrs_amd.potential import solve_potential
rows, cols = 1000, 1200
for name, cond in [('uniform', np.ones((rows, cols))),
                   ('smooth x100', 1.0 + 99.0 * (np.sin(np.arange(cols)[None, :] / 40.) * np.cos(np.arange(rows)[:, None] / 30.) > 0)),
                   ('lognormal s=3', np.exp(3.0 * np.random.default_rng(0).normal(size=(rows, cols))))]:
    for cyc in ('V', 'K'):
        pot, st = solve_potential(cond, 0., rel_tol=1e-8, max_iterations=1500, return_stats=True, cycle=cyc)
        print(name, cyc, st, flush=True)
