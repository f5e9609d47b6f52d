"""Solver on small speckled rasters: iterations / residual.  (Was the A/B of the double pairwise
aggregation experiment, profiles/r01_notes.md; the SSRS_AMG_DOUBLE switch went with the revert.)"""
import os, sys, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ssrs_amd.potential import solve_potential
rng = np.random.default_rng(3)
for shape, dead in (((96, 128), 0.0), ((96, 128), 0.5), ((500, 600), 0.0), ((500, 600), 0.5), ((1500, 1800), 0.3)):
    cond = np.abs(rng.normal(0.8, 0.6, shape))
    cond[rng.random(shape) < dead] = 0.0
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        pot, st = solve_potential(cond, 0., rel_tol=1e-10, max_iterations=600, return_stats=True)
    print(shape, dead, st, flush=True)
