"""One case of tests/dev/soak_potential.py by its seed under the solver's switches.
usage: python tests/dev/soak_potential_one.py SEED"""
import os, sys, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import scipy.sparse.linalg as ssl
from ssrs_amd.potential import solve_potential
from soak_potential import assemble

seed = int(sys.argv[1]); rng = np.random.default_rng(seed)
rows, cols = int(rng.integers(6, 90)), int(rng.integers(6, 110))
dirn = float(rng.choice([0., 45., 90., 135., 180., 225., 270., 315., -45., rng.uniform(0, 360)]))
cond = np.abs(rng.normal(0.8, 0.6, (rows, cols))) * 10.0 ** rng.uniform(-3, 1)
dead = rng.choice([0.0, 0.2, 0.5, 0.7])
cond[rng.random((rows, cols)) < dead] = 0.0
if rng.random() < 0.3:
    r0, c0 = int(rng.integers(0, rows - 3)), int(rng.integers(0, cols - 3))
    cond[r0:r0 + rows // 3, c0:c0 + cols // 3] = 0.0
a_mat, b_vec, inodes, bnodes, benergy = assemble(cond, dirn)
x_ref = ssl.spsolve(a_mat, b_vec)
print(dict(seed=seed, rows=rows, cols=cols, dirn=dirn, dead=float(dead)))
for env in ({}, {'SSRS_AMG_NO_BLOCKS': '1'}, {'SSRS_AMG_NU': '2,2'}, {'SSRS_AMG_NU': '2,2', 'SSRS_AMG_NO_BLOCKS': '1'}, {'SSRS_AMG_NO_FUSE': '1'}):
    os.environ.update(env)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        pot, st = solve_potential(cond, dirn, rel_tol=1e-15, max_iterations=3000, return_stats=True)
    for k in env:
        del os.environ[k]
    x_gpu = np.asarray(pot, dtype=np.float64).T.reshape(-1)[inodes]
    print(env or 'default', 'its', st['iterations'], 'conv', st['converged'], 'res %.1e' % st['residual'], 'levels', st['amg_levels'],
          'err %.2e' % float(np.abs(x_gpu - x_ref).max()), flush=True)
