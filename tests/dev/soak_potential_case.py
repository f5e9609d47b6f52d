"""One case of soak_potential.py by seed: python tests/dev/soak_potential_case.py SEED [rel_tol ...]"""
import os, sys, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import scipy.sparse.linalg as ssl
from ssrs_amd.potential import solve_potential
argv = sys.argv
import importlib.util
spec = importlib.util.spec_from_file_location('soak', os.path.join(os.path.dirname(os.path.abspath(__file__)), 'soak_potential.py'))
soak = importlib.util.module_from_spec(spec); spec.loader.exec_module(soak)
seed = int(argv[1]); rng = np.random.default_rng(seed)
rows, cols = int(rng.integers(6, 90)), int(rng.integers(6, 110))
dirn = float(rng.choice([0., 45., 90., 135., 180., 225., 270., 315., -45., rng.uniform(0, 360)]))
cond = np.abs(rng.normal(0.8, 0.6, (rows, cols))) * 10.0 ** rng.uniform(-3, 1)
dead = rng.choice([0.0, 0.2, 0.5, 0.7])
cond[rng.random((rows, cols)) < dead] = 0.0
if rng.random() < 0.3:
    r0, c0 = int(rng.integers(0, rows - 3)), int(rng.integers(0, cols - 3))
    cond[r0:r0 + rows // 3, c0:c0 + cols // 3] = 0.0
a_mat, b_vec, inodes, bnodes, benergy = soak.assemble(cond, dirn)
x_ref = ssl.spsolve(a_mat, b_vec)
for tol in [float(v) for v in argv[2:]] or [1e-15]:
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        pot, st = solve_potential(cond, dirn, rel_tol=tol, max_iterations=3000, return_stats=True)
    x_gpu = np.asarray(pot, dtype=np.float64).T.reshape(-1)[inodes]
    print(rows, cols, dirn, float(dead), 'tol', tol, 'its', st['iterations'], 'res', st['residual'],
          'max |phi - spsolve|', float(np.abs(x_gpu - x_ref).max()), flush=True)
