"""Share of the cells within one f32 ulp of the exact solution (G12) for C1 / G10 / G11 at several solver tolerances and
cycle switches.  usage: python tools/dev/probe_g12_accuracy.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
from conftest import load_golden, load_g10
from ssrs_amd import layers
from ssrs_amd.potential import solve_potential


def ulp(a, b):
    return np.abs(a.view(np.int32).astype(np.int64) - b.view(np.int32).astype(np.int64))


ex = load_golden('g12_exact_potential.npz')
cases = {'c1': load_golden('g8_c1.npz')['orograph_f32'], 'g10': load_g10()['orograph_f32'], 'g11': load_g10('g11_wander.npz')['orograph_f32']}
for env in ({}, {'SSRS_AMG_NO_BLOCKS': '1'}, {'SSRS_AMG_NU': '2,2'}, {'SSRS_AMG_NU': '2,2', 'SSRS_AMG_NO_BLOCKS': '1'}):
    os.environ.update(env)
    for tol in (1e-15, 3e-16, 1e-16):
        row = []
        for tag, oro in cases.items():
            upd = layers.get_above_threshold_speed(oro, 0.75)
            pot, st = solve_potential(upd, 0., rel_tol=tol, return_stats=True)
            stride = int(ex[f'{tag}_stride'])
            u = ulp(np.ascontiguousarray(pot[::stride, ::stride]), ex[f'{tag}_exact_f32'])
            row.append(f'{tag}: {st["iterations"]} it, <=1 ulp {np.mean(u <= 1):.4f}, exact {np.mean(u == 0):.3f}, max {u.max()}')
        print(env or 'default', f'tol {tol:g} |', ' | '.join(row), flush=True)
    for k in env:
        del os.environ[k]
