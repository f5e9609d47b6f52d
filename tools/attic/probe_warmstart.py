"""Does a neighbouring snapshot's potential help as initial guess?  2000 x 2400 synthetic
raster, wind changed from 10 m/s @ 270 deg to the values below."""
import os, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from ssrs_amd import layers
from ssrs_amd.potential import solve_potential
from ssrs_amd.synthetic import synthetic_dem
shape = (2000, 2400)
dem = torch.from_numpy(synthetic_dem(shape, 10.)).cuda()
def solve(ws, wd, guess=None):
    _, upd = layers.updraft_from_dem(dem, 10., ws, wd, threshold=0.75)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        return solve_potential(upd, 0., rel_tol=1e-10, max_iterations=1500, return_stats=True, initial_guess=guess)
p0, s0 = solve(10., 270.)
print('base', s0['iterations'], round(s0['kernel_ms']), 'ms', flush=True)
for ws, wd in ((10., 272.), (9.5, 280.), (8., 300.), (12., 240.)):
    pc, sc = solve(ws, wd)
    pw, sw = solve(ws, wd, guess=p0.double())
    print(f'wind {ws} @ {wd}: cold {sc["iterations"]} its {sc["kernel_ms"]:.0f} ms | warm {sw["iterations"]} its {sw["kernel_ms"]:.0f} ms '
          f'| max |cold - warm| {float((pc - pw).abs().max()):.2e}', flush=True)
