import sys; sys.path.insert(0,'.')
import numpy as np
from ssrs_amd.potential import solve_potential
from ssrs_amd import layers
g8=np.load('tests/golden/g8_c1.npz')
upd = layers.get_above_threshold_speed(g8['orograph_f32'], 0.75)
R, C = upd.shape
ramp = 1000. * (1 - np.arange(R)[:, None] / (R - 1.)) * np.ones((1, C))
for name, guess in (('none (500)', None), ('ramp', ramp)):
    pot, st = solve_potential(upd, 0., rel_tol=1e-8, return_stats=True, initial_guess=guess)
    print(name, st, 'maxabs', np.abs(pot - g8['potential']).max())
