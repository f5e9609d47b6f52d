"""Scattered wind samples at BASELINE's size: a jittered 2 km lattice (31 x 26 points + margin) onto the 5000 x 6000
raster at 10 m -- ssrs_wind_from_triangles against scipy griddata (what the reference calls), seconds for each."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ssrs_amd.wind import interpolate_wind_scattered       # noqa: E402

rows, cols, cell = 5000, 6000, 0.01
rng = np.random.default_rng(1)
gx, gy = np.meshgrid(np.arange(-2., 63., 2.), np.arange(-2., 53., 2.))
x = (gx + rng.uniform(-0.3, 0.3, gx.shape)).ravel()
y = (gy + rng.uniform(-0.3, 0.3, gy.shape)).ravel()
ws = rng.uniform(2., 14., x.size)
wd = (270. + rng.normal(0., 40., x.size)) % 360.
for rep in range(3):
    torch.cuda.synchronize(); t = time.time()
    s, d = interpolate_wind_scattered(x, y, ws, wd, (rows, cols), cell * 1000.)
    torch.cuda.synchronize(); dt = time.time() - t
    print(f'ssrs_wind_from_triangles, {x.size} points -> {rows} x {cols}: {dt * 1e3:.1f} ms (host triangulation + copies included)', flush=True)
from scipy.interpolate import griddata                     # noqa: E402
t = time.time()
xm, ym = np.meshgrid(np.arange(cols) * cell, np.arange(rows) * cell)
east = ws * np.sin(wd * np.pi / 180.); north = ws * np.cos(wd * np.pi / 180.)
pts = np.array([x, y]).T
ie = griddata(pts, east, (xm, ym), method='linear'); inn = griddata(pts, north, (xm, ym), method='linear')
spd = np.sqrt(ie * ie + inn * inn)
print(f'scipy griddata (two components) + speed: {time.time() - t:.1f} s on one host core')
got = s.cpu().numpy()
print('max |speed - scipy|', float(np.nanmax(np.abs(got - spd))), 'NaN cells', int(np.isnan(got).sum()), int(np.isnan(spd).sum()))
