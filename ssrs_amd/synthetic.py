"""Deterministic synthetic terrain and wind inputs.

The reference obtains its rasters from the network (USGS 3DEP, WIND Toolkit:
/root/reference/ssrs/simulator.py:88-125); every BASELINE.json config is
synthetic, so these generators stand in for that L1 layer.  They are host-side
numpy on purpose: PCG64 streams (`default_rng`) are stable across numpy
versions, so the same arrays are produced in the build container, on the GPU
box, and in the golden-vector generator.
"""
import numpy as np


def synthetic_dem(gridsize, resolution, seed=12345, noise=1.5):
    """Sinusoid + noise DEM, f64 (rows, cols), row 0 = south.

    z[r, c] = 1800 + 300 sin(c/37s) cos(r/23s) + 200 sin((r+c)/61s)
              + 80 cos(c/9s) sin(r/11s) + N(0, noise),   s = 100 / resolution
    so feature wavelengths are constant in metres.
    """
    rows, cols = gridsize
    s = 100.0 / float(resolution)
    r = np.arange(rows, dtype=np.float64)[:, None]
    c = np.arange(cols, dtype=np.float64)[None, :]
    z = (1800.0 + 300.0 * np.sin(c / (37.0 * s)) * np.cos(r / (23.0 * s))
         + 200.0 * np.sin((r + c) / (61.0 * s))
         + 80.0 * np.cos(c / (9.0 * s)) * np.sin(r / (11.0 * s)))
    if noise:
        rng = np.random.default_rng(seed)
        z = z + rng.normal(0.0, noise, size=(rows, cols))
    return z


def wind_lattice(region_width_km, spacing_km=2.0, phase=0.0):
    """WTK-shaped wind samples on a regular lattice.

    Returns (x_km[nx], y_km[ny], wspeed[ny, nx], wdirn_deg[ny, nx]) with
    speed = 8 + 3 sin(x/17 + phase) cos(y/13),
    dirn  = 270 + 40 sin(x/23 + y/31 + phase)      (x, y in km).
    60x50 km @ 2 km gives the 31 x 26 points of SURVEY 8(d).
    """
    nx = int(round(region_width_km[0] / spacing_km)) + 1
    ny = int(round(region_width_km[1] / spacing_km)) + 1
    x = np.arange(nx, dtype=np.float64) * spacing_km
    y = np.arange(ny, dtype=np.float64) * spacing_km
    xx, yy = np.meshgrid(x, y)
    wspeed = 8.0 + 3.0 * np.sin(xx / 17.0 + phase) * np.cos(yy / 13.0)
    wdirn = 270.0 + 40.0 * np.sin(xx / 23.0 + yy / 31.0 + phase)
    return x, y, wspeed, wdirn


def ramp_potential(gridsize):
    """Linear-ramp stand-in potential 1000 (1 - r/(R-1)), f32: the exact
    solution of the reference's Dirichlet problem for uniform conductance and
    track_direction 0 (movmodel.py:21-57).  Always *labelled* as such in
    reports; the real field comes from the potential solver."""
    rows, cols = gridsize
    r = np.arange(rows, dtype=np.float64)[:, None]
    p = 1000.0 * (1.0 - r / (rows - 1.0))
    return np.broadcast_to(p, (rows, cols)).astype(np.float32).copy()
