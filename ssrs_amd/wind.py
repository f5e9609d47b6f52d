"""Wind-field preparation (K6) for snapshot / seasonal modes: WTK-shaped
lattice samples -> per-cell speed and direction rasters, following the u/v
recipe of /root/reference/ssrs/simulator.py:778-792."""
import ctypes as C

import numpy as np
import torch

from . import _native as nat
from ._device import stream_ptr, to_dev


def interpolate_wind_lattice(x_km, y_km, wspeed, wdirn, gridsize, resolution):
    """x_km[nx], y_km[ny]: lattice coordinates (uniform spacing) relative to
    the raster's south-west cell centre; wspeed/wdirn: (ny, nx) or (B, ny, nx).
    Returns (wspeed, wdirn) f64 CUDA tensors (rows, cols) or (B, rows, cols)."""
    x = np.asarray(x_km, dtype=np.float64)
    y = np.asarray(y_km, dtype=np.float64)
    nx, ny = x.size, y.size
    dx = float(x[1] - x[0]) if nx > 1 else 1.0
    dy = float(y[1] - y[0]) if ny > 1 else 1.0
    if nx > 2 and not np.allclose(np.diff(x), dx) or ny > 2 and not np.allclose(np.diff(y), dy):
        raise ValueError('wind lattice must be uniformly spaced')
    ws = to_dev(wspeed, torch.float64)
    wd = to_dev(wdirn, torch.float64)
    single = ws.dim() == 2
    if single:
        ws, wd = ws[None], wd[None]
    if tuple(ws.shape[1:]) != (ny, nx) or ws.shape != wd.shape:
        raise ValueError(f'lattice arrays must be (ny, nx) = {(ny, nx)}')
    batch = int(ws.shape[0])
    rows, cols = int(gridsize[0]), int(gridsize[1])
    out_s = torch.empty((batch, rows, cols), dtype=torch.float64, device=ws.device)
    out_d = torch.empty_like(out_s)
    nat.check(nat.lib().ssrs_wind_from_lattice(
        nat.ptr(ws.contiguous()), nat.ptr(wd.contiguous()), nx, ny, C.c_double(x[0]),
        C.c_double(y[0]), C.c_double(dx), C.c_double(dy), C.c_double(resolution / 1000.),
        nat.ptr(out_s), nat.ptr(out_d), rows, cols, batch, stream_ptr()))
    return (out_s[0], out_d[0]) if single else (out_s, out_d)


def interpolate_wind_scattered(x_km, y_km, wspeed, wdirn, gridsize, resolution):
    """The reference's general case (/root/reference/ssrs/simulator.py:765-792): wind samples at SCATTERED points
    x_km[npts], y_km[npts] (relative to the raster's south-west cell centre), wspeed / wdirn (npts,) or (B, npts).
    `scipy.interpolate.griddata(..., method='linear')` is a Delaunay triangulation + barycentric interpolation: the
    triangulation is built here on the host by the same scipy class griddata uses (a few thousand points), the
    30 M cells are interpolated by the HIP kernels behind `ssrs_wind_from_triangles`.  Returns (wspeed, wdirn) f64
    CUDA tensors (rows, cols) or (B, rows, cols); NaN outside the convex hull of the points, as griddata."""
    from scipy.spatial import Delaunay
    x = np.asarray(x_km, dtype=np.float64).ravel()
    y = np.asarray(y_km, dtype=np.float64).ravel()
    if x.size != y.size or x.size < 3:
        raise ValueError('scattered wind samples need x_km, y_km of equal length >= 3')
    pts = np.ascontiguousarray(np.stack([x, y], 1))
    tri = Delaunay(pts)                                        # what griddata -> LinearNDInterpolator builds
    ws = to_dev(wspeed, torch.float64)
    wd = to_dev(wdirn, torch.float64)
    single = ws.dim() == 1
    if single:
        ws, wd = ws[None], wd[None]
    if ws.dim() != 2 or int(ws.shape[1]) != x.size or ws.shape != wd.shape:
        raise ValueError(f'scattered wind arrays must be (npts,) or (B, npts) with npts = {x.size}')
    batch = int(ws.shape[0])
    rows, cols = int(gridsize[0]), int(gridsize[1])
    dev = ws.device
    d_pts = torch.from_numpy(pts).to(dev)
    d_tri = torch.from_numpy(np.ascontiguousarray(tri.simplices.astype(np.int32))).to(dev)
    d_tr = torch.from_numpy(np.ascontiguousarray(tri.transform.astype(np.float64))).to(dev)
    out_s = torch.empty((batch, rows, cols), dtype=torch.float64, device=dev)
    out_d = torch.empty_like(out_s)
    L = nat.lib()
    nbytes = int(L.ssrs_wind_triangles_workspace_bytes(int(x.size), rows, cols, batch))
    scratch = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    nat.check(L.ssrs_wind_from_triangles(
        nat.ptr(d_pts), nat.ptr(d_tri), nat.ptr(d_tr), nat.ptr(ws.contiguous()), nat.ptr(wd.contiguous()),
        int(x.size), int(d_tri.shape[0]), C.c_double(resolution / 1000.), nat.ptr(out_s), nat.ptr(out_d), rows, cols, batch,
        nat.ptr(scratch), C.c_size_t(nbytes), stream_ptr()))
    return (out_s[0], out_d[0]) if single else (out_s, out_d)
