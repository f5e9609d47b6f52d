"""Updraft raster layer functions on the MI355X (K1), behind the reference's
function names and argument meaning (/root/reference/ssrs/layers.py).

Inputs may be numpy arrays (copied to HBM, result returned as numpy) or CUDA
torch tensors (zero-copy, result is a tensor).  All arithmetic is f64 on the
device; there is no CPU path.
"""
import ctypes as C

import numpy as np
import torch

from . import _native as nat
from ._device import (device, stream_ptr, to_dev, float_dev, like_input, ftype,
                      is_tensor)


def _shape2(t):
    if t.dim() != 2:
        raise ValueError(f'expected a 2-D raster, got shape {tuple(t.shape)}')
    return int(t.shape[0]), int(t.shape[1])


def compute_slope_degrees(z_mat, res):
    """layers.py:63-93.  Returns f64 (rows, cols); border cells 0."""
    return slope_aspect(z_mat, res, want_aspect=False)[0]


def compute_aspect_degrees(z_mat, res):
    """layers.py:96-128."""
    return slope_aspect(z_mat, res, want_slope=False)[1]


def slope_aspect(z_mat, res, want_slope=True, want_aspect=True, out_dtype=torch.float64):
    """Both Horn-stencil layers in one pass over an LDS-staged DEM tile."""
    dem = float_dev(z_mat)
    rows, cols = _shape2(dem)
    slope = torch.empty((rows, cols), dtype=out_dtype, device=dem.device) if want_slope else None
    aspect = torch.empty((rows, cols), dtype=out_dtype, device=dem.device) if want_aspect else None
    nat.check(nat.lib().ssrs_slope_aspect(
        nat.ptr(dem), ftype(dem), C.c_double(res), nat.ptr(slope), nat.ptr(aspect),
        nat.SSRS_F64 if out_dtype == torch.float64 else nat.SSRS_F32,
        rows, cols, stream_ptr()))
    return (None if slope is None else like_input(slope, z_mat),
            None if aspect is None else like_input(aspect, z_mat))


def orographic_updraft(wspeed, wdirn, slope, aspect, min_updraft_val=0.,
                       threshold=None, want_orograph=True):
    """Batched compute_orographic_updraft (+ fused threshold).

    wspeed/wdirn: python scalars or 1-D sequences of B scalars (uniform mode),
    or rasters (rows, cols) / (B, rows, cols) (snapshot / seasonal).
    Returns (orograph f32 | None, usable f64 | None) device tensors shaped
    (rows, cols) for a single case, else (B, rows, cols).
    """
    s = float_dev(slope)
    a = float_dev(aspect)
    if a.dtype != s.dtype:
        a = a.to(s.dtype)
    rows, cols = _shape2(s)
    if tuple(a.shape) != (rows, cols):
        raise ValueError('slope and aspect shapes differ')
    uniform = (wspeed.dim() if is_tensor(wspeed) else np.ndim(wspeed)) <= 1
    single = False
    if uniform:
        ws0 = np.atleast_1d(np.asarray(wspeed.cpu() if is_tensor(wspeed) else wspeed,
                                       dtype=np.float64))
        wd0 = np.atleast_1d(np.asarray(wdirn.cpu() if is_tensor(wdirn) else wdirn,
                                       dtype=np.float64))
        if ws0.shape != wd0.shape:
            raise ValueError('wspeed and wdirn lengths differ')
        batch = ws0.size
        single = (wspeed.dim() if is_tensor(wspeed) else np.ndim(wspeed)) == 0
        ws = wd = None
        wtype = nat.SSRS_F32
        ws0p = ws0.ctypes.data_as(C.POINTER(C.c_double))
        wd0p = wd0.ctypes.data_as(C.POINTER(C.c_double))
    else:
        ws = float_dev(wspeed)
        wd = float_dev(wdirn)
        if wd.dtype != ws.dtype:
            wd = wd.to(ws.dtype)
        if ws.shape != wd.shape:
            raise ValueError('wspeed and wdirn shapes differ')
        if ws.dim() == 2:
            single = True
            ws, wd = ws[None], wd[None]
        if tuple(ws.shape[1:]) != (rows, cols):
            raise ValueError('wind raster shape does not match the terrain')
        batch = int(ws.shape[0])
        wtype = ftype(ws)
        ws0p = wd0p = None
    oro = torch.empty((batch, rows, cols), dtype=torch.float32, device=s.device) \
        if want_orograph else None
    use = torch.empty((batch, rows, cols), dtype=torch.float64, device=s.device) \
        if threshold is not None else None
    nat.check(nat.lib().ssrs_orographic_updraft(
        nat.ptr(s), nat.ptr(a), ftype(s), nat.ptr(ws), nat.ptr(wd), wtype, ws0p, wd0p,
        C.c_double(min_updraft_val), nat.ptr(oro),
        C.c_double(-1.0 if threshold is None else threshold), nat.ptr(use),
        rows, cols, batch, stream_ptr()))
    if single:
        oro = None if oro is None else oro[0]
        use = None if use is None else use[0]
    return oro, use


def compute_orographic_updraft(wspeed, wdirn, slope, aspect, min_updraft_val=0.):
    """layers.py:11-22 with the reference's argument order.  wspeed/wdirn may be
    rasters like the reference passes (constant-filled in uniform mode,
    simulator.py:194-195) or plain scalars.  Returns the f32 raster the
    reference persists (simulator.py:198: `orograph.astype(np.float32)`)."""
    oro, _ = orographic_updraft(wspeed, wdirn, slope, aspect, min_updraft_val)
    return like_input(oro, slope)


def get_above_threshold_speed(in_array, threshold):
    """layers.py:171-185 (f64 output; input is rounded through f32 first when it
    is not f32 already, as the reference only ever feeds the saved f32 raster)."""
    x = to_dev(in_array, torch.float32)
    out = torch.empty(x.shape, dtype=torch.float64, device=x.device)
    nat.check(nat.lib().ssrs_threshold_updraft(
        nat.ptr(x), C.c_double(threshold), nat.ptr(out), C.c_size_t(x.numel()),
        stream_ptr()))
    return like_input(out, in_array)


def updraft_from_dem(z_mat, res, wspeed, wdirn, threshold=None, min_updraft_val=0.,
                     want_orograph=True, out=None):
    """Fused uniform-mode raster: DEM -> (orograph f32, usable f64 | None).
    One HBM pass (8-12 B/cell + outputs), no trig; see DESIGN.md K1.
    `out` = (orograph f32, usable f64) CUDA tensors to write into (either may be None)."""
    dem = float_dev(z_mat)
    rows, cols = _shape2(dem)
    oro = use = None
    if out is not None:
        oro, use = out
        for t, dt in ((oro, torch.float32), (use, torch.float64)):
            if t is not None and not (t.is_cuda and t.dtype == dt and tuple(t.shape) == (rows, cols) and t.is_contiguous()):
                raise ValueError(f'out tensors must be contiguous CUDA ({rows}, {cols}) float32 / float64')
    if oro is None and want_orograph:
        oro = torch.empty((rows, cols), dtype=torch.float32, device=dem.device)
    if use is None and threshold is not None:
        use = torch.empty((rows, cols), dtype=torch.float64, device=dem.device)
    nat.check(nat.lib().ssrs_updraft_from_dem(
        nat.ptr(dem), ftype(dem), C.c_double(res), C.c_double(wspeed), C.c_double(wdirn),
        C.c_double(min_updraft_val), nat.ptr(oro),
        C.c_double(-1.0 if threshold is None else threshold), nat.ptr(use),
        rows, cols, stream_ptr()))
    return (None if oro is None else like_input(oro, z_mat),
            None if use is None else like_input(use, z_mat))


def updraft_from_dem_lattice(z_mat, res, x_km, y_km, wspeed, wdirn, threshold=None,
                             min_updraft_val=0., want_orograph=True):
    """Snapshot / seasonal raster in one pass: DEM + wind samples on a regular lattice
    (x_km[nx], y_km[ny] relative to the south-west cell centre; wspeed / wdirn (ny, nx)
    or (B, ny, nx)) -> (orograph f32 | None, usable f64 | None), shaped (rows, cols) for
    one snapshot, else (B, rows, cols).  Equivalent to wind.interpolate_wind_lattice +
    slope_aspect + orographic_updraft without materialising any of their rasters."""
    import numpy as np
    dem = float_dev(z_mat)
    rows, cols = _shape2(dem)
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
    oro = torch.empty((batch, rows, cols), dtype=torch.float32, device=dem.device) if want_orograph else None
    use = torch.empty((batch, rows, cols), dtype=torch.float64, device=dem.device) \
        if threshold is not None else None
    nbytes = nat.lib().ssrs_lattice_workspace_bytes(nx, ny, batch)
    scratch = torch.empty(nbytes, dtype=torch.uint8, device=dem.device)
    nat.check(nat.lib().ssrs_updraft_from_dem_lattice(
        nat.ptr(dem), ftype(dem), C.c_double(res), nat.ptr(ws.contiguous()), nat.ptr(wd.contiguous()),
        nx, ny, C.c_double(x[0]), C.c_double(y[0]), C.c_double(dx), C.c_double(dy),
        C.c_double(min_updraft_val), nat.ptr(oro), C.c_double(-1. if threshold is None else threshold),
        nat.ptr(use), rows, cols, batch, nat.ptr(scratch), C.c_size_t(nbytes), stream_ptr()))
    if single:
        oro = None if oro is None else oro[0]
        use = None if use is None else use[0]
    return oro, use
