"""ORACLE -- ctypes front end of oracle/liboracle_ssrs.so (test infrastructure
only; see oracle/ssrs_oracle.c).  Used by tests/ for parity at sizes the pure
python oracle cannot reach in seconds, and by bench.py's cpu_baseline leg."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, 'liboracle_ssrs.so')
_lib = None


class OrcParams(C.Structure):
    _fields_ = [('rows', C.c_int32), ('cols', C.c_int32), ('burnin', C.c_int32),
                ('memory', C.c_int32), ('max_moves', C.c_double),
                ('nu', C.c_double), ('prior', C.c_double * 9)]


def build(force=False):
    src = os.path.join(_HERE, 'ssrs_oracle.c')
    if (force or not os.path.exists(_LIB_PATH)
            or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src)):
        subprocess.check_call(['make', '-s', '-C', _HERE, 'all'])
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_LIB_PATH)
        _lib.orc_uniform.restype = C.c_double
        _lib.orc_uniform.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64]
        _lib.orc_simulate_tracks.restype = C.c_int64
        _lib.orc_simulate_tracks_ids.restype = C.c_int64
        _lib.orc_num_threads.restype = C.c_int
    return _lib


def _ptr(a, ct):
    return a.ctypes.data_as(C.POINTER(ct)) if a is not None else None


def make_params(grid_shape, move_dirn, memory_parameter=1, scaling_parameter=1.0):
    """Derive the per-run constants exactly as movmodel.py:275-277 does; the
    prior is computed with numpy cos like the reference (movmodel.py:247-257)."""
    from . import ssrs_oracle as orc
    rows, cols = grid_shape
    p = OrcParams()
    p.rows, p.cols = rows, cols
    p.burnin = int(min(rows, cols) / 10)
    p.memory = int(memory_parameter)
    p.max_moves = rows / 2 * cols / 2
    p.nu = float(scaling_parameter)
    prior = orc.get_directional_probs(move_dirn * np.pi / 180.)
    for k in range(9):
        p.prior[k] = float(prior[k])
    return p


def uniform(seed, track, step):
    return lib().orc_uniform(seed, track, step)


def simulate_tracks(move_dirn, starts, grid_shape, memory_parameter=1,
                    scaling_parameter=1.0, updraft=None, potential=None, seed=0,
                    track_id_base=0, want_traj=True, want_hist=True, nthreads=0, max_moves=None, track_ids=None):
    """Returns dict(lengths int32[n], ends int16[n,2], hist uint32[R,C] | None,
    tracks list[int16[n_i,2]] | None, steps int).  `track_ids` (uint64[n]): the global ids of an arbitrary
    subset of a batch (default: track_id_base + 0..n-1)."""
    L = lib()
    rows, cols = grid_shape
    starts = np.ascontiguousarray(np.asarray(starts, dtype=np.int32).reshape(-1, 2))
    n = starts.shape[0]
    p = make_params(grid_shape, move_dirn, memory_parameter, scaling_parameter)
    if max_moves is not None:      # test hook: cap below the reference's R/2 * C/2 (movmodel.py:277)
        p.max_moves = float(max_moves)
    upd = None if updraft is None else np.ascontiguousarray(updraft, dtype=np.float64)
    pot = None if potential is None else np.ascontiguousarray(potential, dtype=np.float32)
    if upd is not None:
        assert upd.shape == (rows, cols)
    if pot is not None:
        assert pot.shape == (rows, cols)
    lengths = np.zeros(n, dtype=np.int32)
    ends = np.zeros((n, 2), dtype=np.int16)
    hist = np.zeros((rows, cols), dtype=np.uint32) if want_hist else None
    ids = None
    if track_ids is not None:
        ids = np.ascontiguousarray(track_ids, dtype=np.uint64)
        assert ids.shape == (n,)
    args = [C.byref(p), _ptr(upd, C.c_double), _ptr(pot, C.c_float),
            _ptr(starts, C.c_int32), C.c_int64(n), C.c_uint64(seed),
            C.c_uint64(track_id_base), _ptr(ids, C.c_uint64)]
    traj = offs = None
    if want_traj:   # pass 1: lengths; pass 2: trajectories
        steps = L.orc_simulate_tracks_ids(*args, None, None, _ptr(lengths, C.c_int32),
                                      None, None, C.c_int(nthreads))
        assert steps >= 0
        offs = np.zeros(n + 1, dtype=np.int64)
        np.cumsum(lengths, out=offs[1:])
        traj = np.zeros((int(offs[-1]), 2), dtype=np.int16)
    steps = L.orc_simulate_tracks_ids(*args, _ptr(hist, C.c_uint32), _ptr(ends, C.c_int16),
                                  _ptr(lengths, C.c_int32), _ptr(traj, C.c_int16),
                                  _ptr(offs, C.c_int64), C.c_int(nthreads))
    if steps < 0:
        raise ValueError('orc_simulate_tracks: bad arguments')
    tracks = None
    if want_traj:
        tracks = [traj[offs[t]:offs[t + 1]] for t in range(n)]
    return dict(lengths=lengths, ends=ends, hist=hist, tracks=tracks, steps=int(steps))


def slope_aspect(z, res):
    z = np.ascontiguousarray(z, dtype=np.float64)
    s = np.empty_like(z)
    a = np.empty_like(z)
    lib().orc_slope_aspect(_ptr(z, C.c_double), C.c_int(z.shape[0]), C.c_int(z.shape[1]),
                           C.c_double(res), _ptr(s, C.c_double), _ptr(a, C.c_double))
    return s, a


def orographic(slope, aspect, wspeed, wdirn, min_val=0.0):
    slope = np.ascontiguousarray(slope, dtype=np.float64)
    aspect = np.ascontiguousarray(aspect, dtype=np.float64)
    ws = wd = None
    ws0 = wd0 = 0.0
    if np.ndim(wspeed) == 0:
        ws0 = float(wspeed)
    else:
        ws = np.ascontiguousarray(wspeed, dtype=np.float64)
    if np.ndim(wdirn) == 0:
        wd0 = float(wdirn)
    else:
        wd = np.ascontiguousarray(wdirn, dtype=np.float64)
    out64 = np.empty_like(slope)
    out32 = np.empty(slope.shape, dtype=np.float32)
    lib().orc_orographic(_ptr(slope, C.c_double), _ptr(aspect, C.c_double),
                         _ptr(ws, C.c_double), _ptr(wd, C.c_double), C.c_double(ws0),
                         C.c_double(wd0), C.c_double(min_val), C.c_size_t(slope.size),
                         _ptr(out64, C.c_double), _ptr(out32, C.c_float))
    return out64, out32


def threshold(oro32, thr):
    oro32 = np.ascontiguousarray(oro32, dtype=np.float32)
    out = np.empty(oro32.shape, dtype=np.float64)
    lib().orc_threshold(_ptr(oro32, C.c_float), C.c_double(thr), C.c_size_t(oro32.size),
                        _ptr(out, C.c_double))
    return out


def cell_weights(updraft, potential, row, col):
    upd = np.ascontiguousarray(updraft, dtype=np.float64)
    pot = None if potential is None else np.ascontiguousarray(potential, dtype=np.float32)
    out = np.empty(8, dtype=np.float64)
    lib().orc_cell_weights(_ptr(upd, C.c_double), _ptr(pot, C.c_float),
                           C.c_int(upd.shape[0]), C.c_int(upd.shape[1]),
                           C.c_int(row), C.c_int(col), _ptr(out, C.c_double))
    return out


def smooth_presence(count, krad):
    count = np.ascontiguousarray(count, dtype=np.uint32)
    out = np.empty(count.shape, dtype=np.float32)
    lib().orc_smooth_presence(_ptr(count, C.c_uint32), C.c_int(count.shape[0]),
                              C.c_int(count.shape[1]), C.c_int(krad), _ptr(out, C.c_float))
    return out


def num_threads():
    return lib().orc_num_threads()
