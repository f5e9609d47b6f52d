"""ctypes binding of libssrs_hip.so (include/ssrs_hip.h).

There is NO CPU fallback: if the library is missing or a call fails, the
functions raise.  The library is built in-tree by ssrs_amd/csrc/build.py
(hipcc, gfx950) and travels with the source tree.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('SSRS_HIP_LIB') or os.path.join(_HERE, 'libssrs_hip.so')   # (override: timing probes only)

SSRS_F32, SSRS_F64 = 0, 1
SSRS_OK, SSRS_ERR_INVALID, SSRS_ERR_HIP, SSRS_ERR_START = 0, -1, -2, -3
SSRS_TRACKS_PROFILE = 1
SSRS_TRACKS_EXACT_ONLY = 2
SSRS_TRACKS_NO_SCHEDULE = 4
SSRS_TRACKS_NO_BINNING = 8
SSRS_TRACKS_RING_TABLE = 16
SSRS_TRACKS_SCATTERED = 32
SSRS_TRACKS_NO_SCATTERED = 64
SSRS_TRACKS_THR_TABLE = 128
SSRS_SOLVE_NO_AMG = 1

EXPORTS = (
    'ssrs_version', 'ssrs_build_flags', 'ssrs_last_error', 'ssrs_device_info', 'ssrs_slope_aspect',
    'ssrs_orographic_updraft', 'ssrs_threshold_updraft', 'ssrs_updraft_from_dem',
    'ssrs_lattice_workspace_bytes', 'ssrs_updraft_from_dem_lattice',
    'ssrs_wind_from_lattice', 'ssrs_wind_triangles_workspace_bytes', 'ssrs_wind_from_triangles', 'ssrs_thermal_seeds', 'ssrs_blur_workspace_bytes',
    'ssrs_gaussian_blur', 'ssrs_track_params_init', 'ssrs_transition_table_build',
    'ssrs_transition_ring_bytes', 'ssrs_transition_ring_build',
    'ssrs_transition_thr_bytes', 'ssrs_transition_thr_build',
    'ssrs_tracks_workspace_bytes', 'ssrs_tracks_workspace_bytes_ex', 'ssrs_tracks_simulate', 'ssrs_uniform_selftest',
    'ssrs_traj_recorder_create', 'ssrs_traj_recorder_destroy', 'ssrs_traj_recorder_complete',
    'ssrs_traj_recorder_used', 'ssrs_tracks_simulate_rec', 'ssrs_tracks_gather', 'ssrs_tracks_simulate_h64',
    'ssrs_hist_reduce', 'ssrs_presence_count', 'ssrs_presence_workspace_bytes', 'ssrs_presence_smooth',
    'ssrs_presence_smooth_u64',
    'ssrs_presence_normalise_add', 'ssrs_presence_normalise_f32',
    'ssrs_potential_workspace_bytes', 'ssrs_potential_solve',
)


class SsrsTrackParams(C.Structure):
    _fields_ = [('rows', C.c_int32), ('cols', C.c_int32), ('burnin', C.c_int32),
                ('memory_parameter', C.c_int32), ('max_moves', C.c_int64),
                ('scaling_parameter', C.c_double), ('prior', C.c_double * 9),
                ('steps_per_launch', C.c_int32), ('flags', C.c_int32)]


class SsrsTrackStats(C.Structure):
    _fields_ = [('total_steps', C.c_int64), ('launches', C.c_int32),
                ('kernel_ms', C.c_float), ('wall_ms', C.c_float), ('hist_ms', C.c_float),
                ('window_launches', C.c_int32), ('tile_launches', C.c_int32),
                ('block_window_launches', C.c_int32), ('wander_sorts', C.c_int32),
                ('timed_launches', C.c_int32), ('first_move_ms', C.c_float),
                ('block_window_ms', C.c_float), ('block_window_timed', C.c_int32),
                ('block_window_steps', C.c_int64), ('roam_launches', C.c_int32), ('reserved0', C.c_int32),
                ('roam_wave_pairs', C.c_int64), ('roam_slow_wave_pairs', C.c_int64),
                ('roam_shuffles', C.c_int32), ('roam_wide_launches', C.c_int32)]


class SsrsSolveStats(C.Structure):
    _fields_ = [('iterations', C.c_int32), ('converged', C.c_int32),
                ('residual', C.c_double), ('kernel_ms', C.c_float),
                ('amg_levels', C.c_int32), ('amg_coarsest', C.c_int32),
                ('setup_ms', C.c_float), ('workspace_used', C.c_uint64)]


class SsrsError(RuntimeError):
    def __init__(self, code, message):
        super().__init__(f'libssrs_hip error {code}: {message}')
        self.code = code


_lib = None


def lib():
    """Load libssrs_hip.so once; raise loudly when it is absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f'{LIB_PATH} is missing: build it with `python ssrs_amd/csrc/build.py` '
                '(hipcc, gfx950). ssrs_amd has no CPU fallback.')
        L = C.CDLL(LIB_PATH)
        L.ssrs_version.restype = C.c_int
        if L.ssrs_build_flags() & 1 and not os.environ.get('SSRS_ALLOW_PROBE_LIB'):
            raise ImportError(f'{LIB_PATH} is a timing-probe build (results are wrong on purpose); '
                              'set SSRS_ALLOW_PROBE_LIB=1 for timing runs only')
        L.ssrs_last_error.restype = C.c_char_p
        L.ssrs_tracks_workspace_bytes.restype = C.c_size_t
        L.ssrs_tracks_workspace_bytes.argtypes = [C.c_int64]
        L.ssrs_lattice_workspace_bytes.restype = C.c_size_t
        L.ssrs_wind_triangles_workspace_bytes.restype = C.c_size_t
        L.ssrs_wind_triangles_workspace_bytes.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int]
        L.ssrs_lattice_workspace_bytes.argtypes = [C.c_int, C.c_int, C.c_int]
        L.ssrs_tracks_workspace_bytes_ex.restype = C.c_size_t
        L.ssrs_tracks_workspace_bytes_ex.argtypes = [C.c_int64, C.c_int, C.c_int, C.c_int]
        L.ssrs_traj_recorder_create.restype = C.c_void_p
        L.ssrs_traj_recorder_create.argtypes = [C.c_void_p, C.c_size_t]
        L.ssrs_traj_recorder_destroy.restype = None
        L.ssrs_traj_recorder_destroy.argtypes = [C.c_void_p]
        L.ssrs_traj_recorder_complete.argtypes = [C.c_void_p]
        L.ssrs_traj_recorder_used.restype = C.c_size_t
        L.ssrs_traj_recorder_used.argtypes = [C.c_void_p]
        L.ssrs_transition_thr_bytes.restype = C.c_size_t
        L.ssrs_transition_thr_bytes.argtypes = [C.c_int, C.c_int]
        L.ssrs_transition_ring_bytes.restype = C.c_size_t
        L.ssrs_transition_ring_bytes.argtypes = [C.c_int, C.c_int]
        if hasattr(L, 'ssrs_presence_workspace_bytes'):
            L.ssrs_presence_workspace_bytes.restype = C.c_size_t
            L.ssrs_presence_workspace_bytes.argtypes = [C.c_int, C.c_int, C.c_int]
        if hasattr(L, 'ssrs_blur_workspace_bytes'):
            L.ssrs_blur_workspace_bytes.restype = C.c_size_t
            L.ssrs_blur_workspace_bytes.argtypes = [C.c_int, C.c_int, C.c_double]
        if hasattr(L, 'ssrs_potential_workspace_bytes'):
            L.ssrs_potential_workspace_bytes.restype = C.c_size_t
            L.ssrs_potential_workspace_bytes.argtypes = [C.c_int, C.c_int]
        _lib = L
    return _lib


def check(rc):
    if rc != SSRS_OK:
        msg = lib().ssrs_last_error().decode('utf-8', 'replace')
        if rc == SSRS_ERR_INVALID or rc == SSRS_ERR_START:
            raise ValueError(f'libssrs_hip: {msg}')
        raise SsrsError(rc, msg)


def ptr(t):
    """void* of a torch tensor (None -> NULL)."""
    return C.c_void_p(0 if t is None else t.data_ptr())
