"""Track stepper host side (K2/K3), behind the reference's function names
(/root/reference/ssrs/movmodel.py).  The per-step work runs in
libssrs_hip.so; this module only prepares arguments (start cells, the
directional prior, which needs the host's cos) and owns the device buffers.

Random stream: the reference draws from numpy's global MT19937 inside a fork
pool (movmodel.py:312, simulator.py:360), which is irreproducible; here
u(seed, track_id, step) is Philox4x32-10 (include/ssrs_hip.h "Uniform
contract"), so results do not depend on how tracks are sharded.
"""
import ctypes as C
import os
import threading
from math import floor, ceil

import numpy as np
import torch

from . import _native as nat
from ._device import device, stream_ptr, to_dev, is_tensor

# movmodel.py:131-141
neighbour_deltas = [np.array([k // 3 - 1, k % 3 - 1]) for k in range(9)]
neighbour_delta_norms_inv = np.array(
    [[1 / np.sqrt(2.), 1., 1 / np.sqrt(2.)], [1., 0., 1.],
     [1 / np.sqrt(2.), 1., 1 / np.sqrt(2.)]], dtype=np.float32)


def get_starting_indices(ntracks, sbounds, stype, twidth, tres):
    """movmodel.py:144-182 -> (rows, cols) int arrays.  Host-side integer logic;
    'random' consumes numpy's legacy global stream with the same single
    np.random.randint call as the reference, so a seeded run picks the same
    start cells."""
    if (sbounds[1] < sbounds[0] or sbounds[3] < sbounds[2] or sbounds[0] < 0.
            or sbounds[2] < 0. or sbounds[1] > twidth[0] or sbounds[3] > twidth[1]):
        raise ValueError('track_start_region incompatible with terrain_width!')
    res_km = tres / 1000.
    x_max = ceil(twidth[0] / res_km)
    y_max = ceil(twidth[1] / res_km)
    x_lo = min(max(floor(sbounds[0] / res_km) - 1, 1), x_max - 2)
    x_hi = max(min(ceil(sbounds[1] / res_km), x_max - 1), 2)
    y_lo = min(max(floor(sbounds[2] / res_km) - 1, 1), y_max - 2)
    y_hi = max(min(ceil(sbounds[3] / res_km), y_max - 1), 2)
    nx, ny = x_hi - x_lo, y_hi - y_lo
    base_count = nx * ny

    def cell(idx):      # candidate list is x-major: idx = ix * ny + iy
        idx = np.asarray(idx, dtype=np.int64)
        return y_lo + idx % ny, x_lo + idx // ny

    if stype == 'structured':
        pick = np.round(np.linspace(0, base_count - 1, ntracks % base_count)).astype(np.int64)
        if ntracks > base_count:
            whole = np.tile(np.arange(base_count, dtype=np.int64), ntracks // base_count)
            idx = np.concatenate((whole, pick))
        else:
            idx = pick
    elif stype == 'random':
        idx = np.random.randint(0, base_count, ntracks)
    else:
        raise ValueError((f'Model:Invalid sim_start_type of {stype}\n'
                          'Options: structured, random'))
    rows, cols = cell(idx)
    return rows.astype(int), cols.astype(int)


def get_directional_probs(theta):
    """movmodel.py:247-257: cos lobe toward heading theta (radians), entries
    below 0.01 zeroed, row order flipped so that +row = north."""
    ang = np.array([[3 * np.pi / 4, np.pi, 5 * np.pi / 4],
                    [np.pi / 2, np.nan, 3 * np.pi / 2],
                    [np.pi / 4, 0., 7 * np.pi / 4]])
    lobe = np.cos(ang + theta)
    lobe[1, 1] = 0.
    lobe[lobe < 0.01] = 0.
    return lobe.flatten()


def make_track_params(grid_shape, move_dirn, memory_parameter=1, scaling_parameter=1.,
                      steps_per_launch=0, profile=False, exact_only=False, schedule=True,
                      binning=True, ring=False, scattered=None, thr=False):
    rows, cols = int(grid_shape[0]), int(grid_shape[1])
    p = nat.SsrsTrackParams()
    nat.check(nat.lib().ssrs_track_params_init(C.byref(p), rows, cols, int(memory_parameter),
                                               C.c_double(scaling_parameter)))
    prior = get_directional_probs(move_dirn * np.pi / 180.)
    for k in range(9):
        p.prior[k] = float(prior[k])
    p.steps_per_launch = int(steps_per_launch)
    p.flags = (nat.SSRS_TRACKS_PROFILE if profile else 0) | \
        (nat.SSRS_TRACKS_EXACT_ONLY if exact_only else 0) | \
        (0 if schedule else nat.SSRS_TRACKS_NO_SCHEDULE) | \
        (0 if binning else nat.SSRS_TRACKS_NO_BINNING) | \
        (nat.SSRS_TRACKS_RING_TABLE if ring else 0) | \
        (nat.SSRS_TRACKS_THR_TABLE if thr else 0) | \
        (0 if scattered is None else (nat.SSRS_TRACKS_SCATTERED if scattered else nat.SSRS_TRACKS_NO_SCATTERED))
    return p


THR_MAX_CELLS = 1 << 26        # the threshold table is addressed with 32-bit offsets


def table_kind(table, rows=None, cols=None):
    """'f64' | 'ring' | 'thr' | None for a tensor made by build_transition_table.  The kind is the
    tensor's dtype (float64 rows, float32 ring records, int32 threshold dwords), so it survives
    clone / to / pickling; with rows, cols the size is checked against the library's layout."""
    if table is None:
        return None
    kind = {torch.float64: 'f64', torch.float32: 'ring', torch.int32: 'thr'}.get(table.dtype)
    if kind is None:
        raise ValueError(f'a transition table is float64, float32 (ring) or int32 (threshold), not {table.dtype}')
    if rows is not None:
        want = {'f64': rows * cols * 8, 'ring': nat.lib().ssrs_transition_ring_bytes(rows, cols) // 4,
                'thr': nat.lib().ssrs_transition_thr_bytes(rows, cols) // 4}[kind]
        if table.numel() != want or not table.is_contiguous():
            raise ValueError(f'this {table.dtype} tensor is not a {kind} table of a {rows} x {cols} raster '
                             f'({table.numel()} elements, expected {want} contiguous)')
    return kind


def build_transition_table(updraft, potential, ring=False, thr=False, move_dirn=None):
    """Per-cell move weights for the table stepper: 8 x f64 per cell (any
    memory_parameter); with ring=True the f32 ring table (10 x f32 per cell, a
    1-D float32 tensor) of the three-candidate stepper; with thr=True the threshold table
    (8 dwords per cell: the two 16-bit decision thresholds for each of the eight last moves, a 1-D
    int32 tensor) of the threshold stepper -- it belongs to ONE heading, `move_dirn` (degrees),
    recorded in the table's header and checked by the stepper.  Both compact forms serve
    memory_parameter 1 / nu 1 only."""
    upd = to_dev(updraft, torch.float64)
    pot = to_dev(potential, torch.float32)
    rows, cols = int(upd.shape[0]), int(upd.shape[1])
    if pot is not None and tuple(pot.shape) != (rows, cols):
        raise ValueError('updraft and potential shapes differ')
    if thr:
        if move_dirn is None:
            raise ValueError('the threshold table needs move_dirn (it holds the prior fallback of that heading)')
        prior = np.ascontiguousarray(get_directional_probs(float(move_dirn) * np.pi / 180.), dtype=np.float64)
        nbytes = nat.lib().ssrs_transition_thr_bytes(rows, cols)
        # int32: the kind travels with the tensor (the ring table is float32); the heading travels
        # in the table's own header, which ssrs_tracks_simulate checks against its prior
        table = torch.empty(nbytes // 4, dtype=torch.int32, device=upd.device)
        nat.check(nat.lib().ssrs_transition_thr_build(
            nat.ptr(upd), nat.ptr(pot), prior.ctypes.data_as(C.POINTER(C.c_double)), nat.ptr(table),
            rows, cols, stream_ptr()))
        return table
    if ring:
        nbytes = nat.lib().ssrs_transition_ring_bytes(rows, cols)
        table = torch.empty(nbytes // 4, dtype=torch.float32, device=upd.device)
        nat.check(nat.lib().ssrs_transition_ring_build(
            nat.ptr(upd), nat.ptr(pot), nat.ptr(table), rows, cols, stream_ptr()))
        return table
    table = torch.empty((rows, cols, 8), dtype=torch.float64, device=upd.device)
    nat.check(nat.lib().ssrs_transition_table_build(
        nat.ptr(upd), nat.ptr(pot), nat.ptr(table), rows, cols, stream_ptr()))
    return table


def ring_table_applies(memory_parameter, scaling_parameter, want_tracks, exact_only, steps_per_launch):
    """The f32 ring table serves the reference's default movement model only.
    `want_tracks` here means the two-pass trajectory form (the generic kernel writes the
    points itself); recorded trajectories (simulate_tracks' default) keep the ring path."""
    return (int(memory_parameter) == 1 and float(scaling_parameter) == 1.0 and not want_tracks
            and not exact_only and int(steps_per_launch) % 2 == 0)


def default_record_pool_bytes(n, rows, cols):
    """Pool for recorded trajectories: a launch of S = 512 steps needs 4 B x S x live slots,
    so a batch that crosses the raster in ~rows steps needs about n x rows x 4 B x 1.5;
    bounded by a quarter of the free HBM (the pool is only scratch: when it runs out the
    run falls back to the two-pass form)."""
    free, _ = torch.cuda.mem_get_info()
    want = max(int(n) * 4 * max(int(rows), int(cols)) * 3, 1 << 30)     # small batches: lists pad to 256 slots
    return int(max(1 << 20, min(want, free // 4))) // 256 * 256


# The stepper's scratch is large (8.3 KB per track + the pair / fine tables and histogram copies behind it:
# 13-15 GB for a million tracks at 5000 x 6000) and lives only for the duration of a call.  Allocating and
# freeing it around every call left the caching allocator with a freed block of that size which smaller
# requests of the next pass then split, so that the next 13 GB request went to hipMalloc again (~80 ms,
# inside somebody's timed region: profiles/r03_notes.md, the 105.8 ms outlier of round 2's 1 M-track
# sweep).  Each host thread keeps its last workspace per device instead (calls of one thread are serial
# and return synchronised; Simulator's worker threads each hold their own).
_ws_local = threading.local()


def _workspace(nbytes, dev):
    if os.environ.get('SSRS_NO_WS_CACHE'):
        return torch.empty(nbytes, dtype=torch.uint8, device=dev)
    cache = getattr(_ws_local, 'cache', None)
    if cache is None:
        cache = _ws_local.cache = {}
    key = (dev.type, dev.index)
    buf = cache.get(key)
    if buf is None or buf.numel() < nbytes or buf.numel() > 4 * max(nbytes, 1 << 26):
        cache.pop(key, None)
        buf = None                                  # (the old block goes back before the new one is asked for)
        buf = cache[key] = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    return buf[:nbytes]


def release_workspaces():
    """Drop this thread's cached stepper workspaces."""
    _ws_local.cache = {}


class TrackBatch:
    """Result of simulate_tracks: device tensors + lazy host views."""

    def __init__(self, lengths, ends, hist, traj, offsets, stats, replay=None):
        self.lengths = lengths        # int32 (n)          trajectory points per track
        self.ends = ends              # int16 (n, 2)       last point [row, col]
        self.hist = hist              # uint32-in-int32 (rows, cols) or None
        self.traj = traj              # int16 (sum lengths, 2) or None
        self.offsets = offsets        # int64 (n + 1) or None
        self.stats = stats            # dict(total_steps, launches, kernel_ms, wall_ms)
        self._replay = replay         # chunked trajectories: callable(t0, t1) -> int16 (points, 2) host array

    @property
    def total_points(self):
        """Sum of the trajectory lengths (4 bytes each as int16 pairs)."""
        return int(self.lengths.sum(dtype=torch.int64).item())

    def iter_tracks(self):
        """The tracks one by one, int16 (n_i, 2) each, in track order.  When the trajectories did not fit
        the device budget (Sum lengths x 4 B: 1 TB for 100k tracks on the solved 10 m field, where a
        third of them take max_moves = 7.5e6) they are produced range by range: the lengths of the
        finished pass give the ranges, and each range is stepped again with the generic kernel writing
        every point at its offset (the same counter-based streams: the same tracks)."""
        if self.traj is not None:
            traj = self.traj.cpu().numpy()
            off = self.offsets.cpu().numpy()
            for i in range(off.size - 1):
                yield traj[off[i]:off[i + 1]]
            return
        if self._replay is None:
            raise ValueError('simulate_tracks was called with want_tracks=False')
        lengths = self.lengths.cpu().numpy().astype(np.int64)
        budget_points = max(int(self.stats['traj_chunk_bytes']) // 4, 1)
        t0, n = 0, lengths.size
        while t0 < n:
            t1, pts = t0, 0
            while t1 < n and (t1 == t0 or pts + lengths[t1] <= budget_points):
                pts += lengths[t1]
                t1 += 1
            traj = self._replay(t0, t1)
            off = np.concatenate(([0], np.cumsum(lengths[t0:t1])))
            for i in range(t1 - t0):
                yield traj[off[i]:off[i + 1]]
            t0 = t1

    def tracks(self, max_bytes=64 << 30):
        """List[int16 (n_i, 2)] like the reference's pool.map result (iter_tracks() streams them)."""
        need = self.total_points * 4
        if need > max_bytes:
            raise MemoryError(f'the trajectories of this batch are {need / 2**30:.1f} GiB (Sum lengths x 4 B): '
                              'stream them with iter_tracks() or run with save_tracks=False')
        return list(self.iter_tracks())


def simulate_tracks(move_dirn, starts, grid_shape, memory_parameter=1,
                    scaling_parameter=1., updraft_field=None, potential_field=None, *,
                    seed=0, track_id_base=0, table=None, use_table=None, hist=None,
                    want_hist=True, want_tracks=False, steps_per_launch=0, profile=False,
                    exact_only=False, schedule=True, binning=True, ring=None, scattered=None,
                    max_moves=None, record=True, record_pool_bytes=None, thr=None, traj_budget_bytes=None, hist64=False):
    """generate_simulated_tracks for a whole batch (movmodel.py:264-318 under
    simulator.py:360-369) + presence histogram (movmodel.py:410-419).

    starts: int (n, 2) [row, col].  updraft_field f64 / potential_field f32
    rasters (numpy or CUDA tensors); both None = 'drw'.  `table` (from
    build_transition_table) or use_table=True selects the one-fetch-per-step
    path; default: table when it pays (many steps per cell).  A float32
    `table` is the ring table (build_transition_table(..., ring=True)), an int32 one the
    threshold table; when the
    table is built here, the threshold table (build_transition_table(..., thr=True)) is
    picked whenever it applies (memory 1, nu 1, rows * cols < 2^26); ring=True / thr=False
    ask for the ring table, ring=False for the f64 table.
    `hist` (int32/uint32 CUDA tensor) is accumulated into when given.  hist64=True (no trajectories): the counts come back
    as an int64 tensor -- the kernels count into a uint32 scratch raster that the library empties into it every other batch
    (ssrs_tracks_simulate_h64): the trap cells of a solved 10 m field pass 2^32 visits from ~250 000 tracks of one call on;
    `hist`, when given, must then be int64 and is accumulated into.

    want_tracks: trajectories (TrackBatch.tracks()).  record=True (default) keeps every
    launch's visited cells in a device pool and assembles the trajectories afterwards
    (ssrs_tracks_simulate_rec + ssrs_tracks_gather): ONE simulation pass on whichever
    stepper path applies.  record=False, or a pool that ran out, takes the two-pass
    form: lengths first, then the same counter-based streams again with the generic
    kernel writing each point at its final offset.  When Sum lengths x 4 B exceeds
    `traj_budget_bytes` (default: a third of the free HBM, at most 16 GiB) no trajectory tensor is
    allocated at all: TrackBatch.iter_tracks() then steps the batch again range by range.
    """
    rows, cols = int(grid_shape[0]), int(grid_shape[1])
    dev = device()
    st = to_dev(np.asarray(starts) if not is_tensor(starts) else starts, torch.int32)
    st = st.reshape(-1, 2).contiguous()
    n = int(st.shape[0])
    upd = to_dev(updraft_field, torch.float64)
    pot = to_dev(potential_field, torch.float32)
    for name, f in (('updraft_field', upd), ('potential_field', pot)):
        if f is not None and tuple(f.shape) != (rows, cols):
            raise ValueError(f'{name} shape {tuple(f.shape)} != grid_shape {(rows, cols)}')
    if pot is not None and upd is None:
        raise ValueError('potential_field needs updraft_field')
    two_pass = bool(want_tracks) and not record
    if table is None and upd is not None:
        if use_table is None:
            # building the table is one streaming pass over the raster (0.5 ms at 5000 x 6000);
            # a batch pays for it with its first few hundred thousand steps -- and tracks that
            # wander in a basin of the potential take millions each (G11), so anything but a
            # handful of tracks takes the table
            use_table = n * max(rows, cols) >= rows * cols // 64 or n >= 4096
        if use_table:
            f32_ok = ring_table_applies(memory_parameter, scaling_parameter, two_pass,
                                        exact_only, steps_per_launch)
            if thr is None:
                thr = f32_ok and ring is None and rows * cols < THR_MAX_CELLS
            if ring is None:
                ring = f32_ok and not thr
            table = build_transition_table(upd, pot, ring=bool(ring) and not thr, thr=bool(thr),
                                           move_dirn=move_dirn)
    kind = table_kind(table, rows, cols)
    is_thr = kind == 'thr'
    is_ring = kind in ('ring', 'thr')          # the f32 / dword family: default movement model only
    if is_ring and not ring_table_applies(memory_parameter, scaling_parameter, two_pass,
                                          exact_only, steps_per_launch):
        raise ValueError('the ring table needs memory_parameter 1, scaling_parameter 1, no '
                         'two-pass trajectory output, exact_only=False and an even steps_per_launch')
    p = make_track_params((rows, cols), move_dirn, memory_parameter, scaling_parameter,
                          steps_per_launch, profile, exact_only, schedule, binning,
                          ring=is_ring and not is_thr, scattered=scattered, thr=is_thr)
    if max_moves is not None:          # probe hook: cap below the reference's R/2 * C/2 (movmodel.py:277)
        p.max_moves = int(max_moves)
    hist_scratch = None
    if hist64:
        if want_tracks:
            raise ValueError('hist64 goes with want_tracks=False')
        if hist is None:
            hist = torch.zeros((rows, cols), dtype=torch.int64, device=dev)
        elif hist.dtype != torch.int64:
            raise ValueError('hist64=True: `hist` must be an int64 tensor')
        hist_scratch = torch.zeros((rows, cols), dtype=torch.int32, device=dev)
    if hist is None and want_hist:
        hist = torch.zeros((rows, cols), dtype=torch.int32, device=dev)
    lengths = torch.empty(n, dtype=torch.int32, device=dev)
    ends = torch.empty((n, 2), dtype=torch.int16, device=dev)
    # room for private histogram copies (used only once a batch is scattered): up to 64,
    # within 4 GB
    copies = 0
    if hist is not None:
        copies = int(min(64, (4 << 30) // (rows * cols * 4)))
        copies = copies if copies >= 2 else 0
    ws_bytes = nat.lib().ssrs_tracks_workspace_bytes_ex(n, rows, cols, copies)
    ws = _workspace(ws_bytes, dev)
    stats = nat.SsrsTrackStats()

    def run(hist_t, traj_t, off_t):
        nat.check(nat.lib().ssrs_tracks_simulate(
            C.byref(p), nat.ptr(upd), nat.ptr(pot), nat.ptr(table), nat.ptr(st),
            C.c_int64(n), C.c_uint64(int(seed) & 0xFFFFFFFFFFFFFFFF),
            C.c_uint64(int(track_id_base)), nat.ptr(hist_t), nat.ptr(ends),
            nat.ptr(lengths), nat.ptr(traj_t), nat.ptr(off_t), nat.ptr(ws),
            C.c_size_t(ws_bytes), C.byref(stats), stream_ptr()))

    traj = offsets = None
    recorded = simulated = False
    if traj_budget_bytes is None:
        traj_budget_bytes = min(16 << 30, torch.cuda.mem_get_info()[0] // 3)
    if want_tracks and record and n > 0:
        # one pass: every launch's visits stay in the pool, the gather assembles them
        pool_bytes = int(record_pool_bytes) if record_pool_bytes is not None else \
            default_record_pool_bytes(n, rows, cols)
        pool = torch.empty(max(256, pool_bytes // 256 * 256), dtype=torch.uint8, device=dev)
        rec = nat.lib().ssrs_traj_recorder_create(nat.ptr(pool), C.c_size_t(pool.numel()))
        if not rec:
            nat.check(nat.SSRS_ERR_INVALID)
        try:
            nat.check(nat.lib().ssrs_tracks_simulate_rec(
                C.byref(p), nat.ptr(upd), nat.ptr(pot), nat.ptr(table), nat.ptr(st),
                C.c_int64(n), C.c_uint64(int(seed) & 0xFFFFFFFFFFFFFFFF),
                C.c_uint64(int(track_id_base)), nat.ptr(hist), nat.ptr(ends),
                nat.ptr(lengths), C.c_void_p(rec), nat.ptr(ws), C.c_size_t(ws_bytes),
                C.byref(stats), stream_ptr()))
            if nat.lib().ssrs_traj_recorder_complete(C.c_void_p(rec)) and \
                    int(lengths.sum(dtype=torch.int64).item()) * 4 <= int(traj_budget_bytes):
                offsets = torch.zeros(n + 1, dtype=torch.int64, device=dev)
                torch.cumsum(lengths, 0, out=offsets[1:])
                total = int(offsets[-1].item())
                traj = torch.empty((total, 2), dtype=torch.int16, device=dev)
                cursor = torch.empty(n, dtype=torch.int32, device=dev)
                nat.check(nat.lib().ssrs_tracks_gather(
                    C.c_void_p(rec), nat.ptr(st), C.c_int64(n), nat.ptr(offsets), nat.ptr(traj),
                    nat.ptr(cursor), C.c_size_t(4 * n), stream_ptr()))
                torch.cuda.current_stream().synchronize()      # the pool dies with this scope
                recorded = True
            simulated = True                  # lengths, end cells and the histogram are final
        finally:
            nat.lib().ssrs_traj_recorder_destroy(C.c_void_p(rec))
            del pool
        if not recorded and is_ring:
            # the two-pass form runs the generic kernel, which reads the f64 table
            table = build_transition_table(upd, pot, ring=False)
            p = make_track_params((rows, cols), move_dirn, memory_parameter, scaling_parameter,
                                  steps_per_launch, profile, exact_only, schedule, binning, ring=False,
                                  scattered=scattered)
            if max_moves is not None:
                p.max_moves = int(max_moves)
    replay = None
    if want_tracks and not recorded:
        # pass 1: lengths only (done already when a recorded run ran out of pool); pass 2
        # replays the same counter-based streams and writes every point at its final
        # offset (no per-track cap).
        if not simulated:
            run(hist, None, None)
        offsets = torch.zeros(n + 1, dtype=torch.int64, device=dev)
        torch.cumsum(lengths, 0, out=offsets[1:])
        total = int(offsets[-1].item())
        if total * 4 <= int(traj_budget_bytes):
            traj = torch.empty((total, 2), dtype=torch.int16, device=dev)
            run(None, traj, offsets)
        else:
            # too long for one tensor (the reference would hold them as host lists, simulator.py:360-385):
            # range by range on demand, each range one second pass of its own tracks
            offsets = None
            two_table, two_p = table, p

            def replay(t0, t1):
                sub = st[t0:t1].contiguous()
                m = t1 - t0
                sub_len = torch.empty(m, dtype=torch.int32, device=dev)
                sub_end = torch.empty((m, 2), dtype=torch.int16, device=dev)
                off = torch.zeros(m + 1, dtype=torch.int64, device=dev)
                torch.cumsum(lengths[t0:t1], 0, out=off[1:])
                out = torch.empty((int(off[-1].item()), 2), dtype=torch.int16, device=dev)
                wsb = nat.lib().ssrs_tracks_workspace_bytes(m)
                wss = torch.empty(wsb, dtype=torch.uint8, device=dev)
                st2 = nat.SsrsTrackStats()
                nat.check(nat.lib().ssrs_tracks_simulate(
                    C.byref(two_p), nat.ptr(upd), nat.ptr(pot), nat.ptr(two_table), nat.ptr(sub),
                    C.c_int64(m), C.c_uint64(int(seed) & 0xFFFFFFFFFFFFFFFF),
                    C.c_uint64(int(track_id_base) + t0), None, nat.ptr(sub_end), nat.ptr(sub_len),
                    nat.ptr(out), nat.ptr(off), nat.ptr(wss), C.c_size_t(wsb), C.byref(st2), stream_ptr()))
                if not torch.equal(sub_len, lengths[t0:t1]):
                    raise RuntimeError('trajectory replay: a track changed its length between the passes')
                return out.cpu().numpy()
    elif hist64:
        nat.check(nat.lib().ssrs_tracks_simulate_h64(
            C.byref(p), nat.ptr(upd), nat.ptr(pot), nat.ptr(table), nat.ptr(st),
            C.c_int64(n), C.c_uint64(int(seed) & 0xFFFFFFFFFFFFFFFF),
            C.c_uint64(int(track_id_base)), nat.ptr(hist_scratch), nat.ptr(hist), nat.ptr(ends),
            nat.ptr(lengths), nat.ptr(ws), C.c_size_t(ws_bytes), C.byref(stats), stream_ptr()))
    elif not want_tracks:
        run(hist, None, None)
    return TrackBatch(lengths, ends, hist, traj, offsets, replay=replay, stats=
                      dict(total_steps=int(stats.total_steps), launches=int(stats.launches),
                           kernel_ms=float(stats.kernel_ms), wall_ms=float(stats.wall_ms),
                           hist_ms=float(stats.hist_ms), window_launches=int(stats.window_launches),
                           tile_launches=int(stats.tile_launches),
                           block_window_launches=int(stats.block_window_launches),
                           wander_sorts=int(stats.wander_sorts), timed_launches=int(stats.timed_launches),
                           first_move_ms=float(stats.first_move_ms), recorded=bool(recorded),
                           traj_chunk_bytes=int(traj_budget_bytes),
                           block_window_ms=float(stats.block_window_ms),
                           block_window_timed=int(stats.block_window_timed),
                           block_window_steps=int(stats.block_window_steps),
                           roam_launches=int(stats.roam_launches),
                           roam_fine_settled=int(stats.reserved0),
                           roam_wave_pairs=int(stats.roam_wave_pairs),
                           roam_slow_wave_pairs=int(stats.roam_slow_wave_pairs),
                           roam_shuffles=int(stats.roam_shuffles),
                           roam_wide_launches=int(stats.roam_wide_launches)))


def generate_simulated_tracks(move_dirn, start_location, grid_shape, memory_parameter=1,
                              scaling_parameter=1., updraft_field=None, potential_field=None,
                              *, seed=0, track_id=0):
    """Reference signature (movmodel.py:264-272) for ONE track -> int16 (n, 2).
    The two keyword-only arguments name the track's random stream."""
    res = simulate_tracks(move_dirn, [list(start_location)], grid_shape, memory_parameter,
                          scaling_parameter, updraft_field, potential_field, seed=seed,
                          track_id_base=track_id, use_table=False, want_hist=False,
                          want_tracks=True)
    return res.tracks()[0]


def uniforms(seed, track, step):
    """u(seed, track, step) evaluated on the device (rocRAND Philox engine)."""
    tr = to_dev(np.asarray(track, dtype=np.uint64).view(np.int64), torch.int64).reshape(-1)
    sp = to_dev(np.asarray(step, dtype=np.uint64).view(np.int64), torch.int64).reshape(-1)
    if tr.numel() != sp.numel():
        raise ValueError('track and step lengths differ')
    out = torch.empty(tr.numel(), dtype=torch.float64, device=tr.device)
    nat.check(nat.lib().ssrs_uniform_selftest(
        C.c_uint64(int(seed) & 0xFFFFFFFFFFFFFFFF), nat.ptr(tr), nat.ptr(sp), nat.ptr(out),
        C.c_size_t(tr.numel()), stream_ptr()))
    return out.cpu().numpy()


# ---------------------------------------------------------------- re-exports
from .presence import (compute_presence_counts, compute_smooth_presence_counts)  # noqa: E402
from . import potential as _potential  # noqa: E402


class MovModel:
    """Fluid-flow movement model (movmodel.py:10-128).  Same constructor; the
    three-call sequence of the reference (boundary nodes, assemble, solve)
    is kept for compatibility, but nothing is assembled: `solve` runs the
    matrix-free GPU solver."""

    def __init__(self, move_dirn, grid_shape):
        self.move_dirn = move_dirn
        self.grid_shape = grid_shape

    def get_boundary_nodes(self):
        return _potential.get_boundary_nodes(self.move_dirn, self.grid_shape)

    def assemble_sparse_linear_system(self):
        """No-op placeholder: the operator is applied matrix-free on the GPU."""
        return None, None, None

    def solve_sparse_linear_system(self, conductivity, bnodes=None, benergy=None,
                                   row_inds=None, col_inds=None, facs=None, **kwargs):
        """Potential f32 (rows, cols).  bnodes/benergy are re-derived from
        move_dirn (they are a pure function of it in the reference too)."""
        return _potential.solve_potential(conductivity, self.move_dirn, **kwargs)
