#!/usr/bin/env python3
"""Golden-vector generator: runs the REFERENCE hot-path modules in the build
container and stores inputs + outputs as small .npz fixtures (SURVEY.md 8(c),
G1..G12).  Fixtures are data; the reference's source never enters the repo and
never travels to the GPU box.

How the reference is loaded: /root/reference/ssrs/{movmodel,layers}.py are
imported *by file path* (so `ssrs/__init__.py`, which needs pathos/rasterio/
network, is bypassed) after two in-process shims that do not touch the
reference files: `numpy.int = int` (alias removed in NumPy 1.24, used at
movmodel.py:134,137) and an empty `richdem` module (only referenced by the
unused compute_*_richdem_degrees, layers.py:146-167).

RNG injection: `np.random.choice` is replaced *in this process* by its own
algorithm with the uniform made explicit (cumsum -> /last -> searchsorted
'right'); `check_choice_equivalence()` proves the replacement reproduces the
legacy-MT trajectories bit for bit before any Philox-driven golden is made.

Usage:  python tests/golden/generate_golden.py [--skip-c1] [--skip-10m]
While generating, every vector is also compared with oracle/ssrs_oracle.py and
the script aborts on any mismatch.
"""
import argparse
import hashlib
import importlib.util
import os
import sys
import types
import multiprocessing as mp

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = '/root/reference/ssrs'


def load_reference():
    np.int = int                                           # shim (i)
    sys.modules.setdefault('richdem', types.ModuleType('richdem'))  # shim (ii)

    def _load(name, path):
        spec = importlib.util.spec_from_file_location(name, path)
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        return mod
    return (_load('ref_movmodel', os.path.join(REF, 'movmodel.py')),
            _load('ref_layers', os.path.join(REF, 'layers.py')))


mm, ly = load_reference()
from oracle import ssrs_oracle as orc          # noqa: E402
from oracle.philox import uniform53, TrackUniforms   # noqa: E402
from ssrs_amd.synthetic import synthetic_dem, wind_lattice  # noqa: E402

_ORIG_CHOICE = np.random.choice


def choice_with_uniform(next_uniform):
    """np.random.choice(a, p=p) restated with an explicit uniform source."""
    def _choice(a, p=None):
        cdf = np.cumsum(np.asarray(p, dtype=np.float64))
        cdf = cdf / cdf[-1]
        return int(np.searchsorted(cdf, next_uniform(), side='right'))
    return _choice


class StepCounter:
    def __init__(self, fn):
        self.fn, self.k = fn, 0

    def __call__(self):
        u = self.fn(self.k)
        self.k += 1
        return u


def ref_track(track_dirn, start, shape, mem, nu, updraft, potential, uniform):
    """One reference track with np.random.choice driven by uniform(step)."""
    np.random.choice = choice_with_uniform(StepCounter(uniform))
    try:
        return mm.generate_simulated_tracks(track_dirn, [int(start[0]), int(start[1])],
                                            shape, mem, nu, updraft, potential)
    finally:
        np.random.choice = _ORIG_CHOICE


def save(name, **arrays):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **arrays)
    print(f'  wrote {name}: {os.path.getsize(path) / 1024:.1f} KiB')


def small_case(rows=96, cols=128, res=100., wspeed=10., wdirn=270., thr=0.75,
               dem_seed=12345):
    z = synthetic_dem((rows, cols), res, seed=dem_seed)
    slope = ly.compute_slope_degrees(z, res)
    aspect = ly.compute_aspect_degrees(z, res)
    oro = ly.compute_orographic_updraft(wspeed * np.ones((rows, cols)),
                                        wdirn * np.ones((rows, cols)),
                                        slope, aspect)
    oro32 = oro.astype(np.float32)
    upd = ly.get_above_threshold_speed(oro32, thr)
    return z, slope, aspect, oro, oro32, upd


# ---------------------------------------------------------------------------
def g1_constants():
    print('G1 constants / masks / directional priors')
    deltas = np.array([np.asarray(d) for d in mm.neighbour_deltas])
    masks = np.array([mm.get_track_restrictions(int(d[0]), int(d[1]))
                      for d in deltas])
    thetas = np.array([0., 45., 90., 135., 180., 225., 270., 315., -45., 17.5])
    priors = np.array([mm.get_directional_probs(t * np.pi / 180.) for t in thetas])
    assert np.array_equal(deltas, np.array(orc.NEIGHBOUR_DELTAS))
    assert np.array_equal(mm.neighbour_delta_norms_inv, orc.NEIGHBOUR_DELTA_NORMS_INV)
    for d, m in zip(deltas, masks):
        assert np.array_equal(m, orc.get_track_restrictions(int(d[0]), int(d[1])))
    for t, p in zip(thetas, priors):
        assert np.array_equal(p, orc.get_directional_probs(t * np.pi / 180.))
    nudges = []
    for r, c, nr, nc in [(0, 0, 9, 11), (1, 1, 9, 11), (2, 5, 9, 11), (7, 9, 9, 11),
                         (8, 10, 9, 11), (6, 8, 9, 11), (4, 0, 9, 11), (1, 9, 9, 11)]:
        out = mm.move_away_from_boundary(r, c, nr, nc)
        assert out == orc.move_away_from_boundary(r, c, nr, nc)
        nudges.append([r, c, nr, nc, out[0], out[1]])
    save('g1_constants.npz', deltas=deltas,
         norms_inv=mm.neighbour_delta_norms_inv, masks=masks,
         thetas_deg=thetas, priors=priors, nudges=np.array(nudges))


def g2_raster():
    print('G2 raster 96x128')
    rows, cols, res = 96, 128, 100.
    z, slope, aspect, oro, oro32, upd = small_case(rows, cols, res)
    x, y, ws, wd = wind_lattice((cols * res / 1000., rows * res / 1000.), 2.0)
    # spatially varying wind raster (bilinear from the lattice, host numpy)
    from scipy.interpolate import RegularGridInterpolator
    xs = np.arange(cols) * res / 1000.
    ys = np.arange(rows) * res / 1000.
    pts = np.stack(np.meshgrid(ys, xs, indexing='ij'), -1)
    wsr = RegularGridInterpolator((y, x), ws, bounds_error=False, fill_value=None)(pts)
    wdr = RegularGridInterpolator((y, x), wd, bounds_error=False, fill_value=None)(pts)
    oro_var = ly.compute_orographic_updraft(wsr, wdr, slope, aspect)
    oro_min = ly.compute_orographic_updraft(10. * np.ones_like(z), 45. * np.ones_like(z),
                                            slope, aspect, 0.05)
    upd_var = ly.get_above_threshold_speed(oro_var.astype(np.float32), 0.75)
    assert upd.dtype == np.float64 and upd_var.dtype == np.float64
    # oracle check: slope/aspect/orographic are the same numpy expressions
    assert np.array_equal(slope, orc.compute_slope_degrees(z, res))
    assert np.array_equal(aspect, orc.compute_aspect_degrees(z, res))
    assert np.array_equal(oro, orc.compute_orographic_updraft(10., 270., slope, aspect))
    assert np.array_equal(oro_var, orc.compute_orographic_updraft(wsr, wdr, slope, aspect))
    np.testing.assert_allclose(orc.get_above_threshold_speed(oro32, 0.75), upd,
                               rtol=1e-14, atol=0)
    save('g2_raster.npz', dem=z, res=res, slope=slope, aspect=aspect,
         orograph=oro, orograph_f32=oro32, updraft=upd,
         wspeed_var=wsr, wdirn_var=wdr, orograph_var=oro_var,
         updraft_var=upd_var, orograph_min=oro_min)


def g3_threshold():
    print('G3 threshold sweep')
    v = np.concatenate([np.linspace(0., 3., 1201),
                        [0.01, np.nextafter(0.01, 1), 0.75, np.nextafter(0.75, 0),
                         np.nextafter(0.75, 1), 1e-9, 0.0100001]]).astype(np.float32)
    out = {}
    for thr in (0.75, 0.5, 1.2):
        ref = ly.get_above_threshold_speed(v, thr)
        assert ref.dtype == np.float64
        np.testing.assert_allclose(orc.get_above_threshold_speed(v, thr), ref,
                                   rtol=1e-14, atol=0)
        out[f'out_t{int(thr * 100)}'] = ref
    # the dtype trap: first element above threshold -> f32 output
    trap = ly.get_above_threshold_speed(np.array([1.0, 0.5], dtype=np.float32), 0.75)
    save('g3_threshold.npz', v=v, trap_dtype=np.array(str(trap.dtype)), **out)


def g4_starts():
    print('G4 starting indices')
    out = {}
    np.random.seed(30)
    r, c = mm.get_starting_indices(1000, (5, 55, 1, 2), 'random', (60., 50.), 100.)
    np.random.seed(30)
    r2, c2 = orc.get_starting_indices(1000, (5, 55, 1, 2), 'random', (60., 50.), 100.)
    assert np.array_equal(r, r2) and np.array_equal(c, c2)
    out['rand_rows'], out['rand_cols'] = r, c
    base = None
    for n in (5, 1000, 6000, 5151, 12000):
        r, c = mm.get_starting_indices(n, (5, 55, 1, 2), 'structured', (60., 50.), 100.)
        r2, c2 = orc.get_starting_indices(n, (5, 55, 1, 2), 'structured', (60., 50.), 100.)
        assert np.array_equal(r, r2) and np.array_equal(c, c2), n
        out[f'struct{n}_rows'], out[f'struct{n}_cols'] = r, c
    # a 10 m band (C2 shape) and an edge-hugging region
    np.random.seed(31)
    r, c = mm.get_starting_indices(64, (0, 60, 0, 0.5), 'random', (60., 50.), 10.)
    np.random.seed(31)
    r2, c2 = orc.get_starting_indices(64, (0, 60, 0, 0.5), 'random', (60., 50.), 10.)
    assert np.array_equal(r, r2) and np.array_equal(c, c2)
    out['edge_rows'], out['edge_cols'] = r, c
    save('g4_starts.npz', **out)


def g5_potential():
    print('G5 potential 48x64 (reference spsolve)')
    rows, cols = 48, 64
    _, _, _, _, _, upd = small_case(rows, cols, 100.)
    out = {'updraft': upd}
    for dirn in (0., 180., -45., 90., 30.):
        model = mm.MovModel(dirn, (rows, cols))
        bn, be = model.get_boundary_nodes()
        ri, ci, fa = model.assemble_sparse_linear_system()
        pot = model.solve_sparse_linear_system(upd, bn, be, ri, ci, fa)
        obn, obe = orc.get_boundary_nodes(dirn, (rows, cols))
        assert np.array_equal(bn, obn) and np.array_equal(be, obe)
        ori, oci, ofa = orc.neighbour_lists((rows, cols))
        assert np.array_equal(ri, ori) and np.array_equal(ci, oci)
        assert np.array_equal(fa, ofa)
        opot = orc.solve_potential(upd, dirn)
        np.testing.assert_allclose(opot, pot, rtol=2e-6, atol=1e-4)
        tag = f'd{int(dirn % 360)}'
        out[f'pot_{tag}'] = pot
        out[f'bnodes_{tag}'] = bn
        out[f'benergy_{tag}'] = be
    save('g5_potential.npz', **out)


def g6_move_probs():
    print('G6 move probabilities')
    rng = np.random.default_rng(606)
    n = 500
    w = rng.uniform(-0.5, 1.5, size=(n, 9)) * rng.choice([1e-9, 1., 1e3], size=(n, 1))
    w[rng.random((n, 9)) < 0.25] = 0.
    w[:50] = 0.                                   # all-zero rows
    w[50:60, 3] = np.nan                          # NaN rows
    w[60:80] = -np.abs(w[60:80])                  # all negative -> clipped to 0
    masks = np.array([mm.get_track_restrictions(*d) for d in
                      [(0, 0), (-1, -1), (-1, 0), (-1, 1), (0, -1), (0, 1),
                       (1, -1), (1, 0), (1, 1)]] + [np.zeros(9, dtype=int)])
    dirns = np.array([0., 90., 200.])
    nus = np.array([1.0, 0.5, 2.0])
    res = np.empty((len(dirns), len(nus), len(masks), n, 9))
    import contextlib
    import io
    for a, dirn in enumerate(dirns):
        for b, nu in enumerate(nus):
            for m, mask in enumerate(masks):
                for i in range(n):
                    with contextlib.redirect_stdout(io.StringIO()):
                        res[a, b, m, i] = mm.generate_move_probabilities(
                            w[i], dirn, nu, mask.astype(bool))
                    mine = orc.generate_move_probabilities(w[i], dirn, nu, mask)
                    if nu == 1.0:
                        assert np.array_equal(res[a, b, m, i], mine), (a, b, m, i)
                    else:
                        np.testing.assert_allclose(mine, res[a, b, m, i], rtol=1e-14)
    # keep the fixture small: nu=1 fully, others on the first 200 rows
    save('g6_move_probs.npz', w=w, masks=masks, dirns=dirns, nus=nus,
         probs_nu1=res[:, 0], probs_other=res[:, 1:, :, :120])


def check_choice_equivalence(upd, pot, shape):
    """Legacy-MT path: original np.random.choice vs the explicit-uniform
    replacement fed by np.random.random_sample -> identical trajectories."""
    starts = [(5, 20), (8, 64), (3, 100)]
    np.random.seed(77)
    a = [mm.generate_simulated_tracks(0., list(s), shape, 1, 1., upd, pot) for s in starts]
    np.random.seed(77)
    np.random.choice = choice_with_uniform(np.random.random_sample)
    try:
        b = [mm.generate_simulated_tracks(0., list(s), shape, 1, 1., upd, pot) for s in starts]
    finally:
        np.random.choice = _ORIG_CHOICE
    np.random.seed(77)
    c = [orc.generate_simulated_tracks(0., s, shape, 1, 1., upd, pot) for s in starts]
    for x, y, z in zip(a, b, c):
        assert np.array_equal(x, y), 'choice replacement is not equivalent'
        assert np.array_equal(x, z), 'oracle (legacy MT) differs from reference'
    return starts, a


def g7_tracks():
    print('G7 trajectories 96x128, 64 tracks, Philox + legacy MT')
    rows, cols = 96, 128
    shape = (rows, cols)
    _, _, _, _, oro32, upd = small_case(rows, cols, 100.)
    model = mm.MovModel(0., shape)
    bn, be = model.get_boundary_nodes()
    ri, ci, fa = model.assemble_sparse_linear_system()
    pot = model.solve_sparse_linear_system(upd, bn, be, ri, ci, fa)
    mt_starts, mt_tracks = check_choice_equivalence(upd, pot, shape)
    rng = np.random.default_rng(707)
    ntr = 64
    srows = rng.integers(0, 12, ntr)          # includes border rows -> nudges
    scols = rng.integers(0, cols, ntr)
    seed = 30
    out = dict(updraft=upd, potential=pot, orograph_f32=oro32,
               start_rows=srows, start_cols=scols, seed=seed,
               mt_starts=np.array(mt_starts), mt_seed=77,
               mt_lengths=np.array([len(t) for t in mt_tracks]),
               mt_tracks=np.concatenate(mt_tracks))
    cases = [('ff_m1', 0., 1, 1.0, upd, pot), ('ff_m3', 0., 3, 1.0, upd, pot),
             ('ff_d135_m2', 135., 2, 1.0, upd, None),
             ('drw_m1', 0., 1, 1.0, None, None), ('drw_d250_m3', 250., 3, 1.0, None, None),
             ('ff_m1_nu05', 0., 1, 0.5, upd, pot)]
    for tag, dirn, mem, nu, u_, p_ in cases:
        tracks = []
        for t in range(ntr):
            uni = TrackUniforms(seed, t)
            tr = ref_track(dirn, (srows[t], scols[t]), shape, mem, nu, u_, p_, uni)
            mine = orc.generate_simulated_tracks(dirn, (srows[t], scols[t]), shape,
                                                 mem, nu, u_, p_, uniform=uni)
            if nu == 1.0:
                assert np.array_equal(tr, mine), (tag, t)
            tracks.append(tr)
        lens = np.array([len(t) for t in tracks])
        print(f'    {tag}: steps mean {lens.mean():.0f} max {lens.max()}')
        out[f'{tag}_lengths'] = lens
        out[f'{tag}_tracks'] = np.concatenate(tracks)
        out[f'{tag}_params'] = np.array([dirn, mem, nu, u_ is not None, p_ is not None])
    save('g7_tracks.npz', **out)


def g9_presence():
    print('G9 presence counts / smoothing')
    shape = (40, 50)
    rng = np.random.default_rng(909)
    tracks = []
    for _ in range(5):
        n = int(rng.integers(30, 200))
        pts = np.stack([rng.integers(0, shape[0], n), rng.integers(0, shape[1], n)], 1)
        tracks.append(pts.astype(np.int16))
    tracks.append(np.array([[0, 0], [0, 49], [39, 0], [39, 49], [0, 0]], dtype=np.int16))
    counts = mm.compute_presence_counts(tracks, shape)
    assert np.array_equal(counts, orc.compute_presence_counts(tracks, shape))
    out = dict(lengths=np.array([len(t) for t in tracks]),
               tracks=np.concatenate(tracks), counts=counts)
    for rad in (2, 5, 13):
        sm = mm.compute_smooth_presence_counts(tracks, shape, rad)
        assert np.array_equal(sm, orc.compute_smooth_presence_counts(tracks, shape, rad))
        out[f'smooth_r{rad}'] = sm
    save('g9_presence.npz', **out)


# --------------------------------------------------------------------------- C1
_C1 = {}


def _c1_worker(t):
    uni = TrackUniforms(_C1['seed'], t)
    tr = ref_track(0., _C1['starts'][t], _C1['shape'], 1, 1., _C1['upd'], _C1['pot'], uni)
    return tr


def g8_c1(ntracks=1000, procs=8):
    print('G8 config C1: 500x600 @100 m, 1000 tracks, seed 30 (reference, slow)')
    rows, cols, res = 500, 600, 100.
    shape = (rows, cols)
    z, slope, aspect, oro, oro32, upd = small_case(rows, cols, res)
    model = mm.MovModel(0., shape)
    bn, be = model.get_boundary_nodes()
    ri, ci, fa = model.assemble_sparse_linear_system()
    print('   reference potential solve ...', flush=True)
    pot = model.solve_sparse_linear_system(upd, bn, be, ri, ci, fa)
    opot = orc.solve_potential(upd, 0.)
    np.testing.assert_allclose(opot, pot, rtol=2e-6, atol=1e-4)
    np.random.seed(30)
    srows, scols = mm.get_starting_indices(ntracks, (5, 55, 1, 2), 'random',
                                           (60., 50.), res)
    _C1.update(seed=30, starts=list(zip(srows, scols)), shape=shape, upd=upd, pot=pot)
    print('   reference tracks ...', flush=True)
    with mp.get_context('fork').Pool(procs) as pool:
        tracks = pool.map(_c1_worker, range(ntracks), chunksize=8)
    lens = np.array([len(t) for t in tracks])
    ends = np.array([t[-1] for t in tracks])
    sha = hashlib.sha256()
    for t in tracks:
        sha.update(np.ascontiguousarray(t, dtype='<i2').tobytes())
    hist = np.zeros(shape, dtype=np.int32)
    for t in tracks:
        np.add.at(hist, (t[:, 0].astype(int), t[:, 1].astype(int)), 1)
    ref_hist16 = mm.compute_presence_counts(tracks[:20], shape)
    assert np.array_equal(ref_hist16, orc.compute_presence_counts(tracks[:20], shape))
    krad = orc.presence_kernel_radius(1000., res, shape)
    print(f'   steps mean {lens.mean():.0f} max {lens.max()}  krad {krad}; smoothing ...',
          flush=True)
    smooth = mm.compute_smooth_presence_counts(tracks, shape, krad)
    smooth = smooth / np.amax(smooth)
    save('g8_c1.npz', orograph_f32=oro32, potential=pot, start_rows=srows,
         start_cols=scols, seed=30, lengths=lens, ends=ends,
         traj_sha256=np.array(sha.hexdigest()), hist=hist, krad=krad,
         presence_strided=smooth[::8, ::8].astype(np.float32),
         presence_max_raw=np.float32(np.amax(mm.compute_smooth_presence_counts(
             tracks, shape, krad))),
         first_tracks=np.concatenate(tracks[:8]), first_lengths=lens[:8])


# ------------------------------------------------------------ G10: the 10 m regime
_G10 = {}


def _g10_worker(t):
    uni = TrackUniforms(_G10['seed'], t)
    return ref_track(0., _G10['starts'][t], _G10['shape'], 1, 1., _G10['upd'], _G10['pot'], uni)


def shuffle_f32(a):
    """f32 raster -> byte planes (all byte 0, all byte 1, ...): the sign/exponent planes
    deflate well, so the .npz stays small.  tests/conftest.py:unshuffle_f32 undoes it."""
    a = np.ascontiguousarray(a, dtype='<f4')
    return np.ascontiguousarray(a.view(np.uint8).reshape(-1, 4).T)


G10_WINDOW = (1000, 1200, 1000, 1200)      # row0, col0, rows, cols of the C2 DEM


def g10_10m(ntracks=256, procs=8):
    """The regime of BASELINE configs[1..4]: a 1000 x 1200 window of the 5000 x 6000 @10 m
    DEM of SURVEY 8(d) (rows 1000.., cols 1200..: a lee slope with 1/4-1/3 live cells in the
    south, a windward slope with > 80 % live cells in the north; at 10 m the per-cell noise
    term dominates the slopes, so both phases are speckled).  Reference potential (assemble +
    SuperLU, movmodel.py:59-128) and reference tracks (:264-318) under the Philox injection."""
    r0, c0, rows, cols = G10_WINDOW
    print(f'G10 10 m regime: {rows}x{cols} window at ({r0}, {c0}) of the C2 DEM, {ntracks} tracks, '
          'seed 30 (reference, slow)')
    res = 10.
    shape = (rows, cols)
    z = synthetic_dem((5000, 6000), res)[r0:r0 + rows, c0:c0 + cols].copy()
    slope = ly.compute_slope_degrees(z, res)
    aspect = ly.compute_aspect_degrees(z, res)
    oro = ly.compute_orographic_updraft(10. * np.ones(shape), 270. * np.ones(shape), slope, aspect)
    oro32 = oro.astype(np.float32)
    upd = ly.get_above_threshold_speed(oro32, 0.75)
    assert upd.dtype == np.float64
    model = mm.MovModel(0., shape)
    bn, be = model.get_boundary_nodes()
    ri, ci, fa = model.assemble_sparse_linear_system()
    print('   reference potential solve ...', flush=True)
    pot = model.solve_sparse_linear_system(upd, bn, be, ri, ci, fa)
    del ri, ci, fa
    opot = orc.solve_potential(upd, 0.)
    np.testing.assert_allclose(opot, pot, rtol=2e-6, atol=1e-4)
    # local extrema of the f32 field away from the Dirichlet rows: the exact solution is
    # discrete-harmonic (no interior extrema); plateaus over live clusters are allowed
    p = pot.astype(np.float64)
    inner = p[1:-1, 1:-1]
    nb = np.stack([p[1 + dr:rows - 1 + dr, 1 + dc:cols - 1 + dc]
                   for dr in (-1, 0, 1) for dc in (-1, 0, 1) if (dr, dc) != (0, 0)])
    strict_min = int((inner[None] < nb).all(0)[1:-1].sum())
    strict_max = int((inner[None] > nb).all(0)[1:-1].sum())
    width_km = (cols * res / 1000., rows * res / 1000.)
    np.random.seed(30)
    srows, scols = mm.get_starting_indices(ntracks, (1., width_km[0] - 1., 0.1, 0.3), 'random',
                                           width_km, res)
    _G10.update(seed=30, starts=list(zip(srows, scols)), shape=shape, upd=upd, pot=pot)
    print('   reference tracks ...', flush=True)
    with mp.get_context('fork').Pool(procs) as pool:
        tracks = pool.map(_g10_worker, range(ntracks), chunksize=4)
    lens = np.array([len(t) for t in tracks])
    ends = np.array([t[-1] for t in tracks])
    mine = c_oracle_tracks(starts=np.stack([srows, scols], 1), shape=shape, upd=upd, pot=pot)
    assert np.array_equal(mine['lengths'], lens), 'C oracle lengths differ from the reference'
    for a, b in zip(mine['tracks'], tracks):
        assert np.array_equal(a, b), 'C oracle trajectory differs from the reference'
    sha = hashlib.sha256()
    for t in tracks:
        sha.update(np.ascontiguousarray(t, dtype='<i2').tobytes())
    hist = np.zeros(shape, dtype=np.int32)
    for t in tracks:
        np.add.at(hist, (t[:, 0].astype(int), t[:, 1].astype(int)), 1)
    hr, hc = np.nonzero(hist)
    steps = lens - 1
    print(f'   steps/track mean {steps.mean():.0f} median {np.median(steps):.0f} max {steps.max()} '
          f'(max_moves {rows // 2 * (cols // 2)}); dead cells {np.mean(upd == 0):.3f}; '
          f'strict interior minima {strict_min}, maxima {strict_max}', flush=True)
    save('g10_10m.npz', shape=np.array(shape), res=res, window=np.array(G10_WINDOW),
         orograph_f32_planes=shuffle_f32(oro32), potential_planes=shuffle_f32(pot),
         start_rows=srows, start_cols=scols, seed=30, lengths=lens, ends=ends,
         traj_sha256=np.array(sha.hexdigest()),
         hist_rows=hr.astype(np.int16), hist_cols=hc.astype(np.int16),
         hist_vals=hist[hr, hc].astype(np.int32),
         first_tracks=np.concatenate(tracks[:4]), first_lengths=lens[:4],
         dead_fraction=np.mean(upd == 0), strict_minima=strict_min, strict_maxima=strict_max,
         max_moves=rows // 2 * (cols // 2))


def c_oracle_tracks(starts, shape, upd, pot):
    from oracle import c_oracle
    return c_oracle.simulate_tracks(0., starts, shape, 1, 1., upd, pot, seed=30,
                                    want_traj=True, want_hist=False)


# ------------------------------------------- G11: tracks that wander until max_moves
def g11_wander(ntracks=64, procs=8):
    """The whole 60 x 50 km domain of SURVEY 8(d) at 50 m (1000 x 1200 cells, the largest
    size SuperLU factorises here in about a minute): with the reference's own potential
    about half of the reference's tracks never leave the raster -- they reach a basin of
    the potential field whose outlet the f32 field does not resolve, circle there and stop
    at max_moves = 300 000 (movmodel.py:277,285).  Pins that regime (the one the 10 m
    configs live in: tools/attic/probe_traps.py) to the reference: potential, lengths, end cells
    and the sha256 over every point of the 64 tracks."""
    rows, cols, res = 1000, 1200, 50.
    print(f'G11 wandering regime: {rows}x{cols} @50 m, {ntracks} tracks, seed 30 (reference, slow)')
    shape = (rows, cols)
    z, slope, aspect, oro, oro32, upd = small_case(rows, cols, res)
    assert upd.dtype == np.float64
    model = mm.MovModel(0., shape)
    bn, be = model.get_boundary_nodes()
    ri, ci, fa = model.assemble_sparse_linear_system()
    print('   reference potential solve ...', flush=True)
    pot = model.solve_sparse_linear_system(upd, bn, be, ri, ci, fa)
    del ri, ci, fa
    opot = orc.solve_potential(upd, 0.)
    np.testing.assert_allclose(opot, pot, rtol=2e-6, atol=1e-4)
    np.random.seed(30)
    srows, scols = mm.get_starting_indices(ntracks, (5, 55, 1, 2), 'random', (60., 50.), res)
    starts = np.stack([srows, scols], 1)
    # the C port first (seconds): how many tracks run into max_moves
    mine = c_oracle_tracks(starts=starts, shape=shape, upd=upd, pot=pot)
    max_moves = rows // 2 * (cols // 2)
    print(f'   C oracle: {np.mean(mine["lengths"] > max_moves):.2f} of the tracks stop at max_moves; '
          'reference tracks ...', flush=True)
    _G10.update(seed=30, starts=list(zip(srows, scols)), shape=shape, upd=upd, pot=pot)
    with mp.get_context('fork').Pool(procs) as pool:
        tracks = pool.map(_g10_worker, range(ntracks), chunksize=1)
    lens = np.array([len(t) for t in tracks])
    ends = np.array([t[-1] for t in tracks])
    assert np.array_equal(mine['lengths'], lens), 'C oracle lengths differ from the reference'
    for a, b in zip(mine['tracks'], tracks):
        assert np.array_equal(a, b), 'C oracle trajectory differs from the reference'
    sha = hashlib.sha256()
    for t in tracks:
        sha.update(np.ascontiguousarray(t, dtype='<i2').tobytes())
    hist = np.zeros(shape, dtype=np.int32)
    for t in tracks:
        np.add.at(hist, (t[:, 0].astype(int), t[:, 1].astype(int)), 1)
    hr, hc = np.nonzero(hist)
    steps = lens - 1
    print(f'   steps/track median {np.median(steps):.0f} max {steps.max()} (max_moves {max_moves}); '
          f'at max_moves {np.mean(steps >= max_moves):.3f}; busiest cell {hist.max()} visits', flush=True)
    save('g11_wander.npz', shape=np.array(shape), res=res,
         orograph_f32_planes=shuffle_f32(oro32), potential_planes=shuffle_f32(pot),
         start_rows=srows, start_cols=scols, seed=30, lengths=lens, ends=ends,
         traj_sha256=np.array(sha.hexdigest()),
         hist_rows=hr.astype(np.int16), hist_cols=hc.astype(np.int16),
         hist_vals=hist[hr, hc].astype(np.int32), max_moves=max_moves,
         dead_fraction=np.mean(upd == 0))


# ----------------------------- G12: exact solutions of the reference's potential systems
def reference_system(conductivity, move_dirn, numpy2=True):
    """The linear system of MovModel.solve_sparse_linear_system (movmodel.py:98-120) as
    matrices, entry for entry in the reference's arithmetic: (A csc, b, inner nodes, boundary
    nodes, boundary energy).  `numpy2`: `harmonic_mean(...) / fac` divides the python float
    1e-08 of a dead pair by an np.float32 -- an f32 value under NumPy >= 2 (NEP 50, the NumPy
    of this container, so the goldens carry it), an f64 value under the NumPy < 1.24 the
    reference was written for (it uses np.int).  g12 checks that SuperLU on the numpy2 form
    returns the golden field bit for bit."""
    import scipy.sparse as ss
    nrow, ncol = conductivity.shape
    n = nrow * ncol
    model = mm.MovModel(move_dirn, (nrow, ncol))
    bnodes, benergy = model.get_boundary_nodes()
    ri, ci, facs = model.assemble_sparse_linear_system()
    r = ri.astype(np.int64)
    c = ci.astype(np.int64)
    ca = conductivity[r % nrow, r // nrow]
    cb = conductivity[c % nrow, c // nrow]
    live = (ca != 0) & (cb != 0)
    with np.errstate(divide='ignore'):
        dead = (np.float32(1e-08) / facs).astype(np.float64) if numpy2 else 1e-08 / facs.astype(np.float64)
        vals = np.where(live, (2. / (1. / ca + 1. / cb)) / facs, dead)
    g_csr = ss.coo_matrix((vals, (r, c)), shape=(n, n)).tocsr()
    g_csr.data = g_csr.data / np.repeat(np.add.reduceat(g_csr.data, g_csr.indptr[:-1]),
                                        np.diff(g_csr.indptr))
    inodes = np.setdiff1d(np.arange(0, n), bnodes, assume_unique=True)
    g_inner_csc = g_csr[inodes, :].tocoo().tocsc()
    b_vec = g_inner_csc[:, bnodes].dot(benergy)
    a_matrix = ss.eye(np.size(inodes)).tocsc() - g_inner_csc[:, inodes]
    return a_matrix, b_vec, inodes, bnodes, benergy


def exact_potential(conductivity, move_dirn, ref_field=None, sweeps=4, numpy2=True):
    """The exact solution of that system: SuperLU + iterative refinement with the residual
    accumulated in x87 extended precision (converges: condition ~1e10 x 2^-64 << 1)."""
    import scipy.sparse.linalg as ssl
    nrow, ncol = conductivity.shape
    a_mat, b_vec, inodes, bnodes, benergy = reference_system(conductivity, move_dirn, numpy2)
    lu = ssl.splu(a_mat)
    x0 = lu.solve(b_vec)

    def field(xi):
        e = np.empty(nrow * ncol)
        e[inodes] = xi
        e[bnodes] = benergy
        return e.reshape(ncol, nrow).T

    if ref_field is not None:
        same = np.mean(field(x0).astype(np.float32) == ref_field)
        print(f'      SuperLU on the restated system reproduces {same:.6f} of the reference field bit for bit')
        assert same == 1.0
    a_csr = a_mat.tocsr()
    a_csr.sort_indices()
    dl = a_csr.data.astype(np.longdouble)
    bl = b_vec.astype(np.longdouble)
    x = x0.astype(np.longdouble)
    for it in range(sweeps):
        res = bl - np.add.reduceat(dl * x[a_csr.indices], a_csr.indptr[:-1])
        dx = lu.solve(res.astype(np.float64))
        x = x + dx.astype(np.longdouble)
        print(f'      refinement {it}: max |dx| {np.abs(dx).max():.3e}', flush=True)
    assert np.abs(dx).max() < 1e-6
    return field(x.astype(np.float64))


def g12_exact():
    """How good is the reference's own field?  For C1 (G8), the 10 m window (G10) and the
    50 m domain (G11): the exact solution of the reference's system, f32-rounded, on a strided
    sample -- the yardstick for ssrs_potential_solve, since SuperLU's field is itself several
    f32 ulp off at these condition numbers."""
    print('G12 exact potentials (SuperLU + extended-precision refinement)')
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    from conftest import unshuffle_f32
    out = {}
    for tag, fname, stride in (('c1', 'g8_c1.npz', 2), ('g10', 'g10_10m.npz', 3), ('g11', 'g11_wander.npz', 3)):
        g = np.load(os.path.join(HERE, fname))
        if 'orograph_f32' in g:
            oro32, ref = g['orograph_f32'], g['potential']
        else:
            shape = tuple(int(v) for v in g['shape'])
            oro32 = unshuffle_f32(g['orograph_f32_planes'], shape)
            ref = unshuffle_f32(g['potential_planes'], shape)
        upd = ly.get_above_threshold_speed(oro32, 0.75)
        print(f'   {tag}: {upd.shape}', flush=True)
        exact = exact_potential(upd, 0., ref_field=ref)           # the system the goldens were made with
        d = np.abs(ref.astype(np.float64) - exact)
        e32 = exact.astype(np.float32)
        ulp = np.abs(e32.view(np.int32).astype(np.int64) - ref.view(np.int32))
        print(f'      reference field vs exact: max {d.max():.3e}, mean {d.mean():.3e}; '
              f'{np.mean(ulp == 0):.3f} of its cells are the correctly rounded value, max {ulp.max()} ulp')
        # the same system with f64 dead entries (NumPy < 1.24 semantics; what ssrs_potential_solve
        # implements): its exact solution is the yardstick stored here
        exact1 = exact_potential(upd, 0., numpy2=False)
        print(f'      f32 vs f64 dead-pair entries move the exact solution by up to '
              f'{np.abs(exact1 - exact).max():.3e}')
        out[f'{tag}_numpy2_shift'] = np.abs(exact1 - exact).max()
        e32 = exact1.astype(np.float32)
        out[f'{tag}_stride'] = stride
        out[f'{tag}_exact_f32'] = e32[::stride, ::stride].copy()
        out[f'{tag}_ref_max_err'] = d.max()
        out[f'{tag}_ref_mean_err'] = d.mean()
        out[f'{tag}_ref_max_ulp'] = ulp.max()
        out[f'{tag}_ref_exact_share'] = np.mean(ulp == 0)
    save('g12_exact_potential.npz', **out)


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('--skip-c1', action='store_true')
    ap.add_argument('--skip-10m', action='store_true')
    ap.add_argument('--only', default='')
    args = ap.parse_args()
    todo = [g1_constants, g2_raster, g3_threshold, g4_starts, g5_potential,
            g6_move_probs, g7_tracks, g9_presence]
    if not args.skip_c1:
        todo.append(g8_c1)
    if not args.skip_10m:
        todo.append(g10_10m)
        todo.append(g11_wander)
        todo.append(g12_exact)
    for fn in todo:
        if args.only and args.only not in fn.__name__:
            continue
        fn()
    print('done')
