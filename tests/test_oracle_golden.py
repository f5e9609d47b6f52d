"""Pins the oracle (numpy restatement + C port) against every golden vector
captured from the reference (tests/golden/*.npz, generate_golden.py)."""
import hashlib

import numpy as np
import pytest

from oracle import ssrs_oracle as orc
from oracle import c_oracle
from oracle.philox import TrackUniforms


def split(flat, lengths):
    off = np.concatenate([[0], np.cumsum(lengths)])
    return [flat[off[i]:off[i + 1]] for i in range(len(lengths))]


def test_g1_constants(golden):
    g = golden('g1_constants.npz')
    assert np.array_equal(np.array(orc.NEIGHBOUR_DELTAS), g['deltas'])
    assert np.array_equal(orc.NEIGHBOUR_DELTA_NORMS_INV, g['norms_inv'])
    assert orc.NEIGHBOUR_DELTA_NORMS_INV.dtype == np.float32
    for d, m in zip(g['deltas'], g['masks']):
        assert np.array_equal(orc.get_track_restrictions(int(d[0]), int(d[1])), m)
    for t, p in zip(g['thetas_deg'], g['priors']):
        assert np.array_equal(orc.get_directional_probs(t * np.pi / 180.), p)
    for r, c, nr, nc, er, ec in g['nudges']:
        assert orc.move_away_from_boundary(r, c, nr, nc) == (er, ec)


def test_g2_raster(golden):
    g = golden('g2_raster.npz')
    res = float(g['res'])
    assert np.array_equal(orc.compute_slope_degrees(g['dem'], res), g['slope'])
    assert np.array_equal(orc.compute_aspect_degrees(g['dem'], res), g['aspect'])
    assert np.array_equal(orc.compute_orographic_updraft(10., 270., g['slope'], g['aspect']),
                          g['orograph'])
    assert np.array_equal(orc.compute_orographic_updraft(g['wspeed_var'], g['wdirn_var'],
                                                         g['slope'], g['aspect']),
                          g['orograph_var'])
    assert np.array_equal(orc.compute_orographic_updraft(10., 45., g['slope'], g['aspect'], 0.05),
                          g['orograph_min'])
    np.testing.assert_allclose(orc.get_above_threshold_speed(g['orograph_f32'], 0.75),
                               g['updraft'], rtol=1e-14, atol=0)
    # C port (glibc libm instead of numpy's loops): a few ulp
    s, a = c_oracle.slope_aspect(g['dem'], res)
    np.testing.assert_allclose(s, g['slope'], rtol=1e-13, atol=1e-14)
    np.testing.assert_allclose(a, g['aspect'], rtol=1e-13, atol=1e-12)
    o64, o32 = c_oracle.orographic(g['slope'], g['aspect'], 10., 270.)
    np.testing.assert_allclose(o64, g['orograph'], rtol=1e-13, atol=1e-15)
    ulp = np.abs(o32.view(np.int32).astype(np.int64) - g['orograph_f32'].view(np.int32))
    assert ulp.max() <= 1
    np.testing.assert_allclose(c_oracle.threshold(g['orograph_f32'], 0.75), g['updraft'],
                               rtol=1e-12, atol=1e-15)


def test_g3_threshold(golden):
    g = golden('g3_threshold.npz')
    for thr in (0.75, 0.5, 1.2):
        want = g[f'out_t{int(thr * 100)}']
        np.testing.assert_allclose(orc.get_above_threshold_speed(g['v'], thr), want,
                                   rtol=1e-14, atol=0)
        np.testing.assert_allclose(c_oracle.threshold(g['v'], thr), want, rtol=1e-12, atol=1e-15)
    assert str(g['trap_dtype']) == 'float32'      # the documented reference quirk


def test_g4_starts(golden):
    g = golden('g4_starts.npz')
    np.random.seed(30)
    r, c = orc.get_starting_indices(1000, (5, 55, 1, 2), 'random', (60., 50.), 100.)
    assert np.array_equal(r, g['rand_rows']) and np.array_equal(c, g['rand_cols'])
    for n in (5, 1000, 6000, 5151, 12000):
        r, c = orc.get_starting_indices(n, (5, 55, 1, 2), 'structured', (60., 50.), 100.)
        assert np.array_equal(r, g[f'struct{n}_rows']) and np.array_equal(c, g[f'struct{n}_cols'])
    np.random.seed(31)
    r, c = orc.get_starting_indices(64, (0, 60, 0, 0.5), 'random', (60., 50.), 10.)
    assert np.array_equal(r, g['edge_rows']) and np.array_equal(c, g['edge_cols'])
    with pytest.raises(ValueError):
        orc.get_starting_indices(5, (5, 65, 1, 2), 'random', (60., 50.), 100.)
    with pytest.raises(ValueError):
        orc.get_starting_indices(5, (5, 55, 1, 2), 'bogus', (60., 50.), 100.)


def test_g5_potential(golden):
    g = golden('g5_potential.npz')
    for dirn in (0., 180., -45., 90., 30.):
        tag = f'd{int(dirn % 360)}'
        bn, be = orc.get_boundary_nodes(dirn, (48, 64))
        assert np.array_equal(bn, g[f'bnodes_{tag}']) and np.array_equal(be, g[f'benergy_{tag}'])
        pot = orc.solve_potential(g['updraft'], dirn)
        assert pot.dtype == np.float32
        np.testing.assert_allclose(pot, g[f'pot_{tag}'], rtol=2e-6, atol=1e-4)


def test_g6_move_probabilities(golden):
    g = golden('g6_move_probs.npz')
    w, masks = g['w'], g['masks']
    for a, dirn in enumerate(g['dirns']):
        for m, mask in enumerate(masks):
            for i in range(0, w.shape[0], 3):
                got = orc.generate_move_probabilities(w[i], dirn, 1.0, mask)
                assert np.array_equal(got, g['probs_nu1'][a, m, i]), (a, m, i)
    for b, nu in enumerate(g['nus'][1:]):
        for m in (0, 3, 9):
            for i in range(0, 120, 7):
                got = orc.generate_move_probabilities(w[i], g['dirns'][1], nu, masks[m])
                np.testing.assert_allclose(got, g['probs_other'][1, b, m, i], rtol=1e-14)


G7 = ['ff_m1', 'ff_m3', 'ff_d135_m2', 'drw_m1', 'drw_d250_m3', 'ff_m1_nu05']


@pytest.mark.parametrize('tag', G7)
def test_g7_tracks_c_oracle(golden, tag):
    g = golden('g7_tracks.npz')
    dirn, mem, nu, hu, hp = g[tag + '_params']
    starts = np.stack([g['start_rows'], g['start_cols']], 1)
    res = c_oracle.simulate_tracks(dirn, starts, (96, 128), int(mem), nu,
                                   g['updraft'] if hu else None, g['potential'] if hp else None,
                                   seed=int(g['seed']))
    assert np.array_equal(res['lengths'], g[tag + '_lengths'])
    assert np.array_equal(np.concatenate(res['tracks']), g[tag + '_tracks'])
    want = split(g[tag + '_tracks'], g[tag + '_lengths'])
    assert np.array_equal(res['ends'], np.array([t[-1] for t in want]))
    assert int(res['hist'].sum()) == int(g[tag + '_lengths'].sum())


@pytest.mark.parametrize('tag', ['ff_m1', 'drw_d250_m3', 'ff_m3'])
def test_g7_tracks_python_oracle(golden, tag):
    g = golden('g7_tracks.npz')
    dirn, mem, nu, hu, hp = g[tag + '_params']
    want = split(g[tag + '_tracks'], g[tag + '_lengths'])
    for t in range(0, 64, 9):
        got = orc.generate_simulated_tracks(
            dirn, (g['start_rows'][t], g['start_cols'][t]), (96, 128), int(mem), nu,
            g['updraft'] if hu else None, g['potential'] if hp else None,
            uniform=TrackUniforms(int(g['seed']), t))
        assert got.dtype == np.int16 and np.array_equal(got, want[t])


def test_g7_legacy_mt_stream(golden):
    """Reference's own RNG source: global MT19937, one double per step."""
    g = golden('g7_tracks.npz')
    want = split(g['mt_tracks'], g['mt_lengths'])
    np.random.seed(int(g['mt_seed']))
    for s, w in zip(g['mt_starts'], want):
        got = orc.generate_simulated_tracks(0., tuple(s), (96, 128), 1, 1., g['updraft'],
                                            g['potential'])
        assert np.array_equal(got, w)


def test_g8_c1_c_oracle(golden):
    """Config C1 (500x600, 1000 tracks, seed 30) with the C port, all cores."""
    g = golden('g8_c1.npz')
    upd = orc.get_above_threshold_speed(g['orograph_f32'], 0.75)
    starts = np.stack([g['start_rows'], g['start_cols']], 1)
    res = c_oracle.simulate_tracks(0., starts, (500, 600), 1, 1., upd, g['potential'],
                                   seed=int(g['seed']))
    assert np.array_equal(res['lengths'], g['lengths'])
    assert np.array_equal(res['ends'], g['ends'])
    sha = hashlib.sha256()
    for t in res['tracks']:
        sha.update(np.ascontiguousarray(t, dtype='<i2').tobytes())
    assert sha.hexdigest() == str(g['traj_sha256'])
    assert np.array_equal(res['hist'].astype(np.int32), g['hist'])
    sm = c_oracle.smooth_presence(res['hist'], int(g['krad']))
    np.testing.assert_allclose(sm.max(), g['presence_max_raw'], rtol=3e-7)
    np.testing.assert_allclose((sm / sm.max())[::8, ::8], g['presence_strided'], rtol=3e-7,
                               atol=1e-9)


def test_g9_presence(golden):
    g = golden('g9_presence.npz')
    tracks = split(g['tracks'], g['lengths'])
    assert np.array_equal(orc.compute_presence_counts(tracks, (40, 50)), g['counts'])
    for rad in (2, 5, 13):
        assert np.array_equal(orc.compute_smooth_presence_counts(tracks, (40, 50), rad),
                              g[f'smooth_r{rad}'])
        c = c_oracle.smooth_presence(g['counts'].astype(np.uint32), rad)
        ulp = np.abs(c.view(np.int32).astype(np.int64) - g[f'smooth_r{rad}'].view(np.int32))
        assert ulp.max() <= 1
    assert orc.presence_kernel_radius(1000., 100., (500, 600)) == 10
    assert orc.presence_kernel_radius(1000., 10., (5000, 6000)) == 100
    assert orc.presence_kernel_radius(10., 100., (500, 600)) == 2


def test_cell_weights_match_window(golden):
    g = golden('g7_tracks.npz')
    prior = orc.get_directional_probs(0.)
    for r, c in [(5, 5), (40, 100), (94, 126), (1, 1)]:
        w = orc.window_weights(r, c, g['updraft'], g['potential'], prior)
        got = c_oracle.cell_weights(g['updraft'], g['potential'], r, c)
        want = np.array([max(v, 0.) for k, v in enumerate(w) if k != 4])
        assert np.array_equal(got, want)


def test_g10_10m_regime_c_oracle(g10):
    """G10: 1000 x 1200 window of the 10 m DEM, reference potential (assemble + SuperLU) and
    256 reference tracks: the C port reproduces every trajectory (sha256), the numpy
    restatement the first four."""
    import hashlib
    shape = g10['shape']
    upd = orc.get_above_threshold_speed(g10['orograph_f32'], 0.75)
    assert abs(float(np.mean(upd == 0)) - float(g10['dead_fraction'])) < 1e-12
    starts = np.stack([g10['start_rows'], g10['start_cols']], 1)
    res = c_oracle.simulate_tracks(0., starts, shape, 1, 1., upd, g10['potential'],
                                   seed=int(g10['seed']), want_traj=True, want_hist=True)
    assert np.array_equal(res['lengths'], g10['lengths'])
    assert np.array_equal(res['ends'], g10['ends'])
    sha = hashlib.sha256()
    for t in res['tracks']:
        sha.update(np.ascontiguousarray(t, dtype='<i2').tobytes())
    assert sha.hexdigest() == str(g10['traj_sha256'])
    assert np.array_equal(res['hist'].astype(np.int32), g10['hist'])
    # the reference's tracks cross this raster in ~1 step per row: nothing wanders
    steps = g10['lengths'] - 1
    assert steps.max() < 2 * shape[0] and steps.max() < int(g10['max_moves']) // 100
    off = 0
    for t in range(2):
        n = int(g10['first_lengths'][t])
        tr = orc.generate_simulated_tracks(0., (int(starts[t, 0]), int(starts[t, 1])), shape, 1, 1.,
                                           upd, g10['potential'], uniform=TrackUniforms(int(g10['seed']), t))
        assert np.array_equal(tr, g10['first_tracks'][off:off + n])
        off += n


def test_g11_wandering_tracks_c_oracle(g11):
    """G11: the 60 x 50 km domain at 50 m with the reference's own potential: 30 of the
    reference's 64 tracks never leave the raster and stop at max_moves = 300 000
    (movmodel.py:277,285).  The C port reproduces all of them point for point."""
    import hashlib
    shape = g11['shape']
    upd = orc.get_above_threshold_speed(g11['orograph_f32'], 0.75)
    starts = np.stack([g11['start_rows'], g11['start_cols']], 1)
    res = c_oracle.simulate_tracks(0., starts, shape, 1, 1., upd, g11['potential'],
                                   seed=int(g11['seed']), want_traj=True, want_hist=True)
    assert np.array_equal(res['lengths'], g11['lengths'])
    assert np.array_equal(res['ends'], g11['ends'])
    sha = hashlib.sha256()
    for t in res['tracks']:
        sha.update(np.ascontiguousarray(t, dtype='<i2').tobytes())
    assert sha.hexdigest() == str(g11['traj_sha256'])
    assert np.array_equal(res['hist'].astype(np.int32), g11['hist'])
    steps = g11['lengths'] - 1
    mm = int(g11['max_moves'])
    assert mm == 300000 and 0.4 < np.mean(steps >= mm) < 0.55      # the wandering IS the reference's
    assert np.median(steps[steps < mm]) < 3 * shape[0]             # the others cross in ~1 step per row
