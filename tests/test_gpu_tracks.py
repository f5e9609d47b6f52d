"""K2/K3 parity on the MI355X: the HIP stepper (through the C ABI) must be
BIT-EXACT -- trajectories, lengths, endpoints, histogram -- against
  (a) trajectories the reference itself produced (tests/golden/g7, g8), and
  (b) the C oracle on fresh seeded inputs.
Integer path => no tolerance anywhere in this file (nu == 1).
"""
import hashlib

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

G7_CASES = ['ff_m1', 'ff_m3', 'ff_d135_m2', 'drw_m1', 'drw_d250_m3']


def split(flat, lengths):
    off = np.concatenate([[0], np.cumsum(lengths)])
    return [flat[off[i]:off[i + 1]] for i in range(len(lengths))]


def test_uniform_contract_matches_oracle(gpu):
    """rocRAND Philox engine on the device == Random123-pinned oracle."""
    from ssrs_amd import movmodel
    from oracle.philox import uniform53
    rng = np.random.default_rng(1)
    track = rng.integers(0, 2**63, 4096, dtype=np.uint64)
    step = rng.integers(0, 2**40, 4096, dtype=np.uint64)
    track[:8] = [0, 1, 2, 3, 2**32 - 1, 2**32, 2**32 + 1, 2**64 - 1]
    step[:8] = [0, 1, 2, 3, 2**33 - 1, 2**33, 2**33 + 1, 7]
    for seed in (0, 30, 2**64 - 1, 0x123456789ABCDEF):
        got = movmodel.uniforms(seed, track, step)
        want = uniform53(seed, track, step)
        assert np.array_equal(got, want)
        assert got.min() >= 0.0 and got.max() < 1.0


@pytest.mark.parametrize('tag', G7_CASES)
@pytest.mark.parametrize('use_table', [False, True])
def test_g7_reference_trajectories(gpu, golden, tag, use_table):
    from ssrs_amd import movmodel
    g = golden('g7_tracks.npz')
    dirn, mem, nu, has_u, has_p = g[tag + '_params']
    if use_table and not has_u:
        pytest.skip('drw has no table')
    starts = np.stack([g['start_rows'], g['start_cols']], 1)
    res = movmodel.simulate_tracks(
        float(dirn), starts, (96, 128), int(mem), float(nu),
        g['updraft'] if has_u else None, g['potential'] if has_p else None,
        seed=int(g['seed']), use_table=use_table, want_tracks=True, steps_per_launch=16)
    lens = res.lengths.cpu().numpy()
    assert np.array_equal(lens, g[tag + '_lengths'])
    want = split(g[tag + '_tracks'], g[tag + '_lengths'])
    got = res.tracks()
    for t, (a, b) in enumerate(zip(got, want)):
        assert a.dtype == np.int16 and np.array_equal(a, b), f'track {t} differs'
    assert np.array_equal(res.ends.cpu().numpy(), np.array([w[-1] for w in want]))
    hist = np.zeros((96, 128), dtype=np.int64)
    np.add.at(hist, (g[tag + '_tracks'][:, 0].astype(int), g[tag + '_tracks'][:, 1].astype(int)), 1)
    assert np.array_equal(res.hist.cpu().numpy().astype(np.int64), hist)
    assert res.stats['total_steps'] == int(lens.sum() - len(lens))


def test_single_track_reference_signature(gpu, golden):
    from ssrs_amd import movmodel
    g = golden('g7_tracks.npz')
    want = split(g['ff_m1_tracks'], g['ff_m1_lengths'])
    for t in (0, 5, 63):
        tr = movmodel.generate_simulated_tracks(
            0., [int(g['start_rows'][t]), int(g['start_cols'][t])], (96, 128), 1, 1.,
            g['updraft'], g['potential'], seed=int(g['seed']), track_id=t)
        assert np.array_equal(tr, want[t])


def test_nu_half_within_tolerance(gpu, golden):
    """nu != 1 uses pow(): not bit-reproducible across libms, so only the
    statistics are compared."""
    from ssrs_amd import movmodel
    g = golden('g7_tracks.npz')
    starts = np.stack([g['start_rows'], g['start_cols']], 1)
    res = movmodel.simulate_tracks(0., starts, (96, 128), 1, 0.5, g['updraft'], g['potential'],
                                   seed=int(g['seed']), use_table=False)
    lens = res.lengths.cpu().numpy()
    same = (lens == g['ff_m1_nu05_lengths']).mean()
    assert same >= 0.9, f'only {same:.2f} of nu=0.5 tracks have the reference length'


def test_c1_golden_1000_tracks(gpu, golden):
    """BASELINE config 1 (500 x 600 @100 m, 1000 tracks, seed 30): every
    trajectory equals the reference's (sha256 over all int16 points)."""
    from ssrs_amd import movmodel, layers
    g = golden('g8_c1.npz')
    shape = (500, 600)
    upd = layers.get_above_threshold_speed(g['orograph_f32'], 0.75)
    starts = np.stack([g['start_rows'], g['start_cols']], 1)
    for use_table in (False, True):
        res = movmodel.simulate_tracks(0., starts, shape, 1, 1., upd, g['potential'],
                                       seed=int(g['seed']), use_table=use_table,
                                       want_tracks=True)
        assert np.array_equal(res.lengths.cpu().numpy(), g['lengths'])
        assert np.array_equal(res.ends.cpu().numpy(), g['ends'])
        tracks = res.tracks()
        sha = hashlib.sha256()
        for t in tracks:
            sha.update(np.ascontiguousarray(t, dtype='<i2').tobytes())
        assert sha.hexdigest() == str(g['traj_sha256'])
        assert np.array_equal(res.hist.cpu().numpy(), g['hist'])
        first = split(g['first_tracks'], g['first_lengths'])
        for a, b in zip(tracks[:8], first):
            assert np.array_equal(a, b)


def _random_field_case(rows, cols, seed):
    from ssrs_amd.synthetic import synthetic_dem
    from oracle import ssrs_oracle as orc
    rng = np.random.default_rng(seed)
    z = synthetic_dem((rows, cols), 100., seed=seed)
    slope = orc.compute_slope_degrees(z, 100.)
    aspect = orc.compute_aspect_degrees(z, 100.)
    oro = orc.compute_orographic_updraft(10., 270., slope, aspect).astype(np.float32)
    upd = orc.get_above_threshold_speed(oro, 0.75)
    # rough potential: ramp + noise, so that E/W/S moves and dead ends occur
    pot = (1000. * (1 - np.arange(rows)[:, None] / (rows - 1.)) +
           rng.normal(0, 1.5, (rows, cols))).astype(np.float32)
    return upd, pot


@pytest.mark.parametrize('mem,dirn', [(1, 0.), (2, 40.), (8, 180.), (0, 0.)])
def test_vs_c_oracle_fresh_inputs(gpu, mem, dirn):
    from ssrs_amd import movmodel
    from oracle import c_oracle
    rows, cols = 150, 170
    upd, pot = _random_field_case(rows, cols, 99 + mem)
    rng = np.random.default_rng(mem)
    n = 700
    starts = np.stack([rng.integers(0, rows, n), rng.integers(0, cols, n)], 1)
    ref = c_oracle.simulate_tracks(dirn, starts, (rows, cols), mem, 1., upd, pot, seed=7,
                                   track_id_base=1000)
    for use_table in (False, True):
        res = movmodel.simulate_tracks(dirn, starts, (rows, cols), mem, 1., upd, pot, seed=7,
                                       track_id_base=1000, use_table=use_table,
                                       want_tracks=True, steps_per_launch=64)
        assert np.array_equal(res.lengths.cpu().numpy(), ref['lengths'])
        assert np.array_equal(res.ends.cpu().numpy(), ref['ends'])
        assert np.array_equal(res.hist.cpu().numpy().view(np.uint32), ref['hist'])
        for a, b in zip(res.tracks(), ref['tracks']):
            assert np.array_equal(a, b)


def test_nan_and_zero_fields_take_reference_fallbacks(gpu):
    """NaN weights -> directional prior (movmodel.py:228-230); all-zero masked
    weights -> prior (:234-240).  Flat potential gives all-zero weights."""
    from ssrs_amd import movmodel
    from oracle import c_oracle
    rows, cols = 60, 70
    upd = np.full((rows, cols), 0.5)
    pot = np.full((rows, cols), 3.0, dtype=np.float32)          # zero differences
    upd[20:25, 30:40] = np.nan
    pot[40:42, 10:60] = np.nan
    rng = np.random.default_rng(3)
    starts = np.stack([rng.integers(0, 12, 300), rng.integers(0, cols, 300)], 1)
    ref = c_oracle.simulate_tracks(0., starts, (rows, cols), 1, 1., upd, pot, seed=5)
    for use_table in (False, True):
        res = movmodel.simulate_tracks(0., starts, (rows, cols), 1, 1., upd, pot, seed=5,
                                       use_table=use_table, want_tracks=True)
        for a, b in zip(res.tracks(), ref['tracks']):
            assert np.array_equal(a, b)


def test_updraft_only_mode(gpu):
    """No potential: harmonic-mean weights only (MODE_UPDRAFT kernel)."""
    from ssrs_amd import movmodel
    from oracle import c_oracle
    rows, cols = 24, 26
    rng = np.random.default_rng(11)
    upd = rng.uniform(0., 2., (rows, cols))
    starts = np.stack([rng.integers(0, rows, 128), rng.integers(0, cols, 128)], 1)
    ref = c_oracle.simulate_tracks(0., starts, (rows, cols), 1, 1., upd, None, seed=1)
    for use_table in (False, True):
        res = movmodel.simulate_tracks(0., starts, (rows, cols), 1, 1., upd, None, seed=1,
                                       use_table=use_table, want_tracks=True, steps_per_launch=7)
        assert np.array_equal(res.lengths.cpu().numpy(), ref['lengths'])
        for a, b in zip(res.tracks(), ref['tracks']):
            assert np.array_equal(a, b)


def test_max_moves_exit(gpu):
    """A vortex potential keeps tracks orbiting until k == max_moves
    (rows/2*cols/2, movmodel.py:277,285): exercises the non-border exit and a
    launch count that is not a multiple of steps_per_launch."""
    from ssrs_amd import movmodel
    from oracle import c_oracle
    rows, cols = 31, 33
    r = np.arange(rows)[:, None] - 15.
    c = np.arange(cols)[None, :] - 16.
    theta = np.mod(np.arctan2(r, c), 2 * np.pi)
    pot = (100. * (2 * np.pi - theta) + 2. * (np.hypot(r, c) - 8.)**2).astype(np.float32)
    upd = np.ones((rows, cols))
    rng = np.random.default_rng(11)
    starts = np.stack([rng.integers(5, 26, 128), rng.integers(5, 28, 128)], 1)
    ref = c_oracle.simulate_tracks(0., starts, (rows, cols), 1, 1., upd, pot, seed=1)
    assert ref['lengths'].max() == 257 and (ref['lengths'] == 257).sum() > 100
    for use_table in (False, True):
        res = movmodel.simulate_tracks(0., starts, (rows, cols), 1, 1., upd, pot, seed=1,
                                       use_table=use_table, want_tracks=True, steps_per_launch=50)
        assert np.array_equal(res.lengths.cpu().numpy(), ref['lengths'])
        for a, b in zip(res.tracks(), ref['tracks']):
            assert np.array_equal(a, b)


def test_edge_cases_and_errors(gpu):
    from ssrs_amd import movmodel
    upd = np.ones((40, 50))
    pot = np.zeros((40, 50), dtype=np.float32)
    res = movmodel.simulate_tracks(0., np.zeros((0, 2), dtype=int), (40, 50), 1, 1., upd, pot)
    assert res.lengths.numel() == 0 and int(res.hist.sum()) == 0
    with pytest.raises(ValueError):     # start outside the raster
        movmodel.simulate_tracks(0., [[40, 3]], (40, 50), 1, 1., upd, pot)
    with pytest.raises(ValueError):     # reference raises for potential without updraft
        movmodel.simulate_tracks(0., [[4, 3]], (40, 50), 1, 1., None, pot)
    with pytest.raises(ValueError):
        movmodel.simulate_tracks(0., [[4, 3]], (40, 50), 9, 1., upd, pot)
    with pytest.raises(ValueError):
        movmodel.simulate_tracks(0., [[4, 3]], (40, 51), 1, 1., upd, pot)


def test_sharding_invariance_and_hist_checksum(gpu):
    """Tracks split over 'ranks' with track_id_base offsets == one big run:
    the property the multi-GPU path relies on.  Also sum(hist) == sum(lengths)."""
    from ssrs_amd import movmodel
    rows, cols = 300, 320
    upd, pot = _random_field_case(rows, cols, 4)
    rng = np.random.default_rng(8)
    n = 5000
    starts = np.stack([rng.integers(2, 30, n), rng.integers(0, cols, n)], 1)
    whole = movmodel.simulate_tracks(0., starts, (rows, cols), 1, 1., upd, pot, seed=30)
    parts = None
    lens = []
    for lo, hi in [(0, 1234), (1234, 3000), (3000, 5000)]:
        part = movmodel.simulate_tracks(0., starts[lo:hi], (rows, cols), 1, 1., upd, pot,
                                        seed=30, track_id_base=lo, hist=parts)
        parts = part.hist
        lens.append(part.lengths)
    assert torch.equal(torch.cat(lens), whole.lengths)
    assert torch.equal(parts, whole.hist)
    assert int(whole.hist.sum()) == int(whole.lengths.sum())
    again = movmodel.simulate_tracks(0., starts, (rows, cols), 1, 1., upd, pot, seed=30,
                                     use_table=True)
    assert torch.equal(again.hist, whole.hist) and torch.equal(again.ends, whole.ends)


def test_concurrent_streams_from_host_threads(gpu):
    """Seasonal-mode shape (BASELINE configs[4]: one snapshot per stream): four
    host threads, each with its own HIP stream and its own updraft/potential,
    run ssrs_tracks_simulate concurrently; every result equals its serial run."""
    import threading
    from ssrs_amd import movmodel
    rows, cols = 200, 220
    cases = []
    rng = np.random.default_rng(21)
    for s in range(4):
        upd, pot = _random_field_case(rows, cols, 300 + s)
        starts = np.stack([rng.integers(2, 20, 1500), rng.integers(0, cols, 1500)], 1)
        cases.append((torch.from_numpy(upd).cuda(), torch.from_numpy(pot).cuda(),
                      torch.from_numpy(starts.astype(np.int32)).cuda(), 1000 + s))
    serial = [movmodel.simulate_tracks(0., st, (rows, cols), 1, 1., u, p, seed=sd)
              for u, p, st, sd in cases]
    torch.cuda.synchronize()
    results = [None] * 4
    errors = []

    def work(i):
        try:
            stream = torch.cuda.Stream()
            with torch.cuda.stream(stream):
                u, p, st, sd = cases[i]
                results[i] = movmodel.simulate_tracks(0., st, (rows, cols), 1, 1., u, p, seed=sd)
            stream.synchronize()
        except Exception as exc:          # pragma: no cover
            errors.append(exc)

    threads = [threading.Thread(target=work, args=(i,)) for i in range(4)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    for a, b in zip(results, serial):
        assert torch.equal(a.lengths, b.lengths) and torch.equal(a.ends, b.ends)
        assert torch.equal(a.hist, b.hist)


def test_c2_scale_properties(gpu):
    """BASELINE configs[1] at full size (5000 x 6000, 100k tracks): properties
    that need no oracle -- histogram checksum, determinism across data paths and
    schedules, and a 2000-track sample bit-exact against the C oracle."""
    from ssrs_amd import movmodel, layers
    from ssrs_amd.synthetic import synthetic_dem, ramp_potential
    from oracle import c_oracle
    rows, cols, res = 5000, 6000, 10.
    dem = torch.from_numpy(synthetic_dem((rows, cols), res)).cuda()
    _, upd = layers.updraft_from_dem(dem, res, 10., 270., threshold=0.75)
    pot = torch.from_numpy(ramp_potential((rows, cols))).cuda()
    np.random.seed(30)
    r, c = movmodel.get_starting_indices(100000, (5, 55, 1, 2), 'random', (60., 50.), res)
    starts = np.stack([r, c], 1).astype(np.int32)
    a = movmodel.simulate_tracks(0., starts, (rows, cols), 1, 1., upd, pot, seed=30, use_table=True)
    assert int(a.hist.sum()) == int(a.lengths.sum())
    assert int(a.lengths.min()) >= 4800 and int(a.lengths.max()) <= 4902
    b = movmodel.simulate_tracks(0., starts, (rows, cols), 1, 1., upd, pot, seed=30, use_table=True,
                                 schedule=False, binning=False, exact_only=True)
    assert torch.equal(a.lengths, b.lengths) and torch.equal(a.ends, b.ends)
    assert torch.equal(a.hist, b.hist)
    m = 2000
    ref = c_oracle.simulate_tracks(0., starts[:m], (rows, cols), 1, 1., upd.cpu().numpy(),
                                   pot.cpu().numpy(), seed=30, want_traj=False)
    assert np.array_equal(a.lengths[:m].cpu().numpy(), ref['lengths'])
    assert np.array_equal(a.ends[:m].cpu().numpy(), ref['ends'])


def test_binned_histogram_path_with_trajectories(gpu):
    """>= 8192 tracks switches the histogram to the visit-buffer + LDS-binning
    kernel (k_bin_visits); with trajectories requested at the same time every
    output must still equal the C oracle's, and the A/B switches must agree."""
    from ssrs_amd import movmodel, presence
    from oracle import c_oracle
    rows, cols = 300, 320
    upd, pot = _random_field_case(rows, cols, 77)
    rng = np.random.default_rng(5)
    n = 10000
    starts = np.stack([rng.integers(2, 40, n), rng.integers(0, cols, n)], 1)
    ref = c_oracle.simulate_tracks(0., starts, (rows, cols), 1, 1., upd, pot, seed=11,
                                   track_id_base=5)
    got = movmodel.simulate_tracks(0., starts, (rows, cols), 1, 1., upd, pot, seed=11,
                                   track_id_base=5, use_table=True, want_tracks=True)
    assert np.array_equal(got.lengths.cpu().numpy(), ref['lengths'])
    assert np.array_equal(got.ends.cpu().numpy(), ref['ends'])
    assert np.array_equal(got.hist.cpu().numpy().view(np.uint32), ref['hist'])
    assert np.array_equal(got.traj.cpu().numpy(), np.concatenate(ref['tracks']))
    # histogram rebuilt from the stored trajectories (K3' kernel) agrees as well
    again = presence.compute_presence_counts(got.traj, (rows, cols))
    assert torch.equal(again, got.hist)
    for kw in (dict(binning=False), dict(schedule=False), dict(use_table=False)):
        kw = dict(dict(use_table=True), **kw)
        alt = movmodel.simulate_tracks(0., starts, (rows, cols), 1, 1., upd, pot, seed=11,
                                       track_id_base=5, **kw)
        assert torch.equal(alt.hist, got.hist) and torch.equal(alt.lengths, got.lengths)


# ------------------------------------------------------------------ ring table
# The f32 ring table (one 12-byte gather per step) must give the reference's
# tracks bit for bit: its guarded decision falls back to the exact sequence on
# the raw windows whenever f32 rounding could matter.

def _no_traj_result(res):
    return (res.lengths.cpu().numpy(), res.ends.cpu().numpy(),
            res.hist.cpu().numpy().view(np.uint32))


@pytest.mark.parametrize('ring', [False, True])
def test_ring_table_c1_golden(gpu, golden, ring):
    from ssrs_amd import movmodel, layers
    g = golden('g8_c1.npz')
    upd = layers.get_above_threshold_speed(g['orograph_f32'], 0.75)
    starts = np.stack([g['start_rows'], g['start_cols']], 1)
    for spl in (0, 64):
        res = movmodel.simulate_tracks(0., starts, (500, 600), 1, 1., upd, g['potential'],
                                       seed=int(g['seed']), use_table=True, ring=ring,
                                       steps_per_launch=spl)
        lens, ends, hist = _no_traj_result(res)
        assert np.array_equal(lens, g['lengths'])
        assert np.array_equal(ends, g['ends'])
        assert np.array_equal(hist, g['hist'].view(np.uint32))


@pytest.mark.parametrize('tag', ['ff_m1', 'ff_d135_m2', 'ff_m1_nu05'])
def test_ring_table_g7_cases(gpu, golden, tag):
    """memory 1 cases use the ring table; others must be refused loudly."""
    from ssrs_amd import movmodel
    g = golden('g7_tracks.npz')
    dirn, mem, nu, has_u, has_p = g[tag + '_params']
    starts = np.stack([g['start_rows'], g['start_cols']], 1)
    args = (float(dirn), starts, (96, 128), int(mem), float(nu), g['updraft'],
            g['potential'] if has_p else None)
    if int(mem) == 1 and float(nu) == 1.0:
        res = movmodel.simulate_tracks(*args, seed=int(g['seed']), use_table=True, ring=True,
                                       steps_per_launch=16)
        assert np.array_equal(res.lengths.cpu().numpy(), g[tag + '_lengths'])
    else:
        with pytest.raises(ValueError):
            movmodel.simulate_tracks(*args, seed=int(g['seed']), use_table=True, ring=True)


@pytest.mark.parametrize('case', ['rough', 'nan_zero', 'updraft_only', 'tiny', 'huge'])
def test_ring_table_vs_c_oracle(gpu, case):
    """Fresh inputs incl. the rows the fast path must hand to the exact one:
    NaN (poisoned) and all-zero rows, weights below the f32 range, f32 overflow."""
    from ssrs_amd import movmodel
    from oracle import c_oracle
    rows, cols = 150, 170
    upd, pot = _random_field_case(rows, cols, 31)
    if case == 'nan_zero':
        upd = upd.copy(); pot = pot.copy()
        upd[20:25, 30:40] = np.nan
        pot[60:62, 10:160] = np.nan
        pot[90:110, :] = 3.0                       # zero differences
    elif case == 'updraft_only':
        pot = None
    elif case == 'tiny':
        pot = (pot.astype(np.float64) * 1e-37).astype(np.float32)    # weights ~1e-40 .. 1e-37
    elif case == 'huge':
        upd = upd * 1e30
        pot = (pot.astype(np.float64) * 1e30).astype(np.float32)     # weights up to ~1e62: f32 inf
    rng = np.random.default_rng(5)
    n = 900
    starts = np.stack([rng.integers(0, rows, n), rng.integers(0, cols, n)], 1)
    ref = c_oracle.simulate_tracks(20., starts, (rows, cols), 1, 1., upd, pot, seed=11,
                                   track_id_base=77, want_traj=False)
    for ring, scattered in ((False, None), (True, False), (True, True)):
        res = movmodel.simulate_tracks(20., starts, (rows, cols), 1, 1., upd, pot, seed=11,
                                       track_id_base=77, use_table=True, ring=ring,
                                       scattered=scattered, steps_per_launch=32)
        lens, ends, hist = _no_traj_result(res)
        assert np.array_equal(lens, ref['lengths']), (case, ring, scattered)
        assert np.array_equal(ends, ref['ends']), (case, ring, scattered)
        assert np.array_equal(hist, ref['hist']), (case, ring, scattered)


def test_ring_table_large_batch_equals_f64_table(gpu):
    """Binning + per-XCD lists + ring table at a size where all of them are on."""
    from ssrs_amd import movmodel
    rows, cols = 600, 900
    upd, pot = _random_field_case(rows, cols, 3)
    rng = np.random.default_rng(8)
    n = 20000
    starts = np.stack([rng.integers(1, 12, n), rng.integers(0, cols, n)], 1)
    a = _no_traj_result(movmodel.simulate_tracks(0., starts, (rows, cols), 1, 1., upd, pot, seed=3,
                                                 use_table=True, ring=False))
    b = _no_traj_result(movmodel.simulate_tracks(0., starts, (rows, cols), 1, 1., upd, pot, seed=3,
                                                 use_table=True, ring=True))
    for x, y in zip(a, b):
        assert np.array_equal(x, y)
    assert int(b[2].sum()) == int(b[0].sum())


@pytest.mark.parametrize('rows,cols,n,dirn', [(5, 5, 3, 0.), (6, 40, 65, 90.), (33, 7, 129, 200.),
                                              (64, 31000, 300, 0.), (257, 263, 9000, 315.)])
def test_odd_shapes_all_paths_agree_with_oracle(gpu, rows, cols, n, dirn):
    """Smallest legal raster, single waves, rasters wider than the binning window,
    batches on both sides of the binning threshold: ring table, f64 table and window
    gathers all equal the oracle."""
    from ssrs_amd import movmodel
    from oracle import c_oracle
    rng = np.random.default_rng(rows * 1000 + cols)
    upd = np.abs(rng.normal(0.8, 0.6, (rows, cols)))
    upd[rng.random((rows, cols)) < 0.3] = 0.0                     # dead cells
    pot = (1000. * (1 - np.arange(rows)[:, None] / max(rows - 1., 1.)) +
           rng.normal(0, 2.0, (rows, cols))).astype(np.float32)
    starts = np.stack([rng.integers(0, rows, n), rng.integers(0, cols, n)], 1)
    ref = c_oracle.simulate_tracks(dirn, starts, (rows, cols), 1, 1., upd, pot, seed=5, want_traj=False)
    for kw in (dict(use_table=True, ring=True), dict(use_table=True, ring=True, scattered=True),
               dict(use_table=True, ring=False), dict(use_table=False)):
        res = movmodel.simulate_tracks(dirn, starts, (rows, cols), 1, 1., upd, pot, seed=5, **kw)
        lens, ends, hist = _no_traj_result(res)
        assert np.array_equal(lens, ref['lengths']), kw
        assert np.array_equal(ends, ref['ends']), kw
        assert np.array_equal(hist, ref['hist']), kw


@pytest.mark.parametrize('mem,use_table', [(2, True), (1, False), (0, True)])
def test_privatised_histogram_in_the_generic_kernels(gpu, mem, use_table):
    """scattered=True from the first launch: wave-private histogram copies folded at the
    end, in the kernels that serve other movement models too."""
    from ssrs_amd import movmodel
    from oracle import c_oracle
    rows, cols = 120, 140
    upd, pot = _random_field_case(rows, cols, 17)
    rng = np.random.default_rng(mem)
    n = 1500
    starts = np.stack([rng.integers(0, rows, n), rng.integers(0, cols, n)], 1)
    ref = c_oracle.simulate_tracks(0., starts, (rows, cols), mem, 1., upd, pot, seed=9, want_traj=False)
    res = movmodel.simulate_tracks(0., starts, (rows, cols), mem, 1., upd, pot, seed=9,
                                   use_table=use_table, ring=False, scattered=True)
    lens, ends, hist = _no_traj_result(res)
    assert np.array_equal(lens, ref['lengths'])
    assert np.array_equal(hist, ref['hist'])


def test_no_histogram_requested(gpu):
    """want_hist=False: no visit buffer, no atomics; lengths and end cells only."""
    from ssrs_amd import movmodel
    from oracle import c_oracle
    rows, cols = 90, 110
    upd, pot = _random_field_case(rows, cols, 23)
    rng = np.random.default_rng(1)
    starts = np.stack([rng.integers(0, rows, 400), rng.integers(0, cols, 400)], 1)
    ref = c_oracle.simulate_tracks(0., starts, (rows, cols), 1, 1., upd, pot, seed=2, want_traj=False)
    for kw in (dict(use_table=True, ring=True), dict(use_table=True, ring=False), dict(use_table=False)):
        res = movmodel.simulate_tracks(0., starts, (rows, cols), 1, 1., upd, pot, seed=2, want_hist=False, **kw)
        assert res.hist is None
        assert np.array_equal(res.lengths.cpu().numpy(), ref['lengths']), kw
        assert np.array_equal(res.ends.cpu().numpy(), ref['ends']), kw


@pytest.mark.parametrize('dirn', [90., 270., 0., 45., 135., 200., 30.])
def test_binned_histogram_for_every_heading(gpu, dirn):
    """East / west batches bin into a transposed histogram (their front is a column);
    north / south into the plain one; oblique fronts through tile buckets (k_tile_sort /
    k_bin_bucket).  Large enough (>= 8192 tracks, starts at the upstream edge) for the
    binning paths to be taken."""
    from ssrs_amd import movmodel
    from oracle import c_oracle
    rows, cols = 260, 330
    upd, _ = _random_field_case(rows, cols, 41)
    th = np.deg2rad(dirn)
    rr = np.arange(rows)[:, None]; cc = np.arange(cols)[None, :]
    along = rr * np.cos(th) + cc * np.sin(th)
    rng = np.random.default_rng(6)
    pot = (1000. * (1 - (along - along.min()) / (along.max() - along.min())) +
           rng.normal(0, 0.05, (rows, cols))).astype(np.float32)
    n = 9000
    t = rng.integers(2, 12, n)
    if dirn == 90.:
        starts = np.stack([rng.integers(0, rows, n), t], 1)
    elif dirn == 270.:
        starts = np.stack([rng.integers(0, rows, n), cols - 1 - t], 1)
    elif np.cos(th) < 0:
        starts = np.stack([rows - 1 - t, rng.integers(0, cols, n)], 1)
    else:
        starts = np.stack([t, rng.integers(0, cols, n)], 1)
    ref = c_oracle.simulate_tracks(dirn, starts, (rows, cols), 1, 1., upd, pot, seed=13, want_traj=False)
    for kw in (dict(ring=True), dict(ring=False)):
        res = movmodel.simulate_tracks(dirn, starts, (rows, cols), 1, 1., upd, pot, seed=13, use_table=True,
                                       profile=True, **kw)
        lens, ends, hist = _no_traj_result(res)
        assert np.array_equal(lens, ref['lengths']), (dirn, kw)
        assert np.array_equal(hist, ref['hist']), (dirn, kw)
        assert res.stats['hist_ms'] > 0.0, 'the binning path was not taken'
        if dirn in (0., 90., 270.):
            assert res.stats['window_launches'] > 0 and res.stats['tile_launches'] == 0
        else:
            assert res.stats['tile_launches'] > 0 and res.stats['window_launches'] == 0


@pytest.mark.parametrize('rows,cols,n,dirn,same_start', [(200, 200, 70000, 45., True), (40, 32000, 9000, 60., False),
                                                         (3000, 1400, 20000, 20., False)])
def test_tile_binning_corner_cases(gpu, rows, cols, n, dirn, same_start):
    """Tile-bucketed binning: 16-bit LDS counters that overflow within one launch (70 000
    tracks from one cell), a raster one tile high and 32 wide, and random starts all over
    the raster (about two visits per cell and launch: the host leaves the tile path)."""
    from ssrs_amd import movmodel
    from oracle import c_oracle
    upd, _ = _random_field_case(rows, cols, 43)
    th = np.deg2rad(dirn)
    rr = np.arange(rows)[:, None]; cc = np.arange(cols)[None, :]
    along = rr * np.cos(th) + cc * np.sin(th)
    rng = np.random.default_rng(8)
    pot = (1000. * (1 - (along - along.min()) / (along.max() - along.min())) +
           rng.normal(0, 0.05, (rows, cols))).astype(np.float32)
    if same_start:
        starts = np.tile(np.array([[3, 3]]), (n, 1))
    else:
        starts = np.stack([rng.integers(0, rows, n), rng.integers(0, cols, n)], 1)
    ref = c_oracle.simulate_tracks(dirn, starts, (rows, cols), 1, 1., upd, pot, seed=21, want_traj=False)
    res = movmodel.simulate_tracks(dirn, starts, (rows, cols), 1, 1., upd, pot, seed=21, use_table=True, ring=True)
    lens, ends, hist = _no_traj_result(res)
    assert np.array_equal(lens, ref['lengths'])
    assert np.array_equal(hist, ref['hist'])
    assert res.stats['tile_launches'] > 0
    if same_start:
        assert int(ref['hist'].max()) > 0x8000


@pytest.mark.parametrize('dirn', [0., 90.])
def test_wandering_batches_move_from_the_window_to_tile_buckets(gpu, dirn):
    """A ramp potential with wells wide enough to turn a track around (+-45 degrees per
    step): a fifth of the tracks circle in a well until max_moves, the row / column window
    starts missing and the host switches the batch to tile buckets (east / west batches
    also from transposed to plain visit keys) - same histogram."""
    from ssrs_amd import movmodel
    from oracle import c_oracle
    rows, cols, n = 240, 260, 9000
    rng = np.random.default_rng(12)
    upd = np.abs(rng.normal(0.8, 0.6, (rows, cols)))
    th = np.deg2rad(dirn)
    rr = np.arange(rows)[:, None]; cc = np.arange(cols)[None, :]
    along = rr * np.cos(th) + cc * np.sin(th)
    pot = 1000. * (1 - (along - along.min()) / (along.max() - along.min()))
    for _ in range(30):
        r0, c0 = rng.integers(20, rows - 20), rng.integers(20, cols - 20)
        pot = pot - 300. * np.exp(-((rr - r0) ** 2 + (cc - c0) ** 2) / (2. * 8. * 8.))
    pot = pot.astype(np.float32)
    starts = np.stack([rng.integers(0, rows, n), rng.integers(0, cols, n)], 1)
    ref = c_oracle.simulate_tracks(dirn, starts, (rows, cols), 1, 1., upd, pot, seed=33, want_traj=False)
    assert int((ref['lengths'] > 4 * (rows + cols)).sum()) > n // 10, 'the case is meant to wander'
    for kw in (dict(ring=True), dict(ring=False)):
        res = movmodel.simulate_tracks(dirn, starts, (rows, cols), 1, 1., upd, pot, seed=33, use_table=True, **kw)
        lens, ends, hist = _no_traj_result(res)
        assert np.array_equal(lens, ref['lengths']), kw
        assert np.array_equal(hist, ref['hist']), kw
        assert res.stats['window_launches'] > 0 and res.stats['tile_launches'] > 0, res.stats


# ------------------------------------------------------------ recorded trajectories
# want_tracks=True records every launch's visits and gathers them afterwards (one
# simulation pass, any stepper path); record=False is the two-pass form.

@pytest.mark.parametrize('dirn,n,kw', [
    (0., 9000, dict(use_table=True, ring=True)),            # row window binning, ring kernel
    (0., 9000, dict(use_table=True, ring=False)),           # f64 three-candidate kernel
    (90., 9000, dict(use_table=True, ring=True)),           # transposed visit keys
    (45., 9000, dict(use_table=True, ring=True)),           # tile buckets
    (200., 700, dict(use_table=True, ring=True)),           # small batch: per-visit atomics
    (0., 700, dict(use_table=False)),                       # window gathers, generic kernel
    (30., 9000, dict(use_table=True, memory=3)),            # generic table kernel
    (0., 9000, dict(use_table=True, ring=True, schedule=False)),   # identity first list
    (0., 9000, dict(use_table=True, ring=True, steps_per_launch=34)),
    (0., 9000, dict(use_table=True, thr=True)),             # threshold stepper: row window
    (270., 9000, dict(use_table=True, thr=True)),           # ... transposed keys (first move plain)
    (135., 9000, dict(use_table=True, thr=True)),           # ... tile buckets
    (20., 600, dict(use_table=True, thr=True, schedule=False)),
])
def test_recorded_trajectories_equal_the_oracle(gpu, dirn, n, kw):
    from ssrs_amd import movmodel
    from oracle import c_oracle
    rows, cols = 260, 330
    upd, _ = _random_field_case(rows, cols, 41)
    th = np.deg2rad(dirn)
    rr = np.arange(rows)[:, None]; cc = np.arange(cols)[None, :]
    along = rr * np.cos(th) + cc * np.sin(th)
    rng = np.random.default_rng(8)
    pot = (1000. * (1 - (along - along.min()) / (along.max() - along.min())) +
           rng.normal(0, 0.05, (rows, cols))).astype(np.float32)
    starts = np.stack([rng.integers(2, rows - 2, n), rng.integers(2, cols - 2, n)], 1)
    kw = dict(kw)
    mem = kw.pop('memory', 1)
    ref = c_oracle.simulate_tracks(dirn, starts, (rows, cols), mem, 1., upd, pot, seed=17, track_id_base=3)
    res = movmodel.simulate_tracks(dirn, starts, (rows, cols), mem, 1., upd, pot, seed=17, track_id_base=3,
                                   want_tracks=True, **kw)
    assert res.stats['recorded'], 'the trajectory pool was exhausted'
    assert np.array_equal(res.lengths.cpu().numpy(), ref['lengths'])
    assert np.array_equal(res.ends.cpu().numpy(), ref['ends'])
    assert np.array_equal(res.hist.cpu().numpy().view(np.uint32), ref['hist'])
    assert np.array_equal(res.traj.cpu().numpy(), np.concatenate(ref['tracks']))


def test_record_pool_overflow_falls_back_to_two_passes(gpu):
    from ssrs_amd import movmodel
    from oracle import c_oracle
    rows, cols = 120, 150
    upd, pot = _random_field_case(rows, cols, 5)
    rng = np.random.default_rng(9)
    starts = np.stack([rng.integers(2, 20, 3000), rng.integers(0, cols, 3000)], 1)
    ref = c_oracle.simulate_tracks(0., starts, (rows, cols), 1, 1., upd, pot, seed=4)
    hist0 = torch.full((rows, cols), 5, dtype=torch.int32, device='cuda')     # accumulated into
    for pool, expect in ((1 << 20, False), (None, True)):
        hist = hist0.clone()
        res = movmodel.simulate_tracks(0., starts, (rows, cols), 1, 1., upd, pot, seed=4, use_table=True,
                                       want_tracks=True, record_pool_bytes=pool, steps_per_launch=64, hist=hist)
        assert res.stats['recorded'] == expect
        assert np.array_equal(res.traj.cpu().numpy(), np.concatenate(ref['tracks']))
        assert np.array_equal((res.hist - 5).cpu().numpy().view(np.uint32), ref['hist'])
    two = movmodel.simulate_tracks(0., starts, (rows, cols), 1, 1., upd, pot, seed=4, use_table=True,
                                   want_tracks=True, record=False)
    assert not two.stats['recorded'] and torch.equal(two.traj, res.traj)


# -------------------------------------------------------------- threshold table
# The threshold table (two decision thresholds per cell and last move, one 8-byte gather and
# two comparisons per step) is the default stepper path; like the ring table it must give the
# reference's tracks bit for bit, handing near-ties and flagged rows to the exact sequence.

def test_thr_table_c1_golden(gpu, golden):
    from ssrs_amd import movmodel, layers
    g = golden('g8_c1.npz')
    upd = layers.get_above_threshold_speed(g['orograph_f32'], 0.75)
    starts = np.stack([g['start_rows'], g['start_cols']], 1)
    for spl in (0, 64, 2):
        res = movmodel.simulate_tracks(0., starts, (500, 600), 1, 1., upd, g['potential'],
                                       seed=int(g['seed']), use_table=True, thr=True, steps_per_launch=spl)
        lens, ends, hist = _no_traj_result(res)
        assert np.array_equal(lens, g['lengths'])
        assert np.array_equal(ends, g['ends'])
        assert np.array_equal(hist, g['hist'].view(np.uint32))
    rec = movmodel.simulate_tracks(0., starts, (500, 600), 1, 1., upd, g['potential'], seed=int(g['seed']),
                                   use_table=True, thr=True, want_tracks=True)
    assert rec.stats['recorded']
    sha = hashlib.sha256()
    for t in rec.tracks():
        sha.update(np.ascontiguousarray(t, dtype='<i2').tobytes())
    assert sha.hexdigest() == str(g['traj_sha256'])


@pytest.mark.parametrize('tag', ['ff_m1', 'ff_d135_m2', 'ff_m1_nu05'])
def test_thr_table_g7_cases(gpu, golden, tag):
    """G7 starts include border rows: the burn-in nudge goes through the flagged entries."""
    from ssrs_amd import movmodel
    g = golden('g7_tracks.npz')
    dirn, mem, nu, has_u, has_p = g[tag + '_params']
    starts = np.stack([g['start_rows'], g['start_cols']], 1)
    args = (float(dirn), starts, (96, 128), int(mem), float(nu), g['updraft'],
            g['potential'] if has_p else None)
    if int(mem) == 1 and float(nu) == 1.0:
        res = movmodel.simulate_tracks(*args, seed=int(g['seed']), use_table=True, thr=True,
                                       steps_per_launch=16, want_tracks=True)
        assert np.array_equal(res.lengths.cpu().numpy(), g[tag + '_lengths'])
        for a, b in zip(res.tracks(), split(g[tag + '_tracks'], g[tag + '_lengths'])):
            assert np.array_equal(a, b)
    else:
        with pytest.raises(ValueError):
            movmodel.simulate_tracks(*args, seed=int(g['seed']), use_table=True, thr=True)


@pytest.mark.parametrize('case', ['rough', 'nan_zero', 'updraft_only', 'tiny', 'huge', 'flat', 'south', 'scales'])
def test_thr_table_vs_c_oracle(gpu, case):
    """Fresh inputs incl. every flagged kind of row: NaN (poisoned) and all-zero rows (masked prior
    thresholds), weights below the f32 range and beyond it, a flat potential (the prior decides
    every step) and a heading against which the masked prior vanishes (reversal rows)."""
    from ssrs_amd import movmodel
    from oracle import c_oracle
    rows, cols = 150, 170
    upd, pot = _random_field_case(rows, cols, 31)
    dirn = 20.
    if case == 'nan_zero':
        upd = upd.copy(); pot = pot.copy()
        upd[20:25, 30:40] = np.nan
        pot[60:62, 10:160] = np.nan
        pot[90:110, :] = 3.0                       # zero differences
    elif case == 'updraft_only':
        pot = None
    elif case == 'tiny':
        pot = (pot.astype(np.float64) * 1e-37).astype(np.float32)
    elif case == 'huge':
        upd = upd * 1e30
        pot = (pot.astype(np.float64) * 1e30).astype(np.float32)
    elif case == 'flat':
        pot = np.full_like(pot, 7.0)
    elif case == 'scales':                         # the f32 table builder: every magnitude in one field
        r2 = np.random.default_rng(77)
        upd = upd * 10. ** r2.uniform(-9, 39, upd.shape)
        upd[r2.random(upd.shape) < 0.01] = np.inf
        band = 10. ** r2.integers(-44, 8, rows // 10 + 1).astype(np.float64)
        pot = (pot.astype(np.float64) * np.repeat(band, 10)[:rows, None]).astype(np.float32)
    elif case == 'south':
        dirn = 180.                                # the potential still pulls north: reversals
        pot = pot.copy()
        pot[40:110, :] = pot[40, 0]
    rng = np.random.default_rng(5)
    n = 900
    starts = np.stack([rng.integers(0, rows, n), rng.integers(0, cols, n)], 1)
    ref = c_oracle.simulate_tracks(dirn, starts, (rows, cols), 1, 1., upd, pot, seed=11,
                                   track_id_base=77, want_traj=False)
    for scattered in (None, True):
        res = movmodel.simulate_tracks(dirn, starts, (rows, cols), 1, 1., upd, pot, seed=11,
                                       track_id_base=77, use_table=True, thr=True,
                                       scattered=scattered, steps_per_launch=32)
        lens, ends, hist = _no_traj_result(res)
        assert np.array_equal(lens, ref['lengths']), (case, scattered)
        assert np.array_equal(ends, ref['ends']), (case, scattered)
        assert np.array_equal(hist, ref['hist']), (case, scattered)


def test_thr_table_large_batch_equals_f64_table(gpu):
    from ssrs_amd import movmodel
    rows, cols = 600, 900
    upd, pot = _random_field_case(rows, cols, 3)
    rng = np.random.default_rng(8)
    n = 20000
    starts = np.stack([rng.integers(1, 12, n), rng.integers(0, cols, n)], 1)
    for dirn in (0., 90., 45.):
        a = _no_traj_result(movmodel.simulate_tracks(dirn, starts, (rows, cols), 1, 1., upd, pot, seed=3,
                                                     use_table=True, ring=False))
        res = movmodel.simulate_tracks(dirn, starts, (rows, cols), 1, 1., upd, pot, seed=3, use_table=True, thr=True)
        assert movmodel.table_kind is not None
        b = _no_traj_result(res)
        for x, y in zip(a, b):
            assert np.array_equal(x, y), dirn
        assert int(b[2].sum()) == int(b[0].sum())


@pytest.mark.parametrize('dirn', [0., 180.])
def test_front_kernel_variants_give_the_oracles_integers(gpu, dirn):
    """North / south fronts on a raster at least 256 columns wide, through the variants of the front kernel
    behind their switches: SSRS_TRACKS_LDS_ROWS (table rows staged in LDS by a fifth wave with
    global_load_lds, read through a tag / entry / tag sequence; opt-in: measured slower than the gather,
    profiles/r03_notes.md section 7) and SSRS_TRACKS_NO_CHEAP_EXACT (near-ties straight to the full exact
    sequence instead of the seven-division decision).  Same lengths, ends and histogram as the C oracle."""
    import os
    from ssrs_amd import movmodel
    from oracle import c_oracle
    rows, cols = 700, 520
    upd, _ = _random_field_case(rows, cols, 52)
    rng = np.random.default_rng(21)
    rr = np.arange(rows, dtype=np.float64)[:, None]
    along = rr if dirn == 0. else (rows - 1 - rr)
    pot = (1000. * (1 - along / (rows - 1)) + rng.normal(0, 0.05, (rows, cols))).astype(np.float32)
    n = 12000
    t = rng.integers(2, 30, n)
    starts = np.stack([t if dirn == 0. else rows - 1 - t, rng.integers(0, cols, n)], 1)
    ref = c_oracle.simulate_tracks(dirn, starts, (rows, cols), 1, 1., upd, pot, seed=17, want_traj=False)
    for switch, value in ((None, ''), ('SSRS_TRACKS_LDS_ROWS', '1'), ('SSRS_TRACKS_LDS_ROWS', '2'),
                          ('SSRS_TRACKS_NO_CHEAP_EXACT', '1')):
        if switch:
            os.environ[switch] = value
        try:
            res = movmodel.simulate_tracks(dirn, starts, (rows, cols), 1, 1., upd, pot, seed=17, use_table=True,
                                           thr=True)
        finally:
            if switch:
                del os.environ[switch]
        lens, ends, hist = _no_traj_result(res)
        assert res.stats['window_launches'] > 0, switch
        assert np.array_equal(lens, ref['lengths']), (dirn, switch)
        assert np.array_equal(ends, ref['ends']), (dirn, switch)
        assert np.array_equal(hist, ref['hist']), (dirn, switch)


def test_thr_table_belongs_to_one_heading(gpu):
    from ssrs_amd import movmodel
    upd, pot = _random_field_case(40, 50, 1)
    table = movmodel.build_transition_table(upd, pot, thr=True, move_dirn=0.)
    assert movmodel.table_kind(table) == 'thr'
    with pytest.raises(ValueError):
        movmodel.simulate_tracks(90., [[5, 5]], (40, 50), 1, 1., upd, pot, table=table)
    with pytest.raises(ValueError):
        movmodel.build_transition_table(upd, pot, thr=True)
    # the kind and the heading travel IN the table (dtype / header in the guard band), not in Python
    # attributes: a clone is still a threshold table of heading 0 and nothing else
    import torch
    copy = table.clone()
    assert movmodel.table_kind(copy, 40, 50) == 'thr'
    ref = movmodel.simulate_tracks(0., [[5, 5], [7, 20]], (40, 50), 1, 1., upd, pot, table=table, seed=4)
    got = movmodel.simulate_tracks(0., [[5, 5], [7, 20]], (40, 50), 1, 1., upd, pot, table=copy, seed=4)
    assert torch.equal(ref.lengths, got.lengths) and torch.equal(ref.hist, got.hist)
    with pytest.raises(ValueError, match='threshold table'):
        movmodel.simulate_tracks(90., [[5, 5]], (40, 50), 1, 1., upd, pot, table=copy)
    with pytest.raises(ValueError):           # the dwords seen as floats are not a ring table of this raster
        movmodel.simulate_tracks(0., [[5, 5]], (40, 50), 1, 1., upd, pot, table=copy.view(torch.float32))
    ring = movmodel.build_transition_table(upd, pot, ring=True)
    with pytest.raises(ValueError):           # nor is a ring table a threshold table
        movmodel.simulate_tracks(0., [[5, 5]], (40, 50), 1, 1., upd, pot, table=ring.clone().view(torch.int32))
    with pytest.raises(ValueError):           # a table of another raster
        movmodel.simulate_tracks(0., [[5, 5]], (40, 48), 1, 1., np.ascontiguousarray(upd[:, :48]),
                                 np.ascontiguousarray(pot[:, :48]), table=copy)


def test_one_band_batch_is_redealt_over_the_lists(gpu):
    """All tracks start in one eighth of the raster's width: seven of the eight per-XCD lists are
    empty, the host re-deals the live tracks (k_rebalance_lists pseudo-launch) and grows the
    launches to the visit buffer's room; lengths, end cells, histogram and recorded trajectories
    must not notice (SSRS_TRACKS_NO_REBALANCE / _FIXED_STEPS are the A/B switches)."""
    import os
    from ssrs_amd import movmodel
    from oracle import c_oracle
    rows, cols = 420, 900
    upd, pot = _random_field_case(rows, cols, 12)
    pot = pot.copy()
    rr, cc = np.arange(rows)[:, None], np.arange(cols)[None, :]
    for r0, c0 in ((150, 60), (260, 95), (330, 40)):          # wells: detours, a few tracks circle until max_moves
        pot -= (600. * np.exp(-((rr - r0) ** 2 + (cc - c0) ** 2) / (2. * 9. ** 2))).astype(np.float32)
    rng = np.random.default_rng(21)
    n = 12000
    starts = np.stack([rng.integers(2, 30, n), rng.integers(5, 110, n)], 1)
    cap = 6000
    ref = c_oracle.simulate_tracks(0., starts, (rows, cols), 1, 1., upd, pot, seed=8, max_moves=cap)
    got = movmodel.simulate_tracks(0., starts, (rows, cols), 1, 1., upd, pot, seed=8, use_table=True, thr=True,
                                   max_moves=cap, steps_per_launch=64)
    assert np.array_equal(got.lengths.cpu().numpy(), ref['lengths'])
    assert np.array_equal(got.ends.cpu().numpy(), ref['ends'])
    assert np.array_equal(got.hist.cpu().numpy().view(np.uint32), ref['hist'])
    rec = movmodel.simulate_tracks(0., starts, (rows, cols), 1, 1., upd, pot, seed=8, use_table=True, thr=True,
                                   max_moves=cap, steps_per_launch=64, want_tracks=True)
    assert rec.stats['recorded']
    assert np.array_equal(rec.traj.cpu().numpy(), np.concatenate(ref['tracks']))
    assert torch.equal(rec.hist, got.hist)
    os.environ['SSRS_TRACKS_NO_REBALANCE'] = '1'
    os.environ['SSRS_TRACKS_FIXED_STEPS'] = '1'
    try:
        plain = movmodel.simulate_tracks(0., starts, (rows, cols), 1, 1., upd, pot, seed=8, use_table=True, thr=True,
                                         max_moves=cap, steps_per_launch=64)
    finally:
        del os.environ['SSRS_TRACKS_NO_REBALANCE'], os.environ['SSRS_TRACKS_FIXED_STEPS']
    assert torch.equal(plain.hist, got.hist) and torch.equal(plain.lengths, got.lengths)
    assert plain.stats['launches'] > got.stats['launches'], 'the re-deal / grown launches did not happen'


def test_block_windows_for_batches_that_roam_basins(gpu):
    """Wells in the potential trap a good part of a 20k batch until max_moves: the threshold stepper
    leaves the row window, goes through tile buckets while the rest of the batch still travels, then
    sorts the survivors into histogram windows (k_wander_windows / k_deal_sorted) and counts them in
    LDS (k_step_thr<6>: tombstoned lists, reversal rows in the fast path).  Lengths, end cells and
    histogram are the oracle's; SSRS_TRACKS_NO_BLOCK_WINDOW / SSRS_TRACKS_NO_REV are the A/B switches."""
    import os
    from ssrs_amd import movmodel
    from oracle import c_oracle
    rows, cols = 700, 1100
    upd, pot = _random_field_case(rows, cols, 5)
    pot = pot.copy()
    rr, cc = np.arange(rows)[:, None], np.arange(cols)[None, :]
    for r0, c0, w in ((200, 300, 14.), (330, 820, 18.), (340, 330, 10.), (520, 600, 16.)):
        pot -= (700. * np.exp(-((rr - r0) ** 2 + (cc - c0) ** 2) / (2. * w ** 2))).astype(np.float32)
    rng = np.random.default_rng(77)
    n = 20000
    starts = np.stack([rng.integers(2, 12, n), rng.integers(5, cols - 5, n)], 1)
    cap = 9000
    ref = c_oracle.simulate_tracks(0., starts, (rows, cols), 1, 1., upd, pot, seed=4, max_moves=cap, want_traj=False)
    assert (ref['lengths'] - 1 >= cap).mean() > 0.05           # enough of them are trapped
    os.environ['SSRS_TRACKS_FIXED_STEPS'] = '1'               # launches stay 128 steps deep: enough batches to get there
    got = movmodel.simulate_tracks(0., starts, (rows, cols), 1, 1., upd, pot, seed=4, use_table=True, thr=True,
                                   max_moves=cap, steps_per_launch=128)
    assert got.stats['block_window_launches'] > 0 and got.stats['wander_sorts'] > 0, got.stats
    assert got.stats['roam_launches'] > 0, got.stats             # two moves per 64-byte roam-table entry (k_step_roam)
    assert np.array_equal(got.lengths.cpu().numpy(), ref['lengths'])
    assert np.array_equal(got.ends.cpu().numpy(), ref['ends'])
    assert np.array_equal(got.hist.cpu().numpy().view(np.uint32), ref['hist'])
    for switch in ('SSRS_TRACKS_NO_BLOCK_WINDOW', 'SSRS_TRACKS_NO_ROAM_TABLE', 'SSRS_TRACKS_DEAL_ROUND_ROBIN', 'SSRS_TRACKS_NO_REV'):
        os.environ[switch] = '1'
        try:
            other = movmodel.simulate_tracks(0., starts, (rows, cols), 1, 1., upd, pot, seed=4, use_table=True, thr=True,
                                             max_moves=cap, steps_per_launch=128)
        finally:
            del os.environ[switch]
        assert torch.equal(other.hist, got.hist) and torch.equal(other.lengths, got.lengths), switch
        if switch == 'SSRS_TRACKS_NO_ROAM_TABLE':
            assert other.stats['roam_launches'] == 0 and other.stats['block_window_launches'] > 0
    del os.environ['SSRS_TRACKS_FIXED_STEPS']
    assert other.stats['block_window_launches'] > 0            # (the last switch only changes the kernel variant)
    grown = movmodel.simulate_tracks(0., starts, (rows, cols), 1, 1., upd, pot, seed=4, use_table=True, thr=True,
                                     max_moves=cap, steps_per_launch=128)
    assert torch.equal(grown.hist, got.hist) and torch.equal(grown.lengths, got.lengths)


def test_roam_launches_while_tracks_still_wait_for_their_release(gpu):
    """Soak case 355781144 (tests/dev/soak_tracks.py): a 10 x 213 raster, 9 000 tracks, launches of 16 steps,
    `scattered=True` -- the pair-table kernel runs from the second launch on, while most tracks still wait for
    their release by the coherent schedule.  A wave that left such a launch early (the stop flag of k_step_roam)
    released its tracks one launch late, on the odd half of a Philox block: lengths off by one or two.  Waves
    with a waiting lane now run the launch out."""
    import os
    from ssrs_amd import movmodel
    from oracle import c_oracle
    rng = np.random.default_rng(355781144)
    rows, cols = int(rng.integers(5, 400)), int(rng.integers(5, 500))
    assert rng.random() >= 0.15
    n = int(rng.choice([1, 7, 64, 65, 300, 2000, 9000, 20000]))
    dirn = float(rng.choice([0., 45., 90., 135., 180., 225., 270., 315., rng.uniform(0, 360)]))
    kind = rng.choice(['rough', 'smooth', 'flat', 'speckle', 'nan', 'wells', 'scales'])
    assert (rows, cols, n, dirn, str(kind)) == (10, 213, 9000, 315., 'scales')
    upd = np.abs(rng.normal(0.8, 0.6, (rows, cols)))
    ramp = 1000. * (1 - np.arange(rows)[:, None] / max(rows - 1., 1.))
    pot = (ramp + rng.normal(0, rng.choice([0.01, 1.0, 30.0]), (rows, cols))).astype(np.float32)
    upd = upd * 10. ** rng.uniform(-9, 39, upd.shape)
    upd[rng.random((rows, cols)) < 0.01] = np.inf
    band = 10. ** rng.integers(-44, 8, rows // 8 + 1).astype(np.float64)
    pot = (pot.astype(np.float64) * np.repeat(band, 8)[:rows, None]).astype(np.float32)
    starts = np.stack([rng.integers(0, rows, n), rng.integers(0, cols, n)], 1)
    s = int(rng.integers(0, 2**31))
    ref = c_oracle.simulate_tracks(dirn, starts, (rows, cols), 1, 1., upd, pot, seed=s, want_traj=False)
    for switch in (None, 'SSRS_TRACKS_NO_ROAM_STOP'):
        if switch:
            os.environ[switch] = '1'
        try:
            res = movmodel.simulate_tracks(dirn, starts, (rows, cols), 1, 1., upd, pot, seed=s, steps_per_launch=16,
                                           use_table=True, thr=True, scattered=True)
        finally:
            if switch:
                del os.environ[switch]
        assert res.stats['roam_launches'] > 0, switch
        lens, ends, hist = _no_traj_result(res)
        assert np.array_equal(lens, ref['lengths']), switch
        assert np.array_equal(ends, ref['ends']), switch
        assert np.array_equal(hist, ref['hist']), switch


@pytest.mark.parametrize('dirn', [0., 180., 90., 315.])
def test_pair_table_launch_with_tracks_on_the_boundary_rows(gpu, dirn):
    """The one memory fault of round 3 (profiles/r03_notes.md section 2, found by the soak): k_step_roam issues the
    NEXT pair-table gather before it knows whether the lane steps at all, and a lane standing on a boundary cell
    (waiting for its release on row 0 / 1 / rows - 2 / rows - 1, or just finished there) formed an index up to two
    rows outside the raster.  The index is clamped into the table now (tracks.hip, `cb = min(cell_b, last_cell)`).
    Here every track STARTS on the outer two rings of a tiny raster, launches are 16 steps long and the batch is
    flagged scattered, so pair-table launches start while most tracks still stand on those cells."""
    from ssrs_amd import movmodel
    from oracle import c_oracle
    rows, cols = 9, 150
    rng = np.random.default_rng(int(dirn) + 5)
    upd = np.abs(rng.normal(0.8, 0.6, (rows, cols)))
    pot = (1000. * (1 - np.arange(rows)[:, None] / (rows - 1.)) + rng.normal(0, 30., (rows, cols))).astype(np.float32)
    n = 9000
    ring_r = rng.choice([0, 1, rows - 2, rows - 1], n)
    ring_c = rng.choice([0, 1, cols - 2, cols - 1], n)
    on_row = rng.random(n) < 0.7
    starts = np.stack([np.where(on_row, ring_r, rng.integers(0, rows, n)),
                       np.where(on_row, rng.integers(0, cols, n), ring_c)], 1)
    ref = c_oracle.simulate_tracks(dirn, starts, (rows, cols), 1, 1., upd, pot, seed=11, want_traj=False)
    res = movmodel.simulate_tracks(dirn, starts, (rows, cols), 1, 1., upd, pot, seed=11, steps_per_launch=16,
                                   use_table=True, thr=True, scattered=True)
    assert res.stats['roam_launches'] > 0, res.stats
    lens, ends, hist = _no_traj_result(res)
    assert np.array_equal(lens, ref['lengths'])
    assert np.array_equal(ends, ref['ends'])
    assert np.array_equal(hist, ref['hist'])


def test_block_windows_when_nearly_every_track_is_trapped(gpu):
    """8192 tracks (the lists have no spare slots), a wide trough that catches most of them: the
    padded deal of the wander sort does not fit and falls back to the dense one (blocks may then mix
    windows; their strays are global atomics), and a batch flagged SCATTERED starts in block windows
    before any sort.  Results are the oracle's either way."""
    import os
    from ssrs_amd import movmodel
    from oracle import c_oracle
    rows, cols = 300, 2200
    upd, pot = _random_field_case(rows, cols, 9)
    pot = pot.copy()
    rr, cc = np.arange(rows)[:, None], np.arange(cols)[None, :]
    for c0 in (300, 1100, 1900):                               # three pits side by side
        pot -= (900. * np.exp(-((rr - 120) ** 2 / (2. * 12. ** 2) + (cc - c0) ** 2 / (2. * 260. ** 2)))).astype(np.float32)
    rng = np.random.default_rng(3)
    n = 8192
    starts = np.stack([rng.integers(2, 10, n), rng.integers(5, cols - 5, n)], 1)
    cap = 5000
    ref = c_oracle.simulate_tracks(0., starts, (rows, cols), 1, 1., upd, pot, seed=6, max_moves=cap, want_traj=False)
    assert (ref['lengths'] - 1 >= cap).mean() > 0.5
    os.environ['SSRS_TRACKS_FIXED_STEPS'] = '1'
    try:
        got = movmodel.simulate_tracks(0., starts, (rows, cols), 1, 1., upd, pot, seed=6, use_table=True, thr=True,
                                       max_moves=cap, steps_per_launch=64)
        sc = movmodel.simulate_tracks(0., starts, (rows, cols), 1, 1., upd, pot, seed=6, use_table=True, thr=True,
                                      max_moves=cap, steps_per_launch=64, scattered=True)
    finally:
        del os.environ['SSRS_TRACKS_FIXED_STEPS']
    assert got.stats['block_window_launches'] > 0 and got.stats['wander_sorts'] > 0, got.stats
    assert sc.stats['block_window_launches'] > 0 and sc.stats['roam_launches'] > 0, sc.stats
    for res in (got, sc):
        assert np.array_equal(res.lengths.cpu().numpy(), ref['lengths'])
        assert np.array_equal(res.ends.cpu().numpy(), ref['ends'])
        assert np.array_equal(res.hist.cpu().numpy().view(np.uint32), ref['hist'])
    # the same with 512- / 1024-lane roaming blocks forced (whole groups of 2 / 4 list blocks per window; found by the soak:
    # 9 000 tracks made lists of 5 blocks, whose dense deal rounded up to 6 and ran past the list -- lists are whole groups
    # of 4 blocks since), on this batch and on one whose size is no multiple of anything
    rng = np.random.default_rng(4)
    n2 = 9000
    starts2 = np.stack([rng.integers(2, 10, n2), rng.integers(5, cols - 5, n2)], 1)
    ref2 = c_oracle.simulate_tracks(0., starts2, (rows, cols), 1, 1., upd, pot, seed=6, max_moves=cap, want_traj=False)
    for width in ('2', '4'):
        os.environ['SSRS_TRACKS_ROAM_WIDTH'] = width
        os.environ['SSRS_TRACKS_FIXED_STEPS'] = '1'
        try:
            for st_, rf_ in ((starts, ref), (starts2, ref2)):
                for scattered in (False, True):
                    res = movmodel.simulate_tracks(0., st_, (rows, cols), 1, 1., upd, pot, seed=6, use_table=True, thr=True,
                                                   max_moves=cap, steps_per_launch=64, scattered=scattered)
                    assert res.stats['roam_wide_launches'] > 0, (width, scattered, res.stats)
                    assert np.array_equal(res.lengths.cpu().numpy(), rf_['lengths']), (width, scattered)
                    assert np.array_equal(res.ends.cpu().numpy(), rf_['ends']), (width, scattered)
                    assert np.array_equal(res.hist.cpu().numpy().view(np.uint32), rf_['hist']), (width, scattered)
        finally:
            del os.environ['SSRS_TRACKS_ROAM_WIDTH'], os.environ['SSRS_TRACKS_FIXED_STEPS']
