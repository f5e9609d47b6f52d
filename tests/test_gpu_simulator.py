"""Drop-in boundary: ssrs_amd.Config / Simulator driven like the reference's
examples (/root/reference/examples/example_jem.py:43-57) with injected terrain,
checked against the oracle pipeline on the same inputs."""
import os
import pickle
from dataclasses import replace

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def make_config(tmp_path, **kw):
    from ssrs_amd import Config
    base = Config(run_name='t', out_dir=str(tmp_path), sim_seed=30, region_width_km=(8., 6.),
                  resolution=100., track_count=200, track_start_region=(1, 7, 0.2, 0.6),
                  track_direction=0.)
    return replace(base, **kw)


def test_uniform_fluidflow_end_to_end(gpu, tmp_path):
    from ssrs_amd import Simulator
    from oracle import ssrs_oracle as orc, c_oracle
    cfg = make_config(tmp_path)
    sim = Simulator(cfg, terrain='synthetic')
    assert sim.gridsize == (60, 80) and sim.case_ids == ['s10d270']
    oro_file = os.path.join(sim.mode_data_dir, 's10d270_orograph.npy')
    oro = np.load(oro_file)
    assert oro.dtype == np.float32 and oro.shape == (60, 80)
    dem = sim.get_terrain_elevation()
    ref_oro = orc.compute_orographic_updraft(
        10., 270., orc.compute_slope_degrees(dem, 100.), orc.compute_aspect_degrees(dem, 100.))
    d = np.abs(oro.view(np.int32).astype(np.int64) - ref_oro.astype(np.float32).view(np.int32))
    assert d.max() <= 1
    upd = sim.load_updrafts('s10d270')[0]
    assert upd.dtype == np.float64
    np.testing.assert_allclose(upd, orc.get_above_threshold_speed(oro, 0.75), rtol=1e-12, atol=1e-15)

    sim.simulate_tracks()
    pot_file = os.path.join(sim.mode_data_dir, 's10d270_d0_t75_fluidflow_r0_potential.npy')
    trk_file = os.path.join(sim.mode_data_dir, 's10d270_d0_t75_fluidflow_r0_tracks.pkl')
    pot = np.load(pot_file)
    assert pot.dtype == np.float32
    np.testing.assert_allclose(pot, orc.solve_potential(upd, 0.), rtol=0, atol=1e-3)
    with open(trk_file, 'rb') as f:
        tracks = pickle.load(f)
    assert len(tracks) == 200 and tracks[0].dtype == np.int16 and tracks[0].shape[1] == 2
    # same start cells as the reference would draw, same tracks as the oracle
    np.random.seed(30)
    r, c = orc.get_starting_indices(200, (1, 7, 0.2, 0.6), 'random', (8., 6.), 100.)
    ref = c_oracle.simulate_tracks(0., np.stack([r, c], 1), (60, 80), 1, 1., upd, pot, seed=30)
    for a, b in zip(tracks, ref['tracks']):
        assert np.array_equal(a, b)

    out = sim.compute_presence_map(radius=300.)
    assert out.dtype == np.float32 and out.max() == 1.0
    saved = np.load(os.path.join(sim.mode_data_dir, 'summary_presence.npy'))
    assert np.array_equal(saved, out)
    sm = orc.smooth_presence_from_counts(ref['hist'].astype(np.int64), 3)
    sm = sm / sm.max()
    np.testing.assert_allclose(out, sm / sm.max(), rtol=1e-6, atol=1e-7)
    sim.plot_presence_map(radius=300.)      # also via the reference's method name
    # second call finds the cached potential (file contract); like the
    # reference it draws NEW start cells from the advanced numpy stream
    sim.simulate_tracks()
    sim.plot_simulated_tracks()             # stub, must not raise


def test_drw_and_structured_starts(gpu, tmp_path):
    from ssrs_amd import Simulator
    from oracle import ssrs_oracle as orc, c_oracle
    cfg = make_config(tmp_path, movement_model='drw', track_start_type='structured',
                      track_count=77, track_direction=40., track_dirn_restrict=3,
                      run_name='drw')
    sim = Simulator(cfg, terrain='synthetic')
    sim.simulate_tracks()
    with open(os.path.join(sim.mode_data_dir, 's10d270_d40_t75_drw_r0_tracks.pkl'), 'rb') as f:
        tracks = pickle.load(f)
    r, c = orc.get_starting_indices(77, (1, 7, 0.2, 0.6), 'structured', (8., 6.), 100.)
    ref = c_oracle.simulate_tracks(40., np.stack([r, c], 1), (60, 80), 3, 1., None, None, seed=30)
    assert len(tracks) == 77
    for a, b in zip(tracks, ref['tracks']):
        assert np.array_equal(a, b)


def test_snapshot_mode_with_injected_wind_rasters(gpu, tmp_path):
    from ssrs_amd import Simulator
    from oracle import ssrs_oracle as orc
    cfg = make_config(tmp_path, sim_mode='snapshot', run_name='snap', track_count=50)
    rows, cols = 60, 80
    rr, cc = np.mgrid[0:rows, 0:cols]
    ws = 8. + 3. * np.sin(cc / 17.) * np.cos(rr / 13.)
    wd = 270. + 40. * np.sin(cc / 23. + rr / 31.)
    sim = Simulator(cfg, terrain='synthetic',
                    wind=[dict(datetime=(2010, 6, 17, 13), wspeed=ws, wdirn=wd)])
    assert sim.case_ids == ['y2010m06d17h13']
    oro = np.load(os.path.join(sim.mode_data_dir, 'y2010m06d17h13_orograph.npy'))
    dem = sim.get_terrain_elevation()
    ref = orc.compute_orographic_updraft(ws, wd, orc.compute_slope_degrees(dem, 100.),
                                         orc.compute_aspect_degrees(dem, 100.)).astype(np.float32)
    d = np.abs(oro.view(np.int32).astype(np.int64) - ref.view(np.int32))
    assert d.max() <= 1
    sim.simulate_tracks()
    assert os.path.exists(os.path.join(
        sim.mode_data_dir, 'y2010m06d17h13_d0_t75_fluidflow_r0_tracks.pkl'))


def test_constructor_errors(tmp_path):
    from ssrs_amd import Simulator
    with pytest.raises(NotImplementedError):
        Simulator(make_config(tmp_path, run_name='e1'))
    with pytest.raises(ValueError):
        Simulator(make_config(tmp_path, run_name='e2'), terrain=np.zeros((3, 3)))


def test_seasonal_mode_lattice_wind_and_thermals(gpu, tmp_path):
    """Seasonal mode: several injected WTK-shaped lattice snapshots (K6) batched
    through K1, plus one thermal realisation per case (a5): file contract of
    simulator.py:200-228 and the [orograph] + [orograph + thermals] updraft list."""
    from ssrs_amd import Simulator
    from ssrs_amd.synthetic import wind_lattice
    from ssrs_amd.wind import interpolate_wind_lattice
    from oracle import ssrs_oracle as orc
    cfg = make_config(tmp_path, sim_mode='seasonal', run_name='seas', track_count=40,
                      thermals_realization_count=1, movement_model='drw')
    wind = []
    for s in range(3):
        x, y, ws, wd = wind_lattice((8., 6.), 2.0, phase=2 * np.pi * s / 3)
        wind.append(dict(datetime=(2010, 4, 1 + s, 12), x_km=x, y_km=y, wspeed=ws, wdirn=wd))
    sim = Simulator(cfg, terrain='synthetic', wind=wind)
    assert sim.case_ids == ['y2010m04d01h12', 'y2010m04d02h12', 'y2010m04d03h12']
    dem = sim.get_terrain_elevation()
    slope, aspect = orc.compute_slope_degrees(dem, 100.), orc.compute_aspect_degrees(dem, 100.)
    for item, cid in zip(wind, sim.case_ids):
        s, d = interpolate_wind_lattice(item['x_km'], item['y_km'], item['wspeed'], item['wdirn'],
                                        (60, 80), 100.)
        ref = orc.compute_orographic_updraft(s.cpu().numpy(), d.cpu().numpy(), slope, aspect)
        oro = np.load(os.path.join(sim.mode_data_dir, f'{cid}_orograph.npy'))
        dd = np.abs(oro.view(np.int32).astype(np.int64) - ref.astype(np.float32).view(np.int32))
        assert dd.max() <= 1
        th = np.load(os.path.join(sim.mode_data_dir, f'{cid}_r0_thermals.npy'))
        assert th.dtype == np.float32 and th.shape == (60, 80) and th.min() >= 0
    ups = sim.load_updrafts(sim.case_ids[0])
    assert len(ups) == 2 and ups[0].dtype == np.float64
    sim.simulate_tracks()                       # 3 cases x 2 realisations
    for cid in sim.case_ids:
        for real in (0, 1):
            assert os.path.exists(os.path.join(sim.mode_data_dir, f'{cid}_d0_t75_drw_r{real}_tracks.pkl'))
    out = sim.compute_presence_map(radius=300.)
    assert out.shape == (60, 80) and out.max() == 1.0


def test_presence_map_from_saved_tracks_matches_in_memory(gpu, tmp_path):
    """plot_presence_map in a fresh process state reads <id>_tracks.pkl like the
    reference (simulator.py:525-529); the map must equal the one computed from
    the device histogram kept by simulate_tracks."""
    from ssrs_amd import Simulator
    cfg = make_config(tmp_path, run_name='pm', track_count=150)
    sim = Simulator(cfg, terrain='synthetic')
    sim.simulate_tracks()
    first = sim.compute_presence_map(radius=400.)
    sim2 = Simulator(cfg, terrain='synthetic')      # same out_dir: finds orograph/potential/tracks
    second = sim2.compute_presence_map(radius=400.)
    assert np.array_equal(first, second)
