"""Trajectory output and the presence histogram at workloads where tracks wander to max_moves (VERDICT r2 item
6).  The reference keeps every trajectory as a host list and pickles it (/root/reference/ssrs/simulator.py:360-385)
and counts visits in int16 (movmodel.py:410-419, wraps at 32 767).  Here: trajectories too long for one device
tensor are stepped again range by range and streamed into the pickle; a `<id>_tracks.pkl` beyond
Config.max_tracks_file_gb is refused with its size BEFORE anything is allocated; a uint32 count that wrapped is
detected by the histogram's checksum, and batches beyond Config.hist_safe_tracks are added up in 64 bits."""
import os
import pickle
from dataclasses import replace

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _case():
    from test_gpu_tracks import _random_field_case
    rows, cols = 96, 128
    upd, pot = _random_field_case(rows, cols, 7)
    rng = np.random.default_rng(5)
    n = 300
    starts = np.stack([rng.integers(1, 12, n), rng.integers(0, cols, n)], 1)
    return (rows, cols), upd, pot, starts


def test_chunked_trajectories_equal_the_oracle(gpu):
    """A pool too small to record and a device budget too small for the trajectory tensor: iter_tracks() steps the
    batch again in ranges of a few tracks; every point equals the oracle's."""
    from ssrs_amd import movmodel
    from oracle import c_oracle
    shape, upd, pot, starts = _case()
    ref = c_oracle.simulate_tracks(0., starts, shape, 1, 1., upd, pot, seed=11)
    whole = movmodel.simulate_tracks(0., starts, shape, 1, 1., upd, pot, seed=11, want_tracks=True)
    assert whole.traj is not None
    got = movmodel.simulate_tracks(0., starts, shape, 1, 1., upd, pot, seed=11, want_tracks=True,
                                   record_pool_bytes=4096, traj_budget_bytes=16 * 1024)
    assert got.traj is None and not got.stats['recorded']           # neither recorded nor held in one tensor
    assert got.total_points == int(ref['lengths'].sum())
    tracks = list(got.iter_tracks())
    assert len(tracks) == len(ref['tracks'])
    for a, b in zip(tracks, ref['tracks']):
        assert a.dtype == np.int16 and np.array_equal(a, b)
    assert np.array_equal(got.hist.cpu().numpy().view(np.uint32), ref['hist'])      # counted once, in the first pass
    assert torch.equal(got.hist, whole.hist)
    with pytest.raises(MemoryError, match='GiB'):
        got.tracks(max_bytes=1000)
    # a track id offset (a shard of a larger batch) replays under the same stream keys
    sh = movmodel.simulate_tracks(0., starts[100:], shape, 1, 1., upd, pot, seed=11, track_id_base=100, want_tracks=True,
                                  record_pool_bytes=4096, traj_budget_bytes=16 * 1024)
    for a, b in zip(sh.iter_tracks(), ref['tracks'][100:]):
        assert np.array_equal(a, b)


def test_simulator_streams_the_pickle_and_refuses_absurd_sizes(gpu, tmp_path):
    from ssrs_amd import Config, Simulator
    base = Config(run_name='s', out_dir=str(tmp_path), sim_seed=30, region_width_km=(8., 6.), resolution=100.,
                  track_count=200, track_start_region=(1, 7, 0.2, 0.6), track_direction=0.)
    sim = Simulator(base, terrain='synthetic')
    sim.simulate_tracks()
    with open(os.path.join(sim.mode_data_dir, 's10d270_d0_t75_fluidflow_r0_tracks.pkl'), 'rb') as f:
        tracks = pickle.load(f)
    assert type(tracks) is list and len(tracks) == 200 and all(t.dtype == np.int16 and t.shape[1] == 2 for t in tracks)
    points = sum(len(t) for t in tracks)
    # the same run with a file limit below its size: a ValueError that names the size, no pickle written
    small = Simulator(replace(base, run_name='s2', max_tracks_file_gb=points * 4 / 2 ** 30 / 2), terrain='synthetic')
    with pytest.raises(ValueError, match='GiB'):
        small.simulate_tracks()
    assert not os.path.exists(os.path.join(small.mode_data_dir, 's10d270_d0_t75_fluidflow_r0_tracks.pkl'))


def test_single_rank_histogram_guard_and_64_bit_sub_batches(gpu, tmp_path):
    """A narrow, deep well in the potential (written into the <id>_potential.npy cache of the file contract,
    simulator.py:262-272) traps every track until max_moves = 2.25e6: 80 000 tracks put 1.8e11 visits onto
    ~180 cells, 2.7 % of them (4.8e9 > 2^32) onto the hottest.  One uint32 histogram wraps -- the checksum guard sees it, warns
    and steps the batch again as two halves; sub-batches of 20 000 tracks (Config.hist_safe_tracks) never wrap; both ways the
    counts are added up in 64 bits and add up to the points of the tracks, cell for cell the same."""
    from ssrs_amd import Config, Simulator
    rows = cols = 3000
    cfg = Config(run_name='w', out_dir=str(tmp_path), sim_seed=5, region_width_km=(30., 30.), resolution=10.,
                 track_count=80_000, track_start_region=(14.5, 15.5, 14., 14.6), track_direction=0., save_tracks=False)
    flat = np.zeros((rows, cols))
    sim = Simulator(cfg, terrain=flat)                       # flat terrain: the updraft is zero everywhere (weights 1e-6 floor)
    rr, cc = np.arange(rows, dtype=np.float64)[:, None], np.arange(cols, dtype=np.float64)[None, :]
    d2 = (rr - 1500.) ** 2 + (cc - 1500.) ** 2
    pot = (1000. * (1. - rr / (rows - 1.)) + 4.0 * np.sqrt(d2) - 3000. * np.exp(-d2 / (2. * 2.5 ** 2))).astype(np.float32)
    np.save(os.path.join(sim.mode_data_dir, 's10d270_d0_t75_fluidflow_r0_potential.npy'), pot)
    with pytest.warns(RuntimeWarning, match='wrapped'):
        sim.simulate_tracks()
    halves = sim._presence_counts[('s10d270', 0)]
    assert halves.dtype == torch.int64 and int(halves.max().item()) > 2 ** 32
    safe = Simulator(replace(cfg, run_name='w2', hist_safe_tracks=20_000), terrain=flat)
    np.save(os.path.join(safe.mode_data_dir, 's10d270_d0_t75_fluidflow_r0_potential.npy'), pot)
    safe.simulate_tracks()
    hist = safe._presence_counts[('s10d270', 0)]
    assert hist.dtype == torch.int64
    steps = sum(st['total_steps'] for st in [safe.last_stats[('s10d270', 0)]])
    assert int(hist.sum().item()) == steps + 80_000
    assert int(hist.max().item()) > 2 ** 32                  # the cell that wrapped in the single histogram
    assert torch.equal(hist, halves)
    # the same batch in ONE call with the counts in 64 bits inside the library (ssrs_tracks_simulate_h64: what sub-batches of
    # more than 100 000 tracks get; forced here): nothing wraps, nothing is split, the same counts
    import warnings
    lib64 = Simulator(replace(cfg, run_name='w3'), terrain=flat)
    lib64._HIST64_FROM_TRACKS = 10_000
    np.save(os.path.join(lib64.mode_data_dir, 's10d270_d0_t75_fluidflow_r0_potential.npy'), pot)
    with warnings.catch_warnings():
        warnings.simplefilter('error', RuntimeWarning)
        lib64.simulate_tracks()
    assert torch.equal(lib64._presence_counts[('s10d270', 0)], halves)
    out = safe.compute_presence_map(radius=100.)
    assert out.dtype == np.float32 and float(out.max()) == 1.0 and np.isfinite(out).all()
