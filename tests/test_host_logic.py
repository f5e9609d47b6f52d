"""Host-side logic of the product package (no GPU, no oracle in the product
path): start cells, directional prior, Dirichlet sets, Config surface,
sharding arithmetic, synthetic inputs."""
import dataclasses
import re

import numpy as np
import pytest

from ssrs_amd import movmodel, potential
from ssrs_amd.config import Config
from ssrs_amd.distributed import shard_range
from ssrs_amd.synthetic import synthetic_dem, wind_lattice, ramp_potential

REF_FIELDS = [  # /root/reference/ssrs/config.py:14-67, in order, with defaults
    ('run_name', 'default'), ('out_dir', None), ('max_cores', 8), ('sim_seed', -1),
    ('sim_mode', 'uniform'), ('print_verbose', False), ('southwest_lonlat', (-106.21, 42.78)),
    ('projected_crs', 'ESRI:102008'), ('region_width_km', (60., 50.)), ('resolution', 100.),
    ('uniform_winddirn', 270.), ('uniform_windspeed', 10.),
    ('snapshot_datetime', (2010, 6, 17, 13)), ('seasonal_start', (3, 20)),
    ('seasonal_end', (5, 15)), ('seasonal_timeofday', 'daytime'), ('seasonal_count', 8),
    ('wtk_source', 'AWS'), ('wtk_orographic_height', 100), ('wtk_thermal_height', 100),
    ('wtk_interp_type', 'linear'), ('thermals_realization_count', 0),
    ('updraft_threshold', 0.75), ('movement_model', 'fluidflow'), ('track_direction', 0),
    ('track_count', 1000), ('track_start_region', (5, 55, 1, 2)),
    ('track_start_type', 'random'), ('track_stochastic_nu', 1.), ('track_dirn_restrict', 1),
    ('turbine_minimum_hubheight', 50.), ('turbine_mrkr_size', 3.), ('fig_height', 6.),
    ('fig_dpi', 200),
]


def test_config_is_field_compatible_with_reference():
    fields = dataclasses.fields(Config)
    names = [f.name for f in fields]
    assert names[:len(REF_FIELDS)] == [n for n, _ in REF_FIELDS]
    cfg = Config()
    for name, default in REF_FIELDS:
        if default is not None:
            assert getattr(cfg, name) == default, name
    assert Config.turbine_mrkr_styles == ('1k', '2k', '3k', '4k', '+k', 'xk', '*k', '.k', 'ok')
    assert 'turbine_mrkr_styles' not in names            # class attribute, not a field
    c2 = dataclasses.replace(cfg, sim_mode='snapshot', track_count=5)
    assert c2.sim_mode == 'snapshot' and c2.track_count == 5
    text = str(cfg)
    assert ':::: General settings' in text and 'track_count = 1000' in text
    assert Config(**dataclasses.asdict(cfg)) == cfg       # Simulator's copy path


def test_starting_indices_vs_golden(golden):
    g = golden('g4_starts.npz')
    np.random.seed(30)
    r, c = movmodel.get_starting_indices(1000, (5, 55, 1, 2), 'random', (60., 50.), 100.)
    assert np.array_equal(r, g['rand_rows']) and np.array_equal(c, g['rand_cols'])
    for n in (5, 1000, 6000, 5151, 12000):
        r, c = movmodel.get_starting_indices(n, (5, 55, 1, 2), 'structured', (60., 50.), 100.)
        assert np.array_equal(r, g[f'struct{n}_rows']) and np.array_equal(c, g[f'struct{n}_cols'])
    np.random.seed(31)
    r, c = movmodel.get_starting_indices(64, (0, 60, 0, 0.5), 'random', (60., 50.), 10.)
    assert np.array_equal(r, g['edge_rows']) and np.array_equal(c, g['edge_cols'])
    with pytest.raises(ValueError, match='incompatible'):
        movmodel.get_starting_indices(5, (5, 65, 1, 2), 'random', (60., 50.), 100.)
    with pytest.raises(ValueError, match='Invalid sim_start_type'):
        movmodel.get_starting_indices(5, (5, 55, 1, 2), 'bogus', (60., 50.), 100.)


def test_directional_prior_and_constants_vs_golden(golden):
    g = golden('g1_constants.npz')
    for t, p in zip(g['thetas_deg'], g['priors']):
        assert np.array_equal(movmodel.get_directional_probs(t * np.pi / 180.), p)
    assert np.array_equal(movmodel.neighbour_delta_norms_inv, g['norms_inv'])
    assert np.array_equal(np.array(movmodel.neighbour_deltas), g['deltas'])


def test_boundary_nodes_vs_golden(golden):
    g = golden('g5_potential.npz')
    for dirn in (0., 180., -45., 90., 30.):
        tag = f'd{int(dirn % 360)}'
        bn, be = potential.get_boundary_nodes(dirn, (48, 64))
        assert np.array_equal(bn, g[f'bnodes_{tag}']) and np.array_equal(be, g[f'benergy_{tag}'])
        mask, vals = potential.dirichlet_rasters(dirn, (48, 64))
        assert mask.sum() == np.unique(bn).size
        assert set(np.unique(vals[mask == 1])) <= {0., 1000.}
    bn, be = movmodel.MovModel(0., (48, 64)).get_boundary_nodes()
    assert np.array_equal(bn, g['bnodes_d0'])


def test_shard_range_partitions_exactly():
    for n, w in [(100000, 1), (1000000, 8), (10, 3), (7, 8), (0, 4)]:
        parts = [shard_range(n, r, w) for r in range(w)]
        assert parts[0][0] == 0 and parts[-1][1] == n
        assert all(a[1] == b[0] for a, b in zip(parts, parts[1:]))
        sizes = [hi - lo for lo, hi in parts]
        assert max(sizes) - min(sizes) <= 1


def test_synthetic_inputs_are_deterministic():
    a = synthetic_dem((50, 60), 100.)
    b = synthetic_dem((50, 60), 100.)
    assert np.array_equal(a, b) and a.dtype == np.float64 and a.shape == (50, 60)
    assert abs(a.mean() - 1800.) < 250.
    x, y, ws, wd = wind_lattice((60., 50.))
    assert x.size == 31 and y.size == 26 and ws.shape == (26, 31)
    p = ramp_potential((5000, 6000))
    assert p.dtype == np.float32 and p[0, 0] == 1000. and p[-1, -1] == 0.
    assert (np.diff(p[:, 0].astype(np.float64)) < 0).all()


def test_product_never_imports_oracle():
    """The product path must not route through the oracle (or any CPU fallback)."""
    import os
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'ssrs_amd')
    pat = re.compile(r'^\s*(from|import)\s+(\.\.)?oracle\b|from\s+oracle\b', re.M)
    for dirpath, _, files in os.walk(root):
        for f in files:
            if f.endswith(('.py', '.hip', '.h')):
                text = open(os.path.join(dirpath, f)).read()
                assert not pat.search(text), f'{f} imports the oracle'
                assert 'liboracle' not in text, f'{f} references the oracle library'


def test_tools_never_import_oracle():
    """Development scripts that need the oracle live under tests/dev; tools/ is product-side."""
    import os
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tools')
    pat = re.compile(r'^\s*(from|import)\s+oracle\b', re.M)
    for f in os.listdir(root):
        if f.endswith('.py'):
            assert not pat.search(open(os.path.join(root, f)).read()), f'tools/{f} imports the oracle'


def test_ring_table_applies_only_to_the_reference_default_model():
    from ssrs_amd.movmodel import ring_table_applies
    assert ring_table_applies(1, 1.0, False, False, 0)
    assert ring_table_applies(1, 1.0, False, False, 64)
    assert not ring_table_applies(2, 1.0, False, False, 0)       # longer direction memory
    assert not ring_table_applies(0, 1.0, False, False, 0)       # whole-history restriction
    assert not ring_table_applies(1, 0.5, False, False, 0)       # nu != 1 needs pow()
    assert not ring_table_applies(1, 1.0, True, False, 0)        # trajectories: generic kernel
    assert not ring_table_applies(1, 1.0, False, True, 0)        # exact-only A/B
    assert not ring_table_applies(1, 1.0, False, False, 7)       # odd steps per launch


def test_step_case_cuts_equal_sub_batches():
    """More than hist_safe_tracks tracks of a case are stepped in EQUAL sub-batches (a pass lasts as long as its longest
    track chain, so a short last sub-batch would cost a full pass for a fraction of the work), ids stay contiguous, and the
    64-bit sum is taken whenever any rank's share needs it (simulator.py: _step_case)."""
    import torch
    from ssrs_amd import movmodel
    from ssrs_amd.simulator import Simulator
    sim = object.__new__(Simulator)
    sim.hist_safe_tracks, sim.track_direction, sim.gridsize = 140_000, 0., (4, 5)
    sim.track_dirn_restrict, sim.track_stochastic_nu, sim.save_tracks, sim.steps_per_launch = 1, 1., False, 0
    calls = []

    class Batch:
        def __init__(self, n):
            self.hist = torch.zeros((4, 5), dtype=torch.int32)
            self.hist[0, 0] = n
            self.lengths = torch.ones(n, dtype=torch.int32)
            self.ends = torch.zeros((n, 2), dtype=torch.int16)
            self.stats, self.total_points = dict(total_steps=0), n

    def fake(move_dirn, sub, gridsize, *a, track_id_base=0, **kw):
        calls.append((track_id_base, int(sub.shape[0])))
        return Batch(int(sub.shape[0]))
    real = movmodel.simulate_tracks
    movmodel.simulate_tracks = fake
    try:
        for n, want in ((1_000_000, [125_000] * 8), (140_000, [140_000]), (140_001, [70_001, 70_000]), (300_000, [100_000] * 3)):
            calls.clear()
            out = sim._step_case(torch.zeros((n, 2), dtype=torch.int32), 1000, (None, None), 30, None, widest_share=n)
            assert [c[1] for c in calls] == want, (n, calls)
            assert [c[0] for c in calls] == [1000 + sum(want[:k]) for k in range(len(want))]
            assert int(out.hist[0, 0]) == n and (out.hist.dtype == torch.int64) == (n > 140_000)
            assert out.lengths.numel() == n
    finally:
        movmodel.simulate_tracks = real


def test_step_case_splits_a_sub_batch_whose_uint32_counts_wrapped():
    """A sub-batch whose histogram checksum is short (a cell passed 2^32 - 1) is stepped again as two halves, in track-id
    order, with a warning, and everything is added up in 64 bits; below Simulator._MIN_SPLIT_TRACKS it is an error
    (simulator.py: _step_case)."""
    import torch
    from ssrs_amd import movmodel
    from ssrs_amd.distributed import HistogramOverflow
    from ssrs_amd.simulator import Simulator
    sim = object.__new__(Simulator)
    sim.hist_safe_tracks, sim.track_direction, sim.gridsize = 250_000, 0., (4, 5)
    sim.track_dirn_restrict, sim.track_stochastic_nu, sim.save_tracks, sim.steps_per_launch = 1, 1., False, 0
    calls = []
    limit = [30_000]

    class Batch:
        def __init__(self, n):
            self.hist = torch.zeros((4, 5), dtype=torch.int32)
            self.hist[0, 0] = n if n <= limit[0] else n - 7          # more than `limit` tracks: 7 visits "missing"
            self.lengths = torch.ones(n, dtype=torch.int32)
            self.ends = torch.zeros((n, 2), dtype=torch.int16)
            self.stats, self.total_points = dict(total_steps=0), n

    def fake(move_dirn, sub, gridsize, *a, track_id_base=0, **kw):
        calls.append((track_id_base, int(sub.shape[0])))
        return Batch(int(sub.shape[0]))
    real = movmodel.simulate_tracks
    movmodel.simulate_tracks = fake
    try:
        with pytest.warns(RuntimeWarning, match='wrapped'):
            out = sim._step_case(torch.zeros((100_000, 2), dtype=torch.int32), 500, (None, None), 30, None, widest_share=100_000)
        kept = [c for c in calls if c[1] <= limit[0]]
        assert [c[1] for c in kept] == [25_000] * 4 and [c[0] for c in kept] == [500, 25_500, 50_500, 75_500]
        assert [c[1] for c in calls if c[1] > limit[0]] == [100_000, 50_000, 50_000]
        assert out.hist.dtype == torch.int64 and int(out.hist[0, 0]) == 100_000 and out.lengths.numel() == 100_000
        limit[0] = 0                                                  # every batch wraps: ends in the error, not in a loop
        with pytest.warns(RuntimeWarning), pytest.raises(HistogramOverflow, match='missing'):
            sim._step_case(torch.zeros((1000, 2), dtype=torch.int32), 0, (None, None), 30, None, widest_share=1000)
    finally:
        movmodel.simulate_tracks = real


def test_bench_cuts_sub_batches_like_the_simulator():
    """bench.py --hist-safe-tracks defaults to Config.hist_safe_tracks, and both hand sub-batches of more than 100 000 tracks to
    the library's 64-bit counts (ssrs_tracks_simulate_h64)."""
    import os, re
    from ssrs_amd import Config
    from ssrs_amd.simulator import Simulator
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = open(os.path.join(root, 'bench.py')).read()
    m = re.search(r"--hist-safe-tracks', type=int, default=([\d_]+)", src)
    assert m and int(m.group(1).replace('_', '')) == Config().hist_safe_tracks
    m = re.search(r"in_lib = safe > ([\d_]+)", src)
    assert m and int(m.group(1).replace('_', '')) == Simulator._HIST64_FROM_TRACKS
