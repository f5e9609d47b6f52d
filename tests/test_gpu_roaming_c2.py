"""The roaming path at its real size: BASELINE configs[1] -- 5000 x 6000 @10 m, the
build's own K5 potential -- where 44 % of a batch reaches a basin of the field and circles there
until max_moves (/root/reference/ssrs/movmodel.py:285-317: `while k < max_moves` with the exit test
only at the raster's edge).  That regime runs in the block-window launches (the wander sort, windows
chosen from the data, tombstoned lists); until this file its parity tests ran on synthetic Gaussian
wells at 700 x 1100.  Here: 20 000 tracks against the C oracle on the SAME rasters with max_moves
capped on both sides (the oracle does 4-5e8 steps in seconds on the box's host threads), and an
uncapped batch checked through what needs no oracle."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
SHAPE, RES = (5000, 6000), 10.
CAP = 60_000


@pytest.fixture(scope='module')
def c2_field(gpu):
    """DEM -> K1 -> usable updraft -> K5 potential at the default tolerance, on the device and on the host."""
    from ssrs_amd import layers, movmodel
    from ssrs_amd.potential import solve_potential
    from ssrs_amd.synthetic import synthetic_dem
    dem = torch.from_numpy(synthetic_dem(SHAPE, RES)).cuda()
    _, upd = layers.updraft_from_dem(dem, RES, 10., 270., threshold=0.75)
    del dem
    pot, st = solve_potential(upd, 0., return_stats=True)
    assert st['converged'], st
    np.random.seed(30)
    r, c = movmodel.get_starting_indices(100_000, (5, 55, 1, 2), 'random', (60., 50.), RES)
    starts = np.stack([r, c], 1).astype(np.int32)
    return dict(upd=upd, pot=pot, upd_h=upd.cpu().numpy(), pot_h=pot.cpu().numpy(), starts=starts, shared={})


def test_roaming_batch_vs_oracle_at_c2(c2_field):
    from ssrs_amd import movmodel
    from oracle import c_oracle
    f = c2_field
    n = 20_000
    starts = f['starts'][:n]
    ref = c_oracle.simulate_tracks(0., starts, SHAPE, 1, 1., f['upd_h'], f['pot_h'], seed=30, want_traj=False,
                                   max_moves=CAP)
    got = movmodel.simulate_tracks(0., starts, SHAPE, 1, 1., f['upd'], f['pot'], seed=30, max_moves=CAP)
    st = got.stats
    # the batch really went through the roaming machinery
    assert st['block_window_launches'] > 0 and st['wander_sorts'] > 0 and st['roam_launches'] > 0, st
    lengths = got.lengths.cpu().numpy()
    assert np.array_equal(lengths, ref['lengths'])
    assert np.array_equal(got.ends.cpu().numpy(), ref['ends'])
    hist = got.hist.cpu().numpy().view(np.uint32)
    assert np.array_equal(hist, ref['hist'])
    assert int(hist.sum(dtype=np.uint64)) == st['total_steps'] + n == ref['steps'] + n
    at_cap = float(np.mean(lengths - 1 >= CAP))
    assert 0.25 < at_cap < 0.6, at_cap          # the basins hold 30-45 % of a batch on this field
    f['shared']['survivors'] = at_cap
    # the same batch through the tile buckets only / through round 2's one-gather-per-move kernel
    # (A/B switches of the block windows and of the roam table): same integers
    import os
    for switch in ('SSRS_TRACKS_NO_BLOCK_WINDOW', 'SSRS_TRACKS_NO_ROAM_TABLE'):
        os.environ[switch] = '1'
        try:
            alt = movmodel.simulate_tracks(0., starts, SHAPE, 1, 1., f['upd'], f['pot'], seed=30, max_moves=CAP)
        finally:
            del os.environ[switch]
        assert alt.stats['roam_launches'] == 0
        assert (alt.stats['block_window_launches'] == 0) == (switch == 'SSRS_TRACKS_NO_BLOCK_WINDOW')
        assert torch.equal(alt.lengths, got.lengths) and torch.equal(alt.hist, got.hist), switch


def test_uncapped_batch_properties_at_c2(c2_field):
    """No oracle can follow 7.5e6 moves per track; what holds without one: the histogram counts every
    point once, tracks end on the raster's edge or at max_moves exactly, every track that finished under
    the cap has the same length uncapped, and the share that wanders to max_moves is bounded by the
    capped run's share of survivors (measured here: 44 % alive after 60 000 moves, 30 % at max_moves --
    a third of the tracks in the basins find a way out within 7.5e6 moves)."""
    from ssrs_amd import movmodel
    f = c2_field
    n = 2_000
    starts = f['starts'][:n]
    got = movmodel.simulate_tracks(0., starts, SHAPE, 1, 1., f['upd'], f['pot'], seed=30)
    lengths = got.lengths.cpu().numpy().astype(np.int64)
    ends = got.ends.cpu().numpy()
    mm = SHAPE[0] // 2 * (SHAPE[1] // 2)
    hist = got.hist.cpu().numpy().view(np.uint32)
    assert int(hist.sum(dtype=np.uint64)) == got.stats['total_steps'] + n == int((lengths - 1).sum()) + n
    assert lengths.max() == mm + 1
    on_edge = (ends[:, 0] == 0) | (ends[:, 0] == SHAPE[0] - 1) | (ends[:, 1] == 0) | (ends[:, 1] == SHAPE[1] - 1)
    assert np.all(on_edge | (lengths == mm + 1))
    share = float(np.mean(lengths == mm + 1))
    # the first 2 000 tracks of the capped batch are these very tracks
    capped = movmodel.simulate_tracks(0., starts, SHAPE, 1, 1., f['upd'], f['pot'], seed=30, max_moves=CAP)
    cl = capped.lengths.cpu().numpy().astype(np.int64)
    survivors = float(np.mean(cl - 1 >= CAP))
    assert 0.2 < share <= survivors < 0.6, (share, survivors)
    # and every track that finished under the cap has the same length uncapped
    done = cl - 1 < CAP
    assert np.array_equal(cl[done], lengths[done])


def test_uncapped_roaming_tracks_vs_oracle_at_c2(c2_field):
    """The regime the headline spends 99 % of its time in, against the oracle WITHOUT a cap (VERDICT r3 item 1;
    /root/reference/ssrs/movmodel.py:285-317): the 20 000-track batch runs to max_moves = 7.5e6 on the GPU --
    ~130 pair-table launches of 65 536 steps, the stop flag, the periodic re-deal -- and eight contiguous id
    ranges of 64 tracks (finishers, tracks that leave a basin after 1e5..7e6 moves, tracks at max_moves) are
    stepped uncapped by the C oracle under their global ids (~1.3e9 steps): lengths and end cells must be equal.
    Then the same batch with the re-deal after EVERY settled batch and with none at all, and with 512- / 1024-lane blocks:
    identical integers."""
    import os
    from ssrs_amd import movmodel
    from oracle import c_oracle
    f = c2_field
    n = 20_000
    starts = f['starts'][:n]
    mm = SHAPE[0] // 2 * (SHAPE[1] // 2)
    got = movmodel.simulate_tracks(0., starts, SHAPE, 1, 1., f['upd'], f['pot'], seed=30)
    st = got.stats
    assert st['roam_launches'] > 50 and st['roam_shuffles'] > 0, st
    lengths = got.lengths.cpu().numpy()
    ends = got.ends.cpu().numpy()
    ids = np.concatenate([np.arange(b, b + 64) for b in range(137, n, n // 8)][:8])
    ref = c_oracle.simulate_tracks(0., starts[ids], SHAPE, 1, 1., f['upd_h'], f['pot_h'], seed=30, want_traj=False,
                                   want_hist=False, track_ids=ids)
    sub = lengths[ids].astype(np.int64) - 1
    late = int(np.sum((sub > CAP) & (sub < mm)))
    assert int(np.sum(sub == mm)) >= 64 and late >= 16, (int(np.sum(sub == mm)), late)     # the sample holds the regime
    assert np.array_equal(lengths[ids], ref['lengths'])
    assert np.array_equal(ends[ids], ref['ends'])
    assert ref['steps'] > 1.0e9
    for shuffle in ('1', '0'):
        os.environ['SSRS_TRACKS_ROAM_SHUFFLE'] = shuffle
        try:
            alt = movmodel.simulate_tracks(0., starts, SHAPE, 1, 1., f['upd'], f['pot'], seed=30)
        finally:
            del os.environ['SSRS_TRACKS_ROAM_SHUFFLE']
        assert (alt.stats['roam_shuffles'] > st['roam_shuffles']) if shuffle == '1' else (alt.stats['roam_shuffles'] == 0), \
            (shuffle, alt.stats['roam_shuffles'], st['roam_shuffles'])
        assert torch.equal(alt.lengths, got.lengths) and torch.equal(alt.ends, got.ends) and torch.equal(alt.hist, got.hist), shuffle
    # 512- and 1024-lane blocks (two / four list blocks of one window per CU: what a batch with more survivors than one
    # round of 256-lane blocks gets, forced here): identical integers
    for width in ('2', '4'):
        os.environ['SSRS_TRACKS_ROAM_WIDTH'] = width
        try:
            alt = movmodel.simulate_tracks(0., starts, SHAPE, 1, 1., f['upd'], f['pot'], seed=30)
        finally:
            del os.environ['SSRS_TRACKS_ROAM_WIDTH']
        assert alt.stats['roam_wide_launches'] > 50 and st['roam_wide_launches'] == 0, (width, alt.stats)
        assert torch.equal(alt.lengths, got.lengths) and torch.equal(alt.ends, got.ends) and torch.equal(alt.hist, got.hist), width


def test_config2_share_125k_tracks_on_the_solved_field(c2_field):
    """One GPU's share of BASELINE configs[2] (1 M tracks over 8 GPUs = 125 000 per GPU, ids 0..124 999 = rank 0's
    shard) on K5's field at full size, uncapped.  No oracle follows 3.3e11 steps; what holds at this size: the
    histogram counts every point once, every track ends on the raster's edge or at max_moves exactly, and the
    tracks of the capped oracle run that FINISHED under the cap (4 000-track prefix, the host's threads, seconds)
    have the same length and end cell here."""
    from ssrs_amd import movmodel
    from oracle import c_oracle
    f = c2_field
    n = 125_000
    np.random.seed(30)
    r, c = movmodel.get_starting_indices(1_000_000, (5, 55, 1, 2), 'random', (60., 50.), RES)
    starts = np.stack([r, c], 1).astype(np.int32)[:n]
    got = movmodel.simulate_tracks(0., starts, SHAPE, 1, 1., f['upd'], f['pot'], seed=30)
    lengths = got.lengths.cpu().numpy().astype(np.int64)
    ends = got.ends.cpu().numpy()
    mm = SHAPE[0] // 2 * (SHAPE[1] // 2)
    hist = got.hist.cpu().numpy().view(np.uint32)
    assert int(hist.sum(dtype=np.uint64)) == got.stats['total_steps'] + n == int(lengths.sum())
    assert got.stats['roam_launches'] > 50 and got.stats['total_steps'] > 2.5e11
    on_edge = (ends[:, 0] == 0) | (ends[:, 0] == SHAPE[0] - 1) | (ends[:, 1] == 0) | (ends[:, 1] == SHAPE[1] - 1)
    assert np.all(on_edge | (lengths == mm + 1))
    assert 0.2 < float(np.mean(lengths == mm + 1)) < 0.45
    m = 4_000
    ref = c_oracle.simulate_tracks(0., starts[:m], SHAPE, 1, 1., f['upd_h'], f['pot_h'], seed=30, want_traj=False,
                                   want_hist=False, max_moves=CAP)
    done = ref['lengths'].astype(np.int64) - 1 < CAP
    assert 0.4 < float(np.mean(done)) < 0.75
    assert np.array_equal(lengths[:m][done], ref['lengths'][done])
    assert np.array_equal(ends[:m][done], ref['ends'][done])
    assert np.all(lengths[:m][~done] - 1 >= CAP)


def test_quarter_million_tracks_start_wide_and_finish_narrow(c2_field):
    """One call of 250 000 tracks on K5's field at full size, as `Simulator` cuts its sub-batches (Config.hist_safe_tracks): more
    tracks roam at first than one round of 256-lane blocks holds, so the deals make 512-lane blocks until ~63 000 are left and
    256-lane blocks after (tracks.hip: roam_width, chosen from the live count k_wander_windows reads out).  Against the same
    call with 256-lane blocks only (SSRS_TRACKS_ROAM_WIDE=0, two rounds of blocks): identical lengths, end cells and 64-bit
    counts; the counts hold every point once, and a trap cell is past 2^32 - 1 or close to it (why the counts are 64-bit:
    ssrs_tracks_simulate_h64)."""
    import os
    from ssrs_amd import movmodel
    f = c2_field
    n = 250_000
    np.random.seed(30)
    r, c = movmodel.get_starting_indices(n, (5, 55, 1, 2), 'random', (60., 50.), RES)
    starts = np.stack([r, c], 1).astype(np.int32)
    got = movmodel.simulate_tracks(0., starts, SHAPE, 1, 1., f['upd'], f['pot'], seed=30, hist64=True)
    st = got.stats
    assert 0 < st['roam_wide_launches'] < st['roam_launches'], st
    assert got.hist.dtype == torch.int64
    assert int(got.hist.sum().item()) == st['total_steps'] + n == int(got.lengths.sum(dtype=torch.int64).item())
    assert int(got.hist.max().item()) > 3.5e9
    os.environ['SSRS_TRACKS_ROAM_WIDE'] = '0'
    try:
        alt = movmodel.simulate_tracks(0., starts, SHAPE, 1, 1., f['upd'], f['pot'], seed=30, hist64=True)
    finally:
        del os.environ['SSRS_TRACKS_ROAM_WIDE']
    assert alt.stats['roam_wide_launches'] == 0
    assert torch.equal(alt.lengths, got.lengths) and torch.equal(alt.ends, got.ends) and torch.equal(alt.hist, got.hist)
