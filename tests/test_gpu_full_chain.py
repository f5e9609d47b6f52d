"""north_star's parity sentence end to end at config C1: from the DEM
alone -- K1 -> K5 -> K2/K3 -> K4 with the BUILD'S OWN orograph and potential -- against G8,
the reference's run of the same configuration (/root/reference/ssrs/simulator.py:189-198,
230-243, 259-288, 332-386, 508-546)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _ulp(a, b):
    return np.abs(a.view(np.int32).astype(np.int64) - b.view(np.int32).astype(np.int64))


def test_c1_from_the_dem_with_the_builds_own_fields(gpu, golden):
    from ssrs_amd import layers, movmodel, presence
    from ssrs_amd.potential import solve_potential
    from ssrs_amd.synthetic import synthetic_dem
    g = golden('g8_c1.npz')
    shape, res = (500, 600), 100.
    dem = synthetic_dem(shape, res)
    # K1: DEM -> orograph f32 (+ usable updraft f64)
    oro, upd = layers.updraft_from_dem(dem, res, 10., 270., threshold=0.75)
    u = _ulp(np.asarray(oro), g['orograph_f32'])
    assert u.max() <= 1, f'orograph differs by {u.max()} f32 ulp'
    print(f'orograph: {np.mean(u == 0):.4f} of the cells bit-identical, the rest 1 ulp')
    # K5: usable updraft -> potential, library defaults; stated tolerance 1e-3 of 0..1000
    pot = solve_potential(upd, 0.)
    d = np.abs(pot.astype(np.float64) - g['potential'].astype(np.float64))
    print(f'potential: max |d| {d.max():.2e}, <= 1 ulp in {np.mean(_ulp(pot, g["potential"]) <= 1):.3f} of the cells')
    assert d.max() <= 1e-3
    # K2/K3: the reference's start cells (legacy RNG, seed 30) and Philox streams
    np.random.seed(30)
    r, c = movmodel.get_starting_indices(1000, (5, 55, 1, 2), 'random', (60., 50.), res)
    assert np.array_equal(r, g['start_rows']) and np.array_equal(c, g['start_cols'])
    out = movmodel.simulate_tracks(0., np.stack([r, c], 1), shape, 1, 1., upd, pot, seed=30, use_table=True)
    L = out.lengths.cpu().numpy()
    ends = out.ends.cpu().numpy()
    same_end = np.mean((ends == g['ends']).all(1))
    same_all = np.mean((L == g['lengths']) & (ends == g['ends']).all(1))
    print(f'tracks: identical end cell {same_end:.3f}, identical length and end {same_all:.3f}; '
          f'steps mean {L.mean():.0f} vs reference {g["lengths"].mean():.0f}')
    # A track is a chaotic function of the f32 potential: one differing ulp flips a move and
    # the track's random stream decides differently from there on.  C1 tracks are long (3000
    # steps on 500 rows), the two direct/iterative solves agree to ~3e-4 (5 f32 ulp; SuperLU
    # itself is only good to that at a condition number of 1e10), so few tracks stay identical
    # and the comparison is statistical: steps per track and the presence map.
    # The mean of 1000 heavy-tailed lengths (402 .. 20 883 steps) is itself a noisy number: the bound is four
    # standard errors of the difference of two such means, from the reference's own spread (measured: 3352
    # against 3060 steps, z = 2.0; profiles/r03_end_to_end_tolerances.txt)
    se = float(g['lengths'].std(ddof=1)) / np.sqrt(len(L)) * np.sqrt(2.)
    print(f'steps per track: {L.mean() - 1:.0f} vs {g["lengths"].mean() - 1:.0f}, z = {(L.mean() - g["lengths"].mean()) / se:.2f}')
    assert abs(L.mean() - g['lengths'].mean()) < 4. * se
    np.save(os.path.join(os.environ.get('GRAFT_REPO_ROOT', '.'), 'gpurun_out', 'c1_hip_potential.npy'), pot) \
        if os.path.isdir(os.path.join(os.environ.get('GRAFT_REPO_ROOT', '.'), 'gpurun_out')) else None
    # K4: smoothed, normalised presence map vs the reference's (strided sample of G8).  The
    # map is a density estimate of 1000 tracks; tracks that diverged land elsewhere, so the
    # stated tolerance is statistical: mean |d| <= 5e-4, max |d| <= 0.10 of the 0..1 range, correlation >= 0.997
    krad = presence.presence_kernel_radius(1000., res, shape)
    assert krad == int(g['krad'])
    sm = presence.smooth_presence_counts(out.hist, krad)
    acc = torch.zeros(shape, dtype=torch.float64, device='cuda')
    presence.normalise_add(sm, acc)
    pm = presence.normalise_to_f32(acc).cpu().numpy()[::8, ::8]
    dd = np.abs(pm.astype(np.float64) - g['presence_strided'])
    print(f'presence map: mean |d| {dd.mean():.4f}, max |d| {dd.max():.4f}, corr '
          f'{np.corrcoef(pm.ravel(), g["presence_strided"].ravel())[0, 1]:.4f}')
    stat = (dd.mean(), dd.max(), np.corrcoef(pm.ravel(), g['presence_strided'].ravel())[0, 1])
    # and with the reference's potential in place of K5's, the chain is exact again
    out2 = movmodel.simulate_tracks(0., np.stack([r, c], 1), shape, 1, 1., upd, g['potential'], seed=30, use_table=True)
    same2 = np.mean((out2.lengths.cpu().numpy() == g['lengths']) & (out2.ends.cpu().numpy() == g['ends']).all(1))
    print(f'tracks on K1 orograph + reference potential: identical {same2:.3f}')
    assert same2 >= 0.99          # the <= 1 ulp orograph cells may flip a handful of tracks
    # The yardstick is measured here, not assumed (profiles/r04_chain_noise.txt).  Sampling noise is NOT what
    # limits the agreement: another Philox seed on the reference's field reproduces the golden map to mean |d|
    # 1e-4, max |d| 0.015, correlation 0.9998.  The potential is: a track is a chaotic function of the f32
    # field, so the map is only defined up to the field's own uncertainty -- perturbing the REFERENCE's field by
    # random -1 / 0 / +1 f32 ulp (SuperLU's field is up to 12 ulp from the exact solution, G12) moves its map by
    # mean |d| 2.4e-4 .. 5.8e-4, max |d| 0.06 .. 0.14, correlation 0.983 .. 0.998.  K5's field (which is closer to the
    # exact solution than SuperLU's) must move the map no more than the worst of three such perturbations does.
    # (Rounds 1-3 asserted fixed numbers, about twice what one build showed: 5e-4 / 0.10 / 0.997; the V(1,1)
    # solver's field gives 2.0e-4 / 0.095 / 0.9952, the V(2,2) one 1.7e-4 / 0.047 / 0.9985: two realisations.)
    def metrics(potential, seed=30):
        o = movmodel.simulate_tracks(0., np.stack([r, c], 1), shape, 1, 1., upd, potential, seed=seed, use_table=True)
        a = torch.zeros(shape, dtype=torch.float64, device='cuda')
        presence.normalise_add(presence.smooth_presence_counts(o.hist, krad), a)
        m = presence.normalise_to_f32(a).cpu().numpy()[::8, ::8]
        d_ = np.abs(m.astype(np.float64) - g['presence_strided'])
        return d_.mean(), d_.max(), np.corrcoef(m.ravel(), g['presence_strided'].ravel())[0, 1]
    sampling = metrics(g['potential'], seed=31)
    print(f'another seed on the reference field: mean |d| {sampling[0]:.4f}, max |d| {sampling[1]:.4f}, corr {sampling[2]:.4f}')
    rng = np.random.default_rng(0)
    noise = []
    for _ in range(3):
        step = rng.integers(-1, 2, g['potential'].shape).astype(np.int32)
        noise.append(metrics((g['potential'].view(np.int32) + step).view(np.float32)))
        print(f'reference field +- 1 ulp: mean |d| {noise[-1][0]:.4f}, max |d| {noise[-1][1]:.4f}, corr {noise[-1][2]:.4f}')
    noise = (max(n[0] for n in noise), max(n[1] for n in noise), min(n[2] for n in noise))
    assert stat[0] <= noise[0] and stat[1] <= noise[1] and stat[2] >= noise[2], (stat, noise)
