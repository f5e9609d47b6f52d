"""K3'/K4 presence density and K5 potential solver on the MI355X vs golden
vectors from the reference (g5, g8, g9) and vs the oracle."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def split(flat, lengths):
    off = np.concatenate([[0], np.cumsum(lengths)])
    return [flat[off[i]:off[i + 1]] for i in range(len(lengths))]


def ulp_diff_f32(a, b):
    a = np.ascontiguousarray(a, dtype=np.float32).view(np.int32).astype(np.int64)
    b = np.ascontiguousarray(b, dtype=np.float32).view(np.int32).astype(np.int64)
    return np.abs(a - b)


def test_presence_counts_and_smoothing_vs_golden(gpu, golden):
    from ssrs_amd import movmodel, presence
    g = golden('g9_presence.npz')
    tracks = split(g['tracks'], g['lengths'])
    counts = movmodel.compute_presence_counts(tracks, (40, 50))
    assert np.array_equal(counts, g['counts'])                    # integers: exact
    for rad in (2, 5, 13):
        sm = movmodel.compute_smooth_presence_counts(tracks, (40, 50), rad)
        assert sm.dtype == np.float32
        # integer chord sums x 1/ntaps vs scipy's f64 accumulation: <= 1 f32 ulp
        assert ulp_diff_f32(sm, g[f'smooth_r{rad}']).max() <= 1
    with pytest.raises(ValueError):    # the reference raises IndexError here
        movmodel.compute_presence_counts([np.array([[40, 0]], dtype=np.int16)], (40, 50))


def test_presence_c1_golden(gpu, golden):
    """C1: smoothed + normalised presence map of the 1000 reference tracks."""
    from ssrs_amd import presence
    g = golden('g8_c1.npz')
    krad = int(g['krad'])
    assert krad == presence.presence_kernel_radius(1000., 100., (500, 600))
    sm = presence.smooth_presence_counts(torch.from_numpy(g['hist']).cuda(), krad)
    assert abs(float(sm.max()) - float(g['presence_max_raw'])) <= 2e-7 * float(g['presence_max_raw'])
    acc = torch.zeros((500, 600), dtype=torch.float64, device='cuda')
    presence.normalise_add(sm, acc)                               # prprob /= amax; case += prprob
    out = presence.normalise_to_f32(acc).cpu().numpy()
    np.testing.assert_allclose(out[::8, ::8], g['presence_strided'], rtol=3e-7, atol=1e-9)
    assert out.max() == 1.0


def test_presence_smoothing_large_radius_vs_oracle(gpu):
    from ssrs_amd import presence
    from oracle import c_oracle
    rng = np.random.default_rng(2)
    cnt = rng.integers(0, 50, (70, 90)).astype(np.int32)
    for rad in (2, 9, 35, 60):
        got = presence.smooth_presence_counts(cnt, rad)
        ref = c_oracle.smooth_presence(cnt.astype(np.uint32), rad)
        assert ulp_diff_f32(got, ref).max() <= 1


def test_boundary_nodes_vs_golden(golden):
    from ssrs_amd.potential import get_boundary_nodes
    g = golden('g5_potential.npz')
    for dirn in (0., 180., -45., 90., 30.):
        tag = f'd{int(dirn % 360)}'
        bn, be = get_boundary_nodes(dirn, (48, 64))
        assert np.array_equal(bn, g[f'bnodes_{tag}']) and np.array_equal(be, g[f'benergy_{tag}'])


@pytest.mark.parametrize('dirn', [0., 180., -45., 90., 30.])
def test_potential_vs_reference_spsolve(gpu, golden, dirn):
    """AMG-preconditioned solve vs the reference's SuperLU solution (f32
    output) on the 48 x 64 golden, all boundary configurations.  Tolerance:
    2 f32 ulps of the 0..1000 range (1.3e-4)."""
    from ssrs_amd.potential import solve_potential, dirichlet_rasters
    g = golden('g5_potential.npz')
    pot, st = solve_potential(g['updraft'], dirn, rel_tol=1e-12, return_stats=True)
    ref = g[f'pot_d{int(dirn % 360)}']
    assert st['converged'] and st['iterations'] < 200, st
    assert pot.dtype == np.float32 and pot.shape == ref.shape
    np.testing.assert_allclose(pot, ref, rtol=0, atol=1.3e-4)
    mask, vals = dirichlet_rasters(dirn, ref.shape)
    assert np.array_equal(pot[mask == 1], vals[mask == 1].astype(np.float32))
    # determinism: the hierarchy uses no float atomics
    pot2 = solve_potential(g['updraft'], dirn, rel_tol=1e-12)
    assert np.array_equal(pot, pot2)


def test_potential_c1_vs_golden(gpu, golden):
    """Config C1 (500 x 600): the reference needs 1.3 s assembly + 10.8 s SuperLU
    (BASELINE.md); tolerance 1e-3 of the 0..1000 range on the f32 field."""
    from ssrs_amd import layers
    from ssrs_amd.potential import solve_potential
    g = golden('g8_c1.npz')
    upd = layers.get_above_threshold_speed(g['orograph_f32'], 0.75)
    pot, st = solve_potential(upd, 0., rel_tol=1e-10, return_stats=True)
    print('C1 potential solve:', st)
    assert st['converged'] and st['amg_levels'] >= 5, st
    np.testing.assert_allclose(pot, g['potential'], rtol=0, atol=1e-3)


def test_plain_bicgstab_switch_still_works_on_benign_problem(gpu):
    """A/B switch: without the AMG the solver is the v1 Jacobi-scaled BiCGStab,
    fine for uniform conductance (exact answer = linear ramp)."""
    from ssrs_amd.potential import solve_potential
    cond = np.ones((40, 30))
    ramp = 1000. * (1 - np.arange(40)[:, None] / 39.) * np.ones((1, 30))
    for amg in (True, False):
        pot, st = solve_potential(cond, 0., rel_tol=1e-12, max_iterations=5000,
                                  return_stats=True, use_amg=amg)
        assert st['converged'], st
        np.testing.assert_allclose(pot, ramp, rtol=0, atol=2e-3 if not amg else 2e-4)


def test_potential_two_phase_raster_iteration_budget(gpu):
    """Regression guard for the aggregation criterion (amg.hip: strong_link): a 10-m
    synthetic raster is a two-phase medium (about half the cells have zero usable
    updraft).  With the symmetric strength of connection and unbounded joins the
    V-cycle needs ~100 iterations; the one-sided criterion does not converge in 850.
    Also: the answer is a discrete harmonic function, so it obeys the maximum
    principle, and K-cycle / V-cycle agree."""
    import torch
    from ssrs_amd import layers
    from ssrs_amd.potential import solve_potential
    from ssrs_amd.synthetic import synthetic_dem
    dem = torch.from_numpy(synthetic_dem((700, 900), 10.)).cuda()
    _, upd = layers.updraft_from_dem(dem, 10., 10., 270., threshold=0.75)
    dead = float((upd <= 0).double().mean().item())
    assert 0.3 < dead < 0.85, dead
    pot, st = solve_potential(upd, 0., rel_tol=1e-9, max_iterations=400, return_stats=True)
    assert st['converged'] and st['iterations'] <= 250, st
    p = pot.cpu().numpy()
    assert p.min() >= -1e-3 and p.max() <= 1000.001
    assert abs(p[0].mean() - 1000.) < 1e-3 and abs(p[-1].mean()) < 1e-3      # Dirichlet rows
    pot_k, st_k = solve_potential(upd, 0., rel_tol=1e-9, max_iterations=400, return_stats=True,
                                  cycle='K', kdepth=3)
    assert st_k['converged'], st_k
    np.testing.assert_allclose(pot_k.cpu().numpy(), p, rtol=0, atol=2e-3)


def test_potential_workspace_tight_first_try_and_exhaustion(gpu, golden):
    """The size query is an upper bound (1.5 KB per cell); the host side first reserves 1.1 KB per
    cell (the hierarchy takes ~840 B).  A workspace that really is too small must fail loudly, not
    corrupt memory."""
    import ctypes as C
    from ssrs_amd import layers, _native as nat
    from ssrs_amd.potential import solve_potential, dirichlet_rasters
    from ssrs_amd._device import to_dev, stream_ptr
    g = golden('g8_c1.npz')
    upd = layers.get_above_threshold_speed(g['orograph_f32'], 0.75)
    pot, st = solve_potential(upd, 0., rel_tol=1e-10, return_stats=True)
    assert st['converged'] and 0 < st['workspace_used'] <= st['workspace_bytes']
    assert st['workspace_bytes'] < nat.lib().ssrs_potential_workspace_bytes(500, 600)
    assert 500 < st['workspace_used'] / (500 * 600) < 1100, st
    cond = to_dev(upd, torch.float64)
    mask_h, vals_h = dirichlet_rasters(0., (500, 600))
    mask, vals = to_dev(mask_h), to_dev(vals_h)
    out = torch.empty((500, 600), dtype=torch.float32, device='cuda')
    small = 10 * 8 * 500 * 600 + (8 << 20)                       # the solver's vectors + a sliver
    ws = torch.empty(small, dtype=torch.uint8, device='cuda')
    rc = nat.lib().ssrs_potential_solve(nat.ptr(cond), nat.ptr(mask), nat.ptr(vals), None, nat.ptr(out), 500, 600,
                                        C.c_double(1e-10), 100, 0, nat.ptr(ws), C.c_size_t(small), None, stream_ptr())
    assert rc == nat.SSRS_ERR_INVALID and b'workspace' in nat.lib().ssrs_last_error()


def test_potential_cycle_switches_agree(gpu):
    """The solver's A/B switches on one small two-phase raster: the fused level 0 of the V(1,1) cycle is bit-identical to the
    unfused kernels (SSRS_AMG_NO_FUSE), and the V(2,2) cycle of rounds 1-3 (SSRS_AMG_NU=2,2) converges to the same field
    within the solver's own uncertainty."""
    import os
    from ssrs_amd.potential import solve_potential
    rng = np.random.default_rng(5)
    cond = np.abs(rng.normal(0.8, 0.6, (300, 420)))
    cond[rng.random(cond.shape) < 0.45] = 0.0
    cond[100:160, 50:300] = 0.0
    base, st = solve_potential(cond, 0., return_stats=True)
    assert st['converged']
    for env, val, exact in (('SSRS_AMG_NO_FUSE', '1', True), ('SSRS_AMG_NU', '2,2', False)):
        os.environ[env] = val
        try:
            alt, st2 = solve_potential(cond, 0., return_stats=True)
        finally:
            del os.environ[env]
        assert st2['converged'], env
        if exact:
            assert st2['iterations'] == st['iterations'] and np.array_equal(alt, base), env
        else:
            assert np.abs(alt.astype(np.float64) - base.astype(np.float64)).max() <= 2e-3, env


def test_potential_falls_back_when_bicgstab_breaks_down(gpu):
    """Found by tests/dev/soak_potential.py (master seed 777, case 342679122: 72 x 88, heading 180, 70 % dead cells):
    under the default cycle BiCGStab on the exact operator exhausts its restarts at 2.4e-14 with single cells 2.6e-2 off
    the direct solve.  The solver then goes back to PCG's iterate and removes the east-edge defect by defect correction
    (potential.hip: k_quirk_defect); SSRS_SOLVE_NO_FALLBACK shows the state before."""
    import os
    from oracle import ssrs_oracle as orc
    from ssrs_amd.potential import solve_potential
    rng = np.random.default_rng(342679122)
    rows, cols = int(rng.integers(6, 90)), int(rng.integers(6, 110))
    dirn = float(rng.choice([0., 45., 90., 135., 180., 225., 270., 315., -45., rng.uniform(0, 360)]))
    cond = np.abs(rng.normal(0.8, 0.6, (rows, cols))) * 10.0 ** rng.uniform(-3, 1)
    cond[rng.random((rows, cols)) < rng.choice([0.0, 0.2, 0.5, 0.7])] = 0.0
    if rng.random() < 0.3:                                       # (the soak's contiguous dead block: same draws)
        r0, c0 = int(rng.integers(0, rows - 3)), int(rng.integers(0, cols - 3))
        cond[r0:r0 + rows // 3, c0:c0 + cols // 3] = 0.0
    assert (rows, cols, dirn) == (72, 88, 180.)
    ref = orc.solve_potential(cond, dirn).astype(np.float64)
    pot, st = solve_potential(cond, dirn, rel_tol=1e-15, max_iterations=3000, return_stats=True)
    assert st['converged'] and st['residual'] <= 1e-15, st
    assert np.abs(pot - ref).max() < 5e-3
    os.environ['SSRS_SOLVE_NO_FALLBACK'] = '1'
    try:
        _, st0 = solve_potential(cond, dirn, rel_tol=1e-15, max_iterations=3000, return_stats=True)
    finally:
        del os.environ['SSRS_SOLVE_NO_FALLBACK']
    assert not st0['converged'], st0                          # the case still needs it (and giving BiCGStab up early is the cheaper way: 400 against 720 iterations)
