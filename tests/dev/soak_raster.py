"""Randomised soak of the raster kernels (K1) against the numpy oracle: random shapes,
DEM roughness, resolutions, winds, thresholds.  Tolerances as in tests/test_gpu_raster.py.
python tests/dev/soak_raster.py [seconds]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from ssrs_amd import layers
from oracle import ssrs_oracle as orc


def ulp_diff_f32(a, b):
    a = np.ascontiguousarray(a, dtype=np.float32).view(np.int32).astype(np.int64)
    b = np.ascontiguousarray(b, dtype=np.float32).view(np.int32).astype(np.int64)
    return np.abs(a - b)


budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.
t0 = time.time(); n_case = 0; cells = 0; worst_ulp = 0; ident = []; worst_abs = 0.0; n_cancel = 0
master = np.random.default_rng(77)
while time.time() - t0 < budget:
    seed = int(master.integers(0, 2**31)); rng = np.random.default_rng(seed)
    rows, cols = int(rng.integers(3, 700)), int(rng.integers(3, 900))
    res = float(rng.choice([10., 30., 100., rng.uniform(5, 200)]))
    z = 1500. + rng.normal(0, rng.choice([0.0, 0.5, 20., 300.]), (rows, cols)) + \
        200. * np.sin(np.arange(cols)[None, :] / rng.uniform(3, 80)) * np.cos(np.arange(rows)[:, None] / rng.uniform(3, 80))
    if rng.random() < 0.2:
        z = np.round(z)                                   # plateaus: dz_dx == 0 cells (the 1e-10 quirk)
    ws, wd = float(rng.uniform(0.5, 25.)), float(rng.choice([0., 90., 180., 270., rng.uniform(-360, 720)]))
    thr = float(rng.choice([0.75, 0.3, 2.0]))
    slope, aspect = orc.compute_slope_degrees(z, res), orc.compute_aspect_degrees(z, res)
    oro_ref = orc.compute_orographic_updraft(ws, wd, slope, aspect).astype(np.float32)
    s, a = layers.slope_aspect(z, res)
    np.testing.assert_allclose(s, slope, rtol=1e-12, atol=1e-13)
    np.testing.assert_allclose(a, aspect, rtol=1e-12, atol=1e-11)
    oro = layers.compute_orographic_updraft(ws, wd, slope, aspect)
    d1 = ulp_diff_f32(oro, oro_ref)
    oro_f, usable_f = layers.updraft_from_dem(z, res, ws, wd, threshold=thr)
    d2 = ulp_diff_f32(np.asarray(oro_f), oro_ref)
    # cells where cos(aspect - wdirn) cancels to ~1e-16 carry no information in their low bits
    # (the reference's own result there is rounding noise of the degree->radian conversion):
    # more than 1 f32 ulp is accepted only if the absolute difference is below 1e-12 * wspeed
    for d, got in ((d1, np.asarray(oro)), (d2, np.asarray(oro_f))):
        bad = d > 1
        if bad.any():
            absd = np.abs(got.astype(np.float64) - oro_ref.astype(np.float64))[bad].max()
            worst_abs = max(worst_abs, float(absd))
            if absd > 1e-12 * ws:
                print('MISMATCH', dict(seed=seed, rows=rows, cols=cols, res=res, ws=ws, wd=wd), int(d.max()), absd, flush=True)
                sys.exit(1)
            n_cancel += int(bad.sum())
        worst_ulp = max(worst_ulp, int(d[~bad].max()) if (~bad).any() else 0)
    ident.append(float((d2 == 0).mean()))
    use_ref = orc.get_above_threshold_speed(np.asarray(oro_f), thr)
    np.testing.assert_allclose(np.asarray(usable_f), use_ref, rtol=1e-12, atol=1e-15)
    n_case += 1; cells += rows * cols
print(f'soak ok: {n_case} cases, {cells:.3e} cells, worst f32 ulp difference {worst_ulp} '
      f'({n_cancel} cancellation cells beyond that, largest absolute difference {worst_abs:.1e}), '
      f'bit-identical fraction of the fused orograph: min {min(ident):.5f} mean {np.mean(ident):.5f}', flush=True)
