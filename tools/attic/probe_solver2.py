"""A/B of the AMG aggregation criterion and cycle on the GPU: C1 golden raster and
synthetic-DEM rasters at 10 m (1000 x 1200, optionally larger)."""
import sys, time; sys.path.insert(0, '.')
import numpy as np, torch
from ssrs_amd.potential import solve_potential
from ssrs_amd import layers
from ssrs_amd.synthetic import synthetic_dem

cases = []
g8 = np.load('tests/golden/g8_c1.npz')
cases.append(('C1 500x600', layers.get_above_threshold_speed(g8['orograph_f32'], 0.75), g8['potential']))
for shape in [(1000, 1200)] + ([tuple(int(v) for v in a.split('x')) for a in sys.argv[1:]]):
    dem = torch.from_numpy(synthetic_dem(shape, 10.)).cuda()
    _, upd = layers.updraft_from_dem(dem, 10., 10., 270., threshold=0.75)
    cases.append((f'synthetic {shape[0]}x{shape[1]}', upd, None))
variants = [('one-sided V (old)', dict(one_sided=True)),
            ('symmetric V', dict()),
            ('symmetric K3', dict(cycle='K', kdepth=3)),
            ('symmetric K2', dict(cycle='K', kdepth=2))]
for name, upd, ref in cases:
    for vn, kw in variants:
        torch.cuda.synchronize(); t = time.time()
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            pot, st = solve_potential(upd, 0., rel_tol=1e-8, max_iterations=800, return_stats=True, **kw)
        torch.cuda.synchronize(); dt = time.time() - t
        err = '' if ref is None else f' maxabs {np.abs(np.asarray(pot) - ref).max():.2e}'
        print(f'{name:22s} {vn:20s} its {st["iterations"]:5d} conv {st["converged"]} res {st["residual"]:.1e} '
              f'levels {st["amg_levels"]} coarsest {st["amg_coarsest"]} solve {st["kernel_ms"] / 1e3:.2f}s wall {dt:.2f}s{err}',
              flush=True)
