import sys; sys.path.insert(0,'.')
import numpy as np, torch
from ssrs_amd.potential import solve_potential
rows, cols = 1000, 1200
r = np.arange(rows)[:, None]; c = np.arange(cols)[None, :]
blob = (np.sin(c / 40.) * np.cos(r / 30.) > 0.2).astype(float)
rng = np.random.default_rng(0)
media = {
  'binary blobs (0 / 1)': blob,
  'binary blobs (1e-6 / 1), no zero rule': np.where(blob > 0, 1.0, 1e-6),
  'smooth log gradient 1e-10..1': 10.0 ** (-10 * (0.5 + 0.5 * np.sin(c / 40.) * np.cos(r / 30.))),
  'blobs with smooth edge to 0': np.clip(np.sin(c / 40.) * np.cos(r / 30.), 0, None) ** 5,
  'speckle 0/1 (p=0.5)': (rng.random((rows, cols)) < 0.5).astype(float),
}
for name, cond in media.items():
    for cyc in ('V', 'K'):
        pot, st = solve_potential(cond, 0., rel_tol=1e-8, max_iterations=800, return_stats=True, cycle=cyc)
        print(f'{name:40s} {cyc} its {st["iterations"]:5d} conv {st["converged"]} res {st["residual"]:.1e} ms {st["kernel_ms"]:.0f} lv {st["amg_levels"]} coarsest {st["amg_coarsest"]}', flush=True)
