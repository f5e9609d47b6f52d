"""One solved-field pass with the library's own counters (roam table: wave-pairs, slow wave-pairs)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from ssrs_amd import layers, movmodel
from ssrs_amd.potential import solve_potential
from ssrs_amd.synthetic import synthetic_dem
SHAPE, RES = (5000, 6000), 10.
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
cap = int(sys.argv[2]) if len(sys.argv) > 2 else None
dem = torch.from_numpy(synthetic_dem(SHAPE, RES)).cuda()
_, upd = layers.updraft_from_dem(dem, RES, 10., 270., threshold=0.75)
pot = solve_potential(upd, 0.)
np.random.seed(30)
r, c = movmodel.get_starting_indices(n, (5, 55, 1, 2), 'random', (60., 50.), RES)
starts = np.stack([r, c], 1).astype(np.int32)
table = movmodel.build_transition_table(upd, pot, thr=True, move_dirn=0.)
for rep in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    o = movmodel.simulate_tracks(0., starts, SHAPE, 1, 1., upd, pot, seed=30, table=table, profile=True, max_moves=cap)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    st = o.stats
    print(f'pass {rep}: {dt:.3f} s, {st["total_steps"] / dt:.3e} steps/s; launches {st["launches"]}, block-window {st["block_window_launches"]} '
          f'(roam {st["roam_launches"]}), {st["block_window_ms"]:.1f} ms for {st["block_window_steps"]:.3e} steps = '
          f'{st["block_window_ms"] * 1e3 / max(st["block_window_steps"], 1) * 1e3:.4f} ns per step; roam wave-pairs {st["roam_wave_pairs"]:.3e}, '
          f'slow {st["roam_slow_wave_pairs"]:.3e} ({st["roam_slow_wave_pairs"] / max(st["roam_wave_pairs"], 1):.4f})', flush=True)
