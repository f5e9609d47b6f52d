"""In-kernel steps/s of the roaming stepper against the number of tracks: 100k tracks leave ~44k survivors in
~240 thinned blocks (one per CU); the kernel's throughput bound is every CU holding a FULL block of 256."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from ssrs_amd import layers, movmodel
from ssrs_amd.potential import solve_potential
from ssrs_amd.synthetic import synthetic_dem
SHAPE, RES = (5000, 6000), 10.
dem = torch.from_numpy(synthetic_dem(SHAPE, RES)).cuda()
_, upd = layers.updraft_from_dem(dem, RES, 10., 270., threshold=0.75)
pot = solve_potential(upd, 0.)
table = movmodel.build_transition_table(upd, pot, thr=True, move_dirn=0.)
for n in (50_000, 100_000, 125_000, 140_000, 150_000, 200_000, 300_000):
    np.random.seed(30)
    r, c = movmodel.get_starting_indices(n, (5, 55, 1, 2), 'random', (60., 50.), RES)
    starts = np.stack([r, c], 1).astype(np.int32)
    o = movmodel.simulate_tracks(0., starts, SHAPE, 1, 1., upd, pot, seed=30, table=table, profile=True, max_moves=600_000)
    st = o.stats
    alive = int((o.lengths - 1 >= 600_000).sum())
    print(f'{n} tracks, {alive} at the cap: block-window launches {st["block_window_launches"]} (pair table {st["roam_launches"]}), '
          f'{st["block_window_steps"] / max(st["block_window_ms"], 1e-9) * 1e3:.3e} steps/s in them, waves {st["roam_wave_pairs"] / max(st["roam_launches"], 1) / 32768:.0f}', flush=True)
