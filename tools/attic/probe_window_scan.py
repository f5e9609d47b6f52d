"""Scan windows of the C2 DEM for track trapping (f32 plateau basins): HIP solver at the default
tolerance, 2048 tracks from the window's southern band; reports the share of tracks that take
more than 10 x rows steps.  usage: probe_window_scan.py [rows cols]"""
import os, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from ssrs_amd import layers, movmodel
from ssrs_amd.potential import solve_potential
from ssrs_amd.synthetic import synthetic_dem
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
cols = int(sys.argv[2]) if len(sys.argv) > 2 else 1200
res = 10.
full = synthetic_dem((5000, 6000), res)
rng = np.random.default_rng(1010)
n = 2048
starts = np.stack([rng.integers(10, 30, n), rng.integers(1, cols - 1, n)], 1)
for r0 in range(0, 5000 - rows + 1, rows):
    for c0 in range(0, 6000 - cols + 1, cols):
        dem = torch.from_numpy(full[r0:r0 + rows, c0:c0 + cols].copy()).cuda()
        _, upd = layers.updraft_from_dem(dem, res, 10., 270., threshold=0.75)
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            pot, st = solve_potential(upd, 0., return_stats=True)
        out = movmodel.simulate_tracks(0., starts, (rows, cols), 1, 1., upd, pot, seed=30, use_table=True)
        L = out.lengths.cpu().numpy() - 1
        mm = rows // 2 * (cols // 2)
        print(f'window ({r0:4d},{c0:4d}) dead {float((upd <= 0).double().mean()):.2f} it {st["iterations"]:3d}: steps median {np.median(L):.0f} '
              f'p95 {np.percentile(L, 95):.0f} max {L.max()} ; > 10 rows: {np.mean(L > 10 * rows):.3f}, at max_moves: {np.mean(L >= mm):.3f}', flush=True)
