"""Batch time of 100k tracks at C2 on a ramp along the heading: threshold table (default) against the ring table,
headings 0 / 45 / 90 / 180 (which histogram path each one takes is in the stats)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from ssrs_amd import layers, movmodel
from ssrs_amd.synthetic import synthetic_dem
rows, cols, res, n = 5000, 6000, 10., 100000
dem = torch.from_numpy(synthetic_dem((rows, cols), res)).cuda()
_, upd = layers.updraft_from_dem(dem, res, 10., 270., threshold=0.75)
rr = np.arange(rows, dtype=np.float64)[:, None]; cc = np.arange(cols, dtype=np.float64)[None, :]
for dirn in (0., 45., 90., 180.):
    rng = np.random.default_rng(30)
    th = np.deg2rad(dirn)
    along = rr * np.cos(th) + cc * np.sin(th)
    pot = torch.from_numpy((1000. * (1. - (along - along.min()) / (along.max() - along.min()))).astype(np.float32)).cuda()
    t = rng.uniform(100, 200, n); s = rng.uniform(0.1, 0.9, n)
    up_r = t if np.cos(th) > 1e-9 else rows - 1 - t
    up_c = t if np.sin(th) > 1e-9 else cols - 1 - t
    pick = rng.random(n) < (0.5 if abs(np.sin(th) * np.cos(th)) > 1e-9 else (1.0 if abs(np.cos(th)) > 0.5 else 0.0))
    r = np.where(pick, up_r, s * rows); c = np.where(pick, s * cols, up_c)
    starts = np.stack([np.clip(r, 1, rows - 2), np.clip(c, 1, cols - 2)], 1).astype(np.int32)
    for name, kw in (('thr', dict(thr=True)), ('ring', dict(ring=True))):
        table = movmodel.build_transition_table(upd, pot, move_dirn=dirn, **kw)
        hist = torch.zeros((rows, cols), dtype=torch.int32, device='cuda')
        best = 1e9
        for rep in range(3):
            hist.zero_(); torch.cuda.synchronize(); t0 = time.time()
            out = movmodel.simulate_tracks(dirn, starts, (rows, cols), 1, 1., upd, pot, seed=30, table=table, hist=hist)
            torch.cuda.synchronize(); best = min(best, time.time() - t0)
        st = out.stats
        print(f'heading {dirn:5.1f} {name:4s}: {best * 1e3:7.2f} ms per batch, {st["total_steps"] / n:7.0f} steps/track, launches {st["launches"]}, '
              f'window {st["window_launches"]} tiles {st["tile_launches"]} block windows {st["block_window_launches"]}', flush=True)
