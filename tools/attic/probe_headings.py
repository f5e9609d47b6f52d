"""C2 workload for other headings: linear ramp potential along the heading (stand-in), starts
in a band at the upstream edge.  How do the coherent schedule and the binning cope?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from ssrs_amd import layers, movmodel
from ssrs_amd.synthetic import synthetic_dem
rows, cols, res, n = 5000, 6000, 10., 100000
dem = torch.from_numpy(synthetic_dem((rows, cols), res)).cuda()
_, upd = layers.updraft_from_dem(dem, res, 10., 270., threshold=0.75)
rr = np.arange(rows, dtype=np.float64)[:, None]; cc = np.arange(cols, dtype=np.float64)[None, :]
rng = np.random.default_rng(30)
for dirn in ([float(v) for v in sys.argv[1:]] or (0., 90., 45., 135., 30., 250.)):
    th = np.deg2rad(dirn)
    along = rr * np.cos(th) + cc * np.sin(th)                 # distance along the heading (north = +row)
    pot = torch.from_numpy((1000. * (1. - (along - along.min()) / (along.max() - along.min()))).astype(np.float32)).cuda()
    # start band: 100-200 cells from the upstream edge
    t = rng.uniform(100, 200, n); s = rng.uniform(0.1, 0.9, n)
    if dirn in (0., 180.):
        r = t if dirn == 0. else rows - 1 - t; c = s * cols
    elif dirn in (90., 270.):
        c = t if dirn == 90. else cols - 1 - t; r = s * rows
    elif os.environ.get('PROBE_SOUTH_BAND'):
        # the Simulator's default start band (along the south edge) whatever the heading
        r = t if np.cos(th) > 0 else rows - 1 - t; c = s * cols
    else:
        # oblique: start bands along the two upstream edges
        up_r = t if np.cos(th) > 0 else rows - 1 - t
        up_c = t if np.sin(th) > 0 else cols - 1 - t
        pick = rng.random(n) < 0.5
        r = np.where(pick, up_r, s * rows)
        c = np.where(pick, s * cols, up_c)
    starts = np.stack([np.clip(r, 1, rows - 2), np.clip(c, 1, cols - 2)], 1).astype(np.int32)
    table = movmodel.build_transition_table(upd, pot, ring=True)
    hist = torch.zeros((rows, cols), dtype=torch.int32, device='cuda')
    for rep in range(2):
        hist.zero_(); torch.cuda.synchronize(); t0 = time.time()
        out = movmodel.simulate_tracks(dirn, starts, (rows, cols), 1, 1., upd, pot, seed=30, table=table, hist=hist, profile=True)
        torch.cuda.synchronize(); dt = time.time() - t0
    L = out.lengths.cpu().numpy()
    print(f'heading {dirn:5.0f}: {dt * 1e3:7.2f} ms, steps/track {L.mean():.0f}, stepper {out.stats["kernel_ms"]:.2f} ms, '
          f'binning {out.stats["hist_ms"]:.2f} ms, {out.stats["total_steps"] / dt / 1e9:.1f} G steps/s', flush=True)
