"""One oblique-heading batch at C2 for rocprofv3: python tools/attic/probe_one_heading.py 45"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from ssrs_amd import layers, movmodel
from ssrs_amd.synthetic import synthetic_dem
rows, cols, res, n = 5000, 6000, 10., 100000
dirn = float(sys.argv[1]) if len(sys.argv) > 1 else 45.
dem = torch.from_numpy(synthetic_dem((rows, cols), res)).cuda()
_, upd = layers.updraft_from_dem(dem, res, 10., 270., threshold=0.75)
rr = np.arange(rows, dtype=np.float64)[:, None]; cc = np.arange(cols, dtype=np.float64)[None, :]
rng = np.random.default_rng(30)
th = np.deg2rad(dirn)
along = rr * np.cos(th) + cc * np.sin(th)
pot = torch.from_numpy((1000. * (1. - (along - along.min()) / (along.max() - along.min()))).astype(np.float32)).cuda()
t = rng.uniform(100, 200, n); s = rng.uniform(0.1, 0.9, n)
up_r = t if np.cos(th) > 0 else rows - 1 - t
up_c = t if np.sin(th) > 0 else cols - 1 - t
pick = rng.random(n) < 0.5
r = np.where(pick, up_r, s * rows); c = np.where(pick, s * cols, up_c)
starts = np.stack([np.clip(r, 1, rows - 2), np.clip(c, 1, cols - 2)], 1).astype(np.int32)
table = movmodel.build_transition_table(upd, pot, ring=True)
hist = torch.zeros((rows, cols), dtype=torch.int32, device='cuda')
for rep in range(3):
    hist.zero_()
    out = movmodel.simulate_tracks(dirn, starts, (rows, cols), 1, 1., upd, pot, seed=30, table=table, hist=hist)
torch.cuda.synchronize()
print('done', int(hist.sum().item()))
