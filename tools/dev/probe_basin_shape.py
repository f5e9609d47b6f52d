"""Which window shape holds a basin's visits?  C2 field, 100k tracks capped at --cap moves; the presence
histogram around each of the two big basins: share of the visits inside the best-placed window of
288 x 128, 144 x 256, 72 x 512 cells (the LDS histogram window of k_step_roam holds 36 864 counters).
usage: python tools/dev/probe_basin_shape.py [--cap 300000]"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ssrs_amd import layers, movmodel                      # noqa: E402
from ssrs_amd.potential import solve_potential             # noqa: E402
from ssrs_amd.synthetic import synthetic_dem               # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--cap', type=int, default=300_000)
args = ap.parse_args()
SHAPE, RES = (5000, 6000), 10.
dem = torch.from_numpy(synthetic_dem(SHAPE, RES)).cuda()
_, upd = layers.updraft_from_dem(dem, RES, 10., 270., threshold=0.75)
del dem
pot = solve_potential(upd, 0.)
np.random.seed(30)
r, c = movmodel.get_starting_indices(100_000, (5, 55, 1, 2), 'random', (60., 50.), RES)
starts = np.stack([r, c], 1).astype(np.int32)
# two runs: the histogram of the moves between cap/2 and cap is what the roaming launches see
a = movmodel.simulate_tracks(0., starts, SHAPE, 1, 1., upd, pot, seed=30, max_moves=args.cap // 2)
b = movmodel.simulate_tracks(0., starts, SHAPE, 1, 1., upd, pot, seed=30, max_moves=args.cap)
h = (b.hist.to(torch.int64) - a.hist.to(torch.int64)).double()
alive = (b.lengths - 1 >= args.cap)
ends = b.ends[alive].cpu().numpy()
print(f'{int(alive.sum())} tracks alive after {args.cap} moves; visits between {args.cap // 2} and {args.cap}: {float(h.sum()):.4e}')
for name, (r0, c0) in (('basin A', (2124, 3904)), ('basin B', (1440, 5056))):
    R0, R1, C0, C1 = max(r0 - 400, 0), min(r0 + 144 + 400, SHAPE[0]), max(c0 - 500, 0), min(c0 + 256 + 500, SHAPE[1])
    sub = h[R0:R1, C0:C1]
    tot = float(sub.sum())
    n_here = int(((ends[:, 0] >= R0) & (ends[:, 0] < R1) & (ends[:, 1] >= C0) & (ends[:, 1] < C1)).sum())
    rows_m = sub.sum(1).cpu().numpy(); cols_m = sub.sum(0).cpu().numpy()
    def span(m, q):
        cs = np.cumsum(m) / m.sum()
        return int(np.searchsorted(cs, q)), int(np.searchsorted(cs, 1 - q))
    print(f'{name}: {n_here} tracks end here, {tot:.4e} visits in the neighbourhood; cells visited {int((sub > 0).sum())}')
    for q in (1e-2, 1e-3, 1e-4):
        rl, rh = span(rows_m, q); cl, ch = span(cols_m, q)
        print(f'   rows holding all but {2 * q:.0e} of the visits: {R0 + rl}..{R0 + rh} ({rh - rl + 1}); cols {C0 + cl}..{C0 + ch} ({ch - cl + 1})')
    cs2 = torch.zeros((sub.shape[0] + 1, sub.shape[1] + 1), dtype=torch.float64, device=sub.device)
    cs2[1:, 1:] = sub.cumsum(0).cumsum(1)
    for wr, wc in ((288, 128), (144, 256), (72, 512), (192, 192), (288, 256)):
        if wr > sub.shape[0] or wc > sub.shape[1]:
            continue
        s = cs2[wr:, wc:] - cs2[:-wr, wc:] - cs2[wr:, :-wc] + cs2[:-wr, :-wc]
        best = float(s.max())
        idx = int(s.argmax()); br, bc = idx // s.shape[1], idx % s.shape[1]
        print(f'   window {wr} x {wc}: best origin ({R0 + br}, {C0 + bc}) holds {best / tot:.5f} of the visits (strays {1 - best / tot:.2e})')
    at = cs2[r0 - R0 + 144, c0 - C0 + 256] - cs2[r0 - R0, c0 - C0 + 256] - cs2[r0 - R0 + 144, c0 - C0] + cs2[r0 - R0, c0 - C0]
    print(f'   the window the run placed, 144 x 256 at ({r0}, {c0}): {float(at) / tot:.5f}')
