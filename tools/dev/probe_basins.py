"""How are the wandering tracks of the solved 10 m field distributed in space?  100k tracks capped
at 60000 steps; occupancy of 64 x 64 tiles and of 144 x 256 windows by the capped tracks."""
import os, sys, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from ssrs_amd import layers, movmodel
from ssrs_amd.potential import solve_potential
from ssrs_amd.synthetic import synthetic_dem
shape = (5000, 6000); res = 10.
dem = torch.from_numpy(synthetic_dem(shape, res)).cuda()
_, upd = layers.updraft_from_dem(dem, res, 10., 270., threshold=0.75)
with warnings.catch_warnings():
    warnings.simplefilter('ignore')
    pot = solve_potential(upd, 0.)
n, cap = 100000, 60000
np.random.seed(30)
r, c = movmodel.get_starting_indices(n, (5, 55, 1, 2), 'random', (60., 50.), res)
starts = np.stack([r, c], 1)
out = movmodel.simulate_tracks(0., starts, shape, 1, 1., upd, pot, seed=30, use_table=True, max_moves=cap)
L = out.lengths.cpu().numpy() - 1
ends = out.ends.cpu().numpy()
capped = np.nonzero(L >= cap)[0]
e = ends[capped]
print(f'{n} tracks, {capped.size} at the cap of {cap}')
tile = (e[:, 0] // 64) * 128 + e[:, 1] // 64
u, cnt = np.unique(tile, return_counts=True)
order = np.argsort(-cnt)
print(f'{u.size} tiles of 64 x 64 hold them; the 20 fullest:')
for i in order[:20]:
    print(f'   tile row {u[i] // 128:3d} col {u[i] % 128:3d}: {cnt[i]} tracks')
print('cumulative share of the k fullest tiles:', {k: round(float(np.sort(cnt)[::-1][:k].sum() / capped.size), 3) for k in (1, 2, 4, 8, 16, 32, 64, 128)})
# greedy cover by 144 x 256 windows
left = np.ones(capped.size, bool); k = 0
while left.any() and k < 40:
    rr, cc = e[left, 0], e[left, 1]
    # densest window on a coarse grid
    h, re_, ce_ = np.histogram2d(rr, cc, bins=[np.arange(0, 5001 + 36, 36), np.arange(0, 6001 + 64, 64)])
    s = np.zeros_like(h)
    for a in range(4):
        for b in range(4):
            s[:h.shape[0] - a if a else None, :h.shape[1] - b if b else None] += h[a:, b:]
    i, j = np.unravel_index(np.argmax(s), s.shape)
    r0, c0 = i * 36, j * 64
    m = left & (e[:, 0] >= r0) & (e[:, 0] < r0 + 144) & (e[:, 1] >= c0) & (e[:, 1] < c0 + 256)
    print(f'   window rows {r0}..{r0 + 143} cols {c0}..{c0 + 255}: {int(m.sum())} tracks')
    left &= ~m; k += 1
print('left after', k, 'windows:', int(left.sum()))
