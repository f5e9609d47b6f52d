"""Where do the roaming survivors of a solved-field batch step, and how much table do they touch?
C2 field (K5 potential), 100k tracks capped at 60 000 moves; the survivors' end cells then start a
fresh recorded batch of `--steps` moves whose trajectories are analysed on the device:
distinct cells / distinct (cell, last move) states per 4096-step chunk and per basin window,
per-track excursion per chunk.  Sizes a table that would have to sit in L2 / LDS for them.
usage: python tools/dev/probe_roam_footprint.py [--steps 16384] [--tracks 100000]"""
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
ap.add_argument('--steps', type=int, default=16384)
ap.add_argument('--tracks', type=int, default=100_000)
args = ap.parse_args()
SHAPE, RES, CAP = (5000, 6000), 10., 60_000
dem = torch.from_numpy(synthetic_dem(SHAPE, RES)).cuda()
_, upd = layers.updraft_from_dem(dem, RES, 10., 270., threshold=0.75)
del dem
pot = solve_potential(upd, 0.)
np.random.seed(30)
r, c = movmodel.get_starting_indices(args.tracks, (5, 55, 1, 2), 'random', (60., 50.), RES)
starts = np.stack([r, c], 1).astype(np.int32)
a = movmodel.simulate_tracks(0., starts, SHAPE, 1, 1., upd, pot, seed=30, max_moves=CAP)
alive = (a.lengths - 1 >= CAP)
ends = a.ends[alive].to(torch.int32)
print(f'{int(alive.sum())} of {args.tracks} tracks alive after {CAP} moves')
# coarse windows like k_wander_windows: 144 x 256 cells
wr, wc = ends[:, 0] // 144, ends[:, 1] // 256
key = wr * 64 + wc
uk, cnt = torch.unique(key, return_counts=True)
order = torch.argsort(cnt, descending=True)
for i in order[:6].tolist():
    print(f'  window rows {int(uk[i]) // 64 * 144}.. cols {int(uk[i]) % 64 * 256}..: {int(cnt[i])} tracks')
b = movmodel.simulate_tracks(0., ends, SHAPE, 1, 1., upd, pot, seed=31, max_moves=args.steps, want_tracks=True,
                             record_pool_bytes=int(ends.shape[0]) * (args.steps + 1024) * 5)
assert b.stats['recorded'], 'pool too small'
off = b.offsets
traj = b.traj.to(torch.int32)
n = int(ends.shape[0])
full = (b.lengths == args.steps + 1)
print(f'{int(full.sum())} of {n} restarted tracks take all {args.steps} moves (burn-in nudges apply to none of them: '
      f'rows {int(ends[:, 0].min())}..{int(ends[:, 0].max())})')
idx = torch.nonzero(full).flatten()
# (tracks x steps+1) matrix of the full-length tracks
base = off[idx]
pts = traj[(base[:, None] + torch.arange(args.steps + 1, device=traj.device)[None, :]).flatten()].reshape(len(idx), args.steps + 1, 2)
cell = pts[..., 0].long() * SHAPE[1] + pts[..., 1].long()
dr = pts[:, 1:, 0] - pts[:, :-1, 0]
dc = pts[:, 1:, 1] - pts[:, :-1, 1]
move = (dr + 1) * 3 + (dc + 1)                              # k index of the move that led to point i + 1
state = cell[:, 1:] * 9 + move.long()                       # (cell, last move) when standing on point i + 1
CH = 4096
for lo in range(0, args.steps - CH + 1, CH):
    cs = cell[:, 1 + lo:1 + lo + CH]
    ss = state[:, lo:lo + CH]
    ncell = int(torch.unique(cs).numel())
    nstate = int(torch.unique(ss).numel())
    rows_span = (pts[:, 1 + lo:1 + lo + CH, 0].max(1).values - pts[:, 1 + lo:1 + lo + CH, 0].min(1).values).float()
    cols_span = (pts[:, 1 + lo:1 + lo + CH, 1].max(1).values - pts[:, 1 + lo:1 + lo + CH, 1].min(1).values).float()
    q = torch.tensor([0.5, 0.9, 0.99, 1.0], device=rows_span.device)
    print(f'steps {lo}..{lo + CH}: {ncell} distinct cells, {nstate} distinct (cell, last move) states '
          f'({nstate * 4 / 1024:.0f} KB at 4 B, {nstate * 16 / 1024:.0f} KB at 16 B, {nstate * 64 / 1024:.0f} KB at 64 B); '
          f'per-track row span p50/90/99/max {torch.quantile(rows_span, q).tolist()}, col span {torch.quantile(cols_span, q).tolist()}')
    # visits concentration: share of the visits on the hottest states
    us, uc = torch.unique(ss, return_counts=True)
    uc = torch.sort(uc, descending=True).values.double()
    cum = torch.cumsum(uc, 0) / uc.sum()
    for k in (256, 1024, 4096, 16384, 65536):
        if k <= uc.numel():
            print(f'    hottest {k} states hold {float(cum[k - 1]):.4f} of the visits')
for ch in (64, 256, 1024):
    m = (args.steps // ch) * ch
    pr = pts[:, 1:1 + m, 0].reshape(len(idx), -1, ch)
    pc = pts[:, 1:1 + m, 1].reshape(len(idx), -1, ch)
    rs = (pr.max(2).values - pr.min(2).values).float().flatten()
    cs2 = (pc.max(2).values - pc.min(2).values).float().flatten()
    q = torch.tensor([0.5, 0.9, 0.99, 0.999, 1.0], device=rs.device)
    print(f'excursion per {ch}-step chunk: rows p50/90/99/99.9/max {torch.quantile(rs[:4_000_000], q).tolist()}, '
          f'cols {torch.quantile(cs2[:4_000_000], q).tolist()}')
# how often is a step a 2-cycle (back to the cell of two steps ago)?
back = (cell[:, 2:] == cell[:, :-2]).double().mean()
print(f'share of steps that return to the cell of two steps before: {float(back):.3f}')
