"""LEAN (lengths-only) pass against the oracle on the g7 golden raster."""
import numpy as np, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ssrs_amd import movmodel
from oracle import c_oracle
g = np.load('tests/golden/g7_tracks.npz')
tag = 'ff_m1'
dirn, mem, nu, has_u, has_p = g[tag + '_params']
starts = np.stack([g['start_rows'], g['start_cols']], 1)
for spl in (16, 256):
    res = movmodel.simulate_tracks(float(dirn), starts, (96, 128), int(mem), float(nu), g['updraft'], g['potential'],
                                   seed=int(g['seed']), use_table=True, want_tracks=False, steps_per_launch=spl)
    lens = res.lengths.cpu().numpy(); ends = res.ends.cpu().numpy()
    want = g[tag + '_lengths']
    bad = np.nonzero(lens != want)[0]
    print('spl', spl, 'mismatching tracks', len(bad), 'of', len(want))
    for t in bad[:8]:
        print(' track', t, 'start', starts[t], 'len', lens[t], 'want', want[t], 'end', ends[t])
