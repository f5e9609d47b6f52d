import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), 'tests'))
import numpy as np, torch
from ssrs_amd import movmodel
from oracle import c_oracle
from test_gpu_tracks import _random_field_case
rows, cols = 420, 900
upd, pot = _random_field_case(rows, cols, 12)
pot = pot.copy()
rr, cc = np.arange(rows)[:, None], np.arange(cols)[None, :]
for r0, c0 in ((150, 60), (260, 95), (330, 40)):
    pot -= (600. * np.exp(-((rr - r0) ** 2 + (cc - c0) ** 2) / (2. * 9. ** 2))).astype(np.float32)
rng = np.random.default_rng(21)
n = 12000
starts = np.stack([rng.integers(2, 30, n), rng.integers(5, 110, n)], 1)
cap = 6000
ref = c_oracle.simulate_tracks(0., starts, (rows, cols), 1, 1., upd, pot, seed=8, max_moves=cap, want_traj=False)
def run(env):
    for k, v in env.items(): os.environ[k] = v
    try:
        o = movmodel.simulate_tracks(0., starts, (rows, cols), 1, 1., upd, pot, seed=8, use_table=True, thr=True, max_moves=cap, steps_per_launch=64)
    finally:
        for k in env: del os.environ[k]
    L = o.lengths.cpu().numpy(); E = o.ends.cpu().numpy(); H = o.hist.cpu().numpy().view(np.uint32)
    bad = np.nonzero(L != ref['lengths'])[0]
    print(env, {k: o.stats[k] for k in ('launches', 'block_window_launches', 'roam_launches', 'wander_sorts', 'roam_wave_pairs', 'roam_slow_wave_pairs')},
          'lengths differ:', len(bad), 'ends differ:', int((E != ref['ends']).any(1).sum()), 'hist cells differ:', int((H != ref['hist']).sum()),
          'hist sum', int(H.sum()), 'ref', int(ref['hist'].sum()))
    for t in bad[:6]:
        print('   track', t, 'start', starts[t], 'gpu len', L[t], 'ref', ref['lengths'][t], 'gpu end', E[t], 'ref', ref['ends'][t])
    d = np.argwhere(H != ref['hist'])
    for r, c in d[:8]:
        print('   cell', r, c, 'gpu', H[r, c], 'ref', ref['hist'][r, c])
run({})
run({'SSRS_TRACKS_NO_REBALANCE': '1', 'SSRS_TRACKS_FIXED_STEPS': '1'})
run({'SSRS_TRACKS_NO_REBALANCE': '1', 'SSRS_TRACKS_FIXED_STEPS': '1', 'SSRS_TRACKS_NO_ROAM_TABLE': '1'})
run({'SSRS_TRACKS_FIXED_STEPS': '1'})
run({'SSRS_TRACKS_NO_REBALANCE': '1'})
os.environ['SSRS_TRACKS_DEBUG'] = '1'
run({'SSRS_TRACKS_FIXED_STEPS': '1', 'SSRS_TRACKS_NO_ROAM_TABLE': '1'})
