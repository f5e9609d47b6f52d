import os, sys
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, 'tests'))
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
def run(tag, **env):
    for k, v in env.items(): os.environ[k] = v
    try:
        g = movmodel.simulate_tracks(0., starts, (rows, cols), 1, 1., upd, pot, seed=8, use_table=True, thr=True, max_moves=cap, steps_per_launch=64)
    finally:
        for k in env: del os.environ[k]
    L = g.lengths.cpu().numpy(); H = g.hist.cpu().numpy().view(np.uint32); E = g.ends.cpu().numpy()
    dh = H.astype(np.int64) - ref['hist'].astype(np.int64)
    print(tag, 'lengths differ', int((L != ref['lengths']).sum()), 'ends differ', int((E != ref['ends']).any(1).sum()),
          'hist cells differ', int((dh != 0).sum()), 'sum diff', int(dh.sum()), 'launches', g.stats.get('launches'), flush=True)
    if (dh != 0).any():
        r, c = np.nonzero(dh)
        print('   rows', r.min(), r.max(), 'cols', c.min(), c.max(), 'first few', [(int(a), int(b), int(dh[a, b])) for a, b in zip(r[:8], c[:8])])
run('default')
run('fixed+norebalance', SSRS_TRACKS_NO_REBALANCE='1', SSRS_TRACKS_FIXED_STEPS='1')
run('fixed', SSRS_TRACKS_FIXED_STEPS='1')
run('norebalance', SSRS_TRACKS_NO_REBALANCE='1')
run('fixed+norebalance, no window', SSRS_TRACKS_NO_REBALANCE='1', SSRS_TRACKS_FIXED_STEPS='1', SSRS_TRACKS_NO_BLOCK_WINDOW='1')
run('fixed+norebalance, no rev', SSRS_TRACKS_NO_REBALANCE='1', SSRS_TRACKS_FIXED_STEPS='1', SSRS_TRACKS_NO_REV='1')
