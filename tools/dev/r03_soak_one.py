"""One soak case by its seed (tests/dev/soak_tracks.py prints it on a mismatch): the same generator, every
threshold-table variant under the A/B switches, what differs from the C oracle.
usage: python tools/dev/r03_soak_one.py <seed> [spl]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ssrs_amd import movmodel                              # noqa: E402
from oracle import c_oracle                                # noqa: E402

seed = int(sys.argv[1])
rng = np.random.default_rng(seed)
rows, cols = int(rng.integers(5, 400)), int(rng.integers(5, 500))
if rng.random() < 0.15:                                   # more than one tile column (1024 cells)
    rows, cols = int(rng.integers(5, 200)), int(rng.integers(1025, 2600))
n = int(rng.choice([1, 7, 64, 65, 300, 2000, 9000, 20000]))
dirn = float(rng.choice([0., 45., 90., 135., 180., 225., 270., 315., rng.uniform(0, 360)]))
kind = rng.choice(['rough', 'smooth', 'flat', 'speckle', 'nan', 'wells', 'scales'])
if kind == 'wells' and rows * cols > 40000:               # wandering tracks run to rows / 2 * cols / 2 moves
    kind = 'rough'
upd = np.abs(rng.normal(0.8, 0.6, (rows, cols)))
if kind in ('speckle', 'nan'):
    upd[rng.random((rows, cols)) < 0.5] = 0.0
ramp = 1000. * (1 - np.arange(rows)[:, None] / max(rows - 1., 1.))
if kind == 'flat':
    pot = np.full((rows, cols), 7.0, dtype=np.float32)
elif kind == 'wells':                                     # tracks circle in the wells: window -> tile buckets
    rr, cc = np.arange(rows)[:, None], np.arange(cols)[None, :]
    pot = ramp + 0. * cc
    for _ in range(int(rng.integers(1, 12))):
        r0, c0 = rng.integers(0, rows), rng.integers(0, cols)
        pot = pot - rng.uniform(100., 900.) * np.exp(-((rr - r0) ** 2 + (cc - c0) ** 2) / (2. * rng.uniform(3., 12.) ** 2))
    pot = pot.astype(np.float32)
elif kind == 'smooth':
    pot = (ramp + 0 * upd).astype(np.float32)
else:
    pot = (ramp + rng.normal(0, rng.choice([0.01, 1.0, 30.0]), (rows, cols))).astype(np.float32)
if kind == 'nan':
    upd[rng.random((rows, cols)) < 0.01] = np.nan
if kind == 'scales':                                      # every magnitude f32 has (and some it has not)
    upd = upd * 10. ** rng.uniform(-9, 39, upd.shape)
    upd[rng.random((rows, cols)) < 0.01] = np.inf
    band = 10. ** rng.integers(-44, 8, rows // 8 + 1).astype(np.float64)
    pot = (pot.astype(np.float64) * np.repeat(band, 8)[:rows, None]).astype(np.float32)
starts = np.stack([rng.integers(0, rows, n), rng.integers(0, cols, n)], 1)
s = int(rng.integers(0, 2**31))
mem = int(rng.choice([1, 1, 1, 0, 2, 3, 8]))

ref = c_oracle.simulate_tracks(dirn, starts, (rows, cols), mem, 1., upd, pot, seed=s, want_traj=False)
spl = int(rng.choice([0, 2, 7, 16, 64, 512]))
if len(sys.argv) > 2:
    spl = int(sys.argv[2])
print(dict(seed=seed, rows=rows, cols=cols, n=n, dirn=dirn, kind=str(kind), spl=spl, mem=mem, steps=int(ref['steps'])))
for kw in (dict(use_table=True, thr=True), dict(use_table=True, thr=True, scattered=True), dict(use_table=True, thr=True, schedule=False),
           dict(use_table=True, thr=True, binning=False)):
    for switch in ('', 'SSRS_TRACKS_NO_ROAM_STOP', 'SSRS_TRACKS_NO_ROAM_TABLE', 'SSRS_TRACKS_NO_BLOCK_WINDOW', 'SSRS_TRACKS_NO_CHEAP_EXACT',
                   'SSRS_TRACKS_NO_FINE_TABLE'):
        if switch:
            os.environ[switch] = '1'
        try:
            res = movmodel.simulate_tracks(dirn, starts, (rows, cols), mem, 1., upd, pot, seed=s, steps_per_launch=spl, **kw)
        finally:
            if switch:
                del os.environ[switch]
        le = res.lengths.cpu().numpy(); en = res.ends.cpu().numpy(); hi = res.hist.cpu().numpy().view(np.uint32)
        bad_l = np.nonzero(le != ref['lengths'])[0]
        bad_e = np.nonzero((en != ref['ends']).any(1))[0]
        dh = hi.astype(np.int64) - ref['hist'].astype(np.int64)
        st = res.stats
        print(kw, switch or '(default)', 'lengths differ:', len(bad_l), 'ends differ:', len(bad_e), 'hist cells differ:', int((dh != 0).sum()),
              'sum', int(dh.sum()), '| roam launches', st.get('roam_launches'), 'block window', st.get('block_window_launches'), 'sorts', st.get('wander_sorts'),
              'launches', st.get('launches'))
        if len(bad_l):
            i = bad_l[:5]
            print('    tracks', i.tolist(), 'gpu lengths', le[i].tolist(), 'oracle', ref['lengths'][i].tolist(), 'starts', starts[i].tolist())
