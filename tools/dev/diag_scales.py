import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), 'tests'))
import numpy as np
from ssrs_amd import movmodel
from oracle import c_oracle
from test_gpu_tracks import _random_field_case
rows, cols = 150, 170
for mode in ['upd_scale', 'upd_inf', 'pot_bands', 'all']:
    upd, pot = _random_field_case(rows, cols, 31)
    r2 = np.random.default_rng(77)
    sc = 10. ** r2.uniform(-9, 39, upd.shape); infm = r2.random(upd.shape) < 0.01
    band = 10. ** r2.integers(-44, 8, rows // 10 + 1).astype(np.float64)
    if mode in ('upd_scale', 'all'): upd = upd * sc
    if mode in ('upd_inf', 'all'): upd = upd.copy(); upd[infm] = np.inf
    if mode in ('pot_bands', 'all'): pot = (pot.astype(np.float64) * np.repeat(band, 10)[:rows, None]).astype(np.float32)
    rng = np.random.default_rng(5); n = 900
    starts = np.stack([rng.integers(0, rows, n), rng.integers(0, cols, n)], 1)
    ref = c_oracle.simulate_tracks(20., starts, (rows, cols), 1, 1., upd, pot, seed=11, track_id_base=77, want_traj=False)
    for name, kw in [('window', dict(use_table=False)), ('f64', dict(use_table=True, ring=False)), ('ring', dict(use_table=True, ring=True)), ('thr', dict(use_table=True, thr=True))]:
        res = movmodel.simulate_tracks(20., starts, (rows, cols), 1, 1., upd, pot, seed=11, track_id_base=77, steps_per_launch=32, **kw)
        l = res.lengths.cpu().numpy(); bad = np.flatnonzero(l != ref['lengths'])
        print(mode, name, 'mismatching tracks', bad.size, bad[:5], flush=True)
