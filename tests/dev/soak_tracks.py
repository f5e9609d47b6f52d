"""Randomised soak: random shapes, fields, headings, batch sizes and switches; every GPU
data path against the C oracle.  python tests/dev/soak_tracks.py [seconds]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from ssrs_amd import movmodel
from oracle import c_oracle

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.
t0 = time.time(); n_case = 0; n_steps = 0
master = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 2026)       # [seconds] [master seed]
while time.time() - t0 < budget:
    seed = int(master.integers(0, 2**31))
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
    variants = [dict(use_table=True, ring=False), dict(use_table=True, ring=False, scattered=True), dict(use_table=False)]
    if mem == 1 and spl % 2 == 0:
        variants += [dict(use_table=True, ring=True), dict(use_table=True, ring=True, scattered=True),
                     dict(use_table=True, ring=True, schedule=False), dict(use_table=True, ring=True, binning=False),
                     dict(use_table=True, thr=True), dict(use_table=True, thr=True, scattered=True),
                     dict(use_table=True, thr=True, schedule=False), dict(use_table=True, thr=True, binning=False)]
    check_traj = rng.random() < 0.15 and int(ref['steps']) < 3e6
    if check_traj:                                            # recorded trajectories, any path
        full = c_oracle.simulate_tracks(dirn, starts, (rows, cols), mem, 1., upd, pot, seed=s)
        kw = dict(variants[int(rng.integers(0, len(variants)))])
        res = movmodel.simulate_tracks(dirn, starts, (rows, cols), mem, 1., upd, pot, seed=s, steps_per_launch=spl,
                                       want_tracks=True, **kw)
        if not (np.array_equal(res.traj.cpu().numpy(), np.concatenate(full['tracks'])) and
                np.array_equal(res.hist.cpu().numpy().view(np.uint32), ref['hist'])):
            print('TRAJECTORY MISMATCH', dict(seed=seed, rows=rows, cols=cols, n=n, dirn=dirn, kind=kind, spl=spl, mem=mem), kw, flush=True)
            sys.exit(1)
    if kind == 'wells' and n >= 9000 and mem == 1:
        # many short launches: the batch gets to the wander sort and the block windows (k_step_thr<6>)
        os.environ['SSRS_TRACKS_FIXED_STEPS'] = '1'
        try:
            res = movmodel.simulate_tracks(dirn, starts, (rows, cols), mem, 1., upd, pot, seed=s, steps_per_launch=32, use_table=True, thr=True)
        finally:
            del os.environ['SSRS_TRACKS_FIXED_STEPS']
        if not (np.array_equal(res.lengths.cpu().numpy(), ref['lengths']) and np.array_equal(res.ends.cpu().numpy(), ref['ends']) and
                np.array_equal(res.hist.cpu().numpy().view(np.uint32), ref['hist'])):
            print('MISMATCH (short launches)', dict(seed=seed, rows=rows, cols=cols, n=n, dirn=dirn, kind=kind, mem=mem), res.stats, flush=True)
            sys.exit(1)
        n_windowed = globals().get('n_windowed', 0) + (1 if res.stats['block_window_launches'] else 0)
        globals()['n_windowed'] = n_windowed
    for kw in variants:
        res = movmodel.simulate_tracks(dirn, starts, (rows, cols), mem, 1., upd, pot, seed=s, steps_per_launch=spl, **kw)
        ok = (np.array_equal(res.lengths.cpu().numpy(), ref['lengths']) and
              np.array_equal(res.ends.cpu().numpy(), ref['ends']) and
              np.array_equal(res.hist.cpu().numpy().view(np.uint32), ref['hist']))
        if not ok:
            print('MISMATCH', dict(seed=seed, rows=rows, cols=cols, n=n, dirn=dirn, kind=kind, spl=spl, mem=mem), kw, flush=True)
            sys.exit(1)
    n_case += 1; n_steps += int(ref['steps'])
    if n_case % 20 == 0:
        print(f'{n_case} cases, {n_steps:.3e} oracle steps, {time.time() - t0:.0f} s', flush=True)
print(f'soak ok: {n_case} cases x 3-11 GPU variants, {n_steps:.3e} steps each; {globals().get("n_windowed", 0)} short-launch wells cases reached the block windows', flush=True)
