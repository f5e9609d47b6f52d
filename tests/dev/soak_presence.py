"""Randomised soak of the presence kernels (K4) against the oracle: random histograms
(sparse, dense, heavy cells), radii from 1 to half the raster.  python tests/dev/soak_presence.py [s]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from ssrs_amd import presence
from oracle import ssrs_oracle as orc

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.
t0 = time.time(); n_case = 0; worst = 0.0
master = np.random.default_rng(99)
while time.time() - t0 < budget:
    seed = int(master.integers(0, 2**31)); rng = np.random.default_rng(seed)
    rows, cols = int(rng.integers(4, 260)), int(rng.integers(4, 300))
    krad = int(rng.integers(1, max(2, min(rows, cols) // 2)))
    kind = rng.choice(['sparse', 'dense', 'heavy'])
    if kind == 'sparse':
        h = (rng.random((rows, cols)) < 0.02) * rng.integers(1, 50, (rows, cols))
    elif kind == 'dense':
        h = rng.integers(0, 2000, (rows, cols))
    else:
        h = rng.integers(0, 5, (rows, cols)); h[rng.integers(0, rows), rng.integers(0, cols)] = 3_000_000
    h = h.astype(np.int64)
    ref = orc.smooth_presence_from_counts(h, krad)
    got = np.asarray(presence.smooth_presence_counts(h.astype(np.int32), krad))
    scale = max(float(np.abs(ref).max()), 1e-30)
    err = float(np.abs(got.astype(np.float64) - ref.astype(np.float64)).max() / scale)
    worst = max(worst, err)
    if err > 3e-7:                                           # ~2 f32 ulp of the largest value
        print('MISMATCH', dict(seed=seed, rows=rows, cols=cols, krad=krad, kind=str(kind)), err, flush=True)
        sys.exit(1)
    n_case += 1
print(f'soak ok: {n_case} cases, worst difference {worst:.1e} of the map maximum', flush=True)
