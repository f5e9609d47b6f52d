import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from ssrs_amd import movmodel
rows = cols = 3000
rr, cc = np.arange(rows, dtype=np.float64)[:, None], np.arange(cols, dtype=np.float64)[None, :]
d2 = (rr - 1500.) ** 2 + (cc - 1500.) ** 2
upd = np.zeros((rows, cols))
rng = np.random.default_rng(1)
n = 8192
starts = np.stack([rng.integers(1400, 1460, n), rng.integers(1450, 1550, n)], 1)
for name, pot in (('cone+gauss', 1000. * (1. - rr / (rows - 1.)) + 4.0 * np.sqrt(d2) - 3000. * np.exp(-d2 / (2. * 2.5 ** 2))),
                  ('cone only', 4.0 * np.sqrt(d2)),
                  ('gauss wide', 1000. * (1. - rr / (rows - 1.)) - 900. * np.exp(-d2 / (2. * 60. ** 2)))):
    o = movmodel.simulate_tracks(0., starts, (rows, cols), 1, 1., upd, pot.astype(np.float32), seed=5, max_moves=200_000)
    L = o.lengths.cpu().numpy() - 1
    h = o.hist.cpu().numpy().view(np.uint32)
    e = o.ends.cpu().numpy()
    print(name, 'steps mean', L.mean(), 'at cap', (L >= 200_000).mean(), 'hist max', h.max(), 'share of hottest cell', h.max() / h.sum(),
          'cells > 1e-3 of visits', int((h > 1e-3 * h.sum()).sum()), 'ends rows', e[:, 0].min(), e[:, 0].max(), o.stats['launches'], o.stats['roam_launches'], flush=True)
