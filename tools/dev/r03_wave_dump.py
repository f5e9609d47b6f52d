"""Analysis of the per-wave records a -DSSRS_DEBUG_WAVE_DUMP build prints (tools/dev/r03_wave_dump.sh):
clocks per pair of moves against live lanes, slow pairs and strays of each wave."""
import sys

import numpy as np

rows = [l.split()[1:] for l in open(sys.argv[1]) if l.startswith('W ')]
a = np.array(rows, dtype=np.int64)
blk, wv, lanes, dt, npairs, nslow, stray, r0, c0 = a.T[:9]
live = lanes > 0
cpp = dt / np.maximum(npairs, 1)
print(f'{int(live.sum())} waves with live lanes of {len(a)}; pairs per wave {np.unique(npairs[live]).tolist()}')
print('clocks per pair, quantiles 0 / 0.1 / 0.5 / 0.9 / 0.99 / 1:', np.round(np.quantile(cpp[live], [0, 0.1, 0.5, 0.9, 0.99, 1]), 1).tolist())
for name, v in (('live lanes', lanes), ('slow pairs', nslow), ('strays', stray)):
    print(f'correlation of clocks per pair with {name}: {np.corrcoef(cpp[live], v[live])[0, 1]:.3f}')
X = np.stack([np.ones(live.sum()), lanes[live], nslow[live] / npairs[live], stray[live] / npairs[live]], 1)
coef = np.linalg.lstsq(X, cpp[live], rcond=None)[0]
print('least squares: clocks per pair = %.1f %+.2f x lanes %+.0f x (slow pairs / pairs) %+.0f x (strays / pairs); residual std %.1f of %.1f'
      % (*coef, np.std(cpp[live] - X @ coef), np.std(cpp[live])))
for key in sorted(set(zip(r0[live].tolist(), c0[live].tolist()))):
    m = live & (r0 == key[0]) & (c0 == key[1])
    print(f'window at {key}: {int(m.sum())} waves, clocks per pair min / median / 0.9 / max', np.round(np.quantile(cpp[m], [0, 0.5, 0.9, 1]), 1).tolist(),
          'lanes %.1f, strays per pair %.4f, slow pairs %.4f' % (lanes[m].mean(), (stray[m] / npairs[m]).mean(), (nslow[m] / npairs[m]).mean()))
o = np.argsort(-cpp * live)
print('the slowest waves:')
for i in o[:6]:
    print(f'   block {blk[i]} wave {wv[i]}: {lanes[i]} lanes, {cpp[i]:.1f} clocks per pair, {nslow[i]} slow pairs, {stray[i]} strays, window {r0[i]}, {c0[i]}')
