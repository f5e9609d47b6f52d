"""Wandering regime (smooth 10-m terrain, solved potential, tracks up to max_moves): what
bounds a step -- the table gather, the histogram atomic, or neither?"""
import os, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from ssrs_amd import layers, movmodel
from ssrs_amd.potential import solve_potential
from ssrs_amd.synthetic import synthetic_dem
shape = (5000, 6000)
dem = torch.from_numpy(synthetic_dem(shape, 10.)).cuda()
_, upd = layers.updraft_from_dem(dem, 10., 10., 270., threshold=0.75)
with warnings.catch_warnings():
    warnings.simplefilter('ignore')
    pot, st = solve_potential(upd, 0., max_iterations=3000, return_stats=True)   # library default rel_tol
print('solve', st['iterations'], 'dead fraction', float((upd <= 0).double().mean()), flush=True)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
cap = int(sys.argv[2]) if len(sys.argv) > 2 else 60000
np.random.seed(30)
r, c = movmodel.get_starting_indices(n, (5, 55, 1, 2), 'random', (60., 50.), 10.)
starts = np.stack([r, c], 1)
table = movmodel.build_transition_table(upd, pot, ring=True)
for name, kw in (("auto (window -> tiles -> copies)", dict()), ("private copies + zero mask", dict(scattered=True)),
                 ('no histogram', dict(scattered=False, want_hist=False))):
    torch.cuda.synchronize(); t = time.time()
    out = movmodel.simulate_tracks(0., starts, shape, 1, 1., upd, pot, seed=30, table=table, profile=True, max_moves=cap, **kw)
    torch.cuda.synchronize(); dt = time.time() - t
    L = out.lengths.cpu().numpy()
    if out.hist is not None:
        h = out.hist
        print(f'   histogram: {int(h.sum().item()):.3e} visits in {int((h > 0).sum().item()):.3e} distinct cells, max count {int(h.max().item())}', flush=True)
    print(f'{name:32s} steps {out.stats["total_steps"]:.3e} mean {L.mean():.0f} max {L.max()} launches {out.stats["launches"]} '
          f'wall {dt:.2f} s  {out.stats["total_steps"] / dt / 1e9:.2f} G steps/s  (stepper {out.stats["kernel_ms"]:.1f} ms, binning {out.stats["hist_ms"]:.1f} ms)', flush=True)
