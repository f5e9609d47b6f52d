"""A/B probe of stepper variants on the C2 workload (one process, interleaved)."""
import sys, time; sys.path.insert(0, '.')
import numpy as np, torch
from ssrs_amd import layers, movmodel
from ssrs_amd.synthetic import synthetic_dem, ramp_potential
rows, cols, res = 5000, 6000, 10.
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
dem = torch.from_numpy(synthetic_dem((rows, cols), res)).cuda()
np.random.seed(30)
r, c = movmodel.get_starting_indices(n, (5, 55, 1, 2), 'random', (60., 50.), res)
starts = torch.from_numpy(np.stack([r, c], 1).astype(np.int32)).cuda()
pot = torch.from_numpy(ramp_potential((rows, cols))).cuda()
_, upd = layers.updraft_from_dem(dem, res, 10., 270., threshold=0.75)
table = movmodel.build_transition_table(upd, pot)
hist = torch.zeros((rows, cols), dtype=torch.int32, device='cuda')
variants = {
    'table+hist': dict(table=table, want_hist=True),
    'table nohist': dict(table=table, want_hist=False),
    'table+hist S=64': dict(table=table, want_hist=True, steps_per_launch=64),
    'table+hist S=1024': dict(table=table, want_hist=True, steps_per_launch=1024),
    'table+hist nosched': dict(table=table, want_hist=True, schedule=False),
    'table nohist nosched': dict(table=table, want_hist=False, schedule=False),
    'direct+hist': dict(use_table=False, want_hist=True),
    'table+hist exact': dict(table=table, want_hist=True, exact_only=True),
    'table+hist nobin': dict(table=table, want_hist=True, binning=False),
}
res_ms = {k: [] for k in variants}
for rep in range(3):
    for name, kw in variants.items():
        kw = dict(kw)
        h = hist if kw.pop('want_hist') else None
        if h is not None:
            h.zero_()
        out = movmodel.simulate_tracks(0., starts, (rows, cols), 1, 1., upd, pot, seed=30, hist=h,
                                       want_hist=h is not None, profile=True, **kw)
        res_ms[name].append(out.stats['kernel_ms'] + out.stats['hist_ms'])
for name, v in res_ms.items():
    print(f'{name:24s} kernel_ms min {min(v):8.3f} med {sorted(v)[1]:8.3f}  steps {out.stats["total_steps"]}')
