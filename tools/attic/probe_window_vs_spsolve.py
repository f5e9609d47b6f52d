"""HIP potential solver vs the reference's direct solve on a window of the 10 m DEM (the field
comes from oracle.solve_potential = assemble + SuperLU, staged in scratch/ by the build container):
field difference, and the same tracks stepped through both fields."""
import os, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from ssrs_amd import layers, movmodel
from ssrs_amd.potential import solve_potential
g = np.load(sys.argv[1])
oro32, ref = g['oro32'], g['pot']
rows, cols = oro32.shape
upd = layers.get_above_threshold_speed(torch.from_numpy(oro32).cuda(), 0.75)
rng = np.random.default_rng(1010)
n = 4096
starts = np.stack([rng.integers(10, 30, n), rng.integers(1, cols - 1, n)], 1)
refd = torch.from_numpy(ref).cuda()
base = movmodel.simulate_tracks(0., starts, (rows, cols), 1, 1., upd, refd, seed=30, use_table=True)
Lb = base.lengths.cpu().numpy()
print(f'reference field: steps mean {Lb.mean():.0f} median {np.median(Lb):.0f} max {Lb.max()}', flush=True)
for tol in (1e-8, 1e-12, 1e-15):
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        pot, st = solve_potential(upd, 0., rel_tol=tol, return_stats=True)
    d = (pot.double() - refd.double()).abs()
    ulp = (pot.view(torch.int32).long() - refd.view(torch.int32).long()).abs()
    out = movmodel.simulate_tracks(0., starts, (rows, cols), 1, 1., upd, pot, seed=30, use_table=True)
    L = out.lengths.cpu().numpy()
    same = np.mean(L == Lb)
    same_end = float((out.ends == base.ends).all(1).double().mean())
    print(f'rel_tol {tol:g}: {st["iterations"]} it, res {st["residual"]:.1e}; max |d| {float(d.max()):.3e}, mean {float(d.mean()):.3e}, '
          f'bit-identical cells {float((ulp == 0).double().mean()):.4f}, <=1 ulp {float((ulp <= 1).double().mean()):.4f}, max ulp {int(ulp.max())}; '
          f'tracks: steps mean {L.mean():.0f} median {np.median(L):.0f} max {L.max()}; same length {same:.3f}, same end cell {same_end:.3f}', flush=True)
