import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, 'tests')
import numpy as np, torch
from conftest import load_golden
from ssrs_amd import layers, movmodel, presence
from ssrs_amd.potential import solve_potential
from ssrs_amd.synthetic import synthetic_dem
g = load_golden('g8_c1.npz')
shape=(500,600); res=100.
dem = synthetic_dem(shape, res)
oro, upd = layers.updraft_from_dem(dem, res, 10., 270., threshold=0.75)
np.random.seed(30)
r, c = movmodel.get_starting_indices(1000, (5, 55, 1, 2), 'random', (60., 50.), res)
krad = presence.presence_kernel_radius(1000., res, shape)
def pmap(pot, seed=30):
    out = movmodel.simulate_tracks(0., np.stack([r, c], 1), shape, 1, 1., upd, pot, seed=seed, use_table=True)
    acc = torch.zeros(shape, dtype=torch.float64, device='cuda')
    presence.normalise_add(presence.smooth_presence_counts(out.hist, krad), acc)
    L=out.lengths.cpu().numpy()
    return presence.normalise_to_f32(acc).cpu().numpy()[::8, ::8], L
def stat(pm):
    dd = np.abs(pm.astype(np.float64) - g['presence_strided'])
    return f'mean|d| {dd.mean():.5f} max|d| {dd.max():.4f} corr {np.corrcoef(pm.ravel(), g["presence_strided"].ravel())[0,1]:.5f}'
ref = g['potential']
print('reference potential seed 30:', stat(pmap(ref)[0]))
for s in (31, 32, 33): print(f'reference potential seed {s}:', stat(pmap(ref, s)[0]))
rng = np.random.default_rng(0)
for k in (1, 2, 4, 8):
    for rep in range(3):
        step = rng.integers(-k, k + 1, ref.shape).astype(np.int32)
        p = (ref.view(np.int32) + step).view(np.float32)
        pm, L = pmap(p)
        print(f'reference +- {k} ulp (rep {rep}):', stat(pm), 'mean steps', L.mean())
for nu in ('2,2', '1,1', '1,2', '2,1'):
    os.environ['SSRS_AMG_NU'] = nu
    pot = solve_potential(upd, 0.)
    pm, L = pmap(pot)
    print(f'K5 nu {nu}: max |pot-ref| {np.abs(pot.astype(np.float64)-ref).max():.2e}', stat(pm), 'mean steps', L.mean())
