"""The other BASELINE.json configurations, measured once each (MI355X):
  C1  uniform mode, 60 x 50 km @ 100 m, 1000 tracks, through ssrs_amd.Simulator end to end
  C4  snapshot raster chain at 10 m: WTK-shaped lattice -> per-cell wind -> slope/aspect -> updraft
  C5  the same for 16 snapshots per launch (seasonal batching), per-snapshot cost
(the track part of C4 / C5 is the regime of bench.py --potential solve)."""
import os, sys, time, tempfile, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from ssrs_amd import Config, Simulator, layers
from ssrs_amd.wind import interpolate_wind_lattice
from ssrs_amd.synthetic import synthetic_dem, wind_lattice


def sync_time(fn, reps=3):
    fn(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize(); t = time.time(); fn(); torch.cuda.synchronize()
        best = min(best, time.time() - t)
    return best


# ---- C1 through the Simulator (files on disk like the reference)
with tempfile.TemporaryDirectory() as out, warnings.catch_warnings():
    warnings.simplefilter('ignore')
    cfg = Config(run_name='c1', out_dir=out, max_cores=1, region_width_km=(60., 50.), resolution=100.,
                 sim_mode='uniform', uniform_winddirn=270., uniform_windspeed=10., track_direction=0.,
                 track_count=1000, sim_seed=30)
    t = time.time()
    sim = Simulator(cfg, terrain='synthetic')
    t_init = time.time() - t
    t = time.time(); sim.simulate_tracks(); torch.cuda.synchronize(); t_sim = time.time() - t
    t = time.time(); sim.plot_presence_map(); torch.cuda.synchronize(); t_pres = time.time() - t
    print(f'C1 Simulator: init + updraft {t_init:.2f} s, potential + 1000 tracks + pickles {t_sim:.2f} s, '
          f'presence map {t_pres:.2f} s   (reference: potential 12 s, tracks 26 s, BASELINE.md)', flush=True)

# ---- C4 / C5 raster chain at 10 m
rows, cols, res = 5000, 6000, 10.
dem = torch.from_numpy(synthetic_dem((rows, cols), res)).cuda()
slope, aspect = layers.slope_aspect(dem, res)
for B in (1, 16):
    lat = [wind_lattice((60., 50.), phase=2 * np.pi * s / 256) for s in range(B)]
    x, y = lat[0][0], lat[0][1]
    ws = np.stack([l[2] for l in lat]); wd = np.stack([l[3] for l in lat])
    def chain():
        s_r, d_r = interpolate_wind_lattice(x, y, ws if B > 1 else ws[0], wd if B > 1 else wd[0], (rows, cols), res)
        layers.orographic_updraft(s_r, d_r, slope, aspect, threshold=0.75)
    dt = sync_time(chain)
    print(f'C{4 if B == 1 else 5} raster chain, {B} snapshot(s) per launch: {dt * 1e3:.2f} ms = {dt / B * 1e3:.2f} ms per snapshot, '
          f'{rows * cols * B / dt / 1e6:.0f} Mcells/s', flush=True)
for B in (1, 16):
    lat = [wind_lattice((60., 50.), phase=2 * np.pi * s / 256) for s in range(B)]
    x, y = lat[0][0], lat[0][1]
    ws = np.stack([l[2] for l in lat]); wd = np.stack([l[3] for l in lat])
    ws_d, wd_d = torch.from_numpy(ws).cuda(), torch.from_numpy(wd).cuda()
    dt = sync_time(lambda: layers.updraft_from_dem_lattice(dem, res, x, y, ws_d if B > 1 else ws_d[0], wd_d if B > 1 else wd_d[0], threshold=0.75))
    print(f'C{4 if B == 1 else 5} fused DEM + lattice kernel, {B} snapshot(s) per launch: {dt * 1e3:.2f} ms = {dt / B * 1e3:.3f} ms per snapshot, '
          f'{rows * cols * B / dt / 1e6:.0f} Mcells/s', flush=True)
dt = sync_time(lambda: layers.slope_aspect(dem, res))
print(f'slope + aspect from the DEM (once per terrain): {dt * 1e3:.2f} ms, {rows * cols / dt / 1e6:.0f} Mcells/s', flush=True)
