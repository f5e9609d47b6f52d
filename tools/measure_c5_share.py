"""BASELINE configs[4], ONE GPU's share at its size, end to end through ssrs_amd.Simulator: seasonal mode, 32 synthetic
WTK-shaped wind snapshots (of the 256: 8 GPUs x 32), 60 x 50 km @10 m, 10 000 fluidflow tracks per snapshot, default
solver tolerance, save_tracks=False (/root/reference/ssrs/simulator.py:200-215 orographs per snapshot, :259-288 one
potential per snapshot, :348-369 tracks per snapshot, :520-546 presence map).  Prints seconds per phase.
usage: python tools/measure_c5_share.py [snapshots=32] [tracks=10000]"""
import os, sys, time, tempfile, threading, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from ssrs_amd import Config, Simulator, movmodel
from ssrs_amd import potential as potential_mod
from ssrs_amd.synthetic import wind_lattice

nsnap = int(sys.argv[1]) if len(sys.argv) > 1 else 32
ntracks = int(sys.argv[2]) if len(sys.argv) > 2 else 10_000
lock = threading.Lock()
acc = dict(solve_s=0.0, solve_n=0, solve_its=0, tracks_s=0.0, tracks_n=0, steps=0, at_max=0, ntr=0)
_solve, _sim = potential_mod.solve_potential, movmodel.simulate_tracks


def solve(*a, **k):
    torch.cuda.current_stream().synchronize(); t = time.perf_counter()
    k['return_stats'] = True
    out, st = _solve(*a, **k)
    torch.cuda.current_stream().synchronize()
    with lock:
        acc['solve_s'] += time.perf_counter() - t; acc['solve_n'] += 1; acc['solve_its'] += st['iterations']
    return out


def sim(*a, **k):
    torch.cuda.current_stream().synchronize(); t = time.perf_counter()
    out = _sim(*a, **k)
    torch.cuda.current_stream().synchronize()
    L = out.lengths.cpu().numpy().astype(np.int64) - 1
    with lock:
        acc['tracks_s'] += time.perf_counter() - t; acc['tracks_n'] += 1; acc['steps'] += out.stats['total_steps']
        acc['at_max'] += int((L >= 7_500_000).sum()); acc['ntr'] += L.size
    return out


potential_mod.solve_potential = solve
movmodel.simulate_tracks = sim
wind = []
for s in range(nsnap):
    x, y, ws, wd = wind_lattice((60., 50.), 2.0, phase=2 * np.pi * s / 256)
    wind.append(dict(datetime=(2010, 1 + s // 28, 1 + s % 28, 12), x_km=x, y_km=y, wspeed=ws, wdirn=wd))
with tempfile.TemporaryDirectory() as out, warnings.catch_warnings():
    warnings.simplefilter('ignore')
    cfg = Config(run_name='c5', out_dir=out, max_cores=8, region_width_km=(60., 50.), resolution=10., sim_mode='seasonal',
                 track_direction=0., track_count=ntracks, sim_seed=30, save_tracks=False, print_verbose=False)
    t = time.perf_counter(); sim_ = Simulator(cfg, terrain='synthetic', wind=wind); torch.cuda.synchronize(); t_init = time.perf_counter() - t
    t = time.perf_counter(); sim_.simulate_tracks(); torch.cuda.synchronize(); t_sim = time.perf_counter() - t
    t = time.perf_counter(); sim_.compute_presence_map(radius=1000.); torch.cuda.synchronize(); t_pres = time.perf_counter() - t
print(f'C5 share: {nsnap} snapshots x {ntracks} tracks on one MI355X, end to end {t_init + t_sim + t_pres:.1f} s')
print(f'  constructor (DEM, K1 lattice batches, {nsnap} orograph .npy of 120 MB): {t_init:.1f} s')
print(f'  simulate_tracks: {t_sim:.1f} s wall = K5 {acc["solve_s"]:.1f} s in {acc["solve_n"]} solves ({acc["solve_s"] / max(acc["solve_n"], 1):.2f} s, '
      f'{acc["solve_its"] / max(acc["solve_n"], 1):.0f} iterations each; incl. waiting for the GPU when several run at once) '
      f'+ tracks {acc["tracks_s"]:.1f} s summed over {acc["tracks_n"]} batches ({acc["steps"]:.3e} steps, '
      f'{acc["at_max"] / max(acc["ntr"], 1):.3f} of the tracks at max_moves) + potential .npy I/O and the rest')
print(f'  presence map ({nsnap} smoothings at radius 100 cells + the /max ladder + summary_presence.npy): {t_pres:.1f} s')
