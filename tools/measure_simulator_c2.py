"""ssrs_amd.Simulator end to end at BASELINE configs[1] (5000 x 6000 @10 m): constructor (K1 +
file), simulate_tracks, presence map.  Three runs: the ramp stand-in potential (seeded through
the <id>_potential.npy cache of the file contract) with and without tracks.pkl, and the solved
potential (10 000 and 100 000 tracks, no pickle: a third of them run to max_moves)."""
import os, sys, time, tempfile, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from ssrs_amd import Config, Simulator
from ssrs_amd.synthetic import synthetic_dem, ramp_potential

shape = (5000, 6000)
dem = synthetic_dem(shape, 10.)


def run(tag, tracks, save, seed_ramp):
    with tempfile.TemporaryDirectory() as out, warnings.catch_warnings():
        warnings.simplefilter('ignore')
        cfg = Config(run_name='c2', out_dir=out, region_width_km=(60., 50.), resolution=10., sim_mode='uniform',
                     uniform_winddirn=270., uniform_windspeed=10., track_direction=0., track_count=tracks,
                     sim_seed=30, save_tracks=save)
        torch.cuda.synchronize(); t = time.time()
        sim = Simulator(cfg, terrain=dem)
        torch.cuda.synchronize(); t_init = time.time() - t
        if seed_ramp:
            np.save(sim._get_potential_fname(sim.case_ids[0], 0, sim.mode_data_dir) + '.npy', ramp_potential(shape))
        t = time.time(); sim.simulate_tracks(); torch.cuda.synchronize(); t_sim = time.time() - t
        st = list(sim.last_stats.values())[0]
        t = time.time(); sim.plot_presence_map(); torch.cuda.synchronize(); t_pres = time.time() - t
        pk = [f for f in os.listdir(sim.mode_data_dir) if f.endswith('.pkl')]
        size = sum(os.path.getsize(os.path.join(sim.mode_data_dir, f)) for f in pk) / 1e9
        print(f'{tag}: constructor (K1 + orograph.npy) {t_init:.2f} s; simulate_tracks {t_sim:.2f} s '
              f'({st["total_steps"]:.3e} steps, stepper kernels {st["kernel_ms"] / 1e3:.3f} s, recorded {st.get("recorded")}; '
              f'tracks.pkl {size:.2f} GB); presence map (radius 1 km = 100 cells) {t_pres:.2f} s', flush=True)


run('ramp potential, 100k tracks, save_tracks=False', 100_000, False, True)
run('ramp potential, 100k tracks, save_tracks=True ', 100_000, True, True)
run('solved potential, 10k tracks, save_tracks=False', 10_000, False, False)
run('solved potential, 100k tracks, save_tracks=False', 100_000, False, False)      # the bench's workload, through the API
