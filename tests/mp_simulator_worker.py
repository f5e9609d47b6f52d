"""One rank of a multi-process Simulator run (NOT a test module: started as a fresh child
process by tests/test_gpu_multirank.py, one process per rank, all ranks on cuda:0 with the gloo
backend -- RCCL refuses two ranks on one device; on an 8-GPU node the same code runs under nccl
with one device per rank).  usage: mp_simulator_worker.py RANK WORLD PORT OUT_DIR MODE"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def build(mode, out_dir, run_name):
    import numpy as np
    from ssrs_amd import Config, Simulator
    from ssrs_amd.synthetic import wind_lattice
    base = dict(run_name=run_name, out_dir=out_dir, sim_seed=30, region_width_km=(8., 6.),
                resolution=100., track_count=301, track_start_region=(1, 7, 0.2, 0.6),
                track_direction=0.)
    if mode == 'uniform':
        return Simulator(Config(**base), terrain='synthetic')
    if mode == 'subbatch':
        # 301 tracks, at most 150 per uint32 histogram: with two ranks rank 0 steps 151 (two sub-batches, its
        # counts widen to 64 bits) and rank 1 exactly 150 -- the ranks must still meet in the same collectives
        return Simulator(Config(hist_safe_tracks=150, **base), terrain='synthetic')
    if mode == 'fileguard':
        # ~301 tracks x ~70 points x 4 B = ~80 KB merged; the limit sits between one rank's share and the
        # merged file: every rank must refuse (a rank raising alone would leave its peer in a barrier)
        return Simulator(Config(max_tracks_file_gb=float(os.environ['SSRS_TEST_FILE_GB']), **base), terrain='synthetic')
    if mode == 'unseeded':            # uniform mode without a seed: the ranks must agree on rank 0's draws
        base['sim_seed'] = -1
        return Simulator(Config(**base), terrain='synthetic')
    if mode == 'snapshot':
        x, y, ws, wd = wind_lattice((8., 6.), 2.0)
        return Simulator(Config(sim_mode='snapshot', **base), terrain='synthetic',
                         wind=[dict(datetime=(2010, 6, 17, 13), x_km=x, y_km=y, wspeed=ws, wdirn=wd)])
    if mode == 'seasonal':
        wind = []
        for s in range(5):
            x, y, ws, wd = wind_lattice((8., 6.), 2.0, phase=2 * np.pi * s / 5)
            wind.append(dict(datetime=(2010, 4, 1 + s, 12), x_km=x, y_km=y, wspeed=ws, wdirn=wd))
        base['track_count'] = 120
        return Simulator(Config(sim_mode='seasonal', **base), terrain='synthetic', wind=wind)
    raise ValueError(mode)


def main():
    rank, world, port, out_dir, mode = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4], sys.argv[5]
    import torch
    import torch.distributed as dist
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = port
    torch.cuda.set_device(0)
    if world > 1:
        # a lost peer must fail this rank fast (non-zero exit), not park it for the default 30 minutes
        from datetime import timedelta
        dist.init_process_group('gloo', rank=rank, world_size=world, timeout=timedelta(seconds=180))
    sim = build(mode, out_dir, f'{mode}_w{world}')
    if mode == 'fileguard':
        try:
            sim.simulate_tracks()
        except ValueError as exc:
            print('REFUSED:', exc, flush=True)
            if world > 1:
                dist.barrier()          # every rank got here: nobody is parked in a collective
                dist.destroy_process_group()
            sys.exit(7)
        sys.exit(0)
    sim.simulate_tracks()
    if mode == 'unseeded':
        import json
        with open(os.path.join(out_dir, f'seeds_w{world}_r{rank}.json'), 'w') as f:
            json.dump(sorted((list(k), v) for k, v in sim.last_seeds.items()), f)
    sim.compute_presence_map(radius=300.)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
