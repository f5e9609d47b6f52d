#!/usr/bin/env python3
"""Headline benchmark (BASELINE.json): simulated tracks/s + updraft-raster
Mcells/s on the 60x50 km @10 m uniform-mode raster (5000 x 6000 cells),
100k tracks per MI355X.

One "step" = one pass of the hot path over one batch of synthetic input, with
every input already resident in HBM:
    DEM --K1 fused raster--> orograph f32 + usable updraft f64
        --K2a--> per-cell transition table (updraft x potential)
        --K2b/K3--> 100k tracks stepped to completion + uint32 presence histogram
        [N > 1: one RCCL sum-reduce of the histogram to rank 0]
Tracks shard over ranks by global track id (weak scaling: 100k tracks per GPU).

Launch: `python bench.py` (1 GPU) or
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
 --master-port P bench.py --gpus N --steps K --warmup W`.
Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s (spec)
STEP_BYTES = 76                 # 9x4 + 9x4 window + 4 B point/RMW
RASTER_BYTES_PER_CELL = 20      # fused K1 as benchmarked: f64 DEM in (8) + f32 out (4)
#                                 + f64 usable out (8); SURVEY's 12 B/cell figure is
#                                 the f32-in/f32-out elementwise kernel


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--tracks', type=int, default=100_000, help='tracks per GPU')
    ap.add_argument('--resolution', type=float, default=10.0)
    ap.add_argument('--width-km', type=float, nargs=2, default=(60.0, 50.0))
    ap.add_argument('--direct', action='store_true', help='3x3 window gathers, no table')
    ap.add_argument('--steps-per-launch', type=int, default=0)
    ap.add_argument('--cpu-seconds', type=float, default=20.0,
                    help='target CPU time of the cpu_baseline sample (0 = skip)')
    ap.add_argument('--potential', default='ramp', choices=['ramp', 'solve'])
    ap.add_argument('--solve-iterations', type=int, default=2000)
    ap.add_argument('--solved-tracks', type=int, default=10_000,
                    help='tracks of the solved_potential leg (outside the timed region; 0 = skip)')
    ap.add_argument('--no-chain-probe', action='store_true',
                    help='skip the 16 384-track dependent-chain measurement (profiling runs: keeps per-kernel averages clean)')
    ap.add_argument('--ref-cpu-tracks', type=int, default=3,
                    help='tracks of the reference-equivalent (numpy restatement) CPU rate (0 = skip)')
    ap.add_argument('--no-binning', action='store_true', help='per-step global atomics for the histogram')
    ap.add_argument('--no-schedule', action='store_true', help='disable the coherent schedule')
    ap.add_argument('--exact-only', action='store_true', help='disable the fast decision path')
    ap.add_argument('--f64-table', action='store_true', help='8 x f64 transition table instead of the threshold table')
    ap.add_argument('--ring-table', action='store_true', help='f32 ring table (round 1 stepper) instead of the threshold table')
    ap.add_argument('--dem-noise', type=float, default=1.5,
                    help='sigma of the per-cell DEM noise in metres (SURVEY 8(d) prescribes 1.5 at every resolution)')
    return ap.parse_args()


def cpu_baseline(args, gridsize, dem, pot, starts, seed, steps_per_track):
    """Oracle (C port of the reference algorithm, OpenMP) on the host cores, on
    a bounded sample of the same workload: the full-grid raster once and the
    first M tracks (same global ids / Philox streams as the GPU run)."""
    from oracle import c_oracle
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    t0 = time.perf_counter()
    slope, aspect = c_oracle.slope_aspect(dem, args.resolution)
    _, oro32 = c_oracle.orographic(slope, aspect, 10.0, 270.0)
    upd = c_oracle.threshold(oro32, 0.75)
    t_raster = time.perf_counter() - t0
    # calibrate, then size the sample for ~cpu_seconds of stepping
    t0 = time.perf_counter()
    cal = c_oracle.simulate_tracks(0.0, starts[:2 * cores], gridsize, 1, 1.0, upd, pot,
                                   seed=seed, want_traj=False, want_hist=True, nthreads=cores)
    t_cal = max(time.perf_counter() - t0, 1e-6)
    rate = cal['steps'] / t_cal
    m = int(min(len(starts), max(2 * cores, rate * args.cpu_seconds / max(steps_per_track, 1))))
    t0 = time.perf_counter()
    run = c_oracle.simulate_tracks(0.0, starts[:m], gridsize, 1, 1.0, upd, pot, seed=seed,
                                   want_traj=False, want_hist=True, nthreads=cores)
    t_run = time.perf_counter() - t0
    ref_equiv = None
    if args.ref_cpu_tracks > 0:
        # the reference's own arithmetic speed: the numpy restatement (oracle/ssrs_oracle.py, the
        # same per-step numpy calls as /root/reference/ssrs/movmodel.py:264-318, which cannot travel)
        from oracle import ssrs_oracle as orc
        from oracle.philox import TrackUniforms
        t0 = time.perf_counter()
        nsteps = 0
        for t in range(args.ref_cpu_tracks):
            tr = orc.generate_simulated_tracks(0.0, (int(starts[t, 0]), int(starts[t, 1])), gridsize, 1, 1.0,
                                               upd, pot, uniform=TrackUniforms(seed, t))
            nsteps += len(tr) - 1
            assert len(tr) == run['lengths'][t], 'numpy restatement and C port disagree'
        t_ref = time.perf_counter() - t0
        ref_equiv = {'steps_per_s_per_core': nsteps / t_ref, 'cores': 1, 'tracks': args.ref_cpu_tracks,
                     'steps': nsteps, 'seconds': t_ref,
                     'what': 'numpy restatement of generate_simulated_tracks, one process, first tracks of '
                             'this workload; SURVEY measured 12.1 k steps/s/core for the reference itself'}
    return {
        'value': m / t_run, 'unit': 'tracks/s', 'cores': cores, 'kind': 'port',
        'reference_equivalent': ref_equiv,
        'sample': (f'C/OpenMP oracle port, first {m} of the {len(starts)} tracks of this '
                   f'workload ({run["steps"]} steps in {t_run:.1f} s) on {cores} host threads; '
                   f'raster chain on the full grid once ({t_raster:.1f} s)'),
        'steps_per_s': run['steps'] / t_run,
        'steps_per_s_per_core': run['steps'] / t_run / cores,
        'raster_mcells_per_s': gridsize[0] * gridsize[1] / t_raster / 1e6,
    }, run, m


def build_table(args, movmodel, upd, pot):
    if args.f64_table or args.exact_only:
        return movmodel.build_transition_table(upd, pot)
    if args.ring_table:
        return movmodel.build_transition_table(upd, pot, ring=True)
    return movmodel.build_transition_table(upd, pot, thr=True, move_dirn=0.0)


def chain_probe(args, movmodel, layers, dem, pot, starts_h, gridsize, res, seed):
    """What bounds the stepper: one step of a track is a chain of dependent instructions
    (Philox -> decision -> next address -> 12-byte gather -> ...).  A batch of 16 384 tracks is
    one wave per CU, nothing to overlap with: launch time / steps = the chain's latency.  The
    full batch (1.5 waves per SIMD) cannot run a launch faster than S x that latency."""
    import torch
    _, upd = layers.updraft_from_dem(dem, res, 10.0, 270.0, threshold=0.75)
    table = build_table(args, movmodel, upd, pot)
    n = 16384
    sub = torch.from_numpy(starts_h[:n]).to(dem.device)
    best = None
    for _ in range(3):
        o = movmodel.simulate_tracks(0.0, sub, gridsize, 1, 1.0, upd, pot, seed=seed, table=table,
                                     profile=True, exact_only=args.exact_only, want_hist=False)
        L = o.lengths.cpu().numpy() - 1
        S = args.steps_per_launch or 512
        full = int(np.min(L)) // S                    # launches in which every track steps all S times
        if full < 1:
            continue
        # kernel_ms covers all launches; the first `full` ones are S steps deep for every wave
        us = o.stats['kernel_ms'] * 1e3 / o.stats['launches'] / S
        best = us if best is None else min(best, us)
    return {'tracks': n, 'us_per_step_lone_wave': best,
            'launch_floor_ms': None if best is None else best * (args.steps_per_launch or 512) / 1e3,
            'what': 'average launch duration / steps per launch of a 16 384-track batch (one wave per CU, no '
                    'histogram): the latency of one step\'s dependent chain, an upper bound (late launches are '
                    'shallower than S steps)'}


def solved_leg(args, movmodel, layers, dem, starts_h, gridsize, res, seed):
    """The same workload on the potential of the build's own solver, outside the
    timed region and on a bounded batch: on this DEM about 44 % of the tracks reach a basin of
    the field, circle there and stop at max_moves = 7.5e6 (reference behaviour, pinned at 50 m by
    tests/golden/g11_wander.npz), so 100k tracks take ~12 s per pass."""
    import torch
    import warnings
    from ssrs_amd.potential import solve_potential
    _, upd = layers.updraft_from_dem(dem, res, 10.0, 270.0, threshold=0.75)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        pot, sst = solve_potential(upd, 0.0, max_iterations=args.solve_iterations, return_stats=True)
    torch.cuda.synchronize()
    t_solve = time.perf_counter() - t0
    n = min(args.solved_tracks, len(starts_h))
    sub = torch.from_numpy(starts_h[:n]).to(dem.device)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    table = build_table(args, movmodel, upd, pot)
    o = movmodel.simulate_tracks(0.0, sub, gridsize, 1, 1.0, upd, pot, seed=seed, table=table, profile=True)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    L = o.lengths.cpu().numpy() - 1
    mm = gridsize[0] // 2 * (gridsize[1] // 2)
    assert int(o.hist.sum().item()) == o.stats['total_steps'] + n, 'histogram checksum failed (solved leg)'
    return {
        'tracks': n, 'tracks_per_s': n / dt, 'steps_per_s': o.stats['total_steps'] / dt, 'seconds': dt,
        'steps_per_track_mean': float(L.mean()), 'steps_per_track_median': float(np.median(L)),
        'steps_per_track_max': int(L.max()), 'share_at_max_moves': float(np.mean(L >= mm)), 'max_moves': mm,
        'launches': o.stats['launches'], 'window_launches': o.stats['window_launches'],
        'tile_launches': o.stats['tile_launches'], 'block_window_launches': o.stats['block_window_launches'],
        'solver': {'iterations': sst['iterations'], 'residual': sst['residual'], 'converged': sst['converged'],
                   'seconds': t_solve, 'rel_tol': 1e-15, 'amg_levels': sst['amg_levels'],
                   'setup_ms': sst.get('setup_ms'), 'workspace_used_gb': round(sst.get('workspace_used', 0) / 1e9, 2),
                   'workspace_reserved_gb': round(sst.get('workspace_bytes', 0) / 1e9, 2)},
        'what': 'first tracks of the same start list through ssrs_potential_solve\'s field (library default '
                'tolerance), one pass, table build included; not part of `value`',
    }


def main():
    args = parse()
    import torch
    import torch.distributed as dist
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        raise SystemExit(f'--gpus {args.gpus} but WORLD_SIZE={world}: launch with '
                         f'torch.distributed.run --nproc-per-node {args.gpus}')
    assert torch.cuda.is_available(), 'bench.py needs a GPU'
    # one rank per GPU; SSRS_BENCH_BACKEND=gloo lets several ranks share one GPU to
    # rehearse the N > 1 code path on a single-GPU box (RCCL refuses duplicate devices)
    backend = os.environ.get('SSRS_BENCH_BACKEND', 'nccl')
    if backend != 'nccl':
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        if backend == 'nccl':
            dist.init_process_group('nccl', rank=rank, world_size=world,
                                    device_id=torch.device('cuda', local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    from ssrs_amd import layers, movmodel, _native
    from ssrs_amd.distributed import shard_range, reduce_histogram
    from ssrs_amd.synthetic import synthetic_dem, ramp_potential
    _native.lib()

    res = args.resolution
    cols = int(round(args.width_km[0] * 1000.0 / res))
    rows = int(round(args.width_km[1] * 1000.0 / res))
    gridsize = (rows, cols)
    ncells = rows * cols
    seed = 30
    n_total = args.tracks * world
    # identical on every rank: replicated rasters, global start list
    dem_h = synthetic_dem(gridsize, res, noise=args.dem_noise)
    np.random.seed(seed)
    srows, scols = movmodel.get_starting_indices(n_total, (5, 55, 1, 2), 'random',
                                                 tuple(args.width_km), res)
    starts_h = np.stack([srows, scols], 1).astype(np.int32)
    lo, hi = shard_range(n_total, rank, world)
    dev = torch.device('cuda', local_rank)
    dem = torch.from_numpy(dem_h).to(dev)
    starts = torch.from_numpy(starts_h[lo:hi]).to(dev)
    if args.potential == 'solve':
        from ssrs_amd.potential import solve_potential
        _, upd0 = layers.updraft_from_dem(dem, res, 10.0, 270.0, threshold=0.75)
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            pot, sst = solve_potential(upd0, 0.0, max_iterations=args.solve_iterations,
                                       return_stats=True)      # library default rel_tol (1e-15)
        pot_label = (f'ssrs_potential_solve (AMG-PCG): {sst["iterations"]} iterations, |r|/|b| = '
                     f'{sst["residual"]:.1e}, {sst["kernel_ms"] / 1e3:.0f} s (outside the timed region)')
        del upd0
    else:
        pot = torch.from_numpy(ramp_potential(gridsize)).to(dev)
        pot_label = ('LABELLED STAND-IN: linear ramp 1000(1-r/(R-1)) (exact solution for '
                     'uniform conductance); the reference spsolve is infeasible at 3e7 cells')
    # two histograms: the RCCL reduce of step i runs under the kernels of step i + 1
    hists = [torch.zeros(gridsize, dtype=torch.int32, device=dev) for _ in range(2 if world > 1 else 1)]
    pending = [None] * len(hists)
    step_no = [0]

    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    acc = dict(raster_ms=0.0, table_ms=0.0, step_kernel_ms=0.0, step_wall_ms=0.0,
               hist_ms=0.0, steps=0, launches=0)

    def one_step(timed):
        slot = step_no[0] % len(hists)
        step_no[0] += 1
        hist = hists[slot]
        if pending[slot] is not None:        # this buffer's previous reduce (two steps ago)
            pending[slot].wait()
            pending[slot] = None
        hist.zero_()
        ev[0].record()
        oro, upd = layers.updraft_from_dem(dem, res, 10.0, 270.0, threshold=0.75)
        ev[1].record()
        table = None if args.direct else build_table(args, movmodel, upd, pot)
        ev[2].record()
        out = movmodel.simulate_tracks(0.0, starts, gridsize, 1, 1.0, upd, pot, seed=seed,
                                       track_id_base=lo, table=table, use_table=not args.direct,
                                       hist=hist, steps_per_launch=args.steps_per_launch,
                                       profile=True, exact_only=args.exact_only,
                                       schedule=not args.no_schedule, binning=not args.no_binning)
        pending[slot] = reduce_histogram(hist, dst=0, async_op=True)
        ev[3].record()
        if timed:
            # simulate_tracks returned after its last launch completed, so the
            # events are final; no device-wide sync (it would wait for the reduce)
            ev[2].synchronize()
            acc['raster_ms'] += ev[0].elapsed_time(ev[1])
            acc['table_ms'] += ev[1].elapsed_time(ev[2])
            acc['step_kernel_ms'] += out.stats['kernel_ms']
            acc['step_wall_ms'] += out.stats['wall_ms']
            acc['hist_ms'] += out.stats['hist_ms']
            acc['steps'] += out.stats['total_steps']
            acc['launches'] += out.stats['launches']
            acc['timed_launches'] = acc.get('timed_launches', 0) + out.stats.get('timed_launches', out.stats['launches'])
            acc['first_move_ms'] = acc.get('first_move_ms', 0.0) + out.stats.get('first_move_ms', 0.0)
        return out

    def drain():
        for i, w in enumerate(pending):
            if w is not None:
                w.wait()
                pending[i] = None

    for _ in range(args.warmup):
        one_step(False)
    drain()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    last = None
    for _ in range(args.steps):
        last = one_step(True)
    drain()                                  # every step's reduce is inside the timed region
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    hist = hists[(step_no[0] - 1) % len(hists)]
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        tot = torch.tensor([acc['steps']], dtype=torch.int64, device=dev)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        total_steps_all = int(tot.item())
    else:
        total_steps_all = acc['steps']

    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    K = args.steps
    lengths = last.lengths.cpu().numpy()
    steps_per_track = float(lengths.mean() - 1)
    # checksum of checksums: every trajectory point was counted exactly once
    assert int(hist.sum().item()) == total_steps_all // K + n_total, 'histogram checksum failed'
    kernel_s = acc['step_kernel_ms'] / 1e3
    first_moves = K if acc.get('first_move_ms', 0.0) > 0 else 0
    main_launches = acc.get('timed_launches', acc['launches']) - first_moves
    main_kernel_ms = acc['step_kernel_ms'] - acc.get('first_move_ms', 0.0)
    # bytes the chosen data path really requests per step (read + 4 B visit / histogram update)
    # (window gathers 18 x 4 + 4; f64 table row 64 + 24 + 4; f64 three candidates 24 + 4; ring triple
    # 12 + 4; threshold dword 4 + 4)
    moved_bytes = 76 if args.direct else (92 if args.exact_only else (28 if args.f64_table else (16 if args.ring_table else 8)))
    achieved = acc['steps'] * STEP_BYTES / kernel_s / 1e9 if kernel_s > 0 else 0.0
    moved_gbps = acc['steps'] * moved_bytes / kernel_s / 1e9 if kernel_s > 0 else 0.0
    raster_s = acc['raster_ms'] / 1e3 / K
    out = {
        'metric': 'simulated tracks/sec (whole node)',
        'value': n_total * K / elapsed,
        'unit': 'tracks/s',
        'n_gpus': world, 'steps': K, 'warmup': args.warmup,
        'ms_per_step': elapsed / K * 1e3,
        'higher_is_better': True,
        'scaling': 'weak',
        'vs_baseline': None,
        'dtype': 'f64',
        'data': 'synthetic',
        'config': {
            'workload': (f'uniform mode, {args.width_km[0]:g}x{args.width_km[1]:g} km @{res:g} m '
                         f'({rows}x{cols} grid), {args.tracks} tracks per GPU, wind 10 m/s @270, '
                         f'threshold 0.75, direction 0, seed 30 (BASELINE.json configs[1])'
                         + ('' if args.dem_noise == 1.5 else f', DEM noise sigma {args.dem_noise:g} m')),
            'tracks_total': n_total,
            'parallelism': f'track-sharded x{world}, replicated rasters'
                           + (f', {"RCCL" if backend == "nccl" else backend} histogram reduce (async, under the next step)'
                              if world > 1 else ''),
            'stepper_path': ('direct 3x3 gathers' if args.direct else
                             ('f64 transition table' if (args.f64_table or args.exact_only)
                              else ('f32 ring table, exact fallback on the raw windows' if args.ring_table else
                                    'threshold table (two 16-bit decision thresholds per cell and last move), '
                                    'exact fallback on the raw windows'))),
            'potential': pot_label,
        },
        'steps_per_s': total_steps_all / elapsed,
        'steps_per_track_mean': steps_per_track,
        'steps_per_track_max': int(lengths.max() - 1),
        'raster_mcells_per_s': ncells / raster_s / 1e6 if raster_s > 0 else None,
        'raster_gbps': ncells * RASTER_BYTES_PER_CELL / raster_s / 1e9 if raster_s > 0 else None,
        'phase_ms_per_step': {
            'raster_k1': acc['raster_ms'] / K, 'table_k2a': acc['table_ms'] / K,
            'stepper_kernels_k2b': acc['step_kernel_ms'] / K,
            'histogram_binning_k3': acc['hist_ms'] / K,
            'stepper_wall': acc['step_wall_ms'] / K,
        },
        'roofline': {
            'kernel': ('k_step_tracks' if (args.direct or args.f64_table or args.exact_only)
                       else ('k_step_lean<ring>' if args.ring_table else 'k_step_thr')) + ' (K2 stepper, rank 0)',
            # what bounds it: the dependent chain of one step (see dependent_chain below and
            # profiles/r02_stepper_chain.md), not HBM: the kernel moves 8 B per step
            'bound': 'latency',
            # achieved = bytes the shipped data path really requests per step (the 4-byte
            # threshold entry + the 4-byte visit) x steps / sum of the stepper launch durations
            'achieved': moved_gbps, 'peak': HBM_PEAK_GBPS, 'unit': 'GB/s',
            'frac': moved_gbps / HBM_PEAK_GBPS,
            'traffic': None,
            'bytes_per_step': moved_bytes,
            # launches of THIS kernel per bench step and their average duration (HIP events on the launch
            # stream): the one-iteration first-move launch of the generic kernel is counted apart, so the
            # average is the one rocprofv3 reports for the kernel (profiles/r02_final_kernel_stats.md)
            'launches': main_launches // K,
            'avg_launch_ms': main_kernel_ms / max(main_launches, 1),
            'avg_bytes_per_launch': acc['steps'] * moved_bytes / max(main_launches, 1),
            'first_move_launch_ms': acc.get('first_move_ms', 0.0) / K,
            # SURVEY 8(d)'s 76 B/step is the gather volume of the REFERENCE's formulation (18 window
            # reads + 1 point); the shipped path precomputes the windows into a table, so this
            # figure can exceed the peak and is not a roofline fraction
            'model_bytes_per_step': STEP_BYTES,
            'model_gbps': achieved,
            'model_frac': achieved / HBM_PEAK_GBPS,
        },
    }
    # HBM bytes per launch from the PMC passes (profiles/, separate --pmc runs): only quoted when
    # THIS run is the configuration those passes profiled
    pmc = os.path.join(ROOT, 'profiles', 'pmc_traffic.json')
    default_variant = (world == 1 and args.potential == 'ramp' and not (args.direct or args.f64_table or args.exact_only or args.ring_table
                       or args.no_binning or args.no_schedule) and args.tracks == 100_000 and res == 10.0
                       and args.steps_per_launch in (0, 512) and args.dem_noise == 1.5)
    if os.path.exists(pmc) and default_variant:
        try:
            with open(pmc) as f:
                rec = json.load(f)
            out['roofline']['traffic'] = rec.get('k_step_tracks_bytes_per_launch')
            out['roofline']['traffic_source'] = ('profile constant (not counted in this run): '
                                                 + str(rec.get('source')))
        except Exception:
            pass
    if world == 1 and not args.direct and not args.no_chain_probe:
        out['roofline']['dependent_chain'] = chain_probe(args, movmodel, layers, dem, pot, starts_h, gridsize, res, seed)
    if world == 1 and args.solved_tracks > 0 and args.potential != 'solve':
        out['solved_potential'] = solved_leg(args, movmodel, layers, dem, starts_h, gridsize, res, seed)
    if world == 1 and args.cpu_seconds > 0:
        cpu, run, m = cpu_baseline(args, gridsize, dem_h, pot.cpu().numpy(), starts_h, seed,
                                   steps_per_track)
        out['cpu_baseline'] = cpu
        # same tracks, same streams: the GPU's lengths for the sample must agree
        # unless the usable-updraft rasters differ in the last f64 bits
        out['cpu_baseline']['sample_lengths_equal_gpu'] = bool(
            np.array_equal(run['lengths'], lengths[:m]))
    print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
