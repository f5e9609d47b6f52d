#!/usr/bin/env python3
"""Headline benchmark (BASELINE.json): simulated tracks/s + updraft-raster
Mcells/s on the 60x50 km @10 m uniform-mode raster (5000 x 6000 cells),
100k tracks per MI355X, through the potential field of the build's own solver
("from the build's solver if landed"; /root/reference/ssrs/simulator.py:259-288
computes it before the tracks are mapped, :360-369).

One "step" = one pass of the hot path over one batch of synthetic input, with
every input already resident in HBM:
    DEM --K1 fused raster--> orograph f32 + usable updraft f64
        --K2a--> per-cell transition table (updraft x potential)
        --K2b/K3--> 100k tracks stepped to completion + uint32 presence histogram
        [N > 1: one RCCL sum-reduce of the histogram to rank 0, widened to 64 bits when needed]
The potential is solved ONCE, outside the timed region (the reference caches it on disk,
simulator.py:266-272).  On this field ~44 % of the tracks reach a basin and circle there until
max_moves = 7.5e6 (reference behaviour, tests/golden/g11_wander.npz): a pass is ~2.6e11 steps, and the
figure of merit beside tracks/s is steps/s.  The linear-ramp stand-in potential of rounds 1-2
(a batch that crosses the raster as one front, 4 850 steps per track) is kept as the labelled
secondary object `stand_in`.

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
# steps/s of the stepper once the GPU is FULL of waves: what a latency-bound batch of 100k tracks is
# measured against (`throughput_frac`).  Front-shaped batches (k_step_thr<4>): 1 M tracks per GPU on the
# ramp, every SIMD holding several waves (profiles/r02_scale_tracks.txt, 40.6 M tracks/s x 4 850 steps).
# Roaming batches (k_step_roam, one block of 144 KB LDS per CU): the best the same kernel reaches in one call, 260 000 tracks
# (512-lane blocks while ~105 000 roam, 256-lane blocks for the ~80 000 that stay to max_moves; profiles/r04_roam_fill.txt:
# 3.64e11; rounds 3-4 quoted 2.84e11 / 3.1e11 for 140 000 tracks in 256-lane blocks).  A pass takes the time of its longest
# track chain whatever the number of tracks, so this is what the same 1.35 s could carry, not something a 100k-track pass
# can reach.
THROUGHPUT_BOUND_STEPS_PER_S = 2.0e11
ROAM_THROUGHPUT_BOUND_STEPS_PER_S = 3.6e11       # profiles/r04_roam_fill.txt: 260 000 tracks per call (3.1e11 before the 512-lane blocks)
# Independent yardsticks of k_step_roam (not the builder's own kernel at another batch size):
#  * VALU issue: a wave64 instruction holds its SIMD16 for 4 clocks, so 256 CUs x 4 SIMDs issue 2.4e9 / 4 x 1024 wave-instructions
#    per second; the kernel spends 118.6 of them per wave-pair (SQ_INSTS_VALU per launch / wave-pairs per launch,
#    profiles/r04_solved_counters.md), and a wave-pair is 2 x (live lanes of the wave) steps -- live lanes from this run's stats
#  * the CU's gather path: a fully divergent dwordx4 gather that hits L2 is served at 0.448 lane-loads per clock and CU however
#    many waves ask (tools/microbench/gather_latency.hip, profiles/r04_gather_latency.txt); one gather per lane and pair of steps
GPU_CLOCK_HZ, GPU_CUS = 2.4e9, 256
ROAM_VALU_PER_WAVE_PAIR = 118.6
GATHER_LANES_PER_CLK_PER_CU = 0.448


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=3)
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--tracks', type=int, default=100_000, help='tracks per GPU')
    ap.add_argument('--hist-safe-tracks', type=int, default=250_000,
                    help='tracks per sub-batch of a pass (Config.hist_safe_tracks: a uint32 histogram is safe for this many on the '
                         'solved field, and their ~105 000 roaming tracks are one round of 512-lane blocks)')
    ap.add_argument('--resolution', type=float, default=10.0)
    ap.add_argument('--width-km', type=float, nargs=2, default=(60.0, 50.0))
    ap.add_argument('--direct', action='store_true', help='3x3 window gathers, no table')
    ap.add_argument('--steps-per-launch', type=int, default=0)
    ap.add_argument('--cpu-seconds', type=float, default=20.0,
                    help='target CPU time of the cpu_baseline sample (0 = skip)')
    ap.add_argument('--cpu-cap', type=int, default=60_000,
                    help='max_moves of the cpu_baseline sample on the solved field (the GPU repeats the '
                         'sample under the same cap for the in-run parity check)')
    ap.add_argument('--potential', default='solve', choices=['ramp', 'solve'])
    ap.add_argument('--solve-iterations', type=int, default=2000)
    ap.add_argument('--stand-in-steps', type=int, default=10,
                    help='timed passes of the ramp stand-in leg (after the timed region; 0 = skip)')
    ap.add_argument('--full-chip-tracks', type=int, default=250_000,
                    help='tracks of the `full_chip` leg: ONE pass of that many tracks on the solved field after the timed region '
                         '(what the stepper carries once the batch fills the SIMDs; default-variant runs at N=1 only; 0 = skip)')
    ap.add_argument('--no-chain-probe', action='store_true',
                    help='skip the 16 384-track dependent-chain measurement (profiling runs: keeps per-kernel averages clean)')
    ap.add_argument('--ref-cpu-seconds', type=float, default=2.0,
                    help='seconds of stepping of the reference-equivalent (numpy restatement) CPU rate (0 = skip)')
    ap.add_argument('--no-binning', action='store_true', help='per-step global atomics for the histogram')
    ap.add_argument('--no-schedule', action='store_true', help='disable the coherent schedule')
    ap.add_argument('--exact-only', action='store_true', help='disable the fast decision path')
    ap.add_argument('--f64-table', action='store_true', help='8 x f64 transition table instead of the threshold table')
    ap.add_argument('--ring-table', action='store_true', help='f32 ring table (round 1 stepper) instead of the threshold table')
    ap.add_argument('--dem-noise', type=float, default=1.5,
                    help='sigma of the per-cell DEM noise in metres (SURVEY 8(d) prescribes 1.5 at every resolution)')
    return ap.parse_args()


def host_cpu():
    """CPU model, physical cores and hardware threads of the box (printed beside the baseline)."""
    model, cores = None, set()
    try:
        phys = core = None
        with open('/proc/cpuinfo') as f:
            for line in f:
                if line.startswith('model name') and model is None:
                    model = line.split(':', 1)[1].strip()
                elif line.startswith('physical id'):
                    phys = line.split(':', 1)[1].strip()
                elif line.startswith('core id'):
                    core = line.split(':', 1)[1].strip()
                elif not line.strip():
                    if phys is not None and core is not None:
                        cores.add((phys, core))
                    phys = core = None
    except OSError:
        pass
    threads = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    return model, (len(cores) or None), threads


def cpu_baseline(args, gridsize, dem_h, oro_gpu_h, upd_gpu_h, pot_h, starts, seed, cap):
    """Oracle (C port of the reference algorithm, OpenMP) on the host cores, on a bounded sample of
    the same workload: the full-grid raster chain once and the first M tracks (same global ids /
    Philox streams as the GPU run), every track capped at `cap` moves when given (on the solved
    field a track that reaches a basin would otherwise take 7.5e6: ~4 core-minutes each).  The
    stepping sample reads the GPU's own usable-updraft raster, so that its lengths and histogram
    can be compared bit for bit with a GPU run of the same sample under the same cap."""
    from oracle import c_oracle
    model, phys_cores, threads = host_cpu()
    t0 = time.perf_counter()
    slope, aspect = c_oracle.slope_aspect(dem_h, args.resolution)
    _, oro32 = c_oracle.orographic(slope, aspect, 10.0, 270.0)
    upd = c_oracle.threshold(oro32, 0.75)
    t_raster = time.perf_counter() - t0
    # the f32 orograph is the file contract (simulator.py:198); the usable updraft is exp() of it in
    # f64, where the device's and the host's libm differ in the last bit of a quarter of the cells
    same_oro = float(np.mean(oro32 == oro_gpu_h))
    upd_rel = float(np.max(np.abs(upd - upd_gpu_h) / np.maximum(np.abs(upd), 1e-300)))
    del slope, aspect, oro32, upd
    # calibrate, then size the sample for ~cpu_seconds of stepping
    t0 = time.perf_counter()
    cal = c_oracle.simulate_tracks(0.0, starts[:2 * threads], gridsize, 1, 1.0, upd_gpu_h, pot_h,
                                   seed=seed, want_traj=False, want_hist=True, nthreads=threads, max_moves=cap)
    t_cal = max(time.perf_counter() - t0, 1e-6)
    rate = cal['steps'] / t_cal
    per_track = max(cal['steps'] / (2 * threads), 1.0)
    m = int(min(len(starts), max(2 * threads, rate * args.cpu_seconds / per_track)))
    t0 = time.perf_counter()
    run = c_oracle.simulate_tracks(0.0, starts[:m], gridsize, 1, 1.0, upd_gpu_h, pot_h, seed=seed,
                                   want_traj=False, want_hist=True, nthreads=threads, max_moves=cap)
    t_run = time.perf_counter() - t0
    ref_equiv = None
    if args.ref_cpu_seconds > 0:
        # the reference's own arithmetic speed: the numpy restatement (oracle/ssrs_oracle.py, the
        # same per-step numpy calls as /root/reference/ssrs/movmodel.py:264-318, which cannot travel)
        from oracle import ssrs_oracle as orc
        from oracle.philox import TrackUniforms
        t0 = time.perf_counter()
        nsteps = ntr = 0
        ref_cap = min(cap or 20_000, 20_000)
        while time.perf_counter() - t0 < args.ref_cpu_seconds and ntr < m:
            tr = orc.generate_simulated_tracks(0.0, (int(starts[ntr, 0]), int(starts[ntr, 1])), gridsize, 1, 1.0,
                                               upd_gpu_h, pot_h, uniform=TrackUniforms(seed, ntr), max_moves=ref_cap)
            nsteps += len(tr) - 1
            assert len(tr) == min(int(run['lengths'][ntr]), ref_cap + 1), 'numpy restatement and C port disagree'
            ntr += 1
        t_ref = time.perf_counter() - t0
        ref_equiv = {'steps_per_s_per_core': nsteps / t_ref, 'cores': 1, 'tracks': ntr,
                     'steps': nsteps, 'seconds': t_ref, 'max_moves_cap': ref_cap,
                     'what': 'numpy restatement of generate_simulated_tracks, one process, first tracks of '
                             'this workload; SURVEY measured 12.1 k steps/s/core for the reference itself'}
    return {
        'value': m / t_run, 'unit': 'tracks/s', 'cores': threads, 'kind': 'port',
        'cpu_model': model, 'physical_cores': phys_cores, 'hardware_threads_used': threads,
        'max_moves_cap': cap,
        'reference_equivalent': ref_equiv,
        'sample': (f'C/OpenMP oracle port, first {m} of the {len(starts)} tracks of this workload'
                   + (f', every track capped at {cap} moves' if cap else '')
                   + f' ({run["steps"]} steps in {t_run:.1f} s) on {threads} host threads'
                   + (f' ({phys_cores} physical cores, {model})' if phys_cores else '')
                   + f'; raster chain on the full grid once ({t_raster:.1f} s)'),
        'steps_per_s': run['steps'] / t_run,
        'steps_per_s_per_core': run['steps'] / t_run / threads,
        'raster_mcells_per_s': gridsize[0] * gridsize[1] / t_raster / 1e6,
        'orograph_f32_cells_identical_to_gpu': same_oro,
        'usable_updraft_max_rel_diff_vs_gpu': upd_rel,
    }, run, m


def build_table(args, movmodel, upd, pot):
    if args.f64_table or args.exact_only:
        return movmodel.build_transition_table(upd, pot)
    if args.ring_table:
        return movmodel.build_transition_table(upd, pot, ring=True)
    return movmodel.build_transition_table(upd, pot, thr=True, move_dirn=0.0)


def chain_probe(args, movmodel, layers, dem, pot, starts_h, gridsize, res, seed):
    """What bounds the stepper: one step of a track is a chain of dependent instructions
    (Philox -> decision -> next address -> gather -> ...).  A batch of 16 384 tracks is
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
            'potential': 'ramp stand-in',
            'what': 'average launch duration / steps per launch of a 16 384-track batch (one wave per CU, no '
                    'histogram): the latency of one step\'s dependent chain, an upper bound (late launches are '
                    'shallower than S steps)'}


class _MergedPass:
    """The sub-batches of one pass as one result: lengths in track order, stats added up."""

    def __init__(self, outs):
        import torch
        self.lengths = torch.cat([o.lengths for o in outs])
        keys = set().union(*[o.stats.keys() for o in outs])
        self.stats = {}
        for k in keys:
            vals = [o.stats.get(k, 0) for o in outs]
            if all(isinstance(v, (int, float)) and not isinstance(v, bool) for v in vals):
                self.stats[k] = sum(vals)
            else:
                self.stats[k] = vals[0]


class Passes:
    """K timed passes of the hot path on one potential field (the main leg and the stand-in leg)."""

    def __init__(self, args, mods, dem, pot, starts, lo, gridsize, res, seed, world, dev):
        import torch
        self.args, self.dem, self.pot, self.starts, self.lo = args, dem, pot, starts, lo
        self.layers, self.movmodel, self.reduce_histogram = mods
        self.gridsize, self.res, self.seed, self.world = gridsize, res, seed, world
        # two histograms: the RCCL reduce of step i runs under the kernels of step i + 1
        self.hists = [torch.zeros(gridsize, dtype=torch.int32, device=dev) for _ in range(2 if world > 1 else 1)]
        self.pending = [None] * len(self.hists)
        self.reduced = [None] * len(self.hists)
        self.step_no = 0
        self.ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        # K1 through the C ABI with its arguments prepared once and its two output rasters reused: between the
        # start event and the launch the host does ~5 us of work, so both are enqueued while the stream is
        # still zeroing the histogram (~25 us) and ev[0] -> ev[1] is the kernel's own duration in every leg
        # (round 3's figure of the headline leg was the host's launch latency on an idle stream)
        import ctypes as C
        from ssrs_amd import _native as nat
        from ssrs_amd._device import stream_ptr
        self.k1_out = (torch.empty(gridsize, dtype=torch.float32, device=dev),
                       torch.empty(gridsize, dtype=torch.float64, device=dev))
        k1_args = (nat.ptr(dem), nat.SSRS_F64, C.c_double(res), C.c_double(10.0), C.c_double(270.0), C.c_double(0.0),
                   nat.ptr(self.k1_out[0]), C.c_double(0.75), nat.ptr(self.k1_out[1]), int(gridsize[0]), int(gridsize[1]))
        assert dem.dtype == torch.float64 and dem.is_contiguous()
        fn, check = nat.lib().ssrs_updraft_from_dem, nat.check
        self.k1 = lambda: check(fn(*k1_args, stream_ptr()))
        self.acc = dict(raster_ms=0.0, table_ms=0.0, step_kernel_ms=0.0, step_wall_ms=0.0, hist_ms=0.0, steps=0,
                        launches=0, timed_launches=0, first_move_ms=0.0, block_window_ms=0.0, block_window_timed=0,
                        block_window_steps=0, block_window_launches=0, window_launches=0, tile_launches=0,
                        wander_sorts=0, roam_launches=0, roam_wave_pairs=0, roam_slow_wave_pairs=0, roam_fine_settled=0)

    def one_step(self, timed):
        import torch
        args, ev, acc = self.args, self.ev, self.acc
        slot = self.step_no % len(self.hists)
        self.step_no += 1
        hist = self.hists[slot]
        if self.pending[slot] is not None:        # this buffer's previous reduce (two steps ago)
            self.pending[slot].wait()
            self.pending[slot] = None
        hist.zero_()
        ev[0].record()
        self.k1()                                 # = layers.updraft_from_dem(dem, res, 10, 270, threshold=0.75, out=k1_out)
        ev[1].record()
        oro, upd = self.k1_out
        table = None if args.direct else build_table(args, self.movmodel, upd, self.pot)
        ev[2].record()
        n = int(self.starts.shape[0])
        # equal sub-batches of at most --hist-safe-tracks tracks (a last short one would take a full pass's time --
        # a pass lasts as long as its longest track chain -- for a fraction of the work)
        nsub = max(1, -(-n // max(int(args.hist_safe_tracks), 1)))
        safe = max(1, -(-n // nsub))
        outs = []
        hist64 = None
        # a sub-batch of more than ~240 000 tracks takes the trap cells of the solved field past 2^32 visits (1.7e4 per track):
        # sub-batches larger than 100 000 tracks count in 64 bits inside the library (ssrs_tracks_simulate_h64: the kernels'
        # uint32 raster is emptied into the 64-bit one every other batch), as ssrs_amd.Simulator does; smaller ones count
        # into the uint32 raster and are added up in 64 bits here when there are several (Config.hist_safe_tracks)
        in_lib = safe > 100_000 and not args.direct
        for b0 in range(0, n, safe):
            if n > safe or in_lib:
                if hist64 is None:
                    hist64 = torch.zeros(self.gridsize, dtype=torch.int64, device=hist.device)
                if not in_lib:
                    hist.zero_()
            o = self.movmodel.simulate_tracks(0.0, self.starts[b0:b0 + safe], self.gridsize, 1, 1.0, upd, self.pot, seed=self.seed,
                                              track_id_base=self.lo + b0, table=table, use_table=not args.direct,
                                              hist=hist64 if in_lib else hist, hist64=in_lib, steps_per_launch=args.steps_per_launch,
                                              profile=True, exact_only=args.exact_only,
                                              schedule=not args.no_schedule, binning=not args.no_binning)
            outs.append(o)
            if hist64 is not None and not in_lib:
                hist64 += hist.to(torch.int64) & 0xFFFFFFFF
        out = outs[0] if len(outs) == 1 else _MergedPass(outs)
        # (widens to 64 bits by itself when the ranks' largest counts could wrap a 32-bit sum)
        self.pending[slot] = self.reduce_histogram(hist if hist64 is None else hist64, dst=0, async_op=True)
        self.reduced[slot] = self.pending[slot].result if self.pending[slot] is not None else (hist if hist64 is None else hist64)
        ev[3].record()
        if timed:
            # simulate_tracks returned after its last launch completed, so the
            # events are final; no device-wide sync (it would wait for the reduce)
            ev[2].synchronize()
            acc['raster_ms'] += ev[0].elapsed_time(ev[1])
            acc['table_ms'] += ev[1].elapsed_time(ev[2])
            st = out.stats
            acc['step_kernel_ms'] += st['kernel_ms']
            acc['step_wall_ms'] += st['wall_ms']
            acc['hist_ms'] += st['hist_ms']
            acc['steps'] += st['total_steps']
            for k in ('launches', 'timed_launches', 'first_move_ms', 'block_window_ms', 'block_window_timed',
                      'block_window_steps', 'block_window_launches', 'window_launches', 'tile_launches', 'wander_sorts',
                      'roam_launches', 'roam_wave_pairs', 'roam_slow_wave_pairs', 'roam_fine_settled'):
                acc[k] += st.get(k, 0)
        return out

    def k1_kernel_seconds(self, launches=20):
        """K1's own duration: `launches` back-to-back launches between two events on a busy stream (the per-pass
        figure `raster_ms` brackets ONE launch that follows a host-synchronised stepper call: launch latency and
        the clocks' ramp after the idle gap are in it, 0.15 against 0.126 ms)."""
        import torch
        for _ in range(3):
            self.k1()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(launches):
            self.k1()
        e1.record()
        e1.synchronize()
        return e0.elapsed_time(e1) / 1e3 / launches

    def drain(self):
        for i, w in enumerate(self.pending):
            if w is not None:
                w.wait()
                self.pending[i] = None

    def run(self, steps, warmup):
        import torch
        import torch.distributed as dist
        for _ in range(warmup):
            self.one_step(False)
        self.drain()
        if self.world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        last = None
        for _ in range(steps):
            last = self.one_step(True)
        self.drain()                                  # every step's reduce is inside the timed region
        if self.world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        self.last_hist = self.reduced[(self.step_no - 1) % len(self.hists)]
        return elapsed, last


def stepper_roofline(args, acc, K, solved):
    """`roofline` of the pass's dominant kernel, from HIP events the library records around every
    stepper launch on the launch stream (SSRS_TRACKS_PROFILE)."""
    kernel_s = acc['step_kernel_ms'] / 1e3
    # bytes the chosen data path really requests per step (window gathers 18 x 4 + 4; f64 table row
    # 64 + 24 + 4; f64 three candidates 24 + 4; ring triple 12 + 4; threshold dword 4 + a 4-byte
    # visit / 4 and an LDS add in block-window launches)
    table_path = not (args.direct or args.f64_table or args.exact_only or args.ring_table)
    moved_bytes = 76 if args.direct else (92 if args.exact_only else (28 if args.f64_table else (16 if args.ring_table else 8)))
    bw_share = acc['block_window_ms'] / acc['step_kernel_ms'] if acc['step_kernel_ms'] > 0 else 0.0
    bound_sps = THROUGHPUT_BOUND_STEPS_PER_S
    if solved and table_path and bw_share > 0.5:
        # the pass is dominated by the block-window launches of the roaming survivors; the visits are counted in LDS
        ms, steps, n = acc['block_window_ms'], acc['block_window_steps'], acc['block_window_timed']
        if acc.get('roam_launches', 0) * 2 > acc['block_window_launches']:
            # pair table: one 16-byte entry per PAIR of moves
            bytes_per_step = 8
            name = 'k_step_roam<true> (K2 stepper of roaming batches: pair table, two moves per 16-byte gather, block histogram windows in LDS; rank 0)'
            bound_sps = ROAM_THROUGHPUT_BOUND_STEPS_PER_S
        else:
            bytes_per_step = 4          # one 4-byte threshold entry per move
            name = 'k_step_thr<6, false, true> (K2 stepper, block histogram windows; rank 0)'
    else:
        first_moves = K if acc['first_move_ms'] > 0 else 0
        n = acc['timed_launches'] - first_moves
        ms = acc['step_kernel_ms'] - acc['first_move_ms']
        steps = acc['steps']
        bytes_per_step = moved_bytes
        name = ('k_step_tracks' if (args.direct or args.f64_table or args.exact_only)
                else ('k_step_lean<ring>' if args.ring_table else 'k_step_thr<4, true, false>')) + ' (K2 stepper, rank 0)'
    sec = ms / 1e3
    extra = {}
    if 'k_step_roam' in name and acc.get('roam_wave_pairs', 0) > 0 and steps > 0 and sec > 0:
        lanes = steps / 2.0 / acc['roam_wave_pairs']                  # live lanes of an average wave
        issue = GPU_CLOCK_HZ / 4.0 * 4 * GPU_CUS / ROAM_VALU_PER_WAVE_PAIR * lanes * 2.0
        gather = GATHER_LANES_PER_CLK_PER_CU * GPU_CUS * GPU_CLOCK_HZ * 2.0
        extra = {'live_lanes_per_wave': lanes,
                 'valu_issue_bound_steps_per_s': issue, 'valu_issue_frac': steps / sec / issue,
                 'valu_issue_bound_at_64_lanes_steps_per_s': issue / lanes * 64.0,
                 'gather_path_bound_steps_per_s': gather, 'gather_path_frac': steps / sec / gather,
                 'bounds_source': 'VALU: 118.6 wave-VALU per wave-pair (profiles/r04_solved_counters.md) at 1 wave-instruction per 4 '
                                  'clocks and SIMD; gather path: 0.448 lane-loads per clock and CU (profiles/r04_gather_latency.txt); '
                                  'live lanes from this run'}
    gbps = steps * bytes_per_step / sec / 1e9 if sec > 0 else 0.0
    model = steps * STEP_BYTES / sec / 1e9 if sec > 0 else 0.0
    sps = steps / sec if sec > 0 else 0.0
    return {
        'kernel': name,
        # what bounds it: the dependent chain of one step (divergent gather -> decode -> next address) in
        # waves that are alone on their SIMDs, not HBM (the tables stay in L2: TCC hit rate 99.8 %)
        'bound': 'latency',
        # achieved = bytes the shipped data path really requests per step x steps / sum of this
        # kernel's launch durations
        'achieved': gbps, 'peak': HBM_PEAK_GBPS, 'unit': 'GB/s', 'frac': gbps / HBM_PEAK_GBPS,
        'traffic': None,
        'bytes_per_step': bytes_per_step,
        'launches': n // max(K, 1), 'avg_launch_ms': ms / max(n, 1),
        'avg_bytes_per_launch': steps * bytes_per_step / max(n, 1),
        'share_of_stepper_time': (ms / acc['step_kernel_ms']) if acc['step_kernel_ms'] > 0 else None,
        'steps_per_s_in_kernel': sps,
        # the number that can approach 1: steps/s of this kernel against the stepper's measured rate
        # with the GPU full of waves (a latency-bound batch leaves SIMDs idle)
        'throughput_frac': sps / bound_sps,
        'throughput_bound_steps_per_s': bound_sps,
        'first_move_launch_ms': acc['first_move_ms'] / max(K, 1),
        # SURVEY 8(d)'s 76 B/step is the gather volume of the REFERENCE's formulation (18 window
        # reads + 1 point); the shipped path precomputes the windows into a table, so this
        # figure can exceed the peak and is not a roofline fraction
        'model_bytes_per_step': STEP_BYTES, 'model_gbps': model, 'model_frac': model / HBM_PEAK_GBPS,
        **extra,
    }, kernel_s


def attach_traffic(roof, variant_ok):
    """HBM bytes per launch from the PMC passes (profiles/, separate --pmc runs): only quoted when
    THIS run is the configuration those passes profiled, and labelled as a profile constant."""
    pmc = os.path.join(ROOT, 'profiles', 'pmc_traffic.json')
    if not (os.path.exists(pmc) and variant_ok):
        return
    try:
        with open(pmc) as f:
            rec = json.load(f)
        # {"solved": <summarize_pmc.py record of a solved-field pass>, "ramp": <... of the stand-in>}
        solved = 'k_step_roam' in roof['kernel'] or '<6' in roof['kernel']
        ent = rec.get('solved' if solved else 'ramp') or {}
        want = 'k_step_roam' if 'k_step_roam' in roof['kernel'] else ('k_step_thr<6' if solved else 'k_step_thr<4')
        for kname, k in (ent.get('kernels') or {}).items():
            if want in kname:
                roof['traffic'] = k.get('corrected_bytes')
                roof['traffic_kernel'] = kname
                roof['traffic_source'] = 'profile constant (not counted in this run): ' + str(ent.get('source'))
                break
    except Exception:
        pass


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
    max_moves = int(np.ceil(rows / 2 * cols / 2))
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
    solved = args.potential == 'solve'
    ramp = torch.from_numpy(ramp_potential(gridsize)).to(dev)
    ramp_label = ('LABELLED STAND-IN: linear ramp 1000(1-r/(R-1)) (exact solution for '
                  'uniform conductance); the reference spsolve is infeasible at 3e7 cells')
    solver = None
    if solved:
        from ssrs_amd.potential import solve_potential
        _, upd0 = layers.updraft_from_dem(dem, res, 10.0, 270.0, threshold=0.75)
        import warnings
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            pot, sst = solve_potential(upd0, 0.0, max_iterations=args.solve_iterations,
                                       return_stats=True)      # library default rel_tol (1e-15)
        torch.cuda.synchronize()
        t_solve = time.perf_counter() - t0
        pot_label = (f'ssrs_potential_solve (K5, aggregation AMG + PCG, default tolerance 1e-15): {sst["iterations"]} '
                     f'iterations, |r|/|b| = {sst["residual"]:.1e}, {t_solve:.1f} s, once, outside the timed region '
                     f'(the reference caches the field on disk, simulator.py:266-272)')
        solver = {'iterations': sst['iterations'], 'residual': sst['residual'], 'converged': sst['converged'],
                  'seconds': t_solve, 'kernel_seconds': sst['kernel_ms'] / 1e3, 'setup_seconds': (sst.get('setup_ms') or 0.0) / 1e3,
                  'seconds_what': 'wall time of the call, first use of a 33 GB workspace included (hipMalloc); kernel_seconds = '
                                  'the Krylov iteration between two HIP events, setup_seconds = the AMG hierarchy',
                  'rel_tol': 1e-15, 'amg_levels': sst['amg_levels'], 'setup_ms': sst.get('setup_ms'),
                  'workspace_used_gb': round(sst.get('workspace_used', 0) / 1e9, 2),
                  'workspace_reserved_gb': round(sst.get('workspace_bytes', 0) / 1e9, 2)}
        del upd0
    else:
        pot, pot_label = ramp, ramp_label

    mods = (layers, movmodel, reduce_histogram)
    main_leg = Passes(args, mods, dem, pot, starts, lo, gridsize, res, seed, world, dev)
    elapsed, last = main_leg.run(args.steps, args.warmup)
    acc = main_leg.acc
    hist = main_leg.last_hist
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
    L = lengths.astype(np.int64) - 1
    # checksum of checksums: every trajectory point was counted exactly once (the widened sum of a
    # multi-rank run is int64; a single rank's uint32 counts are summed as such)
    hsum = int(hist.sum().item()) if hist.dtype == torch.int64 else int((hist.view(torch.int32).to(torch.int64) & 0xFFFFFFFF).sum().item())
    assert hsum == total_steps_all // K + n_total, 'histogram checksum failed'
    roof, kernel_s = stepper_roofline(args, acc, K, solved)
    raster_s = main_leg.k1_kernel_seconds()
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
                           + (f', {"RCCL" if backend == "nccl" else backend} histogram reduce (async, under the next step; '
                              f'64-bit when the ranks\' counts could wrap 32)' if world > 1 else ''),
            'stepper_path': ('direct 3x3 gathers' if args.direct else
                             ('f64 transition table' if (args.f64_table or args.exact_only)
                              else ('f32 ring table, exact fallback on the raw windows' if args.ring_table else
                                    'threshold table (two 16-bit decision thresholds per cell and last move); once the batch roams, '
                                    'the pair table (a state\'s thresholds and its three successors\', two moves per 16-byte gather) '
                                    'and a 32-bit table for near-ties; exact fallback on the raw windows'))),
            'potential': pot_label,
        },
        # tracks/s depends on the terrain through the track lengths: the comparable figure is steps/s
        'steps_per_s': total_steps_all / elapsed,
        'steps_per_track_mean': float(L.mean()),
        'steps_per_track_median': float(np.median(L)),
        'steps_per_track_max': int(L.max()),
        'share_at_max_moves': float(np.mean(L >= max_moves)), 'max_moves': max_moves,
        'launches_per_step': {'all': acc['launches'] // K, 'row_window': acc['window_launches'] // K,
                              'tile_buckets': acc['tile_launches'] // K, 'block_windows': acc['block_window_launches'] // K,
                              'wander_sorts': acc['wander_sorts'] // K, 'pair_table': acc['roam_launches'] // K},
        'roam': {'wave_pairs': acc['roam_wave_pairs'] // K, 'slow_wave_pairs': acc['roam_slow_wave_pairs'] // K,
                 'near_ties_settled_at_32_bits': acc['roam_fine_settled'] // K,
                 'what': 'pair-table launches: pairs of moves run by their waves; how many sent a lane through the '
                         'single-move sequence (near-ties, flag entries); near-ties the 32-bit fine table settled'},
        'raster_mcells_per_s': ncells / raster_s / 1e6 if raster_s > 0 else None,
        'raster_gbps': ncells * RASTER_BYTES_PER_CELL / raster_s / 1e9 if raster_s > 0 else None,
        'raster_timing': 'K1 kernel duration: 20 back-to-back launches between two HIP events (phase_ms_per_step.raster_k1 is the '
                         'one launch inside a pass, host launch latency included)',
        'phase_ms_per_step': {
            'raster_k1': acc['raster_ms'] / K, 'table_k2a': acc['table_ms'] / K,
            'stepper_kernels_k2b': acc['step_kernel_ms'] / K,
            'histogram_binning_k3': acc['hist_ms'] / K,
            'stepper_wall': acc['step_wall_ms'] / K,
        },
        'roofline': roof,
    }
    if solver:
        out['solver'] = solver
    default_variant = (world == 1 and not (args.direct or args.f64_table or args.exact_only or args.ring_table
                       or args.no_binning or args.no_schedule) and args.tracks == 100_000 and res == 10.0
                       and args.steps_per_launch in (0, 512) and args.dem_noise == 1.5)
    attach_traffic(roof, default_variant)
    if world == 1 and not args.direct and not args.no_chain_probe:
        roof['dependent_chain'] = chain_probe(args, movmodel, layers, dem, ramp, starts_h, gridsize, res, seed)
    if world == 1 and solved and args.stand_in_steps > 0:
        # the round-1/2 headline, kept so that the series survives: the same pass on the ramp
        leg = Passes(args, mods, dem, ramp, starts, lo, gridsize, res, seed, world, dev)
        el2, last2 = leg.run(args.stand_in_steps, 2)
        k2 = args.stand_in_steps
        roof2, _ = stepper_roofline(args, leg.acc, k2, False)
        attach_traffic(roof2, default_variant)
        l2 = last2.lengths.cpu().numpy().astype(np.int64) - 1
        out['stand_in'] = {
            'potential': ramp_label, 'steps': k2, 'warmup': 2,
            'value': n_total * k2 / el2, 'unit': 'tracks/s', 'ms_per_step': el2 / k2 * 1e3,
            'steps_per_s': leg.acc['steps'] / el2, 'steps_per_track_mean': float(l2.mean()),
            'steps_per_track_max': int(l2.max()),
            'raster_mcells_per_s': ncells / leg.k1_kernel_seconds() / 1e6,
            'phase_ms_per_step': {'raster_k1': leg.acc['raster_ms'] / k2, 'table_k2a': leg.acc['table_ms'] / k2,
                                  'stepper_kernels_k2b': leg.acc['step_kernel_ms'] / k2,
                                  'histogram_binning_k3': leg.acc['hist_ms'] / k2},
            'roofline': roof2,
            'what': 'the same pass on the linear-ramp stand-in potential (the headline of rounds 1-2): the batch crosses '
                    'the raster as one front; after the timed region, not part of `value`',
        }
    if world == 1 and solved and default_variant and args.full_chip_tracks > 0:
        # what the same stepper carries when the batch fills the chip (one call of 250 000 tracks: 512-lane roaming blocks while
        # ~105 000 tracks roam, 256-lane blocks after): one warm-up-free pass, histogram checked by its checksum
        nf = int(args.full_chip_tracks)
        np.random.seed(seed)
        fr, fc = movmodel.get_starting_indices(nf, (5, 55, 1, 2), 'random', tuple(args.width_km), res)
        leg = Passes(args, mods, dem, pot, torch.from_numpy(np.stack([fr, fc], 1).astype(np.int32)).to(dev), 0, gridsize, res, seed, 1, dev)
        el3, last3 = leg.run(1, 0)
        l3 = last3.lengths.cpu().numpy().astype(np.int64)
        st3 = last3.stats
        h3 = leg.last_hist
        counted = int((h3.to(torch.int64) & (0xFFFFFFFF if h3.dtype == torch.int32 else -1)).sum().item())
        out['full_chip'] = {
            'tracks': nf, 'value': nf / el3, 'unit': 'tracks/s', 'ms_per_step': el3 * 1e3, 'steps_per_s': leg.acc['steps'] / el3,
            'steps_per_s_in_roam_kernels': (leg.acc['block_window_steps'] / (leg.acc['block_window_ms'] * 1e-3)
                                            if leg.acc['block_window_ms'] > 0 else None),
            'roam_launches': int(st3.get('roam_launches', 0)), 'roam_wide_launches': int(st3.get('roam_wide_launches', 0)),
            'share_at_max_moves': float(np.mean(l3 - 1 >= max_moves)),
            'histogram_counts_every_point_once': bool(counted == int(l3.sum())),
            'what': 'one pass of this many tracks in ONE call on the same field, after the timed region and not part of `value`: '
                    'the stepper with two waves per SIMD while more tracks roam than 256 CUs x 256 lanes hold '
                    '(k_step_roam<REV, 512>), narrow blocks after (profiles/r04_roam_fill.txt)',
        }
        del leg, last3, h3
    if world == 1 and args.cpu_seconds > 0:
        cap = args.cpu_cap if solved else None
        oro_gpu, upd_gpu = layers.updraft_from_dem(dem, res, 10.0, 270.0, threshold=0.75)
        upd_gpu_h = upd_gpu.cpu().numpy()
        cpu, run, m = cpu_baseline(args, gridsize, dem_h, oro_gpu.cpu().numpy(), upd_gpu_h, pot.cpu().numpy(),
                                   starts_h, seed, cap)
        out['cpu_baseline'] = cpu
        # same tracks, same streams, same inputs, same cap: the GPU's lengths and histogram for the
        # sample must be the oracle's, bit for bit
        if cap:
            sub = torch.from_numpy(starts_h[:m]).to(dev)
            g = movmodel.simulate_tracks(0.0, sub, gridsize, 1, 1.0, upd_gpu, pot, seed=seed, max_moves=cap,
                                         table=build_table(args, movmodel, upd_gpu, pot))
            gl = g.lengths.cpu().numpy()
            gh = g.hist.cpu().numpy().view(np.uint32)
            cpu['sample_lengths_equal_gpu'] = bool(np.array_equal(run['lengths'], gl))
            cpu['sample_histogram_equal_gpu'] = bool(np.array_equal(run['hist'], gh))
            cpu['sample_share_at_cap'] = float(np.mean(gl - 1 >= cap))
            cpu['sample_gpu_stats'] = {k: g.stats[k] for k in ('launches', 'block_window_launches', 'wander_sorts')}
            # uncapped, the sample's tracks are the first m of the timed pass: those that finished under
            # the cap have the same length there
            done = run['lengths'] - 1 < cap
            cpu['sample_finished_lengths_equal_timed_pass'] = bool(np.array_equal(run['lengths'][done], lengths[:m][done]))
        else:
            cpu['sample_lengths_equal_gpu'] = bool(np.array_equal(run['lengths'], lengths[:m]))
    print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
