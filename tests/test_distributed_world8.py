"""The N = 8 host logic on CPU (gloo, eight processes): what an 8-GPU node runs around the kernels
(/root/reference/ssrs/simulator.py:347-381 is the process pool the shards replace; :518-546 the
normalisation ladder of the seasonal sum).  No 8-GPU node was ever available to the builder, so
everything that can go wrong WITHOUT a GPU at world 8 is exercised here: shard remainders, the
widening reduce when only some ranks arrive widened, 256 cases over 8 ranks, the seasonal
presence sum through `Simulator.compute_presence_map`, and `_step_case`'s rank-independent
widening decision (shard sizes differ by one, the dtypes must not)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

WORLD = 8


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def test_shard_range_remainders_world8():
    from ssrs_amd.distributed import shard_range, shard_cases
    for n in (0, 1, 7, 8, 9, 15, 100_000, 1_000_000, 1_600_001, 12_345_677):
        spans = [shard_range(n, r, WORLD) for r in range(WORLD)]
        assert spans[0][0] == 0 and spans[-1][1] == n
        assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))          # contiguous, in rank order
        sizes = [hi - lo for lo, hi in spans]
        assert max(sizes) - min(sizes) <= 1 and sizes == sorted(sizes, reverse=True)
        assert max(sizes) == -(-n // WORLD)                                  # Simulator's `widest_share`
    cases = [f'y2010m{m:02d}d{d:02d}h12' for m in range(1, 9) for d in range(1, 33)]      # 256 snapshots
    got = [shard_cases(cases, rank=r, world_size=WORLD) for r in range(WORLD)]
    assert sum(got, []) == cases and all(len(g) == 32 for g in got)
    got = [shard_cases(cases[:250], rank=r, world_size=WORLD) for r in range(WORLD)]
    assert sum(got, []) == cases[:250] and sorted(map(len, got)) == [31] * 6 + [32] * 2


class _FakeBatch:
    def __init__(self, hist, npoints):
        self.hist, self.lengths, self.ends = hist, torch.tensor([npoints], dtype=torch.int32), torch.zeros((1, 2), dtype=torch.int16)
        self.stats = dict(total_steps=npoints)
        self.total_points = npoints


def _worker(rank, world, port, out_dir):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    from datetime import timedelta
    dist.init_process_group('gloo', rank=rank, world_size=world, timeout=timedelta(seconds=120))
    from ssrs_amd import distributed as D
    from ssrs_amd import movmodel, presence
    from ssrs_amd.simulator import Simulator
    shape = (6, 7)

    # 1. ranks 0..2 arrive with int64 counts (they stepped sub-batches), the others with int32: every rank
    #    must take the same collectives and end up with the same 64-bit sum
    def rank_counts(r):
        m = np.random.default_rng(100 + r).integers(0, 1000, shape).astype(np.int64)
        m[2, 3] = 3_000_000_000 if r == 1 else 5
        return m

    def mixed(r):
        m = rank_counts(r)
        return torch.from_numpy(m.copy()) if r < 3 else torch.from_numpy(m.astype(np.uint32).view(np.int32).copy())
    want = sum(rank_counts(r) for r in range(world))
    out = D.reduce_histogram(mixed(rank), all_ranks=True)
    assert out.dtype == torch.int64 and np.array_equal(out.numpy(), want), rank
    # the asynchronous form, the same mix
    work = D.reduce_histogram(mixed(rank), all_ranks=True, async_op=True)
    work.wait()
    assert work.result.dtype == torch.int64 and np.array_equal(work.result.numpy(), want)
    # nobody widened, nothing near 2^32: stays 32-bit on every rank
    out = D.reduce_histogram(torch.from_numpy((rank_counts(rank) % 1000).astype(np.int32)), all_ranks=True)
    assert out.dtype == torch.int32 and np.array_equal(out.numpy(), sum(rank_counts(r) % 1000 for r in range(world)))

    # 2. _step_case: 1 600 001 tracks over 8 ranks in the ratio of the real run (here 17 tracks, step 2):
    #    rank 0 has 3 tracks, the others 2; every rank must come back with the same dtype
    sim = object.__new__(Simulator)
    sim.hist_safe_tracks, sim.track_direction, sim.gridsize = 2, 0., shape
    sim.track_dirn_restrict, sim.track_stochastic_nu, sim.save_tracks, sim.steps_per_launch = 1, 1., False, 0
    total = 17
    lo, hi = D.shard_range(total, rank, world)

    def fake_simulate(move_dirn, sub, gridsize, *a, track_id_base=0, **kw):
        n = int(sub.shape[0])
        hist = torch.zeros(shape, dtype=torch.int32)
        for t in range(track_id_base, track_id_base + n):
            hist[t % shape[0], t % shape[1]] += t + 1
        return _FakeBatch(hist, int(hist.sum()))
    real = movmodel.simulate_tracks
    movmodel.simulate_tracks = fake_simulate
    try:
        batch = sim._step_case(torch.zeros((hi - lo, 2), dtype=torch.int32), lo, (None, None), 30, None,
                               widest_share=-(-total // world))
    finally:
        movmodel.simulate_tracks = real
    assert batch.hist.dtype == torch.int64, (rank, hi - lo)          # also on the ranks with exactly `step` tracks
    summed = D.reduce_histogram(batch.hist, all_ranks=True)
    want = np.zeros(shape, dtype=np.int64)
    for t in range(total):
        want[t % shape[0], t % shape[1]] += t + 1
    assert np.array_equal(summed.numpy(), want)

    # 3. seasonal mode: 256 cases over 8 ranks through compute_presence_map's ladder (per case /max, sum over
    #    the cases of ALL ranks, /max), with CPU stand-ins for the K4 kernels
    cases = [f'c{i:03d}' for i in range(256)]
    sim = object.__new__(Simulator)
    sim.case_ids, sim.thermals_realization_count = cases, 0
    sim.gridsize, sim.resolution, sim.mode_data_dir = shape, 100., out_dir
    sim._presence_device = lambda: torch.device('cpu')

    def counts(case_id):
        return torch.from_numpy(np.random.default_rng(int(case_id[1:])).integers(0, 50, shape).astype(np.float64))
    sim._counts_for = lambda case_id, real_id: counts(case_id)
    saved = (presence.smooth_presence_counts, presence.normalise_add, presence.normalise_to_f32)
    presence.smooth_presence_counts = lambda c, krad: c.clone()

    def normalise_add(a, acc):
        a /= a.max()
        acc += a
    presence.normalise_add = normalise_add
    presence.normalise_to_f32 = lambda s: (s / s.max()).to(torch.float32)
    try:
        assert len(sim.my_case_ids()) == 32 and not sim._shards_tracks()
        got = sim.compute_presence_map(radius=300.)
    finally:
        presence.smooth_presence_counts, presence.normalise_add, presence.normalise_to_f32 = saved
    ref = torch.zeros(shape, dtype=torch.float64)
    for c in cases:
        x = counts(c)
        ref += x / x.max()
    ref = (ref / ref.max()).to(torch.float32).numpy()
    # (the order of the f64 additions differs from a single process's: ranks' partial sums are added)
    assert np.allclose(got, ref, rtol=0, atol=1e-6), rank
    if rank == 0:
        assert np.array_equal(np.load(os.path.join(out_dir, 'summary_presence.npy')), got)
        np.save(os.path.join(out_dir, 'ok.npy'), np.ones(1))
    dist.barrier()
    dist.destroy_process_group()


def test_world8_host_logic(tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(WORLD, port, str(tmp_path)), nprocs=WORLD, join=True)
    assert os.path.exists(tmp_path / 'ok.npy')
