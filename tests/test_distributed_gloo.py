"""N > 1 path on CPU: two gloo ranks shard the track ids, each produces its
shard's histogram (here with the C oracle standing in for the GPU stepper --
the sharding + reduce host logic is what is under test), and the reduced
histogram must equal the single-process run bit for bit."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _case():
    from oracle import ssrs_oracle as orc
    from ssrs_amd.synthetic import synthetic_dem
    rows, cols = 64, 80
    z = synthetic_dem((rows, cols), 100.)
    oro = orc.compute_orographic_updraft(10., 270., orc.compute_slope_degrees(z, 100.),
                                         orc.compute_aspect_degrees(z, 100.)).astype(np.float32)
    upd = orc.get_above_threshold_speed(oro, 0.75)
    pot = (1000. * (1 - np.arange(rows)[:, None] / (rows - 1.)) * np.ones((1, cols))).astype(np.float32)
    rng = np.random.default_rng(0)
    starts = np.stack([rng.integers(1, 8, 301), rng.integers(0, cols, 301)], 1)
    return (rows, cols), upd, pot, starts


def _worker(rank, world, port, out_dir):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from oracle import c_oracle
    from ssrs_amd.distributed import shard_range, reduce_histogram, gather_track_summaries
    shape, upd, pot, starts = _case()
    lo, hi = shard_range(len(starts), rank, world)
    res = c_oracle.simulate_tracks(0., starts[lo:hi], shape, 1, 1., upd, pot, seed=30,
                                   track_id_base=lo, want_traj=False, nthreads=1)
    hist = torch.from_numpy(res['hist'].view(np.int32).copy())
    reduce_histogram(hist, dst=0)
    lens, ends = gather_track_summaries(torch.from_numpy(res['lengths']),
                                        torch.from_numpy(res['ends']), dst=0)
    if rank == 0:
        np.save(os.path.join(out_dir, 'hist.npy'), hist.numpy())
        np.save(os.path.join(out_dir, 'lengths.npy'), torch.cat(lens).numpy())
        np.save(os.path.join(out_dir, 'ends.npy'), torch.cat(ends).numpy())
    # all_ranks=True leaves the same sum everywhere
    h2 = torch.from_numpy(res['hist'].view(np.int32).copy())
    reduce_histogram(h2, all_ranks=True)
    np.save(os.path.join(out_dir, f'all_{rank}.npy'), h2.numpy())
    # seasonal mode: cases are sharded contiguously, the per-case presence sums are added
    from ssrs_amd.distributed import shard_cases, reduce_presence_sum
    cases = [f'case{i}' for i in range(7)]
    mine = shard_cases(cases)
    got = [None] * world
    dist.all_gather_object(got, mine)
    assert sum(got, []) == cases and max(map(len, got)) - min(map(len, got)) <= 1
    summ = torch.full((4, 5), float(len(mine)), dtype=torch.float64)
    assert float(reduce_presence_sum(summ)[0, 0]) == len(cases)
    # the pipelined form bench.py uses: two buffers, reduce i in flight while i + 1 is filled
    bufs = [torch.from_numpy(res['hist'].view(np.int32).copy()) for _ in range(2)]
    works = [reduce_histogram(b, all_ranks=True, async_op=True) for b in bufs]
    for w in works:
        assert w is not None
        w.wait()
    assert all(torch.equal(b, h2) for b in bufs)
    # wrap guard: 3e9 visits in one cell on every rank cannot be summed in 32 bits
    big = np.zeros((3, 4), dtype=np.uint32)
    big[1, 2] = 3_000_000_000
    big[0, 0] = 7 + rank
    hb = torch.from_numpy(big.view(np.int32).copy())
    wide = reduce_histogram(hb, all_ranks=True)
    assert wide.dtype == torch.int64 and wide.shape == hb.shape
    assert int(wide[1, 2]) == 3_000_000_000 * world and int(wide[0, 0]) == sum(7 + r for r in range(world))
    # the asynchronous form widens as well (it used to raise from wait(): an 8-GPU run on the
    # solved field, ~1e9 visits per trap cell and rank, would have aborted by design)
    hb = torch.from_numpy(big.view(np.int32).copy())
    work = reduce_histogram(hb, all_ranks=True, async_op=True)
    work.wait()
    assert work.result.dtype == torch.int64 and int(work.result[1, 2]) == 3_000_000_000 * world
    assert int(work.result[0, 0]) == sum(7 + r for r in range(world))
    small = reduce_histogram(torch.from_numpy(res['hist'].view(np.int32).copy()), all_ranks=True, async_op=True)
    small.wait()
    assert small.result.dtype == torch.int32 and torch.equal(small.result, h2)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('world', [2, 3])
def test_sharded_histogram_reduce_equals_single_run(tmp_path, world):
    from oracle import c_oracle
    port = _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    shape, upd, pot, starts = _case()
    ref = c_oracle.simulate_tracks(0., starts, shape, 1, 1., upd, pot, seed=30, want_traj=False,
                                   nthreads=2)
    hist = np.load(tmp_path / 'hist.npy').view(np.uint32)
    assert np.array_equal(hist, ref['hist'])
    assert np.array_equal(np.load(tmp_path / 'lengths.npy'), ref['lengths'])
    assert np.array_equal(np.load(tmp_path / 'ends.npy'), ref['ends'])
    for r in range(world):
        assert np.array_equal(np.load(tmp_path / f'all_{r}.npy').view(np.uint32), ref['hist'])


def test_reduce_is_noop_without_process_group():
    from ssrs_amd.distributed import reduce_histogram
    h = torch.arange(12, dtype=torch.int32).reshape(3, 4)
    assert reduce_histogram(h) is h
    assert reduce_histogram(h, async_op=True) is None
    from ssrs_amd.distributed import shard_cases, reduce_presence_sum
    assert shard_cases(['a', 'b', 'c']) == ['a', 'b', 'c']
    assert shard_cases(['a', 'b', 'c'], rank=1, world_size=2) == ['c']
    t = torch.ones(3, dtype=torch.float64)
    assert reduce_presence_sum(t) is t
