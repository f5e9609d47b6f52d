"""Multi-GPU sharding of the track batch.

Tracks are independent given read-only rasters (the reference maps them over a
process pool, /root/reference/ssrs/simulator.py:360), so each rank simulates a
contiguous range of GLOBAL track ids against its own replica of the rasters.
The uniform stream is keyed by the global id, so the union over ranks equals a
single-GPU run bit for bit.  The one exchange step is a sum-reduce of the
uint32 presence histogram (120 MB at 5000 x 6000) -- RCCL over xGMI when the
process group is 'nccl', gloo in the CPU tests.
"""
import torch
import torch.distributed as dist


def shard_range(ntracks, rank, world_size):
    """Contiguous global-id range [lo, hi) of `rank` (sizes differ by <= 1)."""
    base, rem = divmod(int(ntracks), int(world_size))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_cases(case_ids, rank=None, world_size=None, group=None):
    """Seasonal / snapshot mode shards by CASE: the wind snapshots are the
    outer independent unit, each rank takes a contiguous share of them (all of its
    tracks), and only the per-case normalised presence sums are exchanged.  Without a
    process group every case stays here."""
    if rank is None or world_size is None:
        if not (dist.is_available() and dist.is_initialized()):
            return list(case_ids)
        rank, world_size = dist.get_rank(group), dist.get_world_size(group)
    lo, hi = shard_range(len(case_ids), rank, world_size)
    return list(case_ids)[lo:hi]


def reduce_presence_sum(summary, group=None):
    """Sum of the per-case normalised presence maps (f64) over ranks, on every rank:
    the one exchange step of seasonal mode (120-240 MB at 5000 x 6000).  The following
    /max of plot_presence_map (simulator.py:546) is then the same on all ranks."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return summary
    dist.all_reduce(summary, op=dist.ReduceOp.SUM, group=group)
    return summary


class HistogramOverflow(OverflowError):
    """A uint32 presence count may have wrapped (single-rank guard of Simulator.simulate_tracks;
    the multi-rank reduce widens instead of raising)."""


def _local_max(flat):
    """Upper bound of the largest count of a uint32-in-int32 histogram as an int64 scalar
    tensor: one pass (aminmax), no host synchronisation.  A negative int32 is a count >= 2^31;
    any such count makes the bound 2^32 - 1 (conservative: the guard then takes the wide path)."""
    lo, hi = torch.aminmax(flat)
    return torch.where(lo < 0, torch.full_like(hi, 0, dtype=torch.int64) + ((1 << 32) - 1), hi.to(torch.int64)).reshape(1)


class _GuardedWork:
    """Handle of an asynchronous guarded reduce: wait() orders the current stream after the
    collective; `result` is the tensor that holds the sum (the caller's `hist`, or its int64
    widening when a 32-bit sum could have wrapped)."""

    def __init__(self, work, result):
        self._work, self.result = work, result

    def wait(self):
        self._work.wait()
        return True


def reduce_histogram(hist, dst=0, group=None, all_ranks=False, async_op=False, guard=True):
    """Sum of the presence histogram over ranks; returns the tensor that holds it.

    Counts are uint32 stored in an int32 tensor; two's-complement addition makes the
    int32 sum bit-identical to the uint32 sum as long as no cell passes 2^32 - 1.  With
    `guard` (default) the ranks first add up their largest counts (one int64 scalar):
    below 2^32 the sum cannot wrap and `hist` is reduced in place (120 MB at
    5000 x 6000); otherwise the counts are widened to int64 and THAT tensor is reduced
    and returned (presence.smooth_presence_counts accepts it).  Tracks that circle in a
    pocket of the potential field until max_moves put ~1e9 visits into single cells
    per 100k tracks, so eight such ranks would wrap a 32-bit sum.  No-op without a
    process group.

    async_op=True returns a handle (or None when there is nothing to reduce): the bound is
    agreed on first (one scalar all-reduce and a host read, microseconds), then the 32-bit sum
    in place or the widened 64-bit sum is queued on the collective's stream and runs under the
    next batch's stepper launches; ``wait()`` orders the current stream after it and
    ``handle.result`` is the tensor that holds the sum (the caller must not touch `hist`
    before ``wait()``).  With guard=False every rank must pass the same dtype."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return None if async_op else hist

    def run(t, **kw):
        if all_ranks:
            return dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group, **kw)
        return dist.reduce(t, dst=dst, op=dist.ReduceOp.SUM, group=group, **kw)

    # Every rank takes the SAME sequence of collectives whatever its local dtype: a rank that stepped its
    # tracks in sub-batches arrives with int64 counts while a rank one track short of the split arrives
    # with int32 (shard sizes differ by one).  The ranks therefore agree on the width first: an already
    # widened rank contributes 2^32 to the bound, which sends every rank down the wide path.
    already_wide = hist.dtype == torch.int64
    flat = hist.reshape(-1) if already_wide else hist.view(torch.int32).reshape(-1)
    need_wide = already_wide
    if guard:
        local = torch.full((1,), 1 << 32, dtype=torch.int64, device=hist.device) if already_wide \
            else _local_max(flat)
        dist.all_reduce(local, op=dist.ReduceOp.SUM, group=group)
        need_wide = int(local.item()) >= (1 << 32)
    wide = None
    if need_wide:
        wide = flat if already_wide else flat.to(torch.int64) & 0xFFFFFFFF
    if async_op:
        if wide is not None:
            return _GuardedWork(run(wide, async_op=True), wide.reshape(hist.shape))
        return _GuardedWork(run(flat, async_op=True), hist)
    if wide is not None:
        run(wide)
        return wide.reshape(hist.shape)
    run(flat)
    return hist


def gather_track_summaries(lengths, ends, dst=0, group=None):
    """Optional: lengths/endpoints of every shard on `dst` (8 B per track)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return [lengths], [ends]
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    sizes = [torch.zeros(1, dtype=torch.int64, device=lengths.device) for _ in range(world)]
    dist.all_gather(sizes, torch.tensor([lengths.numel()], dtype=torch.int64,
                                        device=lengths.device), group=group)
    nmax = int(max(int(s.item()) for s in sizes))
    # int32 on the wire: gloo has no int16 collectives
    pad_l = torch.zeros(nmax, dtype=torch.int32, device=lengths.device)
    pad_e = torch.zeros((nmax, 2), dtype=torch.int32, device=ends.device)
    pad_l[:lengths.numel()] = lengths
    pad_e[:ends.shape[0]] = ends
    outs_l = [torch.zeros_like(pad_l) for _ in range(world)] if rank == dst else None
    outs_e = [torch.zeros_like(pad_e) for _ in range(world)] if rank == dst else None
    dist.gather(pad_l, outs_l, dst=dst, group=group)
    dist.gather(pad_e, outs_e, dst=dst, group=group)
    if rank != dst:
        return None, None
    return ([o[:int(s.item())].to(lengths.dtype) for o, s in zip(outs_l, sizes)],
            [o[:int(s.item())].to(ends.dtype) for o, s in zip(outs_e, sizes)])
