"""Round 2's sweep saw 105.8 ms per bench step at 1 M tracks on its first run (24.6 ms re-run, same kernel times).
Per-step wall time of the bench's pass on the ramp at 1 M tracks next to the caching allocator's device
allocations, with the stepper's workspace allocated per call (SSRS_NO_WS_CACHE=1) and kept per thread."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from ssrs_amd import layers, movmodel
from ssrs_amd.synthetic import synthetic_dem, ramp_potential
SHAPE, RES = (5000, 6000), 10.
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
dem = torch.from_numpy(synthetic_dem(SHAPE, RES)).cuda()
pot = torch.from_numpy(ramp_potential(SHAPE)).cuda()
np.random.seed(30)
r, c = movmodel.get_starting_indices(n, (5, 55, 1, 2), 'random', (60., 50.), RES)
starts = torch.from_numpy(np.stack([r, c], 1).astype(np.int32)).cuda()
hist = torch.zeros(SHAPE, dtype=torch.int32, device='cuda')
print('workspace per call' if os.environ.get('SSRS_NO_WS_CACHE') else 'workspace kept per thread', flush=True)
for step in range(7):
    st0 = torch.cuda.memory_stats()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    hist.zero_()
    oro, upd = layers.updraft_from_dem(dem, RES, 10., 270., threshold=0.75)
    table = movmodel.build_transition_table(upd, pot, thr=True, move_dirn=0.)
    out = movmodel.simulate_tracks(0., starts, SHAPE, 1, 1., upd, pot, seed=30, table=table, hist=hist, profile=True)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    st1 = torch.cuda.memory_stats()
    print(f'step {step}: {dt * 1e3:.1f} ms wall, stepper kernels {out.stats["kernel_ms"]:.1f} ms + binning {out.stats["hist_ms"]:.1f} ms; '
          f'hipMalloc calls {st1["num_device_alloc"] - st0["num_device_alloc"]}, frees {st1["num_device_free"] - st0["num_device_free"]}, '
          f'retries {st1["num_alloc_retries"] - st0["num_alloc_retries"]}, reserved {st1["reserved_bytes.all.current"] / 2**30:.1f} GiB', flush=True)
    del oro, upd, table, out
    if step == 3:
        # what a sweep does between sizes: other tensors come and go
        junk = [torch.empty(int(2.5e9), dtype=torch.uint8, device='cuda') for _ in range(3)]
        del junk
