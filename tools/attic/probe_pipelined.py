"""bench.py's step (K1 + ring table + 100k tracks) with P batches in flight on P host
threads / HIP streams, the way ssrs_amd.Simulator pipelines its cases.  bench.py itself
stays sequential (one batch at a time); this shows what one GPU sustains when several
independent 100k-track batches are available."""
import os, sys, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from ssrs_amd import layers, movmodel
from ssrs_amd.synthetic import synthetic_dem, ramp_potential
rows, cols, res, n = 5000, 6000, 10., 100000
dem = torch.from_numpy(synthetic_dem((rows, cols), res)).cuda()
np.random.seed(30)
r, c = movmodel.get_starting_indices(n, (5, 55, 1, 2), 'random', (60., 50.), res)
starts = torch.from_numpy(np.stack([r, c], 1).astype(np.int32)).cuda()
pot = torch.from_numpy(ramp_potential((rows, cols))).cuda()
K = 8


def worker(hist, checks, idx):
    with torch.cuda.stream(torch.cuda.Stream()):
        for _ in range(K):
            hist.zero_()
            _, upd = layers.updraft_from_dem(dem, res, 10., 270., threshold=0.75)
            table = movmodel.build_transition_table(upd, pot, ring=True)
            out = movmodel.simulate_tracks(0., starts, (rows, cols), 1, 1., upd, pot, seed=30, table=table, hist=hist)
        torch.cuda.current_stream().synchronize()
        checks[idx] = (int(hist.sum().item()), out.stats['total_steps'])


for P in (1, 2, 3, 4):
    hists = [torch.zeros((rows, cols), dtype=torch.int32, device='cuda') for _ in range(P)]
    checks = [None] * P
    for rep in range(2):                       # first repetition warms allocator and caches
        th = [threading.Thread(target=worker, args=(hists[i], checks, i)) for i in range(P)]
        torch.cuda.synchronize(); t = time.time()
        for x in th: x.start()
        for x in th: x.join()
        torch.cuda.synchronize(); dt = time.time() - t
    assert all(s == st + n for s, st in checks), checks      # every trajectory point counted once
    print(f'{P} batch(es) in flight: {P * K * n / dt / 1e6:.1f} M tracks/s, {dt / (P * K) * 1e3:.2f} ms per batch', flush=True)
