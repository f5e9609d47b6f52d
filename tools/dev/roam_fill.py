"""In-kernel steps/s of the roaming stepper against the number of tracks of one call (the bench's field, max_moves
capped at 600 000 unless CAP=0), with a digest of the results so that two builds of the library (SSRS_HIP_LIB) can be
compared integer for integer.  usage: [SSRS_HIP_LIB=...] [CAP=600000] python tools/dev/roam_fill.py [n1 n2 ...]"""
import hashlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from ssrs_amd import layers, movmodel
from ssrs_amd.potential import solve_potential
from ssrs_amd.synthetic import synthetic_dem
SHAPE, RES = (5000, 6000), 10.
cap = int(os.environ.get('CAP', '600000')) or None
counts = [int(v) for v in sys.argv[1:]] or [50_000, 100_000, 125_000, 140_000, 150_000, 200_000, 300_000]
dem = torch.from_numpy(synthetic_dem(SHAPE, RES)).cuda()
_, upd = layers.updraft_from_dem(dem, RES, 10., 270., threshold=0.75)
os.environ.setdefault('SSRS_AMG_NU', '2,2')          # (the field of rounds 1-3: comparable digests across rounds)
pot = solve_potential(upd, 0.)
table = movmodel.build_transition_table(upd, pot, thr=True, move_dirn=0.)
print(f'# library {os.environ.get("SSRS_HIP_LIB", "libssrs_hip.so")}, max_moves {cap}', flush=True)
for n in counts:
    np.random.seed(30)
    r, c = movmodel.get_starting_indices(n, (5, 55, 1, 2), 'random', (60., 50.), RES)
    starts = np.stack([r, c], 1).astype(np.int32)
    torch.cuda.synchronize(); t = time.time()
    o = movmodel.simulate_tracks(0., starts, SHAPE, 1, 1., upd, pot, seed=30, table=table, profile=True, max_moves=cap)
    torch.cuda.synchronize(); wall = time.time() - t
    st = o.stats
    alive = int((o.lengths - 1 >= (cap or 7_500_000)).sum())
    dig = hashlib.sha256(o.lengths.cpu().numpy().tobytes() + o.ends.cpu().numpy().tobytes() + o.hist.cpu().numpy().tobytes()).hexdigest()[:12]
    print(f'{n} tracks, {alive} at the cap: wall {wall * 1e3:.0f} ms, {n / wall:.0f} tracks/s, block-window launches {st["block_window_launches"]} '
          f'(pair table {st["roam_launches"]}, wide {st["roam_wide_launches"]}), {st["block_window_steps"] / max(st["block_window_ms"], 1e-9) * 1e3:.3e} steps/s in them, '
          f'wave-pairs/launch/32768 {st["roam_wave_pairs"] / max(st["roam_launches"], 1) / 32768:.0f}, digest {dig}', flush=True)
