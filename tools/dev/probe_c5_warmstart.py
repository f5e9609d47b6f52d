"""C5 at full size: does the potential of snapshot s help as the initial guess of snapshot s + 1 (phase 2 pi / 256 apart)?
Default tolerance.  Also: two solves on two streams at once against one after the other."""
import os, sys, time, warnings, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from ssrs_amd import layers
from ssrs_amd.potential import solve_potential
from ssrs_amd.synthetic import synthetic_dem, wind_lattice
SHAPE, RES = (5000, 6000), 10.
dem = torch.from_numpy(synthetic_dem(SHAPE, RES)).cuda()
def updraft(s):
    x, y, ws, wd = wind_lattice((60., 50.), 2.0, phase=2 * np.pi * s / 256)
    _, upd = layers.updraft_from_dem_lattice(dem, RES, x, y, ws, wd, threshold=0.75)
    return upd
def solve(upd, guess=None, tol=1e-15):
    torch.cuda.synchronize(); t = time.perf_counter()
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        p, st = solve_potential(upd, 0., rel_tol=tol, return_stats=True, initial_guess=guess)
    torch.cuda.synchronize()
    return p, st, time.perf_counter() - t
u0, u1, u8 = updraft(0), updraft(1), updraft(8)
p0, s0, t0 = solve(u0)
print(f'snapshot 0 cold: {s0["iterations"]} its, {t0:.2f} s (set-up {s0["setup_ms"]:.0f} ms)', flush=True)
for name, u in (('1', u1), ('8', u8)):
    pc, sc, tc = solve(u)
    pw, sw, tw = solve(u, guess=p0.double())
    print(f'snapshot {name}: cold {sc["iterations"]} its {tc:.2f} s | warm from snapshot 0: {sw["iterations"]} its {tw:.2f} s | '
          f'max |cold - warm| {float((pc - pw).abs().max()):.2e}; |p - p0| max {float((pc - p0).abs().max()):.2f}', flush=True)
for tol in (1e-12, 1e-10, 1e-8):
    p, s, t = solve(u0, tol=tol)
    print(f'snapshot 0 at rel_tol {tol:g}: {s["iterations"]} its {t:.2f} s, max |p - p(1e-15)| {float((p - p0).abs().max()):.2e}', flush=True)
# two at once
res = [None, None]
def work(i, u):
    with torch.cuda.stream(torch.cuda.Stream()):
        res[i] = solve_potential(u, 0., return_stats=True)
        torch.cuda.current_stream().synchronize()
torch.cuda.synchronize(); t = time.perf_counter()
th = [threading.Thread(target=work, args=(i, u)) for i, u in enumerate((u0, u1))]
[x.start() for x in th]; [x.join() for x in th]
torch.cuda.synchronize(); t2 = time.perf_counter() - t
print(f'two solves on two streams / threads at once: {t2:.2f} s ({t2 / 2:.2f} s each); equal to the serial results: '
      f'{bool(torch.equal(res[0][0], p0))}', flush=True)
