"""K5 at C2 (5000 x 6000, the bench's field) under SSRS_AMG_OMEGAS=a,b: the step sizes of a pair of Jacobi
sweeps of the V-cycle (default 0.7, 0.7; the roots of a Chebyshev polynomial of degree 2 on [lo, 2] of D^-1 A
give two different ones).  Iterations, seconds, max difference to the default field.
usage: python tools/dev/probe_k5_omegas.py "0.7,0.7" "0.5617,1.3895" ..."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ssrs_amd import layers                                # noqa: E402
from ssrs_amd.potential import solve_potential             # noqa: E402
from ssrs_amd.synthetic import synthetic_dem               # noqa: E402

SHAPE, RES = (5000, 6000), 10.
dem = torch.from_numpy(synthetic_dem(SHAPE, RES)).cuda()
_, upd = layers.updraft_from_dem(dem, RES, 10., 270., threshold=0.75)
del dem
ref = None
for om in sys.argv[1:]:
    os.environ['SSRS_AMG_OMEGAS'] = om
    torch.cuda.synchronize()
    t = time.time()
    pot, st = solve_potential(upd, 0., return_stats=True)
    torch.cuda.synchronize()
    dt = time.time() - t
    if ref is None:
        ref = pot.clone()
    print(f'omegas {om}: {st["iterations"]} iterations, converged {st["converged"]}, residual {st["residual"]:.2e}, {dt:.2f} s, '
          f'max |p - p(first)| {float((pot - ref).abs().max()):.3e}', flush=True)
