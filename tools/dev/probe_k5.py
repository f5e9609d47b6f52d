"""K5 at C2 (5000 x 6000 @10 m, the bench's field) under the solver's experiment switches: iterations, seconds,
workspace.  usage: python tools/dev/probe_k5.py [ROWSxCOLS] -- variants are (label, env, kwargs) below."""
import os, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from ssrs_amd import layers
from ssrs_amd.potential import solve_potential
from ssrs_amd.synthetic import synthetic_dem

shape = tuple(int(v) for v in sys.argv[1].split('x')) if len(sys.argv) > 1 else (5000, 6000)
only = sys.argv[2].split(';') if len(sys.argv) > 2 else None
dem = torch.from_numpy(synthetic_dem(shape, 10.)).cuda()
_, upd = layers.updraft_from_dem(dem, 10., 10., 270., threshold=0.75)
del dem
ramp = (1000. * (1. - torch.arange(shape[0], device='cuda', dtype=torch.float64) / (shape[0] - 1.)))[:, None].expand(shape).contiguous()
variants = [
    ('default', {}, {}),
    ('no fuse', {'SSRS_AMG_NO_FUSE': '1'}, {}),
    ('no blocks', {'SSRS_AMG_NO_BLOCKS': '1'}, {}),
    ('nu 1,2', {'SSRS_AMG_NU': '1,2'}, {}),
    ('nu 1,1', {'SSRS_AMG_NU': '1,1'}, {}),
    ('nu 2,1', {'SSRS_AMG_NU': '2,1'}, {}),
    ('no sell', {'SSRS_AMG_NO_SELL': '1'}, {}),
    ('one row', {'SSRS_AMG_L0_ONE_ROW': '1'}, {}),
    ('ramp guess', {}, {'initial_guess': ramp}),
    ('nu 1,1 + ramp', {'SSRS_AMG_NU': '1,1'}, {'initial_guess': ramp}),
]
# single-level K-cycle: SSRS_AMG_K=level,inner
for lev, inner in ((1, 1), (4, 1), (4, 2), (2, 2), (3, 2), (3, 4), (4, 2), (4, 4), (5, 4), (4, 8), (5, 8), (6, 8)):
    variants.append((f'K {lev},{inner}', {'SSRS_AMG_K': f'{lev},{inner}'}, {}))
ref = None
for label, env, kw in variants:
    if only and label not in only:
        continue
    for k, v in env.items():
        os.environ[k] = v
    try:
        torch.cuda.synchronize(); t = time.time()
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            pot, st = solve_potential(upd, 0., return_stats=True, **kw)
        torch.cuda.synchronize(); dt = time.time() - t
    finally:
        for k in env:
            del os.environ[k]
    if ref is None:
        ref = pot
    print(f'{label:24s} its {st["iterations"]:4d} conv {st["converged"]} res {st["residual"]:.1e} solve {st["kernel_ms"] / 1e3:.2f}s '
          f'setup {st["setup_ms"] / 1e3:.2f}s wall {dt:.2f}s levels {st["amg_levels"]} ws {st["workspace_used"] / 1e9:.1f} GB '
          f'max|d| vs first {float((pot - ref).abs().max()):.2e}', flush=True)
