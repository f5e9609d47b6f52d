"""Directional potential (K5) host side: the Dirichlet sets of
MovModel.get_boundary_nodes (/root/reference/ssrs/movmodel.py:21-57) are
small integer logic evaluated here; the linear solve itself runs matrix-free
on the GPU (ssrs_potential_solve), replacing assemble_sparse_linear_system +
solve_sparse_linear_system (movmodel.py:59-128)."""
import ctypes as C

import numpy as np
import torch

from . import _native as nat
from ._device import stream_ptr, to_dev, like_input


def get_boundary_nodes(move_dirn, grid_shape):
    """movmodel.py:21-57 -> (node ids, energies); node id = col * nrow + row.

    The movement direction picks which edge segments are held at 0 ("low",
    downstream) and at 1000 ("high", upstream).  As in the reference the energy
    vector is split at its midpoint, whatever the two list lengths are.
    """
    nrow, ncol = int(grid_shape[0]), int(grid_shape[1])
    cols = np.arange(ncol, dtype=np.int64)
    inner_rows = np.arange(1, nrow - 1, dtype=np.int64)
    north = nrow * (cols + 1) - 1
    south = nrow * cols
    west = inner_rows
    east = (ncol - 1) * nrow + inner_rows
    angle = move_dirn % 90.
    quadrant = int((move_dirn % 360) // 90.)
    col_len = round(ncol * angle / 90.)
    row_len = round(nrow * angle / 90.)
    if quadrant == 0:
        low = (north[col_len:], east[nrow - row_len:])
        high = (south[:ncol - col_len], west[:row_len])
    elif quadrant == 1:
        low = (south[ncol - col_len:], east[:nrow - row_len])
        high = (north[:col_len], west[row_len:])
    elif quadrant == 2:
        low = (south[:ncol - col_len], west[:row_len])
        high = (north[col_len:], east[nrow - row_len:])
    else:
        high = (south[ncol - col_len:], east[:nrow - row_len])
        low = (north[:col_len], west[row_len:])
    nodes = np.concatenate(low + high)
    energy = np.zeros(nodes.size)
    energy[nodes.size // 2:] = 1000.
    return nodes, energy


def dirichlet_rasters(move_dirn, grid_shape):
    """Boundary nodes as (mask u8, values f64) rasters of shape (rows, cols)."""
    nrow, ncol = int(grid_shape[0]), int(grid_shape[1])
    nodes, energy = get_boundary_nodes(move_dirn, (nrow, ncol))
    mask = np.zeros((nrow, ncol), dtype=np.uint8)
    vals = np.zeros((nrow, ncol), dtype=np.float64)
    r, c = nodes % nrow, nodes // nrow
    mask[r, c] = 1
    vals[r, c] = energy
    return mask, vals


def solve_potential(updraft, move_dirn, rel_tol=1e-15, max_iterations=2000,
                    initial_guess=None, return_stats=False, use_amg=True, extra_sweeps=0, cycle='V',
                    strong_rounds=0, kdepth=0, one_sided=False):
    """MovModel(...).solve_sparse_linear_system equivalent -> f32 (rows, cols).

    `updraft` is the conductivity raster (usable updraft, f64); numpy in ->
    numpy out, CUDA tensor in -> tensor out.

    rel_tol is on |r|/|b|.  The default is tight on purpose: conductive clusters
    floating in dead terrain give eigenvalues ~1e-8, so the level of such a cluster
    is only determined to residual x 1e8; 1e-12 left differences of up to 0.6 (of
    1000) against the reference's direct solve on random speckled rasters, 1e-14
    up to 6e-3 when the solver stops just under it, 1e-15 leaves < 1e-3
    (tests/dev/soak_potential.py; the direct solve itself is only good to ~1e-3 at
    condition numbers of 1e10), at ~25 % more iterations than 1e-12.
    """
    cond = to_dev(updraft, torch.float64)
    rows, cols = int(cond.shape[0]), int(cond.shape[1])
    mask_h, vals_h = dirichlet_rasters(move_dirn, (rows, cols))
    mask = to_dev(mask_h)
    vals = to_dev(vals_h)
    guess = to_dev(initial_guess, torch.float64)
    out = torch.empty((rows, cols), dtype=torch.float32, device=cond.device)
    # ssrs_potential_workspace_bytes is a safe upper bound (1.5 KB per cell: 45 GB at 5000 x 6000);
    # the hierarchy really takes ~840 B per cell (tools/attic/probe_solver_footprint.py), so the first
    # try reserves 1.1 KB per cell and only a solve that runs out of it takes the full bound
    full = nat.lib().ssrs_potential_workspace_bytes(rows, cols)
    first = min(full, (1100 * rows * cols + (96 << 20)) // 256 * 256)
    stats = nat.SsrsSolveStats()
    flags = ((0 if use_amg else nat.SSRS_SOLVE_NO_AMG) | (int(extra_sweeps) << 4) |
             (2 if cycle == 'K' else 0) | (int(strong_rounds) << 8) | (int(kdepth) << 12) |
             (4 if one_sided else 0))
    for nbytes in ((first, full) if first < full else (full,)):
        ws = torch.empty(nbytes, dtype=torch.uint8, device=cond.device)
        rc = nat.lib().ssrs_potential_solve(
            nat.ptr(cond), nat.ptr(mask), nat.ptr(vals), nat.ptr(guess), nat.ptr(out),
            rows, cols, C.c_double(rel_tol), int(max_iterations), flags, nat.ptr(ws),
            C.c_size_t(nbytes), C.byref(stats), stream_ptr())
        if rc == nat.SSRS_ERR_INVALID and nbytes < full and \
                b'workspace' in nat.lib().ssrs_last_error():
            del ws
            continue
        nat.check(rc)
        break
    if not stats.converged:
        import warnings
        warnings.warn(f'potential solve stopped at |r|/|b| = {stats.residual:.3e} after '
                      f'{stats.iterations} iterations (rel_tol {rel_tol:g})')
    res = like_input(out, updraft)
    if return_stats:
        return res, dict(iterations=int(stats.iterations), converged=bool(stats.converged),
                         residual=float(stats.residual), kernel_ms=float(stats.kernel_ms),
                         amg_levels=int(stats.amg_levels), amg_coarsest=int(stats.amg_coarsest),
                         setup_ms=float(stats.setup_ms), workspace_used=int(stats.workspace_used),
                         workspace_bytes=int(nbytes))
    return res
