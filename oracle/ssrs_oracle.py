"""ORACLE -- CPU restatement of the SSRS hot path (test infrastructure only).

This file is the *checker*, never the product: only `tests/`,
`__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py` may import it.
The product path (`ssrs_amd/`) must not import anything under `oracle/`.

It restates, in plain numpy / python, the algorithm of the reference's two
hot-path modules, citing the reference file:line each function follows
(paths relative to /root/reference):

  raster     ssrs/layers.py    :11-22 (orographic), :63-128 (slope/aspect),
                               :171-185 (threshold)
  stepper    ssrs/movmodel.py  :131-141 (constants), :144-182 (starts),
                               :185-261 (restrictions, probabilities),
                               :264-318 (generate_simulated_tracks),
                               :410-439 (presence), :10-128 + :442-447 (potential)

Parity status: PINNED.  Every function is checked against golden vectors
captured from the reference itself (tests/golden/*.npz, produced by
tests/golden/generate_golden.py which imports the reference modules in the
build container); see tests/test_oracle_golden.py.

The arithmetic order matters for bit-exact track decisions and is kept exactly:
f32 potential differences, f64 everything else, numpy's pairwise-8 `np.sum`
for the 9 probabilities, sequential cumsum, `searchsorted(..., 'right')`.
"""
from math import floor, ceil, sqrt

import numpy as np

# ---------------------------------------------------------------------------
# constants (movmodel.py:131-141)
# ---------------------------------------------------------------------------
NEIGHBOUR_DELTAS = [(k // 3 - 1, k % 3 - 1) for k in range(9)]
NEIGHBOUR_DELTA_NORMS_INV = np.zeros((3, 3), dtype=np.float32)
for _k, (_dr, _dc) in enumerate(NEIGHBOUR_DELTAS):
    _dist = sqrt(float(_dr * _dr + _dc * _dc))
    NEIGHBOUR_DELTA_NORMS_INV[_k // 3, _k % 3] = 1.0 / _dist if _dist > 0 else 0.0


# ---------------------------------------------------------------------------
# raster: slope / aspect / orographic updraft / threshold (layers.py)
# ---------------------------------------------------------------------------
def _horn_gradients(z_mat, res):
    """Horn 3x3 differences exactly as layers.py:78-90 / :111-123.
    NB: the reference's "x" is the row axis and "y" the column axis."""
    z_1 = z_mat[:-2, 2:]
    z_2 = z_mat[1:-1, 2:]
    z_3 = z_mat[2:, 2:]
    z_4 = z_mat[:-2, 1:-1]
    z_6 = z_mat[2:, 1:-1]
    z_7 = z_mat[:-2, :-2]
    z_8 = z_mat[1:-1, :-2]
    z_9 = z_mat[2:, :-2]
    dz_dx = ((z_3 + 2 * z_6 + z_9) - (z_1 + 2 * z_4 + z_7)) / (8 * res)
    dz_dy = ((z_1 + 2 * z_2 + z_3) - (z_7 + 2 * z_8 + z_9)) / (8 * res)
    return dz_dx, dz_dy


def compute_slope_degrees(z_mat, res):
    """layers.py:63-93 -- border cells are 0 (NaN -> nan_to_num)."""
    z_mat = np.asarray(z_mat)
    slope = np.zeros_like(z_mat)
    dz_dx, dz_dy = _horn_gradients(z_mat, res)
    slope[1:-1, 1:-1] = np.degrees(np.arctan(np.sqrt(dz_dx**2 + dz_dy**2)))
    return slope


def compute_aspect_degrees(z_mat, res):
    """layers.py:96-128."""
    z_mat = np.asarray(z_mat)
    aspect = np.zeros_like(z_mat)
    dz_dx, dz_dy = _horn_gradients(z_mat, res)
    dz_dx = dz_dx.copy()
    dz_dx[dz_dx == 0.] = 1e-10
    angle = np.degrees(np.arctan(np.divide(dz_dy, dz_dx)))
    angle_mod = 90. * np.divide(dz_dx, np.absolute(dz_dx))
    aspect[1:-1, 1:-1] = 180. - angle + angle_mod
    return aspect


def compute_orographic_updraft(wspeed, wdirn, slope, aspect, min_updraft_val=0.):
    """layers.py:11-22 (that op order)."""
    aspect_diff = np.maximum(0., np.cos((aspect - wdirn) * np.pi / 180.))
    return np.maximum(min_updraft_val, np.multiply(
        wspeed, np.multiply(np.sin(slope * np.pi / 180.), aspect_diff)))


def get_above_threshold_speed(in_array, threshold):
    """layers.py:171-185, vectorised.  The reference runs a python scalar
    function through np.vectorize: each element is first widened to a python
    float (f64), so the arithmetic is f64 whatever the input dtype.  Output is
    f64 (the reference's f32-output quirk when in_array.flat[0] > threshold is
    not reproduced; SURVEY.md section 7 'Updraft dtype trap')."""
    v = np.asarray(in_array).astype(np.float64)
    thr = float(threshold)
    with np.errstate(over='ignore', invalid='ignore'):
        mid = thr * (np.exp(np.power(v / thr, 5)) - 1) / (np.exp(1) - 1)
    out = np.where(v > thr, v, mid)
    return np.where(v > 1e-02, out, 0.)


def interpolate_wind_uv(wspeed_pts, wdirn_pts, interp):
    """simulator.py:778-792: speed/direction -> u/v -> interpolate -> speed,
    direction (degrees in [0, 360)).  `interp` maps a point-value array to the
    raster (the reference uses scipy griddata; synthetic configs use a regular
    lattice, see ssrs_amd.synthetic)."""
    easterly = np.multiply(wspeed_pts, np.sin(wdirn_pts * np.pi / 180.))
    northerly = np.multiply(wspeed_pts, np.cos(wdirn_pts * np.pi / 180.))
    ie = interp(easterly)
    inn = interp(northerly)
    wspeed = np.sqrt(np.square(ie) + np.square(inn))
    wdirn = np.arctan2(ie, inn)
    wdirn = np.mod(wdirn + 2. * np.pi, 2. * np.pi)
    return wspeed, wdirn * 180. / np.pi


# ---------------------------------------------------------------------------
# stepper pieces (movmodel.py:144-261)
# ---------------------------------------------------------------------------
def get_starting_indices(ntracks, sbounds, stype, twidth, tres):
    """movmodel.py:144-182.  Consumes the legacy global numpy RNG for 'random'
    exactly like the reference (one np.random.randint call)."""
    if (sbounds[1] < sbounds[0] or sbounds[3] < sbounds[2] or
            sbounds[0] < 0. or sbounds[2] < 0. or sbounds[1] > twidth[0] or
            sbounds[3] > twidth[1]):
        raise ValueError('track_start_region incompatible with terrain_width!')
    res_km = tres / 1000.
    xind_max = ceil(twidth[0] / res_km)
    yind_max = ceil(twidth[1] / res_km)
    xind_low = min(max(floor(sbounds[0] / res_km) - 1, 1), xind_max - 2)
    xind_upp = max(min(ceil(sbounds[1] / res_km), xind_max - 1), 2)
    yind_low = min(max(floor(sbounds[2] / res_km) - 1, 1), yind_max - 2)
    yind_upp = max(min(ceil(sbounds[3] / res_km), yind_max - 1), 2)
    # np.mgrid[xl:xu, yl:yu] ravel order: x outer, y inner
    nx = xind_upp - xind_low
    ny = yind_upp - yind_low
    xs = np.repeat(np.arange(xind_low, xind_upp), ny)
    ys = np.tile(np.arange(yind_low, yind_upp), nx)
    base_count = nx * ny
    if stype == 'structured':
        idx = np.round(np.linspace(0, base_count - 1, ntracks % base_count))
        idx = idx.astype(int)
        if ntracks > base_count:
            reps = ntracks // base_count
            rows = np.concatenate((np.tile(ys, reps), np.tile(ys, reps)[idx]))
            cols = np.concatenate((np.tile(xs, reps), np.tile(xs, reps)[idx]))
        else:
            rows, cols = ys[idx], xs[idx]
    elif stype == 'random':
        idx = np.random.randint(0, base_count, ntracks)
        rows, cols = ys[idx], xs[idx]
    else:
        raise ValueError((f'Model:Invalid sim_start_type of {stype}\n'
                          'Options: structured, random'))
    return rows.astype(int), cols.astype(int)


def get_track_restrictions(dr, dc):
    """movmodel.py:185-202: cells within +-45 deg of the previous move;
    (0,0) -> all but the centre."""
    mask = np.zeros(9, dtype=int)
    if dr == 0 and dc == 0:
        mask[:] = 1
    else:
        for k, (r, c) in enumerate(NEIGHBOUR_DELTAS):
            if dr != 0 and dc != 0:      # diagonal: rows {dr,0} x cols {0,dc}
                ok = (r in (dr, 0)) and (c in (0, dc))
            elif dr == 0:                # pure column move: that column
                ok = (c == dc)
            else:                        # pure row move: that row
                ok = (r == dr)
            mask[k] = 1 if ok else 0
    mask[4] = 0
    return mask


def move_away_from_boundary(row, col, num_rows, num_cols):
    """movmodel.py:205-217 (asymmetric on purpose: row<=1 but col<=0)."""
    new_row, new_col = row, col
    if row <= 1:
        new_row = row + 2
    elif row >= num_rows - 2:
        new_row = row - 2
    if col <= 0:
        new_col = col + 2
    elif col >= num_cols - 2:
        new_col = col - 2
    return new_row, new_col


def get_directional_probs(theta):
    """movmodel.py:247-257."""
    d = np.zeros((3, 3))
    d[0, :] = [np.cos(np.pi / 4 + theta), np.cos(theta),
               np.cos(7 * np.pi / 4 + theta)]
    d[1, :] = [np.cos(np.pi / 2 + theta), 0, np.cos(3 * np.pi / 2 + theta)]
    d[2, :] = [np.cos(3 * np.pi / 4 + theta), np.cos(np.pi + theta),
               np.cos(5 * np.pi / 4 + theta)]
    d[d < 0.01] = 0.
    return np.flipud(d.clip(min=0.)).flatten()


def pairwise8_sum9(x):
    """numpy's pairwise summation for n == 9 (8-way unrolled block + tail):
    ((x0+x1)+(x2+x3)) + ((x4+x5)+(x6+x7)) + x8."""
    return (((x[0] + x[1]) + (x[2] + x[3])) +
            ((x[4] + x[5]) + (x[6] + x[7]))) + x[8]


def generate_move_probabilities(in_probs, move_dirn, nu_par, dir_bool):
    """movmodel.py:220-244, scalar python floats, explicit summation order."""
    prior = [float(v) for v in get_directional_probs(move_dirn * np.pi / 180.)]
    return _move_probabilities([float(v) for v in in_probs], prior,
                               float(nu_par), [int(b) for b in dir_bool])


def _move_probabilities(w, prior, nu, mask):
    out = list(w)
    if any(v != v for v in out):          # NaN anywhere -> prior (:228-230)
        out = list(prior)
    out = [v if v > 0. else 0. for v in out]   # clip(min=0) (:231)
    out[4] = 0.
    out = [v * float(m) for v, m in zip(out, mask)]
    if not any(v != 0. for v in out):     # all masked weights zero (:234-235)
        out = list(prior)
    out[4] = 0.
    out = [v * float(m) for v, m in zip(out, mask)]
    if not any(v != 0. for v in out):     # prior fully masked too (:239-240)
        out = list(prior)
    s1 = pairwise8_sum9(out)
    out = [v / s1 for v in out]
    if nu != 1.0:
        out = [float(np.power(v, nu)) for v in out]
    s2 = pairwise8_sum9(out)
    return [v / s2 for v in out]


def choose_index(p, u):
    """np.random.choice(range(9), p=p) with its uniform made explicit:
    cdf = cumsum(p); cdf /= cdf[-1]; searchsorted(cdf, u, 'right')
    (numpy/random/mtrand.pyx `choice`, legacy path)."""
    cdf = []
    acc = 0.0
    for v in p:
        acc = acc + v
        cdf.append(acc)
    last = cdf[-1]
    idx = 0
    for c in cdf:
        if c / last <= u:
            idx += 1
    return idx


def window_weights(row, col, updraft_field, potential_field, prior):
    """Raw 9 move weights of movmodel.py:292-306 before masking."""
    if updraft_field is not None:
        win = updraft_field[row - 1:row + 2, col - 1:col + 2]
        win = np.maximum(win, 1e-06)                   # clip(min=1e-06) (:295)
        centre = win[1, 1]
        w = 2.0 / (1.0 / centre + 1.0 / win)           # harmonic mean (:260)
    else:
        if potential_field is not None:
            # the reference multiplies a flat (9,) prior by a (3,3) window
            # here (:299,:305) and numpy raises a broadcast ValueError
            raise ValueError('potential_field needs updraft_field')
        w = np.asarray(prior, dtype=np.float64).reshape(3, 3)
    if potential_field is not None:
        pw = potential_field[row - 1:row + 2, col - 1:col + 2]
        diff = pw[1, 1] - pw                           # stays f32 for f32 field
        diff = np.multiply(diff, NEIGHBOUR_DELTA_NORMS_INV)   # f32*f32 -> f32
        w = np.multiply(w, diff)                       # f64*f32 -> f64
    return [float(v) for v in np.asarray(w, dtype=np.float64).flatten()]


def generate_simulated_tracks(move_dirn, start_location, grid_shape,
                              memory_parameter=1, scaling_parameter=1.,
                              updraft_field=None, potential_field=None,
                              uniform=None, max_moves=None):
    """movmodel.py:264-318.  `uniform(step) -> u in [0,1)` supplies the one
    double per step that np.random.choice would draw; None = the legacy global
    numpy stream (np.random.random_sample), i.e. the reference's own source.
    `max_moves` (test / bench hook, like c_oracle's): a cap below the reference's R/2 * C/2."""
    num_rows, num_cols = grid_shape
    burnin = int(min(num_rows, num_cols) / 10)
    if max_moves is None:
        max_moves = num_rows / 2 * num_cols / 2
    prior = [float(v) for v in get_directional_probs(move_dirn * np.pi / 180.)]
    masks = {d: get_track_restrictions(*d) for d in NEIGHBOUR_DELTAS}
    directions = [(0, 0)]
    row, col = int(start_location[0]), int(start_location[1])
    traj = [(row, col)]
    k = 0
    while k < max_moves:
        if k > burnin:
            if not (0 < row < num_rows - 1 and 0 < col < num_cols - 1):
                break
        else:
            row, col = move_away_from_boundary(row, col, num_rows, num_cols)
        w = window_weights(row, col, updraft_field, potential_field, prior)
        mask = masks[(0, 0)].copy()
        hist = directions[-memory_parameter:] if memory_parameter != 0 \
            else directions
        for d in hist:
            mask = mask & masks[d]
        p = _move_probabilities(w, prior, float(scaling_parameter),
                                [int(m) for m in mask])
        u = uniform(k) if uniform is not None else np.random.random_sample()
        idx = choose_index(p, u)
        dr, dc = NEIGHBOUR_DELTAS[idx]
        row, col = row + dr, col + dc
        traj.append((row, col))
        directions.append((dr, dc))
        k += 1
    return np.array(traj, dtype=np.int16)


# ---------------------------------------------------------------------------
# presence (movmodel.py:410-439, simulator.py:520-546)
# ---------------------------------------------------------------------------
def compute_presence_counts(tracks, gridshape, dtype=np.int64):
    """movmodel.py:410-419.  The reference accumulates in int16 (wraps above
    32767 visits); pass dtype=np.int16 to reproduce that, default is exact."""
    count = np.zeros(gridshape, dtype=np.int64)
    for t in tracks:
        t = np.asarray(t, dtype=np.int64)
        np.add.at(count, (t[:, 0], t[:, 1]), 1)
    return count.astype(dtype)


def disk_kernel(krad):
    """movmodel.py:431-436."""
    krad = int(krad)
    y, x = np.ogrid[-krad:krad + 1, -krad:krad + 1]
    kernel = np.zeros((2 * krad + 1, 2 * krad + 1))
    kernel[x**2 + y**2 <= krad**2] = 1
    return kernel / np.sum(kernel)


def smooth_presence_from_counts(count_mat, radius):
    """movmodel.py:431-439 on an already-built count matrix."""
    import scipy.signal as ssg
    presence = ssg.convolve2d(count_mat, disk_kernel(radius), mode='same')
    return presence.astype(np.float32)


def compute_smooth_presence_counts(tracks, gridshape, radius):
    """movmodel.py:422-439."""
    return smooth_presence_from_counts(
        compute_presence_counts(tracks, gridshape), radius)


def presence_kernel_radius(radius_m, resolution, gridsize):
    """simulator.py:520 + :530."""
    krad = min(max(radius_m / resolution, 2), min(gridsize) / 2)
    return int(round(krad))


# ---------------------------------------------------------------------------
# potential (movmodel.py:10-128, :442-447)
# ---------------------------------------------------------------------------
def harmonic_mean(aval, bval, minval=1e-10):
    """movmodel.py:442-447."""
    if aval != 0 and bval != 0:
        return 2. / (1. / aval + 1 / bval)
    return minval


def get_boundary_nodes(move_dirn, grid_shape):
    """movmodel.py:21-57; node id = col * nrow + row."""
    nrow, ncol = grid_shape
    north = np.array([nrow * (x + 1) - 1 for x in range(ncol)])
    south = np.array([nrow * x for x in range(ncol)])
    west = np.arange(1, nrow - 1)
    east = np.array([(ncol - 1) * nrow + x for x in range(1, nrow - 1)])
    ang = move_dirn % 90.
    quad = (move_dirn % 360) // 90.
    col_len = round(ncol * ang / 90.)
    row_len = round(nrow * ang / 90.)
    if quad == 0:
        low = np.concatenate((north[col_len:], east[nrow - row_len:]))
        high = np.concatenate((south[:ncol - col_len], west[:row_len]))
    elif quad == 1:
        low = np.concatenate((south[ncol - col_len:], east[:nrow - row_len]))
        high = np.concatenate((north[:col_len], west[row_len:]))
    elif quad == 2:
        low = np.concatenate((south[:ncol - col_len], west[:row_len]))
        high = np.concatenate((north[col_len:], east[nrow - row_len:]))
    else:
        high = np.concatenate((south[ncol - col_len:], east[:nrow - row_len]))
        low = np.concatenate((north[:col_len], west[row_len:]))
    nodes = np.concatenate((low, high)).astype(np.int64)
    energy = np.zeros(nodes.size)
    energy[nodes.size // 2:] = 1000.
    return nodes, energy


def neighbour_lists(grid_shape):
    """movmodel.py:59-84 vectorised: for every node i (= col*nrow+row) the
    filtered neighbour list and the 1/sqrt(2)-pattern, which the reference
    assigns by *position in the filtered list* (:78-79).  Returns
    (row_index u4, col_index u4, facs f4) in the reference's order."""
    nrow, ncol = grid_shape
    n = nrow * ncol
    i = np.arange(n, dtype=np.int64)
    north = (i + 1) % nrow == 0
    south = i % nrow == 0
    full = np.stack([i - nrow, i - nrow + 1, i + 1, i + nrow + 1, i + nrow,
                     i + nrow - 1, i - 1, i - nrow - 1], axis=1)
    nb_n = np.stack([i + nrow, i + nrow - 1, i - 1, i - nrow - 1, i - nrow],
                    axis=1)
    nb_s = np.stack([i - nrow, i - nrow + 1, i + 1, i + nrow + 1, i + nrow],
                    axis=1)
    cand = np.full((n, 8), -1, dtype=np.int64)
    cand[:, :] = full
    cand[north, :5] = nb_n[north]
    cand[north, 5:] = -1
    sel_s = south & ~north
    cand[sel_s, :5] = nb_s[sel_s]
    cand[sel_s, 5:] = -1
    valid = (cand >= 0) & (cand < n)
    pos = np.cumsum(valid, axis=1) - 1           # position in filtered list
    rows = np.repeat(i, 8).reshape(n, 8)[valid]
    cols = cand[valid]
    facs = np.where(pos[valid] % 2 == 1, sqrt(2.), 1.).astype(np.float32)
    return rows.astype(np.uint32), cols.astype(np.uint32), facs


def solve_potential(conductivity, move_dirn):
    """movmodel.py:87-128 (+ simulator.py:259-288): row-normalised conductance
    matrix, Dirichlet nodes, scipy SuperLU direct solve.  Returns f32 (R, C)."""
    import scipy.sparse as ss
    import scipy.sparse.linalg as ssl
    conductivity = np.asarray(conductivity, dtype=np.float64)
    nrow, ncol = conductivity.shape
    n = nrow * ncol
    bnodes, benergy = get_boundary_nodes(move_dirn, (nrow, ncol))
    r, c, facs = neighbour_lists((nrow, ncol))
    r = r.astype(np.int64)
    c = c.astype(np.int64)
    ca = conductivity[r % nrow, r // nrow]
    cb = conductivity[c % nrow, c // nrow]
    with np.errstate(divide='ignore'):
        hm = np.where((ca != 0) & (cb != 0), 2. / (1. / ca + 1 / cb), 1e-08)
    vals = hm / facs          # f64 / f4 -> f64, as `harmonic_mean(...) / fac`
    g = ss.coo_matrix((vals, (r, c)), shape=(n, n)).tocsr()
    # row-normalise (movmodel.py:110-112); NB on NumPy>=2 the reference's
    # `1e-08 / np.float32(fac)` is f32-rounded for zero-conductivity pairs -- a
    # 6e-8 relative wobble on a 1e-8 entry, far below the f32 output tolerance.
    row_sums = np.add.reduceat(g.data, g.indptr[:-1])
    g.data = g.data / row_sums[np.repeat(np.arange(n), np.diff(g.indptr))]
    inodes = np.setdiff1d(np.arange(n), bnodes, assume_unique=True)
    gi = g[inodes, :].tocoo().tocsc()
    gii = gi[:, inodes]
    gib = gi[:, bnodes]
    b_vec = gib.dot(benergy)
    a_mat = ss.eye(inodes.size).tocsc() - gii
    ienergy = ssl.spsolve(a_mat, b_vec)
    energy = np.empty(n)
    energy[inodes] = ienergy
    energy[bnodes] = benergy
    return energy.reshape(ncol, nrow).T.astype(np.float32)
