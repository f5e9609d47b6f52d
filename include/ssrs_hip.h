/* ssrs_hip.h -- C ABI of libssrs_hip.so, the MI355X (gfx950) implementation of
 * the SSRS data-parallel hot path: the updraft raster and the stochastic track
 * stepper.
 *
 * SSRS has no FFI/plugin seam of its own: the boundary this library replaces
 * is the set of module-level numeric functions that ssrs/simulator.py imports
 * by name (/root/reference/ssrs/simulator.py:21-28).  Each entry point below
 * cites the reference function it stands in for; INTEGRATION.md shows the
 * ctypes stub a maintainer of the reference would add at each call site.
 *
 * Conventions
 *  - every array pointer is CALLER-OWNED DEVICE memory (e.g. a torch tensor's
 *    data_ptr()), row-major (rows, cols), row 0 = south, unless marked [host];
 *  - `stream` is a hipStream_t passed as void* (NULL = the null stream);
 *  - raster entry points are asynchronous on `stream`; ssrs_tracks_simulate
 *    drives a launch loop and returns after the last launch has completed;
 *  - return value: SSRS_OK (0) or a negative SSRS_ERR_* code, with a
 *    thread-local message available from ssrs_last_error();
 *  - no hidden global state: the random stream is a pure function of
 *    (seed, global track id, step) -- see "Uniform contract" below;
 *  - thread-safe for distinct streams / devices.
 *
 * Uniform contract (replaces the reference's serial global MT19937 draw in
 * np.random.choice, movmodel.py:312, which no parallel run can reproduce):
 *   u(seed, track, step) = ((a >> 5) * 2^26 + (b >> 6)) / 2^53,
 *   (a, b) = words (0,1) [even step] or (2,3) [odd step] of
 *   Philox4x32-10(key = {seed lo, seed hi},
 *                 ctr = {blk lo, blk hi, track lo, track hi}),  blk = step >> 1,
 * i.e. exactly rocRAND's rocrand_init(seed, track, 2*step) + 2 x rocrand().
 */
#ifndef SSRS_HIP_H_
#define SSRS_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SSRS_VERSION 108 /* 0.1.8 */

#define SSRS_OK 0
#define SSRS_ERR_INVALID (-1) /* bad argument (message says which) */
#define SSRS_ERR_HIP (-2)     /* a HIP runtime call failed */
#define SSRS_ERR_START (-3)   /* a start cell lies outside the raster */

/* element type selectors for `const void*` rasters */
#define SSRS_F32 0
#define SSRS_F64 1

int ssrs_version(void);
/* bit 0: a timing-probe build of the library (a kernel stage is stubbed out on purpose, results are
 * wrong): product code, tests and bench.py refuse to run against it */
int ssrs_build_flags(void);
const char *ssrs_last_error(void);
/* name[] receives the device name; returns SSRS_OK or SSRS_ERR_HIP */
int ssrs_device_info(int device, char *name, size_t name_len, int *compute_units,
                     size_t *hbm_bytes);

/* ------------------------------------------------------------------ raster */

/* compute_slope_degrees + compute_aspect_degrees (ssrs/layers.py:63-128):
 * Horn 3x3 gradients on the DEM, slope = deg(atan|grad|), aspect per :124-127,
 * border cells 0.  slope or aspect may be NULL.  f64 arithmetic. */
int ssrs_slope_aspect(const void *dem, int dem_type, double res, void *slope,
                      void *aspect, int out_type, int rows, int cols, void *stream);

/* compute_orographic_updraft (ssrs/layers.py:11-22) for `batch` wind cases over
 * one terrain, optionally fused with get_above_threshold_speed
 * (ssrs/layers.py:171-185, applied to the f32-rounded orograph exactly as
 * Simulator.load_updrafts does after np.save(float32), simulator.py:198,233-242).
 *   slope, aspect   (rows, cols) of in_type
 *   wspeed, wdirn   NULL -> uniform mode, wspeed0[b] / wdirn0[b] ([host], length
 *                   batch); else (batch, rows, cols) rasters of wind_type
 *   orograph        (batch, rows, cols) f32 out, may be NULL
 *   threshold       < 0 -> `usable` is not written
 *   usable          (batch, rows, cols) f64 out (thresholded updraft), may be NULL */
int ssrs_orographic_updraft(const void *slope, const void *aspect, int in_type,
                            const void *wspeed, const void *wdirn, int wind_type,
                            const double *wspeed0, const double *wdirn0,
                            double min_updraft_val, float *orograph,
                            double threshold, double *usable, int rows, int cols,
                            int batch, void *stream);

/* get_above_threshold_speed (ssrs/layers.py:171-185) on n f32 values -> f64. */
int ssrs_threshold_updraft(const float *in, double threshold, double *out,
                           size_t n, void *stream);

/* Fused DEM -> orographic updraft (-> usable updraft): slope/aspect/orographic/
 * threshold of layers.py:11-22,63-128,171-185 in one pass over an LDS-staged
 * DEM tile, uniform wind.  Trig-free: sin(slope) cos(aspect - wdirn) is
 * evaluated from the Horn gradients directly (DESIGN.md "K1"); results agree
 * with the reference expression to a few f64 ulps before the f32 rounding.
 * orograph / usable as above (batch = 1). */
int ssrs_updraft_from_dem(const void *dem, int dem_type, double res, double wspeed,
                          double wdirn, double min_updraft_val, float *orograph,
                          double threshold, double *usable, int rows, int cols,
                          void *stream);

/* Snapshot / seasonal form of ssrs_updraft_from_dem (simulator.py:200-215 with the wind
 * preparation of :765-792 folded in): `batch` wind snapshots given as speed / direction
 * samples on a regular nx x ny lattice (as ssrs_wind_from_lattice; x0, y0, dx, dy in km,
 * the raster's cell size is res / 1000 km) -> orograph (batch, rows, cols) f32 and / or
 * usable updraft (batch, rows, cols) f64.  The DEM is read once for the whole batch and
 * no per-cell wind raster is materialised: bilinear east / north components at the cell,
 * w = -(Y north + X east) / sqrt(d^2 + X^2 + Y^2) (the wind speed cancels).
 * workspace: ssrs_lattice_workspace_bytes(nx, ny, batch) bytes of device scratch. */
size_t ssrs_lattice_workspace_bytes(int nx, int ny, int batch);
int ssrs_updraft_from_dem_lattice(const void *dem, int dem_type, double res,
                                  const double *lattice_speed, const double *lattice_dirn,
                                  int nx, int ny, double x0, double y0, double dx, double dy,
                                  double min_updraft_val, float *orograph, double threshold,
                                  double *usable, int rows, int cols, int batch,
                                  void *workspace, size_t workspace_bytes, void *stream);

/* Wind preparation of snapshot / seasonal modes (ssrs/simulator.py:778-792):
 * `batch` sets of speed/direction samples on a regular nx x ny lattice (origin
 * x0,y0 and spacings dx,dy in the raster's length unit, cell_size likewise;
 * lattice arrays (batch, ny, nx) f64) -> u/v components -> bilinear
 * interpolation to cell centres (clamped at the hull) -> per-cell speed and
 * direction in [0, 360), (batch, rows, cols) f64.  The reference triangulates
 * scattered points with scipy griddata; on a lattice bilinear is its equivalent. */
int ssrs_wind_from_lattice(const double *lattice_speed, const double *lattice_dirn,
                           int nx, int ny, double x0, double y0, double dx, double dy,
                           double cell_size, double *wspeed, double *wdirn, int rows,
                           int cols, int batch, void *stream);

/* The same for SCATTERED samples -- the reference's general case, ssrs/simulator.py:765-776:
 * scipy.interpolate.griddata(points, values, mesh, method='linear'), i.e. a Delaunay triangulation of the sample
 * points and barycentric interpolation inside each triangle, applied to the east / north components (:778-792).
 * The triangulation is the caller's: `points` (npts, 2) f64 in the raster's length unit relative to the centre of
 * cell (0, 0) (x along columns, y along rows), `triangles` (ntri, 3) int32 vertex indices and `transform` (ntri, 3, 2)
 * f64 exactly as scipy.spatial.Delaunay(points) holds them (.simplices, .transform: the 2 x 2 inverse of the edge
 * matrix and the offset r) -- griddata builds that very object.  speed / dirn (batch, npts) f64 -> wspeed / wdirn
 * (batch, rows, cols) f64, direction in [0, 360); cells outside the convex hull are NaN (griddata's fill value).  A
 * cell on an edge shared by two triangles takes the one with the lower index (the interpolant is continuous there;
 * scipy's walk picks either), so results agree with griddata to rounding, not bit for bit.
 * workspace: ssrs_wind_triangles_workspace_bytes(npts, rows, cols, batch) bytes of device scratch. */
size_t ssrs_wind_triangles_workspace_bytes(int npts, int rows, int cols, int batch);
int ssrs_wind_from_triangles(const double *points, const int32_t *triangles, const double *transform,
                             const double *speed, const double *dirn, int npts, int ntri,
                             double cell_size, double *wspeed, double *wdirn, int rows, int cols,
                             int batch, void *workspace, size_t workspace_bytes, void *stream);

/* compute_thermals (ssrs/layers.py:188-214), split in its two stages.
 * ssrs_thermal_seeds: per-cell seeding inside the 10 % border with probability
 * 1/(int(wt)-1), wt = 1000 + |aspect-180|/180*2000, amplitude
 * lognormal(scale + 3, 0.5); counter-based (Philox keyed by seed and cell) --
 * statistical parity only, the reference replays a serial global RNG.
 * ssrs_gaussian_blur: scipy.ndimage.gaussian_filter(sigma, mode='constant'),
 * separable, truncated at 4 sigma.  thermals = blur(seeds, sigma = 4). */
int ssrs_thermal_seeds(const double *aspect, double thermal_intensity_scale,
                       uint64_t seed, double *seeds, int rows, int cols, void *stream);
size_t ssrs_blur_workspace_bytes(int rows, int cols, double sigma);
int ssrs_gaussian_blur(const double *in, double *out, double sigma, int rows, int cols,
                       void *workspace, size_t workspace_bytes, void *stream);

/* ----------------------------------------------------------------- stepper */

/* Per-run constants of generate_simulated_tracks (ssrs/movmodel.py:264-318).
 * Fill with ssrs_track_params_init() or by hand. */
typedef struct SsrsTrackParams {
    int32_t rows, cols;
    int32_t burnin;           /* int(min(rows, cols) / 10), movmodel.py:276 */
    int32_t memory_parameter; /* directions[-m:], 0..8 (0 = whole history) */
    int64_t max_moves;        /* ceil(rows / 2 * cols / 2), movmodel.py:277 */
    double scaling_parameter; /* nu; exact parity is claimed for nu == 1 */
    double prior[9];          /* get_directional_probs(move_dirn * pi / 180),
                                 movmodel.py:247-257, computed by the host */
    int32_t steps_per_launch; /* 0 = default (512) */
    int32_t flags;            /* SSRS_TRACKS_* */
} SsrsTrackParams;

#define SSRS_TRACKS_PROFILE 1    /* time every launch with HIP events (stats) */
#define SSRS_TRACKS_NO_SCHEDULE 4 /* keep caller order, release every track at once
                                    (A/B switch for the coherent schedule) */
#define SSRS_TRACKS_NO_BINNING 8  /* histogram by per-step global atomics instead of the
                                    visit buffer + LDS binning kernel (A/B switch) */
#define SSRS_TRACKS_RING_TABLE 16 /* `table` is the f32 ring table of
                                    ssrs_transition_ring_build (one 12-byte gather per step);
                                    needs updraft (+ potential if the table was built with it)
                                    for the exact decision of near-ties, memory_parameter 1,
                                    scaling_parameter 1, traj NULL, even steps_per_launch */
#define SSRS_TRACKS_SCATTERED 32    /* treat the batch as scattered from the first launch: the
                                      stepper counts visits itself, into wave-private copies of the
                                      histogram when the workspace has room for them (tracks that
                                      circle in a pocket of the field otherwise queue up on single
                                      cells), and the ring stepper reads the per-cell zero-mask
                                      byte before gathering.  Default: large batches go from the
                                      per-step window to tile buckets and only then to this
                                      variant; small ones directly.  Results are identical */
#define SSRS_TRACKS_NO_SCATTERED 64 /* never switch to that variant (A/B) */
#define SSRS_TRACKS_THR_TABLE 128   /* `table` is the threshold table of ssrs_transition_thr_build (one 4-byte
                                      gather and two comparisons per step); needs updraft (+ potential
                                      if the table was built with it) for the exact decision of near-ties,
                                      memory_parameter 1, scaling_parameter 1, traj NULL, even steps_per_launch,
                                      and params->prior equal to the prior the table was built with */
#define SSRS_TRACKS_EXACT_ONLY 2 /* disable the guarded division-free decision
                                   (A/B switch; results are identical) */

typedef struct SsrsTrackStats {
    int64_t total_steps; /* moves taken by all tracks of this call */
    int32_t launches;    /* stepper kernel launches */
    float kernel_ms;     /* sum of stepper launch durations (SSRS_TRACKS_PROFILE) */
    float wall_ms;       /* first launch -> last completion, HIP events */
    float hist_ms;       /* sum of histogram-binning launch durations (PROFILE) */
    int32_t window_launches; /* launches whose visits were binned through the per-step row /
                                column window; */
    int32_t tile_launches;   /* ... through raster-tile buckets; the other launches counted
                                in the stepper (atomics on hist or on its private copies) */
    int32_t block_window_launches; /* ... in the stepper, into a histogram window per block in LDS
                                (threshold table; batches whose survivors roam a few basins) */
    int32_t wander_sorts;    /* times the live tracks were sorted into such windows */
    int32_t timed_launches;  /* stepper launches whose durations make up kernel_ms (PROFILE) */
    float first_move_ms;     /* of which: the one-iteration launch of the generic kernel that makes every
                                track's first move before the threshold stepper takes over (PROFILE; else 0) */
    float block_window_ms;   /* of which: the block-window launches (PROFILE; else 0) */
    int32_t block_window_timed;   /* how many of them were timed */
    int64_t block_window_steps;   /* moves taken in batches of block-window launches (from the per-batch
                                     read-backs; a batch is one or two launches of one kind) */
    int32_t roam_launches;        /* block-window launches that stepped through the windows' roam table
                                     (two moves per 64-byte entry, k_step_roam) */
    int32_t reserved0;            /* near-ties those launches settled with the 32-bit fine table */
    int64_t roam_wave_pairs;      /* pairs of moves run by the waves of those launches ... */
    int64_t roam_slow_wave_pairs; /* ... and how many of them sent some lane through the single-move sequence
                                     (near-ties, flag entries, moves out of the window, burn-in) */
    int32_t roam_shuffles;        /* times a SETTLED roaming batch had the tracks of its windows dealt afresh
                                     (every SSRS_TRACKS_ROAM_SHUFFLE batches, default 16; part of wander_sorts) */
    int32_t roam_wide_launches;   /* of roam_launches: those run with 512-lane blocks (two list blocks of one window per CU: batches
                                     whose survivors outnumber one round of 256-lane blocks; SSRS_TRACKS_ROAM_WIDE) */
} SsrsTrackStats;

/* Fills rows/cols/burnin/max_moves/memory/nu and zeroes the rest; `prior` must
 * still be supplied by the caller (it needs the host libm/numpy cos). */
int ssrs_track_params_init(SsrsTrackParams *p, int rows, int cols,
                           int memory_parameter, double scaling_parameter);

/* Per-cell move weights of movmodel.py:292-306 for every interior cell:
 *   w_k = hm(max(u_c,1e-6), max(u_k,1e-6)) * f64(f32(phi_c - phi_k) * ninv_k),
 * clipped at 0 (:231), the 8 neighbours k = 0,1,2,3,5,6,7,8 of one cell stored
 * as 8 consecutive f64 (64 B, one aligned fetch per step).  A cell with any NaN
 * weight stores NaN in all 8 (the stepper then takes the :228-230 fallback).
 * potential may be NULL (updraft-only weights).  table: rows*cols*8 f64. */
int ssrs_transition_table_build(const double *updraft, const float *potential,
                                double *table, int rows, int cols, void *stream);

/* The same weights for the three-candidate stepper (SSRS_TRACKS_RING_TABLE): after
 * a move only the three cells within +-45 deg are admissible (movmodel.py:185-202),
 * and in clockwise ring order N, NE, E, SE, S, SW, W, NW they are consecutive.  Per
 * cell 10 f32 = the ring-ordered weights rounded to f32, stored as ring 7, 0, 1, ...,
 * 7, 0 (40 B), exact zeros as -0.0f; behind the records one zero-mask byte per cell (bit c:
 * ring weight c is exactly zero).  ring: ssrs_transition_ring_bytes(rows, cols) bytes,
 * 8-byte aligned. */
size_t ssrs_transition_ring_bytes(int rows, int cols);
int ssrs_transition_ring_build(const double *updraft, const float *potential, float *ring,
                               int rows, int cols, void *stream);

/* The decision thresholds themselves (SSRS_TRACKS_THR_TABLE): for every cell and every last move
 * rc (ring position 0..7) one dword T1 | T2 << 16 with T1 = round(2^16 a / (a + b + c)), T2 =
 * round(2^16 (a + b) / (a + b + c)) (clamped to 65535), a, b, c the three admissible weights of
 * movmodel.py:292-309 in ascending neighbour index -- the boundaries np.random.choice's inverse-cdf
 * pick compares the uniform with; the stepper decides on the uniform's top 16 bits and hands a
 * uniform within one unit of a boundary (3e-5 per boundary) to the exact sequence.  A row whose
 * weights are all zero carries the masked prior's thresholds (movmodel.py:234-238); boundary cells,
 * poisoned rows and rows where the unmasked prior decides are flag entries (T1 = 0xFFFF > T2 = code).
 * Eight planes (one per last move) of 4-byte entries at a power-of-two stride,
 * ssrs_transition_thr_bytes(rows, cols) bytes in all, 64-byte aligned; the table belongs to one
 * heading (`prior` [host], 9 doubles = SsrsTrackParams.prior).  rows * cols <= 2^26; the
 * table carries a guard band of (cols + 2) dwords at either end, and the first bytes of the leading band
 * name the table (magic, rows, cols, the nine prior values): ssrs_tracks_simulate reads them back and returns
 * SSRS_ERR_INVALID, before any stepper kernel is launched, for a buffer that is not the threshold table of its
 * raster and prior (88 bytes and one stream wait per call). */
size_t ssrs_transition_thr_bytes(int rows, int cols);
int ssrs_transition_thr_build(const double *updraft, const float *potential, const double *prior,
                              float *thr, int rows, int cols, void *stream);

/* Bytes of device scratch ssrs_tracks_simulate needs for `ntracks` (about 8.3 KB per
 * track: two buffers of 1024 steps x 4 B for the launch's visited cells, in slot and in
 * raster-tile order; a launch takes as many steps as they hold for the live tracks).
 * With the threshold table, a batch whose survivors roam a few basins of the potential
 * field until max_moves (the solved 10 m field: 44 %) is sorted into up to 16 histogram
 * windows of 144 x 256 cells and counted in LDS (SsrsTrackStats.block_window_launches);
 * nothing to size for it. */
size_t ssrs_tracks_workspace_bytes(int64_t ntracks);
/* The same plus (i) the pair table of the roaming regime and the fine table of its near-ties -- 128 + 64
 * bytes per cell (5.76 GB at 5000 x 6000),
 * batches of >= 8192 tracks on rasters below 2^25 cells: with it the block-window launches take two moves
 * per 16-byte gather (SsrsTrackStats.roam_launches), without it they run round 2's one-gather-per-move
 * kernel, 2.1x slower -- and (ii) room for `hist_copies` (2..64) private copies of the histogram.  A
 * workspace of this size lets ssrs_tracks_simulate privatise the histogram once a batch
 * is scattered (many tracks circling in the same pockets of a real potential field make
 * per-step atomics on single cells queue up at the memory side: 3x slower); the copies
 * are added to `hist` before the call returns.  Optional: the smaller workspace works. */
size_t ssrs_tracks_workspace_bytes_ex(int64_t ntracks, int rows, int cols, int hist_copies);

/* generate_simulated_tracks for a batch of tracks (movmodel.py:264-318, driven
 * as Simulator.simulate_tracks does, simulator.py:360-369) + the histogram of
 * compute_presence_counts (movmodel.py:410-419).
 *   updraft    f64 (rows, cols) or NULL;  potential f32 (rows, cols) or NULL
 *              (both NULL = 'drw' mode, simulator.py:370-381)
 *   table      from ssrs_transition_table_build (f64 rows; updraft and potential are
 *              then not read), from ssrs_transition_ring_build with the flag
 *              SSRS_TRACKS_RING_TABLE (updraft / potential are read for the exact
 *              decision of near-ties), or NULL to gather the 3x3 windows of
 *              updraft/potential directly
 *   start_rc   int32 (ntracks, 2) [row, col]
 *   seed       sim_seed + real_id (simulator.py:352); track_id_base = global id
 *              of track 0 of this call (multi-GPU shards pass their offset)
 *   hist       uint32 (rows, cols), ACCUMULATED (+1 per trajectory point), or NULL
 *   end_rc     int16 (ntracks, 2) last point, or NULL
 *   lengths    int32 (ntracks) number of trajectory points, or NULL
 *   traj       int16 pairs, track t at traj[2*traj_offsets[t] ...], or NULL;
 *              traj_offsets int64 (ntracks + 1): exclusive prefix sums of the
 *              lengths of a previous call with the same arguments; points beyond
 *              a track's room [offsets[t], offsets[t+1]) are dropped, never written
 *   workspace  device scratch of ssrs_tracks_workspace_bytes(ntracks)
 *   stats      [host] out, may be NULL */
int ssrs_tracks_simulate(const SsrsTrackParams *params, const double *updraft,
                         const float *potential, const double *table,
                         const int32_t *start_rc, int64_t ntracks, uint64_t seed,
                         uint64_t track_id_base, uint32_t *hist, int16_t *end_rc,
                         int32_t *lengths, int16_t *traj,
                         const int64_t *traj_offsets, void *workspace,
                         size_t workspace_bytes, SsrsTrackStats *stats,
                         void *stream);

/* ssrs_tracks_simulate (no trajectories) with the presence counts in 64 bits: compute_presence_counts' int16 raster
 * (movmodel.py:415) wraps at 32 767 visits and this library's uint32 one at 2^32 - 1, which the trap cells of a solved 10 m
 * field reach from ~250 000 tracks of one call on (1.7e4 visits per track).  The kernels count into `hist_scratch`
 * (uint32 (rows, cols), zeroed by the caller: whatever it holds is counted too) and the library empties it into `hist64`
 * (uint64 (rows, cols), ACCUMULATED) every other batch of launches and before it returns, on `stream`. */
int ssrs_tracks_simulate_h64(const SsrsTrackParams *params, const double *updraft,
                             const float *potential, const double *table,
                             const int32_t *start_rc, int64_t ntracks, uint64_t seed,
                             uint64_t track_id_base, uint32_t *hist_scratch, uint64_t *hist64,
                             int16_t *end_rc, int32_t *lengths, void *workspace,
                             size_t workspace_bytes, SsrsTrackStats *stats, void *stream);

/* Trajectory output in ONE simulation pass (the List[int16 (n_i, 2)] that
 * Simulator.simulate_tracks pickles, simulator.py:360-385).  The two-call form above needs
 * the lengths of an earlier identical simulation for `traj_offsets`.  With a recorder the
 * stepper keeps every launch's visited cells in a caller-supplied device pool (4 bytes per
 * step and live slot); when the call returns the lengths are final, the caller forms the
 * offsets (exclusive prefix sums of `lengths`), allocates `traj` and ssrs_tracks_gather
 * appends every track's points in order.  Every stepper path (ring table included) records.
 *   pool        device scratch, 256-byte aligned; when it is exhausted the simulation goes
 *               on unrecorded (results unaffected), ssrs_traj_recorder_complete() returns 0
 *               and the caller falls back to the two-call form or a larger pool
 *   recorder    reusable: each ssrs_tracks_simulate_rec call starts a fresh record */
typedef struct SsrsTrajRecorder SsrsTrajRecorder;
SsrsTrajRecorder *ssrs_traj_recorder_create(void *pool, size_t pool_bytes);
void ssrs_traj_recorder_destroy(SsrsTrajRecorder *recorder);
int ssrs_traj_recorder_complete(const SsrsTrajRecorder *recorder);
size_t ssrs_traj_recorder_used(const SsrsTrajRecorder *recorder);
/* ssrs_tracks_simulate without traj / traj_offsets, recording into `recorder` */
int ssrs_tracks_simulate_rec(const SsrsTrackParams *params, const double *updraft,
                             const float *potential, const double *table,
                             const int32_t *start_rc, int64_t ntracks, uint64_t seed,
                             uint64_t track_id_base, uint32_t *hist, int16_t *end_rc,
                             int32_t *lengths, SsrsTrajRecorder *recorder, void *workspace,
                             size_t workspace_bytes, SsrsTrackStats *stats, void *stream);
/* traj[2 * traj_offsets[t] ...] <- the int16 (row, col) points of track t, start cell first
 * (asynchronous on `stream`; the pool must stay alive until it has run).
 *   start_rc, ntracks   as passed to ssrs_tracks_simulate_rec
 *   traj_offsets        int64 (ntracks + 1), exclusive prefix sums of `lengths`; points
 *                       beyond a track's room are dropped, never written
 *   cursor_ws           device scratch, 4 bytes per track */
int ssrs_tracks_gather(const SsrsTrajRecorder *recorder, const int32_t *start_rc,
                       int64_t ntracks, const int64_t *traj_offsets, int16_t *traj,
                       void *cursor_ws, size_t cursor_bytes, void *stream);

/* The one exchange step of a track-sharded run (the reference maps the tracks of a
 * case over a process pool, simulator.py:360-369, and adds them up in compute_presence_counts):
 * in-place sum of the ranks' uint32 histograms over xGMI.  `nccl_comm` is an ncclComm_t the caller
 * created with its own RCCL (one process per GPU); root >= 0: ncclReduce to that rank, root < 0:
 * ncclAllReduce.  Asynchronous on `stream`.  RCCL is resolved at run time (the process's own copy
 * first, else librccl.so.1), so the library loads without it.  The sum is 32-bit: a caller whose
 * ranks' largest counts add up to 2^32 or more must widen first (ssrs_amd.distributed does). */
int ssrs_hist_reduce(uint32_t *hist, size_t n, int root, void *nccl_comm, void *stream);

/* --------------------------------------------------------------- presence */

/* compute_presence_counts (ssrs/movmodel.py:410-419) from stored trajectories:
 * hist[row, col] += 1 for each of the npoints int16 (row, col) pairs.  uint32
 * counts (the reference's int16 matrix wraps above 32767 visits).  A point
 * outside the raster is an error (the reference raises IndexError).
 * scratch8: 8 bytes of device scratch. */
int ssrs_presence_count(const int16_t *traj, int64_t npoints, uint32_t *hist,
                        int rows, int cols, void *scratch8, void *stream);

size_t ssrs_presence_workspace_bytes(int rows, int cols, int krad);

/* compute_smooth_presence_counts (ssrs/movmodel.py:422-439) on a count matrix:
 * zero-padded 'same' convolution with the disk kernel (x^2+y^2 <= krad^2)/ntaps,
 * f32 out.  Evaluated as 2*krad+1 chords of row prefix sums: exact in integers,
 * then one multiply by 1/ntaps. */
int ssrs_presence_smooth(const uint32_t *count, int krad, float *out, int rows,
                         int cols, void *workspace, size_t workspace_bytes,
                         void *stream);

/* The same for 64-bit counts: the sum of the ranks' histograms when a 32-bit sum could wrap
 * (tracks that circle in a pocket of the field until max_moves put ~1e9 visits into single
 * cells per 100k tracks).  The chord sums are 64-bit either way. */
int ssrs_presence_smooth_u64(const uint64_t *count, int krad, float *out, int rows,
                             int cols, void *workspace, size_t workspace_bytes,
                             void *stream);

/* The normalisation ladder of Simulator.plot_presence_map (simulator.py:529-546):
 *   acc += src / max(src)     (division in src's precision: f32 for `prprob`,
 *                              f64 for `case_prob`)
 *   out  = f32(src / max(src)) (summary_presence.npy) */
int ssrs_presence_normalise_add(const void *src, int src_type, double *acc, size_t n,
                                void *scratch8, void *stream);
int ssrs_presence_normalise_f32(const double *src, float *out, size_t n,
                                void *scratch8, void *stream);

/* -------------------------------------------------------------- potential */

typedef struct SsrsSolveStats {
    int32_t iterations;
    int32_t converged; /* 1 when |r| <= rel_tol |b| was reached */
    double residual;   /* final |r| / |b| */
    float kernel_ms;
    int32_t amg_levels;   /* levels of the aggregation hierarchy (0 = none) */
    int32_t amg_coarsest; /* nodes on its last level */
    float setup_ms;       /* building the hierarchy (not part of kernel_ms) */
    uint64_t workspace_used; /* bytes of `workspace` really touched (the size query is an upper bound) */
} SsrsSolveStats;

#define SSRS_SOLVE_NO_AMG 1  /* plain BiCGStab (A/B switch; stalls on real rasters) */
#define SSRS_SOLVE_K_CYCLE 2 /* K-cycle (two flexible-CG steps per coarse solve) on the
                               first coarse levels instead of the V-cycle; depth in flag
                               bits 12-15 (0 = 3 levels) */
#define SSRS_SOLVE_ONE_SIDED 4 /* aggregation strength relative to the row maximum only
                                 (A/B switch: the previous criterion; pairs dead with live
                                 cells and needs 2-3x the iterations) */
/* flags bits 4-6: extra pairs of Jacobi sweeps; bits 8-11: strict matching rounds of
 * the one-sided criterion (0 = 4) */

/* Device scratch that always suffices (about 1.5 KB per cell).  The hierarchy really takes ~840 B
 * per cell (SsrsSolveStats.workspace_used reports it); a smaller workspace is accepted and the call
 * fails with SSRS_ERR_INVALID ("workspace exhausted") when it does not suffice. */
size_t ssrs_potential_workspace_bytes(int rows, int cols);

/* MovModel.assemble_sparse_linear_system + solve_sparse_linear_system
 * (ssrs/movmodel.py:59-128) without assembling anything: the row-normalised
 * 8-neighbour conductance operator is applied matrix-free and the Dirichlet
 * problem is solved in f64 by BiCGStab, right-preconditioned with one V-cycle of
 * an aggregation AMG built on the device from the same conductances (the
 * reference factorises with SuperLU).
 *   conductivity  f64 (rows, cols): the usable updraft
 *   fixed_mask    u8  (rows, cols): 1 on Dirichlet cells (get_boundary_nodes,
 *                 movmodel.py:21-57, evaluated by the host)
 *   fixed_values  f64 (rows, cols): boundary energy on those cells
 *   initial_guess f64 (rows, cols) or NULL
 *   potential     f32 (rows, cols) out, as `pot_energy.astype(np.float32)` */
int ssrs_potential_solve(const double *conductivity, const uint8_t *fixed_mask,
                         const double *fixed_values, const double *initial_guess,
                         float *potential, int rows, int cols, double rel_tol,
                         int max_iterations, int flags, void *workspace,
                         size_t workspace_bytes,
                         void *stats /* SsrsSolveStats*, [host], may be NULL */,
                         void *stream);

/* n uniforms of the contract above: out[i] = u(seed, track[i], step[i]).
 * Device self-check of the rocRAND-backed draw used by the stepper. */
int ssrs_uniform_selftest(uint64_t seed, const uint64_t *track,
                          const uint64_t *step, double *out, size_t n,
                          void *stream);

#ifdef __cplusplus
}
#endif
#endif /* SSRS_HIP_H_ */
