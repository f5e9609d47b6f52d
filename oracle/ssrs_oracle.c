/* ORACLE -- plain-C CPU restatement of the SSRS hot path (TEST INFRASTRUCTURE).
 *
 * Never linked into, loaded by, or called from the product path (ssrs_amd/,
 * libssrs_hip.so).  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg load it (through oracle/c_oracle.py).
 *
 * It follows the reference algorithm (paths relative to /root/reference):
 *   ssrs/layers.py:11-22     compute_orographic_updraft
 *   ssrs/layers.py:63-128    compute_slope_degrees / compute_aspect_degrees
 *   ssrs/layers.py:171-185   get_above_threshold_speed
 *   ssrs/movmodel.py:185-261 restrictions, nudge, move probabilities
 *   ssrs/movmodel.py:264-318 generate_simulated_tracks
 *   ssrs/movmodel.py:410-419 compute_presence_counts
 *   ssrs/movmodel.py:422-439 compute_smooth_presence_counts
 * and the Philox uniform contract documented in oracle/philox.py.
 *
 * Parity status: PINNED -- bit-exact against tests/golden/g7_tracks.npz
 * (trajectories produced by the reference itself) and against the numpy
 * restatement oracle/ssrs_oracle.py (tests/test_oracle_golden.py).
 *
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off: no FMA fusion, so
 * every + - * / rounds exactly like numpy's scalar/SSE2 double arithmetic).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------ Philox */
static inline void philox_round(uint32_t c[4], uint32_t k0, uint32_t k1)
{
    uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
    uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
    uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0;
    uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
    c[1] = (uint32_t)p1;
    c[3] = (uint32_t)p0;
    c[0] = n0;
    c[2] = n2;
}

void orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4])
{
    uint32_t c[4] = {ctr[0], ctr[1], ctr[2], ctr[3]};
    uint32_t k0 = key[0], k1 = key[1];
    for (int r = 0; r < 10; ++r) {
        philox_round(c, k0, k1);
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    memcpy(out, c, sizeof(c));
}

double orc_uniform(uint64_t seed, uint64_t track, uint64_t step)
{
    uint64_t blk = step >> 1;
    uint32_t ctr[4] = {(uint32_t)blk, (uint32_t)(blk >> 32),
                       (uint32_t)track, (uint32_t)(track >> 32)};
    uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
    uint32_t w[4];
    orc_philox4x32_10(ctr, key, w);
    uint32_t a = (step & 1) ? w[2] : w[0];
    uint32_t b = (step & 1) ? w[3] : w[1];
    return ((double)(a >> 5) * 67108864.0 + (double)(b >> 6)) / 9007199254740992.0;
}

/* ------------------------------------------------------------------ raster */
/* layers.py:63-128; "x" is the row axis, "y" the column axis (sic). */
void orc_slope_aspect(const double *z, int rows, int cols, double res,
                      double *slope, double *aspect)
{
    const double r2d = 180.0 / M_PI;
#pragma omp parallel for schedule(static)
    for (int r = 0; r < rows; ++r) {
        for (int c = 0; c < cols; ++c) {
            size_t i = (size_t)r * cols + c;
            if (r == 0 || c == 0 || r == rows - 1 || c == cols - 1) {
                if (slope) slope[i] = 0.0;
                if (aspect) aspect[i] = 0.0;
                continue;
            }
            const double *zm = z + (size_t)(r - 1) * cols + c;
            const double *z0 = z + (size_t)r * cols + c;
            const double *zp = z + (size_t)(r + 1) * cols + c;
            double z1 = zm[1], z2 = z0[1], z3 = zp[1];
            double z4 = zm[0], z6 = zp[0];
            double z7 = zm[-1], z8 = z0[-1], z9 = zp[-1];
            double dzdx = ((z3 + 2 * z6 + z9) - (z1 + 2 * z4 + z7)) / (8 * res);
            double dzdy = ((z1 + 2 * z2 + z3) - (z7 + 2 * z8 + z9)) / (8 * res);
            if (slope)
                slope[i] = atan(sqrt(dzdx * dzdx + dzdy * dzdy)) * r2d;
            if (aspect) {
                double dx = dzdx == 0.0 ? 1e-10 : dzdx;
                double ang = atan(dzdy / dx) * r2d;
                double mod = 90.0 * (dx / fabs(dx));
                aspect[i] = 180.0 - ang + mod;
            }
        }
    }
}

/* layers.py:11-22 */
void orc_orographic(const double *slope, const double *aspect,
                    const double *wspeed, const double *wdirn,
                    double wspeed0, double wdirn0, double min_val,
                    size_t n, double *out64, float *out32)
{
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n; ++i) {
        double ws = wspeed ? wspeed[i] : wspeed0;
        double wd = wdirn ? wdirn[i] : wdirn0;
        double ad = cos((aspect[i] - wd) * M_PI / 180.0);
        ad = ad > 0.0 ? ad : 0.0;
        double v = ws * (sin(slope[i] * M_PI / 180.0) * ad);
        v = v > min_val ? v : min_val;
        if (out64) out64[i] = v;
        if (out32) out32[i] = (float)v;
    }
}

/* layers.py:171-185 */
void orc_threshold(const float *in, double thr, size_t n, double *out)
{
    const double em1 = exp(1.0) - 1.0;
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n; ++i) {
        double v = (double)in[i];
        double f = 0.0;
        if (v > 1e-02)
            f = v > thr ? v : thr * (exp(pow(v / thr, 5.0)) - 1.0) / em1;
        out[i] = f;
    }
}

/* ----------------------------------------------------------------- stepper */
typedef struct {
    int32_t rows, cols;
    int32_t burnin;          /* int(min(rows, cols) / 10)       movmodel.py:276 */
    int32_t memory;          /* memory_parameter (0 = whole history, as py [-0:]) */
    double max_moves;        /* rows / 2 * cols / 2 (float)     movmodel.py:277 */
    double nu;               /* scaling_parameter */
    double prior[9];         /* get_directional_probs(move_dirn*pi/180) */
} orc_params;

static const int8_t DR[9] = {-1, -1, -1, 0, 0, 0, 1, 1, 1};
static const int8_t DC[9] = {-1, 0, 1, -1, 0, 1, -1, 0, 1};
/* f32(1/sqrt(2)) as stored in neighbour_delta_norms_inv (movmodel.py:133-141) */
static const float NINV[9] = {0.70710677f, 1.f, 0.70710677f, 1.f, 0.f, 1.f,
                              0.70710677f, 1.f, 0.70710677f};

/* movmodel.py:185-202 as a 9-bit mask per previous direction index */
static uint16_t restriction_mask(int d)
{
    int dr = DR[d], dc = DC[d];
    uint16_t m = 0;
    for (int k = 0; k < 9; ++k) {
        int ok;
        if (dr == 0 && dc == 0) ok = 1;
        else if (dr != 0 && dc != 0) ok = (DR[k] == dr || DR[k] == 0) && (DC[k] == dc || DC[k] == 0);
        else if (dr == 0) ok = DC[k] == dc;
        else ok = DR[k] == dr;
        if (ok && k != 4) m |= (uint16_t)(1u << k);
    }
    return m;
}

static inline double sum9(const double *x)
{   /* numpy pairwise summation, n = 9 */
    return (((x[0] + x[1]) + (x[2] + x[3])) + ((x[4] + x[5]) + (x[6] + x[7]))) + x[8];
}

static inline int all_zero9(const double *x)
{
    for (int k = 0; k < 9; ++k) if (x[k] != 0.0) return 0;
    return 1;
}

/* movmodel.py:220-244 + np.random.choice's cdf search */
static int choose_move(const double *w, const double *prior, double nu,
                       uint16_t mask, double u)
{
    double q[9];
    int has_nan = 0;
    for (int k = 0; k < 9; ++k) has_nan |= (w[k] != w[k]);
    for (int k = 0; k < 9; ++k) {
        double v = has_nan ? prior[k] : w[k];
        q[k] = v > 0.0 ? v : 0.0;
    }
    q[4] = 0.0;
    for (int k = 0; k < 9; ++k) q[k] = q[k] * (double)((mask >> k) & 1);
    if (all_zero9(q)) {
        for (int k = 0; k < 9; ++k) q[k] = prior[k];
        q[4] = 0.0;
        for (int k = 0; k < 9; ++k) q[k] = q[k] * (double)((mask >> k) & 1);
        if (all_zero9(q))
            for (int k = 0; k < 9; ++k) q[k] = prior[k];
    }
    double s1 = sum9(q);
    for (int k = 0; k < 9; ++k) q[k] = q[k] / s1;
    if (nu != 1.0)
        for (int k = 0; k < 9; ++k) q[k] = pow(q[k], nu);
    double s2 = sum9(q);
    for (int k = 0; k < 9; ++k) q[k] = q[k] / s2;
    double cdf[9], acc = 0.0;
    for (int k = 0; k < 9; ++k) { acc = acc + q[k]; cdf[k] = acc; }
    int idx = 0;
    for (int k = 0; k < 9; ++k) idx += (cdf[k] / cdf[8] <= u);
    return idx;
}

/* One track (movmodel.py:264-318).  Returns the trajectory length (points). */
static int64_t run_track(const orc_params *p, const double *updraft,
                         const float *potential, int row, int col,
                         uint64_t seed, uint64_t track_id, const uint16_t *rmask,
                         uint32_t *hist, int16_t *traj, int16_t *end_rc)
{
    const int R = p->rows, C = p->cols;
    int64_t k = 0, npts = 1;
    int mem = p->memory;
    uint8_t ring[64];
    int nring = 1;
    ring[0] = 4;                                 /* initial direction [0, 0] */
    uint16_t running = rmask[4];                 /* for memory == 0 */
    if (traj) { traj[0] = (int16_t)row; traj[1] = (int16_t)col; }
    if (hist) {
#pragma omp atomic
        hist[(size_t)row * C + col] += 1;
    }
    while ((double)k < p->max_moves) {
        if (k > p->burnin) {
            if (!(0 < row && row < R - 1 && 0 < col && col < C - 1)) break;
        } else {                                 /* movmodel.py:205-217 */
            if (row <= 1) row += 2; else if (row >= R - 2) row -= 2;
            if (col <= 0) col += 2; else if (col >= C - 2) col -= 2;
        }
        double w[9];
        if (updraft) {
            double win[9];
            for (int j = 0; j < 9; ++j) {
                double v = updraft[(size_t)(row + DR[j]) * C + (col + DC[j])];
                win[j] = v != v ? v : (v > 1e-06 ? v : 1e-06);
            }
            double ic = 1.0 / win[4];
            for (int j = 0; j < 9; ++j) w[j] = 2.0 / (ic + 1.0 / win[j]);
        } else {
            for (int j = 0; j < 9; ++j) w[j] = p->prior[j];
        }
        if (potential) {
            float pc = potential[(size_t)row * C + col];
            for (int j = 0; j < 9; ++j) {
                float d = pc - potential[(size_t)(row + DR[j]) * C + (col + DC[j])];
                float e = d * NINV[j];
                w[j] = w[j] * (double)e;
            }
        }
        uint16_t mask = rmask[4];
        if (mem == 0) mask = running;
        else {
            int cnt = nring < mem ? nring : mem;
            for (int j = 0; j < cnt; ++j) mask &= rmask[ring[(nring - 1 - j) & 63]];
        }
        double u = orc_uniform(seed, track_id, (uint64_t)k);
        int idx = choose_move(w, p->prior, p->nu, mask, u);
        row += DR[idx];
        col += DC[idx];
        ring[nring & 63] = (uint8_t)idx;
        nring++;
        if (nring >= 128) nring -= 64;           /* keep index bounded, same slots */
        running &= rmask[idx];
        if (traj) { traj[2 * npts] = (int16_t)row; traj[2 * npts + 1] = (int16_t)col; }
        if (hist) {
#pragma omp atomic
            hist[(size_t)row * C + col] += 1;
        }
        npts++;
        k++;
    }
    if (end_rc) { end_rc[0] = (int16_t)row; end_rc[1] = (int16_t)col; }
    return npts;
}

/* Simulate `ntracks` tracks.  hist/end_rc/lengths/traj may each be NULL.
 * traj layout: int16 pairs at traj[2*traj_offsets[t] ...].  Returns total steps
 * (sum of lengths - ntracks) or -1 on bad arguments. */
int64_t orc_simulate_tracks_ids(const orc_params *p, const double *updraft,
                                const float *potential, const int32_t *start_rc,
                                int64_t ntracks, uint64_t seed, uint64_t track_id_base,
                                const uint64_t *track_ids,
                                uint32_t *hist, int16_t *end_rc, int32_t *lengths,
                                int16_t *traj, const int64_t *traj_offsets, int nthreads);

int64_t orc_simulate_tracks(const orc_params *p, const double *updraft,
                            const float *potential, const int32_t *start_rc,
                            int64_t ntracks, uint64_t seed, uint64_t track_id_base,
                            uint32_t *hist, int16_t *end_rc, int32_t *lengths,
                            int16_t *traj, const int64_t *traj_offsets, int nthreads)
{
    return orc_simulate_tracks_ids(p, updraft, potential, start_rc, ntracks, seed, track_id_base, NULL,
                                   hist, end_rc, lengths, traj, traj_offsets, nthreads);
}

/* The same for an arbitrary SUBSET of a batch: track t draws from the stream of global id
 * track_ids[t] (NULL: track_id_base + t) -- the tracks of a batch are independent given their ids
 * (simulator.py:360 maps them over a pool), so any subset can be checked on its own. */
int64_t orc_simulate_tracks_ids(const orc_params *p, const double *updraft,
                                const float *potential, const int32_t *start_rc,
                                int64_t ntracks, uint64_t seed, uint64_t track_id_base,
                                const uint64_t *track_ids,
                                uint32_t *hist, int16_t *end_rc, int32_t *lengths,
                                int16_t *traj, const int64_t *traj_offsets, int nthreads)
{
    if (!p || p->rows < 5 || p->cols < 5 || p->memory < 0 || p->memory > 60) return -1;
    if (potential && !updraft) return -1;
    uint16_t rmask[9];
    for (int d = 0; d < 9; ++d) rmask[d] = restriction_mask(d);
    int64_t total = 0;
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#else
    (void)nthreads;
#endif
#pragma omp parallel for schedule(dynamic, 1) reduction(+ : total)
    for (int64_t t = 0; t < ntracks; ++t) {
        int16_t *tj = (traj && traj_offsets) ? traj + 2 * traj_offsets[t] : NULL;
        int64_t n = run_track(p, updraft, potential, start_rc[2 * t], start_rc[2 * t + 1],
                              seed, track_ids ? track_ids[t] : track_id_base + (uint64_t)t, rmask, hist, tj,
                              end_rc ? end_rc + 2 * t : NULL);
        if (lengths) lengths[t] = (int32_t)n;
        total += n - 1;
    }
    return total;
}

/* Raw 8-neighbour move weights of one cell (clipped at 0, NaN-poisoned), the
 * quantity the HIP transition-table kernel precomputes; order k = 0,1,2,3,5,6,7,8. */
void orc_cell_weights(const double *updraft, const float *potential, int rows,
                      int cols, int row, int col, double out[8])
{
    (void)rows;
    double w[9], win[9];
    for (int j = 0; j < 9; ++j) {
        double v = updraft[(size_t)(row + DR[j]) * cols + (col + DC[j])];
        win[j] = v != v ? v : (v > 1e-06 ? v : 1e-06);
    }
    double ic = 1.0 / win[4];
    float pc = potential ? potential[(size_t)row * cols + col] : 0.f;
    int has_nan = 0;
    for (int j = 0; j < 9; ++j) {
        w[j] = 2.0 / (ic + 1.0 / win[j]);
        if (potential) {
            float d = pc - potential[(size_t)(row + DR[j]) * cols + (col + DC[j])];
            float e = d * NINV[j];
            w[j] = w[j] * (double)e;
        }
        has_nan |= (w[j] != w[j]);
    }
    for (int j = 0, o = 0; j < 9; ++j) {
        if (j == 4) continue;
        out[o++] = has_nan ? NAN : (w[j] > 0.0 ? w[j] : 0.0);
    }
}

/* ---------------------------------------------------------------- presence */
/* movmodel.py:422-439: disk kernel (x^2+y^2 <= k^2)/count, zero-padded 'same'
 * convolution of the count matrix, f32 result.  Direct evaluation, f64 acc in
 * scipy's order is not reproduced (tolerance-based check). */
void orc_smooth_presence(const uint32_t *count, int rows, int cols, int krad,
                         float *out)
{
    int64_t ntaps = 0;
    for (int y = -krad; y <= krad; ++y)
        for (int x = -krad; x <= krad; ++x)
            if (x * x + y * y <= krad * krad) ntaps++;
    const double wgt = 1.0 / (double)ntaps;
#pragma omp parallel for schedule(static)
    for (int r = 0; r < rows; ++r) {
        for (int c = 0; c < cols; ++c) {
            uint64_t acc = 0;
            for (int y = -krad; y <= krad; ++y) {
                int rr = r + y;
                if (rr < 0 || rr >= rows) continue;
                int half = (int)floor(sqrt((double)(krad * krad - y * y)));
                int c0 = c - half < 0 ? 0 : c - half;
                int c1 = c + half >= cols ? cols - 1 : c + half;
                const uint32_t *row = count + (size_t)rr * cols;
                for (int cc = c0; cc <= c1; ++cc) acc += row[cc];
            }
            out[(size_t)r * cols + c] = (float)((double)acc * wgt);
        }
    }
}

int orc_num_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
