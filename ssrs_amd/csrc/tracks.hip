// K2/K3 -- stochastic track stepper + presence histogram for gfx950 (MI355X).
//
// Reference semantics (paths relative to /root/reference):
//   ssrs/movmodel.py:185-202  get_track_restrictions      -> restriction()
//   ssrs/movmodel.py:205-217  move_away_from_boundary     -> nudge in step loop
//   ssrs/movmodel.py:220-244  generate_move_probabilities -> choose_move()
//   ssrs/movmodel.py:264-318  generate_simulated_tracks   -> k_step_thr, k_step_lean, k_step_tracks
//   ssrs/movmodel.py:410-419  compute_presence_counts     -> k_bin_visits[16], k_bin_bucket, k_step_thr<6> / uint32 atomics
//   numpy mtrand `choice`     cumsum, /last, searchsorted 'right'
//
// Structure (MI355X-first, not a port of the per-track python loop):
//   * one lane = one track; a launch advances every live track by up to S
//     steps with its state in registers, then the wave compacts its surviving
//     lanes into the next launch's index list with one ballot + popcount prefix
//     and a single atomic per wave.  Track lengths are heavy-tailed (402..26k
//     steps at 500x600), so later launches run on densely packed waves.
//   * the per-step uniform is counter-based (rocRAND Philox4x32-10 engine),
//     a pure function of (seed, global track id, step): any sharding of tracks
//     over launches / GPUs gives bit-identical trajectories.
//   * four data paths: 3x3 window gathers of the f64 updraft + f32 potential
//     rasters (reference-shaped, k_step_tracks<MODE_FLUIDFLOW / MODE_UPDRAFT>); a
//     precomputed per-cell table of 8 f64 weights (any movement model,
//     k_step_tracks<MODE_TABLE>); for the reference's default model (direction
//     memory 1, nu = 1) the THRESHOLD table -- per cell and last move the two 16-bit
//     decision thresholds, ONE 4-byte gather per step, k_transition_thr + k_step_thr
//     (the default; candidate table in LDS and early gather for fronts, reversal rows
//     in the fast path and a histogram window per block in LDS for batches that roam
//     basins, k_wander_windows / k_deal_sorted) -- and round 1's f32 ring table
//     (k_step_lean, one 12-byte gather, three-tier guarded decision).  Every fast
//     decision hands its near-ties to the exact sequence.
//   * tracks live in one list per XCD (column bands stay in one L2); the histogram is
//     a visit buffer + LDS binning kernel while the batch moves as a front (a row /
//     column window per step for axis-aligned headings, k_bin_visits; tile buckets per
//     launch for oblique ones, k_tile_sort + k_bin_bucket), and wave-private histogram
//     copies once it has scattered (k_fold_copies).
//   * the move decision reproduces the reference's f64 operation order exactly
//     (pairwise-8 sums, sequential cumsum, f32 potential differences); built
//     with -ffp-contract=off so no multiply-add is fused.
#include <rocrand/rocrand_philox4x32_10.h>
#include <hipcub/hipcub.hpp>

#include <cmath>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

#include "common.h"

namespace ssrs {

// ------------------------------------------------------------------ constants
// neighbour k -> (dr, dc) = (k / 3 - 1, k % 3 - 1)            movmodel.py:131-141
__host__ __device__ constexpr int dr_of(int k) { return k / 3 - 1; }
__host__ __device__ constexpr int dc_of(int k) { return k % 3 - 1; }
// f32(1/sqrt(2)) exactly as stored in neighbour_delta_norms_inv
#define SSRS_NINV_DIAG 0.70710677f

// movmodel.py:185-202: cells within +-45 deg of the previous move d; the
// initial direction (0,0) allows everything; the centre is never allowed.
constexpr uint32_t restriction(int d)
{
    const int dr = dr_of(d), dc = dc_of(d);
    uint32_t m = 0;
    for (int k = 0; k < 9; ++k) {
        bool ok = false;
        if (dr == 0 && dc == 0) ok = true;
        else if (dr != 0 && dc != 0)
            ok = (dr_of(k) == dr || dr_of(k) == 0) && (dc_of(k) == dc || dc_of(k) == 0);
        else if (dr == 0) ok = dc_of(k) == dc;
        else ok = dr_of(k) == dr;
        if (ok && k != 4) m |= 1u << k;
    }
    return m;
}
constexpr uint64_t pack_restrictions(int first, int count)
{
    uint64_t v = 0;
    for (int i = 0; i < count; ++i) v |= static_cast<uint64_t>(restriction(first + i)) << (9 * i);
    return v;
}
constexpr uint64_t kRestrictLo = pack_restrictions(0, 7);   // d = 0..6, 63 bits
constexpr uint64_t kRestrictHi = pack_restrictions(7, 2);   // d = 7, 8
constexpr uint32_t kAllButCentre = restriction(4);
static_assert(restriction(0) == 0b000001011u, "(-1,-1) -> k in {0,1,3}");
static_assert(restriction(7) == 0b111000000u, "(1,0) -> k in {6,7,8}");
static_assert(kAllButCentre == 0b111101111u, "(0,0) -> all but centre");

__device__ __forceinline__ uint32_t restriction_of(uint32_t d)
{
    const uint64_t v = d < 7 ? (kRestrictLo >> (9 * d)) : (kRestrictHi >> (9 * (d - 7)));
    return static_cast<uint32_t>(v) & 0x1FFu;
}

// With memory_parameter == 1 (the reference default) every step after the first
// has exactly THREE admissible cells.  candidates(d) = their positions in the
// 8-entry table row (k with the centre removed), ascending, 3 bits each.
constexpr uint32_t candidates(int d)
{
    uint32_t v = 0;
    int n = 0;
    for (int k = 0; k < 9 && n < 3; ++k)
        if ((restriction(d) >> k) & 1u) v |= static_cast<uint32_t>(k < 4 ? k : k - 1) << (3 * n++);
    return v;
}
constexpr uint64_t pack_candidates(int first, int count)
{
    uint64_t v = 0;
    for (int i = 0; i < count; ++i) v |= static_cast<uint64_t>(candidates(first + i)) << (9 * i);
    return v;
}
constexpr uint64_t kCandLo = pack_candidates(0, 7), kCandHi = pack_candidates(7, 2);
static_assert(candidates(7) == (5u | (6u << 3) | (7u << 6)), "(1,0) -> table slots 5,6,7");
static_assert(candidates(5) == (2u | (4u << 3) | (7u << 6)), "(0,1) -> k = 2,5,8 = slots 2,4,7");

__device__ __forceinline__ uint32_t candidates_of(uint32_t d)
{
    const uint64_t v = d < 7 ? (kCandLo >> (9 * d)) : (kCandHi >> (9 * (d - 7)));
    return static_cast<uint32_t>(v) & 0x1FFu;
}

// Ring order of the eight neighbours (clockwise from north): with it the three
// cells admissible after a move are CONSECUTIVE, so a table stored in ring order
// (two entries repeated at the end) yields them with one 12-byte load.
constexpr int kRingK[8] = {7, 8, 5, 2, 1, 0, 3, 6};           // ring position -> k
constexpr int ring_of_k(int k)
{
    for (int c = 0; c < 8; ++c)
        if (kRingK[c] == k) return c;
    return 8;                                                 // centre: "no move yet"
}
constexpr bool ring_matches_restrictions()
{
    for (int c = 0; c < 8; ++c) {
        const uint32_t m = (1u << kRingK[(c + 7) % 8]) | (1u << kRingK[c]) | (1u << kRingK[(c + 1) % 8]);
        if (m != restriction(kRingK[c])) return false;
    }
    return true;
}
static_assert(ring_matches_restrictions(), "restriction(d) = ring neighbours of d");
// order[c]: for the triple (ring c-1, c, c+1) as loaded (x0, x1, x2), which x is
// the 1st / 2nd / 3rd in ascending k (the order np.cumsum runs in); 2 bits each
constexpr uint32_t ring_order(int c)
{
    const int ks[3] = {kRingK[(c + 7) % 8], kRingK[c], kRingK[(c + 1) % 8]};
    uint32_t v = 0;
    int n = 0;
    for (int k = 0; k < 9; ++k)
        for (int j = 0; j < 3; ++j)
            if (ks[j] == k) v |= static_cast<uint32_t>(j) << (2 * n++);
    return v;
}
constexpr uint64_t pack_ring_orders()
{
    uint64_t v = 0;
    for (int c = 0; c < 8; ++c) v |= static_cast<uint64_t>(ring_order(c)) << (6 * c);
    return v;
}
constexpr uint64_t kRingOrder = pack_ring_orders();
constexpr uint32_t kOrdId = 0u | (1u << 2) | (2u << 4), kOrdRot = 2u | (0u << 2) | (1u << 4),
                   kOrdRev = 2u | (1u << 2) | (0u << 4), kOrdSwp = 1u | (0u << 2) | (2u << 4);
static_assert(ring_order(0) == kOrdId && ring_order(6) == kOrdId && ring_order(7) == kOrdId, "x0 x1 x2");
static_assert(ring_order(1) == kOrdRot, "after NE: x2 x0 x1");
static_assert(ring_order(2) == kOrdRev && ring_order(3) == kOrdRev && ring_order(4) == kOrdRev, "x2 x1 x0");
static_assert(ring_order(5) == kOrdSwp, "after SW: x1 x0 x2");
static_assert(ring_order(0) == (0u | (1u << 2) | (2u << 4)), "after N: NW, N, NE are k = 6, 7, 8");
static_assert(ring_order(2) == (2u | (1u << 2) | (0u << 4)), "after E: NE, E, SE are k = 8, 5, 2");
constexpr uint32_t pack_ring_deltas(bool rows)
{
    uint32_t v = 1u << 16;                                    // ring 8 = the centre: no displacement
    for (int c = 0; c < 8; ++c)
        v |= static_cast<uint32_t>((rows ? dr_of(kRingK[c]) : dc_of(kRingK[c])) + 1) << (2 * c);
    return v;
}
constexpr uint32_t kRingDr = pack_ring_deltas(true), kRingDc = pack_ring_deltas(false);
// k_step_roam: the ring position a move leads to, by which of the three intervals the uniform fell into (sel = 0, 1, 2
// in ascending k) and the last move's ring position: 3 bits at 4 x rc; the displacements as signed 2-bit fields at
// 4 x ring position (one shift amount serves both)
constexpr uint32_t pack_ring_next(int sel)
{
    uint32_t v = 0;
    for (int c = 0; c < 8; ++c) v |= static_cast<uint32_t>((c + 7 + ((ring_order(c) >> (2 * sel)) & 3u)) & 7u) << (4 * c);
    return v;
}
constexpr uint32_t kRingNext0 = pack_ring_next(0), kRingNext1 = pack_ring_next(1), kRingNext2 = pack_ring_next(2);
constexpr uint32_t pack_ring_deltas_signed(bool rows)
{
    uint32_t v = 0;
    for (int c = 0; c < 8; ++c)
        v |= (static_cast<uint32_t>(rows ? dr_of(kRingK[c]) : dc_of(kRingK[c])) & 3u) << (4 * c);
    return v;
}
constexpr uint32_t kRingDrS = pack_ring_deltas_signed(true), kRingDcS = pack_ring_deltas_signed(false);
constexpr uint64_t pack_ring_of_k()
{
    uint64_t v = 0;
    for (int k = 0; k < 9; ++k) v |= static_cast<uint64_t>(ring_of_k(k)) << (4 * k);
    return v;
}
constexpr uint64_t kRingOfK = pack_ring_of_k();
constexpr uint64_t pack_k_of_ring()
{
    uint64_t v = 4ull << 32;                                  // ring 8 -> k = 4
    for (int c = 0; c < 8; ++c) v |= static_cast<uint64_t>(kRingK[c]) << (4 * c);
    return v;
}
constexpr uint64_t kKOfRing = pack_k_of_ring();
constexpr int kRingFloats = 10;                               // ring 7, 0, 1, ..., 7, 0 per cell
// ring table buffer = records, 8 bytes of slack for the last 12-byte load, zero-mask bytes
__host__ __device__ constexpr size_t ring_mask_offset(int rows, int cols)
{
    return static_cast<size_t>(rows) * static_cast<size_t>(cols) * kRingFloats * sizeof(float) + 8;
}

// ------------------------------------------------------------------- uniform
// rocRAND Philox4x32-10: key = seed, counter = (blk, track); one 4-word block
// serves two steps.  The engine is built and dropped in registers (stateless).
// SSRS_PROBE_* are timing probes of tools/probe_chain.py (built into libssrs_probe_*.so by
// csrc/build.py --probe; results are wrong on purpose and the product build never defines them):
// NO_PHILOX replaces the generator by two multiplies, NO_GATHER the table gather by constants.
__device__ __forceinline__ uint4 philox_block(uint64_t seed, uint64_t track, uint64_t blk)
{
#ifdef SSRS_PROBE_NO_PHILOX
    const uint32_t h = static_cast<uint32_t>(blk) * 0x9E3779B9u ^ static_cast<uint32_t>(track) * 0x85EBCA6Bu ^
                       static_cast<uint32_t>(seed);
    return make_uint4(h * 0xC2B2AE35u, h ^ 0x27D4EB2Fu, (h >> 3) * 0x165667B1u, h + 0x9E3779B9u);
#endif
    rocrand_state_philox4x32_10 st;
    rocrand_init(seed, track, blk * 4ull, &st);
    return rocrand4(&st);
}

// The same block with the ten rounds written out for gfx950: a round's two three-input xors (product high word ^
// counter word ^ round key) are ONE v_bitop3_b32 each (truth table 0x96) where the compiler makes two v_xor_b32 of
// rocRAND's source -- 17 of the 155 vector instructions of a pair of moves in k_step_roam.  Counter and key as
// rocrand_init(seed, subsequence = track, offset = 4 blk) sets them; k_uniform_selftest compares the two word for word.
__device__ __forceinline__ uint4 philox_block_b3(uint64_t seed, uint64_t track, uint64_t blk)
{
    uint32_t c0 = static_cast<uint32_t>(blk), c1 = static_cast<uint32_t>(blk >> 32);
    uint32_t c2 = static_cast<uint32_t>(track), c3 = static_cast<uint32_t>(track >> 32);
    uint32_t k0 = static_cast<uint32_t>(seed), k1 = static_cast<uint32_t>(seed >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = static_cast<uint64_t>(0xD2511F53u) * c0, p1 = static_cast<uint64_t>(0xCD9E8D57u) * c2;
        const uint32_t n0 = __builtin_amdgcn_bitop3_b32(static_cast<uint32_t>(p1 >> 32), c1, k0, 0x96);
        const uint32_t n2 = __builtin_amdgcn_bitop3_b32(static_cast<uint32_t>(p0 >> 32), c3, k1, 0x96);
        c1 = static_cast<uint32_t>(p1);
        c3 = static_cast<uint32_t>(p0);
        c0 = n0;
        c2 = n2;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    return make_uint4(c0, c1, c2, c3);
}

__device__ __forceinline__ double words_to_uniform(uint32_t a, uint32_t b)
{   // numpy legacy random_sample: 53 bits, [0, 1)
    return (static_cast<double>(a >> 5) * 67108864.0 + static_cast<double>(b >> 6)) *
           (1.0 / 9007199254740992.0);
}

__global__ void k_uniform_selftest(uint64_t seed, const uint64_t *track, const uint64_t *step,
                                   double *out, size_t n)
{
    const size_t i = blockIdx.x * static_cast<size_t>(blockDim.x) + threadIdx.x;
    if (i >= n) return;
    const uint4 w = philox_block(seed, track[i], step[i] >> 1);
    const uint4 v = philox_block_b3(seed, track[i], step[i] >> 1);
    const bool same = w.x == v.x && w.y == v.y && w.z == v.z && w.w == v.w;          // rocRAND's engine and the written-out rounds
    const double u = (step[i] & 1) ? words_to_uniform(w.z, w.w) : words_to_uniform(w.x, w.y);
    out[i] = same ? u : __longlong_as_double(0x7FF8000000000000ll);
}

// ------------------------------------------------------------ move decision
__host__ __device__ __forceinline__ double sum9(const double *x)
{   // numpy pairwise summation for n = 9
    return (((x[0] + x[1]) + (x[2] + x[3])) + ((x[4] + x[5]) + (x[6] + x[7]))) + x[8];
}

// generate_move_probabilities (movmodel.py:220-244) followed by
// np.random.choice's inverse-cdf pick.  w[9] raw weights (w[4] ignored unless NaN).
//
// The reference normalises twice and divides the running sums by their last
// element before comparing with u: 26 f64 divisions per step that only matter
// when u lands within rounding distance of a cdf boundary.  `fast` evaluates the
// same comparison on the UNNORMALISED running sums C_k against u * C_8 and
// falls back to the exact sequence whenever any |C_k - u C_8| is within
// 2^-46 relative: the exact path's cdf_k equals C_k / C_8 up to < 2^-48 (all
// terms non-negative, <= 22 roundings of 2^-53 each), and C_k, u C_8 carry
// <= 9 roundings, so outside that band both paths take the same decision.  The
// result is therefore always the reference's, bit for bit.
__device__ __forceinline__ int choose_move(const double *w, const double *prior, double nu,
                                           uint32_t mask, double u, bool fast)
{
    double q[9];
    bool has_nan = false;
#pragma unroll
    for (int k = 0; k < 9; ++k) has_nan |= (w[k] != w[k]);
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        const double v = has_nan ? prior[k] : w[k];
        q[k] = v > 0.0 ? v : 0.0;                                  // clip(min=0)
    }
    q[4] = 0.0;
    bool any = false;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        const double z = q[k] * 0.0;                               // ix * float(iy)
        q[k] = ((mask >> k) & 1u) ? q[k] : z;
        any |= (q[k] != 0.0);
    }
    if (!any) {                                   // all masked weights are zero
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            q[k] = ((mask >> k) & 1u) && k != 4 ? prior[k] : 0.0;
            any |= (q[k] != 0.0);
        }
        if (!any) {                               // prior fully masked as well
#pragma unroll
            for (int k = 0; k < 9; ++k) q[k] = prior[k];
        }
    }
    if (fast) {
        double c[8];
        double acc = 0.0;
#pragma unroll
        for (int k = 0; k < 8; ++k) { acc = acc + q[k]; c[k] = acc; }
        const double ut = u * (acc + q[8]);
        const double band = ut * 0x1p-46;
        int idx = 0;
        bool near = false;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const double d = c[k] - ut;
            idx += d <= 0.0 ? 1 : 0;
            near |= !(fabs(d) > band);            // also true for NaN / inf
        }
        if (!near) return idx;
    }
    const double s1 = sum9(q);
#pragma unroll
    for (int k = 0; k < 9; ++k) q[k] = q[k] / s1;
    if (nu != 1.0) {
#pragma unroll
        for (int k = 0; k < 9; ++k) q[k] = pow(q[k], nu);
    }
    const double s2 = sum9(q);
    double acc = 0.0;
    double cdf[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        q[k] = q[k] / s2;
        acc = acc + q[k];
        cdf[k] = acc;
    }
    int idx = 0;
#pragma unroll
    for (int k = 0; k < 9; ++k) idx += (cdf[k] / cdf[8] <= u) ? 1 : 0;
    return idx;
}

// Table-mode decision on the 8 pre-clipped weights of one cell (entry j is
// neighbour k = j < 4 ? j : j + 1).  Same guarded comparison as choose_move's
// fast branch, minus everything the table builder already did (clip, NaN
// poisoning).  Returns -1 when the exact sequence must decide: a cdf boundary
// within 2^-46 of u, an all-zero / NaN (poisoned) / infinite masked row.
__device__ __forceinline__ int choose_table_fast(const double *t, uint32_t mask, double u)
{
    double c[8];
    double acc = 0.0;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int k = j < 4 ? j : j + 1;
        acc = acc + (((mask >> k) & 1u) ? t[j] : 0.0);
        c[j] = acc;
    }
    const double ut = u * acc;
    const double band = ut * 0x1p-46;
    int idx = 0;
    double closest = __builtin_inf();            // min_k |C_k - u C_8|
#pragma unroll
    for (int j = 0; j < 7; ++j) {               // k = 8 is never counted: cdf_8 = 1 > u
        const double d = c[j] - ut;
        const int le = d <= 0.0 ? 1 : 0;
        idx += (j == 3) ? 2 * le : le;          // cdf_4 == cdf_3 (centre weight is 0)
        closest = fmin(closest, fabs(d));
    }
    // NaN / inf rows: acc > 0 is false for NaN; inf gives band = inf >= closest
    const bool near = !(acc > 0.0) || !(closest > band);
    return near ? -1 : idx;
}

// The same decision when only three cells are admissible (their table values
// wa, wb, wc in ascending k): masked entries add 0.0 to the running sum, so the
// three partial sums ARE the masked row's cumulative sums, bit for bit.
// Returns the position (0..2) of the chosen candidate or -1 (see above).
__device__ __forceinline__ int choose_three_fast(double wa, double wb, double wc, double u)
{
    const double ca = wa + 0.0;                  // -0.0 -> +0.0 like the 0.0-seeded sum
    const double cb = ca + wb;
    const double acc = cb + wc;
    const double ut = u * acc;
    const double band = ut * 0x1p-46;
    const double da = ca - ut, db = cb - ut;
    const int sel = da > 0.0 ? 0 : (db > 0.0 ? 1 : 2);
    const double closest = fmin(fmin(fabs(da), fabs(db)), fmin(ut, fabs(acc - ut)));
    const bool near = !(acc > 0.0) || !(closest > band);
    return near ? -1 : sel;
}

// Ring-table form: the three weights are the f64 table values rounded to f32
// (relative error <= 2^-24 each, or < 2^-126 absolute when tiny), so the running
// sums and u * total are within 2^-23 * total of the f64 quantities the exact
// sequence compares: outside a band of 2^-21 * total both decide alike.  Totals
// that are tiny, infinite (f32 overflow) or NaN (poisoned row) go to the exact path.
__device__ __forceinline__ int choose_three_ring(float fa, float fb, float fc, double u)
{
    const double ca = static_cast<double>(fa);
    const double cb = ca + static_cast<double>(fb);
    const double acc = cb + static_cast<double>(fc);
    const double ut = u * acc;
    const double band = acc * 0x1p-21;
    const double da = ca - ut, db = cb - ut;
    const int sel = da > 0.0 ? 0 : (db > 0.0 ? 1 : 2);
    const double closest = fmin(fabs(da), fabs(db));
    const bool near = !(acc > 0x1p-90) || !(closest > band);
    return near ? -1 : sel;
}

// First tier of the ring decision, entirely in f32 (the f64 / conversion
// instructions were the larger part of a step's VALU time): uf = u truncated to its
// top 24 bits.  Error budget relative to the true total A: table values 2^-24 each,
// two f32 additions 2^-24 each, u truncation and product rounding 2^-24 each, the
// subtractions 2^-24: |(C_j - u A) - (c_j - uf acc)| <= 2^-21 A.  Band 2^-19 acc; a
// candidate boundary inside the band (probability ~8e-6 per step) goes to
// choose_three_ring (f64 sums, full u, band 2^-21), which may go to the exact sequence.
__device__ __forceinline__ int choose_three_ring_f32(float fa, float fb, float fc, float uf)
{
    const float cb = fa + fb;
    const float acc = cb + fc;
    const float ut = uf * acc;
    const float band = acc * 0x1p-19f;
    const float da = fa - ut, db = cb - ut;
    const int sel = da > 0.f ? 0 : (db > 0.f ? 1 : 2);
    const float closest = fminf(fabsf(da), fabsf(db));
    const bool near = !(acc > 0x1p-90f) || !(closest > band);
    return near ? -1 : sel;
}

// Raw 3x3 move weights of movmodel.py:292-306 at an interior cell.
template <bool HAS_POT>
__device__ __forceinline__ void window_weights(const double *__restrict__ updraft,
                                               const float *__restrict__ potential,
                                               int cols, int row, int col, double *w)
{
    const size_t centre = static_cast<size_t>(row) * cols + col;
    double win[9];
#pragma unroll
    for (int j = 0; j < 9; ++j) {
        const double v = updraft[centre + dr_of(j) * cols + dc_of(j)];
        win[j] = v != v ? v : (v > 1e-06 ? v : 1e-06);            // clip(min=1e-06)
    }
    const double ic = 1.0 / win[4];
#pragma unroll
    for (int j = 0; j < 9; ++j) w[j] = 2.0 / (ic + 1.0 / win[j]);  // harmonic mean
    if (HAS_POT) {
        const float pc = potential[centre];
#pragma unroll
        for (int j = 0; j < 9; ++j) {
            const float d = pc - potential[centre + dr_of(j) * cols + dc_of(j)];
            const float ninv = (j == 4) ? 0.f : ((j & 1) ? 1.f : SSRS_NINV_DIAG);
            const float e = d * ninv;                              // stays f32
            w[j] = w[j] * static_cast<double>(e);
        }
    }
}

// The exact decision at one tenth of the price (round 3).  A near-tie of the 16-bit thresholds sends its lane --
// and with it its wave -- through window_weights + choose_move: 18 + 26 f64 divisions, ~20 000 clocks for a wave
// alone on its SIMD, every ~250 wave-steps.  With direction memory 1 only THREE cells are admissible, and when all
// 18 inputs of the window are finite and moderate (every weight is then finite: no NaN rule, no inf x 0) the
// reference's sequence reduces to: the three weights in ascending k (same operations as window_weights), clipped
// at 0 -- their running sums ARE the masked row's cumulative sums bit for bit (the other cells add 0.0) -- decided
// by choose_three_fast's guarded comparison; all three zero: the masked prior's exact thresholds (k_ctl_init),
// as k_step_lean does.  7 divisions.  Returns the neighbour index k, or -1: the full sequence decides (inputs not
// finite or huge, a comparison within 2^-46, the prior picking the centre).
template <bool HAS_POT>
__device__ __forceinline__ int exact_three(const double *__restrict__ updraft, const float *__restrict__ potential, int cols,
                                           int row, int col, uint32_t last_k, const double *__restrict__ thr, double u)
{
    const size_t centre = static_cast<size_t>(row) * cols + col;
    const uint32_t mask = restriction_of(last_k);
    double v[9];
    float p[9];
    bool fine = true;
#pragma unroll
    for (int j = 0; j < 9; ++j) {
        v[j] = updraft[centre + dr_of(j) * cols + dc_of(j)];
        fine &= (v[j] <= 1e150) & (v[j] >= -1e150);                  // (false for NaN)
        if (HAS_POT) {
            p[j] = potential[centre + dr_of(j) * cols + dc_of(j)];
            fine &= fabsf(p[j]) <= 1e38f;
        }
    }
    if (!fine) return -1;
    const double ic = 1.0 / (v[4] > 1e-06 ? v[4] : 1e-06);
    double w3[3] = {0.0, 0.0, 0.0};
    int k3[3] = {0, 0, 0};
    int n = 0;
#pragma unroll
    for (int j = 0; j < 9; ++j) {
        if (j == 4) continue;
        const bool adm = (mask >> j) & 1u;
        // (uniform control flow: the division is done for every j of the lane's mask only through the select below)
        if (adm) {
            double w = 2.0 / (ic + 1.0 / (v[j] > 1e-06 ? v[j] : 1e-06));      // harmonic mean
            if (HAS_POT) {
                const float d = p[4] - p[j];
                const float ninv = (j & 1) ? 1.f : SSRS_NINV_DIAG;
                const float e = d * ninv;                                      // stays f32
                w = w * static_cast<double>(e);
            }
            w = w > 0.0 ? w : 0.0;                                             // clip(min=0)
            if (n == 0) { w3[0] = w; k3[0] = j; } else if (n == 1) { w3[1] = w; k3[1] = j; } else { w3[2] = w; k3[2] = j; }
            ++n;
        }
    }
    if (n != 3) return -1;
    if (w3[0] == 0.0 && w3[1] == 0.0 && w3[2] == 0.0) {
        // every admissible weight is exactly zero: the directional prior decides (movmodel.py:234-240)
        const double *t = thr + 9u * last_k;
        int idx = 0;
#pragma unroll
        for (int k = 0; k < 9; ++k) idx += t[k] <= u ? 1 : 0;
        return (idx == 4 || idx > 8) ? -1 : idx;
    }
    const int sel = choose_three_fast(w3[0], w3[1], w3[2], u);
    return sel < 0 ? -1 : (sel == 0 ? k3[0] : (sel == 1 ? k3[1] : k3[2]));
}

// ------------------------------------------------------------ transition table
// One thread per cell: 9+9 cached reads, one 64-B row of 8 f64 written.  Blocks
// own 64 x 16 cell tiles walked in an XCD-aware order (blocks b and b+8 share an
// XCD, so each XCD gets one contiguous band of tiles): the rows above and below
// a tile are then found in the same XCD's L2 instead of being re-fetched from
// HBM by another XCD (PMC before: 2.7x read over-fetch with row-strip blocks).
constexpr int kTabW = 64, kTabH = 16;
__device__ __forceinline__ int xcd_band(int b, int n)
{
    const int q = n / 8, r = n % 8, x = b % 8, j = b / 8;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + j;
}

// RING: the same weights rounded to f32 and stored in ring order, 10 floats per
// cell (ring 7, 0, 1, ..., 7, 0), for the three-candidate stepper (k_step_lean).
template <bool RING>
__global__ __launch_bounds__(kBlock) void k_transition_table(
    const double *__restrict__ updraft, const float *__restrict__ potential,
    void *__restrict__ table_out, int rows, int cols, int tiles_x, int ntiles)
{
    // LDS tiles (+1-cell halo) of the clipped updraft reciprocals and of the
    // potential: every cell's 1/max(u, 1e-6) is needed by its 9 neighbours, so it
    // is computed once here instead of 9 times (the harmonic mean of
    // movmodel.py:260 is 2 / (1/u_c + 1/u_k): identical operands, identical bits).
    constexpr int LW = kTabW + 2, LH = kTabH + 2;
    __shared__ double s_inv[LW * LH];
    __shared__ float s_pot[LW * LH];
    const int t = xcd_band(blockIdx.x, ntiles);
    const int r0 = (t / tiles_x) * kTabH, c0 = (t % tiles_x) * kTabW;
    for (int i = threadIdx.x; i < LW * LH; i += kBlock) {
        const int lr = i / LW, lc = i - lr * LW;
        int gr = r0 - 1 + lr, gc = c0 - 1 + lc;
        gr = gr < 0 ? 0 : (gr >= rows ? rows - 1 : gr);   // clamped cells feed border
        gc = gc < 0 ? 0 : (gc >= cols ? cols - 1 : gc);   // outputs only (zeroed below)
        const size_t g = static_cast<size_t>(gr) * cols + gc;
        const double v = updraft[g];
        const double w = v != v ? v : (v > 1e-06 ? v : 1e-06);   // clip(min=1e-06), NaN kept
        s_inv[i] = 1.0 / w;
        s_pot[i] = potential ? potential[g] : 0.f;
    }
    __syncthreads();
    const int lc = static_cast<int>(threadIdx.x % kTabW) + 1;
    const int col = c0 + lc - 1;
    if (col >= cols) return;
    for (int lr = static_cast<int>(threadIdx.x / kTabW) + 1; lr <= kTabH; lr += kBlock / kTabW) {
        const int row = r0 + lr - 1;
        if (row >= rows) break;
        const size_t i = static_cast<size_t>(row) * cols + col;
        double w[9];
        const bool interior = row > 0 && col > 0 && row < rows - 1 && col < cols - 1;
        if (interior) {
            const double ic = s_inv[lr * LW + lc];
            const float pc = s_pot[lr * LW + lc];
#pragma unroll
            for (int j = 0; j < 9; ++j) {
                const int o = (lr + dr_of(j)) * LW + lc + dc_of(j);
                w[j] = 2.0 / (ic + s_inv[o]);                       // harmonic mean
                if (potential) {
                    const float d = pc - s_pot[o];
                    const float ninv = (j == 4) ? 0.f : ((j & 1) ? 1.f : SSRS_NINV_DIAG);
                    const float e = d * ninv;                       // stays f32
                    w[j] = w[j] * static_cast<double>(e);
                }
            }
        } else {
#pragma unroll
            for (int j = 0; j < 9; ++j) w[j] = 0.0;
        }
        bool has_nan = false;
#pragma unroll
        for (int j = 0; j < 9; ++j) has_nan |= (w[j] != w[j]);
        if (RING) {
            // a weight that is EXACTLY zero after the clip is stored as -0.0f (the sign bit is
            // free: weights are >= 0), so the stepper can tell "all three admissible weights
            // are zero" (-> the reference's fallback to the directional prior, common in
            // real potential fields) from positive weights below the f32 range (+0.0f).
            // Rows with an infinite weight are poisoned like NaN rows: the prior shortcut
            // relies on masked-out weights times 0.0 being 0.0 (movmodel.py:231).
            bool bad = has_nan;
#pragma unroll
            for (int j = 0; j < 9; ++j) bad |= (w[j] - w[j] != 0.0);     // inf or NaN
            float f[kRingFloats];
#pragma unroll
            for (int j = 0; j < kRingFloats; ++j) {
                const double v = w[kRingK[(j + 7) % 8]];
                f[j] = bad ? __builtin_nanf("") : (v > 0.0 ? static_cast<float>(v) : -0.0f);
            }
            float2 *dst = reinterpret_cast<float2 *>(static_cast<float *>(table_out) + i * kRingFloats);
#pragma unroll
            for (int k = 0; k < kRingFloats / 2; ++k) dst[k] = make_float2(f[2 * k], f[2 * k + 1]);
            // zero mask (one byte per cell behind the records): bit c = ring weight c is exactly
            // zero.  Tracks that wander through dead terrain read only this byte per step.
            uint32_t zm = 0;
#pragma unroll
            for (int c = 0; c < 8; ++c) zm |= (!bad && !(w[kRingK[c]] > 0.0)) ? (1u << c) : 0u;
            (static_cast<uint8_t *>(table_out) + ring_mask_offset(rows, cols))[i] = static_cast<uint8_t>(zm);
        } else {
            double2 *dst = reinterpret_cast<double2 *>(static_cast<double *>(table_out) + i * 8);
            double o[8];
#pragma unroll
            for (int j = 0, k = 0; j < 9; ++j) {
                if (j == 4) continue;
                o[k++] = has_nan ? __builtin_nan("") : (w[j] > 0.0 ? w[j] : 0.0);
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) dst[k] = make_double2(o[2 * k], o[2 * k + 1]);
        }
    }
}

// -------------------------------------------------------------------- stepper
struct alignas(16) TrackState {
    int32_t pos;     // row | col << 16   (both < 32768)
    int32_t k;       // moves taken; < 0 = dead (bad start cell)
    uint32_t dirs;   // last 8 move indices, 4 bits each, newest in bits 0-3
    uint32_t aux;    // bits 0-8: AND of restrictions over the whole history
                     // (memory == 0); bits 9-31: release step (coherent schedule)
};

// Live tracks are kept in kXcd separate lists, one per XCD: blocks are dealt to
// the XCDs round-robin (block b runs on XCD b % 8), so block b serves list b % 8
// and a track stays on the XCD it was dealt to.  With the coherent schedule the
// lists are contiguous bands of the across-track coordinate: the table rows a
// band walks over are fetched into ONE XCD's L2 instead of all eight.
constexpr int kXcd = 8;
struct alignas(16) TrackCtl {
    uint32_t count[4][kXcd];     // live tracks of list x entering launch i at count[i & 3][x]
    uint32_t error;              // != 0: some start cell was outside the raster
    uint32_t par_min;            // smallest along-track start coordinate (schedule)
    unsigned long long steps;    // total moves taken
    unsigned long long strays;   // visits the binning kernel could not place in its LDS window
    uint32_t bin_done;           // blocks of the running k_bin_visits16 that have finished (read-back by the last)
    uint32_t pad;
    double prior[9];             // directional prior of this call (read by the slow paths)
    unsigned long long roam_slow;   // k_step_roam: wave-pairs in which some lane took the slow path
    unsigned long long roam_pairs;  // k_step_roam: wave-pairs run
    unsigned long long dbg_tsum, dbg_tmax, dbg_waves, dbg_slowmax;   // SSRS_TRACKS_DEBUG_ROAM: wave lifetimes of one launch
    uint32_t roam_stop;                                              // k_step_roam: launch + 1 of the launch whose first wave is through its steps
    uint32_t deal_live;                                              // k_wander_windows: live tracks it found (the host picks the deal's block width from it)
    unsigned long long dbg_waits;                                    // same: polls of stepping waves that waited for a staged row
    unsigned long long dbg_span, dbg_span_max;                       // same: virtual-row span of the blocks of a front (sum << 20 | blocks; max)
};

static_assert(sizeof(TrackCtl) <= 128 * sizeof(uint32_t), "final read-back slot of pinned_counts()");

// block histogram windows of wandering batches (k_step_thr<6>, k_wander_windows)
// 144 rows: 144 KB of LDS, one block per CU (rounds 2-3); 72 rows: 72 KB, two blocks per CU (-DSSRS_WIN_ROWS=72, A/B)
#ifndef SSRS_WIN_ROWS
#define SSRS_WIN_ROWS 144
#endif
constexpr int kWinRows = SSRS_WIN_ROWS, kWinCols = 256;
static_assert(kWinRows == 144 || kWinRows == 72, "a window is 4 or 2 rows of coarse bins");
constexpr int kWanderWindows = 16;
constexpr uint32_t kWanderMix = 256;                 // low bits of a sort key: a per-sort hash of the track (k_wander_keys)
constexpr uint32_t kDealBlocks = kWinRows == 144 ? 232 : 464;   // blocks the contiguous deal spreads the live tracks over (+ one per
                                                   // window in use and the padding: under the 256 CUs x blocks per CU)
constexpr int kBinRows = 36, kBinCols = kWinCols / 4;
constexpr int kWinBinRows = kWinRows / kBinRows;
constexpr int kWanderBins = 16384;           // 64 KB of LDS
struct WanderWindows {
    int32_t n;
    int32_t r0[kWanderWindows], c0[kWanderWindows];
};



// Coherent schedule.  Tracks are independent, so the order in which lanes pick
// them up and the global step at which each one starts are free choices that
// cannot change any result (the uniform is keyed by track id and the track's
// own step count).  Both are chosen for locality: tracks are sorted by their
// across-track coordinate (lanes of a wave walk neighbouring columns) and a
// track that starts d cells ahead of the rearmost one along the movement
// direction is released d steps later, so the whole batch sweeps the raster as
// one front a few rows deep: table rows and histogram rows are then shared
// through L2 instead of being re-fetched from HBM by every track.
struct PlanGeom {
    double cos_t, sin_t;   // movement direction (north = +row, clockwise)
    int offset;            // rows + cols: makes both coordinates non-negative
};
__device__ __forceinline__ void plan_coords(const PlanGeom &g, int row, int col, int &par, int &perp)
{
    par = static_cast<int>(lrint(row * g.cos_t + col * g.sin_t)) + g.offset;
    perp = static_cast<int>(lrint(col * g.cos_t - row * g.sin_t)) + g.offset;
}

// sort key = perp << par_bits | par: both coordinates lie in [0, 2 offset], so rasters up to
// ~16k x 16k sort on 32-bit keys (half the passes' traffic, one pass fewer than 40 bits)
template <typename KeyT>
__global__ __launch_bounds__(kBlock) void k_plan_keys(const int32_t *__restrict__ start_rc,
                                                     long long ntracks, PlanGeom g,
                                                     KeyT *__restrict__ keys,
                                                     int32_t *__restrict__ vals, TrackCtl *ctl, int par_bits)
{
    const long long t = blockIdx.x * static_cast<long long>(kBlock) + threadIdx.x;
    int par = 0x7fffffff;
    if (t < ntracks) {
        int perp;
        plan_coords(g, start_rc[2 * t], start_rc[2 * t + 1], par, perp);
        par = par < 0 ? 0 : par;
        perp = perp < 0 ? 0 : perp;
        const int top = (1 << par_bits) - 1;
        keys[t] = (static_cast<KeyT>(perp > top ? top : perp) << par_bits) | static_cast<KeyT>(par > top ? top : par);
        vals[t] = static_cast<int32_t>(t);
    }
    // block minimum, one atomic per block (one per wave: 1500 atomics on one word took 15 us)
    __shared__ int s_min;
    if (threadIdx.x == 0) s_min = 0x7fffffff;
    __syncthreads();
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const int o = __shfl_down(par, off);
        par = o < par ? o : par;
    }
    if ((threadIdx.x & 63) == 0 && par != 0x7fffffff) atomicMin(&s_min, par);
    __syncthreads();
    if (threadIdx.x == 0 && s_min != 0x7fffffff) atomicMin(&ctl->par_min, static_cast<uint32_t>(s_min));
}

__device__ __forceinline__ int wander_window_of(const WanderWindows *__restrict__ w, int n, int row, int col)
{
    int id = n;
    for (int q = n - 1; q >= 0; --q)
        if (static_cast<uint32_t>(row - w->r0[q]) < static_cast<uint32_t>(kWinRows) &&
            static_cast<uint32_t>(col - w->c0[q]) < static_cast<uint32_t>(kWinCols)) id = q;
    return id;
}

enum { MODE_PRIOR = 0, MODE_UPDRAFT = 1, MODE_FLUIDFLOW = 2, MODE_TABLE = 3 };

struct RoamEntry;

struct StepArgs {
    int rows, cols, burnin, memory;
    long long max_k;
    double nu;
    const double *prior;         // 9 doubles in device memory (kept out of the SGPR budget)
    const double *updraft;
    const float *potential;
    const double *table;
    unsigned long long seed, track_base;
    uint32_t *hist;
    int16_t *end_rc;
    int32_t *lengths;
    int16_t *traj;
    const long long *traj_off;
    TrackState *state;
    const int32_t *list_in;      // NULL = identity (first launch)
    int32_t *list_out;
    TrackCtl *ctl;
    int launch;                  // index of this launch
    int fast;                    // guarded division-free decision (see choose_move)
    int steps;                   // S
    int coherent;                // tracks carry a release step in aux
    uint32_t *visits;            // [steps][visit_stride] visited cell per slot (K3 binning), or NULL
    long long visit_stride;
    uint32_t cap;                // slots per XCD list (multiple of kBlock); list x = [x*cap, (x+1)*cap)
    long long it_base;           // k_step_thr: global iteration of this launch's first step
    int plane_shift;             // k_step_thr: log2 of the byte stride between the table's eight planes
    uint32_t guard;              // k_step_thr: bytes of guard band before plane 0 (table points at the band)
    int v16_offset;              // k_step_thr<4>: plan offset (rows + cols): first start row = ctl->par_min - v16_offset
    int pf_dir, pf_rc;           // k_step_thr prefetch wave: row direction of the front (+1 north, -1 south),
                                 // ring position of the heading
    uint32_t vcap;               // slots per XCD list in the visit buffer (= cap, or the launch's own
                                 // bound when its visits are recorded for trajectory output)
    const double *thr;           // [9][9] prior-fallback thresholds (k_ctl_init)
    const uint8_t *zmask;        // ring table's zero-mask bytes (scattered variant), or NULL
    uint32_t *hist_copies;       // privatised histogram copies (scattered batches), or NULL
    int ncopies;
    const WanderWindows *wander; // k_step_thr<6>: the windows of the last wander sort (n = 0: none yet)
    const RoamEntry *roam;       // k_step_roam: the pair table (8 entries per cell of the raster), or NULL
    const void *fine;            // k_step_roam: 32-bit boundaries per (cell, last move) for near-ties, or NULL
    int ordered;                 // k_step_tracks: block-ordered list reservation (see there)
    int roam_stop;               // k_step_roam: a launch ends when its FIRST wave is through its steps (A/B: SSRS_TRACKS_NO_ROAM_STOP)
    int lr_wait;                 // k_step_thr<.., LR>: waves ahead of the ring wait for their row (SSRS_TRACKS_LDS_ROWS=2)
    int cheap_exact;             // k_step_thr: near-ties through exact_three first (A/B: SSRS_TRACKS_NO_CHEAP_EXACT)
    unsigned long long *dbg_buf; // diagnostic builds (SSRS_DEBUG_WAVE_DUMP): one record per wave
    int debug_roam;              // k_step_roam: wave lifetimes into the control block (SSRS_TRACKS_DEBUG_ROAM)
    uint32_t vis_r, vis_c;       // visit key = row * vis_r + col * vis_c: (cols, 1), or (1, rows) when
                                 // the front is a column (east / west headings: transposed binning)
};

__global__ __launch_bounds__(kBlock) void k_tracks_init(
    const int32_t *__restrict__ start_rc, long long ntracks, int rows, int cols,
    uint32_t *hist, int16_t *traj, const long long *traj_off, int32_t *lengths,
    int16_t *end_rc, TrackState *state, TrackCtl *ctl, PlanGeom g, int coherent, uint32_t cap)
{
    const long long t = blockIdx.x * static_cast<long long>(kBlock) + threadIdx.x;
    if (t < kXcd) {
        // the first launch's list (sorted or identity) is cut into kXcd runs of cap slots
        const long long left = ntracks - t * static_cast<long long>(cap);
        ctl->count[0][t] = static_cast<uint32_t>(left < 0 ? 0 : (left > cap ? cap : left));
        ctl->count[1][t] = ctl->count[2][t] = ctl->count[3][t] = 0;
        if (t == 0) ctl->steps = 0;
    }
    if (t >= ntracks) return;
    const int row = start_rc[2 * t], col = start_rc[2 * t + 1];
    TrackState s;
    s.dirs = 0x44444444u;        // "no move yet" = (0,0) in every history slot
    uint32_t delay = 0;
    if (coherent) {
        int par, perp;
        plan_coords(g, row, col, par, perp);
        par = par < 0 ? 0 : par;
        const uint32_t pmin = ctl->par_min;
        delay = static_cast<uint32_t>(par) > pmin ? static_cast<uint32_t>(par) - pmin : 0u;
        if (delay > 0x7FFFFFu) delay = 0x7FFFFFu;
        // even delays keep the parity of k wave-uniform: one Philox block serves two
        // steps, and a wave with mixed parities would evaluate it on every iteration
        delay &= ~1u;
    }
    s.aux = kAllButCentre | (delay << 9);
    if (row < 0 || col < 0 || row >= rows || col >= cols) {
        atomicOr(&ctl->error, 1u);
        s.pos = 0;
        s.k = -1;
        if (lengths) lengths[t] = 0;
        if (end_rc) { end_rc[2 * t] = -1; end_rc[2 * t + 1] = -1; }
    } else {
        s.pos = row | (col << 16);
        s.k = 0;
        if (hist) atomicAdd(&hist[static_cast<size_t>(row) * cols + col], 1u);
        if (traj)
            reinterpret_cast<uint32_t *>(traj)[traj_off[t]] =
                static_cast<uint32_t>(row & 0xFFFF) | (static_cast<uint32_t>(col) << 16);
    }
    state[t] = s;
}

// The general stepper: any movement model, any data path, optional trajectory output.
// The reference's default configuration runs in k_step_lean instead.
template <int MODE>
__global__ __launch_bounds__(kBlock) void k_step_tracks(const StepArgs a)
{
    // a.ordered: one list reservation per block with its waves in order (the first-move launch of a front whose
    // later launches stage table rows in LDS: a block's tracks must stay neighbours)
    __shared__ uint32_t s_cnt[kBlock / 64 + 1];
    if (a.ordered) {
        if (threadIdx.x <= kBlock / 64) s_cnt[threadIdx.x] = 0u;
        __syncthreads();
    }
    TrackCtl *ctl = a.ctl;
    const int in_slot = a.launch & 3, out_slot = (a.launch + 1) & 3;
    const uint32_t xcd = blockIdx.x % kXcd;
    const uint32_t nlive = ctl->count[in_slot][xcd];
    const uint32_t il = (blockIdx.x / kXcd) * kBlock + threadIdx.x;   // position in list xcd
    const uint32_t i = xcd * a.cap + il;                               // slot in the list arrays
    const uint32_t iv = xcd * a.vcap + il;                             // slot in the visit buffer
    if (blockIdx.x == 0 && threadIdx.x < kXcd)
        ctl->count[(a.launch + 2) & 3][threadIdx.x] = 0;   // free slot of launch+1's output
    // whole waves past the live list leave at once (wave-uniform)
    if ((il & ~63u) >= nlive) return;

    bool active = il < nlive;
    const int32_t t = active ? (a.list_in ? a.list_in[i] : static_cast<int32_t>(i)) : 0;
    TrackState s = {0, -1, 0, 0};
    if (active) s = a.state[t];
    active = active && s.k >= 0;
    int row = s.pos & 0xFFFF, col = (s.pos >> 16) & 0xFFFF;
    int k = s.k;                                   // max_moves < 2^31 (checked by the host)
    const int max_k = static_cast<int>(a.max_k);
    uint32_t dirs = s.dirs, run = s.aux & 0x1FFu;
    // release step relative to this launch's first global step (both < 2^31)
    const long long rel64 = (a.coherent ? static_cast<long long>(s.aux >> 9) : 0) -
                            static_cast<long long>(a.launch) * a.steps;
    const int release = rel64 > 0x7fffffffLL ? 0x7fffffff : (rel64 < 0 ? 0 : static_cast<int>(rel64));
    const unsigned long long track = a.track_base + static_cast<unsigned long long>(t);
    const long long toff = (a.traj && active) ? a.traj_off[t] : 0;
    // room of this track in traj: a caller whose offsets do not come from this
    // very simulation must not be able to make the kernel write out of bounds
    const long long troom = (a.traj && active) ? a.traj_off[t + 1] - toff : 0;
    uint32_t pend_a = 0, pend_b = 0;   // words (2,3) of the current Philox block
    bool have_pending = false;
    uint32_t moved = 0;
    int last_it = -1;

    const int lane_id = threadIdx.x & 63;

    // Loop head of movmodel.py:285-291 for the track's current (row, col, k):
    // `done` = the while/break exit, (er, ec) = the cell whose 3x3 window the
    // step evaluates (the burn-in nudge applied).  Evaluated one step ahead so
    // that the table entry of (er, ec) can be fetched before it is needed.
    bool done = false;
    int er = row, ec = col;
    auto loop_head = [&]() {
        done = !(k < max_k);
        er = row;
        ec = col;
        if (!done) {
            if (k > a.burnin) {
                done = !(0 < row && row < a.rows - 1 && 0 < col && col < a.cols - 1);
            } else {
                if (er <= 1) er += 2; else if (er >= a.rows - 2) er -= 2;
                if (ec <= 0) ec += 2; else if (ec >= a.cols - 2) ec -= 2;
            }
        }
    };
    double2 t0 = {0.0, 0.0}, t1 = t0, t2 = t0, t3 = t0;      // prefetched table entry
    auto fetch_entry = [&]() {
        if (MODE == MODE_TABLE) {
            const double2 *src = reinterpret_cast<const double2 *>(
                a.table + (static_cast<size_t>(er) * a.cols + ec) * 8);
            t0 = src[0]; t1 = src[1]; t2 = src[2]; t3 = src[3];
        }
    };
    if (active) loop_head();
    fetch_entry();

    for (int it = 0; it < a.steps; ++it) {
        if (!__any(active)) break;
        bool stepped = false;
        if (active && it >= release) {
            if (done) {
                if (a.lengths) a.lengths[t] = static_cast<int32_t>(k + 1);
                if (a.end_rc)
                    reinterpret_cast<uint32_t *>(a.end_rc)[t] =
                        static_cast<uint32_t>(row & 0xFFFF) | (static_cast<uint32_t>(col) << 16);
                active = false;
            } else {
                // ---- uniform for step k
                uint32_t wa, wb;
                if ((k & 1) && have_pending) {
                    wa = pend_a; wb = pend_b;
                } else {
                    const uint4 w4 = philox_block(a.seed, track, static_cast<unsigned long long>(k) >> 1);
                    if (k & 1) { wa = w4.z; wb = w4.w; }
                    else { wa = w4.x; wb = w4.y; pend_a = w4.z; pend_b = w4.w; }
                }
                have_pending = !(k & 1);
                const double u = words_to_uniform(wa, wb);
                // ---- direction memory (movmodel.py:307-309)
                uint32_t mask = kAllButCentre;
                if (a.memory == 1) {                 // the reference default: last move only
                    mask = restriction_of(dirs & 0xFu);
                } else if (a.memory == 0) {
                    mask = run;
                } else {
                    uint32_t d = dirs;
                    for (int j = 0; j < a.memory; ++j) {
                        mask &= restriction_of(d & 0xFu);
                        d >>= 4;
                    }
                }
                // ---- weights + decision
                int idx = -1;
                if (MODE == MODE_TABLE) {
                    const double tt[8] = {t0.x, t0.y, t1.x, t1.y, t2.x, t2.y, t3.x, t3.y};
                    if (a.fast) idx = choose_table_fast(tt, mask, u);
                    if (__builtin_expect(idx < 0, 0)) {
                        // taken ~1e-12 of the time.  The empty asm makes the inputs opaque
                        // INSIDE the branch: without it hipcc hoists ~100 instructions of the
                        // exact path (clip / mask / NaN scan) above the branch into every step
                        double o[8];
#pragma unroll
                        for (int j = 0; j < 8; ++j) { o[j] = tt[j]; asm volatile("" : "+v"(o[j])); }
                        asm volatile("" : "+v"(mask));
                        const double w[9] = {o[0], o[1], o[2], o[3], 0.0, o[4], o[5], o[6], o[7]};
                        double pr[9];
#pragma unroll
                        for (int j = 0; j < 9; ++j) pr[j] = a.prior[j];
                        idx = choose_move(w, pr, a.nu, mask, u, false);
                    }
                } else {
                    double w[9];
                    if (MODE == MODE_FLUIDFLOW) {
                        window_weights<true>(a.updraft, a.potential, a.cols, er, ec, w);
                    } else if (MODE == MODE_UPDRAFT) {
                        window_weights<false>(a.updraft, a.potential, a.cols, er, ec, w);
                    } else {
#pragma unroll
                        for (int j = 0; j < 9; ++j) w[j] = a.prior[j];
                    }
                    double pr[9];
#pragma unroll
                    for (int j = 0; j < 9; ++j) pr[j] = a.prior[j];
                    idx = choose_move(w, pr, a.nu, mask, u, a.fast != 0);
                }
                row = er + idx / 3 - 1;
                col = ec + idx % 3 - 1;
                dirs = (dirs << 4) | static_cast<uint32_t>(idx);
                if (a.memory == 0) run &= restriction_of(static_cast<uint32_t>(idx));
                ++k;
                ++moved;
                stepped = true;
                loop_head();                       // head of the NEXT step
                if (a.traj && k < troom)
                    reinterpret_cast<uint32_t *>(a.traj)[toff + k] =
                        static_cast<uint32_t>(row & 0xFFFF) | (static_cast<uint32_t>(col) << 16);
            }
        }
        // Next step's table entry: fetched by EVERY lane at the top level of the
        // loop (no divergent phi, so the compiler does not wait on it here) and
        // BEFORE this step's histogram atomic: vmcnt retires in issue order, so
        // a fetch queued behind the atomic would pay the atomic's latency too.
        // (er, ec) is always a cell of the raster, also for finished lanes.
        fetch_entry();
        // ---- presence histogram (K3).  Every lane issues exactly one VMEM op
        // behind the four fetches on every iteration, so the compiler waits for
        // the fetches with vmcnt(1) (a conditional op would force vmcnt(0) and put
        // its round trip on every step's critical path).
        //  * binning mode: the visited cell goes to visits[it][slot] with a
        //    coalesced store; k_bin_visits turns each step's row of the buffer
        //    into histogram counts through an LDS window (device-scope atomics are
        //    memory-side on MI355X: ~2e10 64-B requests/s chip-wide, which made
        //    per-step atomics the stepper's limit: 10.6 ms vs 6.3 ms without)
        //  * fallback: one atomic per lane (idle lanes add 0 to their own cell)
        if (a.visits) {
            a.visits[static_cast<long long>(it) * a.visit_stride + iv] =
                stepped ? __umul24(static_cast<uint32_t>(row), a.vis_r) + __umul24(static_cast<uint32_t>(col), a.vis_c)
                        : 0xFFFFFFFFu;
        } else if (a.hist) {
            uint32_t *h = a.hist;
            if (a.hist_copies)
                h = a.hist_copies + static_cast<size_t>((i >> 6) % static_cast<uint32_t>(a.ncopies)) *
                                        (static_cast<size_t>(a.rows) * a.cols);
            atomicAdd(&h[static_cast<size_t>(row) * a.cols + col], stepped ? 1u : 0u);
        }
        last_it = it;
    }
    // a wave that ran out of live lanes early still owns its slots of the buffer
    if (a.visits)
        for (int it = last_it + 1; it < a.steps; ++it)
            a.visits[static_cast<long long>(it) * a.visit_stride + iv] = 0xFFFFFFFFu;

    // ---- wave-level compaction of the survivors into the next launch's list
    const unsigned long long live = __ballot(active);
    const int lane = lane_id;
    const int nsurv = __popcll(live);
    uint32_t base = 0;
    if (a.ordered) {
        const int wv = threadIdx.x >> 6;
        if (lane == 0) s_cnt[wv] = static_cast<uint32_t>(nsurv);
        __syncthreads();
        if (threadIdx.x == 0) {
            uint32_t tot = 0;
            for (int q = 0; q < kBlock / 64; ++q) tot += s_cnt[q];
            s_cnt[kBlock / 64] = tot ? atomicAdd(&ctl->count[out_slot][xcd], tot) : 0u;
        }
        __syncthreads();
        base = s_cnt[kBlock / 64];
        for (int q = 0; q < wv; ++q) base += s_cnt[q];
    } else {
        if (lane == 0 && nsurv) base = atomicAdd(&ctl->count[out_slot][xcd], static_cast<uint32_t>(nsurv));
        base = __shfl(base, 0);
    }
    if (active) {
        const int rank = __popcll(live & ((1ull << lane) - 1ull));
        a.list_out[xcd * a.cap + base + rank] = t;
        TrackState o;
        o.pos = row | (col << 16);
        o.k = k;
        o.dirs = dirs;
        o.aux = run | (s.aux & ~0x1FFu);
        a.state[t] = o;
    }
    // one atomic per wave for the step total
    unsigned long long m = moved;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m += __shfl_down(m, off);
    if (lane == 0 && m) atomicAdd(&ctl->steps, m);
}

// Decision thresholds of the prior fallback (movmodel.py:234-244): when every admissible
// weight is exactly zero the move distribution depends on (prior, last move) only.
// thr[d * 9 + k] = cdf_k / cdf_8 for last move d, computed with the arithmetic of
// choose_move's exact branch (same pairwise sums, same divisions), so that
// "count of thr <= u" IS np.random.choice's pick.
struct PriorArg { double v[9]; };

// In-band identity of a threshold table: the first bytes of its leading guard band (the speculative
// gathers that land there never use what they read).  ssrs_tracks_simulate refuses a table whose
// header does not name this raster and this prior: the f32 / dword table kinds are flat buffers,
// and a table of another heading has the same layout.
constexpr unsigned long long kThrMagic = 0x3152485453525353ull;      // "SSRSTHR1"
struct ThrHeader {
    unsigned long long magic;
    int32_t rows, cols;
    double prior[9];
};
static_assert(sizeof(ThrHeader) <= 256, "the header lives in the leading guard band (>= 256 bytes)");

// Start of a call, one kernel instead of two memsets, a copy and a threshold kernel (each of
// those cost the stream ~6 us): clears the control block, stores the prior, par_min = max, and
// the prior-fallback thresholds from the by-value prior.
__global__ void k_ctl_init(TrackCtl *ctl, const PriorArg pr, double *__restrict__ thr, WanderWindows *wander,
                           const ThrHeader *__restrict__ header, int rows, int cols)
{
    const int d = threadIdx.x;
    if (d == 33) wander->n = 0;
    if (d < 32) reinterpret_cast<uint32_t *>(ctl->count)[d] = 0;
    if (d == 32) {
        // error bit 1: the threshold table was not built for this raster and this prior
        uint32_t bad = 0;
        if (header) {
            bad = (header->magic != kThrMagic || header->rows != rows || header->cols != cols) ? 2u : 0u;
            for (int k = 0; k < 9; ++k) bad |= (header->prior[k] != pr.v[k]) ? 2u : 0u;
        }
        ctl->error = bad; ctl->par_min = 0xFFFFFFFFu; ctl->steps = 0; ctl->strays = 0; ctl->bin_done = 0; ctl->pad = 0; ctl->roam_slow = 0; ctl->roam_pairs = 0;
        ctl->dbg_tsum = ctl->dbg_tmax = ctl->dbg_waves = ctl->dbg_slowmax = 0;
        ctl->dbg_span = ctl->dbg_span_max = ctl->dbg_waits = 0;
        ctl->roam_stop = ctl->deal_live = 0;
    }
    if (d < 9) ctl->prior[d] = pr.v[d];
    if (d >= 9) return;
    const double *prior = pr.v;
    const uint32_t mask = restriction_of(static_cast<uint32_t>(d));
    double q[9];
    bool any = false;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        q[k] = ((mask >> k) & 1u) && k != 4 ? prior[k] : 0.0;
        any |= (q[k] != 0.0);
    }
    if (!any) {
#pragma unroll
        for (int k = 0; k < 9; ++k) q[k] = prior[k];
    }
    const double s1 = sum9(q);
#pragma unroll
    for (int k = 0; k < 9; ++k) q[k] = q[k] / s1;
    const double s2 = sum9(q);
    double acc = 0.0, cdf[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        q[k] = q[k] / s2;
        acc = acc + q[k];
        cdf[k] = acc;
    }
#pragma unroll
    for (int k = 0; k < 9; ++k) thr[d * 9 + k] = cdf[k] / cdf[8];
}

// ---------------------------------------------------------------- lean stepper
// The production configuration (table, memory_parameter 1, nu = 1, no trajectory
// output, even S) written flat: a step is bound by its chain of dependent
// instructions, and about 100 of the generic kernel's ~230 per step were exec-mask
// bookkeeping of nested divergent branches.  Here
//  * the Philox block is evaluated on even iterations by every lane (release
//    delays are even and S is even, so the parity of a track's step count equals
//    the parity of the iteration for every lane: no per-lane branch),
//  * the loop head of the next step (done / nudged cell) is a pure function of
//    (row, col, k) and is recomputed unconditionally by all lanes,
//  * the move itself is predicated with selects,
//  * lengths / end cells of tracks that finish are written after the loop (a
//    finished lane's row, col, k are frozen),
// which leaves two rare wave-level branches in the loop: the exact decision
// (first step of a track, near-ties, poisoned rows) and the burn-in nudge.
// Results are identical to k_step_tracks<MODE_TABLE, true> (same tests).
constexpr uint32_t pack_slot_deltas(bool rows)
{
    uint32_t v = 0;
    for (int j = 0; j < 8; ++j) {
        const int k = j < 4 ? j : j + 1;
        v |= static_cast<uint32_t>((rows ? dr_of(k) : dc_of(k)) + 1) << (2 * j);
    }
    return v;
}
constexpr uint32_t kSlotDr = pack_slot_deltas(true), kSlotDc = pack_slot_deltas(false);

struct __attribute__((packed, aligned(4))) RingTriple { float x0, x1, x2; };

// RING: a.table is the f32 ring table (ssrs_transition_ring_build): ONE 12-byte
// gather per step instead of three 8-byte ones (a 64-lane gather costs the CU's
// address unit ~64 cycles whatever its width: with 6 waves per CU the three
// gathers were half of a step's time); near-ties are decided by the exact
// sequence on the raw 3x3 windows of updraft / potential.
// ZMASK (ring table only; chosen by the host once a batch no longer moves as a front):
// with real potential fields most steps of a long track happen in dead terrain where all
// three admissible weights are exactly zero and the prior decides.  The 12-byte gather
// of such a step is a random HBM access that only delivers three zeros; the cell's
// zero-mask byte says the same, and a wandering track reuses its 128-cell line for many
// steps.  Lanes whose mask bits are not all set gather as usual (one more dependent
// load, which is why the front-shaped regime does not use this variant).
// VT: visit keys of the transposed histogram (east / west fronts), col * rows + row.
template <bool RING, bool ZMASK = false, bool VT = false>
__global__ __launch_bounds__(kBlock) void k_step_lean(const StepArgs a)
{
    TrackCtl *ctl = a.ctl;
    const int in_slot = a.launch & 3, out_slot = (a.launch + 1) & 3;
    const uint32_t xcd = blockIdx.x % kXcd;
    const uint32_t nlive = ctl->count[in_slot][xcd];
    const uint32_t il = (blockIdx.x / kXcd) * kBlock + threadIdx.x;
    const uint32_t i = xcd * a.cap + il;
    const uint32_t iv = xcd * a.vcap + il;
    if (blockIdx.x == 0 && threadIdx.x < kXcd) ctl->count[(a.launch + 2) & 3][threadIdx.x] = 0;
    if ((il & ~63u) >= nlive) return;

    bool active = il < nlive;
    const int32_t t = active ? (a.list_in ? a.list_in[i] : static_cast<int32_t>(i)) : 0;
    TrackState s = {0, -1, 0, 0};
    if (active) s = a.state[t];
    active = active && s.k >= 0;
    const bool was_active = active;
    int row = s.pos & 0xFFFF, col = (s.pos >> 16) & 0xFFFF;
    int k = s.k;
    const int max_k = static_cast<int>(a.max_k);
    uint32_t dirs = s.dirs;
    // RING: ring position of the last move (8 = none yet) instead of its k
    uint32_t rc = static_cast<uint32_t>(kRingOfK >> (4 * (dirs & 0xFu))) & 0xFu;
    (void)rc;
    const long long rel64 = (a.coherent ? static_cast<long long>(s.aux >> 9) : 0) -
                            static_cast<long long>(a.launch) * a.steps;
    const int release = rel64 > 0x7fffffffLL ? 0x7fffffff : (rel64 < 0 ? 0 : static_cast<int>(rel64));
    const unsigned long long track = a.track_base + static_cast<unsigned long long>(t);
    uint32_t pend_a = 0, pend_b = 0, moved = 0;
    const uint32_t ucols = static_cast<uint32_t>(a.cols);
    const uint32_t in_rows = static_cast<uint32_t>(a.rows - 2), in_cols = static_cast<uint32_t>(a.cols - 2);

    bool done;
    int er, ec;
    // movmodel.py:285-291 for (row, col, k); rows, cols <= 32767
    auto loop_head = [&]() {
        const bool interior = static_cast<uint32_t>(row - 1) < in_rows && static_cast<uint32_t>(col - 1) < in_cols;
        done = !(k < max_k) || (k > a.burnin && !interior);
        er = row;
        ec = col;
        if (__any(k <= a.burnin)) {                       // first burnin steps of a track only
            if (k <= a.burnin) {
                if (er <= 1) er += 2; else if (er >= a.rows - 2) er -= 2;
                if (ec <= 0) ec += 2; else if (ec >= a.cols - 2) ec -= 2;
            }
        }
    };
    double ta = 0.0, tb = 0.0, tc = 0.0;
    uint32_t cand = 0;
    RingTriple x = {0.f, 0.f, 0.f};
    (void)ta; (void)tb; (void)tc; (void)cand; (void)x;
    auto fetch_entry = [&]() {
        const uint32_t cell = __umul24(static_cast<uint32_t>(er), ucols) + static_cast<uint32_t>(ec);
        if (RING) {
            // triple (ring rc-1, rc, rc+1) = floats rc .. rc+2 of the cell's record
            const float *src = reinterpret_cast<const float *>(a.table) +
                               static_cast<size_t>(cell) * kRingFloats + (rc & 7u);
            if (ZMASK) {
                const uint32_t m = a.zmask[cell];
                const uint32_t tri = (((m << 8) | m) >> ((rc + 7u) & 7u)) & 7u;   // bits rc-1, rc, rc+1
                if (tri == 7u && rc != 8u) x = RingTriple{-0.0f, -0.0f, -0.0f};
                else x = *reinterpret_cast<const RingTriple *>(src);
            } else {
#ifdef SSRS_PROBE_NO_GATHER
                (void)src;
                x = RingTriple{1.0f, 2.0f + static_cast<float>(er & 1), 1.5f};
#else
                x = *reinterpret_cast<const RingTriple *>(src);
#endif
            }
        } else {
            cand = candidates_of(dirs & 0xFu);
            const double *src = a.table + static_cast<size_t>(cell) * 8;
            ta = src[cand & 7u];
            tb = src[(cand >> 3) & 7u];
            tc = src[cand >> 6];
        }
    };
    loop_head();
    fetch_entry();

    int last_it = -1;
    auto one_step = [&](const int it, const bool even) {
        const bool go = active && it >= release;
        const bool st = go && !done;
        active = active && !(go && done);                 // finished: row, col, k stay frozen
        // ---- uniform for step k (parity of k == parity of it, see above)
        uint32_t w0, w1;
        if (even) {
            const uint4 w4 = philox_block(a.seed, track, static_cast<unsigned long long>(static_cast<uint32_t>(k) >> 1));
            w0 = w4.x; w1 = w4.y; pend_a = w4.z; pend_b = w4.w;
        } else {
            w0 = pend_a; w1 = pend_b;
        }
        // ---- decision among the three admissible cells
        int nrow, ncol;
        if (RING) {
            const float uf = static_cast<float>(w0 >> 8) * 0x1p-24f;     // top 24 bits of u, exact
            const uint32_t ord = static_cast<uint32_t>(kRingOrder >> (6 * (rc & 7u))) & 63u;
            // (x0, x1, x2) -> ascending k.  Only four orders occur (static_asserts below);
            // written as two-way selects on the bit patterns so that they stay v_cndmask
            // (a three-way pick became branches that dragged the vmcnt wait above Philox)
            const uint32_t b0 = __float_as_uint(x.x0), b1 = __float_as_uint(x.x1), b2 = __float_as_uint(x.x2);
            const bool rot = ord == ring_order(1), rev = ord == ring_order(2), swp = ord == ring_order(5);
            const uint32_t ba = (rot || rev) ? b2 : (swp ? b1 : b0);
            const uint32_t bb = (rot || swp) ? b0 : b1;
            const uint32_t bc = rev ? b0 : (rot ? b1 : b2);
            int sel = choose_three_ring_f32(__uint_as_float(ba), __uint_as_float(bb), __uint_as_float(bc), uf);
            bool slow = st && (sel < 0 || rc == 8u);
            double u = 0.0;
            uint32_t prior_nc = 8u;
            if (__builtin_expect(__any(slow), 0)) {
                u = words_to_uniform(w0, w1);
                if (slow && rc != 8u) {
                    if ((ba & bb & bc) == 0x80000000u && (ba | bb | bc) == 0x80000000u) {
                        // all three admissible weights are exactly zero: the directional
                        // prior decides (movmodel.py:234-240); thresholds precomputed per
                        // last move with the exact arithmetic (k_ctl_init)
                        const double *t = a.thr + 9u * (static_cast<uint32_t>(kKOfRing >> (4 * rc)) & 0xFu);
                        int idx = 0;
#pragma unroll
                        for (int k = 0; k < 9; ++k) idx += t[k] <= u ? 1 : 0;
                        prior_nc = static_cast<uint32_t>(kRingOfK >> (4 * idx)) & 0xFu;
                        slow = prior_nc >= 8u;      // a pick of the centre (unmasked prior): exact path
                    } else {
                        // second tier: f64 sums and the full 53-bit u (band 2^-21)
                        sel = choose_three_ring(__uint_as_float(ba), __uint_as_float(bb), __uint_as_float(bc), u);
                        slow = sel < 0;
                    }
                }
            }
            // ring position of the chosen cell: rc - 1 + (which x it was)
            uint32_t nc = (rc + 7u + ((ord >> (2 * (sel < 0 ? 0 : sel))) & 3u)) & 7u;
            nc = prior_nc < 8u ? prior_nc : nc;
            if (__builtin_expect(__any(slow), 0)) {
                if (slow) {
                    // first step of a track (8 admissible cells), near-ties, poisoned rows:
                    // the reference's exact sequence on the raw windows (movmodel.py:292-312)
                    double w[9];
                    if (a.potential) window_weights<true>(a.updraft, a.potential, a.cols, er, ec, w);
                    else window_weights<false>(a.updraft, a.potential, a.cols, er, ec, w);
                    double pr[9];
#pragma unroll
                    for (int j = 0; j < 9; ++j) pr[j] = a.prior[j];
                    const uint32_t last = static_cast<uint32_t>(kKOfRing >> (4 * rc)) & 0xFu;
                    const int idx = choose_move(w, pr, 1.0, restriction_of(last), u, false);
                    nc = static_cast<uint32_t>(kRingOfK >> (4 * idx)) & 0xFu;
                }
            }
            nrow = er + static_cast<int>((kRingDr >> (2 * nc)) & 3u) - 1;
            ncol = ec + static_cast<int>((kRingDc >> (2 * nc)) & 3u) - 1;
            rc = st ? nc : rc;
        } else {
            const double u = words_to_uniform(w0, w1);
            const uint32_t last = dirs & 0xFu;
            const int sel = choose_three_fast(ta, tb, tc, u);
            const uint32_t slot = (cand >> (3 * (sel < 0 ? 0 : sel))) & 7u;
            int idx = static_cast<int>(slot + (slot >= 4u ? 1u : 0u));
            const bool slow = st && (sel < 0 || last == 4u);
            if (__builtin_expect(__any(slow), 0)) {
                if (slow) {
                    // first step of a track (8 admissible cells), near-ties, poisoned rows:
                    // the reference's exact sequence on the full row (movmodel.py:220-244)
                    const double *src = a.table + static_cast<size_t>(__umul24(static_cast<uint32_t>(er), ucols) +
                                                                      static_cast<uint32_t>(ec)) * 8;
                    double o[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) o[j] = __builtin_nontemporal_load(src + j);
                    const double w[9] = {o[0], o[1], o[2], o[3], 0.0, o[4], o[5], o[6], o[7]};
                    double pr[9];
#pragma unroll
                    for (int j = 0; j < 9; ++j) pr[j] = a.prior[j];
                    idx = choose_move(w, pr, 1.0, restriction_of(last), u, false);
                }
            }
            const int j8 = idx - (idx > 4 ? 1 : 0);
            nrow = er + static_cast<int>((kSlotDr >> (2 * j8)) & 3u) - 1;
            ncol = ec + static_cast<int>((kSlotDc >> (2 * j8)) & 3u) - 1;
            dirs = st ? ((dirs << 4) | static_cast<uint32_t>(idx)) : dirs;
        }
        // ---- move (predicated), head of the next step (unconditional: a pure
        // function of row, col, k, so idle lanes recompute what they had)
        row = st ? nrow : row;
        col = st ? ncol : col;
        k += st ? 1 : 0;
        moved += st ? 1u : 0u;
        loop_head();
        fetch_entry();
        // ---- presence histogram (see k_step_tracks)
        const uint32_t cell = __umul24(static_cast<uint32_t>(row), ucols) + static_cast<uint32_t>(col);
        if (a.visits) {
            const uint32_t key = VT ? __umul24(static_cast<uint32_t>(col), static_cast<uint32_t>(a.rows)) +
                                          static_cast<uint32_t>(row)
                                    : cell;
            a.visits[static_cast<long long>(it) * a.visit_stride + iv] = st ? key : 0xFFFFFFFFu;
        } else if (a.hist) {
            uint32_t *h = a.hist;
            // scattered variant: wave-private copy, so that same-address atomics of
            // different waves do not queue up (compiled out of the front-shaped variant)
            if ((ZMASK || !RING) && a.hist_copies)
                h = a.hist_copies + static_cast<size_t>((i >> 6) % static_cast<uint32_t>(a.ncopies)) *
                                        (static_cast<size_t>(a.rows) * ucols);
            atomicAdd(&h[cell], st ? 1u : 0u);
        }
        last_it = it;
    };
    for (int it = 0; it < a.steps; it += 2) {             // a.steps is even (host)
        if (!__any(active)) break;
        one_step(it, true);
        one_step(it + 1, false);
    }
    if (a.visits)
        for (int it = last_it + 1; it < a.steps; ++it)
            a.visits[static_cast<long long>(it) * a.visit_stride + iv] = 0xFFFFFFFFu;

    // tracks that finished in this launch
    if (was_active && !active) {
        if (a.lengths) a.lengths[t] = static_cast<int32_t>(k + 1);
        if (a.end_rc)
            reinterpret_cast<uint32_t *>(a.end_rc)[t] =
                static_cast<uint32_t>(row & 0xFFFF) | (static_cast<uint32_t>(col) << 16);
    }
    // survivors -> next launch's list of this XCD
    const unsigned long long live = __ballot(active);
    const int lane = threadIdx.x & 63;
    const int nsurv = __popcll(live);
    uint32_t base = 0;
    if (lane == 0 && nsurv) base = atomicAdd(&ctl->count[out_slot][xcd], static_cast<uint32_t>(nsurv));
    base = __shfl(base, 0);
    if (active) {
        const int rank = __popcll(live & ((1ull << lane) - 1ull));
        a.list_out[xcd * a.cap + base + rank] = t;
        TrackState o;
        o.pos = row | (col << 16);
        o.k = k;
        o.dirs = RING ? (static_cast<uint32_t>(kKOfRing >> (4 * rc)) & 0xFu) : dirs;
        o.aux = s.aux;
        a.state[t] = o;
    }
    unsigned long long m = moved;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m += __shfl_down(m, off);
    if (lane == 0 && m) atomicAdd(&ctl->steps, m);
}

__device__ __forceinline__ void split_cell(uint32_t c, uint32_t cols, double inv_cols, uint32_t &r, uint32_t &cc)
{
    r = static_cast<uint32_t>(static_cast<double>(c) * inv_cols);          // c < 2^31: off by one at most
    if (r * cols > c) --r;
    if ((r + 1u) * cols <= c) ++r;
    cc = c - r * cols;
}

// ------------------------------------------------------------ threshold stepper
// k_step_lean still spends ~150 instructions and 7 wave-level branches on a step, and a wave
// that is alone on its SIMD (100k tracks = 1.5 waves per SIMD) pays 6-9 clocks per instruction
// and ~25 per branch whatever the instruction is (tools/microbench/latency.hip): the step is
// bound by its instruction COUNT, not by Philox (hidden under the gather) nor by HBM.  The
// threshold table moves everything that depends on the cell and the last move only -- picking
// the three admissible weights, ordering them by k, summing, normalising -- into the table
// builder.  Per (cell, last move rc) the table holds the two decision thresholds themselves as
// 16-bit fixed point in one dword,
//     T1 = round(2^16 a / (a + b + c)),   T2 = round(2^16 (a + b) / (a + b + c))   (a b c in ascending k,
//     both clamped to 65535)
// and a step is: one 4-byte gather, two integer subtractions from the top 16 bits of the uniform,
// two sign bits -> the chosen cell.  |T - 2^16 cdf_k/cdf_8| <= 0.55 (rounding 0.5; the builder's
// f32 arithmetic 6e-7 relative = 0.04 units; the reference's normalise-twice sequence
// 1e-15; the clamp only matters where
// the uniform's top bits are 65534 / 65535, which the band covers) and the uniform's top 16 bits
// ufi satisfy ufi <= 2^16 u < ufi + 1, so ufi - T >= 1 means cdf_k/cdf_8 <= u and ufi - T <= -2
// means it is not (0.45 units of margin either way): the decision is the reference's unless ufi - T is -1 or 0 (3e-5 per boundary),
// where the exact sequence on the raw windows decides.  Everything irregular is an entry with
// T1 > T2 (impossible otherwise): T1 = 0xFFFF and T2 =
//   0  poisoned row    (a NaN / infinite weight: exact sequence, movmodel.py:228-230)
//   1  boundary cell   (k > burnin: the track ends; else the burn-in nudge: exact sequence)
//   2  reversal        (the three weights AND the masked prior are all zero: the unmasked prior
//                       decides among its own cells, movmodel.py:239-240: thresholds thr9)
// A row whose three weights are zero while the masked prior is not carries the PRIOR's two
// thresholds (movmodel.py:234-238), so the common fallback needs no branch at all.  The cells the
// burn-in nudge moves (rows <= 1, >= rows - 2, cols >= cols - 2) are recognised from the cell
// index while some lane of the wave is in its burn-in (a separate phase of the loop).
// The first move of a track (eight admissible cells) is made by one iteration of the generic
// kernel before the first launch of this one.
// Eight planes (one per last move) of 4-byte entries: neighbouring tracks gather from the same
// few cache lines (32 cells per 128-byte line), which is what the CU's address unit charges for.
constexpr uint32_t kThrPoison = 0x0000FFFFu, kThrBoundary = 0x0001FFFFu, kThrReversal = 0x0002FFFFu;

// The table is eight planes (one per last move rc) of 4-byte entries, plane p at byte offset
// p << thr_plane_shift: a power-of-two stride makes a step's address one shift-or
__host__ __device__ constexpr int thr_plane_shift(int rows, int cols)
{
    const unsigned long long bytes = static_cast<unsigned long long>(rows) * static_cast<unsigned long long>(cols) * 4ull;
    int sh = 8;
    while ((1ull << sh) < bytes) ++sh;
    return sh;
}

__host__ __device__ __forceinline__ uint32_t thr_pack(double b1, double b2)
{   // two boundaries in [0, 1] -> T1 | T2 << 16 (f32 is plenty: 24 bits against 16)
    const float t1 = static_cast<float>(b1) * 65536.0f + 0.5f, t2 = static_cast<float>(b2) * 65536.0f + 0.5f;   // round half up
    uint32_t u1 = t1 >= 65535.0f ? 65535u : static_cast<uint32_t>(t1);
    uint32_t u2 = t2 >= 65535.0f ? 65535u : static_cast<uint32_t>(t2);
    if (u1 > u2) u1 = u2;                                              // b1 <= b2 up to rounding
    return u1 | (u2 << 16);
}

struct ThrPrior {
    uint32_t zero_e[8];      // entry of the masked prior after last move rc
    uint32_t reversal;       // bit rc: the masked prior is all zero as well
    uint32_t thr9[9];        // thresholds of the unmasked prior (2^16 units, clamped), k = 0..8
    // the unmasked prior's cells are the three admissible ones after a move rev_rc (a cosine lobe
    // along a raster axis): a reversal row is then an ordinary row of last move rev_rc with entry rev_e
    uint32_t rev_ok, rev_rc, rev_e;
};

// The builder works in f32 throughout: a threshold only has to land within 0.45 units of 2^-16
// of the reference's boundary (the band above), i.e. 7e-6 in probability, and an f32 weight is
// good to 3e-7 relative (v_rcp_f32 is 1 ulp).  What f32 cannot represent goes to the exact
// sequence instead (a poisoned entry):
//   * an updraft above 1e20 (v_rcp_f32 flushes denormal results): its reciprocal is staged as NaN,
//     which poisons the nine cells around it;
//   * a NaN / infinite weight anywhere in the window (the reference's own rule for NaN,
//     movmodel.py:228-230; f32 overflow joins it): the sum of the eight raw weights is not finite;
//   * a row whose total leaves [1e-30, 1e30] (denormal weights have large RELATIVE errors; the
//     reciprocal of a huge total is flushed).
// Staged reciprocals carry a factor 2^-24, so that a weight 2^24 d / (1/u_c + 1/u_q) with d > 0
// never underflows to zero (>= 8.4 x 1.4e-45): "all three weights are zero" is `tot == 0`, as in
// f64.  The common factor drops out of the thresholds.
// No branches and ~190 vector instructions per cell: a wave64 instruction occupies its SIMD for
// four clocks, so the first version's 365 (+ 45 branches, one per case of every row) took 300 us
// whatever the loads and stores did (probe builds SSRS_PROBE_K2A_*).  v_cvt_pknorm_u16_f32 packs
// both thresholds in one instruction (round(65535 x), saturating: x carries 65536 / 65535).
template <bool HASPOT>
__global__ __launch_bounds__(kBlock) void k_transition_thr(
    const double *__restrict__ updraft, const float *__restrict__ potential,
    uint32_t *__restrict__ table_out, int rows, int cols, int tiles_x, int ntiles, const ThrPrior pr, int plane_shift,
    ThrHeader *__restrict__ header, const PriorArg heading)
{
    constexpr int LW = kTabW + 2, LH = kTabH + 2;
    __shared__ float2 s_cell[LW * LH];                       // x: 2^-24 / clipped updraft, y: potential
    if (blockIdx.x == 0 && threadIdx.x < 9) {
        header->prior[threadIdx.x] = heading.v[threadIdx.x];
        if (threadIdx.x == 0) { header->magic = kThrMagic; header->rows = rows; header->cols = cols; }
    }
    const int t = xcd_band(blockIdx.x, ntiles);
    const int r0 = (t / tiles_x) * kTabH, c0 = (t % tiles_x) * kTabW;
    for (int i = threadIdx.x; i < LW * LH; i += kBlock) {
        const int lr = i / LW, lc = i - lr * LW;
        int gr = r0 - 1 + lr, gc = c0 - 1 + lc;
        gr = gr < 0 ? 0 : (gr >= rows ? rows - 1 : gr);
        gc = gc < 0 ? 0 : (gc >= cols ? cols - 1 : gc);
        const size_t g = static_cast<size_t>(gr) * cols + gc;
        float2 c;
#ifdef SSRS_PROBE_K2A_NOLOAD
        const float v = 1.0f + (g & 255);
        c.y = HASPOT ? 1000.f - (g & 1023) : 0.f;
#else
        const float v = static_cast<float>(updraft[g]);
        c.y = HASPOT ? potential[g] : 0.f;
#endif
        const float w = fmaxf(v, 1e-06f);                    // clip(min=1e-6); NaN is restored below
        c.x = (v <= 1e20f) ? __builtin_amdgcn_rcpf(w) * 5.9604645e-08f : __builtin_nanf("");
        s_cell[i] = c;
    }
    __syncthreads();
    const int lc = static_cast<int>(threadIdx.x % kTabW) + 1;
    const int col = c0 + lc - 1;
    if (col >= cols) return;
    uint32_t zero_row[8];                                    // wave-uniform
#pragma unroll
    for (int rc = 0; rc < 8; ++rc) zero_row[rc] = ((pr.reversal >> rc) & 1u) ? kThrReversal : pr.zero_e[rc];
    for (int lr = static_cast<int>(threadIdx.x / kTabW) + 1; lr <= kTabH; lr += kBlock / kTabW) {
        const int row = r0 + lr - 1;
        if (row >= rows) break;
        const uint32_t off = (static_cast<uint32_t>(row) * static_cast<uint32_t>(cols) + static_cast<uint32_t>(col)) * 4u;   // < 2^29
        const bool interior = row > 0 && col > 0 && row < rows - 1 && col < cols - 1;
        float w[9];
        const float2 cc = s_cell[lr * LW + lc];
        float sum = 0.f;                                     // not finite <=> some raw weight (or the centre) is not
#pragma unroll
        for (int j = 0; j < 9; ++j) {
            if (j == 4) continue;
            const float2 cq = s_cell[(lr + dr_of(j)) * LW + lc + dc_of(j)];
            float v = __builtin_amdgcn_rcpf(cc.x + cq.x);    // 2^23 x harmonic mean
            if (HASPOT) {
                const float d = cc.y - cq.y;
                v *= (j & 1) ? d : d * SSRS_NINV_DIAG;       // d x ninv stays f32 in the reference as well
            }
            sum += v;
            w[j] = fmaxf(v, 0.f);                            // clip(min=0)
        }
        const bool bad = !(fabsf(sum) <= 3e38f);
        const bool ovr = bad | !interior;
        const uint32_t ovr_e = interior ? kThrPoison : kThrBoundary;
#pragma unroll
        for (int rc = 0; rc < 8; ++rc) {
            const uint32_t ord = ring_order(rc);
            const int ring3[3] = {(rc + 7) % 8, rc, (rc + 1) % 8};
            const int ka = kRingK[ring3[ord & 3u]], kb = kRingK[ring3[(ord >> 2) & 3u]], kc = kRingK[ring3[(ord >> 4) & 3u]];
            const float ab = w[ka] + w[kb], tot = ab + w[kc];                   // np.cumsum's order
            const float r = __builtin_amdgcn_rcpf(tot) * 1.0000153f;            // 65536 / 65535
            const auto pk = __builtin_amdgcn_cvt_pknorm_u16(w[ka] * r, ab * r); // T1 | T2 << 16, T1 <= T2
            uint32_t e = __builtin_bit_cast(uint32_t, pk);
            // as integers: 1e-30 <= tot <= 1e30 (tot >= 0 or NaN)
            const bool fine = (__float_as_uint(tot) - 0x0DA24260u) <= (0x7149F2CAu - 0x0DA24260u);
            e = fine ? e : kThrPoison;
            asm volatile("" : "+v"(e));                      // or the compiler branches around the six lines above
            e = (tot == 0.f) ? zero_row[rc] : e;
            e = ovr ? ovr_e : e;
            // plane rc holds the entries of last move rc for all cells, 4 bytes each
#ifdef SSRS_PROBE_K2A_NOSTORE
            if (e == 0x12345678u)
#endif
#ifdef SSRS_PROBE_K2A_PAD      // planes off the power-of-two stride (inside the slack of the last plane; timing only)
            *reinterpret_cast<uint32_t *>(reinterpret_cast<char *>(table_out) + (static_cast<size_t>(rc) << plane_shift) - static_cast<size_t>(rc) * SSRS_PROBE_K2A_PAD + off) = e;
#elif defined(SSRS_K2A_NT)
            __builtin_nontemporal_store(e, reinterpret_cast<uint32_t *>(reinterpret_cast<char *>(table_out) + (static_cast<size_t>(rc) << plane_shift) + off));
#else
            *reinterpret_cast<uint32_t *>(reinterpret_cast<char *>(table_out) + (static_cast<size_t>(rc) << plane_shift) + off) = e;
#endif
        }
    }
}

// Histogram mode of k_step_thr, a template parameter so that the loop body carries no
// wave-uniform branches (each costs a lone wave ~25 clocks, taken or not):
//   0 none, 1 visit buffer (plain keys), 2 visit buffer (transposed keys), 3 atomics on
//   hist / its private copies, 4 visit buffer with 16-bit keys relative to the front's row
//   (north-bound fronts: a visit is its offset from cell (first start row + iteration - 1, 0);
//   halves what the stepper writes and k_bin_visits16 reads; a visit out of that range is
//   counted by the lane itself), 6 a histogram window of the block in LDS (wandering tracks: on
//   the solved 10 m field 44 % of a batch ends up roaming two basins of ~130 x 210 cells until
//   max_moves, 1.8e8 visits per launch onto ~12 000 cells; per-step global atomics queue up on
//   those few lines and per-lane caches of a few cells never hit.  The host sorts the live tracks
//   by 64 x 64 tile (k_wander_keys), so a block's 256 tracks share a basin; the block counts into
//   a kWinRows x kWinCols window around them with LDS atomics and flushes the non-zero cells once
//   per launch; a visit outside the window is a global atomic and a stray)
// The fifth wave of a block (PF): the batch moves as a front, so every step touches table rows no
// one has loaded yet and 4 of 5 gathers contain a lane that waits for HBM (87 % of the L2 requests
// hit, but a 12-line gather waits for its slowest line).  A wave cannot prefetch for itself --
// loads return to a wave in order, the needed gather would queue behind the prefetch -- but ANOTHER
// wave can: this one steps no tracks; it reads where the block's 256 tracks are, and every eight
// iterations it streams the table rows the front will reach 8..16 iterations later (three planes
// around the heading, the block's column span plus a drift margin) through the XCD's L2 with
// coalesced loads.  Paced by the iteration the stepping waves publish in LDS; nobody ever waits for
// it, its waits are bounded, so it cannot hang the block.  North / south headings only (the span is
// contiguous along a raster row).
struct PfArgs {              // by value: taking the kernel arguments' address would move them to scratch
    const int32_t *list_in;
    const TrackState *state;
    const void *table;
    TrackCtl *ctl;
    long long it_base;
    uint32_t cap;
    int coherent, steps, pf_dir, pf_rc, rows, cols, plane_shift;
    int debug;
};
__device__ __forceinline__ void thr_prefetch_wave(const PfArgs a, uint32_t xcd, uint32_t nlive, const volatile int *s_it)
{
    const int lane = threadIdx.x & 63;
    const uint32_t base = (blockIdx.x / kXcd) * kBlock;
    int vr_lo = 0x7fffffff, vr_hi = -0x7fffffff, c_lo = 0x7fffffff, c_hi = -0x7fffffff;
    for (int q = 0; q < kBlock / 64; ++q) {
        const uint32_t il = base + q * 64 + lane;
        if (il >= nlive) continue;
        const uint32_t i = xcd * a.cap + il;
        const int32_t t = a.list_in ? a.list_in[i] : static_cast<int32_t>(i);
        const TrackState s = a.state[t];
        if (s.k < 0) continue;
        const int row = s.pos & 0xFFFF, col = (s.pos >> 16) & 0xFFFF;
        const long long rel64 = (a.coherent ? static_cast<long long>(s.aux >> 9) : 0) + 1 - a.it_base;
        const int rel = rel64 > a.steps ? a.steps : (rel64 < 0 ? 0 : static_cast<int>(rel64));
        const int vr = row - a.pf_dir * rel;              // row at iteration `it` (once released): vr + pf_dir * it
        vr_lo = vr < vr_lo ? vr : vr_lo;  vr_hi = vr > vr_hi ? vr : vr_hi;
        c_lo = col < c_lo ? col : c_lo;   c_hi = col > c_hi ? col : c_hi;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        int o = __shfl_xor(vr_lo, off); vr_lo = o < vr_lo ? o : vr_lo;
        o = __shfl_xor(vr_hi, off); vr_hi = o > vr_hi ? o : vr_hi;
        o = __shfl_xor(c_lo, off); c_lo = o < c_lo ? o : c_lo;
        o = __shfl_xor(c_hi, off); c_hi = o > c_hi ? o : c_hi;
    }
    if (vr_lo > vr_hi) return;                            // no live track in this block
    if (vr_hi - vr_lo > 16) vr_hi = vr_lo + 16;           // stragglers are not worth rows of traffic
    const char *tab = reinterpret_cast<const char *>(a.table);
    uint32_t acc = 0;
    constexpr int kEvery = 8, kLead = 8;
    for (int it0 = 0; it0 < a.steps; it0 += kEvery) {
        // stay at most kEvery iterations ahead of the block's stepping waves; leave with them
        for (int spin = 0; spin < 4096 && *s_it < it0 - kEvery; ++spin) __builtin_amdgcn_s_sleep(16);
        if (*s_it >= 0x3fffffff) break;
        const int margin = 8 + static_cast<int>(2.5f * sqrtf(static_cast<float>(it0 + kLead + kEvery)));
        int c0 = c_lo - margin, c1 = c_hi + margin;
        c0 = c0 < 0 ? 0 : c0;
        c1 = c1 >= a.cols ? a.cols - 1 : c1;
        // rows the front reaches at iterations [it0 + kLead, it0 + kLead + kEvery)
        const int first = (a.pf_dir > 0 ? vr_lo : vr_hi) + a.pf_dir * (it0 + kLead);
        const int nrow = kEvery + (vr_hi - vr_lo);
        const uint32_t span = static_cast<uint32_t>(c1 - c0) + 1u;
        // four rows at a time, two 512-byte pieces of three planes each: 24 loads in flight
        for (int j0 = 0; j0 < nrow; j0 += 4) {
            uint32_t v[24];
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const int r = first + a.pf_dir * (j0 + jj);
                const bool row_ok = j0 + jj < nrow && r >= 0 && r < a.rows;
                const uint32_t cellb = static_cast<uint32_t>(row_ok ? r : 0) * static_cast<uint32_t>(a.cols) + static_cast<uint32_t>(c0);
                const uint32_t lo = (cellb & ~1u) * 4u;                  // 8-byte aligned
                const uint32_t hi = (cellb + span) * 4u;                 // end (exclusive)
#pragma unroll
                for (int p = 0; p < 3; ++p) {
                    const uint32_t plane = static_cast<uint32_t>((a.pf_rc + 7 + p) & 7) << a.plane_shift;
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const uint32_t off = lo + static_cast<uint32_t>(lane) * 8u + static_cast<uint32_t>(h) * 512u;
                        v[(jj * 3 + p) * 2 + h] = (row_ok && off < hi) ? reinterpret_cast<const uint2 *>(tab + (plane | off))->x : 0u;
                    }
                }
            }
#pragma unroll
            for (int q = 0; q < 24; ++q) acc ^= v[q];
        }
    }
    if (acc == 0x9E3779B9u && a.steps < 0) a.ctl->pad = 1;      // keeps the loads alive
}

// LR (round 3): the fifth wave of a front's block STAGES the table rows in LDS instead of pulling them through L2.
// A north / south front advances a row per iteration, its block's 256 tracks span a few hundred columns, and
// only the three planes around the heading are read: 16 rows x 3 planes x 384 columns of entries are 72 KB (two
// blocks per CU).  The staging wave fills a ring of row slots ahead of the slowest stepping wave (each wave
// publishes its iteration), a slot carries the row it holds as a tag, and a stepping lane reads
// tag, entry, tag (three LDS reads, ~100 clocks, against 300-600 for the gather through L2): equal tags on both
// sides of the entry mean the slot was not being rewritten (a wave's LDS operations execute in order, and the
// staging wave invalidates the tag before it rewrites a slot and sets it after).  Anything else -- another plane,
// a column outside the window, a row not staged yet or already gone -- is the global gather of before.
constexpr int kLrRows = 16, kLrCols = 256;          // 48 KB: one dwordx4 wave-load + one ds_write_b128 per row and plane
// LDS accesses of the ring by address space: a `volatile` generic pointer compiles to flat_load ... sc0 sc1 with a
// wait after each (measured: the staged variant 2.5x SLOWER than the gather it replaces); these are ds_read / ds_write
typedef __attribute__((address_space(3))) uint32_t lds_u32;
__device__ __forceinline__ uint32_t lds_addr(const void *p)
{
    return static_cast<uint32_t>(reinterpret_cast<uintptr_t>((const lds_u32 *)p));
}
__device__ __forceinline__ uint32_t lds_ld(const void *p)
{
    return __hip_atomic_load((const lds_u32 *)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void lds_st(void *p, uint32_t v)
{
    __hip_atomic_store((lds_u32 *)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
struct LrShared {
    int c0, vr_lo, vr_hi, on;          // window origin (even), the block's virtual rows at iteration 0, any live track
    int tag[kLrRows];                  // row held by each slot, -1: none / being rewritten
    int itw[kBlock / 64];              // iteration each stepping wave has reached (0x3fffffff: left)
};

__device__ __forceinline__ void thr_stage_geometry(const PfArgs a, uint32_t xcd, uint32_t nlive, LrShared *g)
{
    const int lane = threadIdx.x & 63;
    const uint32_t base = (blockIdx.x / kXcd) * kBlock;
    int vr_lo = 0x7fffffff, vr_hi = -0x7fffffff, c_lo = 0x7fffffff, c_hi = -0x7fffffff;
    for (int q = 0; q < kBlock / 64; ++q) {
        const uint32_t il = base + q * 64 + lane;
        if (il >= nlive) continue;
        const uint32_t i = xcd * a.cap + il;
        const int32_t t = a.list_in ? a.list_in[i] : static_cast<int32_t>(i);
        const TrackState s = a.state[t];
        if (s.k < 0) continue;
        const int row = s.pos & 0xFFFF, col = (s.pos >> 16) & 0xFFFF;
        const long long rel64 = (a.coherent ? static_cast<long long>(s.aux >> 9) : 0) + 1 - a.it_base;
        const int rel = rel64 > a.steps ? a.steps : (rel64 < 0 ? 0 : static_cast<int>(rel64));
        const int vr = row - a.pf_dir * rel;              // row at iteration `it` (once released): vr + pf_dir * it
        vr_lo = vr < vr_lo ? vr : vr_lo;  vr_hi = vr > vr_hi ? vr : vr_hi;
        c_lo = col < c_lo ? col : c_lo;   c_hi = col > c_hi ? col : c_hi;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        int o = __shfl_xor(vr_lo, off); vr_lo = o < vr_lo ? o : vr_lo;
        o = __shfl_xor(vr_hi, off); vr_hi = o > vr_hi ? o : vr_hi;
        o = __shfl_xor(c_lo, off); c_lo = o < c_lo ? o : c_lo;
        o = __shfl_xor(c_hi, off); c_hi = o > c_hi ? o : c_hi;
    }
    if (lane < kLrRows) lds_st(&g->tag[lane], 0xFFFFFFFFu);
    if (lane < kBlock / 64) lds_st(&g->itw[lane], (base + lane * 64 >= nlive) ? 0x3fffffffu : 0xFFFFFFFFu);     // (waves past the list leave at once)
    if (lane == 0) {
        const bool on = vr_lo <= vr_hi;
        if (a.debug && on) {
            atomicAdd(&a.ctl->dbg_span, (static_cast<unsigned long long>(vr_hi - vr_lo) << 20) | 1ull);
            atomicMax(&a.ctl->dbg_span_max, static_cast<unsigned long long>(vr_hi - vr_lo));
        }
        if (vr_hi - vr_lo > kLrRows / 2) { if (a.pf_dir > 0) vr_hi = vr_lo + kLrRows / 2; else vr_lo = vr_hi - kLrRows / 2; }   // stragglers fall back
        int c0 = (c_lo + c_hi + 1 - kLrCols) / 2;                           // window centred on the block's columns
        const int cmax = a.cols - kLrCols;                                  // (>= 0: host)
        c0 = c0 < 0 ? 0 : (c0 > cmax ? cmax : c0);
        lds_st(&g->c0, static_cast<uint32_t>(c0));
        lds_st(&g->vr_lo, static_cast<uint32_t>(vr_lo));
        lds_st(&g->vr_hi, static_cast<uint32_t>(vr_hi));
        lds_st(&g->on, on ? 1u : 0u);
    }
}

__device__ __forceinline__ void thr_stage_wave(const PfArgs a, LrShared *g, uint32_t *ring)
{
    const int lane = threadIdx.x & 63;
    if (!lds_ld(&g->on)) return;
    __builtin_amdgcn_s_setprio(3);                       // ahead of the stepping waves that share its SIMD
    const int dir = a.pf_dir, c0 = static_cast<int>(lds_ld(&g->c0)), vr_lo = static_cast<int>(lds_ld(&g->vr_lo)),
              vr_hi = static_cast<int>(lds_ld(&g->vr_hi));
    const int first = dir > 0 ? vr_lo : vr_hi;           // the row the front needs at iteration 0
    const char *tab = reinterpret_cast<const char *>(a.table);
    typedef __attribute__((address_space(1))) const void gmem_t;
    typedef __attribute__((address_space(3))) void lmem_t;
    // rows go from global memory straight into the ring (global_load_lds_dwordx4: 64 lanes x 16 bytes = one row of
    // one plane per instruction, no registers in between), kBatch rows = 6 loads at a time and TWO batches in
    // flight: a batch's tags are published when the NEXT batch has been issued and `s_waitcnt vmcnt(6)` says the
    // older six have landed.  Every other LDS access of this loop is inline asm: the compiler orders what it
    // sees of LDS against outstanding LDS-DMA with vmcnt(0), which would take the second batch out of flight.
    constexpr int kBatch = 2;
    static_assert(kLrCols == 256 && kLrRows % kBatch == 0, "64 lanes x 4 entries; a batch never wraps inside itself");
    int next = first, pend = 0, pend_row = 0;
    uint32_t n_blocked = 0, n_batches = 0;
    const unsigned long long t_begin = __builtin_amdgcn_s_memtime();
    const uint32_t itw_addr = lds_addr(&g->itw[0]), tag_addr = lds_addr(&g->tag[0]);
    auto publish = [&](int row0) {
        // lanes 0..kBatch-1: tag[slot of row0 + lane] = that row (rows outside the raster keep their -1)
        const int row = row0 + dir * lane;
        if (lane < kBatch && row >= 0 && row < a.rows)
            asm volatile("ds_write_b32 %0, %1" : : "v"(tag_addr + 4u * static_cast<uint32_t>(row & (kLrRows - 1))), "v"(row) : "memory");
    };
    for (int spin = 0; spin < (1 << 20); ++spin) {       // (bounded: the stepping waves always leave)
        int w0, w1, w2, w3;
        static_assert(kBlock / 64 == 4, "four stepping waves");
        asm volatile("ds_read_b32 %0, %4\n\tds_read_b32 %1, %4 offset:4\n\tds_read_b32 %2, %4 offset:8\n\t"
                     "ds_read_b32 %3, %4 offset:12\n\ts_waitcnt lgkmcnt(0)"
                     : "=&v"(w0), "=&v"(w1), "=&v"(w2), "=&v"(w3) : "v"(itw_addr) : "memory");
        int itmin = w0 < w1 ? w0 : w1;
        itmin = w2 < itmin ? w2 : itmin;
        itmin = w3 < itmin ? w3 : itmin;
        itmin = __builtin_amdgcn_readfirstlane(itmin);
        if (itmin >= 0x3fffffff) break;                   // every stepping wave has left
        itmin = itmin < 0 ? 0 : itmin;
        // the slot of row `next` still holds row next -/+ kLrRows, which the slowest wave may need until it has
        // passed it; and nothing beyond the rows this launch can reach
        const int ahead = dir * (next - first);           // rows ahead of iteration 0's front
        if (ahead + kBatch - 1 > itmin + kLrRows - 1 || ahead > a.steps + (vr_hi - vr_lo)) {
            if (pend) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                publish(pend_row);
                pend = 0;
            } else {
                ++n_blocked;
                __builtin_amdgcn_s_sleep(1);
            }
            continue;
        }
        ++n_batches;
        {   // the batch's slots are in flight from here on
            const int row = next + dir * lane;
            if (lane < kBatch)
                asm volatile("ds_write_b32 %0, %1" : : "v"(tag_addr + 4u * static_cast<uint32_t>(row & (kLrRows - 1))), "v"(-1) : "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
#pragma unroll
        for (int q = 0; q < kBatch; ++q) {
            const int row = next + dir * q, slot = row & (kLrRows - 1);
            const int rr = row < 0 ? 0 : (row >= a.rows ? a.rows - 1 : row);      // (never published when clamped)
            const uint32_t cellb = (static_cast<uint32_t>(rr) * static_cast<uint32_t>(a.cols) + static_cast<uint32_t>(c0)) * 4u;
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                const uint32_t plane = static_cast<uint32_t>((a.pf_rc + 7 + p) & 7) << a.plane_shift;
                // (4-byte aligned in global memory: row x cols + c0 is any integer)
                __builtin_amdgcn_global_load_lds((gmem_t *)(tab + (plane + cellb + static_cast<uint32_t>(lane) * 16u)),
                                                 (lmem_t *)&ring[(slot * 3 + p) * kLrCols], 16, 0, 0);
            }
        }
        if (pend) {
            asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            static_assert(kBatch * 3 == 6, "the count above");
            publish(pend_row);
        }
        pend = 1;
        pend_row = next;
        next += dir * kBatch;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // (nothing of this wave is in flight into LDS when it leaves)
    if (a.debug && lane == 0) {
        // diagnostics: batches staged, polls that found the ring full, rows staged, the wave's lifetime in clocks
        atomicAdd(&a.ctl->roam_pairs, (static_cast<unsigned long long>(n_batches) << 32) | n_blocked);
        atomicAdd(&a.ctl->roam_slow, ((static_cast<unsigned long long>(dir * (next - first)) & 0xFFFFFFFFull) << 32) |
                                         ((__builtin_amdgcn_s_memtime() - t_begin) >> 8));
    }
}

// REV: reversal rows decided in the fast path (ThrPrior::rev_*).  In the basins of a solved field a
// track falls to the bottom of a pit (a move south, say), finds every way on uphill and the masked
// prior empty, takes the unmasked prior's move north and falls back: every other step is a reversal,
// and as a flag entry each one sent its whole wave through the slow path (1030 issue clocks per
// wave-step against 490 on the ramp).  Three more instructions on the chain: variants without the
// prefetch wave only.
template <int HM, bool PF = false, bool REV = false, bool LR = false>
__global__ __launch_bounds__(PF ? kBlock + 64 : kBlock) void k_step_thr(const StepArgs a, const ThrPrior pr)
{
    static_assert(!(HM == 6 && PF), "the block window is for batches without a front");
    static_assert(!LR || (PF && !REV), "staged rows: fronts with a fifth wave");
    __shared__ uint32_t s_win[HM == 6 ? kWinRows * kWinCols : 1];
    __shared__ __attribute__((aligned(16))) uint32_t s_ring[LR ? kLrRows * 3 * kLrCols : 1];
    __shared__ LrShared s_lr;
    __shared__ uint32_t s_cnt[kBlock / 64 + 1];            // LR: survivors per wave (block-ordered list reservation)
    if (LR && threadIdx.x <= kBlock / 64) s_cnt[threadIdx.x] = 0u;
    __shared__ int s_box[4];
    if (HM == 6) {
        for (int q = threadIdx.x; q < kWinRows * kWinCols; q += kBlock) s_win[q] = 0u;
        if (threadIdx.x == 0) { s_box[0] = s_box[1] = 0x7fffffff; s_box[2] = s_box[3] = -1; }
        __syncthreads();
    }
    TrackCtl *ctl = a.ctl;
    const int in_slot = a.launch & 3, out_slot = (a.launch + 1) & 3;
    const uint32_t xcd = blockIdx.x % kXcd;
    const uint32_t nlive = ctl->count[in_slot][xcd];
    __shared__ int s_it;                                  // PF: iteration reached by the stepping waves
    // Candidate table: row rc = the three admissible moves after last move rc in the order of the
    // decision (a: u < T1, b: u < T2, c: else) as {cell delta a, b, c, packed (nc | dr << 3 | dc << 5)
    // of a, b, c, plane offset a, b, c, -}.  A lane reads its row when its last move is known, long
    // before the gather returns; what then depends on the gather is two subtractions, two compares and
    // the selects of the delta and of the plane.  Plane offsets include the guard band and the byte
    // offset is a SUM (plane + 4 cell, modulo 2^32): a speculative neighbour of a boundary cell lies
    // below plane 0 or beyond plane 7, inside the bands.
    __shared__ alignas(16) uint32_t s_lut[8 * 8];
    if (threadIdx.x < 8) {
        const uint32_t r8 = threadIdx.x;
        const uint32_t ord8 = static_cast<uint32_t>(kRingOrder >> (6u * r8)) & 63u;
        uint32_t pack = 0;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const uint32_t ncj = (r8 + 7u + ((ord8 >> (2 * j)) & 3u)) & 7u;
            const uint32_t drj = (kRingDr >> (2u * ncj)) & 3u, dcj = (kRingDc >> (2u * ncj)) & 3u;
            s_lut[r8 * 8 + j] = drj * static_cast<uint32_t>(a.cols) + dcj - (static_cast<uint32_t>(a.cols) + 1u);
            s_lut[r8 * 8 + 4 + j] = a.guard + (ncj << a.plane_shift);
            pack |= (ncj | (drj << 3) | (dcj << 5)) << (8 * j);
        }
        s_lut[r8 * 8 + 3] = pack;
        s_lut[r8 * 8 + 7] = 0;
    }
    if (!PF) __syncthreads();
    if (PF) {
        const PfArgs pa = {a.list_in, a.state, reinterpret_cast<const char *>(a.table) + a.guard, a.ctl, a.it_base, a.cap, a.coherent, a.steps,
                           a.pf_dir, a.pf_rc, a.rows, a.cols, a.plane_shift, a.debug_roam};
        if (threadIdx.x == 0) s_it = -1;
        if (LR && threadIdx.x >= kBlock) thr_stage_geometry(pa, xcd, nlive, &s_lr);
        __syncthreads();
        if (threadIdx.x >= kBlock) {
            if (LR) thr_stage_wave(pa, &s_lr, s_ring);
            else thr_prefetch_wave(pa, xcd, nlive, &s_it);
            return;
        }
    }
    const uint32_t il = (blockIdx.x / kXcd) * kBlock + threadIdx.x;
    const uint32_t i = xcd * a.cap + il;
    const uint32_t iv = xcd * a.vcap + il;
    if (blockIdx.x == 0 && threadIdx.x < kXcd) ctl->count[(a.launch + 2) & 3][threadIdx.x] = 0;
    if (HM != 6 && (il & ~63u) >= nlive) return;          // (HM 6: every wave meets the block's barriers)

    bool live0 = il < nlive;
    int32_t t = live0 ? (a.list_in ? a.list_in[i] : static_cast<int32_t>(i)) : 0;
    if (HM == 6 && t < 0) { live0 = false; t = 0; }      // tombstone (k_deal_sorted, this kernel's own dead)
    TrackState s = {0, -1, 0, 0};
    if (live0) s = a.state[t];
    uint32_t rc = static_cast<uint32_t>(kRingOfK >> (4 * (s.dirs & 0xFu))) & 0xFu;
    // (every track has made its first move before the first launch of this kernel: rc < 8)
    live0 = live0 && s.k >= 0 && rc < 8u;
    // HM 6: the window the host's sort filled this block from; before the first sort (or for tracks
    // outside every window) the bounding box of the block's tracks decides
    int wr = 0, wc = 0, win_r0 = 0, win_c0 = 0;
    uint32_t win_stray = 0;
    if (HM == 6) {
        const int row0 = s.pos & 0xFFFF, col0 = (s.pos >> 16) & 0xFFFF;
        int r_lo = live0 ? row0 : 0x7fffffff, r_hi = live0 ? row0 : -1, c_lo = live0 ? col0 : 0x7fffffff, c_hi = live0 ? col0 : -1;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            int o = __shfl_xor(r_lo, off); r_lo = o < r_lo ? o : r_lo;
            o = __shfl_xor(r_hi, off); r_hi = o > r_hi ? o : r_hi;
            o = __shfl_xor(c_lo, off); c_lo = o < c_lo ? o : c_lo;
            o = __shfl_xor(c_hi, off); c_hi = o > c_hi ? o : c_hi;
        }
        if ((threadIdx.x & 63) == 0 && r_hi >= 0) {
            atomicMin(&s_box[0], r_lo); atomicMin(&s_box[1], c_lo);
            atomicMax(&s_box[2], r_hi); atomicMax(&s_box[3], c_hi);
        }
        __syncthreads();
        const bool fits = s_box[2] - s_box[0] < kWinRows && s_box[3] - s_box[1] < kWinCols;
        // window of the block's first live track (all of them, after a padded deal)
        const int nwin = a.wander->n;
        int wid = live0 && nwin > 0 ? wander_window_of(a.wander, nwin, row0, col0) : 0x7fffffff;
        wid = wid >= nwin ? 0x7fffffff : wid;
        {
            const unsigned long long lm = __ballot(wid != 0x7fffffff);
            wid = lm ? __shfl(wid, __ffsll(static_cast<long long>(lm)) - 1) : 0x7fffffff;
        }
        __shared__ int s_wid[kBlock / 64];
        if ((threadIdx.x & 63) == 0) s_wid[threadIdx.x >> 6] = wid;
        __syncthreads();
        wid = 0x7fffffff;
        for (int q = kBlock / 64 - 1; q >= 0; --q) wid = s_wid[q] != 0x7fffffff ? s_wid[q] : wid;
        if (wid != 0x7fffffff && threadIdx.x == 0) {
            s_box[0] = a.wander->r0[wid]; s_box[2] = s_box[0] + kWinRows - 1;
            s_box[1] = a.wander->c0[wid]; s_box[3] = s_box[1] + kWinCols - 1;
        }
        const bool placed = wid != 0x7fffffff;
        __syncthreads();
        if (!placed && !fits && threadIdx.x == 0) {
            // a box larger than the window: the first wave's box decides
            s_box[0] = r_lo; s_box[1] = c_lo; s_box[2] = r_hi; s_box[3] = c_hi;
        }
        __syncthreads();
        if (s_box[2] >= 0) {
            // centred; a box still larger than the window: its middle
            win_r0 = (s_box[0] + s_box[2] + 1 - kWinRows) / 2;
            win_c0 = (s_box[1] + s_box[3] + 1 - kWinCols) / 2;
        }
        wr = row0 - win_r0;
        wc = col0 - win_c0;
    }
    const uint32_t ucols = static_cast<uint32_t>(a.cols), urows = static_cast<uint32_t>(a.rows);
    const uint32_t ncell = urows * ucols;
    uint32_t cell = __umul24(static_cast<uint32_t>(s.pos & 0xFFFF), ucols) + (static_cast<uint32_t>(s.pos >> 16) & 0xFFFFu);
    rc &= 7u;
    int k = s.k;
    // this launch covers the global iterations [it_base, it_base + steps); a track is released at
    // global iteration delay + 1 (its first move was made before, so k - iteration stays even:
    // one Philox block serves the even / odd pair of iterations).  A lane steps while
    // rel <= it < rel + span; span shrinks to the iterations left until max_moves and drops to 0
    // when the track ends.
    const long long rel64 = (a.coherent ? static_cast<long long>(s.aux >> 9) : 0) + 1 - a.it_base;
    const int rel = rel64 > 0x3fffffffLL ? 0x3fffffff : (rel64 < 0 ? 0 : static_cast<int>(rel64));
    const long long left = a.max_k - k;                                          // k < max_k  <=>  it - rel < left
    uint32_t span = !live0 ? 0u : (left > 0x3fffffffLL ? 0x3fffffffu : (left < 0 ? 0u : static_cast<uint32_t>(left)));
    const long long burn64 = static_cast<long long>(rel) + (a.burnin - k);      // k <= burnin  <=>  it <= it_burn
    const int it_burn = !live0 ? -1 : (burn64 > 0x3fffffffLL ? 0x3fffffff : (burn64 < -1 ? -1 : static_cast<int>(burn64)));
    const int blk0 = (k - rel) >> 1;          // Philox block of iteration 0 (k - rel is even; negative before the release)
    const unsigned long long track = a.track_base + static_cast<unsigned long long>(t);
    const char *tab = reinterpret_cast<const char *>(a.table);
    uint32_t pend_a = 0, pend_b = 0;
    const uint32_t back = ucols + 1u;
    // last iteration at which some lane of this wave is still in its burn-in (zone test needed)
    int wave_burn = it_burn;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const int o = __shfl_xor(wave_burn, off);
        wave_burn = o > wave_burn ? o : wave_burn;
    }
    wave_burn = __builtin_amdgcn_readfirstlane(wave_burn);
    const uint32_t psh = static_cast<uint32_t>(a.plane_shift);
    uint32_t plane = a.guard + (rc << psh);
    uint32_t e = *reinterpret_cast<const uint32_t *>(tab + (plane + (cell << 2)));
    uint4 lutA = *reinterpret_cast<const uint4 *>(&s_lut[rc * 8]), lutB = *reinterpret_cast<const uint4 *>(&s_lut[rc * 8 + 4]);
    const uint32_t rev_e = pr.rev_e, rev_rc = pr.rev_rc;
    // column of the cell, kept up to date only while the burn-in phase of the loop runs
    uint32_t colv = static_cast<uint32_t>(s.pos >> 16) & 0xFFFFu;
    const uint32_t zone_lo = 2u * ucols, zone_hi = (urows - 2u) * ucols, zone_col = ucols - 2u;
    uint32_t *vrow = (HM == 1 || HM == 2) ? a.visits + iv : nullptr;              // this lane's slot, row `it`
    uint16_t *vrow16 = HM == 4 ? reinterpret_cast<uint16_t *>(a.visits) + iv : nullptr;
    // HM 4: cell (first start row + global iteration - 1, 0); may start below 0: compared as int
    int32_t vbase = HM == 4 ? (static_cast<int32_t>(ctl->par_min) - a.v16_offset + static_cast<int32_t>(a.it_base) - 1) *
                                  static_cast<int32_t>(ucols) : 0;
    uint32_t *hbase = a.hist;
    if (HM == 3 && a.hist_copies)
        hbase = a.hist_copies + static_cast<size_t>((i >> 6) % static_cast<uint32_t>(a.ncopies)) * static_cast<size_t>(ncell);
    int it = 0;
    const bool cheap_exact = a.cheap_exact != 0;
    // LR: row and window column of the cell (the staged rows are addressed by them), the staging wave's window
    const int lr_c0 = LR ? static_cast<int>(lds_ld(&s_lr.c0)) : 0;
    const uint32_t lr_h0 = static_cast<uint32_t>(a.pf_rc + 7) & 7u;              // first of the three staged planes
    int lrow = s.pos & 0xFFFF, lcol = ((s.pos >> 16) & 0xFFFF) - lr_c0;
    const uint32_t lr_tag0 = lds_addr(s_lr.tag), lr_ring0 = lds_addr(s_ring);   // LDS byte addresses
    uint32_t lr_miss = 0;                                                       // wave-steps that fell back to the gather
    uint32_t lr_why[4] = {0u, 0u, 0u, 0u};
    uint32_t lr_waits = 0;                                                      // polls of a wave that waited for its row
    auto report = [&](int v) __attribute__((always_inline)) {                   // progress of this wave, for the fifth one
        if ((threadIdx.x & 63) == 0) {
            if (LR) lds_st(&s_lr.itw[(threadIdx.x >> 6) & (kBlock / 64 - 1)], static_cast<uint32_t>(v));
            else atomicMax(&s_it, v);
        }
    };

    auto one_step = [&](const bool even, const bool burn, const bool publish) __attribute__((always_inline)) {
        if (PF && publish && even && (LR || (it & 7) == 0)) report(it);
        // st: all ones when this lane steps now
        const uint32_t stm = (static_cast<uint32_t>(it - rel) < span) ? 0xFFFFFFFFu : 0u;
        uint32_t w0, w1;
        if (even) {
            const uint4 w4 = philox_block_b3(a.seed, track, static_cast<unsigned long long>(static_cast<uint32_t>(blk0 + (it >> 1))));
            w0 = w4.x; w1 = w4.y; pend_a = w4.z; pend_b = w4.w;
        } else {
            w0 = pend_a; w1 = pend_b;
        }
        const uint32_t ufi = w0 >> 16;                               // top 16 bits of u
        // Two decodes.  LUT: the candidates of the last move come from the LDS table and the next gather
        // is issued before anything else is known -- shortest dependent chain, for launches that wait
        // on their gathers (fronts: misses, a dozen lines per gather).  ARITH (the REV variants outside
        // the burn-in): the candidates are worked out from the ring position after the entry is known,
        // one gather after the special test -- measured on the solved field: 3.23 s per pass against
        // 3.50 s with the LUT decode and its reversal merge on the chain.
        const bool arith = REV && !burn;
        int32_t d1, d2;
        uint32_t nc, dr = 0u, dc = 0u, cell_n = cell, pl = plane, e_n = e;
        if (!arith) {
            const uint32_t la = lutA.x, lb = lutA.y, lc = lutA.z, lpk = lutA.w, pa = lutB.x, pb = lutB.y, pc = lutB.z;
            d1 = static_cast<int32_t>(ufi) - static_cast<int32_t>(e & 0xFFFFu);
            d2 = static_cast<int32_t>(ufi) - static_cast<int32_t>(e >> 16);
            // a (u < T1), b (u < T2) or c; a lane that does not step now stays.  Flag entries and near-ties
            // take one of the three as well (a neighbour cell: inside the table or its guard bands) and
            // are put right below.
            const bool s1 = d1 < 0, s2 = d2 < 0;
            const uint32_t pcur = plane;
            uint32_t dl = s2 ? lb : lc;
            pl = s2 ? pb : pc;
            dl = s1 ? la : dl;  pl = s1 ? pa : pl;
            dl = stm ? dl : 0u; pl = stm ? pl : pcur;
            cell_n = cell + dl;
            // which one it was (off the chain; LR: on it)
            const uint32_t fld = (lpk >> (s1 ? 0u : (s2 ? 8u : 16u))) & 0xFFu;
            nc = fld & 7u; dr = (fld >> 3) & 3u; dc = (fld >> 5) & 3u;
            if (!LR) {
                e_n = *reinterpret_cast<const uint32_t *>(tab + (pl + (cell_n << 2)));
            } else {
                // the entry of the cell the move leads to, from the staged rows: tag, entry, tag (in this order)
                const int row_n = lrow + static_cast<int>((dr - 1u) & stm), col_n = lcol + static_cast<int>((dc - 1u) & stm);
                const uint32_t pidx = ((stm ? nc : rc) - lr_h0) & 7u;
                const uint32_t slot = static_cast<uint32_t>(row_n) & (kLrRows - 1);
                const bool ok = (pidx < 3u) & (static_cast<uint32_t>(col_n) < static_cast<uint32_t>(kLrCols));
                const uint32_t idx = ok ? (slot * 3u + pidx) * kLrCols + static_cast<uint32_t>(col_n) : 0u;
                uint32_t t1, el, t2;
                {
                    const uint32_t ta = lr_tag0 + (slot << 2), ea = lr_ring0 + (idx << 2);
                    asm volatile("ds_read_b32 %0, %3\n\tds_read_b32 %1, %4\n\tds_read_b32 %2, %3\n\ts_waitcnt lgkmcnt(0)"
                                 : "=&v"(t1), "=&v"(el), "=&v"(t2) : "v"(ta), "v"(ea) : "memory");
                }
                bool need = (stm != 0u) & !(ok & (static_cast<int>(t1) == row_n) & (static_cast<int>(t2) == row_n));
                e_n = stm ? el : e;                                    // (a lane that does not step keeps its entry)
                if (__builtin_expect(__any(need), 0)) {
                    // A row that is not there YET (slot empty or still holding the row of a ring turn before): this
                    // wave is ahead of the block's slowest one by most of the ring.  It waits for the staging wave
                    // instead of overtaking it with gathers -- the block ends with its slowest wave either way, and
                    // waves that run apart find nothing staged from then on (measured: the waves of a block
                    // diffuse apart through their exact decisions until the leaders ride the ring's edge, 68 % of
                    // the wave-steps fell back).  Bounded: the slowest wave never waits, so the staging wave
                    // always gets here; a row this launch never stages is gathered after the last poll.
                    const int dirw = a.pf_dir;
                    bool wait = need & ok & ((static_cast<int>(t1) < 0) | (dirw * (static_cast<int>(t1) - row_n) < 0));
                    for (int sp = 0; a.lr_wait && sp < 256 && __any(wait); ++sp) {
                        __builtin_amdgcn_s_sleep(2);
                        const uint32_t ta = lr_tag0 + (slot << 2), ea = lr_ring0 + (idx << 2);
                        asm volatile("ds_read_b32 %0, %3\n\tds_read_b32 %1, %4\n\tds_read_b32 %2, %3\n\ts_waitcnt lgkmcnt(0)"
                                     : "=&v"(t1), "=&v"(el), "=&v"(t2) : "v"(ta), "v"(ea) : "memory");
                        const bool hit = (static_cast<int>(t1) == row_n) & (static_cast<int>(t2) == row_n);
                        if (wait & hit) { e_n = el; need = false; }
                        wait = wait & !hit & ((static_cast<int>(t1) < 0) | (dirw * (static_cast<int>(t1) - row_n) < 0));
                        if (a.debug_roam) ++lr_waits;
                    }
                }
                if (__builtin_expect(__any(need), 0)) {
                    if (need) e_n = *reinterpret_cast<const uint32_t *>(tab + (pl + (cell_n << 2)));
                    ++lr_miss;
                    if (a.debug_roam) {       // which kind (lane-steps): window / plane, slot empty, slot holds an older row, a newer one
                        lr_why[0] += (need && !ok) ? 1u : 0u;
                        lr_why[1] += (need && ok && static_cast<int>(t1) < 0) ? 1u : 0u;
                        lr_why[2] += (need && ok && static_cast<int>(t1) >= 0 && static_cast<int>(t1) < row_n) ? 1u : 0u;
                        lr_why[3] += (need && pidx >= 3u) ? 1u : 0u;          // (of the first kind: the plane)
                    }
                }
            }
        } else {
            // a reversal row is an ordinary row of the move along the heading with the prior's thresholds
            const bool rev = e == kThrReversal;
            const uint32_t eu = rev ? rev_e : e, rcd = rev ? rev_rc : rc;
            d1 = static_cast<int32_t>(ufi) - static_cast<int32_t>(eu & 0xFFFFu);
            d2 = static_cast<int32_t>(ufi) - static_cast<int32_t>(eu >> 16);
            const uint32_t ord = static_cast<uint32_t>(kRingOrder >> __umul24(6u, rcd)) & 63u;
            const uint32_t neg = (static_cast<uint32_t>(d1) >> 31) + (static_cast<uint32_t>(d2) >> 31);     // 2 - sel
            nc = (rcd + 7u + ((ord >> (4u - 2u * neg)) & 3u)) & 7u;
        }
        // ufi - T in {-1, 0}: the uniform is within rounding of a boundary; T1 > T2 (d1 < d2): a flag entry
        bool special = (static_cast<uint32_t>(d1 + 1) < 2u) | (static_cast<uint32_t>(d2 + 1) < 2u) | (d1 < d2);
        if (burn) {
            const bool zone = (cell < zone_lo) | (cell >= zone_hi) | (colv >= zone_col);
            special = special | (zone & (it <= it_burn));
        }
        special = special & (stm != 0u);
        uint32_t base = cell, base_col = colv;
        int base_row = 0;
        const uint32_t cell_before = cell;
        uint32_t go = stm;
        if (__builtin_expect(__any(special), 0)) {
            if (special) {
                bool exact = true;
                if (e == kThrBoundary && k > a.burnin) {
                    go = 0u;                                           // movmodel.py:286-288: the track ends here
                    span = 0u;
                    exact = false;
                } else if (e == kThrReversal && k > a.burnin) {
                    // unmasked prior (movmodel.py:239-240): count of thresholds <= u
                    int idx = 0;
                    bool near = false;
#pragma unroll
                    for (int q = 0; q < 9; ++q) {
                        const int32_t d = static_cast<int32_t>(ufi) - static_cast<int32_t>(pr.thr9[q]);
                        idx += d >= 0 ? 1 : 0;
                        near |= static_cast<uint32_t>(d + 1) < 2u;
                    }
                    const uint32_t pnc = static_cast<uint32_t>(kRingOfK >> (4 * (idx > 8 ? 4 : idx))) & 0xFu;
                    if (!near && pnc < 8u) { nc = pnc; exact = false; }
                }
                if (exact) {
                    // near-ties, poisoned rows, the burn-in nudge: the reference's exact sequence on
                    // the raw windows (movmodel.py:285-312)
                    uint32_t r, c;
                    split_cell(cell, ucols, 1.0 / static_cast<double>(ucols), r, c);
                    int er = static_cast<int>(r), ec = static_cast<int>(c);
                    if (k <= a.burnin) {
                        if (er <= 1) er += 2; else if (er >= a.rows - 2) er -= 2;
                        if (ec <= 0) ec += 2; else if (ec >= a.cols - 2) ec -= 2;
                    }
                    const uint32_t last = static_cast<uint32_t>(kKOfRing >> (4 * rc)) & 0xFu;
                    const double uu = words_to_uniform(w0, w1);
                    int idx = -1;
                    if (cheap_exact) idx = a.potential ? exact_three<true>(a.updraft, a.potential, a.cols, er, ec, last, a.thr, uu)
                                                       : exact_three<false>(a.updraft, a.potential, a.cols, er, ec, last, a.thr, uu);
                    if (idx < 0) {
                        double w[9];
                        if (a.potential) window_weights<true>(a.updraft, a.potential, a.cols, er, ec, w);
                        else window_weights<false>(a.updraft, a.potential, a.cols, er, ec, w);
                        double prr[9];
#pragma unroll
                        for (int j = 0; j < 9; ++j) prr[j] = a.prior[j];
                        idx = choose_move(w, prr, 1.0, restriction_of(last), uu, false);
                    }
                    nc = static_cast<uint32_t>(kRingOfK >> (4 * idx)) & 0xFu;
                    base = __umul24(static_cast<uint32_t>(er), ucols) + static_cast<uint32_t>(ec);
                    base_col = static_cast<uint32_t>(ec);
                    base_row = er;
                    if (HM == 6) { wr += er - static_cast<int>(r); wc += ec - static_cast<int>(c); }      // the nudge
                }
            }
            if (!arith) {
                // the move as decided (every lane: the others find what they already have) and the gather
                // again for the lanes that guessed wrong (the first gather stays ONE load for both paths,
                // issued before this branch)
                dr = (kRingDr >> (2u * nc)) & 3u;
                dc = (kRingDc >> (2u * nc)) & 3u;
                const uint32_t moved_to = base + __umul24(dr, ucols) + dc - back;
                const uint32_t cell_t = (moved_to & go) | (cell & ~go);
                const uint32_t pl_t = ((a.guard + (nc << psh)) & go) | (plane & ~go);
                if ((cell_t != cell_n) | (pl_t != pl)) e_n = *reinterpret_cast<const uint32_t *>(tab + (pl_t + (cell_t << 2)));
                cell_n = cell_t;
                pl = pl_t;
            }
        }
        if (arith) {
            dr = (kRingDr >> (2u * nc)) & 3u;
            dc = (kRingDc >> (2u * nc)) & 3u;
            const uint32_t moved_to = base + __umul24(dr, ucols) + dc - back;
            cell_n = (moved_to & go) | (cell & ~go);
            pl = a.guard + (((nc & go) | (rc & ~go)) << psh);
            e_n = *reinterpret_cast<const uint32_t *>(tab + (pl + (cell_n << 2)));
        }
        // ---- commit (idle lanes keep cell, rc, k)
        cell = cell_n;
        plane = pl;
        rc = (nc & go) | (rc & ~go);
        k -= static_cast<int>(go);                                    // go is 0 or -1
        if (burn) {
            // column of the new cell (the exact path may have moved the base: recompute there)
            const uint32_t moved_col = base == cell_before ? colv + dc - 1u : base_col + dc - 1u;
            colv = (moved_col & go) | (colv & ~go);
        }
        if (LR) {
            // (base_col is only meaningful where the exact path set it: base != cell_before)
            const int nr = (base == cell_before ? lrow : base_row) + static_cast<int>(dr) - 1;
            const int ncl = (base == cell_before ? lcol : static_cast<int>(base_col) - lr_c0) + static_cast<int>(dc) - 1;
            lrow = go ? nr : lrow;
            lcol = go ? ncl : lcol;
        }
        e = e_n;
        if (!arith) {
            lutA = *reinterpret_cast<const uint4 *>(&s_lut[rc * 8]);
            lutB = *reinterpret_cast<const uint4 *>(&s_lut[rc * 8 + 4]);
        }
        // ---- presence histogram (see k_step_tracks)
        if (HM == 1) {
            *vrow = cell | ~go;                                       // idle: 0xFFFFFFFF
            vrow += a.visit_stride;
        } else if (HM == 2) {
            uint32_t r, c;
            split_cell(cell, ucols, 1.0 / static_cast<double>(ucols), r, c);
            *vrow = (__umul24(c, urows) + r) | ~go;
            vrow += a.visit_stride;
        } else if (HM == 3) {
            atomicAdd(&hbase[cell], go & 1u);
        } else if (HM == 6) {
            wr += static_cast<int>((dr - 1u) & go);                    // go is 0 or all ones
            wc += static_cast<int>((dc - 1u) & go);
            const bool inside = (static_cast<uint32_t>(wr) < static_cast<uint32_t>(kWinRows)) &
                                (static_cast<uint32_t>(wc) < static_cast<uint32_t>(kWinCols));
            if (go != 0u && inside) atomicAdd(&s_win[wr * kWinCols + wc], 1u);     // ds_add_u32, nothing returned
            const bool out = (go != 0u) & !inside;
            if (__builtin_expect(__any(out), 0)) {
                if (out) { atomicAdd(&a.hist[cell], 1u); ++win_stray; }
            }
        } else if (HM == 4) {
            const uint32_t key = static_cast<uint32_t>(static_cast<int32_t>(cell) - vbase);
            const bool stray = (go != 0u) & (key >= 0xFFFFu);
            if (__builtin_expect(__any(stray), 0)) {
                // (counted per lane and reported once when the launch ends: one atomic on ctl->strays per
                // wave-step -- 1.6 M on one address in a launch over a solved field -- took 20-38 ms)
                if (stray) { atomicAdd(&a.hist[cell], 1u); ++win_stray; }
            }
            *vrow16 = static_cast<uint16_t>((go != 0u && !stray) ? key : 0xFFFFu);
            vrow16 += a.visit_stride;
            vbase += static_cast<int32_t>(ucols);
        }
        ++it;
    };
    // phase A: some lane of the wave is in its burn-in (the nudge zone is live); phase B: none
    for (; it < a.steps && it <= wave_burn; ) {            // a.steps is even (host)
        if (!__any(static_cast<uint32_t>(it - rel) < span || it < rel)) break;
        one_step(true, true, true);
        one_step(false, true, true);
    }
    // eight iterations per trip: one liveness test, one progress report to the prefetch wave and one
    // backward branch instead of four of each (a wave pays 25-40 clocks per branch, taken or not)
    for (; it + 8 <= a.steps; ) {
        if (!__any(static_cast<uint32_t>(it - rel) < span || (it < rel && span != 0u))) break;
        if (PF) report(it);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (LR && u == 2) report(it);                  // (the staging wave looks at the slowest wave: every four iterations)
            one_step(true, false, false);
            one_step(false, false, false);
        }
    }
    for (; it < a.steps; ) {
        if (!__any(static_cast<uint32_t>(it - rel) < span || (it < rel && span != 0u))) break;
        one_step(true, false, true);
        one_step(false, false, true);
    }
    if (HM == 1 || HM == 2)
        for (; it < a.steps; ++it) {
            *vrow = 0xFFFFFFFFu;
            vrow += a.visit_stride;
        }
    if (HM == 4)
        for (; it < a.steps; ++it) {
            *vrow16 = 0xFFFFu;
            vrow16 += a.visit_stride;
        }
    if (PF) report(0x3fffffff);                                        // nothing left to wait for
    if (LR && a.debug_roam) {
        if ((threadIdx.x & 63) == 0) {
            atomicAdd(&ctl->dbg_tsum, static_cast<unsigned long long>(lr_miss));
            atomicAdd(&ctl->dbg_waves, static_cast<unsigned long long>(it));
            atomicAdd(&ctl->dbg_waits, static_cast<unsigned long long>(lr_waits));
        }
        // (lane-steps by kind, packed: 16 bits of millions would overflow: two 64-bit words, 32 bits each)
        atomicAdd(&ctl->dbg_tmax, (static_cast<unsigned long long>(lr_why[0]) << 32) | lr_why[1]);
        atomicAdd(&ctl->dbg_slowmax, (static_cast<unsigned long long>(lr_why[2]) << 32) | lr_why[3]);
    }
    if (HM == 6) {
        __syncthreads();
        for (int q = threadIdx.x; q < kWinRows * kWinCols; q += kBlock) {
            const uint32_t n = s_win[q];
            // (cells of the window outside the raster were never counted)
            if (n) atomicAdd(&a.hist[static_cast<uint32_t>(win_r0 + q / kWinCols) * ucols + static_cast<uint32_t>(win_c0 + q % kWinCols)], n);
        }
    }
    if (HM == 6 || HM == 4) {
        unsigned long long ws = win_stray;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) ws += __shfl_down(ws, off);
        if ((threadIdx.x & 63) == 0 && ws) atomicAdd(&ctl->strays, ws);
    }

    // a track whose span is used up is finished: it ended at the raster's edge or took max_moves
    const bool active = live0 && span != 0u && k < static_cast<int>(a.max_k);
    uint32_t row, col;
    split_cell(cell, ucols, 1.0 / static_cast<double>(ucols), row, col);
    if (live0 && !active) {
        if (a.lengths) a.lengths[t] = static_cast<int32_t>(k + 1);
        if (a.end_rc) reinterpret_cast<uint32_t *>(a.end_rc)[t] = (row & 0xFFFFu) | (col << 16);
    }
    const unsigned long long live = __ballot(active);
    const int lane = threadIdx.x & 63;
    const int nsurv = __popcll(live);
    uint32_t basei = 0;
    if (HM == 6) {
        // one reservation per BLOCK, the waves in order inside it: the tracks the host sorted into
        // this block stay one run of the next list (per-wave reservations land in arrival order and
        // would shuffle the basins' waves into every block)
        __shared__ uint32_t s_surv[kBlock / 64 + 1];
        const int wv = threadIdx.x >> 6;
        if (lane == 0) s_surv[wv] = static_cast<uint32_t>(nsurv);
        __syncthreads();
        if (threadIdx.x == 0) {
            uint32_t tot = 0;
            for (int q = 0; q < kBlock / 64; ++q) tot += s_surv[q];
            s_surv[kBlock / 64] = tot ? atomicAdd(&ctl->count[out_slot][xcd], static_cast<uint32_t>(kBlock)) : 0xFFFFFFFFu;
        }
        __syncthreads();
        basei = s_surv[kBlock / 64];
        if (basei != 0xFFFFFFFFu) {
            // the block keeps its kBlock slots (every lane its own): the dead leave tombstones
            a.list_out[xcd * a.cap + basei + threadIdx.x] = active ? t : -1;
            if (active) {
                TrackState o;
                o.pos = static_cast<int32_t>(row | (col << 16));
                o.k = k;
                o.dirs = static_cast<uint32_t>(kKOfRing >> (4 * rc)) & 0xFu;
                o.aux = s.aux;
                a.state[t] = o;
            }
        }
    } else if (LR) {
        // one reservation per block, its waves in order: the block's tracks stay neighbours in the next list
        // (per-wave reservations land in arrival order and scatter a block over its XCD's band of columns --
        // harmless for gathers through L2, fatal for a 256-column window in LDS: a third of the lane-steps fell
        // outside it).  The waves past the list and the staging wave have left; a barrier counts the waves
        // that are still there.
        const int wv = threadIdx.x >> 6;
        if (lane == 0) s_cnt[wv] = static_cast<uint32_t>(nsurv);
        __syncthreads();
        if (threadIdx.x == 0) {
            uint32_t tot = 0;
            for (int q = 0; q < kBlock / 64; ++q) tot += s_cnt[q];
            s_cnt[kBlock / 64] = tot ? atomicAdd(&ctl->count[out_slot][xcd], tot) : 0u;
        }
        __syncthreads();
        basei = s_cnt[kBlock / 64];
        for (int q = 0; q < wv; ++q) basei += s_cnt[q];
    } else {
        if (lane == 0 && nsurv) basei = atomicAdd(&ctl->count[out_slot][xcd], static_cast<uint32_t>(nsurv));
        basei = __shfl(basei, 0);
    }
    if (HM != 6 && active) {
        const int rank = __popcll(live & ((1ull << lane) - 1ull));
        a.list_out[xcd * a.cap + basei + rank] = t;
        TrackState o;
        o.pos = static_cast<int32_t>(row | (col << 16));
        o.k = k;
        o.dirs = static_cast<uint32_t>(kKOfRing >> (4 * rc)) & 0xFu;
        o.aux = s.aux;
        a.state[t] = o;
    }
    unsigned long long mv = live0 ? static_cast<unsigned long long>(k - s.k) : 0ull;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) mv += __shfl_down(mv, off);
    if (lane == 0 && mv) atomicAdd(&ctl->steps, mv);
}

// ------------------------------------------------------------ pair table (round 3)
// What a roaming batch costs: on the solved 10 m field 44 % of a batch circles in two basins until
// max_moves = 7.5e6, every lane of a wave somewhere else in its basin, and a step of k_step_thr<6> is one
// fully divergent 4-byte gather.  The CU's address unit takes ~4 clocks per lane of such a load
// (tools/microbench/gather.hip: ~7 per distinct line; 256 lanes per CU -> the measured 1030 clocks per
// iteration) whatever the latency: the round-2 notes called the regime latency-bound; the first version
// of this table (64-byte entries with successor indices, four loads per pair of moves) ran 1.8x SLOWER
// although it halved the instructions and took the decode off the chain, which settled it
// (profiles/r03_notes.md).  So the table serves one thing: fewer divergent loads per move.
//   * a state is (cell, last move); its 16-byte entry holds the thresholds of the state AND of the three
//     cells its move can lead to, so ONE dwordx4 load decides a PAIR of moves (one Philox block = two
//     uniforms); the 8 states of a cell are one 128-byte line;
//   * the table covers the whole raster (128 bytes per cell, 3.84 GB at 5000 x 6000, built in ~2 ms when
//     a batch starts to roam): a first version with a slab per wander window lost 2-3x to the ONE wave
//     per launch that had a lane outside its slab -- a launch lasts as long as its slowest wave.
// Everything irregular stays out of the fast path: a flag entry, a near-tie, the burn-in, a pair cut
// short by max_moves or by the release schedule sends the lane through `slow_step`, the single-move
// sequence of k_step_thr on the plain threshold table (and from there, where needed, through the
// reference's exact sequence on the raw windows), after which the lane looks its state up again.
// Decisions are the same integers compared with the same thresholds as in k_step_thr, so the results
// are identical (same tests; SSRS_TRACKS_NO_ROAM_TABLE is the A/B switch).
struct alignas(16) RoamEntry {
    uint32_t e0, ea, eb, ec;     // raw threshold entries of the state and of the cells its move leads to
                                 // (a: u < T1, b: u < T2, c: else; after the reversal substitution of e0)
};
static_assert(sizeof(RoamEntry) == 16, "one dwordx4 per state");
constexpr size_t kPairMaxCells = 1ull << 25;       // 32-bit byte offsets: 128 bytes per cell

template <bool REV>
__global__ __launch_bounds__(kBlock) void k_roam_build(const char *__restrict__ tab, uint32_t guard, int plane_shift,
                                                      int rows, int cols, RoamEntry *__restrict__ out, const ThrPrior pr)
{
    // one thread per state; a block covers 32 consecutive cells x 8 last moves, the grid walks the raster
    const uint32_t nstate = static_cast<uint32_t>(rows) * static_cast<uint32_t>(cols) * 8u;
    for (uint32_t s = blockIdx.x * kBlock + threadIdx.x; s < nstate; s += gridDim.x * kBlock) {
        const uint32_t rc = s & 7u, cell = s >> 3;
        auto entry = [&](uint32_t c, uint32_t q) {
            return *reinterpret_cast<const uint32_t *>(tab + (guard + (q << plane_shift)) + (c << 2));
        };
        uint4 en = make_uint4(entry(cell, rc), kThrPoison, kThrPoison, kThrPoison);
        const bool rev = REV && en.x == kThrReversal;
        const uint32_t eu = rev ? pr.rev_e : en.x, rcd = rev ? pr.rev_rc : rc;
        if ((eu & 0xFFFFu) <= (eu >> 16)) {
            // (a row that is no flag belongs to an interior cell: its neighbours are cells of the raster)
            const uint32_t ord = static_cast<uint32_t>(kRingOrder >> (6u * rcd)) & 63u;
            uint32_t es[3];
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const uint32_t ncj = (rcd + 7u + ((ord >> (2 * j)) & 3u)) & 7u;
                const uint32_t cj = cell + ((kRingDr >> (2u * ncj)) & 3u) * static_cast<uint32_t>(cols) + ((kRingDc >> (2u * ncj)) & 3u) -
                                    static_cast<uint32_t>(cols) - 1u;
                es[j] = entry(cj, ncj);
            }
            en.y = es[0]; en.z = es[1]; en.w = es[2];
        }
        reinterpret_cast<uint4 *>(out)[s] = en;
    }
}

// Near-ties.  0.6 % of a roaming wave's pairs have a lane whose uniform lies within one unit (2^-16) of a
// threshold; the reference's exact sequence on the raw windows then costs that wave ~25 000 clocks (18 + 26
// f64 divisions in a lone wave), a sixth of its time, and the launch waits for the wave that drew the most of
// them.  The fine table holds the same two boundaries per (cell, last move) as 32-bit fixed point, computed
// in f64 from the reference's own weights (movmodel.py:292-306: harmonic mean x f32 potential difference x
// f32 1/norm, clipped at 0): |T - 2^32 cdf_k/cdf_8| <= 0.5 + 2^32 x ~1e-15 (x = a / (a + b + c) against
// the reference's normalise-twice-and-divide: a dozen roundings of 2^-53), and the uniform's top 32 bits
// u32 satisfy u32 <= 2^32 u < u32 + 1, so u32 - T >= 1 means cdf <= u, u32 - T <= -2 means it is not, and only
// u32 - T in {-1, 0} (5e-10 per boundary) still needs the exact sequence.  Rows the 16-bit table flags
// (boundary, poison, reversal) and rows whose weights are not finite are flags here too (T1 > T2).
struct FineEntry { uint32_t t1, t2; };
struct FinePrior { uint32_t zero_t1[8], zero_t2[8], reversal; };     // the masked prior's boundaries after last move rc

__host__ __device__ __forceinline__ uint32_t fine_fixed(double x)
{   // round(2^32 x) for x in [0, 1], half up, clamped
    const double v = x * 4294967296.0 + 0.5;
    return v >= 4294967295.0 ? 0xFFFFFFFFu : static_cast<uint32_t>(static_cast<unsigned long long>(v));
}

template <bool HAS_POT>
__global__ __launch_bounds__(kBlock) void k_fine_build(const double *__restrict__ updraft, const float *__restrict__ potential,
                                                      FineEntry *__restrict__ out, int rows, int cols, int tiles_x, int ntiles,
                                                      const FinePrior fp)
{
    // the weights exactly as k_transition_table forms them (LDS tile of the clipped updraft's reciprocals
    // and of the potential: identical operands, identical bits)
    constexpr int LW = kTabW + 2, LH = kTabH + 2;
    __shared__ double s_inv[LW * LH];
    __shared__ float s_pot[LW * LH];
    const int t = xcd_band(blockIdx.x, ntiles);
    const int r0 = (t / tiles_x) * kTabH, c0 = (t % tiles_x) * kTabW;
    for (int i = threadIdx.x; i < LW * LH; i += kBlock) {
        const int lr = i / LW, lc = i - lr * LW;
        int gr = r0 - 1 + lr, gc = c0 - 1 + lc;
        gr = gr < 0 ? 0 : (gr >= rows ? rows - 1 : gr);
        gc = gc < 0 ? 0 : (gc >= cols ? cols - 1 : gc);
        const size_t g = static_cast<size_t>(gr) * cols + gc;
        const double v = updraft[g];
        const double w = v != v ? v : (v > 1e-06 ? v : 1e-06);   // clip(min=1e-06), NaN kept
        s_inv[i] = 1.0 / w;
        s_pot[i] = HAS_POT ? potential[g] : 0.f;
    }
    __syncthreads();
    const int lc = static_cast<int>(threadIdx.x % kTabW) + 1;
    const int col = c0 + lc - 1;
    if (col >= cols) return;
    for (int lr = static_cast<int>(threadIdx.x / kTabW) + 1; lr <= kTabH; lr += kBlock / kTabW) {
        const int row = r0 + lr - 1;
        if (row >= rows) break;
        const size_t cell = static_cast<size_t>(row) * cols + col;
        const bool interior = row > 0 && col > 0 && row < rows - 1 && col < cols - 1;
        double w[9];
        bool bad = !interior;
        if (interior) {
            const double ic = s_inv[lr * LW + lc];
            const float pc = s_pot[lr * LW + lc];
#pragma unroll
            for (int j = 0; j < 9; ++j) {
                const int o = (lr + dr_of(j)) * LW + lc + dc_of(j);
                w[j] = 2.0 / (ic + s_inv[o]);                       // harmonic mean
                if (HAS_POT) {
                    const float d = pc - s_pot[o];
                    const float ninv = (j == 4) ? 0.f : ((j & 1) ? 1.f : SSRS_NINV_DIAG);
                    const float e = d * ninv;                       // stays f32
                    w[j] = w[j] * static_cast<double>(e);
                }
                bad |= !(w[j] - w[j] == 0.0);                       // NaN or infinite (the centre's too: movmodel.py:228)
                w[j] = w[j] > 0.0 ? w[j] : 0.0;                     // clip(min=0)
            }
        }
        FineEntry en[8];
#pragma unroll
        for (int rc = 0; rc < 8; ++rc) {
            const uint32_t ord = ring_order(rc);
            const int ring3[3] = {(rc + 7) % 8, rc, (rc + 1) % 8};
            const int ka = kRingK[ring3[ord & 3u]], kb = kRingK[ring3[(ord >> 2) & 3u]], kc = kRingK[ring3[(ord >> 4) & 3u]];
            FineEntry e = {0xFFFFFFFFu, 0u};                         // flag
            if (!bad) {
                const double ab = w[ka] + w[kb], tot = ab + w[kc];   // np.cumsum's order
                if (tot == 0.0) {
                    // all three weights zero: the masked prior decides (movmodel.py:234-238), unless it is empty too
                    if (!((fp.reversal >> rc) & 1u)) { e.t1 = fp.zero_t1[rc]; e.t2 = fp.zero_t2[rc]; }
                } else if (tot - tot == 0.0) {                       // finite
                    e.t1 = fine_fixed(w[ka] / tot);
                    e.t2 = fine_fixed(ab / tot);
                    if (e.t1 > e.t2) e.t1 = e.t2;                    // (a <= a + b up to rounding)
                }
            }
            en[rc] = e;
        }
        uint4 *dst = reinterpret_cast<uint4 *>(out + cell * 8);
#pragma unroll
        for (int q = 0; q < 4; ++q) dst[q] = make_uint4(en[2 * q].t1, en[2 * q].t2, en[2 * q + 1].t1, en[2 * q + 1].t2);
    }
}

// BT = 256: one block of list slots (rounds 3-4).  BT = 512 / 1024 ("wide", round 4): TWO / FOUR consecutive blocks of the list --
// which the wide deal fills from one window -- share the CU's one 144-KB window with 8 / 16 waves, 2 / 4 per SIMD: a batch whose
// survivors outnumber 256 CUs x 256 lanes then still steps in ONE round of blocks (k_deal_sorted, profiles/r04_roam_fill.txt)
template <bool REV, int BT = kBlock>
__global__ __launch_bounds__(BT) void k_step_roam(const StepArgs a, const ThrPrior pr)
{
    __shared__ uint32_t s_win[kWinRows * kWinCols];
    __shared__ int s_box[4];
    __shared__ int s_wid[BT / 64];
    for (int q = threadIdx.x; q < kWinRows * kWinCols; q += BT) s_win[q] = 0u;
    if (threadIdx.x == 0) { s_box[0] = s_box[1] = 0x7fffffff; s_box[2] = s_box[3] = -1; }
    __syncthreads();
    TrackCtl *ctl = a.ctl;
    const int in_slot = a.launch & 3, out_slot = (a.launch + 1) & 3;
    const uint32_t xcd = blockIdx.x % kXcd;
    const uint32_t nlive = ctl->count[in_slot][xcd];
    const uint32_t il = (blockIdx.x / kXcd) * BT + threadIdx.x;
    const uint32_t i = xcd * a.cap + il;
    if (blockIdx.x == 0 && threadIdx.x < kXcd) ctl->count[(a.launch + 2) & 3][threadIdx.x] = 0;

    bool live0 = il < nlive;
    int32_t t = live0 ? (a.list_in ? a.list_in[i] : static_cast<int32_t>(i)) : 0;
    if (t < 0) { live0 = false; t = 0; }                  // tombstone
    TrackState s = {0, -1, 0, 0};
    if (live0) s = a.state[t];
    uint32_t rc = static_cast<uint32_t>(kRingOfK >> (4 * (s.dirs & 0xFu))) & 0xFu;
    live0 = live0 && s.k >= 0 && rc < 8u;                 // (every track has made its first move)
    rc &= 7u;
    int row = s.pos & 0xFFFF, col = (s.pos >> 16) & 0xFFFF;
    // the block's histogram window: as in k_step_thr<6> (the window the sort filled this block from; the
    // bounding box of its tracks otherwise)
    int win_r0 = 0, win_c0 = 0;
    {
        int r_lo = live0 ? row : 0x7fffffff, r_hi = live0 ? row : -1, c_lo = live0 ? col : 0x7fffffff, c_hi = live0 ? col : -1;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            int o = __shfl_xor(r_lo, off); r_lo = o < r_lo ? o : r_lo;
            o = __shfl_xor(r_hi, off); r_hi = o > r_hi ? o : r_hi;
            o = __shfl_xor(c_lo, off); c_lo = o < c_lo ? o : c_lo;
            o = __shfl_xor(c_hi, off); c_hi = o > c_hi ? o : c_hi;
        }
        if ((threadIdx.x & 63) == 0 && r_hi >= 0) {
            atomicMin(&s_box[0], r_lo); atomicMin(&s_box[1], c_lo);
            atomicMax(&s_box[2], r_hi); atomicMax(&s_box[3], c_hi);
        }
        __syncthreads();
        const bool fits = s_box[2] - s_box[0] < kWinRows && s_box[3] - s_box[1] < kWinCols;
        const int nwin = a.wander->n;
        int wid = live0 && nwin > 0 ? wander_window_of(a.wander, nwin, row, col) : 0x7fffffff;
        wid = wid >= nwin ? 0x7fffffff : wid;
        {
            const unsigned long long lm = __ballot(wid != 0x7fffffff);
            wid = lm ? __shfl(wid, __ffsll(static_cast<long long>(lm)) - 1) : 0x7fffffff;
        }
        if ((threadIdx.x & 63) == 0) s_wid[threadIdx.x >> 6] = wid;
        __syncthreads();
        wid = 0x7fffffff;
        for (int q = BT / 64 - 1; q >= 0; --q) wid = s_wid[q] != 0x7fffffff ? s_wid[q] : wid;
        if (wid != 0x7fffffff && threadIdx.x == 0) {
            s_box[0] = a.wander->r0[wid]; s_box[2] = s_box[0] + kWinRows - 1;
            s_box[1] = a.wander->c0[wid]; s_box[3] = s_box[1] + kWinCols - 1;
        }
        const bool placed = wid != 0x7fffffff;
        __syncthreads();
        if (!placed && !fits && threadIdx.x == 0) { s_box[0] = r_lo; s_box[1] = c_lo; s_box[2] = r_hi; s_box[3] = c_hi; }
        __syncthreads();
        if (s_box[2] >= 0) {
            win_r0 = (s_box[0] + s_box[2] + 1 - kWinRows) / 2;
            win_c0 = (s_box[1] + s_box[3] + 1 - kWinCols) / 2;
        }
    }
    const char *pair = reinterpret_cast<const char *>(a.roam);
    const uint32_t ucols = static_cast<uint32_t>(a.cols);
    const uint32_t last_cell = static_cast<uint32_t>(a.rows) * ucols - 1u;
    int k = s.k;
    // iterations [it_base, it_base + steps) of the batch; a lane steps while rel <= it < rel + span
    // (see k_step_thr: rel is odd or 0, k - rel is even, one Philox block per even / odd pair)
    const long long rel64 = (a.coherent ? static_cast<long long>(s.aux >> 9) : 0) + 1 - a.it_base;
    const int rel = rel64 > 0x3fffffffLL ? 0x3fffffff : (rel64 < 0 ? 0 : static_cast<int>(rel64));
    const long long left = a.max_k - k;
    uint32_t span = !live0 ? 0u : (left > 0x3fffffffLL ? 0x3fffffffu : (left < 0 ? 0u : static_cast<uint32_t>(left)));
    const int blk0 = (k - rel) >> 1;
    const unsigned long long track = a.track_base + static_cast<unsigned long long>(t);
    const char *tab = reinterpret_cast<const char *>(a.table);
    const uint32_t psh = static_cast<uint32_t>(a.plane_shift);
    const uint32_t rev_e = pr.rev_e;
    uint32_t win_stray = 0, n_fine = 0;
    const void *fine = a.fine;

    // ---- the lane's place: window coordinates (wr, wc) -- any integers -- the cell as a linear index,
    // and the entry of its state.  `fast`: released and beyond its burn-in
    int wr = 0, wc = 0, lcell = 0;                          // lcell: the same cell as a linear index (what the gather needs)
    bool fast = false;
    uint4 E = make_uint4(kThrPoison, kThrPoison, kThrPoison, kThrPoison);
    auto enter = [&](bool released) __attribute__((always_inline)) {
        wr = row - win_r0;
        wc = col - win_c0;
        lcell = row * a.cols + col;
        fast = live0 && released && span != 0u && k > a.burnin;
        const uint32_t sidx = fast ? ((static_cast<uint32_t>(lcell) << 3) | rc) : 0u;
        E = *reinterpret_cast<const uint4 *>(pair + (sidx << 4));
    };
    enter(rel == 0);
    const uint32_t rev_rc4 = 4u * pr.rev_rc;

    // ---- one move by the rules of k_step_thr's special branch, written plainly (rare: ~1 % of the pairs)
    auto slow_step = [&](const uint32_t w0, const uint32_t w1) __attribute__((always_inline)) {
        const uint32_t ufi = w0 >> 16;
        const uint32_t cell = static_cast<uint32_t>(row) * ucols + static_cast<uint32_t>(col);
        const uint32_t e = *reinterpret_cast<const uint32_t *>(tab + (a.guard + (rc << psh)) + (cell << 2));
        const int32_t d1 = static_cast<int32_t>(ufi) - static_cast<int32_t>(e & 0xFFFFu);
        const int32_t d2 = static_cast<int32_t>(ufi) - static_cast<int32_t>(e >> 16);
        const bool burn = k <= a.burnin;
        bool special = (static_cast<uint32_t>(d1 + 1) < 2u) | (static_cast<uint32_t>(d2 + 1) < 2u) | (d1 < d2);
        if (burn) special |= (row <= 1) | (row >= a.rows - 2) | (col >= a.cols - 2);
        uint32_t nc;
        int br = row, bc = col;
        if (!special) {
            const uint32_t ord = static_cast<uint32_t>(kRingOrder >> (6u * rc)) & 63u;
            const uint32_t neg = (static_cast<uint32_t>(d1) >> 31) + (static_cast<uint32_t>(d2) >> 31);
            nc = (rc + 7u + ((ord >> (4u - 2u * neg)) & 3u)) & 7u;
        } else {
            if (e == kThrBoundary && !burn) { span = 0u; return; }      // movmodel.py:286-288: the track ends here
            bool exact = true;
            nc = 0;
            if (fine && !(d1 < d2) && !(burn && ((row <= 1) | (row >= a.rows - 2) | (col >= a.cols - 2)))) {
                // a near-tie of an ordinary row: the same two boundaries at 32 bits
                const uint2 f = reinterpret_cast<const uint2 *>(fine)[(cell << 3) | rc];
                const uint32_t u32 = (w0 & ~31u) | (w1 >> 27);          // top 32 bits of the 53-bit uniform
                const long long f1 = static_cast<long long>(u32) - static_cast<long long>(f.x);
                const long long f2 = static_cast<long long>(u32) - static_cast<long long>(f.y);
                const bool unsure = (f.x > f.y) | (static_cast<unsigned long long>(f1 + 1) < 2ull) | (static_cast<unsigned long long>(f2 + 1) < 2ull);
                if (!unsure) {
                    const uint32_t ord = static_cast<uint32_t>(kRingOrder >> (6u * rc)) & 63u;
                    const uint32_t neg = (f1 < 0 ? 1u : 0u) + (f2 < 0 ? 1u : 0u);
                    nc = (rc + 7u + ((ord >> (4u - 2u * neg)) & 3u)) & 7u;
                    exact = false;
                    ++n_fine;
                }
            }
            if (e == kThrReversal && !burn) {
                // unmasked prior (movmodel.py:239-240): count of thresholds <= u
                int idx = 0;
                bool near = false;
#pragma unroll
                for (int q = 0; q < 9; ++q) {
                    const int32_t d = static_cast<int32_t>(ufi) - static_cast<int32_t>(pr.thr9[q]);
                    idx += d >= 0 ? 1 : 0;
                    near |= static_cast<uint32_t>(d + 1) < 2u;
                }
                const uint32_t pnc = static_cast<uint32_t>(kRingOfK >> (4 * (idx > 8 ? 4 : idx))) & 0xFu;
                if (!near && pnc < 8u) { nc = pnc; exact = false; }
            }
            if (exact) {
                // near-ties, poisoned rows, the burn-in nudge: the reference's exact sequence on the raw
                // windows (movmodel.py:285-312)
                int er = row, ec = col;
                if (burn) {
                    if (er <= 1) er += 2; else if (er >= a.rows - 2) er -= 2;
                    if (ec <= 0) ec += 2; else if (ec >= a.cols - 2) ec -= 2;
                }
                double w[9];
                if (a.potential) window_weights<true>(a.updraft, a.potential, a.cols, er, ec, w);
                else window_weights<false>(a.updraft, a.potential, a.cols, er, ec, w);
                double prr[9];
#pragma unroll
                for (int j = 0; j < 9; ++j) prr[j] = a.prior[j];
                const uint32_t last = static_cast<uint32_t>(kKOfRing >> (4 * rc)) & 0xFu;
                // (the guarded division-free comparison first: it hands over to the 26 divisions of the literal
                // sequence only within 2^-46 of a boundary, so the pick is the reference's either way)
                const int idx = choose_move(w, prr, 1.0, restriction_of(last), words_to_uniform(w0, w1), true);
                nc = static_cast<uint32_t>(kRingOfK >> (4 * idx)) & 0xFu;
                br = er; bc = ec;
            }
        }
        row = br + static_cast<int>((kRingDr >> (2u * nc)) & 3u) - 1;
        col = bc + static_cast<int>((kRingDc >> (2u * nc)) & 3u) - 1;
        rc = nc;
        ++k;
        const uint32_t hr = static_cast<uint32_t>(row - win_r0), hc = static_cast<uint32_t>(col - win_c0);
        if (hr < static_cast<uint32_t>(kWinRows) && hc < static_cast<uint32_t>(kWinCols)) {
            atomicAdd(&s_win[hr * kWinCols + hc], 1u);
        } else {
            atomicAdd(&a.hist[static_cast<uint32_t>(row) * ucols + static_cast<uint32_t>(col)], 1u);
            ++win_stray;
        }
    };

    int it = 0;
    uint32_t n_pairs = 0, n_slow = 0;                         // wave-uniform
    const unsigned long long t_begin = a.debug_roam ? __builtin_amdgcn_s_memtime() : 0ull;
    auto one_pair = [&]() __attribute__((always_inline)) {
        const uint4 w4 = philox_block_b3(a.seed, track, static_cast<unsigned long long>(static_cast<uint32_t>(blk0 + (it >> 1))));
        const uint32_t ufa = w4.x >> 16, ufb = w4.z >> 16;
        const bool st_a = static_cast<uint32_t>(it - rel) < span, st_b = static_cast<uint32_t>(it + 1 - rel) < span;
        // (copies into locals before any select: see k_step_thr on selects of captured variables)
        const uint32_t q0 = E.x, qa = E.y, qb = E.z, qc = E.w;
        const uint32_t rc0 = rc;
        const uint32_t rc04 = 4u * rc0;                                         // (before the entry is there)
        // What stands between an entry's arrival and the next gather costs ~4.4 clocks per instruction, whatever
        // its depth (profiles/r03_notes.md section 2), so the way to the next state is as short as it gets: which
        // of the three intervals the uniform fell into (two compares), the ring position that leads to from a
        // packed constant (selected by the compares, 3 bits at 3 x last move), the displacement as signed 2-bit
        // fields, the cell as a linear index.  Window coordinates, the visits and the special tests follow the gather.
        // ---- first move (a reversal row is an ordinary row of the move along the heading with the prior's thresholds)
        const bool rva = REV && q0 == kThrReversal;
        const uint32_t eua = rva ? rev_e : q0;
        const uint32_t t4a = rva ? rev_rc4 : rc04;
        const bool ga1 = ufa >= (eua & 0xFFFFu), ga2 = ufa >= (eua >> 16);       // sel = ga1 + ga2 (T1 <= T2 unless flagged)
        const uint32_t nxa = ga2 ? kRingNext2 : (ga1 ? kRingNext1 : kRingNext0);   // (constants: a select of captured variables would be a select of their addresses)
        const uint32_t nca = (nxa >> t4a) & 7u, nca4 = 4u * nca;
        const uint32_t q1 = ga2 ? qc : (ga1 ? qb : qa);                          // the entry of the cell it leads to
        const int dra = __builtin_amdgcn_sbfe(static_cast<int>(kRingDrS), nca4, 2u), dca = __builtin_amdgcn_sbfe(static_cast<int>(kRingDcS), nca4, 2u);
        const int cell_a = lcell + __mul24(dra, a.cols) + dca;
        // ---- second move
        const bool rvb = REV && q1 == kThrReversal;
        const uint32_t eub = rvb ? rev_e : q1;
        const uint32_t t4b = rvb ? rev_rc4 : nca4;
        const bool gb1 = ufb >= (eub & 0xFFFFu), gb2 = ufb >= (eub >> 16);
        const uint32_t nxb = gb2 ? kRingNext2 : (gb1 ? kRingNext1 : kRingNext0);
        const uint32_t ncb = (nxb >> t4b) & 7u, ncb4 = 4u * ncb;
        const int drb = __builtin_amdgcn_sbfe(static_cast<int>(kRingDrS), ncb4, 2u), dcb = __builtin_amdgcn_sbfe(static_cast<int>(kRingDcS), ncb4, 2u);
        const int cell_b = cell_a + __mul24(drb, a.cols) + dcb;
        // the next entry, before anything else is known.  A lane that takes the pair from the table ends on
        // a cell of the raster; one that does not (it stands on a boundary cell, a flag entry, a near-tie)
        // may point up to two rows outside it -- the index is clamped into the table (a negative one wraps to
        // a large unsigned) and the lane looks its state up again below, so what it loads here is never used;
        // nor is what a lane loads that does not step (it holds a cell of the raster all the same)
        {
            const uint32_t cb = static_cast<uint32_t>(cell_b) < last_cell ? static_cast<uint32_t>(cell_b) : last_cell;
            E = *reinterpret_cast<const uint4 *>(pair + ((cb << 7) | (ncb4 << 2)));
        }
        // ---- behind the gather: window coordinates of the two visits, the tests
        const int wra = wr + dra, wca = wc + dca, wrb = wra + drb, wcb = wca + dcb;
        const int32_t a1 = static_cast<int32_t>(ufa) - static_cast<int32_t>(eua & 0xFFFFu);
        const int32_t a2 = static_cast<int32_t>(ufa) - static_cast<int32_t>(eua >> 16);
        const int32_t b1 = static_cast<int32_t>(ufb) - static_cast<int32_t>(eub & 0xFFFFu);
        const int32_t b2 = static_cast<int32_t>(ufb) - static_cast<int32_t>(eub >> 16);
        // ufi - T in {-1, 0}: within rounding of a boundary; T1 > T2 (d1 < d2): a flag entry
        const bool special = (static_cast<uint32_t>(a1 + 1) < 2u) | (static_cast<uint32_t>(a2 + 1) < 2u) | (a1 < a2) |
                             (static_cast<uint32_t>(b1 + 1) < 2u) | (static_cast<uint32_t>(b2 + 1) < 2u) | (b1 < b2);
        const bool go = fast & st_a & st_b & !special;
        // visits: into the block's histogram window in LDS (no branch: a lane that does not go, or whose visit
        // falls outside the window, adds 0 to cell 0); a visit outside the window is a global atomic and a stray
        const bool in_a = go & (static_cast<uint32_t>(wra) < static_cast<uint32_t>(kWinRows)) & (static_cast<uint32_t>(wca) < static_cast<uint32_t>(kWinCols));
        const bool in_b = go & (static_cast<uint32_t>(wrb) < static_cast<uint32_t>(kWinRows)) & (static_cast<uint32_t>(wcb) < static_cast<uint32_t>(kWinCols));
        // the two rare continuations (a visit outside the window; a lane on the single-move sequence) behind ONE
        // scalar test whose mask is formed early: a v_cmp -> s_cbranch_vccnz that is not taken costs a lone wave
        // ~30 clocks (tools/microbench/latency.hip), an s_cmp on a mask that has been in SGPRs for a while next to none
        const bool out = go & !(in_a & in_b);
        const bool slow = (st_a | st_b) & !go;
        const unsigned long long rare = __ballot(out | slow);
        atomicAdd(&s_win[in_a ? (wra << 8) + wca : 0], in_a ? 1u : 0u);        // ds_add_u32, nothing returned
        atomicAdd(&s_win[in_b ? (wrb << 8) + wcb : 0], in_b ? 1u : 0u);
        k += go ? 2 : 0;
        wr = go ? wrb : wr;
        wc = go ? wcb : wc;
        lcell = go ? cell_b : lcell;
        rc = go ? ncb : rc0;
        ++n_pairs;
        if (__builtin_expect(rare != 0ull, 0)) {
            if (__any(out)) {
#ifdef SSRS_PROBE_NO_STRAY_ATOMICS      // timing probe (wrong histogram): what the strays' global atomics cost a pass
                if (go && !in_a) ++win_stray;
                if (go && !in_b) ++win_stray;
#else
                if (go && !in_a) { atomicAdd(&a.hist[cell_a], 1u); ++win_stray; }
                if (go && !in_b) { atomicAdd(&a.hist[cell_b], 1u); ++win_stray; }
#endif
            }
            if (__any(slow)) {
                ++n_slow;
                if (slow) {
                    if (fast) { row = win_r0 + wr; col = win_c0 + wc; }       // the state the lane stands on, as a cell again
#pragma unroll 1
                    for (int h = 0; h < 2; ++h)
                        if (static_cast<uint32_t>(it + h - rel) < span) slow_step(h ? w4.z : w4.x, h ? w4.w : w4.y);
                    enter(true);
                }
            }
        }
        it += 2;
    };
    // A launch ends when its FIRST wave is through its steps: that wave raises the flag (launch + 1: the control
    // block is zeroed per call and launches count up), every other wave whose tracks are all released leaves at
    // its next trip.  A released track's Philox block index follows from its own k, so nothing ties the waves'
    // iteration counts together -- and a wave that is slower (a track on its way
    // out of its basin pays a global atomic per visit outside the block's window: +257 clocks per pair,
    // profiles/r03_roam_waves.txt) no longer holds the other 900 back for the 12 % it is behind.
    const uint32_t stop_stamp = static_cast<uint32_t>(a.launch) + 1u;
    uint32_t stop_seen = 0u;
    bool stopped = false;
    for (; it + 8 <= a.steps; ) {
        // (one scalar test per trip; the flag as it was one trip ago: its load is never waited for on its own)
        // (a wave with a lane that still waits for its release runs the launch out: the host counts a.steps
        // iterations for it, and a track released one launch late would start on the odd half of a Philox block)
        const unsigned long long stepping = __ballot(static_cast<uint32_t>(it - rel) < span), waiting = __ballot(it < rel && span != 0u);
        const bool stop_now = a.roam_stop && waiting == 0ull &&
                              static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(stop_seen))) == stop_stamp;
        if (((stepping | waiting) == 0ull) | stop_now) { stopped = stop_now; break; }
        if (a.roam_stop) stop_seen = __hip_atomic_load(&ctl->roam_stop, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
        for (int u = 0; u < 4; ++u) one_pair();
    }
    for (; it < a.steps && !stopped; ) {                     // a.steps is even (host)
        if (!__any(static_cast<uint32_t>(it - rel) < span || (it < rel && span != 0u))) break;
        one_pair();
    }
    if (a.roam_stop && !stopped && it >= a.steps && (threadIdx.x & 63) == 0) atomicMax(&ctl->roam_stop, stop_stamp);
    if (fast) { row = win_r0 + wr; col = win_c0 + wc; }
    {
        uint32_t nf = n_fine;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) nf += __shfl_down(nf, off);
        if ((threadIdx.x & 63) == 0 && nf) atomicAdd(&ctl->pad, nf);          // near-ties the fine table settled
    }
    if ((threadIdx.x & 63) == 0 && n_pairs) {
        atomicAdd(&ctl->roam_pairs, static_cast<unsigned long long>(n_pairs));
        if (n_slow) atomicAdd(&ctl->roam_slow, static_cast<unsigned long long>(n_slow));
        if (a.debug_roam) {
            const unsigned long long dt = __builtin_amdgcn_s_memtime() - t_begin;
            atomicAdd(&ctl->dbg_tsum, dt);
            atomicMax(&ctl->dbg_tmax, dt);
            atomicAdd(&ctl->dbg_waves, 1ull);
            atomicMax(&ctl->dbg_slowmax, (static_cast<unsigned long long>(n_slow) << 32) | n_pairs);
        }
    }
#ifdef SSRS_DEBUG_WAVE_DUMP
    // (diagnostic build only, tools/dev/r03_wave_dump.sh: one record per wave of ONE launch)
    if (a.debug_roam && a.launch == SSRS_DEBUG_WAVE_DUMP && a.dbg_buf) {
        const unsigned long long dt = __builtin_amdgcn_s_memtime() - t_begin;
        const int lanes = __popcll(__ballot(live0));
        uint32_t wst = win_stray;
        for (int off = 32; off > 0; off >>= 1) wst += __shfl_down(wst, off);
        if ((threadIdx.x & 63) == 0) {
            unsigned long long *r = a.dbg_buf + 8ull * (blockIdx.x * (BT / 64) + (threadIdx.x >> 6));
            r[0] = dt; r[1] = static_cast<unsigned long long>(lanes); r[2] = n_pairs; r[3] = n_slow; r[4] = wst;
            r[5] = static_cast<unsigned long long>(win_r0); r[6] = static_cast<unsigned long long>(win_c0);
            r[7] = static_cast<uint32_t>(__popcll(__ballot(fast)));
        }
    }
#endif
    __syncthreads();
    for (int q = threadIdx.x; q < kWinRows * kWinCols; q += BT) {
        const uint32_t n = s_win[q];
        if (n) atomicAdd(&a.hist[static_cast<uint32_t>(win_r0 + q / kWinCols) * ucols + static_cast<uint32_t>(win_c0 + q % kWinCols)], n);
    }
    {
        unsigned long long ws = win_stray;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) ws += __shfl_down(ws, off);
        if ((threadIdx.x & 63) == 0 && ws) atomicAdd(&ctl->strays, ws);
    }
    const bool active = live0 && span != 0u && k < static_cast<int>(a.max_k);
    if (live0 && !active) {
        if (a.lengths) a.lengths[t] = static_cast<int32_t>(k + 1);
        if (a.end_rc) reinterpret_cast<uint32_t *>(a.end_rc)[t] = (static_cast<uint32_t>(row) & 0xFFFFu) | (static_cast<uint32_t>(col) << 16);
    }
    // one reservation per block, every lane keeps its slot, the dead leave tombstones (k_step_thr<6>)
    __shared__ uint32_t s_surv[BT / 64 + 1];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int nsurv = __popcll(__ballot(active));
    if (lane == 0) s_surv[wv] = static_cast<uint32_t>(nsurv);
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t tot = 0;
        for (int q = 0; q < BT / 64; ++q) tot += s_surv[q];
        s_surv[BT / 64] = tot ? atomicAdd(&ctl->count[out_slot][xcd], static_cast<uint32_t>(BT)) : 0xFFFFFFFFu;
    }
    __syncthreads();
    const uint32_t basei = s_surv[BT / 64];
    if (basei != 0xFFFFFFFFu) {
        a.list_out[xcd * a.cap + basei + threadIdx.x] = active ? t : -1;
        if (active) {
            TrackState o;
            o.pos = static_cast<int32_t>((static_cast<uint32_t>(row) & 0xFFFFu) | (static_cast<uint32_t>(col) << 16));
            o.k = k;
            o.dirs = static_cast<uint32_t>(kKOfRing >> (4 * rc)) & 0xFu;
            o.aux = s.aux;
            a.state[t] = o;
        }
    }
    unsigned long long mv = live0 ? static_cast<unsigned long long>(k - s.k) : 0ull;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) mv += __shfl_down(mv, off);
    if (lane == 0 && mv) atomicAdd(&ctl->steps, mv);
}

// K3 binning: one block per step of the launch.  The coherent schedule keeps the
// whole batch on a front a few raster rows deep, so one step's visits fall into
// a window of a few rows: counted with LDS atomics, flushed with contiguous
// (full-rate) global atomics of the non-zero cells; stragglers outside the
// window fall back to single global atomics.
constexpr int kBinThreads = 1024;
constexpr int kBinCells = 30720;          // 120 KB of LDS
__global__ __launch_bounds__(kBinThreads) void k_bin_visits(const uint32_t *__restrict__ visits,
                                                           long long stride,
                                                           TrackCtl *__restrict__ ctl, int slot,
                                                           uint32_t *__restrict__ hist, int rows,
                                                           int cols, uint32_t cap)
{
    __shared__ uint32_t bins[kBinCells];
    __shared__ uint32_t s_first;
    uint32_t nslots[kXcd];                             // whole waves wrote their slots
#pragma unroll
    for (int x = 0; x < kXcd; ++x) nslots[x] = (ctl->count[slot][x] + 63u) & ~63u;
    const uint32_t *v = visits + static_cast<long long>(blockIdx.x) * stride;
    const uint32_t ncell = static_cast<uint32_t>(rows) * cols;
    const int wrows = kBinCells / cols < 1 ? 0 : (kBinCells / cols > rows ? rows : kBinCells / cols);
    if (threadIdx.x == 0) s_first = 0xFFFFFFFFu;
    for (int k = threadIdx.x; k < wrows * cols; k += kBinThreads) bins[k] = 0;
    __syncthreads();
    // window origin: the row of the smallest visited cell among a SAMPLE of the
    // slots (4 per thread: the first 512 of every list).  A full pass would read the
    // step's visits twice; visits below the sampled origin simply count as strays.
    uint32_t mn = 0xFFFFFFFFu;
    {
        const uint32_t x = threadIdx.x % kXcd;
        uint32_t nx = 0;
#pragma unroll
        for (int y = 0; y < kXcd; ++y) nx = (static_cast<uint32_t>(y) == x) ? nslots[y] : nx;
        for (uint32_t q = 0, j = threadIdx.x / kXcd; q < 4 && j < nx; ++q, j += kBinThreads / kXcd) {
            const uint32_t c = v[x * cap + j];
            mn = c < mn ? c : mn;
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const uint32_t o = __shfl_down(mn, off);
        mn = o < mn ? o : mn;
    }
    if ((threadIdx.x & 63) == 0 && mn != 0xFFFFFFFFu) atomicMin(&s_first, mn);
    __syncthreads();
    if (s_first == 0xFFFFFFFFu) {
        // the sample saw only idle slots (sparse late launches): look at all of them
        __syncthreads();
        mn = 0xFFFFFFFFu;
#pragma unroll
        for (int x = 0; x < kXcd; ++x)
            for (uint32_t j = threadIdx.x; j < nslots[x]; j += kBinThreads) {
                const uint32_t c = v[x * cap + j];
                mn = c < mn ? c : mn;
            }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const uint32_t o = __shfl_down(mn, off);
            mn = o < mn ? o : mn;
        }
        if ((threadIdx.x & 63) == 0 && mn != 0xFFFFFFFFu) atomicMin(&s_first, mn);
        __syncthreads();
    }
    const uint32_t first = s_first;
    if (first == 0xFFFFFFFFu) return;                  // nobody moved in this step
    // first cell of the window: one row below the sampled minimum when the window
    // has room, so that a slightly lagging track is still binned
    uint32_t brow = first / cols;
    if (wrows >= 3 && brow > 0) --brow;
    const uint32_t base = brow * cols;
    const uint32_t wcells = static_cast<uint32_t>(wrows) * cols;
    uint32_t stray = 0;
    // sixteen loads in flight per thread: the loop is bound by the latency of the visit
    // buffer (just written by the stepper), not by the LDS atomics
    constexpr int kU = 16;
#pragma unroll
    for (int x = 0; x < kXcd; ++x)
        for (uint32_t j = threadIdx.x; j < nslots[x]; j += kBinThreads * kU) {
            uint32_t c[kU];
#pragma unroll
            for (int q = 0; q < kU; ++q) {
                const uint32_t jj = j + q * kBinThreads;
                c[q] = jj < nslots[x] ? v[x * cap + jj] : 0xFFFFFFFFu;
            }
#pragma unroll
            for (int q = 0; q < kU; ++q) {
                if (c[q] >= ncell) continue;           // idle slot
                const uint32_t off = c[q] - base;
                if (off < wcells) atomicAdd(&bins[off], 1u);
                else { atomicAdd(&hist[c[q]], 1u); ++stray; }
            }
        }
    // stray count -> host: when the batch no longer moves as a front (tracks
    // trapped or scattered) the host switches back to in-stepper atomics
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) stray += __shfl_down(stray, off);
    if ((threadIdx.x & 63) == 0 && stray) atomicAdd(&ctl->strays, static_cast<unsigned long long>(stray));
    __syncthreads();
    for (uint32_t k = threadIdx.x; k < wcells; k += kBinThreads * 4) {
        uint32_t n[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const uint32_t kk = k + q * kBinThreads;
            n[q] = (kk < wcells && base + kk < ncell) ? bins[kk] : 0u;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q)
            if (n[q]) atomicAdd(&hist[base + k + q * kBinThreads], n[q]);
    }
}


// K3 for 16-bit visit keys (k_step_thr<4>): block b bins iterations 2b and 2b + 1 of the launch into
// ONE LDS window; a key is the visit's offset from cell (first start row + global iteration - 1, 0),
// so the window is simply the first kBinCells offsets from the first iteration's base (the front's
// rows sit at offsets [2 cols, 4 cols), one row further for the second iteration).  Two iterations
// per block halve the zeroing / flushing of the 120-KB window, the larger part of this kernel.
__global__ __launch_bounds__(kBinThreads) void k_bin_visits16(const uint16_t *__restrict__ visits, long long stride, int steps,
                                                             TrackCtl *__restrict__ ctl, int slot,
                                                             uint32_t *__restrict__ hist, int rows, int cols, uint32_t cap,
                                                             int v16_offset, long long it_base, uint32_t *host_out, int host_words)
{
    __shared__ uint32_t bins[kBinCells];
    uint32_t nslots[kXcd];
#pragma unroll
    for (int x = 0; x < kXcd; ++x) nslots[x] = (ctl->count[slot][x] + 63u) & ~63u;
    const int it0 = 2 * static_cast<int>(blockIdx.x);
    const long long base = (static_cast<long long>(ctl->par_min) - v16_offset + it_base + it0 - 1) * cols;
    const long long ncell = static_cast<long long>(rows) * cols;
    for (int k = threadIdx.x; k < kBinCells; k += kBinThreads) bins[k] = 0;
    __syncthreads();
    uint32_t stray = 0;
    constexpr int kU = 8;                                  // 8 x 2 keys in flight per thread
    for (int sub = 0; sub < 2 && it0 + sub < steps; ++sub) {
        const uint16_t *v = visits + static_cast<long long>(it0 + sub) * stride;
        const uint32_t shift = sub ? static_cast<uint32_t>(cols) : 0u;       // the second iteration's base is one row up
#pragma unroll
        for (int x = 0; x < kXcd; ++x) {
            const uint32_t *v2 = reinterpret_cast<const uint32_t *>(v + static_cast<size_t>(x) * cap);   // cap is a multiple of 256
            const uint32_t npair = nslots[x] / 2;
            for (uint32_t j = threadIdx.x; j < npair; j += kBinThreads * kU) {
                uint32_t c[kU];
#pragma unroll
                for (int q = 0; q < kU; ++q) {
                    const uint32_t jj = j + q * kBinThreads;
#ifdef SSRS_PROBE_K3_NOREAD
                    c[q] = jj < npair ? (jj * 2654435761u) % 24000u * 65537u : 0xFFFFFFFFu;
#else
                    c[q] = jj < npair ? v2[jj] : 0xFFFFFFFFu;
#endif
                }
#pragma unroll
                for (int q = 0; q < kU; ++q) {
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        uint32_t key = h ? c[q] >> 16 : c[q] & 0xFFFFu;
                        if (key == 0xFFFFu) continue;      // idle slot (or counted by the stepper)
                        key += shift;
                        if (key < kBinCells) atomicAdd(&bins[key], 1u);
                        else if (base + key >= 0 && base + key < ncell) { atomicAdd(&hist[base + key], 1u); ++stray; }   // a visited cell
                    }
                }
            }
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) stray += __shfl_down(stray, off);
    if ((threadIdx.x & 63) == 0 && stray) {
        // (with a read-back the count must have landed before this block reports itself done: take the
        // atomic's return value, i.e. wait for THIS one -- a fence at the end would also wait for the
        // flush below to drain, 70 -> 175 us per launch)
        const unsigned long long before = atomicAdd(&ctl->strays, static_cast<unsigned long long>(stray));
        if (host_out && before == 0xFFFFFFFFFFFFFFFFull) hist[0] += 0u;      // keeps the returning form
    }
    __syncthreads();
    for (uint32_t k = threadIdx.x; k < kBinCells; k += kBinThreads * 4) {
        uint32_t n[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const uint32_t kk = k + q * kBinThreads;
            n[q] = kk < kBinCells ? bins[kk] : 0u;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q)
#ifdef SSRS_PROBE_K3_NOFLUSH
            if (n[q] == 0x12345678u)
#else
            if (n[q])
#endif
                atomicAdd(&hist[base + k + q * kBinThreads], n[q]);           // n > 0: a visited cell, inside the raster
    }
    // The batch's read-back (live counts, steps, strays: the head of the control block), written to the
    // host's pinned slot by the last block of the batch's last binning kernel: a copy on the stream
    // would cost it a blit kernel and two more dependency gaps (~16 us per launch, profiles/r02_notes.md)
    if (host_out) {
        __shared__ bool s_last;
        __syncthreads();
        if (threadIdx.x == 0) s_last = atomicAdd(&ctl->bin_done, 1u) == gridDim.x - 1u;
        __syncthreads();
        if (s_last) {
            // (no system-scope fence: it would write the whole L2 back, ~100 us; the slot is fine-grained
            // host memory, stored to directly, and the host reads it after the stream's event)
            if (static_cast<int>(threadIdx.x) < host_words)
                __hip_atomic_store(host_out + threadIdx.x,
                                   __hip_atomic_load(reinterpret_cast<const uint32_t *>(ctl) + threadIdx.x, __ATOMIC_RELAXED,
                                                     __HIP_MEMORY_SCOPE_AGENT),
                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            if (threadIdx.x == 0) ctl->bin_done = 0u;
        }
    }
}

// K3 for batches that no longer move as a front.  With real potential fields long
// tracks circle in pockets of the field, many tracks in the same pockets: 9e9 visits fell
// on 1.3e7 distinct cells (one cell: 4.6e7) in a C2 run, and the per-step global atomics
// -- same address, hence serialised at the memory side -- made the stepper seven times
// slower than without a histogram (a track itself revisits little: 353 distinct cells
// per 512 steps, so per-track de-duplication was measured and gained nothing).  What
// helps is privatisation: wave w counts into copy w mod K of the histogram (K copies
// in the caller's workspace when it is large enough, ssrs_tracks_workspace_bytes_ex);
// the copies are added to `hist` once at the end.  5.8 -> 14 (K = 16) / 17.5 (K = 64)
// G steps/s on the C2 run.
// hist64 += hist32, hist32 = 0 (ssrs_tracks_simulate_h64: a roaming batch's trap cells pass 2^32 visits from ~250 000 tracks
// on: the 32-bit raster the kernels count into is emptied into the caller's 64-bit one every other batch and at the end)
__global__ __launch_bounds__(kBlock) void k_drain64(uint32_t *__restrict__ lo, unsigned long long *__restrict__ acc, size_t n)
{
    for (size_t i = blockIdx.x * static_cast<size_t>(kBlock) + threadIdx.x; i < n; i += static_cast<size_t>(gridDim.x) * kBlock) {
        const uint32_t v = lo[i];
        if (v) { acc[i] += v; lo[i] = 0u; }
    }
}

__global__ __launch_bounds__(kBlock) void k_fold_copies(const uint32_t *__restrict__ copies, int ncopies,
                                                       size_t ncell, uint32_t *__restrict__ hist)
{
    for (size_t i = blockIdx.x * static_cast<size_t>(kBlock) + threadIdx.x; i < ncell;
         i += static_cast<size_t>(gridDim.x) * kBlock) {
        uint32_t s = 0;
        for (int c = 0; c < ncopies; ++c) s += copies[static_cast<size_t>(c) * ncell + i];
        if (s) hist[i] += s;
    }
}

// K3 for oblique headings and for batches that wander.  An oblique front diffuses along
// BOTH raster axes (heading 45 deg:
// the moves N, NE, E advance 1, 1, 0 rows): a step's visits span hundreds of rows, no
// per-step row window holds them, and per-visit atomics run at the memory side's ~2e10/s
// (27 ms per 100k tracks at C2 against 5 ms for axis-aligned headings).  But over one
// launch the region the front sweeps receives ~8 visits per cell.  So the launch's visits
// are bucketed by raster TILE (kTileRows x kTileCols cells = 30 720 16-bit LDS counters):
//   k_tile_sort<false>  counts visits per tile        (one read of the visit buffer)
//   k_tile_scan         bucket starts
//   k_tile_sort<true>   copies visits into buckets    (one read, one write)
//   k_bin_bucket        one block per 65 536 visits of a tile: LDS counters, row-wise flush
// 0.8 GB of traffic per launch instead of 5e7 memory-side atomics.  Tracks that circle in
// pockets of a solved potential field (no front at all) concentrate their visits even
// more: such batches move here from the row window as well (19 -> 27 G steps/s).  The order inside a
// bucket is arbitrary; a histogram does not care.  A counter that reaches 0x8000 is
// emptied by the one thread that saw it (LDS atomics return the old value; at most
// 1023 x 16 other increments can land in between, so 16 bits never overflow).
constexpr int kTileThreads = 1024;
constexpr int kTileRows = 30, kTileCols = 1024;              // 30 720 counters, 60 KB: two blocks per CU
constexpr int kTilesMax = 4096;                              // tiles per raster (LDS count array)
constexpr uint32_t kItemVisits = 65536;                       // visits per k_bin_bucket block


__device__ __forceinline__ uint32_t tile_of(uint32_t c, uint32_t cols, double inv_cols, uint32_t ntc)
{
    uint32_t r, cc;
    split_cell(c, cols, inv_cols, r, cc);
    return (r / kTileRows) * ntc + cc / kTileCols;
}

// block (x, b; y): slots [256 b, 256 b + 256) of list x, the y-th of kStepSplit runs of
// steps (the kernel is bound by load latency: 100k lanes alone cannot cover it).  These
// are the same 256 tracks step after step, so a block meets a handful of tiles.
constexpr int kStepSplit = 16;
template <bool SCATTER>
__global__ __launch_bounds__(kBlock) void k_tile_sort(const uint32_t *__restrict__ visits, long long stride, int steps,
                                                     const TrackCtl *__restrict__ ctl, int slot, uint32_t cols,
                                                     double inv_cols, uint32_t ncell, uint32_t cap, uint32_t ntc,
                                                     uint32_t ntiles, uint32_t *__restrict__ tile_count,
                                                     uint32_t *__restrict__ tile_cursor, uint32_t *__restrict__ bucket)
{
    __shared__ uint32_t cnt[kTilesMax];
    __shared__ uint32_t base[SCATTER ? kTilesMax : 1];
    const uint32_t x = blockIdx.x % kXcd, j = (blockIdx.x / kXcd) * kBlock + threadIdx.x;
    const uint32_t nslots = (ctl->count[slot][x] + 63u) & ~63u;
    if ((blockIdx.x / kXcd) * kBlock >= nslots) return;           // whole block idle
    for (uint32_t t = threadIdx.x; t < ntiles; t += kBlock) cnt[t] = 0;
    __syncthreads();
    const uint32_t *v = visits + static_cast<size_t>(x) * cap + j;
    constexpr int kU = 8;
    const int chunk = ((steps + kStepSplit - 1) / kStepSplit + kU - 1) / kU * kU;
    const int it0 = static_cast<int>(blockIdx.y) * chunk;
    steps = steps < it0 + chunk ? steps : it0 + chunk;
    if (!SCATTER) {
        if (j < nslots) {
            // a track stays in one tile for many steps: count runs, one LDS atomic per run
            uint32_t cur = 0xFFFFFFFFu, run = 0;
            for (int it = it0; it < steps; it += kU) {
                uint32_t c[kU];
#pragma unroll
                for (int q = 0; q < kU; ++q) c[q] = it + q < steps ? v[static_cast<long long>(it + q) * stride] : 0xFFFFFFFFu;
#pragma unroll
                for (int q = 0; q < kU; ++q) {
                    if (c[q] >= ncell) continue;
                    const uint32_t t = tile_of(c[q], cols, inv_cols, ntc);
                    if (t != cur) {
                        if (run) atomicAdd(&cnt[cur], run);
                        cur = t; run = 0;
                    }
                    ++run;
                }
            }
            if (run) atomicAdd(&cnt[cur], run);
        }
        __syncthreads();
        for (uint32_t t = threadIdx.x; t < ntiles; t += kBlock)
            if (cnt[t]) atomicAdd(&tile_count[t], cnt[t]);
        return;
    }
    // SCATTER: count again (the reads hit L2 the second time), reserve the block's share of
    // every bucket it feeds, copy.  (Keeping the run's 64 visits in registers between the two
    // sweeps was measured slower, 117 -> 153 us: the unrolled body no longer fits the I-cache.)
    if (j < nslots) {
        uint32_t cur = 0xFFFFFFFFu, run = 0;
        for (int it = it0; it < steps; it += kU) {
            uint32_t c[kU];
#pragma unroll
            for (int q = 0; q < kU; ++q) c[q] = it + q < steps ? v[static_cast<long long>(it + q) * stride] : 0xFFFFFFFFu;
#pragma unroll
            for (int q = 0; q < kU; ++q) {
                if (c[q] >= ncell) continue;
                const uint32_t t = tile_of(c[q], cols, inv_cols, ntc);
                if (t != cur) {
                    if (run) atomicAdd(&cnt[cur], run);
                    cur = t; run = 0;
                }
                ++run;
            }
        }
        if (run) atomicAdd(&cnt[cur], run);
    }
    __syncthreads();
    for (uint32_t t = threadIdx.x; t < ntiles; t += kBlock) {
        const uint32_t n = cnt[t];
        if (n) { base[t] = atomicAdd(&tile_cursor[t], n); cnt[t] = 0; }
    }
    __syncthreads();
    if (j >= nslots) return;                                       // whole waves: nslots is a multiple of 64
    const uint32_t lane = threadIdx.x & 63u;
    for (int it = it0; it < steps; it += kU) {
        uint32_t c[kU];
#pragma unroll
        for (int q = 0; q < kU; ++q) c[q] = it + q < steps ? v[static_cast<long long>(it + q) * stride] : 0xFFFFFFFFu;
#pragma unroll
        for (int q = 0; q < kU; ++q) {
            // a wave's 64 tracks are neighbours: nearly always one tile, then one LDS
            // atomic for the wave instead of 64 on the same address
            const bool active = c[q] < ncell;
            const uint32_t t = active ? tile_of(c[q], cols, inv_cols, ntc) : 0u;
            const unsigned long long act = __ballot(active);
            if (act == 0) continue;
            const int leader = __ffsll(static_cast<long long>(act)) - 1;
            const uint32_t t0 = __shfl(t, leader);
            if (__ballot(active && t == t0) == act) {
                uint32_t first = 0;
                if (lane == static_cast<uint32_t>(leader)) first = atomicAdd(&cnt[t0], static_cast<uint32_t>(__popcll(act)));
                first = __shfl(first, leader);
                if (active) bucket[base[t0] + first + __popcll(act & ((1ull << lane) - 1ull))] = c[q];
            } else if (active) {
                bucket[base[t] + atomicAdd(&cnt[t], 1u)] = c[q];
            }
        }
    }
}

// exclusive scan of the tile counts (<= 4096): bucket starts; the cursors start there
__global__ __launch_bounds__(kTileThreads) void k_tile_scan(const uint32_t *__restrict__ tile_count, uint32_t ntiles,
                                                           uint32_t *__restrict__ tile_start,
                                                           uint32_t *__restrict__ tile_cursor,
                                                           uint32_t *__restrict__ item_start)
{
    using Scan = hipcub::BlockScan<uint32_t, kTileThreads>;
    __shared__ typename Scan::TempStorage tmp;
    constexpr int kPer = kTilesMax / kTileThreads;
    uint32_t n[kPer], sum = 0, items = 0, prefix, iprefix;
#pragma unroll
    for (int q = 0; q < kPer; ++q) {
        const uint32_t t = threadIdx.x * kPer + q;
        n[q] = t < ntiles ? tile_count[t] : 0u;
        sum += n[q];
        items += (n[q] + kItemVisits - 1) / kItemVisits;
    }
    Scan(tmp).ExclusiveSum(sum, prefix);
    __syncthreads();
    Scan(tmp).ExclusiveSum(items, iprefix);
#pragma unroll
    for (int q = 0; q < kPer; ++q) {
        const uint32_t t = threadIdx.x * kPer + q;
        if (t < ntiles) { tile_start[t] = prefix; tile_cursor[t] = prefix; item_start[t] = iprefix; }
        prefix += n[q];
        iprefix += (n[q] + kItemVisits - 1) / kItemVisits;
    }
    if (threadIdx.x == kTileThreads - 1) item_start[ntiles] = iprefix;     // the last thread holds the total
}

__global__ __launch_bounds__(kTileThreads) void k_bin_bucket(const uint32_t *__restrict__ bucket,
                                                            const uint32_t *__restrict__ tile_start,
                                                            const uint32_t *__restrict__ tile_count,
                                                            TrackCtl *__restrict__ ctl, uint32_t *__restrict__ hist,
                                                            uint32_t rows, uint32_t cols, double inv_cols, uint32_t ntc,
                                                            uint32_t ntiles, const uint32_t *__restrict__ item_start)
{
    __shared__ uint32_t bins[kTileRows * kTileCols / 2];
    // work item = up to kItemVisits visits of one tile: busy tiles (the front's densest rows,
    // the pockets wandering tracks circle in) are shared by as many blocks as they need,
    // each with its own counters.  Block -> tile by bisection of the item starts.
    if (blockIdx.x >= item_start[ntiles]) return;
    uint32_t tile = 0;
    for (uint32_t step = 2048; step > 0; step >>= 1)             // kTilesMax = 4096 > tile + step
        if (tile + step < ntiles && item_start[tile + step] <= blockIdx.x) tile += step;
    const uint32_t lo = (blockIdx.x - item_start[tile]) * kItemVisits;
    const uint32_t total = tile_count[tile];
    const uint32_t n = total - lo < kItemVisits ? total - lo : kItemVisits;
    const uint32_t r0 = (tile / ntc) * kTileRows, c0 = (tile % ntc) * kTileCols;
    const uint32_t *b = bucket + tile_start[tile] + lo;
    for (int k = threadIdx.x; k < kTileRows * kTileCols / 2; k += kTileThreads) bins[k] = 0;
    __syncthreads();
    constexpr int kU = 16;
    for (uint32_t i = threadIdx.x; i < n; i += kTileThreads * kU) {
        uint32_t c[kU];
#pragma unroll
        for (int q = 0; q < kU; ++q) {
            const uint32_t ii = i + q * kTileThreads;
            c[q] = ii < n ? b[ii] : 0xFFFFFFFFu;
        }
#pragma unroll
        for (int q = 0; q < kU; ++q) {
            if (c[q] == 0xFFFFFFFFu) continue;
            uint32_t r, cc;
            split_cell(c[q], cols, inv_cols, r, cc);
            const uint32_t idx = (r - r0) * kTileCols + (cc - c0), sh = (idx & 1u) * 16u;
            const uint32_t old = atomicAdd(&bins[idx >> 1], 1u << sh);
            if (((old >> sh) & 0xFFFFu) == 0x8000u) {            // this thread empties the counter
                atomicAdd(&hist[c[q]], 0x8001u);
                atomicSub(&bins[idx >> 1], 0x8001u << sh);
            }
        }
    }
    __syncthreads();
    // flush: consecutive lanes, consecutive cells.  The number of cells flushed tells the
    // host whether bucketing still pays (strays = cells; it stops below 2 visits per cell).
    uint32_t flushed = 0;
    const uint32_t trows = rows - r0 < kTileRows ? rows - r0 : kTileRows;
    const uint32_t tcols = cols - c0 < kTileCols ? cols - c0 : kTileCols;
    for (uint32_t w = threadIdx.x; w < trows * (kTileCols / 2); w += kTileThreads) {
        const uint32_t pair = bins[w];
        if (!pair) continue;
        const uint32_t dr = w / (kTileCols / 2), dc = 2 * (w % (kTileCols / 2));
        uint32_t *h = hist + static_cast<size_t>(r0 + dr) * cols + c0 + dc;
        if (pair & 0xFFFFu) { atomicAdd(h, pair & 0xFFFFu); ++flushed; }
        if ((pair >> 16) && dc + 1 < tcols) { atomicAdd(h + 1, pair >> 16); ++flushed; }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) flushed += __shfl_down(flushed, off);
    if ((threadIdx.x & 63) == 0 && flushed) atomicAdd(&ctl->strays, static_cast<unsigned long long>(flushed));
}

// hist (rows x cols) += transpose of hist_t (cols x rows): the transposed histogram that
// east / west batches bin into (there a step's visits fill a few COLUMNS, which are rows
// of hist_t, so the LDS window of k_bin_visits and its contiguous flush work unchanged)
__global__ __launch_bounds__(kBlock) void k_transpose_add(const uint32_t *__restrict__ hist_t, int rows,
                                                         int cols, uint32_t *__restrict__ hist)
{
    __shared__ uint32_t tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;           // 32 x 8
    const int tiles_c = (cols + 31) / 32;
    const int r0 = (blockIdx.x / tiles_c) * 32, c0 = (blockIdx.x % tiles_c) * 32;
    for (int j = ty; j < 32; j += 8) {                                // read hist_t[c][r], r fastest
        const int c = c0 + j, r = r0 + tx;
        tile[j][tx] = (c < cols && r < rows) ? hist_t[static_cast<size_t>(c) * rows + r] : 0u;
    }
    __syncthreads();
    for (int j = ty; j < 32; j += 8) {                                // write hist[r][c], c fastest
        const int r = r0 + j, c = c0 + tx;
        const uint32_t v = tile[tx][j];
        if (r < rows && c < cols && v) hist[static_cast<size_t>(r) * cols + c] += v;
    }
}


// ------------------------------------------------------------ trajectory record
// Trajectory output without a second simulation pass.  The stepper already writes the
// cell every slot visits at every step of a launch into the visit buffer (the input of
// the binning kernels).  With a recorder each launch gets its OWN region of a caller-
// supplied pool -- [8 list counts][the launch's slot -> track list][visits of S steps] --
// instead of the shared buffer, and once the lengths are final (they give the offsets)
// ssrs_tracks_gather replays the regions in launch order and appends every track's
// visits at its cursor.  4 B written per step while stepping, 4 B read + 4 B written
// in the gather; the ring stepper stays in use (the generic kernel's direct trajectory
// writes needed the lengths of an earlier pass: two full simulations).
struct TrajChunk {
    const uint32_t *counts;      // [kXcd] live slots per list when the launch started
    const int32_t *list;         // [kXcd][vcap] slot -> track, nullptr = identity over `cap`
    const uint32_t *visits;      // [steps][kXcd * vcap]
    uint32_t vcap, cap;
    int steps;
    int transposed;              // visit key = col * rows + row (east / west fronts)
};

// plain per-visit atomics: the histogram of a recorded launch that neither binning path took
__global__ __launch_bounds__(kBlock) void k_count_visits(const uint32_t *__restrict__ visits, uint32_t vcap,
                                                        int steps, const uint32_t *__restrict__ counts,
                                                        uint32_t *__restrict__ hist, uint32_t ncell)
{
    const uint32_t x = blockIdx.x % kXcd, j = (blockIdx.x / kXcd) * kBlock + threadIdx.x;
    const uint32_t nslots = (counts[x] + 63u) & ~63u;
    if (j >= nslots) return;
    const uint32_t *v = visits + static_cast<size_t>(x) * vcap + j;
    const size_t stride = static_cast<size_t>(kXcd) * vcap;
    for (int it = 0; it < steps; ++it) {
        const uint32_t c = v[it * stride];
        if (c < ncell) atomicAdd(&hist[c], 1u);
    }
}

__global__ __launch_bounds__(kBlock) void k_gather_init(const int32_t *__restrict__ start_rc, long long ntracks,
                                                       const long long *__restrict__ off, int16_t *__restrict__ traj,
                                                       uint32_t *__restrict__ cursor)
{
    const long long t = blockIdx.x * static_cast<long long>(kBlock) + threadIdx.x;
    if (t >= ntracks) return;
    cursor[t] = 1;
    if (off[t + 1] - off[t] > 0)
        reinterpret_cast<uint32_t *>(traj)[off[t]] =
            static_cast<uint32_t>(start_rc[2 * t] & 0xFFFF) | (static_cast<uint32_t>(start_rc[2 * t + 1]) << 16);
}

__global__ __launch_bounds__(kBlock) void k_gather_chunk(const TrajChunk ch, uint32_t rows, uint32_t cols,
                                                        const long long *__restrict__ off,
                                                        int16_t *__restrict__ traj, uint32_t *__restrict__ cursor)
{
    const uint32_t x = blockIdx.x % kXcd, j = (blockIdx.x / kXcd) * kBlock + threadIdx.x;
    if (j >= ch.counts[x]) return;               // slots past the live list hold no visits
    const uint32_t t = ch.list ? static_cast<uint32_t>(ch.list[static_cast<size_t>(x) * ch.vcap + j]) : x * ch.cap + j;
    const long long base = off[t];
    const long long room = off[t + 1] - base;    // never write beyond the track's room
    uint32_t *out = reinterpret_cast<uint32_t *>(traj) + base;
    uint32_t c = cursor[t];
    const uint32_t ncell = rows * cols;
    const uint32_t div = ch.transposed ? rows : cols;
    const double inv = 1.0 / static_cast<double>(div);
    const uint32_t *v = ch.visits + static_cast<size_t>(x) * ch.vcap + j;
    const size_t stride = static_cast<size_t>(kXcd) * ch.vcap;
    constexpr int kU = 8;
    for (int it = 0; it < ch.steps; it += kU) {
        uint32_t key[kU];
#pragma unroll
        for (int q = 0; q < kU; ++q) key[q] = it + q < ch.steps ? v[(it + q) * stride] : 0xFFFFFFFFu;
#pragma unroll
        for (int q = 0; q < kU; ++q) {
            if (key[q] >= ncell) continue;
            uint32_t hi, lo;                     // key = hi * div + lo
            split_cell(key[q], div, inv, hi, lo);
            const uint32_t row = ch.transposed ? lo : hi, col = ch.transposed ? hi : lo;
            if (c < room) out[c] = row | (col << 16);
            ++c;
        }
    }
    cursor[t] = c;
}

// Re-deals the live tracks over the kXcd lists (a pseudo-launch between two stepper launches).
// The lists are column bands, dealt to the XCDs; when the survivors of a batch sit in one or two
// bands -- tracks that wander in a few basins of the potential field until max_moves -- one XCD
// steps them all while seven idle, and the launch grid, sized by the longest list, stays large.
__global__ __launch_bounds__(1024) void k_rebalance_lists(const int32_t *__restrict__ list_in, int32_t *__restrict__ list_out,
                                                         TrackCtl *ctl, int in_slot, int out_slot, int zero_slot, uint32_t cap)
{
    __shared__ uint32_t pre[kXcd + 1];
    if (threadIdx.x == 0) {
        uint32_t run = 0;
        for (int x = 0; x < kXcd; ++x) { pre[x] = run; run += ctl->count[in_slot][x]; ctl->count[zero_slot][x] = 0; }
        pre[kXcd] = run;
    }
    __syncthreads();
    const uint32_t total = pre[kXcd];
    for (uint32_t j = threadIdx.x; j < total; j += 1024) {
        int x = 0;
#pragma unroll
        for (int y = 1; y < kXcd; ++y) x += pre[y] <= j ? 1 : 0;
        list_out[(j & (kXcd - 1)) * cap + (j >> 3)] = list_in[static_cast<uint32_t>(x) * cap + (j - pre[x])];
    }
    if (threadIdx.x < kXcd) ctl->count[out_slot][threadIdx.x] = (total + kXcd - 1 - threadIdx.x) / kXcd;
}
static_assert(kXcd == 8, "k_rebalance_lists deals with j & 7 / j >> 3");

// Wandering batches (k_step_thr<6>).  A pseudo-launch between two stepper launches:
//   k_wander_windows  where are the live tracks?  Histogram on a coarse grid (kWinRows / 4 x
//                     kWinCols / 4 cells per bin, in LDS), then greedily the densest window of
//                     4 x 4 bins, kWanderWindows times (each round removes the tracks it covers);
//                     on the solved 10 m field two windows hold 99.96 % of the survivors
//   k_wander_keys     key of a list slot = index of the first window that holds its track (tracks
//                     outside all windows and dead slots sort to the end)
//   hipcub sort       5 bits, the list itself as the values
//   k_deal_sorted     each window's run starts at a multiple of kXcd * kBlock tracks and is dealt
//                     round-robin to the lists, so every block of every list is filled from ONE
//                     window; the gaps are tombstones (track id -1)
// k_step_thr<6> keeps the blocks' slots (it reserves kBlock slots per block with a survivor and
// writes tombstones for its dead), so the sort holds until the host asks for the next one.
__global__ __launch_bounds__(1024) void k_wander_windows(const int32_t *__restrict__ list_in, const TrackState *__restrict__ state,
                                                        TrackCtl *__restrict__ ctl, int in_slot, uint32_t cap,
                                                        int rows, int cols, WanderWindows *__restrict__ out)
{
    __shared__ uint32_t h[kWanderBins];
    __shared__ unsigned long long s_best;
    __shared__ uint32_t s_live;
    const int nbr = (rows + kBinRows - 1) / kBinRows, nbc = (cols + kBinCols - 1) / kBinCols;
    const int nb = nbr * nbc;                                        // <= kWanderBins (host)
    for (int q = threadIdx.x; q < nb; q += 1024) h[q] = 0;
    if (threadIdx.x == 0) s_live = 0u;
    __syncthreads();
    uint32_t mine = 0;
    for (uint32_t i = threadIdx.x; i < cap * kXcd; i += 1024) {
        const uint32_t x = i / cap, il = i - x * cap;
        if (il >= ctl->count[in_slot][x]) continue;
        const int32_t t = list_in[i];
        if (t < 0) continue;
        const int32_t pos = state[t].pos;
        atomicAdd(&h[((pos & 0xFFFF) / kBinRows) * nbc + ((pos >> 16) & 0xFFFF) / kBinCols], 1u);
        ++mine;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) mine += __shfl_down(mine, off);
    if ((threadIdx.x & 63) == 0 && mine) atomicAdd(&s_live, mine);
    __syncthreads();
    if (threadIdx.x == 0) ctl->deal_live = s_live;
    int n = 0;
    for (; n < kWanderWindows; ++n) {
        if (threadIdx.x == 0) s_best = 0ull;
        __syncthreads();
        unsigned long long best = 0ull;
        for (int q = threadIdx.x; q < nb; q += 1024) {
            const int i = q / nbc, j = q - i * nbc;
            uint32_t sum = 0;
            for (int di = 0; di < kWinBinRows && i + di < nbr; ++di)
                for (int dj = 0; dj < 4 && j + dj < nbc; ++dj) sum += h[(i + di) * nbc + j + dj];
            const unsigned long long v = (static_cast<unsigned long long>(sum) << 32) | static_cast<uint32_t>(nb - 1 - q);   // ties: lowest bin
            best = v > best ? v : best;
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const unsigned long long o = __shfl_down(best, off);
            best = o > best ? o : best;
        }
        if ((threadIdx.x & 63) == 0) atomicMax(&s_best, best);
        __syncthreads();
        const unsigned long long win = s_best;
        __syncthreads();
        if ((win >> 32) == 0ull) break;
        const int q = nb - 1 - static_cast<int>(win & 0xFFFFFFFFull), i = q / nbc, j = q - i * nbc;
        if (threadIdx.x < 16) {
            const int di = threadIdx.x >> 2, dj = threadIdx.x & 3;
            if (di < kWinBinRows && i + di < nbr && j + dj < nbc) h[(i + di) * nbc + j + dj] = 0;
        }
        if (threadIdx.x == 0) { out->r0[n] = i * kBinRows; out->c0[n] = j * kBinCols; }
        __syncthreads();
    }
    if (threadIdx.x == 0) out->n = n;
}

__global__ __launch_bounds__(kBlock) void k_wander_keys(const int32_t *__restrict__ list_in, const TrackState *__restrict__ state,
                                                       const TrackCtl *__restrict__ ctl, int in_slot, uint32_t cap,
                                                       const WanderWindows *__restrict__ win, uint32_t *__restrict__ keys, uint32_t salt)
{
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= cap * kXcd) return;
    const uint32_t x = i / cap, il = i - x * cap;
    uint32_t key = kWanderWindows + 1;                               // dead slot
    uint32_t mix = 0;
    if (il < ctl->count[in_slot][x]) {
        const int32_t t = list_in[i];
        if (t >= 0) {
            const int32_t pos = state[t].pos;
            key = static_cast<uint32_t>(wander_window_of(win, win->n, pos & 0xFFFF, (pos >> 16) & 0xFFFF));
            if (key >= static_cast<uint32_t>(win->n)) key = kWanderWindows;   // outside every window
            // salt != 0 (a sort of a settled batch): the tracks of a window in an order of its own every time, so
            // that nobody keeps the same wave-mates -- a wave that carries a track on its way out of the basin
            // is slower, and tracks that stay with it fall behind for good
            uint32_t h = (static_cast<uint32_t>(t) + salt) * 0x9E3779B1u;
            h ^= h >> 15; h *= 0x85EBCA6Bu; h ^= h >> 13;
            mix = salt ? (h & (kWanderMix - 1u)) : 0u;
        }
    }
    keys[i] = key * kWanderMix + mix;
}

__global__ __launch_bounds__(1024) void k_deal_sorted(const uint32_t *__restrict__ sorted_keys, const int32_t *__restrict__ sorted,
                                                     int32_t *__restrict__ list_out, TrackCtl *ctl, int out_slot, int zero_slot, uint32_t cap,
                                                     int contiguous, int width)
{
    // run of key k: [lo[k], lo[k + 1]) in the sorted order, dealt from position off[k] on
    __shared__ uint32_t lo[kWanderWindows + 3], off[kWanderWindows + 3];
    __shared__ uint32_t s_fill;
    const uint32_t slots = cap * kXcd;
    // width 2 / 4 (k_step_roam<REV, 512 / 1024>): a window's run is whole groups of `width` blocks and every list holds whole groups
    const uint32_t uw = static_cast<uint32_t>(width);
    const uint32_t kRun = kXcd * kBlock * uw;
    if (threadIdx.x <= kWanderWindows + 2) {
        // first index whose key is >= threadIdx.x
        uint32_t a = 0, b = slots;
        while (a < b) {
            const uint32_t m = (a + b) / 2;
            if (sorted_keys[m] / kWanderMix < threadIdx.x) a = m + 1; else b = m;
        }
        lo[threadIdx.x] = a;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        // contiguous deal: a block keeps `fill` of its kBlock slots (the rest are tombstones), chosen so that the
        // live tracks make about one block per CU.  A block-window kernel holds 144 KB of LDS, one block per
        // CU, and a divergent gather costs its CU ~4 clocks per lane: 44k survivors in 180 full blocks leave 76
        // CUs idle while the others take 1030 clocks per step.
        const uint32_t live = lo[kWanderWindows + 1];
        uint32_t fill = kBlock;
        if (contiguous) {
            // the blocks must stay under the 256 CUs x blocks per CU (a block beyond the first round of a launch finds the
            // stop flag up and waits for the others to finish the pass): kDealBlocks + one partial block per window IN USE
            // (round 4: the allowance of the unused ones goes to the deal, 246 instead of 232 blocks with two basins =
            // 63 000 instead of 59 392 live tracks in one round) + the padding to whole blocks of every list.
            // Wide: 216 groups + one per window + the padding
            uint32_t in_use = 0;
            for (int k = 0; k <= kWanderWindows; ++k) in_use += lo[k + 1] > lo[k] ? 1u : 0u;
            const uint32_t deal_blocks = uw > 1u ? uw * 216u : kDealBlocks + (kWanderWindows + 1u - in_use);
            fill = (live + deal_blocks - 1) / deal_blocks;
            fill = fill < 64u ? 64u : (fill > kBlock ? kBlock : fill);
        }
        uint32_t run = 0;
        for (int pass = 0; pass < 2; ++pass) {
            run = 0;
            for (int k = 0; k <= kWanderWindows; ++k) {              // (key kWanderWindows + 1 = dead: not dealt)
                off[k] = run;
                uint32_t blocks = (lo[k + 1] - lo[k] + fill - 1) / fill;
                blocks = (blocks + uw - 1u) / uw * uw;
                // round-robin deal: a window's run is whole blocks of EVERY list
                run += contiguous ? blocks * kBlock : (blocks * kBlock + kRun - 1) / kRun * kRun;
            }
            run = (run + kRun - 1) / kRun * kRun;
            if (run <= slots || fill == kBlock) break;
            fill = kBlock;                                           // no room for the thinned blocks
        }
        off[kWanderWindows + 1] = run;
        s_fill = fill;
        if (run > slots) {
            // no room for the padding (nearly every slot is live): dense deal, blocks may mix windows
            run = 0;
            for (int k = 0; k <= kWanderWindows + 1; ++k) { off[k] = run; if (k <= kWanderWindows) run += lo[k + 1] - lo[k]; }
            off[kWanderWindows + 1] = (run + kRun - 1) / kRun * kRun;   // (<= slots: cap is a multiple of 4 blocks, workspace_layout)
            off[kWanderWindows + 2] = 1;                             // dense
        } else {
            off[kWanderWindows + 2] = 0;
        }
    }
    __syncthreads();
    const uint32_t total = off[kWanderWindows + 1];                  // a multiple of kRun
    const bool dense = off[kWanderWindows + 2] != 0;
    const uint32_t fill = s_fill;
    for (uint32_t g = blockIdx.x * 1024u + threadIdx.x; g < total; g += gridDim.x * 1024u) {
        int32_t t = -1;
        if (dense) {
            if (g < lo[kWanderWindows + 1]) t = sorted[g];
        } else {
            int k = 0;
#pragma unroll
            for (int q = 1; q <= kWanderWindows; ++q) k += off[q] <= g ? 1 : 0;
            // a thinned block's tracks are dealt over its four waves evenly (slot = wave x 64 + lane <- position
            // 4 lane + wave): a divergent load takes ~300 clocks plus ~4 per active lane of its wave
            const uint32_t in_run = g - off[k], blk = in_run / kBlock, sl = in_run % kBlock;
            const uint32_t pos = (sl & 63u) * (kBlock / 64u) + (sl >> 6);
            const uint32_t j = lo[k] + blk * fill + pos;
            if (pos < fill && j < lo[k + 1]) t = sorted[j];
        }
        // contiguous: list x takes positions [x total / 8, (x + 1) total / 8), whole blocks of ONE window each
        // (total is a multiple of 8 blocks) -- an XCD then steps one or two windows and its L2 holds their
        // part of the roam table; round-robin (round 2): every window on every XCD
        if (contiguous) list_out[(g / (total / kXcd)) * cap + g % (total / kXcd)] = t;
        else list_out[(g & (kXcd - 1)) * cap + (g >> 3)] = t;
    }
    if (blockIdx.x == 0 && threadIdx.x < kXcd) {
        ctl->count[out_slot][threadIdx.x] = total / kXcd;
        ctl->count[zero_slot][threadIdx.x] = 0;
    }
}

static size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

struct Workspace {
    TrackCtl *ctl;
    double *thr;
    TrackState *state;
    int32_t *list[2];
    unsigned long long *keys[2];
    void *sort_temp;
    size_t sort_temp_bytes;
    uint32_t *bucket;            // the same visits ordered by raster tile (oblique headings)
    uint32_t *tile_count, *tile_start, *tile_cursor, *item_start;   // [kTilesMax] each, item_start one more
    WanderWindows *wander;
    RoamEntry *roam;             // kWanderWindows slabs (batches large enough for the wander sort), or nullptr
    uint32_t *visits;            // [kVisitSteps][visit_stride] visited cells of one launch
    long long visit_stride;
    uint32_t cap;                // slots per XCD list
};

constexpr int kVisitSteps = 1024;  // binning mode covers launches of up to this many steps
constexpr int64_t kWanderMinTracks = 8192;   // smaller batches are never sorted into windows

static size_t sort_temp_size(int64_t n)
{
    size_t bytes = 0;
    (void)hipcub::DeviceRadixSort::SortPairs(nullptr, bytes, static_cast<const unsigned long long *>(nullptr),
                                       static_cast<unsigned long long *>(nullptr),
                                       static_cast<const int32_t *>(nullptr),
                                       static_cast<int32_t *>(nullptr), static_cast<int>(n), 0, 40);
    size_t bytes32 = 0;
    (void)hipcub::DeviceRadixSort::SortPairs(nullptr, bytes32, static_cast<const uint32_t *>(nullptr),
                                       static_cast<uint32_t *>(nullptr), static_cast<const int32_t *>(nullptr),
                                       static_cast<int32_t *>(nullptr), static_cast<int>(n), 0, 32);
    return bytes > bytes32 ? bytes : bytes32;
}

static size_t workspace_layout(int64_t n, char *base, Workspace *ws)
{
    size_t off = 0;
    // kXcd lists of cap slots each; cap is a whole number of the widest blocks (k_step_roam<REV, 1024>: the wide deal rounds
    // every list up to whole groups of blocks, and its dense fall-back must still fit)
    const size_t cap = align_up((static_cast<size_t>(n) + kXcd - 1) / kXcd, 4 * kBlock);
    const size_t slots = cap * kXcd;
    if (ws) ws->cap = static_cast<uint32_t>(cap);
    if (ws) ws->ctl = reinterpret_cast<TrackCtl *>(base + off);
    off = align_up(off + sizeof(TrackCtl), 256);
    if (ws) ws->thr = reinterpret_cast<double *>(base + off);
    off = align_up(off + 81 * sizeof(double), 256);
    if (ws) ws->state = reinterpret_cast<TrackState *>(base + off);
    off = align_up(off + sizeof(TrackState) * static_cast<size_t>(n), 256);
    for (int i = 0; i < 2; ++i) {
        if (ws) ws->list[i] = reinterpret_cast<int32_t *>(base + off);
        off = align_up(off + sizeof(int32_t) * slots, 256);
    }
    for (int i = 0; i < 2; ++i) {
        if (ws) ws->keys[i] = reinterpret_cast<unsigned long long *>(base + off);
        off = align_up(off + sizeof(unsigned long long) * static_cast<size_t>(n), 256);
    }
    const size_t temp = n > 0 ? sort_temp_size(static_cast<int64_t>(slots)) : 0;     // (the wander sort covers every slot)
    if (ws) { ws->sort_temp = base + off; ws->sort_temp_bytes = temp; }
    off = align_up(off + temp, 256);
    const long long stride = static_cast<long long>(slots);
    // one buffer: binning runs on the launch stream right after its stepper launch
    // (a second buffer would only be needed to overlap it with the next launch)
    if (ws) { ws->visits = reinterpret_cast<uint32_t *>(base + off); ws->visit_stride = stride; }
    off = align_up(off + sizeof(uint32_t) * static_cast<size_t>(stride) * kVisitSteps, 256);
    if (ws) ws->bucket = reinterpret_cast<uint32_t *>(base + off);
    off = align_up(off + sizeof(uint32_t) * static_cast<size_t>(stride) * kVisitSteps, 256);
    if (ws) {
        ws->tile_count = reinterpret_cast<uint32_t *>(base + off);
        ws->tile_start = ws->tile_count + kTilesMax;
        ws->tile_cursor = ws->tile_start + kTilesMax;
        ws->item_start = ws->tile_cursor + kTilesMax;
    }
    off = align_up(off + sizeof(uint32_t) * (4 * kTilesMax + 1), 256);
    if (ws) ws->wander = reinterpret_cast<WanderWindows *>(base + off);
    off = align_up(off + sizeof(WanderWindows), 256);
    if (ws) ws->roam = nullptr;             // (the pair table lives behind the regular workspace: pair_table_bytes)
    return off;
}

// pinned host words for the live-count read-back, one set per host thread
constexpr int kFinalSlot = 384;      // words: the read-back ring (8 slots of 40) ends at 320
constexpr int kHeaderSlot = 336;     // words 336..357: a threshold table's header, read back before the first launch
static_assert(kHeaderSlot >= 320 && kHeaderSlot % 2 == 0 && kHeaderSlot * 4 + sizeof(ThrHeader) <= kFinalSlot * 4, "header slot");
static uint32_t *pinned_counts()
{
    static thread_local uint32_t *buf = nullptr;
    if (!buf && hipHostMalloc(reinterpret_cast<void **>(&buf), 512 * sizeof(uint32_t)) != hipSuccess)
        buf = nullptr;
    return buf;
}


}  // namespace ssrs

// the opaque handle of include/ssrs_hip.h: a bump allocator over the caller's pool plus
// the host-side directory of the launches recorded into it
struct SsrsTrajRecorder {
    char *pool;
    size_t bytes, used;
    std::vector<ssrs::TrajChunk> chunks;
    int complete;                // 1: every launch of the last simulation is in `chunks`
    int rows, cols;
    long long ntracks;
};

namespace ssrs {
}  // namespace ssrs

using namespace ssrs;

// Thresholds that depend on the heading's prior only (host, f64, the reference's sequence
// of movmodel.py:234-244 + np.random.choice): the masked prior after each last move rc and
// the unmasked prior.
static void prior_tables(const double *prior, ThrPrior *out)
{
    auto thresholds = [](const double *q9, double *thr) {
        double q[9];
        const double s1 = sum9(q9);
        for (int k = 0; k < 9; ++k) q[k] = q9[k] / s1;
        const double s2 = sum9(q);
        double acc = 0.0, cdf[9];
        for (int k = 0; k < 9; ++k) { q[k] = q[k] / s2; acc = acc + q[k]; cdf[k] = acc; }
        for (int k = 0; k < 9; ++k) thr[k] = cdf[k] / cdf[8];
    };
    *out = ThrPrior{};
    for (int rc = 0; rc < 8; ++rc) {
        const uint32_t mask = restriction(kRingK[rc]);
        double q[9], thr[9];
        bool any = false;
        for (int k = 0; k < 9; ++k) {
            q[k] = ((mask >> k) & 1u) && k != 4 ? prior[k] : 0.0;
            any |= q[k] != 0.0;
        }
        if (!any) { out->reversal |= 1u << rc; continue; }
        thresholds(q, thr);
        // the two boundaries between the three admissible cells in ascending k
        const uint32_t ord = ring_order(rc);
        const int ring3[3] = {(rc + 7) % 8, rc, (rc + 1) % 8};
        const int ka = kRingK[ring3[ord & 3u]], kb = kRingK[ring3[(ord >> 2) & 3u]];
        out->zero_e[rc] = thr_pack(thr[ka], thr[kb]);
    }
    double q[9], thr[9];
    for (int k = 0; k < 9; ++k) q[k] = k == 4 ? 0.0 : prior[k];
    thresholds(q, thr);
    for (int k = 0; k < 9; ++k) out->thr9[k] = thr_pack(thr[k], thr[k]) & 0xFFFFu;
    uint32_t support = 0;
    for (int k = 0; k < 9; ++k) support |= (q[k] != 0.0 ? 1u : 0u) << k;
    for (int rh = 0; rh < 8 && !out->rev_ok; ++rh) {
        if (support == 0 || (support & ~restriction(kRingK[rh])) != 0) continue;
        const uint32_t ord = ring_order(rh);
        const int ring3[3] = {(rh + 7) % 8, rh, (rh + 1) % 8};
        const int ka = kRingK[ring3[ord & 3u]], kb = kRingK[ring3[(ord >> 2) & 3u]];
        out->rev_ok = 1;
        out->rev_rc = static_cast<uint32_t>(rh);
        out->rev_e = thr_pack(thr[ka], thr[kb]);
    }
}

// bit 0: a timing-probe build (csrc/build.py --probe*: some SSRS_PROBE_* switch is defined and the
// results are wrong on purpose).  ssrs_amd refuses to load such a library unless told to.
extern "C" int ssrs_build_flags(void)
{
#if defined(SSRS_PROBE_NO_PHILOX) || defined(SSRS_PROBE_NO_GATHER) || defined(SSRS_PROBE_K2A_NOLOAD) || \
    defined(SSRS_PROBE_K2A_NOSTORE) || defined(SSRS_PROBE_K2A_PAD) || defined(SSRS_K2A_NT) || \
    defined(SSRS_PROBE_K3_NOREAD) || defined(SSRS_PROBE_K3_NOFLUSH) || defined(SSRS_PROBE_NO_STRAY_ATOMICS)
    return 1;
#else
    return 0;
#endif
}

// The masked prior's two boundaries after each last move at 32 bits (k_fine_build's zero rows), by the
// reference's sequence of movmodel.py:234-244 in f64
static void fine_prior_tables(const double *prior, FinePrior *out)
{
    *out = FinePrior{};
    for (int rc = 0; rc < 8; ++rc) {
        const uint32_t mask = restriction(kRingK[rc]);
        double q[9], qq[9], cdf[9];
        bool any = false;
        for (int k = 0; k < 9; ++k) {
            q[k] = ((mask >> k) & 1u) && k != 4 ? prior[k] : 0.0;
            any |= q[k] != 0.0;
        }
        if (!any) { out->reversal |= 1u << rc; continue; }
        const double s1 = sum9(q);
        for (int k = 0; k < 9; ++k) qq[k] = q[k] / s1;
        const double s2 = sum9(qq);
        double acc = 0.0;
        for (int k = 0; k < 9; ++k) { qq[k] = qq[k] / s2; acc = acc + qq[k]; cdf[k] = acc; }
        const uint32_t ord = ring_order(rc);
        const int ring3[3] = {(rc + 7) % 8, rc, (rc + 1) % 8};
        const int ka = kRingK[ring3[ord & 3u]], kb = kRingK[ring3[(ord >> 2) & 3u]];
        out->zero_t1[rc] = fine_fixed(cdf[ka] / cdf[8]);
        out->zero_t2[rc] = fine_fixed(cdf[kb] / cdf[8]);
        if (out->zero_t1[rc] > out->zero_t2[rc]) out->zero_t1[rc] = out->zero_t2[rc];
    }
}

extern "C" int ssrs_track_params_init(SsrsTrackParams *p, int rows, int cols,
                                      int memory_parameter, double scaling_parameter)
{
    SSRS_REQUIRE(p != nullptr, "ssrs_track_params_init: params is NULL");
    SSRS_REQUIRE(rows >= 5 && cols >= 5, "ssrs_track_params_init: need rows, cols >= 5");
    *p = SsrsTrackParams{};
    p->rows = rows;
    p->cols = cols;
    p->burnin = static_cast<int32_t>((rows < cols ? rows : cols) / 10.0);   // int(min/10)
    const double mm = rows / 2.0 * cols / 2.0;                              // movmodel.py:277
    p->max_moves = static_cast<int64_t>(ceil(mm));
    p->memory_parameter = memory_parameter;
    p->scaling_parameter = scaling_parameter;
    return SSRS_OK;
}

extern "C" size_t ssrs_tracks_workspace_bytes(int64_t ntracks)
{
    if (ntracks < 0) ntracks = 0;
    return workspace_layout(ntracks, nullptr, nullptr);
}

// The pair table of a roaming batch (k_step_roam): 8 x 16 bytes per cell, behind the regular workspace; only
// batches large enough for the wander sort roam in block windows, and its byte offsets are 32-bit
static size_t pair_table_bytes(int64_t ntracks, int rows, int cols)
{
    if (ntracks < kWanderMinTracks || rows <= 0 || cols <= 0) return 0;
    const size_t cells = static_cast<size_t>(rows) * static_cast<size_t>(cols);
    // 8 x 16 bytes per cell + the fine table of the near-ties, 8 x 8 bytes per cell
    return cells < kPairMaxCells ? align_up(cells * 8 * sizeof(RoamEntry), 256) + align_up(cells * 8 * sizeof(FineEntry), 256) : 0;
}

extern "C" size_t ssrs_tracks_workspace_bytes_ex(int64_t ntracks, int rows, int cols, int hist_copies)
{
    const size_t base = align_up(ssrs_tracks_workspace_bytes(ntracks), 256) + pair_table_bytes(ntracks, rows, cols);
    if (rows <= 0 || cols <= 0 || hist_copies < 2) return base;
    if (hist_copies > 64) hist_copies = 64;
    return base + static_cast<size_t>(hist_copies) * static_cast<size_t>(rows) * static_cast<size_t>(cols) * sizeof(uint32_t);
}

extern "C" int ssrs_transition_table_build(const double *updraft, const float *potential,
                                           double *table, int rows, int cols, void *stream)
{
    SSRS_REQUIRE(updraft && table, "ssrs_transition_table_build: updraft/table is NULL");
    SSRS_REQUIRE(rows >= 3 && cols >= 3, "ssrs_transition_table_build: need rows, cols >= 3");
    SSRS_REQUIRE((reinterpret_cast<uintptr_t>(table) & 63u) == 0,
                 "ssrs_transition_table_build: table must be 64-byte aligned");
    const int tx = (cols + kTabW - 1) / kTabW, ty = (rows + kTabH - 1) / kTabH, nt = tx * ty;
    hipLaunchKernelGGL(k_transition_table<false>, dim3(static_cast<unsigned>(nt)), dim3(kBlock), 0,
                       as_stream(stream), updraft, potential, static_cast<void *>(table), rows, cols, tx, nt);
    SSRS_HIP_CHECK(hipGetLastError());
    return SSRS_OK;
}

extern "C" size_t ssrs_transition_ring_bytes(int rows, int cols)
{
    if (rows <= 0 || cols <= 0) return 0;
    // records + 8 bytes (the 12-byte load of the last cell's last triple stays inside the
    // buffer) + one zero-mask byte per cell, rounded to whole floats
    const size_t cells = static_cast<size_t>(rows) * static_cast<size_t>(cols);
    return ring_mask_offset(rows, cols) + (cells + 3) / 4 * 4;
}

extern "C" int ssrs_transition_ring_build(const double *updraft, const float *potential,
                                          float *ring, int rows, int cols, void *stream)
{
    SSRS_REQUIRE(updraft && ring, "ssrs_transition_ring_build: updraft/ring is NULL");
    SSRS_REQUIRE(rows >= 3 && cols >= 3, "ssrs_transition_ring_build: need rows, cols >= 3");
    SSRS_REQUIRE((reinterpret_cast<uintptr_t>(ring) & 7u) == 0,
                 "ssrs_transition_ring_build: ring must be 8-byte aligned");
    const int tx = (cols + kTabW - 1) / kTabW, ty = (rows + kTabH - 1) / kTabH, nt = tx * ty;
    hipLaunchKernelGGL(k_transition_table<true>, dim3(static_cast<unsigned>(nt)), dim3(kBlock), 0,
                       as_stream(stream), updraft, potential, static_cast<void *>(ring), rows, cols, tx, nt);
    SSRS_HIP_CHECK(hipGetLastError());
    return SSRS_OK;
}

// The stepper issues the next gather BEFORE it knows whether the lane's entry was a flag (boundary
// cell, ...): such a lane's speculative address is a neighbour of its cell, up to cols + 1 cells
// outside a plane.  A guard band at both ends of the allocation keeps those loads inside it (their
// values are never used).
static size_t thr_guard_bytes(int cols) { return align_up((static_cast<size_t>(cols) + 2) * 4, 256); }

extern "C" size_t ssrs_transition_thr_bytes(int rows, int cols)
{
    if (rows <= 0 || cols <= 0) return 0;
    // eight planes of rows * cols dwords, a power-of-two stride apart, between two guard bands
    return (static_cast<size_t>(8) << thr_plane_shift(rows, cols)) + 2 * thr_guard_bytes(cols);
}

extern "C" int ssrs_transition_thr_build(const double *updraft, const float *potential,
                                         const double *prior, float *thr, int rows, int cols, void *stream)
{
    SSRS_REQUIRE(updraft && thr && prior, "ssrs_transition_thr_build: updraft/prior/thr is NULL");
    SSRS_REQUIRE(rows >= 3 && cols >= 3, "ssrs_transition_thr_build: need rows, cols >= 3");
    SSRS_REQUIRE(static_cast<size_t>(rows) * static_cast<size_t>(cols) <= (1ull << 26),
                 "ssrs_transition_thr_build: the table and its guard bands are addressed with 32-bit offsets (rows * cols <= 2^26)");
    SSRS_REQUIRE((reinterpret_cast<uintptr_t>(thr) & 63u) == 0,
                 "ssrs_transition_thr_build: table must be 64-byte aligned");
    ThrPrior pr;
    prior_tables(prior, &pr);
    const int tx = (cols + kTabW - 1) / kTabW, ty = (rows + kTabH - 1) / kTabH, nt = tx * ty;
    uint32_t *planes = reinterpret_cast<uint32_t *>(reinterpret_cast<char *>(thr) + thr_guard_bytes(cols));
    ThrHeader *header = reinterpret_cast<ThrHeader *>(thr);
    PriorArg heading;
    for (int k = 0; k < 9; ++k) heading.v[k] = prior[k];
    if (potential)
        hipLaunchKernelGGL(k_transition_thr<true>, dim3(static_cast<unsigned>(nt)), dim3(kBlock), 0, as_stream(stream),
                           updraft, potential, planes, rows, cols, tx, nt, pr, thr_plane_shift(rows, cols), header, heading);
    else
        hipLaunchKernelGGL(k_transition_thr<false>, dim3(static_cast<unsigned>(nt)), dim3(kBlock), 0, as_stream(stream),
                           updraft, potential, planes, rows, cols, tx, nt, pr, thr_plane_shift(rows, cols), header, heading);
    SSRS_HIP_CHECK(hipGetLastError());
    return SSRS_OK;
}

extern "C" int ssrs_uniform_selftest(uint64_t seed, const uint64_t *track, const uint64_t *step,
                                     double *out, size_t n, void *stream)
{
    SSRS_REQUIRE(track && step && out, "ssrs_uniform_selftest: NULL pointer");
    if (n == 0) return SSRS_OK;
    hipLaunchKernelGGL(k_uniform_selftest, dim3(static_cast<unsigned>((n + 255) / 256)), dim3(256),
                       0, as_stream(stream), seed, track, step, out, n);
    SSRS_HIP_CHECK(hipGetLastError());
    return SSRS_OK;
}

static int tracks_simulate_impl(const SsrsTrackParams *p, const double *updraft,
                                const float *potential, const double *table,
                                const int32_t *start_rc, int64_t ntracks, uint64_t seed,
                                uint64_t track_id_base, uint32_t *hist, int16_t *end_rc,
                                int32_t *lengths, int16_t *traj, const int64_t *traj_offsets,
                                void *workspace, size_t workspace_bytes,
                                SsrsTrackStats *stats, void *stream, SsrsTrajRecorder *rec,
                                unsigned long long *hist64 = nullptr)
{
    SSRS_REQUIRE(p != nullptr, "ssrs_tracks_simulate: params is NULL");
    SSRS_REQUIRE(hist64 == nullptr || hist != nullptr, "ssrs_tracks_simulate_h64: the 32-bit raster the kernels count into is NULL");
    SSRS_REQUIRE(p->rows >= 5 && p->cols >= 5, "ssrs_tracks_simulate: need rows, cols >= 5 (got %d x %d)",
                 p->rows, p->cols);
    SSRS_REQUIRE(p->rows <= 32767 && p->cols <= 32767,
                 "ssrs_tracks_simulate: rows, cols must fit int16 trajectories (<= 32767)");
    SSRS_REQUIRE(p->memory_parameter >= 0 && p->memory_parameter <= 8,
                 "ssrs_tracks_simulate: memory_parameter must be in 0..8 (got %d)",
                 p->memory_parameter);
    SSRS_REQUIRE(p->max_moves >= 0 && p->max_moves < (1ll << 31) - 1,
                 "ssrs_tracks_simulate: max_moves out of range");
    SSRS_REQUIRE(p->burnin >= 0, "ssrs_tracks_simulate: burnin must be >= 0");
    SSRS_REQUIRE(ntracks >= 0 && ntracks < (1ll << 31) - 64, "ssrs_tracks_simulate: bad ntracks");
    SSRS_REQUIRE(!(potential && !updraft && !table),
                 "ssrs_tracks_simulate: potential_field needs updraft_field (reference raises)");
    SSRS_REQUIRE(!(traj && !traj_offsets), "ssrs_tracks_simulate: traj needs traj_offsets");
    SSRS_REQUIRE(!(table && (reinterpret_cast<uintptr_t>(table) & 63u)),
                 "ssrs_tracks_simulate: table must be 64-byte aligned");
    if (stats) *stats = SsrsTrackStats{};
    if (rec) {
        rec->used = 0;
        rec->chunks.clear();
        rec->complete = 1;
        rec->rows = p->rows;
        rec->cols = p->cols;
        rec->ntracks = ntracks;
    }
    if (ntracks == 0) return SSRS_OK;
    SSRS_REQUIRE(start_rc != nullptr, "ssrs_tracks_simulate: start_rc is NULL");
    SSRS_REQUIRE(workspace && workspace_bytes >= ssrs_tracks_workspace_bytes(ntracks),
                 "ssrs_tracks_simulate: workspace too small (need %zu bytes)",
                 ssrs_tracks_workspace_bytes(ntracks));
    SSRS_REQUIRE((reinterpret_cast<uintptr_t>(workspace) & 255u) == 0,
                 "ssrs_tracks_simulate: workspace must be 256-byte aligned");

    hipStream_t st = as_stream(stream);
    Workspace ws;
    workspace_layout(ntracks, static_cast<char *>(workspace), &ws);
    uint32_t *host_counts = pinned_counts();
    SSRS_REQUIRE(host_counts != nullptr, "ssrs_tracks_simulate: hipHostMalloc failed");

    const int S = p->steps_per_launch > 0 ? p->steps_per_launch : 512;
    const bool profile = (p->flags & SSRS_TRACKS_PROFILE) != 0;
    const int mode = table ? MODE_TABLE : (updraft ? (potential ? MODE_FLUIDFLOW : MODE_UPDRAFT)
                                                   : MODE_PRIOR);

    if (p->flags & SSRS_TRACKS_THR_TABLE) {
        // A threshold table names its raster and prior in its leading guard band.  The header is read back
        // and checked HERE, before any stepper kernel gathers from the table: a table built for a smaller
        // raster would send those gathers out of bounds (k_ctl_init's device-side check of the same
        // header is only acted on when the run is over).  88 bytes and one stream wait per call.
        SSRS_REQUIRE(table != nullptr, "ssrs_tracks_simulate: SSRS_TRACKS_THR_TABLE needs `table`");
        ThrHeader *hh = reinterpret_cast<ThrHeader *>(host_counts + kHeaderSlot);
        SSRS_HIP_CHECK(hipMemcpyAsync(hh, table, sizeof(ThrHeader), hipMemcpyDeviceToHost, st));
        SSRS_HIP_CHECK(hipStreamSynchronize(st));
        bool bad = hh->magic != kThrMagic || hh->rows != p->rows || hh->cols != p->cols;
        for (int k = 0; k < 9; ++k) bad |= hh->prior[k] != p->prior[k];
        if (bad)
            return set_error(SSRS_ERR_INVALID, "ssrs_tracks_simulate: `table` is not a threshold table built by "
                             "ssrs_transition_thr_build for this %d x %d raster and params->prior (nothing was launched)",
                             p->rows, p->cols);
    }

    hipEvent_t ev_first = nullptr, ev_last = nullptr;
    SSRS_HIP_CHECK(hipEventCreate(&ev_first));
    SSRS_HIP_CHECK(hipEventCreate(&ev_last));
    SSRS_HIP_CHECK(hipEventRecord(ev_first, st));
    {
        PriorArg pa;
        for (int k = 0; k < 9; ++k) pa.v[k] = p->prior[k];
        // (a threshold table names its raster and prior in its leading guard band: checked on the device)
        const ThrHeader *header = (p->flags & SSRS_TRACKS_THR_TABLE) ? reinterpret_cast<const ThrHeader *>(table) : nullptr;
        hipLaunchKernelGGL(k_ctl_init, dim3(1), dim3(64), 0, st, ws.ctl, pa, ws.thr, ws.wander, header, p->rows, p->cols);
        SSRS_HIP_CHECK(hipGetLastError());
    }
    const bool coherent = (p->flags & SSRS_TRACKS_NO_SCHEDULE) == 0;
    PlanGeom geom = {};
    {
        const unsigned blocks = static_cast<unsigned>((ntracks + kBlock - 1) / kBlock);
        if (coherent) {
            // movement direction from the prior: its lobe peaks along the heading
            // (movmodel.py:247-257), so the weighted neighbour offsets give it back
            double vr = 0.0, vc = 0.0;
            for (int k = 0; k < 9; ++k) { vr += p->prior[k] * dr_of(k); vc += p->prior[k] * dc_of(k); }
            const double nrm = std::sqrt(vr * vr + vc * vc);
            geom.cos_t = nrm > 0.0 ? vr / nrm : 1.0;
            geom.sin_t = nrm > 0.0 ? vc / nrm : 0.0;
            geom.offset = p->rows + p->cols;
            int par_bits = 1;
            while ((1ll << par_bits) <= 2ll * geom.offset) ++par_bits;       // <= 18 (rows, cols <= 32767)
            size_t temp_bytes = ws.sort_temp_bytes;
            if (2 * par_bits <= 32) {
                uint32_t *k0 = reinterpret_cast<uint32_t *>(ws.keys[0]), *k1 = reinterpret_cast<uint32_t *>(ws.keys[1]);
                hipLaunchKernelGGL(k_plan_keys<uint32_t>, dim3(blocks), dim3(kBlock), 0, st, start_rc,
                                   static_cast<long long>(ntracks), geom, k0, ws.list[1], ws.ctl, par_bits);
                SSRS_HIP_CHECK(hipGetLastError());
                SSRS_HIP_CHECK(hipcub::DeviceRadixSort::SortPairs(
                    ws.sort_temp, temp_bytes, k0, k1, ws.list[1], ws.list[0], static_cast<int>(ntracks), 0, 2 * par_bits, st));
            } else {
                hipLaunchKernelGGL(k_plan_keys<unsigned long long>, dim3(blocks), dim3(kBlock), 0, st, start_rc,
                                   static_cast<long long>(ntracks), geom, ws.keys[0], ws.list[1], ws.ctl, par_bits);
                SSRS_HIP_CHECK(hipGetLastError());
                SSRS_HIP_CHECK(hipcub::DeviceRadixSort::SortPairs(
                    ws.sort_temp, temp_bytes, ws.keys[0], ws.keys[1], ws.list[1], ws.list[0],
                    static_cast<int>(ntracks), 0, 2 * par_bits, st));
            }
        }
        hipLaunchKernelGGL(k_tracks_init, dim3(blocks), dim3(kBlock), 0, st, start_rc,
                           static_cast<long long>(ntracks), p->rows, p->cols, hist, traj,
                           reinterpret_cast<const long long *>(traj_offsets), lengths, end_rc,
                           ws.state, ws.ctl, geom, coherent ? 1 : 0, ws.cap);
        SSRS_HIP_CHECK(hipGetLastError());
    }

    StepArgs a = {};
    a.rows = p->rows; a.cols = p->cols; a.burnin = p->burnin; a.memory = p->memory_parameter;
    a.max_k = p->max_moves; a.nu = p->scaling_parameter;
    a.prior = ws.ctl->prior;
    a.updraft = updraft; a.potential = potential; a.table = table;
    a.seed = seed; a.track_base = track_id_base;
    a.hist = hist; a.end_rc = end_rc; a.lengths = lengths; a.traj = traj;
    a.traj_off = reinterpret_cast<const long long *>(traj_offsets);
    a.state = ws.state; a.ctl = ws.ctl; a.steps = S;
    a.fast = ((p->flags & SSRS_TRACKS_EXACT_ONLY) == 0 && p->scaling_parameter == 1.0) ? 1 : 0;
    a.coherent = coherent ? 1 : 0;
    a.cap = ws.cap;
    a.thr = ws.thr;
    a.wander = ws.wander;
    a.debug_roam = std::getenv("SSRS_TRACKS_DEBUG_ROAM") != nullptr ? 1 : 0;
    a.roam_stop = std::getenv("SSRS_TRACKS_NO_ROAM_STOP") == nullptr ? 1 : 0;
    a.dbg_buf = nullptr;
#ifdef SSRS_DEBUG_WAVE_DUMP
    {
        static unsigned long long *dump_buf = nullptr;          // (diagnostic build: never freed)
        if (!dump_buf && a.debug_roam && hipMalloc(&dump_buf, 8ull * 8 * 4096 * (kBlock / 64)) != hipSuccess) dump_buf = nullptr;
        if (dump_buf) (void)hipMemsetAsync(dump_buf, 0, 8ull * 8 * 4096 * (kBlock / 64), st);
        a.dbg_buf = dump_buf;
    }
#endif
    a.cheap_exact = std::getenv("SSRS_TRACKS_NO_CHEAP_EXACT") == nullptr ? 1 : 0;
    a.vcap = ws.cap;
    const bool lean = p->memory_parameter == 1 && traj == nullptr;
    const bool ring = (p->flags & SSRS_TRACKS_RING_TABLE) != 0;
    const bool thr = (p->flags & SSRS_TRACKS_THR_TABLE) != 0;
    ThrPrior thr_prior = {};
    if (thr) {
        SSRS_REQUIRE(!ring && table && updraft && lean && a.fast && (S & 1) == 0,
                     "ssrs_tracks_simulate: SSRS_TRACKS_THR_TABLE needs table + updraft, memory_parameter 1, "
                     "scaling_parameter 1, no direct trajectory output, no EXACT_ONLY and an even steps_per_launch");
        SSRS_REQUIRE(static_cast<size_t>(p->rows) * static_cast<size_t>(p->cols) <= (1ull << 26),
                     "ssrs_tracks_simulate: the threshold table needs rows * cols <= 2^26");
        prior_tables(p->prior, &thr_prior);
        a.plane_shift = thr_plane_shift(p->rows, p->cols);
        a.guard = static_cast<uint32_t>(thr_guard_bytes(p->cols));
        // prefetch wave: the heading's ring position (0 = north, 4 = south); A/B switch SSRS_TRACKS_NO_PREFETCH
        a.pf_dir = (coherent && std::getenv("SSRS_TRACKS_NO_PREFETCH") == nullptr)
                       ? (geom.cos_t > 0.98 ? 1 : (geom.cos_t < -0.98 ? -1 : 0)) : 0;
        a.pf_rc = a.pf_dir < 0 ? 4 : 0;
    }
    const int mode0 = updraft ? (potential ? MODE_FLUIDFLOW : MODE_UPDRAFT) : MODE_PRIOR;   // thr: first move
    if (ring)
        SSRS_REQUIRE(table && updraft && lean && a.fast && (S & 1) == 0,
                     "ssrs_tracks_simulate: SSRS_TRACKS_RING_TABLE needs table + updraft, memory_parameter 1, "
                     "scaling_parameter 1, no trajectory output, no EXACT_ONLY and an even steps_per_launch");
    // binning needs the coherent front (a step's visits fall into a few rows).  The
    // binning kernel runs on the SAME stream, after its stepper launch: running it on a
    // side stream to overlap the next launch was measured and rejected (its 1024-thread /
    // 120-KB-LDS blocks crowd the latency-bound stepper waves: stepper 6.4 -> 10.2 ms,
    // binning 1.4 -> 3.5 ms; profiles/r01_notes.md).
    // east / west headings: the front is a column, so the batch bins into a TRANSPOSED
    // histogram (needs one raster of extra workspace, see below)
    // oblique headings: visits bucketed by raster tile (k_tile_sort / k_bin_bucket)
    const double off_axis = std::fabs(geom.sin_t) < std::fabs(geom.cos_t) ? std::fabs(geom.sin_t) : std::fabs(geom.cos_t);
    const uint32_t ntc = static_cast<uint32_t>((p->cols + kTileCols - 1) / kTileCols);
    const uint32_t ntiles = ntc * static_cast<uint32_t>((p->rows + kTileRows - 1) / kTileRows);
    // (bucket offsets are 32-bit: a launch must make fewer than 2^32 visits)
    const bool tiles_ok = hist != nullptr && coherent && S <= kVisitSteps && ntracks >= 8192 && ntiles <= kTilesMax &&
                          static_cast<unsigned long long>(ws.cap) * kXcd * static_cast<unsigned long long>(S) < (1ull << 32) &&
                          (p->flags & SSRS_TRACKS_NO_BINNING) == 0 && std::getenv("SSRS_TRACKS_NO_TILES") == nullptr;
    const bool force_tiles = tiles_ok && std::getenv("SSRS_TRACKS_FORCE_TILES") != nullptr;   // A/B switch
    bool tiles_on = tiles_ok && (off_axis > 0.17 || force_tiles);   // more than ~10 degrees off a raster axis
    const bool want_transposed = coherent && !tiles_on && std::fabs(geom.sin_t) > std::fabs(geom.cos_t);
    bool binning = hist != nullptr && coherent && S <= kVisitSteps &&
                   (p->flags & SSRS_TRACKS_NO_BINNING) == 0 &&
                   (want_transposed ? p->rows : p->cols) <= kBinCells;
    a.visits = nullptr;
    a.visit_stride = ws.visit_stride;
    bool binning_on = binning && ntracks >= 8192 && !tiles_on;   // small batches: plain atomics are cheaper
    // SSRS_TRACKS_SCATTERED forces the zero-mask variant from the first launch,
    // SSRS_TRACKS_NO_SCATTERED keeps it off (A/B switches; results are identical)
    const bool never_scattered = (p->flags & SSRS_TRACKS_NO_SCATTERED) != 0;
    bool scattered = (p->flags & SSRS_TRACKS_SCATTERED) != 0;
    if (scattered) binning_on = tiles_on = false;
    // threshold stepper, front outgrown the row window (tracks wander): atomics behind the per-lane
    // block windows in LDS instead of tile buckets (A/B switch SSRS_TRACKS_NO_BLOCK_WINDOW)
    // (never while trajectories are recorded: those launches use the visit-buffer kernels, which
    // know no tombstones)
    const bool cache_ok = thr && hist != nullptr && !(rec && rec->complete) && std::getenv("SSRS_TRACKS_NO_BLOCK_WINDOW") == nullptr;
    bool cached = cache_ok && scattered;
    bool want_wander_sort = cached;
    int wander_sorts = 0, wander_cooldown = 0, stable_batches = 0, upper_from = 0;
    size_t ws_base = align_up(ssrs_tracks_workspace_bytes(ntracks), 256);
    // room behind the regular workspace: first the pair table of the roaming regime (threshold stepper only)
    {
        const size_t pb = thr ? pair_table_bytes(ntracks, p->rows, p->cols) : 0;
        if (pb && workspace_bytes >= ws_base + pb) {
            ws.roam = reinterpret_cast<RoamEntry *>(static_cast<char *>(workspace) + ws_base);
            if (std::getenv("SSRS_TRACKS_NO_FINE_TABLE") == nullptr)          // A/B switch
                a.fine = static_cast<char *>(workspace) + ws_base +
                         align_up(static_cast<size_t>(p->rows) * static_cast<size_t>(p->cols) * 8 * sizeof(RoamEntry), 256);
            ws_base += pb;
        }
        a.roam = ws.roam;
    }
    // pair table (k_step_roam): built when the batch starts to roam; A/B switches SSRS_TRACKS_NO_ROAM_TABLE, SSRS_TRACKS_DEAL_ROUND_ROBIN
    const bool roam_ok = cache_ok && ws.roam != nullptr && std::getenv("SSRS_TRACKS_NO_ROAM_TABLE") == nullptr;
    const bool roam_rev = thr_prior.rev_ok != 0 && std::getenv("SSRS_TRACKS_NO_REV") == nullptr;
    const bool deal_contiguous = std::getenv("SSRS_TRACKS_DEAL_ROUND_ROBIN") == nullptr;
    // LDS-staged table rows for fronts (k_step_thr<.., LR = true>): built and parity-tested, measured SLOWER than the
    // gather behind the prefetch wave (profiles/r03_notes.md section 7), so it is off unless asked for
    // (=2: waves that are ahead of the ring wait for their row instead of gathering it -- slower still)
    const char *lds_rows_env = std::getenv("SSRS_TRACKS_LDS_ROWS");
    const bool lds_rows = lds_rows_env != nullptr;
    a.lr_wait = (lds_rows && std::atoi(lds_rows_env) == 2) ? 1 : 0;
    bool roam_ready = false;
    // wide roaming blocks (k_step_roam<REV, 512>): chosen at a deal when more tracks are alive than one round of 256-lane blocks
    // holds (SSRS_TRACKS_ROAM_WIDE=<live tracks from which on>, 0: never, 1: always), kept until the next deal
    int roam_width = 1;                      // 1, 2, 4: blocks of 256, 512, 1024 lanes
    long long roam_wide_from = static_cast<long long>(kDealBlocks + kWanderWindows - 2) * kBlock + 1;   // (62 977: one more than a narrow deal with three windows in use holds, 246 blocks)
    if (const char *e = std::getenv("SSRS_TRACKS_ROAM_WIDE")) roam_wide_from = std::atoll(e);
    int roam_width_forced = 0;               // SSRS_TRACKS_ROAM_WIDTH=1|2|4: that width at every deal (A/B)
    if (const char *e = std::getenv("SSRS_TRACKS_ROAM_WIDTH")) { const int w = std::atoi(e); if (w == 1 || w == 2 || w == 4) roam_width_forced = w; }
    int roam_wide_launches = 0;
    int roam_launches = 0, stable_roam = 0, since_shuffle = 0, roam_shuffles = 0;
    bool sort_is_periodic = false;
    int roam_shuffle = 16;                   // batches between two shuffles of a settled roaming batch (SSRS_TRACKS_ROAM_SHUFFLE, 0: never)
    if (const char *e = std::getenv("SSRS_TRACKS_ROAM_SHUFFLE")) roam_shuffle = std::atoi(e);
    int roam_steps = 128 * S;                // A/B: SSRS_TRACKS_ROAM_STEPS (4096: 0.0073, 16384: 0.0060, 65536: 0.0053 ns per step at C2)
    if (const char *e = std::getenv("SSRS_TRACKS_ROAM_STEPS")) {
        const int v = std::atoi(e);
        if (v >= 2 && v <= (1 << 20)) roam_steps = v & ~1;
    }
    uint32_t prev_total = 0;
    // (the key arrays hold 2 n words >= the list slots; the coarse grid of k_wander_windows fits its LDS)
    const bool wander_sort_ok = ntracks >= kWanderMinTracks &&
                                static_cast<long long>((p->rows + kBinRows - 1) / kBinRows) * ((p->cols + kBinCols - 1) / kBinCols) <= kWanderBins;
    // private histogram copies live behind the regular workspace when the caller gave room
    const size_t ncell = static_cast<size_t>(p->rows) * static_cast<size_t>(p->cols);
    int ncopies = 0;
    if (hist && workspace_bytes > ws_base) ncopies = static_cast<int>((workspace_bytes - ws_base) / (ncell * sizeof(uint32_t)));
    uint32_t *extra = reinterpret_cast<uint32_t *>(static_cast<char *>(workspace) + ws_base);
    // the first extra raster is the transposed histogram of an east / west batch
    uint32_t *hist_t = nullptr;
    if (want_transposed && binning_on) {
        if (ncopies >= 1) { hist_t = extra; extra += ncell; --ncopies; }
        else { binning = false; binning_on = false; }      // no room: per-step atomics
    }
    if (hist_t) SSRS_HIP_CHECK(hipMemsetAsync(hist_t, 0, sizeof(uint32_t) * ncell, st));
    a.vis_r = hist_t ? 1u : static_cast<uint32_t>(p->cols);
    a.vis_c = hist_t ? static_cast<uint32_t>(p->rows) : 1u;
    ncopies = ncopies > 64 ? 64 : ncopies;
    uint32_t *copies_ptr = ncopies >= 2 ? extra : nullptr;
    bool copies_live = false;
    a.hist_copies = nullptr;
    a.ncopies = 1;
    a.zmask = ring ? reinterpret_cast<const uint8_t *>(table) + ring_mask_offset(p->rows, p->cols) : nullptr;
    unsigned long long seen_steps = 0, seen_strays = 0;
    // Launch loop.  Launches are queued kBatch deep; the live count of a batch
    // is copied back asynchronously and examined while the next batch runs, so
    // the GPU never waits on the host.  Launches past the end see count 0.
    constexpr int kBatch = 2, kRing = 8;
    constexpr int kSlotWords = 40;           // 160 bytes of TrackCtl: counts .. strays
    static_assert(offsetof(TrackCtl, strays) + sizeof(unsigned long long) <= kSlotWords * sizeof(uint32_t), "read-back slot");
    int slot_row[kRing] = {};
    bool slot_block_window[kRing] = {};      // the batch's launches counted in block windows
    long long block_window_steps = 0;
    const bool direct_read_back = std::getenv("SSRS_TRACKS_COPY_READ_BACK") == nullptr;      // A/B switch
    // (profile mode: every batch gets its own event, which is also the start mark of the next launch)
    hipEvent_t ev_batch[kRing] = {};
    if (!profile)
        for (int i = 0; i < kRing; ++i) SSRS_HIP_CHECK(hipEventCreate(&ev_batch[i]));
    // SSRS_TRACKS_PROFILE: ONE event per boundary (an event record costs the stream ~5 us): a stepper
    // launch runs from the mark before it to its own mark (kind 1), its binning kernels from there to
    // theirs (kind 2); an explicit start mark (kind 0) only where something else was queued in between
    std::vector<hipEvent_t> ev_marks;
    std::vector<int> mark_kind;
    bool marks_adjacent = false;             // the last mark is the start of whatever is queued next
    auto mark = [&](int kind) {
        hipEvent_t e;
        if (hipEventCreate(&e) != hipSuccess) return;
        (void)hipEventRecord(e, st);
        ev_marks.push_back(e);
        mark_kind.push_back(kind);
        marks_adjacent = true;
    };
    int launch = 0;
    long long it_done = 0;                   // threshold stepper: iterations of the launches so far (after the first move)
    bool want_rebalance = false;
    int rebalance_cooldown = 0;              // batches to look past after a re-deal (their counts are older than it)
    const bool may_rebalance = std::getenv("SSRS_TRACKS_NO_REBALANCE") == nullptr;
    const bool grow_steps = std::getenv("SSRS_TRACKS_FIXED_STEPS") == nullptr;      // A/B switch
    const bool v16_ok = 5ll * p->cols <= kBinCells && std::getenv("SSRS_TRACKS_NO_VISITS16") == nullptr;
    a.v16_offset = geom.offset;
    // bound on the longest XCD list
    uint32_t upper = static_cast<uint32_t>(ntracks < static_cast<int64_t>(ws.cap) ? ntracks : ws.cap);
    int batches = 0, checked = 0, judge_from = 0, last_Sl = 0;
    int window_launches = 0, tile_launches = 0, block_window_launches = 0;
    bool finished = false;
    int rc = SSRS_OK;
    // Termination: every live track either finishes or takes S moves per
    // launch and k < max_moves, so the live count reaches 0.
    while (!finished && rc == SSRS_OK) {
        if (want_wander_sort && cached && wander_sort_ok && launch > 0) {
            // pseudo-launch: list[launch & 1] -> windows, keys, sort -> padded deal into list[(launch + 1) & 1]
            const uint32_t slots = ws.cap * kXcd;
            uint32_t *k0 = reinterpret_cast<uint32_t *>(ws.keys[0]), *k1 = reinterpret_cast<uint32_t *>(ws.keys[1]);
            int32_t *sorted = reinterpret_cast<int32_t *>(ws.bucket);
            hipLaunchKernelGGL(k_wander_windows, dim3(1), dim3(1024), 0, st, ws.list[launch & 1], ws.state, ws.ctl, launch & 3, ws.cap,
                               p->rows, p->cols, ws.wander);
            hipLaunchKernelGGL(k_wander_keys, dim3((slots + kBlock - 1) / kBlock), dim3(kBlock), 0, st, ws.list[launch & 1], ws.state,
                               ws.ctl, launch & 3, ws.cap, ws.wander, k0, sort_is_periodic ? static_cast<uint32_t>(launch) : 0u);
            size_t temp_bytes = ws.sort_temp_bytes;
            if (hipcub::DeviceRadixSort::SortPairs(ws.sort_temp, temp_bytes, k0, k1, ws.list[launch & 1], sorted,
                                                   static_cast<int>(slots), 0, 13, st) != hipSuccess) {
                rc = set_error(SSRS_ERR_HIP, "wander sort failed");
                break;
            }
            // block width of the deal: 512-lane blocks when more tracks are alive than one round of 256-lane blocks holds (216 x 512
            // in one round; beyond that several rounds either way, 512 lanes never slower: profiles/r04_roam_fill.txt; 1024-lane
            // blocks measured 3-4x slower than either -- four waves per SIMD and a launch that ends with its first wave --,
            // SSRS_TRACKS_ROAM_WIDTH=4 keeps the A/B).  The lists' lengths count tombstones and padding, so the width goes by
            // the live tracks k_wander_windows has just counted: one word read back synchronously, and only when the lists are
            // long enough for the question to arise
            roam_width = 1;
            if (roam_ok && deal_contiguous) {
                if (roam_wide_from == 1) {
                    roam_width = 2;
                } else if (roam_wide_from > 1 && static_cast<long long>(prev_total) >= roam_wide_from) {
                    uint32_t *word = &host_counts[kFinalSlot];
                    if (hipMemcpyAsync(word, &ws.ctl->deal_live, sizeof(uint32_t), hipMemcpyDeviceToHost, st) != hipSuccess ||
                        hipStreamSynchronize(st) != hipSuccess) {
                        rc = set_error(SSRS_ERR_HIP, "live-count read-back at the deal failed");
                        break;
                    }
                    if (static_cast<long long>(*word) >= roam_wide_from) roam_width = 2;
                }
                if (roam_width_forced) roam_width = roam_width_forced;
            }
            hipLaunchKernelGGL(k_deal_sorted, dim3(64), dim3(1024), 0, st, k1, sorted, ws.list[(launch + 1) & 1], ws.ctl,
                               (launch + 1) & 3, (launch + 2) & 3, ws.cap, deal_contiguous ? 1 : 0, roam_width);
            ++launch;
            ++wander_sorts;
            want_wander_sort = false;
            want_rebalance = false;
            wander_cooldown = 3;
            stable_roam = sort_is_periodic ? 2 : 0;      // (a shuffle of a settled batch: the launches stay long)
            if (sort_is_periodic) ++roam_shuffles;
            sort_is_periodic = false;
            marks_adjacent = false;
            // the padded deal makes the lists LONGER (each window's run is rounded up to whole blocks of
            // every list): raise the bound now, and let no batch queued before this point lower it
            // (thinned blocks: at most kDealBlocks + one per window + the padding, 264 blocks = 33 per list)
            const unsigned long long wf = static_cast<unsigned long long>(roam_width);    // (wide: runs are whole groups of blocks)
            unsigned long long padded = static_cast<unsigned long long>(upper) + (kWanderWindows + 1ull) * kBlock * wf;
            const unsigned long long kDealPerList = (wf * kDealBlocks + wf * (kWanderWindows + 1)) / kXcd + 3;      // 34 blocks per list (144 rows)
            if (deal_contiguous && padded < kDealPerList * kBlock) padded = kDealPerList * kBlock;
            upper = padded > ws.cap ? ws.cap : static_cast<uint32_t>(padded);
            upper_from = batches;
        }
        if (cached) want_rebalance = false;          // (the lists carry tombstones; the wander sort deals evenly)
        if (cached && roam_ok && !roam_ready) {
            // the batch starts to roam: the pair table, for the whole raster (~2 ms at 5000 x 6000)
            const char *tabc = reinterpret_cast<const char *>(table);
            const unsigned grid = 256 * 16;
            if (roam_rev) hipLaunchKernelGGL(k_roam_build<true>, dim3(grid), dim3(kBlock), 0, st, tabc, a.guard, a.plane_shift, p->rows, p->cols,
                                             ws.roam, thr_prior);
            else hipLaunchKernelGGL(k_roam_build<false>, dim3(grid), dim3(kBlock), 0, st, tabc, a.guard, a.plane_shift, p->rows, p->cols,
                                    ws.roam, thr_prior);
            if (a.fine) {
                FinePrior fp;
                fine_prior_tables(p->prior, &fp);
                const int tx = (p->cols + kTabW - 1) / kTabW, ty = (p->rows + kTabH - 1) / kTabH, nt = tx * ty;
                FineEntry *fine_out = reinterpret_cast<FineEntry *>(const_cast<void *>(a.fine));
                if (potential) hipLaunchKernelGGL(k_fine_build<true>, dim3(static_cast<unsigned>(nt)), dim3(kBlock), 0, st, updraft, potential,
                                                  fine_out, p->rows, p->cols, tx, nt, fp);
                else hipLaunchKernelGGL(k_fine_build<false>, dim3(static_cast<unsigned>(nt)), dim3(kBlock), 0, st, updraft, potential,
                                        fine_out, p->rows, p->cols, tx, nt, fp);
            }
            roam_ready = true;
            marks_adjacent = false;
        }
        if (want_rebalance && launch > 0) {
            // pseudo-launch: list[launch & 1] -> list[(launch + 1) & 1], counts likewise
            hipLaunchKernelGGL(k_rebalance_lists, dim3(1), dim3(1024), 0, st, ws.list[launch & 1], ws.list[(launch + 1) & 1],
                               ws.ctl, launch & 3, (launch + 1) & 3, (launch + 2) & 3, ws.cap);
            marks_adjacent = false;
            ++launch;
            want_rebalance = false;
            rebalance_cooldown = 3;
        }
        // one launch per batch while launches are long and few (the host then sees the batch die one
        // launch earlier: one empty launch at the end of a short run instead of two); two otherwise
        const int depth = (thr && last_Sl >= 512 && launch < 24) ? 1 : kBatch;
        const int slot = batches % kRing;
        bool read_back_done = false;             // the batch's last binning kernel wrote the slot itself
        bool batch_block_window = false;
        for (int j = 0; j < depth; ++j, ++launch) {
            a.launch = launch;
            a.list_in = (launch == 0 && !coherent) ? nullptr : ws.list[launch & 1];
            a.list_out = ws.list[(launch + 1) & 1];
            // threshold table: launch 0 is ONE iteration of the window-gather kernel for every
            // track at once (the first move has eight admissible cells), the rest are S deep
            const bool first_move = thr && launch == 0;
            const unsigned blocks = kXcd * ((upper + kBlock - 1) / kBlock);
            int Sl = first_move ? 1 : S;
            uint32_t vcap_l = ws.cap;
            long long vstride_l = ws.visit_stride;
            if (thr && !first_move && cached && !(rec && rec->complete) && grow_steps) {
                // no visit buffer to fit: only the read-back interval matters.  Once the roam table is in use the
                // launches are long: a launch lasts as long as its slowest wave (a lane on the slow path --
                // near-ties, a track that leaves its region on its way out of the basin -- holds its wave
                // back), and over more steps the waves' slow episodes average out
                Sl = (roam_ok && roam_ready && stable_roam >= 2) ? roam_steps : 8 * S;
            } else if (thr && !first_move && (binning_on || tiles_on) && !(rec && rec->complete) && grow_steps) {
                // Few live tracks left (the long tail of a batch; tracks that wander until max_moves):
                // the visit buffer then holds MORE iterations of the shrunken lists, and a launch of
                // up to 8 S steps amortises the per-launch kernels (binning, read-back) over them
                const uint32_t vc = (blocks / kXcd) * kBlock;                   // slots per list this launch
                const long long room = (ws.visit_stride * kVisitSteps) / (static_cast<long long>(kXcd) * vc);
                long long grown = room < 8ll * S ? room : 8ll * S;
                grown &= ~1ll;
                if (grown >= S + S / 4) {
                    Sl = static_cast<int>(grown);
                    vcap_l = vc;
                    vstride_l = static_cast<long long>(kXcd) * vc;
                }
            }
            if (std::getenv("SSRS_TRACKS_DEBUG") && (launch < 40 || launch % 500 == 0))
                fprintf(stderr, "[tracks] launch %d upper %u blocks %u Sl %d binning %d tiles %d scattered %d cached %d cap %u\n", launch, upper, blocks, Sl,
                        binning_on ? 1 : 0, tiles_on ? 1 : 0, scattered ? 1 : 0, cached ? 1 : 0, ws.cap);
            a.steps = Sl;
            a.coherent = (coherent && !first_move) ? 1 : 0;
            a.ordered = (first_move && a.pf_dir != 0 && lds_rows && p->cols >= kLrCols) ? 1 : 0;
            a.it_base = thr && launch > 0 ? it_done : 0;
            a.visits = nullptr;
            if (!binning_on && scattered && copies_ptr && !copies_live && !cached) {
                // first scattered launch: zero the private copies, count into them from now on
                if (hipMemsetAsync(copies_ptr, 0, sizeof(uint32_t) * ncell * ncopies, st) != hipSuccess) {
                    rc = set_error(SSRS_ERR_HIP, "histogram copies memset failed");
                    break;
                }
                copies_live = true;
                marks_adjacent = false;
                a.hist_copies = copies_ptr;
                a.ncopies = ncopies;
            }
            // the first-move launch is one iteration deep: its visits go straight to the histogram
            // (one binning block for the whole batch took 170 us); generic kernel, plain keys
            const bool bin_window = binning_on && !first_move, bin_tiles = tiles_on && !first_move;
            const uint32_t keep_r = a.vis_r, keep_c = a.vis_c;
            if (first_move) { a.vis_r = static_cast<uint32_t>(p->cols); a.vis_c = 1u; }
            if (bin_window || bin_tiles) a.visits = ws.visits;
            a.visit_stride = vstride_l;
            a.vcap = vcap_l;
            const uint32_t *rec_counts = nullptr;
            if (rec && rec->complete) {
                // this launch's own region of the pool: [counts][slot -> track list][visits]
                const uint32_t vcap = (blocks / kXcd) * kBlock;
                const bool identity = a.list_in == nullptr;
                const size_t list_bytes = identity ? 0 : align_up(sizeof(int32_t) * kXcd * static_cast<size_t>(vcap), 256);
                const size_t vis_bytes = align_up(sizeof(uint32_t) * kXcd * static_cast<size_t>(vcap) * static_cast<size_t>(Sl), 256);
                const size_t need = 256 + list_bytes + vis_bytes;
                if (rec->bytes - rec->used < need) {
                    rec->complete = 0;           // pool exhausted: the rest of the run is not recorded
                } else {
                    char *base = rec->pool + rec->used;
                    rec->used += need;
                    TrajChunk ch = {};
                    ch.counts = reinterpret_cast<const uint32_t *>(base);
                    ch.list = identity ? nullptr : reinterpret_cast<const int32_t *>(base + 256);
                    ch.visits = reinterpret_cast<const uint32_t *>(base + 256 + list_bytes);
                    ch.vcap = vcap;
                    ch.cap = ws.cap;
                    ch.steps = Sl;
                    ch.transposed = (hist_t && bin_window) ? 1 : 0;
                    hipError_t e1 = hipMemcpyAsync(base, ws.ctl->count[launch & 3], kXcd * sizeof(uint32_t),
                                                   hipMemcpyDeviceToDevice, st);
                    hipError_t e2 = identity ? hipSuccess
                                             : hipMemcpy2DAsync(base + 256, sizeof(int32_t) * vcap, a.list_in,
                                                                sizeof(int32_t) * ws.cap, sizeof(int32_t) * vcap, kXcd,
                                                                hipMemcpyDeviceToDevice, st);
                    if (e1 != hipSuccess || e2 != hipSuccess) { rc = set_error(SSRS_ERR_HIP, "trajectory record copy failed"); break; }
                    rec->chunks.push_back(ch);
                    marks_adjacent = false;
                    a.visits = const_cast<uint32_t *>(ch.visits);
                    a.visit_stride = static_cast<long long>(kXcd) * vcap;
                    a.vcap = vcap;
                    rec_counts = ch.counts;
                }
            }
            if (profile && !marks_adjacent) mark(0);
            bool is_block_window = false;
            // 16-bit visit keys: north-bound front through the row window, nothing recorded
            // (the stepper forms the key base (first start row + iteration - 1) * cols in 32 bits)
            const bool v16 = thr && bin_window && a.visits == ws.visits && a.pf_dir == 1 && !hist_t && v16_ok &&
                             (it_done + Sl + 2ll * geom.offset) * p->cols < (1ll << 31);
            switch (first_move ? mode0 : mode) {
            case MODE_TABLE:
                if (thr) {
                    // front-shaped batches heading north / south get the prefetch wave; once the front has
                    // outgrown the row window (tile buckets) it streams rows nobody reads: 7.40 -> 7.06 s
                    // per 100k tracks on the solved 10 m field without it
                    const bool pf = a.pf_dir != 0 && a.coherent && !scattered && !tiles_on;
                    // (staged rows: opt-in switch SSRS_TRACKS_LDS_ROWS; the window is kLrCols wide)
                    const bool lr = pf && p->cols >= kLrCols && lds_rows;
                    if (v16) {
                        if (lr) hipLaunchKernelGGL((k_step_thr<4, true, false, true>), dim3(blocks), dim3(kBlock + 64), 0, st, a, thr_prior);
                        else hipLaunchKernelGGL((k_step_thr<4, true>), dim3(blocks), dim3(kBlock + 64), 0, st, a, thr_prior);
                        break;
                    }
                    const bool rev = thr_prior.rev_ok != 0 && std::getenv("SSRS_TRACKS_NO_REV") == nullptr;
                    if (cached && !a.visits) {
                        ++block_window_launches;
                        batch_block_window = is_block_window = true;
                        if (roam_ok && roam_ready) {
                            ++roam_launches;
                            if (roam_width > 1) {
                                const unsigned bt = static_cast<unsigned>(roam_width) * kBlock;
                                const unsigned wblocks = kXcd * ((upper + bt - 1) / bt);
                                ++roam_wide_launches;
                                if (roam_width == 2) {
                                    if (rev) hipLaunchKernelGGL((k_step_roam<true, 2 * kBlock>), dim3(wblocks), dim3(bt), 0, st, a, thr_prior);
                                    else hipLaunchKernelGGL((k_step_roam<false, 2 * kBlock>), dim3(wblocks), dim3(bt), 0, st, a, thr_prior);
                                } else {
                                    if (rev) hipLaunchKernelGGL((k_step_roam<true, 4 * kBlock>), dim3(wblocks), dim3(bt), 0, st, a, thr_prior);
                                    else hipLaunchKernelGGL((k_step_roam<false, 4 * kBlock>), dim3(wblocks), dim3(bt), 0, st, a, thr_prior);
                                }
                            } else if (rev) hipLaunchKernelGGL(k_step_roam<true>, dim3(blocks), dim3(kBlock), 0, st, a, thr_prior);
                            else hipLaunchKernelGGL(k_step_roam<false>, dim3(blocks), dim3(kBlock), 0, st, a, thr_prior);
#ifdef SSRS_DEBUG_WAVE_DUMP
                            if (a.debug_roam && launch == SSRS_DEBUG_WAVE_DUMP && a.dbg_buf) {
                                std::vector<unsigned long long> rec(8ull * blocks * (kBlock / 64));
                                (void)hipStreamSynchronize(st);
                                (void)hipMemcpy(rec.data(), a.dbg_buf, rec.size() * 8, hipMemcpyDeviceToHost);
                                // block, wave, live lanes, clocks, pairs, slow pairs, strays, window origin, fast lanes
                                for (uint32_t w = 0; w < blocks * (kBlock / 64); ++w) {
                                    const unsigned long long *r = &rec[8ull * w];
                                    fprintf(stderr, "W %u %u %llu %llu %llu %llu %llu %lld %lld %llu\n", w / (kBlock / 64), w % (kBlock / 64), r[1], r[0], r[2], r[3], r[4],
                                            static_cast<long long>(r[5]), static_cast<long long>(r[6]), r[7]);
                                }
                            }
#endif
                            if (a.debug_roam) {
                                // diagnostics only: one synchronous read of the control block per launch
                                TrackCtl c;
                                (void)hipMemcpyAsync(&host_counts[kFinalSlot], ws.ctl, sizeof(TrackCtl), hipMemcpyDeviceToHost, st);
                                (void)hipStreamSynchronize(st);
                                memcpy(&c, &host_counts[kFinalSlot], sizeof(TrackCtl));
                                if (c.dbg_waves)
                                    fprintf(stderr, "[roam] launch %d blocks %u Sl %d: %llu waves, mean %.0f clk, max %.0f clk (x%.2f); slowest-by-slow-pairs wave: %llu of %llu pairs slow\n",
                                            launch, blocks, Sl, c.dbg_waves, static_cast<double>(c.dbg_tsum) / c.dbg_waves, static_cast<double>(c.dbg_tmax),
                                            static_cast<double>(c.dbg_tmax) * c.dbg_waves / static_cast<double>(c.dbg_tsum), c.dbg_slowmax >> 32, c.dbg_slowmax & 0xFFFFFFFFull);
                                (void)hipMemsetAsync(&ws.ctl->dbg_tsum, 0, 4 * sizeof(unsigned long long), st);
                            }
                            break;
                        }
                        if (rev) hipLaunchKernelGGL((k_step_thr<6, false, true>), dim3(blocks), dim3(kBlock), 0, st, a, thr_prior);
                        else hipLaunchKernelGGL((k_step_thr<6>), dim3(blocks), dim3(kBlock), 0, st, a, thr_prior);
                        break;
                    }
                    if (a.visits && hist_t && binning_on) hipLaunchKernelGGL((k_step_thr<2>), dim3(blocks), dim3(kBlock), 0, st, a, thr_prior);
                    else if (a.visits && lr) hipLaunchKernelGGL((k_step_thr<1, true, false, true>), dim3(blocks), dim3(kBlock + 64), 0, st, a, thr_prior);
                    else if (a.visits && pf) hipLaunchKernelGGL((k_step_thr<1, true>), dim3(blocks), dim3(kBlock + 64), 0, st, a, thr_prior);
                    else if (a.visits && rev) hipLaunchKernelGGL((k_step_thr<1, false, true>), dim3(blocks), dim3(kBlock), 0, st, a, thr_prior);
                    else if (a.visits) hipLaunchKernelGGL((k_step_thr<1>), dim3(blocks), dim3(kBlock), 0, st, a, thr_prior);
                    else if (a.hist && pf) hipLaunchKernelGGL((k_step_thr<3, true>), dim3(blocks), dim3(kBlock + 64), 0, st, a, thr_prior);
                    else if (a.hist && rev) hipLaunchKernelGGL((k_step_thr<3, false, true>), dim3(blocks), dim3(kBlock), 0, st, a, thr_prior);
                    else if (a.hist) hipLaunchKernelGGL((k_step_thr<3>), dim3(blocks), dim3(kBlock), 0, st, a, thr_prior);
                    else if (pf) hipLaunchKernelGGL((k_step_thr<0, true>), dim3(blocks), dim3(kBlock + 64), 0, st, a, thr_prior);
                    else if (rev) hipLaunchKernelGGL((k_step_thr<0, false, true>), dim3(blocks), dim3(kBlock), 0, st, a, thr_prior);
                    else hipLaunchKernelGGL((k_step_thr<0>), dim3(blocks), dim3(kBlock), 0, st, a, thr_prior);
                    break;
                }
                // (the zero-mask variant only pays under in-stepper atomics: with tile buckets it
                // was measured slower, 7.0 -> 8.3 s per 100k wandering tracks)
                if (ring && scattered && !binning_on && !tiles_on) hipLaunchKernelGGL((k_step_lean<true, true>), dim3(blocks), dim3(kBlock), 0, st, a);
                else if (ring && hist_t && binning_on) hipLaunchKernelGGL((k_step_lean<true, false, true>), dim3(blocks), dim3(kBlock), 0, st, a);
                else if (ring) hipLaunchKernelGGL((k_step_lean<true, false>), dim3(blocks), dim3(kBlock), 0, st, a);
                else if (lean && a.fast && (S & 1) == 0 && hist_t && binning_on) hipLaunchKernelGGL((k_step_lean<false, false, true>), dim3(blocks), dim3(kBlock), 0, st, a);
                else if (lean && a.fast && (S & 1) == 0) hipLaunchKernelGGL((k_step_lean<false, false>), dim3(blocks), dim3(kBlock), 0, st, a);
                else hipLaunchKernelGGL(k_step_tracks<MODE_TABLE>, dim3(blocks), dim3(kBlock), 0, st, a);
                break;
            case MODE_FLUIDFLOW: hipLaunchKernelGGL(k_step_tracks<MODE_FLUIDFLOW>, dim3(blocks), dim3(kBlock), 0, st, a); break;
            case MODE_UPDRAFT: hipLaunchKernelGGL(k_step_tracks<MODE_UPDRAFT>, dim3(blocks), dim3(kBlock), 0, st, a); break;
            default: hipLaunchKernelGGL(k_step_tracks<MODE_PRIOR>, dim3(blocks), dim3(kBlock), 0, st, a); break;
            }
            if (profile) mark(is_block_window ? 3 : 1);      // end of the stepper launch
            if (bin_window) {
                ++window_launches;
                if (v16) {
                    uint32_t *out = (j == depth - 1 && direct_read_back) ? &host_counts[kSlotWords * slot] : nullptr;
                    hipLaunchKernelGGL(k_bin_visits16, dim3((Sl + 1) / 2), dim3(kBinThreads), 0, st, reinterpret_cast<const uint16_t *>(a.visits),
                                       a.visit_stride, Sl, ws.ctl, launch & 3, hist, p->rows, p->cols, a.vcap, a.v16_offset, a.it_base,
                                       out, kSlotWords);
                    read_back_done = out != nullptr;
                }
                else if (hist_t)
                    hipLaunchKernelGGL(k_bin_visits, dim3(Sl), dim3(kBinThreads), 0, st, a.visits, a.visit_stride,
                                       ws.ctl, launch & 3, hist_t, p->cols, p->rows, a.vcap);
                else
                    hipLaunchKernelGGL(k_bin_visits, dim3(Sl), dim3(kBinThreads), 0, st, a.visits, a.visit_stride,
                                       ws.ctl, launch & 3, hist, p->rows, p->cols, a.vcap);
                if (profile && !read_back_done) mark(2);      // (else the batch's event below is this mark)
            }
            if (bin_tiles) {
                ++tile_launches;
                const double inv_cols = 1.0 / static_cast<double>(p->cols);
                const uint32_t ucols = static_cast<uint32_t>(p->cols), ucell = static_cast<uint32_t>(ncell);
                (void)hipMemsetAsync(ws.tile_count, 0, sizeof(uint32_t) * ntiles, st);
                hipLaunchKernelGGL((k_tile_sort<false>), dim3(blocks, kStepSplit), dim3(kBlock), 0, st, a.visits, a.visit_stride, Sl,
                                   ws.ctl, launch & 3, ucols, inv_cols, ucell, a.vcap, ntc, ntiles, ws.tile_count,
                                   ws.tile_cursor, ws.bucket);
                hipLaunchKernelGGL(k_tile_scan, dim3(1), dim3(kTileThreads), 0, st, ws.tile_count, ntiles, ws.tile_start,
                                   ws.tile_cursor, ws.item_start);
                hipLaunchKernelGGL((k_tile_sort<true>), dim3(blocks, kStepSplit), dim3(kBlock), 0, st, a.visits, a.visit_stride, Sl,
                                   ws.ctl, launch & 3, ucols, inv_cols, ucell, a.vcap, ntc, ntiles, ws.tile_count,
                                   ws.tile_cursor, ws.bucket);
                // at most S visits per slot of the launch, and one partly filled item per tile
                const unsigned long long max_visits = static_cast<unsigned long long>(blocks) * kBlock * Sl;
                const unsigned items = static_cast<unsigned>(max_visits / kItemVisits) + ntiles;
                hipLaunchKernelGGL(k_bin_bucket, dim3(items), dim3(kTileThreads), 0, st, ws.bucket, ws.tile_start,
                                   ws.tile_count, ws.ctl, hist, static_cast<uint32_t>(p->rows), ucols, inv_cols, ntc,
                                   ntiles, ws.item_start);
                if (profile) mark(2);
            }
            if (rec_counts && hist && !bin_window && !bin_tiles) {    // recorded launch outside both binning paths
                hipLaunchKernelGGL(k_count_visits, dim3(blocks), dim3(kBlock), 0, st, a.visits, a.vcap, Sl, rec_counts, hist,
                                   static_cast<uint32_t>(ncell));
                marks_adjacent = false;
            }
            a.vis_r = keep_r;
            a.vis_c = keep_c;
            if (thr && !first_move) it_done += Sl;
            last_Sl = Sl;
            if (hipGetLastError() != hipSuccess) { rc = set_error(SSRS_ERR_HIP, "stepper launch failed"); break; }
        }
        if (rc != SSRS_OK) break;
        if (hist64 && cached && (batches & 1)) {          // (block windows flush whole launches' counts at once)
            hipLaunchKernelGGL(k_drain64, dim3(4096), dim3(kBlock), 0, st, hist, hist64, ncell);
            marks_adjacent = false;
        }
        // survivors of this batch = input count of the next launch
        // ring slot = the head of the control block in one copy: [4][8] list counts, error, par_min,
        // steps (2 words), strays (2 words); the row this batch's survivors went to is count[launch & 3]
        bool queued = read_back_done || hipMemcpyAsync(&host_counts[kSlotWords * slot], ws.ctl, kSlotWords * sizeof(uint32_t),
                                                       hipMemcpyDeviceToHost, st) == hipSuccess;
        if (queued && profile) {
            const size_t before = ev_marks.size();
            mark(read_back_done ? 2 : 0);
            queued = ev_marks.size() > before;
            if (queued) ev_batch[slot] = ev_marks.back();
        } else if (queued) {
            queued = hipEventRecord(ev_batch[slot], st) == hipSuccess;
            marks_adjacent = false;
        }
        if (!queued) {
            rc = set_error(SSRS_ERR_HIP, "live-count read-back failed");
            break;
        }
        slot_row[slot] = launch & 3;
        slot_block_window[slot] = batch_block_window;
        ++batches;
        // examine every batch but the one just queued (it keeps the GPU busy)
        while (checked < batches - 1) {
            const int cs = checked % kRing;
            if (hipEventSynchronize(ev_batch[cs]) != hipSuccess) { rc = set_error(SSRS_ERR_HIP, "event sync failed"); break; }
            uint32_t c = 0;                         // longest list
            const uint32_t *cnt = &host_counts[kSlotWords * cs + kXcd * slot_row[cs]];
            for (int x = 0; x < kXcd; ++x) c = cnt[x] > c ? cnt[x] : c;
            unsigned long long tot[2];
            memcpy(tot, &host_counts[kSlotWords * cs + offsetof(TrackCtl, steps) / sizeof(uint32_t)], sizeof(tot));
            ++checked;
            if (slot_block_window[cs]) block_window_steps += static_cast<long long>(tot[0] - seen_steps);
            if (c == 0) { finished = true; break; }
            // the live count only shrinks, a stale bound is safe -- except across a wander sort
            upper = (checked - 1 >= upper_from || c > upper) ? c : upper;
            uint32_t total = 0;
            for (int x = 0; x < kXcd; ++x) total += cnt[x];
            if (tiles_on && cache_ok && !force_tiles && a.pf_dir != 0) {
                // a front that outgrew the row window goes through tile buckets while part of the batch
                // still travels; once nobody finishes any more (the survivors roam their basins until
                // max_moves) the block windows take over
                // (nobody finishing YET is not stable: some must have finished, or the batch is older
                // than two raster crossings)
                const bool started = static_cast<long long>(total) * 50 < static_cast<long long>(ntracks) * 49 ||
                                     it_done > 2ll * (p->rows + p->cols);
                if (started && prev_total != 0 && total >= prev_total - prev_total / 32 && ++stable_batches >= 2) {
                    tiles_on = false;
                    cached = true;
                    want_wander_sort = true;
                }
                if (prev_total == 0 || total < prev_total - prev_total / 32) stable_batches = 0;
            }
            prev_total = total;
            if (thr && may_rebalance) {
                if (rebalance_cooldown > 0) --rebalance_cooldown;
                else if (c >= 1024 && 5ull * c >= static_cast<unsigned long long>(total) + 64ull) want_rebalance = true;   // longest list >= 1.6 x the mean
            }
            // binning pays only while the batch moves as a front: once more than a
            // quarter of a batch's visits miss the LDS window, later launches go back
            // to in-stepper atomics
            if (binning_on || tiles_on) {
                // row window: strays = visits outside it (stop above a quarter); tiles:
                // strays = cells flushed (stop below two visits per cell)
                // (batches queued before a switch still report the old path's strays)
                const unsigned long long dsteps = tot[0] - seen_steps, dstray = tot[1] - seen_strays;
                if (checked - 1 >= judge_from && dsteps > 0 && dstray * (tiles_on ? 2 : 4) > dsteps && !force_tiles) {
                    judge_from = batches;
                    if (binning_on && cache_ok && !tiles_ok) {
                        // the front has outgrown the row window and there are no tile buckets
                        binning_on = false;
                        cached = true;
                        want_wander_sort = true;
                        a.vis_r = static_cast<uint32_t>(p->cols);
                        a.vis_c = 1u;
                    } else if (binning_on && tiles_ok) {
                        // the front has outgrown the row window; its visits may still cluster
                        binning_on = false;
                        tiles_on = true;
                        a.vis_r = static_cast<uint32_t>(p->cols);      // plain visit keys from now on
                        a.vis_c = 1u;
                    } else {
                        binning_on = tiles_on = false;
                        a.vis_r = static_cast<uint32_t>(p->cols);
                        a.vis_c = 1u;
                        scattered = !never_scattered;      // no front any more: zero-mask variant
                        cached = cache_ok && scattered;
                        want_wander_sort = cached;
                    }
                }
            }
            if (cached && !(binning_on || tiles_on)) {
                // block windows: strays = visits outside them.  Tracks still on their way into a basin
                // (or out of their block's box) show up here: sort again, a few times at most
                const unsigned long long dsteps = tot[0] - seen_steps, dstray = tot[1] - seen_strays;
                if (std::getenv("SSRS_TRACKS_DEBUG") && checked < 60)
                    fprintf(stderr, "[tracks] batch %d block windows: %llu steps, %llu strays (%.3f), live %u\n", checked, dsteps, dstray,
                            dsteps ? static_cast<double>(dstray) / static_cast<double>(dsteps) : 0.0, total);
                if (wander_cooldown > 0) --wander_cooldown;
                else if (dsteps > 0 && dstray * 64 > dsteps && wander_sorts < 12) want_wander_sort = true;
                else if (roam_shuffle > 0 && ++since_shuffle >= roam_shuffle && stable_roam >= 2 && roam_ready) {
                    want_wander_sort = true;            // settled: every roam_shuffle batches the windows' tracks are dealt afresh
                    sort_is_periodic = true;
                    since_shuffle = 0;
                } else ++stable_roam;                   // settled in its windows: the launches may grow
            }
            // batches that never binned (small, unsorted, very wide rasters) give no stray
            // signal: tracks still alive after four raster crossings are wandering
            if (!binning_on && !tiles_on && !scattered && !never_scattered &&
                (thr ? it_done : static_cast<long long>(launch) * S) > 4ll * (p->rows + p->cols)) {
                scattered = true;
                if (cache_ok && !cached) { cached = true; want_wander_sort = true; }
            }
            seen_steps = tot[0];
            seen_strays = tot[1];
        }
    }
    if (copies_live && rc == SSRS_OK)
        hipLaunchKernelGGL(k_fold_copies, dim3(4096), dim3(kBlock), 0, st, copies_ptr, ncopies, ncell, hist);
    if (hist_t && rc == SSRS_OK)
        hipLaunchKernelGGL(k_transpose_add, dim3(static_cast<unsigned>(((p->rows + 31) / 32) * ((p->cols + 31) / 32))),
                           dim3(kBlock), 0, st, hist_t, p->rows, p->cols, hist);
    if (hist64 && rc == SSRS_OK) hipLaunchKernelGGL(k_drain64, dim3(4096), dim3(kBlock), 0, st, hist, hist64, ncell);
    (void)hipEventRecord(ev_last, st);
    // fetch step total + error flag
    TrackCtl host_ctl = {};
    if (rc == SSRS_OK) {
        if (hipMemcpyAsync(&host_counts[kFinalSlot], ws.ctl, sizeof(TrackCtl), hipMemcpyDeviceToHost, st) != hipSuccess ||
            hipStreamSynchronize(st) != hipSuccess)
            rc = set_error(SSRS_ERR_HIP, "final read-back failed");
        else
            memcpy(&host_ctl, &host_counts[kFinalSlot], sizeof(TrackCtl));
    } else {
        (void)hipStreamSynchronize(st);
    }
    if (a.debug_roam && rc == SSRS_OK && roam_launches == 0 && host_ctl.dbg_waves)
        fprintf(stderr, "[front] staged rows: %llu wave-steps, %llu of them fell back to the gather (%.4f); lane-steps: outside window / plane %llu, "
                        "slot empty %llu, slot holds another row %llu; outside by its plane (of the first) %llu\n", host_ctl.dbg_waves,
                host_ctl.dbg_tsum, static_cast<double>(host_ctl.dbg_tsum) / static_cast<double>(host_ctl.dbg_waves),
                host_ctl.dbg_tmax >> 32, host_ctl.dbg_tmax & 0xFFFFFFFFull, host_ctl.dbg_slowmax >> 32, host_ctl.dbg_slowmax & 0xFFFFFFFFull);
    if (a.debug_roam && rc == SSRS_OK && roam_launches == 0 && host_ctl.roam_pairs)
        fprintf(stderr, "[front] staging waves: %llu batches, %llu polls that found the ring full, %llu rows, %.3e clocks of lifetime in all\n",
                host_ctl.roam_pairs >> 32, host_ctl.roam_pairs & 0xFFFFFFFFull, host_ctl.roam_slow >> 32,
                static_cast<double>(host_ctl.roam_slow & 0xFFFFFFFFull) * 256.0);
    if (a.debug_roam && rc == SSRS_OK && roam_launches == 0 && (host_ctl.dbg_span & 0xFFFFFull))
        fprintf(stderr, "[front] rows between the first and the last track of a block's front (by the row each would be on at iteration 0): "
                        "mean %.1f over %llu blocks, max %llu; the ring holds %d; %llu polls of stepping waves that waited for their row\n",
                static_cast<double>(host_ctl.dbg_span >> 20) / static_cast<double>(host_ctl.dbg_span & 0xFFFFFull),
                host_ctl.dbg_span & 0xFFFFFull, host_ctl.dbg_span_max, kLrRows, host_ctl.dbg_waits);
    if (stats && rc == SSRS_OK) {
        stats->total_steps = static_cast<int64_t>(host_ctl.steps);
        stats->launches = launch;
        stats->window_launches = window_launches;
        stats->tile_launches = tile_launches;
        stats->block_window_launches = block_window_launches;
        stats->wander_sorts = wander_sorts;
        stats->roam_launches = roam_launches;
        stats->roam_shuffles = roam_shuffles;
        stats->roam_wide_launches = roam_wide_launches;
        stats->roam_wave_pairs = static_cast<int64_t>(host_ctl.roam_pairs);
        stats->roam_slow_wave_pairs = static_cast<int64_t>(host_ctl.roam_slow);
        stats->reserved0 = static_cast<int32_t>(host_ctl.pad);                   // near-ties settled by the fine table
        // (batches still unexamined when the loop ended: the last one or two of the run)
        stats->block_window_steps = block_window_steps;
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, ev_first, ev_last) == hipSuccess) stats->wall_ms = ms;
        if (profile) {
            float sum = 0.f, hsum = 0.f;
            int timed = 0;
            for (size_t i = 1; i < ev_marks.size(); ++i) {
                if (mark_kind[i] == 0 || hipEventElapsedTime(&ms, ev_marks[i - 1], ev_marks[i]) != hipSuccess) continue;
                if (mark_kind[i] == 1 || mark_kind[i] == 3) {
                    sum += ms;
                    if (timed == 0 && thr) stats->first_move_ms = ms;      // (launch 0 of a threshold-table call)
                    ++timed;
                    if (mark_kind[i] == 3) { stats->block_window_ms += ms; ++stats->block_window_timed; }
                } else {
                    hsum += ms;
                }
            }
            stats->timed_launches = timed;
            stats->kernel_ms = sum;
            stats->hist_ms = hsum;
        }
    }
    for (hipEvent_t e : ev_marks) (void)hipEventDestroy(e);
    if (!profile)
        for (int i = 0; i < kRing; ++i) (void)hipEventDestroy(ev_batch[i]);
    (void)hipEventDestroy(ev_first);
    (void)hipEventDestroy(ev_last);
    if (rc != SSRS_OK) return rc;
    if (host_ctl.error & 2u)
        return set_error(SSRS_ERR_INVALID, "ssrs_tracks_simulate: `table` is not a threshold table built by "
                         "ssrs_transition_thr_build for this %d x %d raster and params->prior (results discarded)",
                         p->rows, p->cols);
    if (host_ctl.error)
        return set_error(SSRS_ERR_START, "ssrs_tracks_simulate: a start cell lies outside the %d x %d raster",
                         p->rows, p->cols);
    return SSRS_OK;
}

extern "C" int ssrs_tracks_simulate(const SsrsTrackParams *p, const double *updraft,
                                    const float *potential, const double *table,
                                    const int32_t *start_rc, int64_t ntracks, uint64_t seed,
                                    uint64_t track_id_base, uint32_t *hist, int16_t *end_rc,
                                    int32_t *lengths, int16_t *traj, const int64_t *traj_offsets,
                                    void *workspace, size_t workspace_bytes,
                                    SsrsTrackStats *stats, void *stream)
{
    return tracks_simulate_impl(p, updraft, potential, table, start_rc, ntracks, seed, track_id_base, hist, end_rc,
                                lengths, traj, traj_offsets, workspace, workspace_bytes, stats, stream, nullptr);
}

extern "C" int ssrs_tracks_simulate_h64(const SsrsTrackParams *p, const double *updraft,
                                        const float *potential, const double *table,
                                        const int32_t *start_rc, int64_t ntracks, uint64_t seed,
                                        uint64_t track_id_base, uint32_t *hist_scratch, uint64_t *hist64, int16_t *end_rc,
                                        int32_t *lengths, void *workspace, size_t workspace_bytes,
                                        SsrsTrackStats *stats, void *stream)
{
    SSRS_REQUIRE(hist_scratch != nullptr && hist64 != nullptr, "ssrs_tracks_simulate_h64: NULL histogram");
    static_assert(sizeof(unsigned long long) == sizeof(uint64_t), "64-bit counts");
    return tracks_simulate_impl(p, updraft, potential, table, start_rc, ntracks, seed, track_id_base, hist_scratch, end_rc,
                                lengths, nullptr, nullptr, workspace, workspace_bytes, stats, stream, nullptr,
                                reinterpret_cast<unsigned long long *>(hist64));
}

extern "C" SsrsTrajRecorder *ssrs_traj_recorder_create(void *pool, size_t pool_bytes)
{
    if (!pool || (reinterpret_cast<uintptr_t>(pool) & 255u)) {
        set_error(SSRS_ERR_INVALID, "ssrs_traj_recorder_create: pool must be non-NULL and 256-byte aligned");
        return nullptr;
    }
    SsrsTrajRecorder *rec = new (std::nothrow) SsrsTrajRecorder();
    if (!rec) { set_error(SSRS_ERR_INVALID, "ssrs_traj_recorder_create: out of host memory"); return nullptr; }
    rec->pool = static_cast<char *>(pool);
    rec->bytes = pool_bytes;
    rec->used = 0;
    rec->complete = 0;
    rec->rows = rec->cols = 0;
    rec->ntracks = -1;
    return rec;
}

extern "C" void ssrs_traj_recorder_destroy(SsrsTrajRecorder *rec) { delete rec; }

extern "C" int ssrs_traj_recorder_complete(const SsrsTrajRecorder *rec) { return rec ? rec->complete : 0; }

extern "C" size_t ssrs_traj_recorder_used(const SsrsTrajRecorder *rec) { return rec ? rec->used : 0; }

extern "C" int ssrs_tracks_simulate_rec(const SsrsTrackParams *p, const double *updraft,
                                        const float *potential, const double *table,
                                        const int32_t *start_rc, int64_t ntracks, uint64_t seed,
                                        uint64_t track_id_base, uint32_t *hist, int16_t *end_rc,
                                        int32_t *lengths, SsrsTrajRecorder *recorder,
                                        void *workspace, size_t workspace_bytes,
                                        SsrsTrackStats *stats, void *stream)
{
    SSRS_REQUIRE(recorder != nullptr, "ssrs_tracks_simulate_rec: recorder is NULL");
    return tracks_simulate_impl(p, updraft, potential, table, start_rc, ntracks, seed, track_id_base, hist, end_rc,
                                lengths, nullptr, nullptr, workspace, workspace_bytes, stats, stream, recorder);
}

extern "C" int ssrs_tracks_gather(const SsrsTrajRecorder *rec, const int32_t *start_rc, int64_t ntracks,
                                  const int64_t *traj_offsets, int16_t *traj, void *cursor_ws,
                                  size_t cursor_bytes, void *stream)
{
    SSRS_REQUIRE(rec != nullptr, "ssrs_tracks_gather: recorder is NULL");
    SSRS_REQUIRE(rec->complete, "ssrs_tracks_gather: the record is incomplete (pool exhausted or no simulation yet)");
    SSRS_REQUIRE(ntracks == rec->ntracks, "ssrs_tracks_gather: ntracks differs from the recorded simulation");
    if (ntracks == 0) return SSRS_OK;
    SSRS_REQUIRE(start_rc && traj_offsets && traj && cursor_ws, "ssrs_tracks_gather: NULL pointer");
    SSRS_REQUIRE(cursor_bytes >= sizeof(uint32_t) * static_cast<size_t>(ntracks),
                 "ssrs_tracks_gather: cursor scratch too small (4 bytes per track)");
    hipStream_t st = as_stream(stream);
    uint32_t *cursor = static_cast<uint32_t *>(cursor_ws);
    const long long *off = reinterpret_cast<const long long *>(traj_offsets);
    hipLaunchKernelGGL(k_gather_init, dim3(static_cast<unsigned>((ntracks + kBlock - 1) / kBlock)), dim3(kBlock), 0, st,
                       start_rc, static_cast<long long>(ntracks), off, traj, cursor);
    for (const TrajChunk &ch : rec->chunks)
        hipLaunchKernelGGL(k_gather_chunk, dim3(kXcd * (ch.vcap / kBlock)), dim3(kBlock), 0, st, ch,
                           static_cast<uint32_t>(rec->rows), static_cast<uint32_t>(rec->cols), off, traj, cursor);
    SSRS_HIP_CHECK(hipGetLastError());
    return SSRS_OK;
}
