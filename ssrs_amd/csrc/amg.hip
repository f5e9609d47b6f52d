// K5 preconditioner -- aggregation algebraic multigrid for the potential system.
//
// Why: the fluid-flow operator of ssrs/movmodel.py:87-128 links cells by the
// harmonic mean of their updrafts, which spans ten orders of magnitude (1e-8
// for zero-updraft cells, 2e-10 .. 5 elsewhere).  Conductive clusters that float
// in dead terrain give eigenvalues ~1e-8; Krylov methods without a coarse space
// stall (DESIGN.md "K5").  Geometric coarsening fails for the same reason
// (tests/dev/amg_experiment2.py); aggregation along the STRONG couplings works
// (tests/dev/amg_experiment4.py) because a floating cluster collapses into few
// coarse nodes whose level the coarse problem determines directly.
//
// Method (all on the device, deterministic -- no float atomics):
//   level 0   CSR of the symmetric operator L = D - C on the raster numbering
//             (Dirichlet cells = isolated identity rows), natural 8-neighbour
//             weights (the reference's east-edge quirk lives only in the outer
//             Krylov operator, potential.hip)
//   coarsen   pairwise aggregation: every unmatched node proposes to its
//             strongest unmatched neighbour that is SYMMETRICALLY strong
//             (a_ij^2 >= (0.25/8)^2 a_ii a_jj: never across the live/dead phases;
//             ties broken by a symmetric hash), mutual proposals pair up, 8
//             rounds; leftover nodes join the aggregate of their best matched
//             neighbour, any number per aggregate (each node decides alone:
//             order independent); once strict matching stalls, any positive
//             coupling may pair
//   Galerkin  P^T A P by radix-sorting (I,J) keys + reduce-by-key (hipCUB)
//   cycle     V(1,1) since round 4 (x = w D^-1 b, coarse correction, one sweep; V(2,2) in rounds 1-3 and as BiCGStab's
//             fall-back), damped Jacobi (omega 0.7), level 0 in two fused stencil passes on the caller's vectors, dense
//             inverse on the coarsest level (<= 1024 nodes, Gauss-Jordan on the device); profiles/r04_k5.md has what else
//             was measured (black-box MG, smoothed aggregation, K-cycles, an f32 cycle)
#include <hipcub/hipcub.hpp>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "amg.h"

namespace ssrs {

namespace {

constexpr double kOmega = 0.7;
constexpr double kTheta = 0.25;
constexpr int kMatchRounds = 8;
constexpr int kMaxDense = 1024;

struct Bump {
    char *base;
    size_t cap, off;
    void *take(size_t bytes)
    {
        const size_t a = (off + 255) / 256 * 256;
        if (a + bytes > cap) return nullptr;
        off = a + bytes;
        return base + a;
    }
};

constexpr int kVectorRows = 1 << 17;     // levels at least this large use four lanes per row (CSR sweeps)
constexpr int kSellRows = 1 << 17;       // ... and get a sliced ELL copy for the cycle's sweeps

inline int grid_for(size_t n)
{
    size_t b = (n + kBlock - 1) / kBlock;
    if (b < 1) b = 1;
    if (b > (1u << 20)) b = 1u << 20;
    return static_cast<int>(b);
}

__device__ __forceinline__ double pair_weight(double a, double b)
{
    return (a != 0.0 && b != 0.0) ? 2.0 / (1.0 / a + 1.0 / b) : 1e-08;
}

// ---- level 0 from the raster ------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_l0_count(const uint8_t *__restrict__ fixed, int rows,
                                                    int cols, int *__restrict__ cnt)
{
    const size_t n = static_cast<size_t>(rows) * cols;
    for (size_t i = blockIdx.x * static_cast<size_t>(kBlock) + threadIdx.x; i < n;
         i += static_cast<size_t>(gridDim.x) * kBlock) {
        int c = 1;
        if (!fixed[i]) {
            const int r = static_cast<int>(i / cols), cc = static_cast<int>(i % cols);
            for (int dr = -1; dr <= 1; ++dr)
                for (int dc = -1; dc <= 1; ++dc) {
                    if (!dr && !dc) continue;
                    const int rr = r + dr, c2 = cc + dc;
                    if (rr < 0 || rr >= rows || c2 < 0 || c2 >= cols) continue;
                    if (!fixed[static_cast<size_t>(rr) * cols + c2]) ++c;
                }
        }
        cnt[i] = c;
    }
}

__global__ __launch_bounds__(kBlock) void k_l0_fill(const double *__restrict__ cond,
                                                   const uint8_t *__restrict__ fixed, int rows,
                                                   int cols, const int *__restrict__ rowptr,
                                                   int *__restrict__ col, double *__restrict__ val)
{
    const size_t n = static_cast<size_t>(rows) * cols;
    for (size_t i = blockIdx.x * static_cast<size_t>(kBlock) + threadIdx.x; i < n;
         i += static_cast<size_t>(gridDim.x) * kBlock) {
        int p = rowptr[i];
        if (fixed[i]) {
            col[p] = static_cast<int>(i);
            val[p] = 1.0;
            continue;
        }
        const int r = static_cast<int>(i / cols), cc = static_cast<int>(i % cols);
        const double ci = cond[i];
        double diag = 0.0;
        const int pd = p++;                       // diagonal first
        for (int dr = -1; dr <= 1; ++dr)
            for (int dc = -1; dc <= 1; ++dc) {
                if (!dr && !dc) continue;
                const int rr = r + dr, c2 = cc + dc;
                if (rr < 0 || rr >= rows || c2 < 0 || c2 >= cols) continue;
                const size_t j = static_cast<size_t>(rr) * cols + c2;
                double w = pair_weight(ci, cond[j]);
                if (dr && dc) w = w / 1.41421353816986083984375;
                diag += w;
                if (!fixed[j]) {
                    col[p] = static_cast<int>(j);
                    val[p] = -w;
                    ++p;
                }
            }
        col[pd] = static_cast<int>(i);
        val[pd] = diag;
    }
}

// ---- per-level helpers --------------------------------------------------------
// dinv = omega-compensated inverse diagonal; isolated rows are solved exactly
__global__ __launch_bounds__(kBlock) void k_dinv(const int *__restrict__ rowptr,
                                                const int *__restrict__ col,
                                                const double *__restrict__ val, int n,
                                                double *__restrict__ dinv)
{
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
        double d = 1.0;
        for (int p = rowptr[i]; p < rowptr[i + 1]; ++p)
            if (col[p] == i) d = val[p];
        const bool isolated = rowptr[i + 1] - rowptr[i] <= 1;
        dinv[i] = (isolated ? 1.0 / kOmega : 1.0) / d;
    }
}

__device__ __forceinline__ uint32_t edge_hash(uint32_t a, uint32_t b)
{
    const uint32_t lo = a < b ? a : b, hi = a < b ? b : a;
    uint32_t h = lo * 0x9E3779B1u ^ (hi + 0x7F4A7C15u) * 0x85EBCA6Bu;
    h ^= h >> 15; h *= 0xC2B2AE35u; h ^= h >> 13;
    return h;
}

__device__ __forceinline__ unsigned long long edge_priority(double w, uint32_t i, uint32_t j)
{   // positive doubles order like their bit patterns; the low 20 mantissa bits
    // are replaced by a symmetric hash so that equal weights tie-break randomly.
    // Measured alternatives (tools/probe_solver_media.py, profiles/r01_notes.md):
    // hash-only among strong couplings and octave buckets + hash coarsen smooth
    // 1e-10..1 gradients (where heaviest-first handshaking only matches chain
    // tops) but are 2-7x slower on the real SSRS rasters, so heaviest-first stays.
    unsigned long long b = static_cast<unsigned long long>(__double_as_longlong(w));
    return (b & ~0xFFFFFull) | (edge_hash(i, j) & 0xFFFFFu);
}

// Strength of connection.  The rasters are two-phase media: "live" cells with
// conductances ~1 and "dead" cells (zero usable updraft) with 1e-8, both phases
// percolating.  A dead cell sees its link to a live neighbour as strong as any of
// its links (all 1e-8), so a criterion relative to the ROW maximum alone lets dead
// and live nodes pair up once the live node has run out of live partners; such
// mixed aggregates cost 2-3x the V-cycle iterations and break the K-cycle
// (tests/dev/amg_experiment6.py: 497 vs 162 V-cycles, K-cycle 500+ vs 30).  The
// symmetric criterion a_ij^2 >= (theta/8)^2 a_ii a_jj never pairs across the phases.
//   mode 0: symmetric;  1: relative to the row maximum (previous behaviour);
//   2: any positive coupling (after symmetric coarsening has stalled)
__device__ __forceinline__ bool strong_link(double w, double wmax, double di_inv, double dj_inv, int mode)
{
    if (mode == 0) return w * w * di_inv * dj_inv >= (kTheta / 8.0) * (kTheta / 8.0);
    if (mode == 1) return w >= kTheta * wmax;
    return true;
}

// every unmatched node proposes to its best unmatched strong neighbour
__global__ __launch_bounds__(kBlock) void k_propose(const int *__restrict__ rowptr,
                                                   const int *__restrict__ col,
                                                   const double *__restrict__ val, int n,
                                                   const int *__restrict__ match,
                                                   int *__restrict__ prop,
                                                   const double *__restrict__ dinv, int mode)
{
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
        int best = -1;
        if (match[i] < 0) {
            double wmax = 0.0;
            for (int p = rowptr[i]; p < rowptr[i + 1]; ++p)
                if (col[p] != i && -val[p] > wmax) wmax = -val[p];
            unsigned long long bp = 0;
            for (int p = rowptr[i]; p < rowptr[i + 1]; ++p) {
                const int j = col[p];
                const double w = -val[p];
                if (j == i || !(w > 0.0) || match[j] >= 0 || !strong_link(w, wmax, dinv[i], dinv[j], mode)) continue;
                const unsigned long long pr = edge_priority(w, i, j);
                if (pr > bp) { bp = pr; best = j; }
            }
        }
        prop[i] = best;
    }
}

__global__ __launch_bounds__(kBlock) void k_accept(int n, const int *__restrict__ prop,
                                                  int *__restrict__ match)
{
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
        const int j = prop[i];
        if (j >= 0 && prop[j] == i) match[i] = j;
    }
}

// Leftover nodes join the aggregate of their best MATCHED neighbour, any number per
// aggregate: a pocket of dead cells that has collapsed to one node hangs on nodes of
// the live cluster by couplings that are weak seen from the live side, so it never
// finds a mutual partner; with a bounded number of joins per pair such leaves pile up
// around hubs and coarsening stalls (1000 x 1200 raster: stuck at 9936 nodes).
// Symmetric-strong neighbours win over merely row-strong ones.  Each node decides for
// itself, so the result does not depend on execution order.
__global__ __launch_bounds__(kBlock) void k_join(const int *__restrict__ rowptr,
                                                const int *__restrict__ col,
                                                const double *__restrict__ val, int n,
                                                const int *__restrict__ match,
                                                const double *__restrict__ dinv,
                                                int *__restrict__ joined_to, int *__restrict__ flag)
{
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
        const bool isolated = rowptr[i + 1] - rowptr[i] <= 1;
        int join = -1, f = 0;
        if (!isolated) {
            if (match[i] >= 0) {
                f = i < match[i] ? 1 : 0;                 // pair leader = smaller index
            } else {
                double wmax = 0.0;
                for (int p = rowptr[i]; p < rowptr[i + 1]; ++p)
                    if (col[p] != i && -val[p] > wmax) wmax = -val[p];
                unsigned long long bp = 0;
                int best = -1;
                for (int p = rowptr[i]; p < rowptr[i + 1]; ++p) {
                    const int j = col[p];
                    const double w = -val[p];
                    if (j == i || !(w > 0.0) || match[j] < 0 || w < kTheta * wmax) continue;
                    unsigned long long pr = edge_priority(w, i, j) >> 1;
                    if (strong_link(w, wmax, dinv[i], dinv[j], 0)) pr |= 1ull << 63;
                    if (pr > bp) { bp = pr; best = j; }
                }
                if (best >= 0) join = best < match[best] ? best : match[best];
                else f = 1;                               // stays a singleton aggregate
            }
        }
        joined_to[i] = join;
        flag[i] = f;
    }
}

// Level 0 (round 4): the aggregates of the raster are its 2 x 2 blocks, each split into the connected components of its
// symmetrically strong links (4 straight + 2 diagonal links inside a block: a union over four nodes in registers).  No
// matching rounds, no joins, nothing that depends on an order; an aggregate never straddles the live / dead phases.
// Against pairwise matching + joins (tests/dev/attic/ua_block_experiment.py, PCG to 1e-15): C1 115 iterations instead of
// 127, the 10 m window 122 instead of 147, the 50 m domain 83 instead of 91, with a smaller level 1.  Plain 2 x 2 blocks
// (round 1) fail because they glue the phases together; 2 x 3 and 3 x 3 split blocks do not converge at 10 m.
// Output in the form k_assign_agg takes: match = -1 everywhere, joined_to = the leader (smallest cell of the component)
// of every non-leader, flag = 1 for leaders.
__global__ __launch_bounds__(kBlock) void k_block_agg(const int *__restrict__ rowptr, const int *__restrict__ col,
                                                     const double *__restrict__ val, const double *__restrict__ dinv,
                                                     int rows, int cols, int *__restrict__ match, int *__restrict__ joined_to,
                                                     int *__restrict__ flag)
{
    const int bcols = (cols + 1) / 2, brows = (rows + 1) / 2;
    const long long nb = static_cast<long long>(brows) * bcols;
    for (long long b = blockIdx.x * static_cast<long long>(kBlock) + threadIdx.x; b < nb;
         b += static_cast<long long>(gridDim.x) * kBlock) {
        const int br = static_cast<int>(b / bcols), bc = static_cast<int>(b - static_cast<long long>(br) * bcols);
        int cell[4];
        bool live[4];                             // exists and is not an isolated (Dirichlet) row
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int r = 2 * br + (k >> 1), c = 2 * bc + (k & 1);
            const bool ok = r < rows && c < cols;
            cell[k] = ok ? r * cols + c : -1;
            live[k] = ok && rowptr[cell[k] + 1] - rowptr[cell[k]] > 1;
        }
        int lab[4] = {0, 1, 2, 3};
        bool strong[4][4] = {};
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            if (!live[a]) continue;
            for (int p = rowptr[cell[a]]; p < rowptr[cell[a] + 1]; ++p) {
                const int j = col[p];
                const double w = -val[p];
#pragma unroll
                for (int c2 = a + 1; c2 < 4; ++c2)
                    if (live[c2] && j == cell[c2] && w > 0.0 && strong_link(w, 0.0, dinv[cell[a]], dinv[j], 0))
                        strong[a][c2] = true;
            }
        }
#pragma unroll
        for (int pass = 0; pass < 3; ++pass)
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int c2 = a + 1; c2 < 4; ++c2)
                    if (strong[a][c2]) {
                        const int m = lab[a] < lab[c2] ? lab[a] : lab[c2];
                        lab[a] = m;
                        lab[c2] = m;
                    }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (cell[k] < 0) continue;
            match[cell[k]] = -1;
            const bool leader = live[k] && lab[k] == k;
            joined_to[cell[k]] = (live[k] && !leader) ? cell[lab[k]] : -1;
            flag[cell[k]] = leader ? 1 : 0;
        }
    }
}

__global__ __launch_bounds__(kBlock) void k_assign_agg(const int *__restrict__ rowptr, int n,
                                                      const int *__restrict__ match,
                                                      const int *__restrict__ joined_to,
                                                      const int *__restrict__ cid,   // exclusive scan of flag
                                                      int *__restrict__ agg,
                                                      unsigned long long *__restrict__ keys)
{
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
        const bool isolated = rowptr[i + 1] - rowptr[i] <= 1;
        int a = -1;
        if (!isolated) {
            int leader = i;
            if (match[i] >= 0) leader = i < match[i] ? i : match[i];
            else if (joined_to[i] >= 0) leader = joined_to[i];
            a = cid[leader];
        }
        agg[i] = a;
        // (aggregate, member) keys; sorted, they are the member lists of the restriction
        keys[i] = a >= 0 ? (static_cast<unsigned long long>(a) << 32) | static_cast<uint32_t>(i) : ~0ull;
    }
}

// member lists from the sorted keys: memidx[p] = fine node, memptr[a] = first p of aggregate a
__global__ __launch_bounds__(kBlock) void k_member_lists(const unsigned long long *__restrict__ keys, int n,
                                                        int nc, int *__restrict__ memptr,
                                                        int *__restrict__ memidx)
{
    for (int p = blockIdx.x * kBlock + threadIdx.x; p < n; p += gridDim.x * kBlock) {
        const unsigned long long k = keys[p];
        memidx[p] = static_cast<int>(k & 0xFFFFFFFFull);
        const bool parked = k == ~0ull;
        const int a = parked ? nc : static_cast<int>(k >> 32);
        const bool first = p == 0 || (keys[p - 1] >> 32) != (k >> 32);
        if (first) memptr[a] = p;                         // the parked bucket marks the end
        if (p == n - 1 && !parked) memptr[nc] = n;
    }
}

// Galerkin keys: one (I, J) key per fine entry whose ends both have aggregates
__global__ __launch_bounds__(kBlock) void k_galerkin_keys(const int *__restrict__ rowptr,
                                                         const int *__restrict__ col,
                                                         const double *__restrict__ val, int n,
                                                         const int *__restrict__ agg,
                                                         unsigned long long *__restrict__ keys,
                                                         double *__restrict__ vals)
{
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
        const int I = agg[i];
        for (int p = rowptr[i]; p < rowptr[i + 1]; ++p) {
            const int J = agg[col[p]];
            // entries to/from nodes without aggregate are parked behind all real keys
            const bool ok = I >= 0 && J >= 0;
            keys[p] = ok ? (static_cast<unsigned long long>(I) << 32) | static_cast<uint32_t>(J)
                         : ~0ull;
            vals[p] = ok ? val[p] : 0.0;
        }
    }
}

__global__ __launch_bounds__(kBlock) void k_unpack_coarse(const unsigned long long *__restrict__ keys,
                                                         int nnz, int nc, int *__restrict__ rowptr,
                                                         int *__restrict__ col)
{
    for (int e = blockIdx.x * kBlock + threadIdx.x; e < nnz; e += gridDim.x * kBlock) {
        const int I = static_cast<int>(keys[e] >> 32);
        col[e] = static_cast<int>(keys[e] & 0xFFFFFFFFull);
        if (e == 0 || static_cast<int>(keys[e - 1] >> 32) != I) rowptr[I] = e;
        if (e == nnz - 1) rowptr[nc] = nnz;
    }
}

// ---- cycle kernels --------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_jacobi_first(const cv_t *__restrict__ dinv,
                                                        const cv_t *__restrict__ b, int n,
                                                        cv_t *__restrict__ x, double w)
{
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock)
        x[i] = static_cast<cv_t>(w * dinv[i] * b[i]);
}

__global__ __launch_bounds__(kBlock) void k_to_cv(const double *__restrict__ a, int n, cv_t *__restrict__ out)
{
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) out[i] = static_cast<cv_t>(a[i]);
}

// Four lanes per row for the large CSR levels: a thread-per-row walk reads val/col
// with a stride of one row (~12 entries) between lanes; four lanes read four
// consecutive entries, and the partial sums meet in a fixed shuffle tree
// (reproducible).  Level 1 at 5000 x 6000: 1.15 ms -> see profiles/r01_notes.md.
constexpr int kRowLanes = 4;
template <class V>
__device__ __forceinline__ double row_dot4(const int *__restrict__ rowptr, const int *__restrict__ col,
                                           const V *__restrict__ val, const cv_t *__restrict__ x,
                                           int row, int sub)
{
    double ax = 0.0;
    const int end = rowptr[row + 1];
    for (int p = rowptr[row] + sub; p < end; p += kRowLanes) ax += static_cast<double>(val[p]) * x[col[p]];
    ax += __shfl_xor(ax, 1);
    ax += __shfl_xor(ax, 2);
    return ax;
}

// kRowsPerGroup rows per 4-lane group, their loads issued together: with one row per group
// a sweep is a chain of three dependent memory latencies (rowptr -> col / val -> x[col]) and
// the occupancy limit (8192 waves) makes level 1 at 5000 x 6000 latency-bound (0.51 ms for
// 1.4 GB); four independent chains per lane cover the latency.
constexpr int kRowsPerGroup = 4;
template <class V>
__device__ __forceinline__ void rows_dot4(const int *__restrict__ rowptr, const int *__restrict__ col,
                                          const V *__restrict__ val, const cv_t *__restrict__ x,
                                          const long long (&row)[kRowsPerGroup], int n, int sub,
                                          double (&ax)[kRowsPerGroup])
{
    int p[kRowsPerGroup], end[kRowsPerGroup];
#pragma unroll
    for (int u = 0; u < kRowsPerGroup; ++u) {
        const bool ok = row[u] < n;
        p[u] = ok ? rowptr[row[u]] + sub : 0;
        end[u] = ok ? rowptr[row[u] + 1] : 0;
        ax[u] = 0.0;
    }
    // two entries per lane and row in flight (rows of up to 8 non-zeros: the common case)
    int c0[kRowsPerGroup], c1[kRowsPerGroup];
    double v0[kRowsPerGroup], v1[kRowsPerGroup];
#pragma unroll
    for (int u = 0; u < kRowsPerGroup; ++u) {
        const bool a = p[u] < end[u], b = p[u] + kRowLanes < end[u];
        c0[u] = a ? col[p[u]] : -1;
        v0[u] = a ? static_cast<double>(val[p[u]]) : 0.0;
        c1[u] = b ? col[p[u] + kRowLanes] : -1;
        v1[u] = b ? static_cast<double>(val[p[u] + kRowLanes]) : 0.0;
    }
#pragma unroll
    for (int u = 0; u < kRowsPerGroup; ++u) {
        const double x0 = c0[u] >= 0 ? x[c0[u]] : 0.0;
        const double x1 = c1[u] >= 0 ? x[c1[u]] : 0.0;
        ax[u] += v0[u] * x0;
        ax[u] += v1[u] * x1;
    }
#pragma unroll
    for (int u = 0; u < kRowsPerGroup; ++u)                    // longer rows
        for (int q = p[u] + 2 * kRowLanes; q < end[u]; q += kRowLanes) ax[u] += static_cast<double>(val[q]) * x[col[q]];
#pragma unroll
    for (int u = 0; u < kRowsPerGroup; ++u) {
        ax[u] += __shfl_xor(ax[u], 1);
        ax[u] += __shfl_xor(ax[u], 2);
    }
}

template <class V>
__global__ __launch_bounds__(kBlock) void k_jacobi4(const int *__restrict__ rowptr,
                                                   const int *__restrict__ col,
                                                   const V *__restrict__ val,
                                                   const cv_t *__restrict__ dinv,
                                                   const cv_t *__restrict__ b,
                                                   const cv_t *__restrict__ x, int n,
                                                   cv_t *__restrict__ xn, double w)
{
    const int sub = threadIdx.x % kRowLanes;
    const long long groups = static_cast<long long>(gridDim.x) * (kBlock / kRowLanes);
    // whole groups stay in the loop together (the bound is per group, not per lane)
    for (long long i = blockIdx.x * static_cast<long long>(kBlock / kRowLanes) + threadIdx.x / kRowLanes; i < n;
         i += groups * kRowsPerGroup) {
        long long row[kRowsPerGroup];
        double ax[kRowsPerGroup];
#pragma unroll
        for (int u = 0; u < kRowsPerGroup; ++u) row[u] = i + u * groups;
        rows_dot4(rowptr, col, val, x, row, n, sub, ax);
#pragma unroll
        for (int u = 0; u < kRowsPerGroup; ++u)
            if (sub == 0 && row[u] < n) xn[row[u]] = static_cast<cv_t>(x[row[u]] + w * dinv[row[u]] * (b[row[u]] - ax[u]));
    }
}

template <class V>
__global__ __launch_bounds__(kBlock) void k_residual4(const int *__restrict__ rowptr,
                                                     const int *__restrict__ col,
                                                     const V *__restrict__ val,
                                                     const cv_t *__restrict__ b,
                                                     const cv_t *__restrict__ x, int n,
                                                     cv_t *__restrict__ r)
{
    const int sub = threadIdx.x % kRowLanes;
    const long long groups = static_cast<long long>(gridDim.x) * (kBlock / kRowLanes);
    for (long long i = blockIdx.x * static_cast<long long>(kBlock / kRowLanes) + threadIdx.x / kRowLanes; i < n;
         i += groups * kRowsPerGroup) {
        long long row[kRowsPerGroup];
        double ax[kRowsPerGroup];
#pragma unroll
        for (int u = 0; u < kRowsPerGroup; ++u) row[u] = i + u * groups;
        rows_dot4(rowptr, col, val, x, row, n, sub, ax);
#pragma unroll
        for (int u = 0; u < kRowsPerGroup; ++u)
            if (sub == 0 && row[u] < n) r[row[u]] = static_cast<cv_t>(b[row[u]] - ax[u]);
    }
}

// ---- sliced ELL form of a level's f32 entries (AmgLevel::sell_*) and its sweeps
__global__ __launch_bounds__(kBlock) void k_sell_widths(const int *__restrict__ rowptr, int n, int nsl, int *__restrict__ cnt)
{
    const int lane = threadIdx.x & 63;
    for (int s = blockIdx.x * (kBlock / 64) + static_cast<int>(threadIdx.x >> 6); s < nsl; s += gridDim.x * (kBlock / 64)) {
        const int row = s * 64 + lane;
        int len = row < n ? rowptr[row + 1] - rowptr[row] : 0;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) { const int o = __shfl_xor(len, off); len = o > len ? o : len; }
        if (lane == 0) cnt[s] = len * 64;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) cnt[nsl] = 0;
}

__global__ __launch_bounds__(kBlock) void k_sell_fill(const int *__restrict__ rowptr, const int *__restrict__ col,
                                                     const float *__restrict__ val, int n, int nsl,
                                                     const int *__restrict__ sell_ptr, int *__restrict__ sell_col,
                                                     float *__restrict__ sell_val)
{
    const int lane = threadIdx.x & 63;
    for (int s = blockIdx.x * (kBlock / 64) + static_cast<int>(threadIdx.x >> 6); s < nsl; s += gridDim.x * (kBlock / 64)) {
        const int row = s * 64 + lane;
        const int p0 = sell_ptr[s], width = (sell_ptr[s + 1] - p0) / 64;
        const int r0 = row < n ? rowptr[row] : 0, len = row < n ? rowptr[row + 1] - r0 : 0;
        for (int k = 0; k < width; ++k) {
            const bool has = k < len;
            sell_col[p0 + 64 * k + lane] = has ? col[r0 + k] : (row < n ? row : 0);
            sell_val[p0 + 64 * k + lane] = has ? val[r0 + k] : 0.0f;
        }
    }
}

// JAC: xn = x + w D^-1 (b - A x); else r = b - A x.  One wave = one slice; four entries per lane in flight
template <bool JAC>
__global__ __launch_bounds__(kBlock) void k_sweep_sell(const int *__restrict__ sell_ptr, const int *__restrict__ sell_col,
                                                      const float *__restrict__ sell_val, const cv_t *__restrict__ dinv,
                                                      const cv_t *__restrict__ b, const cv_t *__restrict__ x, int n, int nsl,
                                                      cv_t *__restrict__ out, double w)
{
    const int lane = threadIdx.x & 63;
    for (int s = blockIdx.x * (kBlock / 64) + static_cast<int>(threadIdx.x >> 6); s < nsl; s += gridDim.x * (kBlock / 64)) {
        const int p1 = sell_ptr[s + 1];
        int p = sell_ptr[s] + lane;
        double ax = 0.0;
        for (; p + 3 * 64 < p1; p += 4 * 64) {
            const int c0 = sell_col[p], c1 = sell_col[p + 64], c2 = sell_col[p + 128], c3 = sell_col[p + 192];
            const float v0 = sell_val[p], v1 = sell_val[p + 64], v2 = sell_val[p + 128], v3 = sell_val[p + 192];
            const double x0 = x[c0], x1 = x[c1], x2 = x[c2], x3 = x[c3];
            ax += static_cast<double>(v0) * x0;
            ax += static_cast<double>(v1) * x1;
            ax += static_cast<double>(v2) * x2;
            ax += static_cast<double>(v3) * x3;
        }
        for (; p < p1; p += 64) ax += static_cast<double>(sell_val[p]) * x[sell_col[p]];
        const int row = s * 64 + lane;
        if (row < n) out[row] = static_cast<cv_t>(JAC ? x[row] + w * dinv[row] * (b[row] - ax) : b[row] - ax);
    }
}

template <class V>
__global__ __launch_bounds__(kBlock) void k_jacobi(const int *__restrict__ rowptr,
                                                  const int *__restrict__ col,
                                                  const V *__restrict__ val,
                                                  const cv_t *__restrict__ dinv,
                                                  const cv_t *__restrict__ b,
                                                  const cv_t *__restrict__ x, int n,
                                                  cv_t *__restrict__ xn, double w)
{
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
        double ax = 0.0;
        for (int p = rowptr[i]; p < rowptr[i + 1]; ++p) ax += static_cast<double>(val[p]) * x[col[p]];
        xn[i] = static_cast<cv_t>(x[i] + w * dinv[i] * (b[i] - ax));
    }
}

template <class V>
__global__ __launch_bounds__(kBlock) void k_residual(const int *__restrict__ rowptr,
                                                    const int *__restrict__ col,
                                                    const V *__restrict__ val,
                                                    const cv_t *__restrict__ b,
                                                    const cv_t *__restrict__ x, int n,
                                                    cv_t *__restrict__ r)
{
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
        double ax = 0.0;
        for (int p = rowptr[i]; p < rowptr[i + 1]; ++p) ax += static_cast<double>(val[p]) * x[col[p]];
        r[i] = static_cast<cv_t>(b[i] - ax);
    }
}

// The cycle's sweeps on the CSR levels read f32 copies of the entries (8 instead of 12 B
// per non-zero; these kernels move bytes).  Rounded so that the copy stays a symmetric
// M-matrix: off-diagonals (<= 0) toward zero, the diagonal upward, hence still diagonally
// dominant, and a_ij = a_ji round alike.  Set-up (strength, Galerkin, dense inverse) keeps f64.
__global__ __launch_bounds__(kBlock) void k_val32(const int *__restrict__ rowptr, const int *__restrict__ col,
                                                 const double *__restrict__ val, int n, float *__restrict__ out)
{
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock)
        for (int p = rowptr[i]; p < rowptr[i + 1]; ++p)
            out[p] = col[p] == i ? __double2float_ru(val[p]) : __double2float_rz(val[p]);
}

// ---- level 0 matrix-free ----------------------------------------------------
// The finest level is a 9-point stencil on the raster: applying it from the
// reciprocal conductances (8 B per cell, neighbours through L2) moves ~40 B per cell
// and sweep, the CSR form 12 B per non-zero = 108 B per cell at one third of the
// bandwidth (thread-per-row gathers): 3.3 ms vs 0.5 ms per sweep at 5000 x 6000, and
// the five level-0 sweeps were two thirds of a V-cycle.  The CSR copy of level 0 is
// still built: the aggregation and the Galerkin product read it.
struct L0Stencil {
    const cv_t *rinv;         // 1 / cond, 0 where cond == 0 (then every link is 1e-8); the sign
                              // bit marks Dirichlet cells (saves nine byte loads per cell)
    const uint8_t *fixed;
    int rows, cols;
};

__global__ __launch_bounds__(kBlock) void k_l0_rinv(const double *__restrict__ cond,
                                                   const uint8_t *__restrict__ fixed, size_t n,
                                                   double *__restrict__ rinv)
{
    for (size_t i = blockIdx.x * static_cast<size_t>(kBlock) + threadIdx.x; i < n;
         i += static_cast<size_t>(gridDim.x) * kBlock) {
        const double c = cond[i];
        const double v = c != 0.0 ? fabs(1.0 / c) : 0.0;      // conductances are >= 0
        rinv[i] = fixed[i] ? -v : v;
    }
}

// (A x)_i of the level-0 operator.  The weights are those of k_l0_fill up to rounding
// (diagonal links are multiplied by 1/sqrt(2) instead of divided by sqrt(2), and the whole row is
// evaluated in the cycle's precision): this operator only preconditions, what it must be is
// symmetric (w_ij is a function of ri + rj) and diagonally dominant (diag = the sum of the same
// w_ij), which it is by construction.
constexpr double kInvFacDiag = 1.0 / 1.41421353816986083984375;

// One wave = one row segment of 62 cells plus a halo lane on either side: every lane loads
// its own column of the three rows (six loads for x and rinv), the east / west neighbours
// arrive by lane shuffles.  The kernels were bound by the number of load instructions (21
// per cell with a thread-per-cell stencil: 0.66 ms per sweep at 5000 x 6000), not by bytes.
constexpr int kL0Cols = 62;                                   // cells per wave
__device__ __forceinline__ cv_t l0_apply_wave(const L0Stencil &a, const cv_t *__restrict__ x,
                                              int r, int c, bool &centre, size_t &i, cv_t &xi)
{
    const int lane = threadIdx.x & 63;
    const bool col_ok = c >= 0 && c < a.cols;
    const cv_t inf = static_cast<cv_t>(__builtin_inf());
    cv_t xv[3], sv[3];                                        // rows r-1, r, r+1 of this lane's column
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        const int rr = r + d - 1;
        const bool ok = col_ok && rr >= 0 && rr < a.rows;
        const size_t j = static_cast<size_t>(ok ? rr : r) * a.cols + (col_ok ? c : 0);
        xv[d] = ok ? x[j] : static_cast<cv_t>(0);
        sv[d] = ok ? a.rinv[j] : inf;                         // +inf: outside the raster, no link
    }
    i = static_cast<size_t>(r) * a.cols + (col_ok ? c : 0);
    centre = col_ok && lane >= 1 && lane <= kL0Cols;
    xi = xv[1];
    const cv_t si = sv[1], ri = fabs(si);
    cv_t diag = 0, off = 0;
#pragma unroll
    for (int k = 0; k < 9; ++k) {                             // same order as before: rows south to north
        if (k == 4) continue;
        const int d = k / 3, dc = k % 3 - 1;
        cv_t sj = sv[d], xj = xv[d];
        if (dc < 0) { sj = __shfl_up(sj, 1); xj = __shfl_up(xj, 1); }
        if (dc > 0) { sj = __shfl_down(sj, 1); xj = __shfl_down(xj, 1); }
        const cv_t rj = fabs(sj);
        cv_t w = (ri != 0 && rj != 0) ? static_cast<cv_t>(2) / (ri + rj) : static_cast<cv_t>(1e-08);
        if (rj == inf) w = 0;
        if (d != 1 && dc != 0) w = w * static_cast<cv_t>(kInvFacDiag);
        diag += w;
        if (!signbit(sj)) off += w * xj;
    }
    if (signbit(si)) return xv[1];                            // Dirichlet cell: identity row
    return diag * xv[1] - off;
}

// ---- level 0 of the V(1,1) cycle in two passes (f64 cycle only).  Unfused it is: copy rhs -> b (2 array passes), x = w D^-1 b
// (3), r = b - A x (4), [restrict], x += P x_c (2.5), x = x + w D^-1 (b - A x) (5), copy x -> out (2) = 18.5 passes of 240 MB
// at 5000 x 6000; fused: pre (b, dinv, rinv -> x, r: 5) and post (x, agg, x_c, rinv, b, dinv -> out: 6.5) = 11.5.
struct L0Slots { const double *rhs; double *out; };
__global__ void k_set_slots(L0Slots *s, const double *rhs, double *out) { s->rhs = rhs; s->out = out; }

// the stencil of l0_apply_wave on values the caller has in registers: xv / sv = this lane's column of rows r-1, r, r+1
__device__ __forceinline__ double l0_core(const double (&xv)[3], const double (&sv)[3])
{
    const double inf = __builtin_inf();
    const double si = sv[1], ri = fabs(si);
    double diag = 0.0, off = 0.0;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        if (k == 4) continue;
        const int d = k / 3, dc = k % 3 - 1;
        double sj = sv[d], xj = xv[d];
        if (dc < 0) { sj = __shfl_up(sj, 1); xj = __shfl_up(xj, 1); }
        if (dc > 0) { sj = __shfl_down(sj, 1); xj = __shfl_down(xj, 1); }
        const double rj = fabs(sj);
        double w = (ri != 0.0 && rj != 0.0) ? 2.0 / (ri + rj) : 1e-08;
        if (rj == inf) w = 0.0;
        if (d != 1 && dc != 0) w = w * kInvFacDiag;
        diag += w;
        if (!signbit(sj)) off += w * xj;
    }
    if (signbit(si)) return xv[1];                            // Dirichlet cell: identity row
    return diag * xv[1] - off;
}

// pre: x = w D^-1 b and r = b - A x in one pass (the neighbours' x is recomputed from their b and dinv)
__global__ __launch_bounds__(kBlock) void k_l0_pre_fused(const double *__restrict__ rinv, int rows, int cols,
                                                        const double *__restrict__ dinv, const L0Slots *__restrict__ slots,
                                                        double w, double *__restrict__ x, double *__restrict__ r)
{
    const double *__restrict__ b = slots->rhs;
    const int c = (static_cast<int>(blockIdx.x) * (kBlock / 64) + static_cast<int>(threadIdx.x >> 6)) * kL0Cols +
                  static_cast<int>(threadIdx.x & 63) - 1;
    const int row = static_cast<int>(blockIdx.y), lane = threadIdx.x & 63;
    const bool col_ok = c >= 0 && c < cols;
    double xv[3], sv[3], bc = 0.0;
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        const int rr = row + d - 1;
        const bool ok = col_ok && rr >= 0 && rr < rows;
        const size_t j = static_cast<size_t>(ok ? rr : row) * cols + (col_ok ? c : 0);
        const double bj = ok ? b[j] : 0.0;
        xv[d] = ok ? w * dinv[j] * bj : 0.0;
        sv[d] = ok ? rinv[j] : __builtin_inf();
        if (d == 1) bc = bj;
    }
    const double ax = l0_core(xv, sv);
    if (col_ok && lane >= 1 && lane <= kL0Cols) {
        const size_t i = static_cast<size_t>(row) * cols + c;
        x[i] = xv[1];
        r[i] = bc - ax;
    }
}

// post: x' = x + P x_c (recomputed for the neighbours) and out = x' + w D^-1 (b - A x') in one pass
__global__ __launch_bounds__(kBlock) void k_l0_post_fused(const double *__restrict__ rinv, int rows, int cols,
                                                         const double *__restrict__ dinv, const L0Slots *__restrict__ slots,
                                                         const int *__restrict__ agg, const double *__restrict__ xc,
                                                         const double *__restrict__ x, double w)
{
    const double *__restrict__ b = slots->rhs;
    double *__restrict__ out = slots->out;
    const int c = (static_cast<int>(blockIdx.x) * (kBlock / 64) + static_cast<int>(threadIdx.x >> 6)) * kL0Cols +
                  static_cast<int>(threadIdx.x & 63) - 1;
    const int row = static_cast<int>(blockIdx.y), lane = threadIdx.x & 63;
    const bool col_ok = c >= 0 && c < cols;
    double xv[3], sv[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        const int rr = row + d - 1;
        const bool ok = col_ok && rr >= 0 && rr < rows;
        const size_t j = static_cast<size_t>(ok ? rr : row) * cols + (col_ok ? c : 0);
        const int a = ok ? agg[j] : -1;
        xv[d] = ok ? x[j] + (a >= 0 ? xc[a] : 0.0) : 0.0;
        sv[d] = ok ? rinv[j] : __builtin_inf();
    }
    const double ax = l0_core(xv, sv);
    if (col_ok && lane >= 1 && lane <= kL0Cols) {
        const size_t i = static_cast<size_t>(row) * cols + c;
        out[i] = xv[1] + w * dinv[i] * (b[i] - ax);
    }
}

// ---- the same two passes for TWO raster rows per wave (round 4, late): a wave that computes rows 2R and 2R + 1 loads rows
// 2R - 1 .. 2R + 2 once -- 12 loads for two rows of the pre pass instead of 18, 20 instead of 28 in the post pass (the
// kernels are bound by their load instructions) -- and, since level 1's aggregates are the parts of the aligned 2 x 2 blocks
// (k_block_agg), the pre pass has every member of an aggregate in two neighbouring lanes: it writes the restricted residual
// itself (members added in index order, as k_restrict does) instead of writing r for k_restrict to gather.  Same arithmetic
// per cell in the same order as l0_core: bit-identical to the one-row kernels (SSRS_AMG_L0_ONE_ROW) and to the unfused path.
__device__ __forceinline__ void l0_shift4(const double (&v)[4], double (&l)[4], double (&r)[4])
{
#pragma unroll
    for (int k = 0; k < 4; ++k) { l[k] = __shfl_up(v[k], 1); r[k] = __shfl_down(v[k], 1); }
}

// row d0 + 1 of the four loaded rows (d0 = 0: row 2R, d0 = 1: row 2R + 1); l / c / r = west / own / east columns
__device__ __forceinline__ double l0_core4(const double (&xl)[4], const double (&xc)[4], const double (&xr)[4],
                                           const double (&sl)[4], const double (&sc)[4], const double (&sr)[4], int d0)
{
    const double inf = __builtin_inf();
    const double si = sc[d0 + 1], ri = fabs(si);
    double diag = 0.0, off = 0.0;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        if (k == 4) continue;
        const int d = k / 3, dc = k % 3 - 1;
        const double sj = dc < 0 ? sl[d0 + d] : (dc > 0 ? sr[d0 + d] : sc[d0 + d]);
        const double xj = dc < 0 ? xl[d0 + d] : (dc > 0 ? xr[d0 + d] : xc[d0 + d]);
        const double rj = fabs(sj);
        double w = (ri != 0.0 && rj != 0.0) ? 2.0 / (ri + rj) : 1e-08;
        if (rj == inf) w = 0.0;
        if (d != 1 && dc != 0) w = w * kInvFacDiag;
        diag += w;
        if (!signbit(sj)) off += w * xj;
    }
    if (signbit(si)) return xc[d0 + 1];                       // Dirichlet cell: identity row
    return diag * xc[d0 + 1] - off;
}

// pre: x = w D^-1 b, r = b - A x; bc = R r when `agg` is given (r is then not written at all), else r is written
__global__ __launch_bounds__(kBlock) void k_l0_pre2(const double *__restrict__ rinv, int rows, int cols,
                                                   const double *__restrict__ dinv, const L0Slots *__restrict__ slots,
                                                   double w, double *__restrict__ x, double *__restrict__ r,
                                                   const int *__restrict__ agg, double *__restrict__ bc)
{
    const double *__restrict__ b = slots->rhs;
    const int c = (static_cast<int>(blockIdx.x) * (kBlock / 64) + static_cast<int>(threadIdx.x >> 6)) * kL0Cols +
                  static_cast<int>(threadIdx.x & 63) - 1;
    const int row0 = 2 * static_cast<int>(blockIdx.y), lane = threadIdx.x & 63;
    const bool col_ok = c >= 0 && c < cols;
    double xv[4], sv[4], bv[2] = {0.0, 0.0};
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        const int rr = row0 + d - 1;
        const bool ok = col_ok && rr >= 0 && rr < rows;
        const size_t j = static_cast<size_t>(ok ? rr : row0) * cols + (col_ok ? c : 0);
        const double bj = ok ? b[j] : 0.0;
        xv[d] = ok ? w * dinv[j] * bj : 0.0;
        sv[d] = ok ? rinv[j] : __builtin_inf();
        if (d == 1) bv[0] = bj;
        if (d == 2) bv[1] = bj;
    }
    double xl[4], xr[4], sl[4], sr[4];
    l0_shift4(xv, xl, xr);
    l0_shift4(sv, sl, sr);
    const double ax0 = l0_core4(xl, xv, xr, sl, sv, sr, 0), ax1 = l0_core4(xl, xv, xr, sl, sv, sr, 1);
    const bool centre = col_ok && lane >= 1 && lane <= kL0Cols, row1_ok = row0 + 1 < rows;
    const size_t i0 = static_cast<size_t>(row0) * cols + (col_ok ? c : 0), i1 = i0 + cols;
    const double r0 = bv[0] - ax0, r1 = bv[1] - ax1;
    if (centre) {
        x[i0] = xv[1];
        if (row1_ok) x[i1] = xv[2];
        if (!agg) {
            r[i0] = r0;
            if (row1_ok) r[i1] = r1;
        }
    }
    if (agg) {
        // the 2 x 2 block = lanes (l, l + 1) with l odd (c even; a wave's first column is even) x rows (2R, 2R + 1)
        const int a0 = centre ? agg[i0] : -1, a1 = (centre && row1_ok) ? agg[i1] : -1;
        const double q0 = __shfl_down(r0, 1), q1 = __shfl_down(r1, 1);
        const int e0 = __shfl_down(a0, 1), e1 = __shfl_down(a1, 1);
        if ((lane & 1) && centre) {
            const bool east = lane + 1 <= kL0Cols;                  // (lane 63 is the halo: a wave's 62 columns are 31 whole blocks)
            const int am[4] = {a0, east ? e0 : -1, a1, east ? e1 : -1};
            const double rm[4] = {r0, q0, r1, q1};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (am[i] < 0) continue;
                bool first = true;
#pragma unroll
                for (int j = 0; j < i; ++j) first = first && am[j] != am[i];
                if (!first) continue;
                double s = 0.0;
                s += rm[i];
#pragma unroll
                for (int j = i + 1; j < 4; ++j)
                    if (am[j] == am[i]) s += rm[j];
                bc[am[i]] = s;
            }
        }
    }
}

// post: x' = x + P x_c and out = x' + w D^-1 (b - A x'), two rows per wave
__global__ __launch_bounds__(kBlock) void k_l0_post2(const double *__restrict__ rinv, int rows, int cols,
                                                    const double *__restrict__ dinv, const L0Slots *__restrict__ slots,
                                                    const int *__restrict__ agg, const double *__restrict__ xc,
                                                    const double *__restrict__ x, double w)
{
    const double *__restrict__ b = slots->rhs;
    double *__restrict__ out = slots->out;
    const int c = (static_cast<int>(blockIdx.x) * (kBlock / 64) + static_cast<int>(threadIdx.x >> 6)) * kL0Cols +
                  static_cast<int>(threadIdx.x & 63) - 1;
    const int row0 = 2 * static_cast<int>(blockIdx.y), lane = threadIdx.x & 63;
    const bool col_ok = c >= 0 && c < cols;
    double xv[4], sv[4];
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        const int rr = row0 + d - 1;
        const bool ok = col_ok && rr >= 0 && rr < rows;
        const size_t j = static_cast<size_t>(ok ? rr : row0) * cols + (col_ok ? c : 0);
        const int a = ok ? agg[j] : -1;
        xv[d] = ok ? x[j] + (a >= 0 ? xc[a] : 0.0) : 0.0;
        sv[d] = ok ? rinv[j] : __builtin_inf();
    }
    double xl[4], xr[4], sl[4], sr[4];
    l0_shift4(xv, xl, xr);
    l0_shift4(sv, sl, sr);
    const double ax0 = l0_core4(xl, xv, xr, sl, sv, sr, 0), ax1 = l0_core4(xl, xv, xr, sl, sv, sr, 1);
    if (col_ok && lane >= 1 && lane <= kL0Cols) {
        const size_t i0 = static_cast<size_t>(row0) * cols + c, i1 = i0 + cols;
        out[i0] = xv[1] + w * dinv[i0] * (b[i0] - ax0);
        if (row0 + 1 < rows) out[i1] = xv[2] + w * dinv[i1] * (b[i1] - ax1);
    }
}

__global__ __launch_bounds__(kBlock) void k_l0_jacobi(L0Stencil a, const cv_t *__restrict__ dinv,
                                                     const cv_t *__restrict__ b,
                                                     const cv_t *__restrict__ x,
                                                     cv_t *__restrict__ xn, cv_t w)
{
    const int c = (static_cast<int>(blockIdx.x) * (kBlock / 64) + static_cast<int>(threadIdx.x >> 6)) * kL0Cols +
                  static_cast<int>(threadIdx.x & 63) - 1;
    bool centre;
    size_t i;
    cv_t xi;
    const cv_t ax = l0_apply_wave(a, x, static_cast<int>(blockIdx.y), c, centre, i, xi);
    if (centre) xn[i] = xi + w * dinv[i] * (b[i] - ax);
}

__global__ __launch_bounds__(kBlock) void k_l0_residual(L0Stencil a, const cv_t *__restrict__ b,
                                                       const cv_t *__restrict__ x,
                                                       cv_t *__restrict__ r)
{
    const int c = (static_cast<int>(blockIdx.x) * (kBlock / 64) + static_cast<int>(threadIdx.x >> 6)) * kL0Cols +
                  static_cast<int>(threadIdx.x & 63) - 1;
    bool centre;
    size_t i;
    cv_t xi;
    const cv_t ax = l0_apply_wave(a, x, static_cast<int>(blockIdx.y), c, centre, i, xi);
    if (centre) r[i] = b[i] - ax;
}

__global__ __launch_bounds__(kBlock) void k_restrict(const int *__restrict__ memptr,
                                                    const int *__restrict__ memidx,
                                                    const cv_t *__restrict__ r, int nc,
                                                    cv_t *__restrict__ bc)
{
    for (int I = blockIdx.x * kBlock + threadIdx.x; I < nc; I += gridDim.x * kBlock) {
        double s = 0.0;                                   // members in index order: reproducible
        for (int p = memptr[I]; p < memptr[I + 1]; ++p) s += r[memidx[p]];
        bc[I] = static_cast<cv_t>(s);
    }
}

__global__ __launch_bounds__(kBlock) void k_prolong_add(const int *__restrict__ agg,
                                                       const cv_t *__restrict__ xc, int n,
                                                       cv_t *__restrict__ x)
{
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
        const int a = agg[i];
        if (a >= 0) x[i] += xc[a];
    }
}

// ---- dense coarsest level ---------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_dense_fill(const int *__restrict__ rowptr,
                                                      const int *__restrict__ col,
                                                      const double *__restrict__ val, int n,
                                                      double *__restrict__ aug)   // n x 2n [A | I]
{
    const size_t total = static_cast<size_t>(n) * 2 * n;
    for (size_t e = blockIdx.x * static_cast<size_t>(kBlock) + threadIdx.x; e < total;
         e += static_cast<size_t>(gridDim.x) * kBlock) {
        const int i = static_cast<int>(e / (2 * n)), j = static_cast<int>(e % (2 * n));
        aug[e] = (j >= n && j - n == i) ? 1.0 : 0.0;
    }
}
__global__ __launch_bounds__(kBlock) void k_dense_scatter(const int *__restrict__ rowptr,
                                                         const int *__restrict__ col,
                                                         const double *__restrict__ val, int n,
                                                         double *__restrict__ aug)
{
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock)
        for (int p = rowptr[i]; p < rowptr[i + 1]; ++p)
            aug[static_cast<size_t>(i) * 2 * n + col[p]] = val[p];
}
// Gauss-Jordan step k: rows i != k get row_i -= (a_ik / a_kk) row_k; the pivot
// column is saved first so that the update is race free
__global__ __launch_bounds__(kBlock) void k_gj_save(const double *__restrict__ aug, int n, int k,
                                                   double *__restrict__ colk)
{
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock)
        colk[i] = aug[static_cast<size_t>(i) * 2 * n + k];
}
__global__ __launch_bounds__(kBlock) void k_gj_elim(double *__restrict__ aug, int n, int k,
                                                   const double *__restrict__ colk)
{
    const size_t total = static_cast<size_t>(n) * 2 * n;
    const double piv = colk[k];
    for (size_t e = blockIdx.x * static_cast<size_t>(kBlock) + threadIdx.x; e < total;
         e += static_cast<size_t>(gridDim.x) * kBlock) {
        const int i = static_cast<int>(e / (2 * n)), j = static_cast<int>(e % (2 * n));
        if (i == k) continue;
        aug[e] -= (colk[i] / piv) * aug[static_cast<size_t>(k) * 2 * n + j];
    }
}
__global__ __launch_bounds__(kBlock) void k_gj_finish(const double *__restrict__ aug, int n,
                                                     double *__restrict__ inv)
{
    const size_t total = static_cast<size_t>(n) * n;
    for (size_t e = blockIdx.x * static_cast<size_t>(kBlock) + threadIdx.x; e < total;
         e += static_cast<size_t>(gridDim.x) * kBlock) {
        const int i = static_cast<int>(e / n), j = static_cast<int>(e % n);
        inv[e] = aug[static_cast<size_t>(i) * 2 * n + n + j] / aug[static_cast<size_t>(i) * 2 * n + i];
    }
}
// x = inv * b on the last level: one wave per row, lanes stride the row
// (coalesced), shuffle reduction.  (One thread per row walked 800 strided
// loads serially and was ~40 % of a whole V-cycle at 500 x 600.)
__global__ __launch_bounds__(kBlock) void k_dense_apply(const double *__restrict__ inv,
                                                       const cv_t *__restrict__ b, int n,
                                                       cv_t *__restrict__ x)
{
    const int lane = threadIdx.x & 63;
    const int row = (blockIdx.x * kBlock + threadIdx.x) >> 6;
    if (row >= n) return;
    const double *r = inv + static_cast<size_t>(row) * n;
    double s = 0.0;
    for (int j = lane; j < n; j += 64) s += r[j] * b[j];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off);
    if (lane == 0) x[row] = static_cast<cv_t>(s);
}

__global__ __launch_bounds__(kBlock) void k_axpy1(const cv_t *__restrict__ a, int n, cv_t *__restrict__ x)
{
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) x[i] += a[i];
}

__global__ void k_copy(const cv_t *__restrict__ a, cv_t *__restrict__ b, size_t n)
{
    for (size_t i = blockIdx.x * static_cast<size_t>(blockDim.x) + threadIdx.x; i < n;
         i += static_cast<size_t>(gridDim.x) * blockDim.x)
        b[i] = a[i];
}

// the cycle's way in and out: b = rhs / sqrt(norm2) in the cycle's precision, out = x * sqrt(norm2) (M is linear)
__device__ __forceinline__ double cycle_scale(const double *norm2)
{
    const double v = norm2 ? *norm2 : 1.0;
    return (v > 0.0 && v < 1e300) ? sqrt(v) : 1.0;
}
__global__ void k_cycle_in(const double *__restrict__ rhs, const double *__restrict__ norm2, cv_t *__restrict__ b, size_t n)
{
    const double inv = 1.0 / cycle_scale(norm2);
    for (size_t i = blockIdx.x * static_cast<size_t>(blockDim.x) + threadIdx.x; i < n;
         i += static_cast<size_t>(gridDim.x) * blockDim.x)
        b[i] = static_cast<cv_t>(rhs[i] * inv);
}
__global__ void k_cycle_out(const cv_t *__restrict__ x, const double *__restrict__ norm2, double *__restrict__ out, size_t n)
{
    const double sc = cycle_scale(norm2);
    for (size_t i = blockIdx.x * static_cast<size_t>(blockDim.x) + threadIdx.x; i < n;
         i += static_cast<size_t>(gridDim.x) * blockDim.x)
        out[i] = static_cast<double>(x[i]) * sc;
}

// ---- K-cycle (Notay): two flexible-CG steps on a coarse level, each
// preconditioned by the cycle below it; all scalars stay on the device.
struct KScalars {
    double rho1, alpha1, gamma, beta, alpha2, f1, f2;
    double part[3][256];
};

__global__ __launch_bounds__(kBlock) void k_spmv(const int *__restrict__ rowptr,
                                                const int *__restrict__ col,
                                                const double *__restrict__ val,
                                                const cv_t *__restrict__ x, int n,
                                                cv_t *__restrict__ y)
{
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
        double ax = 0.0;
        for (int p = rowptr[i]; p < rowptr[i + 1]; ++p) ax += val[p] * x[col[p]];
        y[i] = static_cast<cv_t>(ax);
    }
}

__device__ __forceinline__ double kblock_sum(double v, double *lds)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
    if ((threadIdx.x & 63) == 0) lds[threadIdx.x >> 6] = v;
    __syncthreads();
    double s = 0.0;
    if (threadIdx.x == 0)
        for (int w = 0; w < kBlock / 64; ++w) s += lds[w];
    __syncthreads();
    return s;
}

// up to three dot products a_k . b_k in one pass (NULL pairs are skipped)
__global__ __launch_bounds__(kBlock) void k_dots(const cv_t *a0, const cv_t *b0,
                                                const cv_t *a1, const cv_t *b1,
                                                const cv_t *a2, const cv_t *b2, int n,
                                                KScalars *s)
{
    __shared__ double lds[kBlock / 64];
    double d0 = 0.0, d1 = 0.0, d2 = 0.0;
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
        d0 += static_cast<double>(a0[i]) * b0[i];
        if (a1) d1 += static_cast<double>(a1[i]) * b1[i];
        if (a2) d2 += static_cast<double>(a2[i]) * b2[i];
    }
    d0 = kblock_sum(d0, lds);
    d1 = kblock_sum(d1, lds);
    d2 = kblock_sum(d2, lds);
    if (threadIdx.x == 0) { s->part[0][blockIdx.x] = d0; s->part[1][blockIdx.x] = d1; s->part[2][blockIdx.x] = d2; }
}

__global__ __launch_bounds__(kBlock) void k_kfinish(KScalars *s, int stage, int nblocks)
{
    __shared__ double lds[kBlock / 64];
    double t[3];
    for (int k = 0; k < 3; ++k) {
        double d = 0.0;
        for (int i = threadIdx.x; i < nblocks; i += kBlock) d += s->part[k][i];
        t[k] = kblock_sum(d, lds);
    }
    if (threadIdx.x != 0) return;
    if (stage == 1) {                    // rho1 = c1.v1, alpha1 = c1.b
        s->rho1 = t[0];
        s->alpha1 = t[1];
        s->f1 = t[0] > 0.0 ? t[1] / t[0] : 0.0;
        s->f2 = 0.0;
    } else {                             // gamma = c2.v1, beta = c2.v2, alpha2 = c2.r1
        s->gamma = t[0];
        s->beta = t[1];
        s->alpha2 = t[2];
        const double rho2 = s->rho1 > 0.0 ? t[1] - t[0] * t[0] / s->rho1 : 0.0;
        if (rho2 > 0.0 && s->rho1 > 0.0) {
            s->f1 = s->alpha1 / s->rho1 - t[0] * t[2] / (s->rho1 * rho2);
            s->f2 = t[2] / rho2;
        }                                 // else keep the one-step answer
    }
}

// ---- single-level K-cycle: flexible CG(1) on one coarse level, scalars on the device
// stage 1: beta = -(z, q_prev) / (p_prev, q_prev);  stage 2: pq = (p, q), alpha = (p, r) / pq
__global__ __launch_bounds__(kBlock) void k_k1finish(KScalars *s, int stage, int nblocks, int linear)
{
    __shared__ double lds[kBlock / 64];
    double t[2];
    for (int k = 0; k < 2; ++k) {
        double d = 0.0;
        for (int i = threadIdx.x; i < nblocks; i += kBlock) d += s->part[k][i];
        t[k] = kblock_sum(d, lds);
    }
    if (threadIdx.x != 0) return;
    if (stage == 1) s->beta = (s->rho1 != 0.0 && !linear) ? -t[0] / s->rho1 : 0.0;          // rho1 holds (p_prev, q_prev)
    else { s->rho1 = t[0]; s->alpha1 = linear ? 1.0 : (t[0] != 0.0 ? t[1] / t[0] : 0.0); }
}
// p = z + beta p (first: p = z)
__global__ __launch_bounds__(kBlock) void k_k1p(const cv_t *__restrict__ z, int n, const KScalars *s, int first,
                                               cv_t *__restrict__ p)
{
    const double beta = first ? 0.0 : s->beta;
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock)
        p[i] = static_cast<cv_t>(z[i] + beta * (first ? 0.0 : static_cast<double>(p[i])));
}
// x += alpha p (first: x = alpha p);  r -= alpha q
__global__ __launch_bounds__(kBlock) void k_k1xr(const cv_t *__restrict__ p, const cv_t *__restrict__ q, int n,
                                                const KScalars *s, int first, cv_t *__restrict__ x, cv_t *__restrict__ r)
{
    const double alpha = s->alpha1;
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
        x[i] = static_cast<cv_t>((first ? 0.0 : static_cast<double>(x[i])) + alpha * p[i]);
        r[i] = static_cast<cv_t>(r[i] - alpha * q[i]);
    }
}
// q = A p with four lanes per row (the CSR sweeps' walk)
template <class V>
__global__ __launch_bounds__(kBlock) void k_spmv4(const int *__restrict__ rowptr, const int *__restrict__ col,
                                                 const V *__restrict__ val, const cv_t *__restrict__ x, int n,
                                                 cv_t *__restrict__ y)
{
    const int sub = threadIdx.x % kRowLanes;
    const long long groups = static_cast<long long>(gridDim.x) * (kBlock / kRowLanes);
    for (long long i = blockIdx.x * static_cast<long long>(kBlock / kRowLanes) + threadIdx.x / kRowLanes; i < n;
         i += groups * kRowsPerGroup) {
        long long row[kRowsPerGroup];
        double ax[kRowsPerGroup];
#pragma unroll
        for (int u = 0; u < kRowsPerGroup; ++u) row[u] = i + u * groups;
        rows_dot4(rowptr, col, val, x, row, n, sub, ax);
#pragma unroll
        for (int u = 0; u < kRowsPerGroup; ++u)
            if (sub == 0 && row[u] < n) y[row[u]] = static_cast<cv_t>(ax[u]);
    }
}

// r1 = b - f1 v1
__global__ __launch_bounds__(kBlock) void k_kresid(const cv_t *__restrict__ b,
                                                  const cv_t *__restrict__ v1, int n,
                                                  const KScalars *s, cv_t *__restrict__ r1)
{
    const double f1 = s->f1;
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock)
        r1[i] = static_cast<cv_t>(b[i] - f1 * v1[i]);
}

// x = f1 c1 + f2 c2
__global__ __launch_bounds__(kBlock) void k_kcombine(const cv_t *__restrict__ c1,
                                                    const cv_t *__restrict__ c2, int n,
                                                    const KScalars *s, cv_t *__restrict__ x)
{
    const double f1 = s->f1, f2 = s->f2;
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock)
        x[i] = static_cast<cv_t>(f1 * c1[i] + f2 * c2[i]);
}

}  // namespace

size_t amg_workspace_bytes(int rows, int cols)
{
    const size_t n = static_cast<size_t>(rows) * cols;
    // level 0: <= 9n entries (12 B each) + ~60 B/node of vectors and maps; the
    // hierarchy shrinks geometrically (x ~3.5 in total); sort scratch: two key and
    // two value buffers of 9n entries (8 B each) + hipCUB temporaries
    size_t bytes = static_cast<size_t>(4.5 * (9.0 * n * 12 + 96.0 * n));
    bytes += 4 * 9 * n * 8 + 3 * 9 * n * 8;
    bytes += static_cast<size_t>(kMaxDense) * kMaxDense * 8 * 3 + (64u << 20);
    return bytes;
}

#define AMG_TAKE(ptr, type, count)                                                    \
    do {                                                                              \
        ptr = static_cast<type *>(bump.take(sizeof(type) * static_cast<size_t>(count))); \
        if (!ptr) return set_error(SSRS_ERR_INVALID, "amg: workspace exhausted (%s)", #ptr); \
    } while (0)

int amg_setup(AmgHierarchy &h, const double *cond, const uint8_t *fixed, int rows, int cols,
              void *workspace, size_t workspace_bytes, hipStream_t st)
{
    h.levels.clear();
    h.dense_inv = nullptr;
    h.l0_blocks = false;
    Bump bump{static_cast<char *>(workspace), workspace_bytes, 0};
    const int n0 = rows * cols;

    // ---- level 0
    AmgLevel L{};
    L.n = n0;
    AMG_TAKE(L.rowptr, int, n0 + 1);
    int *cnt;
    AMG_TAKE(cnt, int, n0 + 1);
    hipLaunchKernelGGL(k_l0_count, dim3(grid_for(n0)), dim3(kBlock), 0, st, fixed, rows, cols, cnt);
    SSRS_HIP_CHECK(hipMemsetAsync(cnt + n0, 0, sizeof(int), st));
    {
        size_t tb = 0;
        SSRS_HIP_CHECK(hipcub::DeviceScan::ExclusiveSum(nullptr, tb, cnt, L.rowptr, n0 + 1, st));
        void *tmp = bump.take(tb);
        if (!tmp) return set_error(SSRS_ERR_INVALID, "amg: workspace exhausted (scan)");
        SSRS_HIP_CHECK(hipcub::DeviceScan::ExclusiveSum(tmp, tb, cnt, L.rowptr, n0 + 1, st));
    }
    SSRS_HIP_CHECK(hipMemcpyAsync(&L.nnz, L.rowptr + n0, sizeof(int), hipMemcpyDeviceToHost, st));
    SSRS_HIP_CHECK(hipStreamSynchronize(st));
    AMG_TAKE(L.col, int, L.nnz);
    AMG_TAKE(L.val, double, L.nnz);
    hipLaunchKernelGGL(k_l0_fill, dim3(grid_for(n0)), dim3(kBlock), 0, st, cond, fixed, rows, cols,
                       L.rowptr, L.col, L.val);
    SSRS_HIP_CHECK(hipGetLastError());
    {
        double *rinv;
        AMG_TAKE(rinv, double, n0);
        hipLaunchKernelGGL(k_l0_rinv, dim3(grid_for(n0)), dim3(kBlock), 0, st, cond, fixed, static_cast<size_t>(n0), rinv);
        h.l0_rinv = rinv;
        cv_t *rinvc;
        AMG_TAKE(rinvc, cv_t, n0);
        hipLaunchKernelGGL(k_to_cv, dim3(grid_for(n0)), dim3(kBlock), 0, st, rinv, n0, rinvc);
        h.l0_rinvc = rinvc;
        void *slots;
        AMG_TAKE(slots, char, 256);
        h.l0_slots = slots;
        h.l0_fixed = fixed;
        h.l0_rows = rows;
        h.l0_cols = cols;
    }

    // sort scratch sized for level 0 (the largest)
    unsigned long long *keys_a, *keys_b;
    double *vals_a, *vals_b;
    AMG_TAKE(keys_a, unsigned long long, L.nnz);
    AMG_TAKE(keys_b, unsigned long long, L.nnz);
    AMG_TAKE(vals_a, double, L.nnz);
    AMG_TAKE(vals_b, double, L.nnz);
    int *d_count;
    AMG_TAKE(d_count, int, 4);
    size_t sort_tb = 0, red_tb = 0, scan_tb = 0;
    SSRS_HIP_CHECK(hipcub::DeviceRadixSort::SortPairs(nullptr, sort_tb, keys_a, keys_b, vals_a, vals_b,
                                                      L.nnz, 0, 64, st));
    SSRS_HIP_CHECK(hipcub::DeviceReduce::ReduceByKey(nullptr, red_tb, keys_b, keys_a, vals_b, vals_a,
                                                     d_count, hipcub::Sum(), L.nnz, st));
    SSRS_HIP_CHECK(hipcub::DeviceScan::ExclusiveSum(nullptr, scan_tb, cnt, cnt, n0 + 1, st));
    size_t cub_tb = sort_tb > red_tb ? sort_tb : red_tb;
    cub_tb = cub_tb > scan_tb ? cub_tb : scan_tb;
    void *cub_tmp = bump.take(cub_tb);
    if (!cub_tmp) return set_error(SSRS_ERR_INVALID, "amg: workspace exhausted (cub)");

    bool permissive = false;
    for (int lev = 0;; ++lev) {
        const int n = L.n;
        AMG_TAKE(L.dinv, double, n);
        AMG_TAKE(L.dinvc, cv_t, n);
        AMG_TAKE(L.x, cv_t, n);
        AMG_TAKE(L.xt, cv_t, n);
        AMG_TAKE(L.b, cv_t, n);
        AMG_TAKE(L.r, cv_t, n);
        if ((lev >= 1 && lev <= h.kdepth) || (lev >= 1 && lev == h.klevel && h.kinner > 0)) {
            AMG_TAKE(L.kb, cv_t, n);
            AMG_TAKE(L.c1, cv_t, n);
            AMG_TAKE(L.v1, cv_t, n);
            AMG_TAKE(L.v2, cv_t, n);
            void *ks;
            AMG_TAKE(ks, char, sizeof(KScalars));
            L.kscal = ks;
        }
        hipLaunchKernelGGL(k_dinv, dim3(grid_for(n)), dim3(kBlock), 0, st, L.rowptr, L.col, L.val, n, L.dinv);
        hipLaunchKernelGGL(k_to_cv, dim3(grid_for(n)), dim3(kBlock), 0, st, L.dinv, n, L.dinvc);
        L.agg = nullptr;
        L.memptr = nullptr;
        L.memidx = nullptr;
        L.nc = 0;
        if ((n <= kMaxDense && lev > 0) || lev >= 48) { h.levels.push_back(L); break; }

        // ---- pairwise aggregation
        int *match, *prop, *joined, *flag, *cid;
        AMG_TAKE(L.agg, int, n);
        AMG_TAKE(match, int, n);
        AMG_TAKE(prop, int, n);
        AMG_TAKE(joined, int, n);
        AMG_TAKE(flag, int, n + 1);
        AMG_TAKE(cid, int, n + 1);
        int nc = 0;
        const bool blocks0 = lev == 0 && h.symmetric && !std::getenv("SSRS_AMG_NO_BLOCKS");      // A/B: pairwise matching on level 0 too
        for (int attempt = 0; attempt < 2; ++attempt) {
            if (blocks0) {
                const long long nb = static_cast<long long>((rows + 1) / 2) * ((cols + 1) / 2);
                hipLaunchKernelGGL(k_block_agg, dim3(grid_for(static_cast<size_t>(nb))), dim3(kBlock), 0, st, L.rowptr, L.col, L.val,
                                   L.dinv, rows, cols, match, joined, flag);
                h.l0_blocks = true;
                SSRS_HIP_CHECK(hipMemsetAsync(flag + n, 0, sizeof(int), st));
                size_t tb = cub_tb;
                SSRS_HIP_CHECK(hipcub::DeviceScan::ExclusiveSum(cub_tmp, tb, flag, cid, n + 1, st));
                SSRS_HIP_CHECK(hipMemcpyAsync(&nc, cid + n, sizeof(int), hipMemcpyDeviceToHost, st));
                SSRS_HIP_CHECK(hipStreamSynchronize(st));
                break;
            }
            SSRS_HIP_CHECK(hipMemsetAsync(match, 0xFF, sizeof(int) * n, st));
            for (int round = 0; round < kMatchRounds; ++round) {
                // symmetric (default): every round strict, any positive coupling once the
                // strict coarsening has stalled; one-sided: strong_rounds strict, then any
                const int mode = h.symmetric ? (permissive ? 2 : 0) : (round < h.strong_rounds ? 1 : 2);
                hipLaunchKernelGGL(k_propose, dim3(grid_for(n)), dim3(kBlock), 0, st, L.rowptr, L.col, L.val, n,
                                   match, prop, L.dinv, mode);
                hipLaunchKernelGGL(k_accept, dim3(grid_for(n)), dim3(kBlock), 0, st, n, prop, match);
            }
            hipLaunchKernelGGL(k_join, dim3(grid_for(n)), dim3(kBlock), 0, st, L.rowptr, L.col, L.val, n, match,
                               L.dinv, joined, flag);
            SSRS_HIP_CHECK(hipMemsetAsync(flag + n, 0, sizeof(int), st));
            {
                size_t tb = cub_tb;
                SSRS_HIP_CHECK(hipcub::DeviceScan::ExclusiveSum(cub_tmp, tb, flag, cid, n + 1, st));
            }
            SSRS_HIP_CHECK(hipMemcpyAsync(&nc, cid + n, sizeof(int), hipMemcpyDeviceToHost, st));
            SSRS_HIP_CHECK(hipStreamSynchronize(st));
            if (!(h.symmetric && !permissive && (nc == 0 || nc > 0.85 * n))) break;
            permissive = true;                     // strict coarsening stalled: redo this level
        }
        if (nc == 0 || nc > 0.93 * n) {           // nothing (left) to coarsen
            L.agg = nullptr;
            h.levels.push_back(L);
            break;
        }
        AMG_TAKE(L.memptr, int, static_cast<size_t>(nc) + 1);
        AMG_TAKE(L.memidx, int, n);
        hipLaunchKernelGGL(k_assign_agg, dim3(grid_for(n)), dim3(kBlock), 0, st, L.rowptr, n, match, joined, cid,
                           L.agg, keys_a);
        {
            size_t tb = cub_tb;
            SSRS_HIP_CHECK(hipcub::DeviceRadixSort::SortKeys(cub_tmp, tb, keys_a, keys_b, n, 0, 64, st));
        }
        hipLaunchKernelGGL(k_member_lists, dim3(grid_for(n)), dim3(kBlock), 0, st, keys_b, n, nc, L.memptr, L.memidx);
        L.nc = nc;

        // ---- Galerkin coarse matrix: sort (I,J) keys, sum duplicates
        hipLaunchKernelGGL(k_galerkin_keys, dim3(grid_for(n)), dim3(kBlock), 0, st, L.rowptr, L.col, L.val, n, L.agg,
                           keys_a, vals_a);
        {
            size_t tb = cub_tb;
            SSRS_HIP_CHECK(hipcub::DeviceRadixSort::SortPairs(cub_tmp, tb, keys_a, keys_b, vals_a, vals_b, L.nnz, 0, 64, st));
            tb = cub_tb;
            SSRS_HIP_CHECK(hipcub::DeviceReduce::ReduceByKey(cub_tmp, tb, keys_b, keys_a, vals_b, vals_a, d_count,
                                                             hipcub::Sum(), L.nnz, st));
        }
        int nuniq = 0;
        SSRS_HIP_CHECK(hipMemcpyAsync(&nuniq, d_count, sizeof(int), hipMemcpyDeviceToHost, st));
        unsigned long long last_key = 0;
        SSRS_HIP_CHECK(hipStreamSynchronize(st));
        SSRS_HIP_CHECK(hipMemcpyAsync(&last_key, keys_a + (nuniq - 1), sizeof(last_key), hipMemcpyDeviceToHost, st));
        SSRS_HIP_CHECK(hipStreamSynchronize(st));
        if (last_key == ~0ull) --nuniq;            // the parked bucket

        if (std::getenv("SSRS_PROGRESS"))
            fprintf(stderr, "[amg] level %d: %d rows, %d nnz -> %d rows, %d nnz%s\n", lev, n, L.nnz, nc, nuniq,
                    permissive ? " (permissive)" : "");
        AmgLevel C{};
        C.n = nc;
        C.nnz = nuniq;
        AMG_TAKE(C.rowptr, int, nc + 1);
        AMG_TAKE(C.col, int, nuniq);
        AMG_TAKE(C.val, double, nuniq);
        hipLaunchKernelGGL(k_unpack_coarse, dim3(grid_for(nuniq)), dim3(kBlock), 0, st, keys_a, nuniq, nc, C.rowptr, C.col);
        SSRS_HIP_CHECK(hipMemcpyAsync(C.val, vals_a, sizeof(double) * nuniq, hipMemcpyDeviceToDevice, st));
        SSRS_HIP_CHECK(hipGetLastError());
        h.levels.push_back(L);
        L = C;
    }
    if (h.levels.empty()) return set_error(SSRS_ERR_INVALID, "amg: empty hierarchy");

    // ---- f32 copies of the coarse matrices for the cycle, in the (now free) sort scratch
    if (!std::getenv("SSRS_AMG_F64")) {
        float *pool = reinterpret_cast<float *>(vals_b);
        const size_t room = 2 * static_cast<size_t>(h.levels[0].nnz);
        size_t used = 0;
        for (size_t lev = 1; lev < h.levels.size(); ++lev) {
            AmgLevel &Lv = h.levels[lev];
            const size_t need = (static_cast<size_t>(Lv.nnz) + 63) / 64 * 64;
            if (used + need > room) break;                   // (never seen: coarse levels shrink 2.3x each)
            Lv.val32 = pool + used;
            used += need;
            hipLaunchKernelGGL(k_val32, dim3(grid_for(Lv.n)), dim3(kBlock), 0, st, Lv.rowptr, Lv.col, Lv.val, Lv.n, Lv.val32);
        }
        SSRS_HIP_CHECK(hipGetLastError());
    }
    // ---- sliced ELL copies of the large levels (the ones whose sweeps ran four lanes per row)
    if (!std::getenv("SSRS_AMG_NO_SELL")) {
        for (size_t lev = 1; lev < h.levels.size(); ++lev) {
            AmgLevel &Lv = h.levels[lev];
            if (Lv.n < kSellRows || !Lv.val32) continue;
            const int nsl = (Lv.n + 63) / 64;
            int *cnt;
            AMG_TAKE(cnt, int, nsl + 1);
            AMG_TAKE(Lv.sell_ptr, int, nsl + 1);
            hipLaunchKernelGGL(k_sell_widths, dim3(grid_for(static_cast<size_t>(nsl) * 64)), dim3(kBlock), 0, st, Lv.rowptr, Lv.n, nsl, cnt);
            size_t tb = cub_tb;
            SSRS_HIP_CHECK(hipcub::DeviceScan::ExclusiveSum(cub_tmp, tb, cnt, Lv.sell_ptr, nsl + 1, st));
            int total = 0;
            SSRS_HIP_CHECK(hipMemcpyAsync(&total, Lv.sell_ptr + nsl, sizeof(int), hipMemcpyDeviceToHost, st));
            SSRS_HIP_CHECK(hipStreamSynchronize(st));
            if (total <= 0 || static_cast<size_t>(total) > 2 * static_cast<size_t>(Lv.nnz) + 64u * 64u) { Lv.sell_ptr = nullptr; continue; }   // (padding would double it: CSR)
            AMG_TAKE(Lv.sell_col, int, total);
            AMG_TAKE(Lv.sell_val, float, total);
            hipLaunchKernelGGL(k_sell_fill, dim3(grid_for(static_cast<size_t>(nsl) * 64)), dim3(kBlock), 0, st, Lv.rowptr, Lv.col, Lv.val32, Lv.n, nsl,
                               Lv.sell_ptr, Lv.sell_col, Lv.sell_val);
        }
        SSRS_HIP_CHECK(hipGetLastError());
    }

    // ---- dense inverse of the coarsest level when it is small enough
    AmgLevel &B = h.levels.back();
    if (B.n <= kMaxDense) {
        const int n = B.n;
        double *aug, *colk;
        AMG_TAKE(aug, double, static_cast<size_t>(n) * 2 * n);
        AMG_TAKE(colk, double, n);
        AMG_TAKE(h.dense_inv, double, static_cast<size_t>(n) * n);
        hipLaunchKernelGGL(k_dense_fill, dim3(grid_for(static_cast<size_t>(n) * 2 * n)), dim3(kBlock), 0, st, B.rowptr, B.col, B.val, n, aug);
        hipLaunchKernelGGL(k_dense_scatter, dim3(grid_for(n)), dim3(kBlock), 0, st, B.rowptr, B.col, B.val, n, aug);
        for (int k = 0; k < n; ++k) {
            hipLaunchKernelGGL(k_gj_save, dim3(grid_for(n)), dim3(kBlock), 0, st, aug, n, k, colk);
            hipLaunchKernelGGL(k_gj_elim, dim3(grid_for(static_cast<size_t>(n) * 2 * n)), dim3(kBlock), 0, st, aug, n, k, colk);
        }
        hipLaunchKernelGGL(k_gj_finish, dim3(grid_for(static_cast<size_t>(n) * n)), dim3(kBlock), 0, st, aug, n, h.dense_inv);
        SSRS_HIP_CHECK(hipGetLastError());
    }
    SSRS_HIP_CHECK(hipStreamSynchronize(st));
    h.workspace_used = bump.off;
    return SSRS_OK;
}

static void solve_level(AmgHierarchy &h, size_t lev, hipStream_t st);


// one wave per slice of 64 rows, at most 16 384 blocks (the kernels stride)
static dim3 sell_grid(int n)
{
    const size_t blocks = (static_cast<size_t>((n + 63) / 64) + kBlock / 64 - 1) / (kBlock / 64);
    return dim3(static_cast<unsigned>(blocks < 1 ? 1 : (blocks > 16384 ? 16384 : blocks)));
}

// level-0 stencil kernels: blockIdx.y = row, a block = 4 waves x 62 cells of it
static dim3 l0_grid(const AmgHierarchy &h)
{
    const int per_block = (kBlock / 64) * kL0Cols;
    return dim3(static_cast<unsigned>((h.l0_cols + per_block - 1) / per_block), static_cast<unsigned>(h.l0_rows));
}

static void launch_jacobi(AmgHierarchy &h, size_t lev, const cv_t *x, cv_t *xn, hipStream_t st, double w = kOmega)
{
    AmgLevel &L = h.levels[lev];
    if (lev == 0 && h.l0_rinvc && !getenv("SSRS_AMG_L0_CSR")) {
        const L0Stencil a{h.l0_rinvc, h.l0_fixed, h.l0_rows, h.l0_cols};
        hipLaunchKernelGGL(k_l0_jacobi, l0_grid(h), dim3(kBlock), 0, st, a, L.dinvc, L.b, x, xn, static_cast<cv_t>(w));
    } else {
        const dim3 g4(grid_for((static_cast<size_t>(L.n) + kRowsPerGroup - 1) / kRowsPerGroup * kRowLanes)), g1(grid_for(L.n));
        if (L.sell_ptr && L.sell_val)
            hipLaunchKernelGGL(k_sweep_sell<true>, sell_grid(L.n), dim3(kBlock), 0, st, L.sell_ptr, L.sell_col, L.sell_val, L.dinvc, L.b, x, L.n,
                               (L.n + 63) / 64, xn, w);
        else if (L.n >= kVectorRows && L.val32)
            hipLaunchKernelGGL(k_jacobi4<float>, g4, dim3(kBlock), 0, st, L.rowptr, L.col, L.val32, L.dinvc, L.b, x, L.n, xn, w);
        else if (L.n >= kVectorRows)
            hipLaunchKernelGGL(k_jacobi4<double>, g4, dim3(kBlock), 0, st, L.rowptr, L.col, L.val, L.dinvc, L.b, x, L.n, xn, w);
        else if (L.val32)
            hipLaunchKernelGGL(k_jacobi<float>, g1, dim3(kBlock), 0, st, L.rowptr, L.col, L.val32, L.dinvc, L.b, x, L.n, xn, w);
        else
            hipLaunchKernelGGL(k_jacobi<double>, g1, dim3(kBlock), 0, st, L.rowptr, L.col, L.val, L.dinvc, L.b, x, L.n, xn, w);
    }
}

static void launch_residual(AmgHierarchy &h, size_t lev, hipStream_t st, const cv_t *x = nullptr)
{
    AmgLevel &L = h.levels[lev];
    if (!x) x = L.x;
    if (lev == 0 && h.l0_rinvc && !getenv("SSRS_AMG_L0_CSR")) {
        const L0Stencil a{h.l0_rinvc, h.l0_fixed, h.l0_rows, h.l0_cols};
        hipLaunchKernelGGL(k_l0_residual, l0_grid(h), dim3(kBlock), 0, st, a, L.b, x, L.r);
    } else {
        const dim3 g4(grid_for((static_cast<size_t>(L.n) + kRowsPerGroup - 1) / kRowsPerGroup * kRowLanes)), g1(grid_for(L.n));
        if (L.sell_ptr && L.sell_val)
            hipLaunchKernelGGL(k_sweep_sell<false>, sell_grid(L.n), dim3(kBlock), 0, st, L.sell_ptr, L.sell_col, L.sell_val, L.dinvc, L.b, x, L.n,
                               (L.n + 63) / 64, L.r, 0.0);
        else if (L.n >= kVectorRows && L.val32)
            hipLaunchKernelGGL(k_residual4<float>, g4, dim3(kBlock), 0, st, L.rowptr, L.col, L.val32, L.b, x, L.n, L.r);
        else if (L.n >= kVectorRows)
            hipLaunchKernelGGL(k_residual4<double>, g4, dim3(kBlock), 0, st, L.rowptr, L.col, L.val, L.b, x, L.n, L.r);
        else if (L.val32)
            hipLaunchKernelGGL(k_residual<float>, g1, dim3(kBlock), 0, st, L.rowptr, L.col, L.val32, L.b, x, L.n, L.r);
        else
            hipLaunchKernelGGL(k_residual<double>, g1, dim3(kBlock), 0, st, L.rowptr, L.col, L.val, L.b, x, L.n, L.r);
    }
}

// One multigrid step at `lev`: rhs L.b -> L.x (smooth, coarse solve, smooth)
static void cycle(AmgHierarchy &h, size_t lev, hipStream_t st)
{
    AmgLevel &L = h.levels[lev];
    const int n = L.n, g = grid_for(n);
    if (lev + 1 == h.levels.size()) {
        if (h.dense_inv) {
            // x = inv b, then one step of iterative refinement against the sparse
            // operator: the Gauss-Jordan inverse of a matrix whose entries span
            // 1e-10 .. 1 is only accurate to a few digits
            const int gd = grid_for(static_cast<size_t>(n) * 64);
            hipLaunchKernelGGL(k_dense_apply, dim3(gd), dim3(kBlock), 0, st, h.dense_inv, L.b, n, L.x);
            hipLaunchKernelGGL(k_residual, dim3(g), dim3(kBlock), 0, st, L.rowptr, L.col, L.val, L.b, L.x, n, L.r);
            hipLaunchKernelGGL(k_dense_apply, dim3(gd), dim3(kBlock), 0, st, h.dense_inv, L.r, n, L.xt);
            hipLaunchKernelGGL(k_axpy1, dim3(g), dim3(kBlock), 0, st, L.xt, n, L.x);
        } else {                                   // stalled coarsening: relax
            hipLaunchKernelGGL(k_jacobi_first, dim3(g), dim3(kBlock), 0, st, L.dinvc, L.b, n, L.x, kOmega);
            for (int s = 0; s < 20; ++s) {
                hipLaunchKernelGGL(k_jacobi, dim3(g), dim3(kBlock), 0, st, L.rowptr, L.col, L.val, L.dinvc, L.b, L.x, n, L.xt, kOmega);
                hipLaunchKernelGGL(k_jacobi, dim3(g), dim3(kBlock), 0, st, L.rowptr, L.col, L.val, L.dinvc, L.b, L.xt, n, L.x, kOmega);
            }
        }
        return;
    }
    if (!h.robust && lev == 0 && h.fuse0) {
        if constexpr (sizeof(cv_t) == sizeof(double)) {
            const double *rinv = reinterpret_cast<const double *>(h.l0_rinvc);
            const L0Slots *slots = static_cast<const L0Slots *>(h.l0_slots);
            AmgLevel &C1 = h.levels[1];
            if (!std::getenv("SSRS_AMG_L0_ONE_ROW")) {
                dim3 g2 = l0_grid(h);
                g2.y = (g2.y + 1) / 2;
                const bool in_place = h.l0_blocks && (kL0Cols % 2) == 0;       // restriction inside the pre pass
                hipLaunchKernelGGL(k_l0_pre2, g2, dim3(kBlock), 0, st, rinv, h.l0_rows, h.l0_cols, reinterpret_cast<const double *>(L.dinvc),
                                   slots, h.om[0], reinterpret_cast<double *>(L.xt), reinterpret_cast<double *>(L.r),
                                   in_place ? L.agg : static_cast<const int *>(nullptr), reinterpret_cast<double *>(C1.b));
                if (!in_place) hipLaunchKernelGGL(k_restrict, dim3(grid_for(C1.n)), dim3(kBlock), 0, st, L.memptr, L.memidx, L.r, C1.n, C1.b);
                solve_level(h, 1, st);
                hipLaunchKernelGGL(k_l0_post2, g2, dim3(kBlock), 0, st, rinv, h.l0_rows, h.l0_cols, reinterpret_cast<const double *>(L.dinvc),
                                   slots, L.agg, reinterpret_cast<const double *>(C1.x), reinterpret_cast<const double *>(L.xt), h.om[0]);
                return;
            }
            hipLaunchKernelGGL(k_l0_pre_fused, l0_grid(h), dim3(kBlock), 0, st, rinv, h.l0_rows, h.l0_cols,
                               reinterpret_cast<const double *>(L.dinvc), slots, h.om[0], reinterpret_cast<double *>(L.xt),
                               reinterpret_cast<double *>(L.r));
            hipLaunchKernelGGL(k_restrict, dim3(grid_for(C1.n)), dim3(kBlock), 0, st, L.memptr, L.memidx, L.r, C1.n, C1.b);
            solve_level(h, 1, st);
            hipLaunchKernelGGL(k_l0_post_fused, l0_grid(h), dim3(kBlock), 0, st, rinv, h.l0_rows, h.l0_cols,
                               reinterpret_cast<const double *>(L.dinvc), slots, L.agg, reinterpret_cast<const double *>(C1.x),
                               reinterpret_cast<const double *>(L.xt), h.om[0]);
        }
        return;
    }
    if (!h.robust && (lev == 0 ? h.nu0 : h.nuc) == 1) {
        // V(1,1): x = w D^-1 b, coarse correction, one sweep with the same step (self-adjoint in the D inner product);
        // the iterate lives in L.xt until the last sweep writes L.x
        hipLaunchKernelGGL(k_jacobi_first, dim3(g), dim3(kBlock), 0, st, L.dinvc, L.b, n, L.xt, h.om[0]);
        launch_residual(h, lev, st, L.xt);
        AmgLevel &C1 = h.levels[lev + 1];
        hipLaunchKernelGGL(k_restrict, dim3(grid_for(C1.n)), dim3(kBlock), 0, st, L.memptr, L.memidx, L.r, C1.n, C1.b);
        solve_level(h, lev + 1, st);
        hipLaunchKernelGGL(k_prolong_add, dim3(g), dim3(kBlock), 0, st, L.agg, C1.x, n, L.xt);
        launch_jacobi(h, lev, L.xt, L.x, st, h.om[0]);
        return;
    }
    // pre-smoothing: 2*sweeps Jacobi sweeps from x = 0
    // (step sizes of a pair of sweeps: h.om[0], h.om[1] before the coarse correction, the same in reverse after it --
    // the smoother stays self-adjoint in the D inner product whatever the two are)
    hipLaunchKernelGGL(k_jacobi_first, dim3(g), dim3(kBlock), 0, st, L.dinvc, L.b, n, L.xt, h.om[0]);
    launch_jacobi(h, lev, L.xt, L.x, st, h.om[1]);
    for (int s = 1; s < h.sweeps; ++s) {
        launch_jacobi(h, lev, L.x, L.xt, st, h.om[0]);
        launch_jacobi(h, lev, L.xt, L.x, st, h.om[1]);
    }
    launch_residual(h, lev, st);
    AmgLevel &C = h.levels[lev + 1];
    hipLaunchKernelGGL(k_restrict, dim3(grid_for(C.n)), dim3(kBlock), 0, st, L.memptr, L.memidx, L.r, C.n, C.b);
    solve_level(h, lev + 1, st);
    hipLaunchKernelGGL(k_prolong_add, dim3(g), dim3(kBlock), 0, st, L.agg, C.x, n, L.x);
    // post-smoothing
    for (int s = 0; s < h.sweeps; ++s) {
        launch_jacobi(h, lev, L.x, L.xt, st, h.om[1]);
        launch_jacobi(h, lev, L.xt, L.x, st, h.om[0]);
    }
}

// Approximate solve of level `lev` (rhs L.b -> L.x): on the first `kdepth`
// coarse levels two FCG steps preconditioned by cycle() (K-cycle), below that a
// plain V-cycle, on the last level the dense inverse.
static void solve_level(AmgHierarchy &h, size_t lev, hipStream_t st)
{
    AmgLevel &L = h.levels[lev];
    if (lev + 1 != h.levels.size() && static_cast<int>(lev) == h.klevel && h.kinner > 0 && L.kscal && !h.robust) {
        // single-level K-cycle: kinner steps of flexible CG(1) on this level's system, each preconditioned by the
        // cycle from here down.  r = L.kb, x = L.c1, p = L.v1, q = L.v2, z = L.x
        const int n = L.n, g = grid_for(n);
        const int gb = g > 256 ? 256 : g;
        const dim3 g4(grid_for((static_cast<size_t>(n) + kRowsPerGroup - 1) / kRowsPerGroup * kRowLanes));
        KScalars *ks = static_cast<KScalars *>(L.kscal);
        const int klin = std::getenv("SSRS_AMG_K_LINEAR") != nullptr ? 1 : 0;     // experiment: plain repeated cycles (alpha 1, beta 0)
        hipLaunchKernelGGL(k_copy, dim3(g), dim3(256), 0, st, L.b, L.kb, static_cast<size_t>(n));
        for (int k = 0; k < h.kinner; ++k) {
            if (k > 0) hipLaunchKernelGGL(k_copy, dim3(g), dim3(256), 0, st, L.kb, L.b, static_cast<size_t>(n));
            cycle(h, lev, st);                                                             // z = B r (in L.x)
            if (k > 0) {
                hipLaunchKernelGGL(k_dots, dim3(gb), dim3(kBlock), 0, st, L.x, L.v2, nullptr, nullptr, nullptr, nullptr, n, ks);
                hipLaunchKernelGGL(k_k1finish, dim3(1), dim3(kBlock), 0, st, ks, 1, gb, klin);
            }
            hipLaunchKernelGGL(k_k1p, dim3(g), dim3(kBlock), 0, st, L.x, n, ks, k == 0 ? 1 : 0, L.v1);
            // q = A p from the f64 entries: the f32 copies the sweeps read round the diagonal up and the off-diagonals toward
            // zero, which multiplies the energy (p, A p) of a floating cluster's level -- a row sum of 1e-8 of the diagonal --
            // by ~100; the step sizes would come out 100 times too small (measured: alpha 0.01, 1 555 outer iterations)
            hipLaunchKernelGGL(k_spmv4<double>, g4, dim3(kBlock), 0, st, L.rowptr, L.col, L.val, L.v1, n, L.v2);
            hipLaunchKernelGGL(k_dots, dim3(gb), dim3(kBlock), 0, st, L.v1, L.v2, L.v1, L.kb, nullptr, nullptr, n, ks);
            hipLaunchKernelGGL(k_k1finish, dim3(1), dim3(kBlock), 0, st, ks, 2, gb, klin);
            hipLaunchKernelGGL(k_k1xr, dim3(g), dim3(kBlock), 0, st, L.v1, L.v2, n, ks, k == 0 ? 1 : 0, L.c1, L.kb);
        }
        hipLaunchKernelGGL(k_copy, dim3(g), dim3(256), 0, st, L.c1, L.x, static_cast<size_t>(n));
        return;
    }
    if (lev + 1 == h.levels.size() || static_cast<int>(lev) > h.kdepth || !L.kscal || (h.klevel > 0 && h.kinner > 0)) {
        cycle(h, lev, st);
        return;
    }
    const int n = L.n, g = grid_for(n);
    int gb = g > 256 ? 256 : g;
    KScalars *ks = static_cast<KScalars *>(L.kscal);
    hipLaunchKernelGGL(k_copy, dim3(g), dim3(256), 0, st, L.b, L.kb, static_cast<size_t>(n));
    cycle(h, lev, st);                                                                    // c1 = B b
    hipLaunchKernelGGL(k_copy, dim3(g), dim3(256), 0, st, L.x, L.c1, static_cast<size_t>(n));
    hipLaunchKernelGGL(k_spmv, dim3(g), dim3(kBlock), 0, st, L.rowptr, L.col, L.val, L.c1, n, L.v1);
    hipLaunchKernelGGL(k_dots, dim3(gb), dim3(kBlock), 0, st, L.c1, L.v1, L.c1, L.kb, nullptr, nullptr, n, ks);
    hipLaunchKernelGGL(k_kfinish, dim3(1), dim3(kBlock), 0, st, ks, 1, gb);
    hipLaunchKernelGGL(k_kresid, dim3(g), dim3(kBlock), 0, st, L.kb, L.v1, n, ks, L.b);   // r1 -> rhs
    cycle(h, lev, st);                                                                    // c2 = B r1 (in L.x)
    hipLaunchKernelGGL(k_spmv, dim3(g), dim3(kBlock), 0, st, L.rowptr, L.col, L.val, L.x, n, L.v2);
    hipLaunchKernelGGL(k_dots, dim3(gb), dim3(kBlock), 0, st, L.x, L.v1, L.x, L.v2, L.x, L.b, n, ks);
    hipLaunchKernelGGL(k_kfinish, dim3(1), dim3(kBlock), 0, st, ks, 2, gb);
    hipLaunchKernelGGL(k_kcombine, dim3(g), dim3(kBlock), 0, st, L.c1, L.x, n, ks, L.xt);
    hipLaunchKernelGGL(k_copy, dim3(g), dim3(256), 0, st, L.xt, L.x, static_cast<size_t>(n));
}

// The V-cycle is ~10 levels x ~8 small kernels: on the coarse levels it is launch
// bound (0.9 ms per PCG iteration at 500 x 600 was mostly launch gaps).  Its
// structure and every pointer are fixed after setup, so the whole cycle is
// captured once into a hipGraph and replayed; rhs/out staging stays outside.
// Capture needs a real (non-null) stream; on the null stream, or if capture
// fails, the kernels are launched directly.
static void ensure_graph(AmgHierarchy &h, hipStream_t st)
{
    const int which = h.robust ? 1 : 0;
    if (h.graph_tried[which]) return;
    h.graph_tried[which] = true;
    if (st == nullptr) return;
    hipGraph_t graph = nullptr;
    hipError_t e = hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal);
    if (e != hipSuccess) {
        if (std::getenv("SSRS_PROGRESS")) fprintf(stderr, "[amg] begin capture failed: %s\n", hipGetErrorString(e));
        (void)hipGetLastError();
        return;
    }
    cycle(h, 0, st);
    e = hipStreamEndCapture(st, &graph);
    if (e != hipSuccess || graph == nullptr) {
        if (std::getenv("SSRS_PROGRESS")) fprintf(stderr, "[amg] end capture failed: %s\n", hipGetErrorString(e));
        (void)hipGetLastError();
        return;
    }
    hipGraphExec_t exec = nullptr;
    e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
    if (e == hipSuccess) h.graph_exec[which] = exec;
    else (void)hipGetLastError();
    if (std::getenv("SSRS_PROGRESS"))
        fprintf(stderr, "[amg] V-cycle graph: %s\n", e == hipSuccess ? "captured" : hipGetErrorString(e));
    (void)hipGraphDestroy(graph);
}

void amg_release(AmgHierarchy &h)
{
    for (int k = 0; k < 2; ++k) {
        if (h.graph_exec[k]) (void)hipGraphExecDestroy(static_cast<hipGraphExec_t>(h.graph_exec[k]));
        h.graph_exec[k] = nullptr;
    }
}

void amg_apply(AmgHierarchy &h, const double *rhs, double *out, const double *norm2, hipStream_t st, bool robust)
{
    AmgLevel &L = h.levels[0];
    h.robust = robust;
    h.fuse0 = sizeof(cv_t) == sizeof(double) && h.l0_rinvc && h.l0_slots && h.nu0 == 1 && h.levels.size() > 1 && h.levels[0].agg &&
              !std::getenv("SSRS_AMG_L0_CSR") && !std::getenv("SSRS_AMG_NO_FUSE");
    ensure_graph(h, st);
    if (!robust && h.fuse0) {
        // the fused level 0 reads `rhs` and writes `out` itself: no staging copies
        hipLaunchKernelGGL(k_set_slots, dim3(1), dim3(1), 0, st, static_cast<L0Slots *>(h.l0_slots), rhs, out);
        void *fexec = h.graph_exec[0];
        if (fexec == nullptr || hipGraphLaunch(static_cast<hipGraphExec_t>(fexec), st) != hipSuccess) cycle(h, 0, st);
        return;
    }
    if (sizeof(cv_t) == sizeof(double)) norm2 = nullptr;      // the scaling only serves the f32 option's range
    hipLaunchKernelGGL(k_cycle_in, dim3(grid_for(L.n)), dim3(256), 0, st, rhs, norm2, L.b, static_cast<size_t>(L.n));
    void *exec = h.graph_exec[robust ? 1 : 0];
    if (exec == nullptr || hipGraphLaunch(static_cast<hipGraphExec_t>(exec), st) != hipSuccess)
        cycle(h, 0, st);
    hipLaunchKernelGGL(k_cycle_out, dim3(grid_for(L.n)), dim3(256), 0, st, h.levels[0].x, norm2, out, static_cast<size_t>(L.n));
}

}  // namespace ssrs
