// Internal interface of the aggregation-AMG preconditioner (amg.hip) used by
// the potential solver (potential.hip).
#pragma once
#include <vector>

#include "common.h"

namespace ssrs {

// Vectors of the V-cycle: f64.  An f32 cycle (-DSSRS_AMG_CYCLE_F32: half the bytes in every kernel of it; the
// right-hand side is scaled to unit norm on the way in, so range is no issue) was built and measured in round 4
// and does NOT work on this operator: a conductive cluster that floats in dead terrain is tied to its surroundings
// 1e-8 .. 1e-10 times more weakly than its cells are tied to each other, so the net residual of the cluster -- what
// the coarse levels need in order to set its level -- is a sum over its cells that cancels to that relative size;
// f32 storage of the residual or of x (6e-8) erases it, and the outer iteration stalls at |r|/|b| ~ 1e-8
// (configs[3]'s field: 2000 iterations, 1.5e-8; profiles/r04_k5.md).  The option stays for that record only.
#ifdef SSRS_AMG_CYCLE_F32
typedef float cv_t;
#else
typedef double cv_t;
#endif

struct AmgLevel {
    int n = 0, nnz = 0, nc = 0;
    int *rowptr = nullptr, *col = nullptr;
    double *val = nullptr, *dinv = nullptr;
    float *val32 = nullptr;    // the same entries rounded to f32 for the cycle's sweeps (levels >= 1)
    // the same entries once more in sliced ELL form (slices of 64 rows, entry k of row 64 s + l at sell_ptr[s] + 64 k + l,
    // short rows padded with (own column, 0)): a wave reads 64 consecutive entries per load and its 64 gathers of x go to
    // neighbouring rows' neighbours -- the large levels' sweeps (amg.hip: k_sweep_sell; SSRS_AMG_NO_SELL: the CSR kernels)
    int *sell_ptr = nullptr, *sell_col = nullptr;
    float *sell_val = nullptr;
    int *agg = nullptr;        // fine node -> coarse node (-1: isolated row), NULL on the last level
    int *memptr = nullptr;     // coarse node I -> its fine nodes memidx[memptr[I] .. memptr[I+1])
    int *memidx = nullptr;
    cv_t *dinvc = nullptr;     // dinv in the cycle's precision (dinv itself stays f64: the set-up's strength test reads it)
    cv_t *x = nullptr, *xt = nullptr, *b = nullptr, *r = nullptr;
    // K-cycle scratch (levels 1 .. kdepth)
    cv_t *kb = nullptr, *c1 = nullptr, *v1 = nullptr, *v2 = nullptr;
    void *kscal = nullptr;
};

struct AmgHierarchy {
    std::vector<AmgLevel> levels;
    double *dense_inv = nullptr;   // inverse of the last level when it is small
    size_t workspace_used = 0;
    void *graph_exec[2] = {nullptr, nullptr};   // hipGraphExec_t of the captured fast / robust cycle (NULL: direct launches)
    bool graph_tried[2] = {false, false};
    bool robust = false;           // the cycle being run / captured: V(2,2) everywhere instead of (nu0, nuc)
    bool symmetric = true;         // symmetric strength of connection (amg.hip: strong_link)
    int strong_rounds = 4;         // matching rounds restricted to strong couplings (of 8)
    int kdepth = 0;                // coarse levels 1..kdepth use the K-cycle (0 = V-cycle)
    int klevel = 0, kinner = 0;    // ONE level solved by `kinner` flexible-CG steps preconditioned by the cycle below it
                                   // (0: off; SSRS_AMG_K=level,inner).  Unlike kdepth's nesting (2^depth visits of the deep
                                   // levels) the levels below are visited kinner times
    int sweeps = 1;                // pairs of Jacobi sweeps before and after the coarse correction
    int nu0 = 1, nuc = 1;          // Jacobi sweeps before / after the coarse correction on level 0 / on the coarser levels:
                                   // 1 (default since round 4: V(1,1), 430 iterations x 5.2 ms at C2) or 2 (rounds 1-3: V(2,2),
                                   // 385 x 8.1 ms); SSRS_AMG_NU=a,b
    double om[2] = {0.7, 0.7};     // step sizes of a pair of sweeps (SSRS_AMG_OMEGAS=a,b for experiments)
    // level 0 applied matrix-free (amg.hip: L0Stencil)
    const double *l0_rinv = nullptr;   // +-1 / cond in f64: the outer operator (potential.hip) reads it
    const cv_t *l0_rinvc = nullptr;    // the same in the cycle's precision
    // level 0 of the V(1,1) cycle fused into two stencil passes that read the caller's right-hand side and write the
    // caller's result directly (amg.hip: k_l0_pre_fused / k_l0_post_fused); the two pointers travel through a device slot
    // so that the captured graph serves every (rhs, out) pair.  A/B: SSRS_AMG_NO_FUSE
    void *l0_slots = nullptr;
    bool fuse0 = false;
    bool l0_blocks = false;        // level 1 = parts of aligned 2 x 2 raster blocks (k_block_agg): the two-row level-0 kernels restrict in place
    const uint8_t *l0_fixed = nullptr;
    int l0_rows = 0, l0_cols = 0;
};

size_t amg_workspace_bytes(int rows, int cols);
// Builds the hierarchy inside `workspace` (device memory, 256-byte aligned).
int amg_setup(AmgHierarchy &h, const double *cond, const uint8_t *fixed, int rows, int cols,
              void *workspace, size_t workspace_bytes, hipStream_t st);
// out = M rhs (one V-cycle); rhs/out: f64 vectors on the raster numbering.  `norm2` (device, may be NULL): a value of
// the order of |rhs|^2 -- the cycle runs on rhs / sqrt(norm2) and the result is scaled back (M is linear)
// robust: the V(2,2) cycle of rounds 1-3 whatever nu0 / nuc say (BiCGStab takes it when it stagnates under V(1,1))
void amg_apply(AmgHierarchy &h, const double *rhs, double *out, const double *norm2, hipStream_t st, bool robust = false);
// Frees host-side resources (the captured graph); device memory is the caller's.
void amg_release(AmgHierarchy &h);

}  // namespace ssrs
