// Internal interface of the aggregation-AMG preconditioner (amg.hip) used by
// the potential solver (potential.hip).
#pragma once
#include <vector>

#include "common.h"

namespace ssrs {

struct AmgLevel {
    int n = 0, nnz = 0, nc = 0;
    int *rowptr = nullptr, *col = nullptr;
    double *val = nullptr, *dinv = nullptr;
    float *val32 = nullptr;    // the same entries rounded to f32 for the cycle's sweeps (levels >= 1)
    int *agg = nullptr;        // fine node -> coarse node (-1: isolated row), NULL on the last level
    int *memptr = nullptr;     // coarse node I -> its fine nodes memidx[memptr[I] .. memptr[I+1])
    int *memidx = nullptr;
    double *x = nullptr, *xt = nullptr, *b = nullptr, *r = nullptr;
    // K-cycle scratch (levels 1 .. kdepth)
    double *kb = nullptr, *c1 = nullptr, *v1 = nullptr, *v2 = nullptr;
    void *kscal = nullptr;
};

struct AmgHierarchy {
    std::vector<AmgLevel> levels;
    double *dense_inv = nullptr;   // inverse of the last level when it is small
    size_t workspace_used = 0;
    void *graph_exec = nullptr;    // hipGraphExec_t of one captured cycle (NULL: direct launches)
    bool graph_tried = false;
    bool symmetric = true;         // symmetric strength of connection (amg.hip: strong_link)
    int strong_rounds = 4;         // matching rounds restricted to strong couplings (of 8)
    int kdepth = 0;                // coarse levels 1..kdepth use the K-cycle (0 = V-cycle)
    int sweeps = 1;                // pairs of Jacobi sweeps before and after the coarse correction
    double om[2] = {0.7, 0.7};     // step sizes of a pair of sweeps (SSRS_AMG_OMEGAS=a,b for experiments)
    // level 0 applied matrix-free (amg.hip: L0Stencil)
    const double *l0_rinv = nullptr;
    const uint8_t *l0_fixed = nullptr;
    int l0_rows = 0, l0_cols = 0;
};

size_t amg_workspace_bytes(int rows, int cols);
// Builds the hierarchy inside `workspace` (device memory, 256-byte aligned).
int amg_setup(AmgHierarchy &h, const double *cond, const uint8_t *fixed, int rows, int cols,
              void *workspace, size_t workspace_bytes, hipStream_t st);
// out = M rhs (one V(2,2) cycle); rhs/out: vectors on the raster numbering
void amg_apply(AmgHierarchy &h, const double *rhs, double *out, hipStream_t st);
// Frees host-side resources (the captured graph); device memory is the caller's.
void amg_release(AmgHierarchy &h);

}  // namespace ssrs
