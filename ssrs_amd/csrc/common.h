// Internal helpers shared by the translation units of libssrs_hip.so.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>

#include "../../include/ssrs_hip.h"

namespace ssrs {

// thread-local message behind ssrs_last_error()
char *error_buffer();
int set_error(int code, const char *fmt, ...);

#define SSRS_HIP_CHECK(expr)                                                      \
    do {                                                                          \
        hipError_t e_ = (expr);                                                   \
        if (e_ != hipSuccess)                                                     \
            return ssrs::set_error(SSRS_ERR_HIP, "%s failed: %s (%s:%d)", #expr,  \
                                   hipGetErrorString(e_), __FILE__, __LINE__);    \
    } while (0)

#define SSRS_REQUIRE(cond, ...)                                     \
    do {                                                            \
        if (!(cond)) return ssrs::set_error(SSRS_ERR_INVALID, __VA_ARGS__); \
    } while (0)

inline hipStream_t as_stream(void *s) { return reinterpret_cast<hipStream_t>(s); }

// MI355X: 256 CUs; memory-bound grids are capped at 8 blocks of 256 per CU and
// grid-stride the rest (cdna_hip_programming.md, Guideline 11).
constexpr int kBlock = 256;
constexpr int kMaxStreamBlocks = 256 * 8;

// Thread-per-cell stencil kernels walk the raster in row segments of kBlock columns
// (row and column without a division per cell).  Tiles of several rows per block, so
// that a block re-reads its own rows from L1 / L2, were measured and are NOT used:
// at 5000 x 6000 the operator application took 1.04 ms with 1, 2, 4 or 8 rows per tile
// and 1.2 ms with 32 (serial rows cost more latency than the re-fetch saves).
struct TileWalk {
    int rows, cols, tile_rows, tiles_x;
    long long ntiles;
};
inline TileWalk make_tile_walk(int rows, int cols)
{
    TileWalk w{rows, cols, 1, (cols + kBlock - 1) / kBlock, 0};
    w.ntiles = static_cast<long long>(rows) * w.tiles_x;
    return w;
}
// for_each_cell(w, [&](size_t i, int r, int c) {...}) over the tiles blockIdx.x, + gridDim.x, ...
template <class F>
__device__ __forceinline__ void for_each_cell(const TileWalk &w, F &&body)
{
    for (long long t = blockIdx.x; t < w.ntiles; t += gridDim.x) {
        const int ty = static_cast<int>(t / w.tiles_x), tx = static_cast<int>(t - static_cast<long long>(ty) * w.tiles_x);
        const int c = tx * kBlock + static_cast<int>(threadIdx.x);
        if (c >= w.cols) continue;
        const int r0 = ty * w.tile_rows;
        const int r1 = r0 + w.tile_rows < w.rows ? r0 + w.tile_rows : w.rows;
        size_t i = static_cast<size_t>(r0) * w.cols + c;
        for (int r = r0; r < r1; ++r, i += w.cols) body(i, r, c);
    }
}

}  // namespace ssrs
