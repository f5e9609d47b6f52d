// K3'/K4 -- presence density for gfx950 (MI355X).
//
// Reference semantics (paths relative to /root/reference):
//   ssrs/movmodel.py:410-419   compute_presence_counts  (python loop, int16!)
//   ssrs/movmodel.py:422-439   compute_smooth_presence_counts: disk kernel
//                              (x^2+y^2 <= k^2)/ntaps, scipy convolve2d 'same'
//   ssrs/simulator.py:520-546  /max per realisation, sum, /max per case, ...
//
// The reference evaluates the (2k+1)^2-tap convolution directly: 1.3e8 MAC at
// 500x600/k=10, 1.2e12 at 5000x6000/k=100 (CPU-infeasible).  Here the disk is a
// stack of 2k+1 horizontal chords, each a difference of two row-prefix sums, so
// a cell costs 2(2k+1) cached reads and the sum is EXACT in integers; one
// multiply by 1/ntaps and the f32 rounding follow.  Counts are uint32 (the
// reference's int16 wraps above 32767 visits -- SURVEY.md section 7).
#include <cmath>
#include <vector>

#include "common.h"

namespace ssrs {

__global__ __launch_bounds__(kBlock) void k_presence_count(const int16_t *__restrict__ traj,
                                                          long long npoints,
                                                          uint32_t *__restrict__ hist, int rows,
                                                          int cols, uint32_t *bad)
{
    for (long long i = blockIdx.x * static_cast<long long>(kBlock) + threadIdx.x; i < npoints;
         i += static_cast<long long>(gridDim.x) * kBlock) {
        const uint32_t p = reinterpret_cast<const uint32_t *>(traj)[i];
        const int r = static_cast<int16_t>(p & 0xFFFF), c = static_cast<int16_t>(p >> 16);
        if (r < 0 || c < 0 || r >= rows || c >= cols) {
            atomicOr(bad, 1u);
            continue;
        }
        atomicAdd(&hist[static_cast<size_t>(r) * cols + c], 1u);
    }
}

// exclusive prefix sums of one raster row per block: P[r][0..cols]
template <typename CountT>
__global__ __launch_bounds__(kBlock) void k_row_prefix(const CountT *__restrict__ count,
                                                      unsigned long long *__restrict__ prefix,
                                                      int rows, int cols)
{
    __shared__ unsigned long long part[kBlock];
    const int r = blockIdx.x;
    const CountT *row = count + static_cast<size_t>(r) * cols;
    unsigned long long *out = prefix + static_cast<size_t>(r) * (cols + 1);
    const int chunk = (cols + kBlock - 1) / kBlock;
    const int lo = threadIdx.x * chunk;
    const int hi = lo + chunk < cols ? lo + chunk : cols;
    unsigned long long s = 0;
    for (int c = lo; c < hi; ++c) s += row[c];
    part[threadIdx.x] = s;
    __syncthreads();
    // Hillis-Steele inclusive scan over the 256 partials
    for (int off = 1; off < kBlock; off <<= 1) {
        unsigned long long v = threadIdx.x >= off ? part[threadIdx.x - off] : 0ull;
        __syncthreads();
        part[threadIdx.x] += v;
        __syncthreads();
    }
    unsigned long long run = part[threadIdx.x] - s;   // exclusive
    for (int c = lo; c < hi; ++c) {
        out[c] = run;
        run += row[c];
    }
    if (threadIdx.x == kBlock - 1) out[cols] = part[kBlock - 1];
}

__global__ __launch_bounds__(kBlock) void k_disk_sum(const unsigned long long *__restrict__ prefix,
                                                    const int *__restrict__ half, int krad,
                                                    double weight, float *__restrict__ out,
                                                    int rows, int cols)
{
    const int r = blockIdx.y;
    const int c = blockIdx.x * kBlock + threadIdx.x;
    if (c >= cols) return;
    unsigned long long acc = 0;
    const int dy0 = r - krad < 0 ? -r : -krad;
    const int dy1 = r + krad >= rows ? rows - 1 - r : krad;
    for (int dy = dy0; dy <= dy1; ++dy) {
        const int h = half[dy + krad];
        const unsigned long long *p = prefix + static_cast<size_t>(r + dy) * (cols + 1);
        const int a = c - h < 0 ? 0 : c - h;
        const int b = c + h + 1 > cols ? cols : c + h + 1;
        acc += p[b] - p[a];
    }
    out[static_cast<size_t>(r) * cols + c] = static_cast<float>(static_cast<double>(acc) * weight);
}

// max over a non-negative array into *slot (bit pattern order == value order)
template <typename T>
__global__ __launch_bounds__(kBlock) void k_max_nonneg(const T *__restrict__ x, size_t n,
                                                      unsigned long long *slot)
{
    double m = 0.0;
    for (size_t i = blockIdx.x * static_cast<size_t>(kBlock) + threadIdx.x; i < n;
         i += static_cast<size_t>(gridDim.x) * kBlock) {
        const double v = static_cast<double>(x[i]);
        m = v > m ? v : m;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const double o = __shfl_down(m, off);
        m = o > m ? o : m;
    }
    if ((threadIdx.x & 63) == 0) atomicMax(slot, static_cast<unsigned long long>(__double_as_longlong(m)));
}

// acc += src / max(src): the division happens in src's precision like numpy's
// in-place `prprob /= np.amax(prprob)` (f32) or `case_prob /= ...` (f64).
template <typename T>
__global__ __launch_bounds__(kBlock) void k_normalise_add(const T *__restrict__ src,
                                                         double *__restrict__ acc, size_t n,
                                                         const unsigned long long *slot)
{
    const T mx = static_cast<T>(__longlong_as_double(static_cast<long long>(*slot)));
    for (size_t i = blockIdx.x * static_cast<size_t>(kBlock) + threadIdx.x; i < n;
         i += static_cast<size_t>(gridDim.x) * kBlock)
        acc[i] = acc[i] + static_cast<double>(src[i] / mx);
}

__global__ __launch_bounds__(kBlock) void k_normalise_f32(const double *__restrict__ src,
                                                         float *__restrict__ out, size_t n,
                                                         const unsigned long long *slot)
{
    const double mx = __longlong_as_double(static_cast<long long>(*slot));
    for (size_t i = blockIdx.x * static_cast<size_t>(kBlock) + threadIdx.x; i < n;
         i += static_cast<size_t>(gridDim.x) * kBlock)
        out[i] = static_cast<float>(src[i] / mx);
}

static inline int grid_for(size_t n)
{
    size_t b = (n + kBlock - 1) / kBlock;
    if (b < 1) b = 1;
    if (b > static_cast<size_t>(kMaxStreamBlocks)) b = kMaxStreamBlocks;
    return static_cast<int>(b);
}

}  // namespace ssrs

using namespace ssrs;

extern "C" size_t ssrs_presence_workspace_bytes(int rows, int cols, int krad)
{
    if (rows < 0 || cols < 0 || krad < 0) return 0;
    size_t prefix = static_cast<size_t>(rows) * (static_cast<size_t>(cols) + 1) * 8;
    size_t half = (static_cast<size_t>(2 * krad + 1) * 4 + 255) / 256 * 256;
    return 256 + half + prefix;
}

extern "C" int ssrs_presence_count(const int16_t *traj, int64_t npoints, uint32_t *hist,
                                   int rows, int cols, void *scratch8, void *stream)
{
    SSRS_REQUIRE(hist && scratch8, "ssrs_presence_count: NULL pointer");
    SSRS_REQUIRE(rows > 0 && cols > 0 && npoints >= 0, "ssrs_presence_count: bad sizes");
    if (npoints == 0) return SSRS_OK;
    SSRS_REQUIRE(traj != nullptr, "ssrs_presence_count: traj is NULL");
    hipStream_t st = as_stream(stream);
    SSRS_HIP_CHECK(hipMemsetAsync(scratch8, 0, 8, st));
    hipLaunchKernelGGL(k_presence_count, dim3(grid_for(static_cast<size_t>(npoints))),
                       dim3(kBlock), 0, st, traj, static_cast<long long>(npoints), hist, rows,
                       cols, static_cast<uint32_t *>(scratch8));
    SSRS_HIP_CHECK(hipGetLastError());
    uint32_t bad = 0;
    SSRS_HIP_CHECK(hipMemcpyAsync(&bad, scratch8, 4, hipMemcpyDeviceToHost, st));
    SSRS_HIP_CHECK(hipStreamSynchronize(st));
    if (bad) return set_error(SSRS_ERR_INVALID, "ssrs_presence_count: a trajectory point lies outside the %d x %d raster (the reference raises IndexError)", rows, cols);
    return SSRS_OK;
}

template <typename CountT>
static int presence_smooth(const CountT *count, int krad, float *out, int rows, int cols,
                           void *workspace, size_t workspace_bytes, void *stream)
{
    SSRS_REQUIRE(count && out && workspace, "ssrs_presence_smooth: NULL pointer");
    SSRS_REQUIRE(rows > 0 && cols > 0 && krad >= 0, "ssrs_presence_smooth: bad sizes");
    SSRS_REQUIRE(workspace_bytes >= ssrs_presence_workspace_bytes(rows, cols, krad),
                 "ssrs_presence_smooth: workspace too small");
    SSRS_REQUIRE((reinterpret_cast<uintptr_t>(workspace) & 255u) == 0,
                 "ssrs_presence_smooth: workspace must be 256-byte aligned");
    hipStream_t st = as_stream(stream);
    // chord half-widths: x^2 + y^2 <= k^2  <=>  |x| <= floor(sqrt(k^2 - y^2))
    std::vector<int> half(2 * krad + 1);
    long long ntaps = 0;
    for (int y = -krad; y <= krad; ++y) {
        int h = static_cast<int>(std::floor(std::sqrt(static_cast<double>(krad) * krad - static_cast<double>(y) * y)));
        while (static_cast<long long>(h + 1) * (h + 1) + static_cast<long long>(y) * y <= static_cast<long long>(krad) * krad) ++h;
        while (h > 0 && static_cast<long long>(h) * h + static_cast<long long>(y) * y > static_cast<long long>(krad) * krad) --h;
        half[y + krad] = h;
        ntaps += 2 * h + 1;
    }
    char *base = static_cast<char *>(workspace);
    int *d_half = reinterpret_cast<int *>(base + 256);
    size_t half_bytes = (static_cast<size_t>(2 * krad + 1) * 4 + 255) / 256 * 256;
    auto *prefix = reinterpret_cast<unsigned long long *>(base + 256 + half_bytes);
    SSRS_HIP_CHECK(hipMemcpyAsync(d_half, half.data(), half.size() * sizeof(int),
                                  hipMemcpyHostToDevice, st));
    SSRS_HIP_CHECK(hipStreamSynchronize(st));   // `half` is a host temporary
    hipLaunchKernelGGL(k_row_prefix<CountT>, dim3(rows), dim3(kBlock), 0, st, count, prefix, rows, cols);
    SSRS_HIP_CHECK(hipGetLastError());
    const double weight = 1.0 / static_cast<double>(ntaps);   // kernel /= np.sum(kernel)
    hipLaunchKernelGGL(k_disk_sum, dim3((cols + kBlock - 1) / kBlock, rows), dim3(kBlock), 0, st,
                       prefix, d_half, krad, weight, out, rows, cols);
    SSRS_HIP_CHECK(hipGetLastError());
    return SSRS_OK;
}

extern "C" int ssrs_presence_smooth(const uint32_t *count, int krad, float *out, int rows,
                                    int cols, void *workspace, size_t workspace_bytes,
                                    void *stream)
{
    return presence_smooth<uint32_t>(count, krad, out, rows, cols, workspace, workspace_bytes, stream);
}

extern "C" int ssrs_presence_smooth_u64(const uint64_t *count, int krad, float *out, int rows,
                                        int cols, void *workspace, size_t workspace_bytes,
                                        void *stream)
{
    return presence_smooth<unsigned long long>(reinterpret_cast<const unsigned long long *>(count), krad, out,
                                               rows, cols, workspace, workspace_bytes, stream);
}

extern "C" int ssrs_presence_normalise_add(const void *src, int src_type, double *acc, size_t n,
                                           void *scratch8, void *stream)
{
    SSRS_REQUIRE(src && acc && scratch8, "ssrs_presence_normalise_add: NULL pointer");
    SSRS_REQUIRE(src_type == SSRS_F32 || src_type == SSRS_F64,
                 "ssrs_presence_normalise_add: bad element type");
    if (n == 0) return SSRS_OK;
    hipStream_t st = as_stream(stream);
    auto *slot = static_cast<unsigned long long *>(scratch8);
    SSRS_HIP_CHECK(hipMemsetAsync(slot, 0, 8, st));
    if (src_type == SSRS_F32) {
        hipLaunchKernelGGL(k_max_nonneg<float>, dim3(grid_for(n)), dim3(kBlock), 0, st,
                           static_cast<const float *>(src), n, slot);
        hipLaunchKernelGGL(k_normalise_add<float>, dim3(grid_for(n)), dim3(kBlock), 0, st,
                           static_cast<const float *>(src), acc, n, slot);
    } else {
        hipLaunchKernelGGL(k_max_nonneg<double>, dim3(grid_for(n)), dim3(kBlock), 0, st,
                           static_cast<const double *>(src), n, slot);
        hipLaunchKernelGGL(k_normalise_add<double>, dim3(grid_for(n)), dim3(kBlock), 0, st,
                           static_cast<const double *>(src), acc, n, slot);
    }
    SSRS_HIP_CHECK(hipGetLastError());
    return SSRS_OK;
}

extern "C" int ssrs_presence_normalise_f32(const double *src, float *out, size_t n,
                                           void *scratch8, void *stream)
{
    SSRS_REQUIRE(src && out && scratch8, "ssrs_presence_normalise_f32: NULL pointer");
    if (n == 0) return SSRS_OK;
    hipStream_t st = as_stream(stream);
    auto *slot = static_cast<unsigned long long *>(scratch8);
    SSRS_HIP_CHECK(hipMemsetAsync(slot, 0, 8, st));
    hipLaunchKernelGGL(k_max_nonneg<double>, dim3(grid_for(n)), dim3(kBlock), 0, st, src, n, slot);
    hipLaunchKernelGGL(k_normalise_f32, dim3(grid_for(n)), dim3(kBlock), 0, st, src, out, n, slot);
    SSRS_HIP_CHECK(hipGetLastError());
    return SSRS_OK;
}
