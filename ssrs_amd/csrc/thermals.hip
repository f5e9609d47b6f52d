// a5 -- random thermal updraft field for gfx950 (MI355X).
//
// Reference semantics (paths relative to /root/reference):
//   ssrs/layers.py:188-214  compute_thermals: inside a 10 % border, each cell
//       draws num1 = randint(1, int(wt)), wt = 1000 + |aspect-180|/180*2000, and
//       seeds a thermal lognormal(scale + 3, 0.5) when num1 == 5; the seed field
//       is blurred with scipy.ndimage.gaussian_filter(sigma=4, mode='constant').
//   ssrs/simulator.py:217-228 one field per realisation, saved as f32.
//
// The reference consumes the serial global MT19937 in a python double loop
// (1-2 draws per cell in row-major order), which no parallel code can replay;
// only STATISTICAL parity is possible.  Here every
// cell owns a Philox4x32-10 block keyed by (seed, cell index): word 0 decides
// the seeding with the reference's probability 1/(int(wt)-1), words 1-3 feed a
// Box-Muller normal for the lognormal amplitude.  The blur is the same
// separable, zero-padded, 4-sigma-truncated Gaussian as scipy's.
#include <rocrand/rocrand_philox4x32_10.h>

#include <cmath>
#include <vector>

#include "common.h"

namespace ssrs {

__global__ __launch_bounds__(kBlock) void k_thermal_seeds(const double *__restrict__ aspect,
                                                         double mu, double sigma,
                                                         unsigned long long seed,
                                                         double *__restrict__ out, int rows,
                                                         int cols)
{
    const size_t n = static_cast<size_t>(rows) * cols;
    const int by = static_cast<int>(0.1 * rows), bx = static_cast<int>(0.1 * cols);
    for (size_t i = blockIdx.x * static_cast<size_t>(kBlock) + threadIdx.x; i < n;
         i += static_cast<size_t>(gridDim.x) * kBlock) {
        const int r = static_cast<int>(i / cols), c = static_cast<int>(i % cols);
        double v = 0.0;
        if (r >= by && r < rows - by && c >= bx && c < cols - bx) {
            rocrand_state_philox4x32_10 st;
            rocrand_init(seed, i, 0, &st);
            const uint4 w = rocrand4(&st);
            const double wt = 1000.0 + (fabs(aspect[i] - 180.0) / 180.0) * 2000.0;
            const int nvals = static_cast<int>(wt) - 1;         // randint(1, int(wt)): nvals values
            // P(num1 == 5) = 1 / nvals (nvals >= 5 always: wt >= 1000)
            const double u0 = static_cast<double>(w.x) * (1.0 / 4294967296.0);
            if (u0 * nvals < 1.0) {
                const double u1 = (static_cast<double>(w.y) + 1.0) * (1.0 / 4294967296.0);  // (0,1]
                const double u2 = static_cast<double>(w.z) * (1.0 / 4294967296.0);
                const double z = sqrt(-2.0 * log(u1)) * cos(2.0 * 3.141592653589793 * u2);
                v = exp(mu + sigma * z);
            }
        }
        out[i] = v;
    }
}

// one separable pass: out[r][c] = sum_k w[k] in[.. + k ..] along `axis`, zero padded
__global__ __launch_bounds__(kBlock) void k_blur_pass(const double *__restrict__ in,
                                                     double *__restrict__ out,
                                                     const double *__restrict__ weights, int radius,
                                                     int rows, int cols, int axis)
{
    const size_t n = static_cast<size_t>(rows) * cols;
    for (size_t i = blockIdx.x * static_cast<size_t>(kBlock) + threadIdx.x; i < n;
         i += static_cast<size_t>(gridDim.x) * kBlock) {
        const int r = static_cast<int>(i / cols), c = static_cast<int>(i % cols);
        double acc = 0.0;
        for (int k = -radius; k <= radius; ++k) {
            const int rr = axis == 0 ? r + k : r, cc = axis == 0 ? c : c + k;
            if (rr < 0 || rr >= rows || cc < 0 || cc >= cols) continue;
            acc += weights[k + radius] * in[static_cast<size_t>(rr) * cols + cc];
        }
        out[i] = acc;
    }
}

}  // namespace ssrs

using namespace ssrs;

static int blocks_for(size_t n)
{
    size_t b = (n + kBlock - 1) / kBlock;
    return static_cast<int>(b < 1 ? 1 : (b > static_cast<size_t>(kMaxStreamBlocks) ? kMaxStreamBlocks : b));
}

extern "C" size_t ssrs_blur_workspace_bytes(int rows, int cols, double sigma)
{
    if (rows <= 0 || cols <= 0 || !(sigma > 0.0)) return 0;
    const int radius = static_cast<int>(4.0 * sigma + 0.5);
    return static_cast<size_t>(rows) * cols * 8 + (static_cast<size_t>(2 * radius + 1) * 8 + 255) / 256 * 256 + 256;
}

extern "C" int ssrs_gaussian_blur(const double *in, double *out, double sigma, int rows, int cols,
                                  void *workspace, size_t workspace_bytes, void *stream)
{
    SSRS_REQUIRE(in && out && workspace, "ssrs_gaussian_blur: NULL pointer");
    SSRS_REQUIRE(rows > 0 && cols > 0 && sigma > 0.0, "ssrs_gaussian_blur: bad arguments");
    SSRS_REQUIRE(workspace_bytes >= ssrs_blur_workspace_bytes(rows, cols, sigma),
                 "ssrs_gaussian_blur: workspace too small");
    // scipy.ndimage._gaussian_kernel1d: radius = int(truncate * sigma + 0.5), truncate = 4
    const int radius = static_cast<int>(4.0 * sigma + 0.5);
    std::vector<double> w(2 * radius + 1);
    double sum = 0.0;
    for (int k = -radius; k <= radius; ++k) { w[k + radius] = std::exp(-0.5 / (sigma * sigma) * k * k); sum += w[k + radius]; }
    for (double &v : w) v /= sum;
    hipStream_t st = as_stream(stream);
    char *base = static_cast<char *>(workspace);
    double *d_w = reinterpret_cast<double *>(base);
    double *tmp = reinterpret_cast<double *>(base + (w.size() * 8 + 255) / 256 * 256);
    SSRS_HIP_CHECK(hipMemcpyAsync(d_w, w.data(), w.size() * 8, hipMemcpyHostToDevice, st));
    SSRS_HIP_CHECK(hipStreamSynchronize(st));
    const size_t n = static_cast<size_t>(rows) * cols;
    // scipy filters axis 0 first, then axis 1
    hipLaunchKernelGGL(k_blur_pass, dim3(blocks_for(n)), dim3(kBlock), 0, st, in, tmp, d_w, radius, rows, cols, 0);
    hipLaunchKernelGGL(k_blur_pass, dim3(blocks_for(n)), dim3(kBlock), 0, st, tmp, out, d_w, radius, rows, cols, 1);
    SSRS_HIP_CHECK(hipGetLastError());
    return SSRS_OK;
}

extern "C" int ssrs_thermal_seeds(const double *aspect, double thermal_intensity_scale,
                                  uint64_t seed, double *seeds, int rows, int cols, void *stream)
{
    SSRS_REQUIRE(aspect && seeds, "ssrs_thermal_seeds: NULL pointer");
    SSRS_REQUIRE(rows > 0 && cols > 0, "ssrs_thermal_seeds: bad sizes");
    const size_t n = static_cast<size_t>(rows) * cols;
    hipLaunchKernelGGL(k_thermal_seeds, dim3(blocks_for(n)), dim3(kBlock), 0, as_stream(stream),
                       aspect, thermal_intensity_scale + 3.0, 0.5,
                       static_cast<unsigned long long>(seed), seeds, rows, cols);
    SSRS_HIP_CHECK(hipGetLastError());
    return SSRS_OK;
}
