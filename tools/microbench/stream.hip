// What HBM gives plain streaming kernels on gfx950: read-only, write-only, copy, and the
// table builder's mix (12 B read, 8 x 4 B written to eight planes per element), 16-byte and 4-byte
// accesses per lane.  The write-heavy kernels of this repo (K1, K2a) are priced against these, and
// v_cvt_pknorm_u16_f32's rounding is checked against round(65535 x) on the way.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
template <typename T> __global__ void k_read(const T *a, T *sink, size_t n)
{
    T acc{};
    for (size_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += size_t(gridDim.x) * blockDim.x) {
        const T v = a[i];
        if constexpr (sizeof(T) == 16) { acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w; } else acc ^= v;
    }
    if constexpr (sizeof(T) == 16) { if (acc.x == 0x12345u) sink[0] = acc; } else if (acc == 0x12345u) sink[0] = acc;
}
template <typename T> __global__ void k_write(T *a, size_t n, T v)
{
    for (size_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += size_t(gridDim.x) * blockDim.x) a[i] = v;
}
template <typename T> __global__ void k_copy(const T *a, T *b, size_t n)
{
    for (size_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += size_t(gridDim.x) * blockDim.x) b[i] = a[i];
}
__global__ void k_mix(const double *u, const float *p, uint32_t *t, size_t n, size_t plane)
{   // one element per lane per iteration: 8 + 4 B read, eight 4-byte stores a plane apart
    for (size_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += size_t(gridDim.x) * blockDim.x) {
        const uint32_t e = static_cast<uint32_t>(u[i]) + __float_as_uint(p[i]);
#pragma unroll
        for (int rc = 0; rc < 8; ++rc) t[rc * plane + i] = e + rc;
    }
}
__global__ void k_mix_raster(const double *z, float *o, double *u, size_t n)
{   // the raster kernel's traffic: 8 B read, 4 + 8 B written per cell
    for (size_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += size_t(gridDim.x) * blockDim.x) {
        const double v = z[i];
        o[i] = static_cast<float>(v); u[i] = v + 1.0;
    }
}
__global__ void k_pknorm(const float *x, uint32_t *o, int n)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) o[i] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pknorm_u16(x[i], x[i] * 1.0000153f));
}
template <typename F> static float timed(F f, int reps = 10)
{
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    f(); hipDeviceSynchronize();
    hipEventRecord(a); for (int i = 0; i < reps; ++i) f(); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); return ms / reps;
}
int main()
{
    const size_t bytes = size_t(1) << 30, cells = 30000000;
    char *a, *b; CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes + 4096));
    CK(hipMemset(a, 1, bytes)); CK(hipMemset(b, 0, bytes));
    for (int grid : {2048, 8192, 65536}) {
        float r16 = timed([&] { hipLaunchKernelGGL(k_read<uint4>, dim3(grid), dim3(256), 0, 0, (const uint4 *)a, (uint4 *)b, bytes / 16); });
        float r4 = timed([&] { hipLaunchKernelGGL(k_read<uint32_t>, dim3(grid), dim3(256), 0, 0, (const uint32_t *)a, (uint32_t *)b, bytes / 4); });
        float w16 = timed([&] { hipLaunchKernelGGL(k_write<uint4>, dim3(grid), dim3(256), 0, 0, (uint4 *)b, bytes / 16, make_uint4(1, 2, 3, 4)); });
        float w4 = timed([&] { hipLaunchKernelGGL(k_write<uint32_t>, dim3(grid), dim3(256), 0, 0, (uint32_t *)b, bytes / 4, 7u); });
        float c16 = timed([&] { hipLaunchKernelGGL(k_copy<uint4>, dim3(grid), dim3(256), 0, 0, (const uint4 *)a, (uint4 *)b, bytes / 16); });
        float mix = timed([&] { hipLaunchKernelGGL(k_mix, dim3(grid), dim3(256), 0, 0, (const double *)a, (const float *)(a + cells * 8), (uint32_t *)b, cells, size_t(1) << 25); });
        float mr = timed([&] { hipLaunchKernelGGL(k_mix_raster, dim3(grid), dim3(256), 0, 0, (const double *)a, (float *)b, (double *)(b + cells * 4), cells); });
        printf("grid %6d: raster mix %.2f TB/s (20 B/cell, %.0f us)\n", grid, cells * 20.0 / mr / 1e9, mr * 1e3);
        printf("grid %6d: read x4 %.2f  read dword %.2f  write x4 %.2f  write dword %.2f  copy x4 %.2f (r+w)  table mix %.2f TB/s (44 B/cell, %.0f us)\n", grid,
               bytes / r16 / 1e9, bytes / r4 / 1e9, bytes / w16 / 1e9, bytes / w4 / 1e9, 2.0 * bytes / c16 / 1e9, cells * 44.0 / mix / 1e9, mix * 1e3);
    }
    // v_cvt_pknorm_u16_f32 against round-half-even(65535 x), saturating
    const int n = 1 << 22;
    std::vector<float> h(n); std::vector<uint32_t> o(n);
    for (int i = 0; i < n; ++i) h[i] = (i < n / 2) ? (i + 0.5f * (i & 1)) / 65535.0f * (65535.0f / (n / 2)) : static_cast<float>(i - n / 2) / (n / 2) * 1.001f;
    float *dx; uint32_t *d_o; CK(hipMalloc(&dx, n * 4)); CK(hipMalloc(&d_o, n * 4));
    CK(hipMemcpy(dx, h.data(), n * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_pknorm, dim3(n / 256), dim3(256), 0, 0, dx, d_o, n);
    CK(hipMemcpy(o.data(), d_o, n * 4, hipMemcpyDeviceToHost));
    double worst = 0, worst2 = 0;
    for (int i = 0; i < n; ++i) {
        const double x = h[i], t = std::fmin(x * 65535.0, 65535.0);
        worst = std::fmax(worst, std::fabs((o[i] & 0xFFFFu) - t));
        const double x2 = static_cast<double>(h[i] * 1.0000153f), t2 = std::fmin(static_cast<double>(h[i]) * 65536.0, 65535.0);
        (void)x2;
        worst2 = std::fmax(worst2, std::fabs((o[i] >> 16) - t2));
    }
    printf("v_cvt_pknorm_u16_f32: max |T - min(65535 x, 65535)| = %.4f;  with x * 65536/65535: max |T - min(65536 x, 65535)| = %.4f\n", worst, worst2);
    return 0;
}
