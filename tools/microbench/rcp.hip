// accuracy of the raw v_rcp_f64 / v_rcp_f32 estimates on gfx950 (max relative error over random inputs)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
#include <random>
__global__ void k(const double *x, double *r64, float *r32, int n)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    r64[i] = __builtin_amdgcn_rcp(x[i]);
    r32[i] = __builtin_amdgcn_rcpf(static_cast<float>(x[i]));
}
int main()
{
    const int n = 1 << 20;
    std::vector<double> h(n), o64(n);
    std::vector<float> o32(n);
    std::mt19937_64 g(1);
    std::uniform_real_distribution<double> u(-30.0, 30.0);
    for (auto &v : h) v = std::pow(10.0, u(g)) * (1.0 + 0.37 * u(g) / 30.0);
    double *dx, *d64; float *d32;
    hipMalloc(&dx, n * 8); hipMalloc(&d64, n * 8); hipMalloc(&d32, n * 4);
    hipMemcpy(dx, h.data(), n * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, d64, d32, n);
    hipMemcpy(o64.data(), d64, n * 8, hipMemcpyDeviceToHost);
    hipMemcpy(o32.data(), d32, n * 4, hipMemcpyDeviceToHost);
    double e64 = 0, e32 = 0;
    for (int i = 0; i < n; ++i) {
        e64 = std::fmax(e64, std::fabs(o64[i] * h[i] - 1.0));
        const double xf = static_cast<double>(static_cast<float>(h[i]));
        e32 = std::fmax(e32, std::fabs(static_cast<double>(o32[i]) * xf - 1.0));
    }
    printf("v_rcp_f64: max relative error %.3e   v_rcp_f32: %.3e\n", e64, e32);
    return 0;
}
