// What the CU's memory path charges a latency-bound stepper per dependent, fully divergent gather:
// 256 blocks (one per CU: each takes 96 KB of LDS so that no two share a CU), W waves per block, L live lanes per
// wave, every live lane chases its own pointer through an L2-resident ring of 16-byte entries (dwordx4 loads, the
// pair table's access) or 4-byte entries (dword).  Per (W, L): clocks per dependent load as one wave sees it, and
// lane-loads per clock and CU (the throughput the CU's TA/TCP path delivers).
// Build: hipcc --offload-arch=gfx950 -O3 gather_latency.hip -o gather_latency
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

template <int WIDE>
__global__ void chase(uint64_t *out, const uint32_t *mem, int iters, int live)
{
    extern __shared__ uint32_t pad[];
    const int lane = threadIdx.x & 63;
    uint32_t idx = (blockIdx.x * 1031u + threadIdx.x * 97u) & 0xFFFFu;
    if (threadIdx.x == 0) pad[0] = 0;
    __syncthreads();
    const uint64_t c0 = clock64();
    if (lane < live) {
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (WIDE) {
                    const uint4 e = reinterpret_cast<const uint4 *>(mem)[idx];
                    idx = e.x ^ (e.y & e.z & e.w & 0u);          // (all four words are consumed)
                } else {
                    idx = mem[idx];
                }
            }
        }
    }
    const uint64_t c1 = clock64();
    if (lane == 0) out[(blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6)) * 2] = c1 - c0;
    if (idx == 0xFFFFFFFFu) out[1] = pad[0];
}

int main()
{
    const int ring = 1 << 16;                       // 65 536 entries: 1 MB of uint4, 256 KB of uint32 -- L2-resident
    std::vector<uint32_t> h4(ring * 4), h1(ring);
    // a permutation with one long cycle whose successive elements are far apart (every lane on its own cache line)
    for (int i = 0; i < ring; ++i) {
        const uint32_t nxt = (i * 40503u + 12345u) & (ring - 1);
        h4[4 * i] = nxt; h4[4 * i + 1] = h4[4 * i + 2] = h4[4 * i + 3] = 0xFFFFFFFFu;
        h1[i] = nxt;
    }
    uint32_t *d4, *d1;
    uint64_t *d_out;
    hipMalloc(&d4, h4.size() * 4); hipMalloc(&d1, h1.size() * 4); hipMalloc(&d_out, 1 << 20);
    hipMemcpy(d4, h4.data(), h4.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(d1, h1.data(), h1.size() * 4, hipMemcpyHostToDevice);
    const int iters = 400, blocks = 256;
    printf("%-8s %5s %5s %14s %18s\n", "load", "waves", "lanes", "clocks/load", "lane-loads/clk/CU");
    for (int wide = 1; wide >= 0; --wide)
        for (int W : {1, 2, 3, 4, 6, 8, 12, 16})
            for (int L : {16, 32, 47, 64}) {
                for (int rep = 0; rep < 2; ++rep) {
                    if (wide) hipLaunchKernelGGL(chase<1>, dim3(blocks), dim3(64 * W), 96 * 1024, 0, d_out, d4, iters, L);
                    else hipLaunchKernelGGL(chase<0>, dim3(blocks), dim3(64 * W), 96 * 1024, 0, d_out, d1, iters, L);
                }
                hipDeviceSynchronize();
                std::vector<uint64_t> o(blocks * W * 2);
                hipMemcpy(o.data(), d_out, o.size() * 8, hipMemcpyDeviceToHost);
                double sum = 0;
                for (int i = 0; i < blocks * W; ++i) sum += static_cast<double>(o[2 * i]);
                const double per_load = sum / (blocks * W) / (iters * 8.0);
                printf("%-8s %5d %5d %14.1f %18.3f\n", wide ? "dwordx4" : "dword", W, L, per_load, W * L / per_load);
            }
    return 0;
}
