// Throughput of one wave-level 4-byte gather on gfx950 as a function of how many distinct 128-byte
// lines its 64 lanes touch (all L2 / L1 resident): what the CU's address unit charges a stepper
// whose lanes read neighbouring table entries.  Build: hipcc --offload-arch=gfx950 -O3 gather.hip -o gather
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

template <int LINES>
__global__ void gather(const uint32_t *mem, uint32_t *out, int iters, uint32_t ring_mask)
{
    // lane l reads dword (base + (l % LINES) * 32 + l / LINES) of a window that moves every iteration
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t pos = (blockIdx.x * 977u + (threadIdx.x >> 6) * 131u) * 4096u;
    const uint32_t lane_off = (lane % LINES) * 32u + (lane / LINES) % 32u;
    uint32_t acc = 0;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            acc += mem[(pos + lane_off) & ring_mask];
            pos += 6000u;             // next "row"
        }
    }
    if (acc == 0x12345u) out[0] = acc;
}

template <int LINES>
void run(const uint32_t *d_mem, uint32_t *d_out, int blocks, uint32_t mask)
{
    const int iters = 400;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(gather<LINES>, dim3(blocks), dim3(256), 0, 0, d_mem, d_out, iters, mask);
    (void)hipEventRecord(e0, 0);
    hipLaunchKernelGGL(gather<LINES>, dim3(blocks), dim3(256), 0, 0, d_mem, d_out, iters, mask);
    (void)hipEventRecord(e1, 0);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double wave_instr = static_cast<double>(blocks) * 4 * iters * 8;
    // cycles of CU time per wave-instruction: ms * 2.4e6 cycles/ms * 256 CUs / wave_instr
    printf("%2d lines per gather, %5d blocks: %8.3f ms, %6.1f CU-clocks per wave-level gather (2.4 GHz, 256 CUs)\n", LINES, blocks, ms,
           ms * 2.4e6 * 256.0 / wave_instr);
}

int main()
{
    const uint32_t n = 1u << 22;            // 16 MB of dwords: L2-resident across the XCDs
    uint32_t *d_mem, *d_out;
    (void)hipMalloc(&d_mem, n * 4);
    (void)hipMalloc(&d_out, 256);
    (void)hipMemset(d_mem, 1, n * 4);
    for (int blocks : {512, 2048, 8192}) {
        run<1>(d_mem, d_out, blocks, n - 1);
        run<2>(d_mem, d_out, blocks, n - 1);
        run<4>(d_mem, d_out, blocks, n - 1);
        run<8>(d_mem, d_out, blocks, n - 1);
        run<16>(d_mem, d_out, blocks, n - 1);
        run<32>(d_mem, d_out, blocks, n - 1);
        run<64>(d_mem, d_out, blocks, n - 1);
    }
    return 0;
}
