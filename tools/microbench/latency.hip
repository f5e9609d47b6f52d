// Lone-wave instruction latencies on gfx950 (what a latency-bound kernel pays per dependent
// instruction): one wave per CU, chains of dependent instructions, timed with s_memtime
// (clock64) and the 100 MHz wall clock.  Build: hipcc --offload-arch=gfx950 -O3 latency.hip -o latency
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define REP16(x) x x x x x x x x x x x x x x x x

template <int KIND>
__global__ void chain(uint64_t *out, const uint32_t *mem, int iters, float seed)
{
    float f = seed + threadIdx.x;
    uint32_t u = threadIdx.x * 2654435761u + 1u;
    uint64_t w = u;
    uint32_t idx = threadIdx.x;
    const uint64_t c0 = clock64();
    const uint64_t w0 = wall_clock64();
    for (int i = 0; i < iters; ++i) {
        if (KIND == 0) { REP16(asm volatile("v_add_f32 %0, %0, %1" : "+v"(f) : "v"(seed));) }
        if (KIND == 1) { REP16(asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(u) : "v"(idx | 1u));) }
        if (KIND == 2) { REP16(asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(w) : "v"(u), "v"(static_cast<uint32_t>(w)) : "vcc"); u = static_cast<uint32_t>(w >> 32);) }
        if (KIND == 3) { REP16(asm volatile("v_cmp_lt_f32 vcc, %1, %0\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(f) : "v"(seed) : "vcc");) }
        if (KIND == 4) { REP16(asm volatile("v_add_f64 %0, %0, %1" : "+v"(*reinterpret_cast<double *>(&w)) : "v"(1.0));) }
        if (KIND == 5) {  // dependent L2-resident loads (pointer chase over a 1 MB ring)
            REP16(idx = mem[idx];)
        }
        if (KIND == 6) {  // v_cmp -> s_cbranch (never taken) per element: cost of a wave-level test
            REP16(asm volatile("v_cmp_gt_f32 vcc, 0, %0\n s_cbranch_vccnz 1f\n v_add_f32 %0, %0, %1\n 1:" : "+v"(f) : "v"(seed) : "vcc");)
        }
        if (KIND == 7) {  // readfirstlane + salu + valu round trip
            REP16(asm volatile("v_readfirstlane_b32 s20, %0\n s_add_u32 s20, s20, 1\n v_add_u32 %0, %0, s20" : "+v"(u) : : "s20");)
        }
        if (KIND == 8) {  // independent VALU (4 chains): issue rate of one wave
            float g = f + 1.f, h = f + 2.f, k = f + 3.f;
            REP16(asm volatile("v_add_f32 %0, %0, %4\n v_add_f32 %1, %1, %4\n v_add_f32 %2, %2, %4\n v_add_f32 %3, %3, %4" : "+v"(f), "+v"(g), "+v"(h), "+v"(k) : "v"(seed));)
            f += g + h + k;
        }
        if (KIND == 9) {  // taken branch
            REP16(asm volatile("s_branch 1f\n s_nop 0\n 1: v_add_f32 %0, %0, %1" : "+v"(f) : "v"(seed));)
        }
        if (KIND == 10) { // s_and_saveexec / s_or exec pair around one VALU
            REP16(asm volatile("v_cmp_lt_f32 vcc, 0, %0\n s_and_saveexec_b64 s[20:21], vcc\n v_add_f32 %0, %0, %1\n s_or_b64 exec, exec, s[20:21]" : "+v"(f) : "v"(seed) : "vcc", "s20", "s21");)
        }
        if (KIND >= 12 && KIND <= 17) {  // four independent chains of one instruction: issue rate of one wave per opcode
            uint32_t g = u + 1u, h = u + 2u, k = u + 3u;
            uint64_t w1 = w + 1, w2 = w + 2, w3 = w + 3;
            if (KIND == 12) { REP16(asm volatile("v_xor_b32 %0, %0, %4\n v_xor_b32 %1, %1, %4\n v_xor_b32 %2, %2, %4\n v_xor_b32 %3, %3, %4" : "+v"(u), "+v"(g), "+v"(h), "+v"(k) : "v"(idx));) }
            if (KIND == 13) { REP16(asm volatile("v_bitop3_b32 %0, %0, %4, s20 bitop3:0x96\n v_bitop3_b32 %1, %1, %4, s20 bitop3:0x96\n v_bitop3_b32 %2, %2, %4, s20 bitop3:0x96\n v_bitop3_b32 %3, %3, %4, s20 bitop3:0x96" : "+v"(u), "+v"(g), "+v"(h), "+v"(k) : "v"(idx) : "s20");) }
            if (KIND == 14) { REP16(asm volatile("v_mad_u64_u32 %0, vcc, %4, %5, 0\n v_mad_u64_u32 %1, vcc, %4, %6, 0\n v_mad_u64_u32 %2, vcc, %4, %7, 0\n v_mad_u64_u32 %3, vcc, %4, %8, 0" : "+v"(w), "+v"(w1), "+v"(w2), "+v"(w3) : "v"(idx | 1u), "v"(u), "v"(g), "v"(h), "v"(k) : "vcc"); u ^= static_cast<uint32_t>(w >> 32); g ^= static_cast<uint32_t>(w1 >> 32); h ^= static_cast<uint32_t>(w2 >> 32); k ^= static_cast<uint32_t>(w3 >> 32);) }
            if (KIND == 15) { REP16(asm volatile("v_add3_u32 %0, %0, %4, -1\n v_add3_u32 %1, %1, %4, -1\n v_add3_u32 %2, %2, %4, -1\n v_add3_u32 %3, %3, %4, -1" : "+v"(u), "+v"(g), "+v"(h), "+v"(k) : "v"(idx));) }
            if (KIND == 16) { REP16(asm volatile("v_lshrrev_b64 %0, 1, %0\n v_lshrrev_b64 %1, 1, %1\n v_lshrrev_b64 %2, 1, %2\n v_lshrrev_b64 %3, 1, %3" : "+v"(w), "+v"(w1), "+v"(w2), "+v"(w3));) }
            if (KIND == 17) { REP16(asm volatile("v_bfe_u32 %0, %0, %4, 2\n v_cndmask_b32 %1, %1, %4, vcc\n v_sub_u32_sdwa %2, %2, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:WORD_0\n v_lshl_or_b32 %3, %3, 7, %4" : "+v"(u), "+v"(g), "+v"(h), "+v"(k) : "v"(idx) : "vcc");) }
            u += g + h + k;
            w += w1 + w2 + w3;
        }
        if (KIND == 11) { // global store + dependent-free VALU (does a store stall issue?)
            REP16(asm volatile("global_store_dword %1, %0, off\n v_add_f32 %0, %0, %2" : "+v"(f) : "v"(out + 4096 + threadIdx.x), "v"(seed) : "memory");)
        }
    }
    const uint64_t c1 = clock64();
    const uint64_t w1 = wall_clock64();
    if (threadIdx.x == 0) { out[blockIdx.x * 4 + 0] = c1 - c0; out[blockIdx.x * 4 + 1] = w1 - w0; }
    if (f == 12345.f || u == 77u || w == 99u || idx == 0xFFFFFFFFu) out[3] = 1;
}

template <int KIND>
void run(const char *name, int per_rep, uint64_t *d_out, const uint32_t *d_mem, int blocks, int threads)
{
    const int iters = 2000;
    hipLaunchKernelGGL(chain<KIND>, dim3(blocks), dim3(threads), 0, 0, d_out, d_mem, iters, 1.0f);
    hipLaunchKernelGGL(chain<KIND>, dim3(blocks), dim3(threads), 0, 0, d_out, d_mem, iters, 1.0f);
    hipDeviceSynchronize();
    uint64_t h[4];
    hipMemcpy(h, d_out, sizeof(h), hipMemcpyDeviceToHost);
    const double n = static_cast<double>(iters) * 16 * per_rep;
    const double ns = h[1] * 10.0;        // wall_clock64: 100 MHz
    printf("%-46s %3d blocks x %4d thr: %7.2f memtime ticks / instr, %7.2f ns / instr (%.0f MHz memtime)\n", name, blocks,
           threads, h[0] / n, ns / n, h[0] / (ns * 1e-3));
}

int main()
{
    uint64_t *d_out;
    uint32_t *d_mem;
    hipMalloc(&d_out, 1 << 20);
    const int ring = 1 << 18;                // 1 MB of uint32: L2-resident
    std::vector<uint32_t> h(ring);
    for (int i = 0; i < ring; ++i) h[i] = (i * 9973u + 12345u) % ring;
    hipMalloc(&d_mem, ring * 4);
    hipMemcpy(d_mem, h.data(), ring * 4, hipMemcpyHostToDevice);
    for (int cfg = 0; cfg < 3; ++cfg) {
        const int blocks = cfg == 0 ? 256 : (cfg == 1 ? 256 : 2048), threads = cfg == 0 ? 64 : 256;
        printf("--- %d blocks of %d threads (%s)\n", blocks, threads, cfg == 0 ? "one wave per CU" : (cfg == 1 ? "one wave per SIMD" : "8 waves per SIMD"));
        run<0>("dependent v_add_f32", 1, d_out, d_mem, blocks, threads);
        run<8>("4 independent v_add_f32 chains", 4, d_out, d_mem, blocks, threads);
        run<1>("dependent v_mul_lo_u32", 1, d_out, d_mem, blocks, threads);
        run<2>("dependent v_mad_u64_u32", 1, d_out, d_mem, blocks, threads);
        run<3>("dependent v_cmp + v_cndmask (per pair)", 1, d_out, d_mem, blocks, threads);
        run<4>("dependent v_add_f64", 1, d_out, d_mem, blocks, threads);
        run<5>("dependent global_load_dword (L2 hit)", 1, d_out, d_mem, blocks, threads);
        run<6>("v_cmp + s_cbranch_vccnz (not taken) + v_add", 1, d_out, d_mem, blocks, threads);
        run<7>("v_readfirstlane + s_add + v_add", 1, d_out, d_mem, blocks, threads);
        run<9>("s_branch (taken) + v_add", 1, d_out, d_mem, blocks, threads);
        run<10>("v_cmp + s_and_saveexec + v_add + s_or exec", 1, d_out, d_mem, blocks, threads);
        run<11>("global_store_dword + v_add", 1, d_out, d_mem, blocks, threads);
        run<12>("4 independent v_xor_b32 chains", 4, d_out, d_mem, blocks, threads);
        run<13>("4 independent v_bitop3_b32 (one SGPR operand)", 4, d_out, d_mem, blocks, threads);
        run<14>("4 independent v_mad_u64_u32 (+ 4 v_xor)", 4, d_out, d_mem, blocks, threads);
        run<15>("4 independent v_add3_u32", 4, d_out, d_mem, blocks, threads);
        run<16>("4 independent v_lshrrev_b64", 4, d_out, d_mem, blocks, threads);
        run<17>("v_bfe_u32, v_cndmask, v_sub_u32_sdwa, v_lshl_or_b32 (independent)", 4, d_out, d_mem, blocks, threads);
    }
    return 0;
}
