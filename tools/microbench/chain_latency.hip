// Latency of dependent instruction chains on ONE wave per SIMD (gfx950): what a serial decoder's symbol step is made of.
//   hipcc --offload-arch=gfx950 -O2 tools/microbench/chain_latency.hip -o /tmp/chain_latency && /tmp/chain_latency
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))
template <int KIND> __global__ void k(uint32_t* out, int iters, uint32_t seed)
{
    uint32_t s = __builtin_amdgcn_readfirstlane(seed), v = threadIdx.x + seed, t = __builtin_amdgcn_readfirstlane(seed + 1);
    uint64_t q = ((uint64_t)s << 32) | t;
    const uint64_t t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
        if (KIND == 0) { REP64(asm volatile("s_add_u32 %0, %0, %1" : "+s"(s) : "s"(t) : "scc");) }
        if (KIND == 1) { REP64(asm volatile("s_lshl_b64 %0, %0, 1" : "+s"(q) :: "scc");) }
        if (KIND == 2) { REP64(asm volatile("v_add_u32 %0, %0, %1" : "+v"(v) : "v"(v));) }
        if (KIND == 3) { REP64(asm volatile("s_and_b32 %0, %0, 63\n v_readlane_b32 %0, %1, %0" : "+s"(s) : "v"(v) : "scc");) }       // SALU -> readlane -> SALU
        if (KIND == 4) { REP64(asm volatile("v_readfirstlane_b32 %0, %1\n v_add_u32 %1, %0, %1" : "+s"(s), "+v"(v));) }        // VALU -> SGPR -> VALU
        if (KIND == 5) { REP64(asm volatile("s_cmp_lg_u32 %0, 0\n s_cbranch_scc1 1f\n s_nop 0\n1:\n s_add_u32 %0, %0, 1" : "+s"(s) :: "scc");) }   // taken forward branch
        if (KIND == 6) { REP64(asm volatile("s_cmp_eq_u32 %0, 0\n s_cbranch_scc1 1f\n s_nop 0\n1:\n s_add_u32 %0, %0, 1" : "+s"(s) :: "scc");) }   // untaken branch
        if (KIND == 7) { REP64(asm volatile("s_cmp_lg_u32 %0, %1\n s_cselect_b32 %0, %0, %1" : "+s"(s) : "s"(t) : "scc");) }
        if (KIND == 8) { REP64(asm volatile("s_bcnt1_i32_b64 %0, %1\n s_lshl_b64 %1, %1, %0" : "+s"(s), "+s"(q) :: "scc");) }
        if (KIND == 11) { REP64(asm volatile("s_add_u32 %0, %0, %1\n s_add_u32 %2, %2, %1" : "+s"(s), "+s"(t), "+s"(seed) :: "scc");) }   // two independent SALU chains
    }
    const uint64_t t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) { out[blockIdx.x * 4] = (uint32_t)(t1 - t0); out[blockIdx.x * 4 + 1] = s + v + (uint32_t)q + t; }
}
template <int KIND> void run(const char* name, int per, int waves_per_simd)
{
    uint32_t* d; (void)hipMalloc(&d, 1 << 20);
    const int iters = 2000;
    const int blocks = 256 * 4 * waves_per_simd;
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(64), 0, 0, d, 10, 3u);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(64), 0, 0, d, iters, 3u);
    uint32_t h[4]; (void)hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
    setvbuf(stdout, nullptr, _IOLBF, 0); printf("%-52s %d wave(s)/SIMD: %7.2f shader clocks per step\n", name, waves_per_simd, (double)h[0] / ((double)iters * 64 * 1) * 1.0 / 1.0 / (per ? 1 : 1));
    (void)hipFree(d);
}
int main()
{
    for (int w : { 1, 4 }) {
        if (w == 1) {
        run<0>("s_add_u32 (dependent)", 1, 1); run<11>("two independent s_add_u32", 1, 1); run<1>("s_lshl_b64 (dependent)", 1, 1); run<2>("v_add_u32 (dependent)", 1, 1);
        run<3>("s_and + v_readlane(SGPR index) -> SGPR (dependent)", 1, 1); run<4>("v_readfirstlane -> v_add (dependent)", 1, 1);
        run<5>("s_cmp + taken forward branch + s_add", 1, 1); run<6>("s_cmp + untaken branch + s_nop + s_add", 1, 1); run<7>("s_cmp + s_cselect (dependent)", 1, 1);
        run<8>("s_bcnt1_b64 + s_lshl_b64 (dependent)", 1, 1);
        } else {
        run<0>("s_add_u32 (dependent)", 1, 4); run<3>("s_and + v_readlane(SGPR index) -> SGPR (dependent)", 1, 4); run<5>("s_cmp + taken forward branch + s_add", 1, 4);
        run<2>("v_add_u32 (dependent)", 1, 4);
        }
    }
    return 0;
}
