// valu_latency.hip -- how many clocks does a wave need per VALU instruction when every instruction reads the result of the one
// before it (chains = 1), of the one two before it (chains = 2) ... for the instruction kinds the PNG Paeth predictor is made of.
// One wave per SIMD (so nothing else hides a stall) and, second table, two waves per SIMD (the PNG kernel's occupancy).
//   hipcc --offload-arch=gfx950 -O3 tools/microbench/valu_latency.hip -o /tmp/valu_latency && /tmp/valu_latency
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint32_t u32;
#define REP16(x) x x x x x x x x x x x x x x x x
template <int OP, int CH> __global__ void k(u32* out, uint64_t* ticks, int iters)
{
    u32 a[4] = { threadIdx.x, threadIdx.x * 3u, threadIdx.x ^ 0x55u, threadIdx.x + 7u }, b = 0x64016402u + threadIdx.x, c = 0x04010400u;
    const uint64_t t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
#define STEP(j) \
        if constexpr (OP == 0) asm volatile("v_pk_add_f16 %0, %0, %1" : "+v"(a[j % CH]) : "v"(b)); \
        else if constexpr (OP == 1) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(a[j % CH]) : "v"(b), "v"(c)); \
        else if constexpr (OP == 2) asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0xe4" : "+v"(a[j % CH]) : "v"(b), "v"(c)); \
        else if constexpr (OP == 3) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[j % CH]) : "v"(b)); \
        else if constexpr (OP == 4) asm volatile("v_pk_max_f16 %0, %0, %0 neg_lo:[0,1] neg_hi:[0,1]" : "+v"(a[j % CH])); \
        else if constexpr (OP == 5) asm volatile("v_lerp_u8 %0, %0, %1, 0" : "+v"(a[j % CH]) : "v"(b)); \
        else if constexpr (OP == 6) asm volatile("v_pk_add_f16 %0, %0, %1\n\tv_and_b32 %2, %2, %1" : "+v"(a[0]), "+v"(b), "+v"(a[1]));  /* pk, then an independent int op */ \
        else if constexpr (OP == 7) asm volatile("v_mov_b32_dpp %0, %0 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[j % CH]));
        STEP(0) STEP(1) STEP(2) STEP(3) STEP(4) STEP(5) STEP(6) STEP(7) STEP(8) STEP(9) STEP(10) STEP(11) STEP(12) STEP(13) STEP(14) STEP(15)
    }
    const uint64_t t1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a[0] ^ a[1] ^ a[2] ^ a[3];
    if (threadIdx.x == 0 && blockIdx.x == 0) *ticks = t1 - t0;
}
template <int OP, int CH> void run(const char* name, int waves_per_simd)
{
    u32* out; uint64_t* ticks; hipMalloc(&out, 1 << 24); hipMalloc(&ticks, 8);
    const int iters = 20000, threads = 256 * waves_per_simd;            // one workgroup per CU: 4 (8) waves = 1 (2) per SIMD
    k<OP, CH><<<256, threads>>>(out, ticks, 10);
    k<OP, CH><<<256, threads>>>(out, ticks, iters);
    uint64_t t; hipMemcpy(&t, ticks, 8, hipMemcpyDeviceToHost);
    const int per_it = OP == 6 ? 32 : 16;
    printf("%-44s chains %d  waves/SIMD %d : %6.2f clocks per instruction of one wave (s_memtime: 100 MHz -> x %.1f)\n", name, CH, waves_per_simd, (double)t / iters / per_it * 24.0, 24.0);
    hipFree(out); hipFree(ticks);
}
int main()
{
    for (int w = 1; w <= 2; ++w) {
        run<3, 1>("v_add_u32", w); run<3, 2>("v_add_u32", w);
        run<0, 1>("v_pk_add_f16", w); run<0, 2>("v_pk_add_f16", w); run<0, 3>("v_pk_add_f16", w); run<0, 4>("v_pk_add_f16", w);
        run<4, 1>("v_pk_max_f16 neg", w); run<4, 2>("v_pk_max_f16 neg", w);
        run<1, 1>("v_perm_b32", w); run<1, 2>("v_perm_b32", w);
        run<2, 1>("v_bitop3_b32", w); run<2, 2>("v_bitop3_b32", w);
        run<5, 1>("v_lerp_u8", w); run<5, 2>("v_lerp_u8", w);
        run<7, 1>("v_mov_b32_dpp wave_shr:1", w); run<7, 2>("v_mov_b32_dpp wave_shr:1", w);
        run<6, 1>("v_pk_add_f16 chain + independent v_and", w);
    }
    return 0;
}
