// microbench.hip -- gfx950 instruction-rate and copy-bandwidth probes that inform the
// JPEG/PNG kernel design (which integer multiply forms are full rate, what a plain
// dwordx4 copy reaches).  Build: hipcc --offload-arch=gfx950 -O3 tools/microbench.hip -o /tmp/microbench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstdlib>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); return 1; } } while (0)

constexpr int ITERS = 4096;

#define RATE_KERNEL(NAME, ASM)                                                          \
__global__ __launch_bounds__(256) void NAME(int* out, int a, int b)                      \
{                                                                                         \
    int x0 = threadIdx.x + a, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7; \
    for (int i = 0; i < ITERS; ++i) {                                                     \
        asm volatile(ASM(%0) ASM(%1) ASM(%2) ASM(%3) ASM(%4) ASM(%5) ASM(%6) ASM(%7)      \
                     : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(b)); \
    }                                                                                     \
    out[blockIdx.x * 256 + threadIdx.x] = x0 ^ x1 ^ x2 ^ x3 ^ x4 ^ x5 ^ x6 ^ x7;          \
}

#define A_ADD(r)    "v_add_u32 " #r ", " #r ", %8\n"
#define A_MULLO(r)  "v_mul_lo_u32 " #r ", " #r ", %8\n"
#define A_MUL24(r)  "v_mul_i32_i24 " #r ", " #r ", %8\n"
#define A_MAD24(r)  "v_mad_i32_i24 " #r ", " #r ", %8, " #r "\n"
#define A_MADU24(r) "v_mad_u32_u24 " #r ", " #r ", %8, " #r "\n"
#define A_DOT2(r)   "v_dot2_i32_i16 " #r ", " #r ", %8, " #r "\n"
#define A_PERM(r)   "v_perm_b32 " #r ", " #r ", %8, " #r "\n"
#define A_MED3(r)   "v_med3_i32 " #r ", " #r ", %8, " #r "\n"
#define A_ASHR(r)   "v_ashrrev_i32 " #r ", 3, " #r "\n"
#define A_DPPQ(r)   "v_mov_b32_dpp " #r ", " #r " quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
#define A_DPPS(r)   "v_mov_b32_dpp " #r ", " #r " row_shl:4 row_mask:0xf bank_mask:0x5\n"
#define A_LSHLADD(r) "v_lshl_add_u32 " #r ", " #r ", 3, %8\n"
#define A_ADD3(r)   "v_add3_u32 " #r ", " #r ", %8, " #r "\n"
#define A_PKADD(r)  "v_pk_add_i16 " #r ", " #r ", %8\n"
#define A_PKMUL(r)  "v_pk_mul_lo_u16 " #r ", " #r ", %8\n"
#define A_PKMAD(r)  "v_pk_mad_i16 " #r ", " #r ", %8, " #r "\n"
#define A_MADI16(r) "v_mad_i32_i16 " #r ", " #r ", %8, " #r "\n"
#define A_SAD(r)    "v_sad_u8 " #r ", " #r ", %8, " #r "\n"
#define A_BFE(r)    "v_bfe_i32 " #r ", " #r ", 0, 16\n"
#define A_MUL64(r)  "v_mul_hi_u32 " #r ", " #r ", %8\n"
#define A_FMA(r)    "v_fma_f32 " #r ", " #r ", %8, " #r "\n"
#define A_FADD(r)   "v_add_f32 " #r ", " #r ", %8\n"
#define A_FMUL(r)   "v_mul_f32 " #r ", " #r ", %8\n"
#define A_FLOOR(r)  "v_floor_f32 " #r ", " #r "\n"
#define A_CVTFI(r)  "v_cvt_f32_i32 " #r ", " #r "\n"
#define A_CVTIF(r)  "v_cvt_i32_f32 " #r ", " #r "\n"
#define A_CVTUB(r)  "v_cvt_f32_ubyte1 " #r ", " #r "\n"
#define A_CVTPK(r)  "v_cvt_pk_u8_f32 " #r ", " #r ", 1, " #r "\n"
#define A_SATPK(r)  "v_sat_pk_u8_i16 " #r ", " #r "\n"
#define A_MAXI(r)   "v_max_i32 " #r ", " #r ", %8\n"
#define A_MINI(r)   "v_min_i32 " #r ", " #r ", %8\n"
#define A_CNDM(r)   "v_cndmask_b32 " #r ", " #r ", %8, vcc\n"
#define A_AND(r)    "v_and_b32 " #r ", " #r ", %8\n"
#define A_XOR(r)    "v_xor_b32 " #r ", " #r ", %8\n"
#define A_SUB(r)    "v_sub_u32 " #r ", " #r ", %8\n"
#define A_LSHL(r)   "v_lshlrev_b32 " #r ", 3, " #r "\n"
#define A_LSHLOR(r) "v_lshl_or_b32 " #r ", " #r ", 3, %8\n"
#define A_ANDOR(r)  "v_and_or_b32 " #r ", " #r ", %8, " #r "\n"
#define A_SADU32(r) "v_sad_u32 " #r ", " #r ", %8, " #r "\n"
#define A_PKMAX(r)  "v_pk_max_i16 " #r ", " #r ", %8\n"
#define A_PKSUB(r)  "v_pk_sub_i16 " #r ", " #r ", %8\n"
#define A_ALIGNB(r) "v_alignbyte_b32 " #r ", " #r ", %8, 1\n"
#define A_PKFMA(r)  "v_pk_fma_f32 " #r ", " #r ", " #r ", " #r "\n"

RATE_KERNEL(k_add, A_ADD)
RATE_KERNEL(k_mullo, A_MULLO)
RATE_KERNEL(k_mul24, A_MUL24)
RATE_KERNEL(k_mad24, A_MAD24)
RATE_KERNEL(k_madu24, A_MADU24)
RATE_KERNEL(k_dot2, A_DOT2)
RATE_KERNEL(k_perm, A_PERM)
RATE_KERNEL(k_med3, A_MED3)
RATE_KERNEL(k_ashr, A_ASHR)
RATE_KERNEL(k_dppq, A_DPPQ)
RATE_KERNEL(k_dpps, A_DPPS)
RATE_KERNEL(k_lshladd, A_LSHLADD)
RATE_KERNEL(k_add3, A_ADD3)
RATE_KERNEL(k_pkadd, A_PKADD)
RATE_KERNEL(k_pkmul, A_PKMUL)
RATE_KERNEL(k_pkmad, A_PKMAD)
RATE_KERNEL(k_madi16, A_MADI16)
RATE_KERNEL(k_sad, A_SAD)
RATE_KERNEL(k_bfe, A_BFE)
RATE_KERNEL(k_mulhi, A_MUL64)
RATE_KERNEL(k_fma, A_FMA)
RATE_KERNEL(k_fadd, A_FADD)
RATE_KERNEL(k_fmul, A_FMUL)
RATE_KERNEL(k_floor, A_FLOOR)
RATE_KERNEL(k_cvtfi, A_CVTFI)
RATE_KERNEL(k_cvtif, A_CVTIF)
RATE_KERNEL(k_cvtub, A_CVTUB)
RATE_KERNEL(k_cvtpk, A_CVTPK)
RATE_KERNEL(k_satpk, A_SATPK)
RATE_KERNEL(k_maxi, A_MAXI)
RATE_KERNEL(k_mini, A_MINI)
RATE_KERNEL(k_cndm, A_CNDM)
RATE_KERNEL(k_and, A_AND)
RATE_KERNEL(k_xor, A_XOR)
RATE_KERNEL(k_sub, A_SUB)
RATE_KERNEL(k_lshl, A_LSHL)
RATE_KERNEL(k_lshlor, A_LSHLOR)
RATE_KERNEL(k_andor, A_ANDOR)
RATE_KERNEL(k_sadu32, A_SADU32)
RATE_KERNEL(k_pkmax, A_PKMAX)
RATE_KERNEL(k_pksub, A_PKSUB)
RATE_KERNEL(k_alignb, A_ALIGNB)

__global__ __launch_bounds__(256) void k_copy16(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t n)
{
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) dst[i] = src[i];
}
// 1 read : 2 write (the rgba16->rgbaf32 shape) and 4 read : 1 write (rgbaf32->rgba8 shape)
__global__ __launch_bounds__(256) void k_expand(const uint2* __restrict__ src, uint4* __restrict__ dst, size_t n)
{
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) { uint2 v = src[i]; dst[i] = make_uint4(v.x, v.y, v.x ^ 1, v.y ^ 1); }
}
__global__ __launch_bounds__(256) void k_shrink(const uint4* __restrict__ src, unsigned* __restrict__ dst, size_t n)
{
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) { uint4 v = src[i]; dst[i] = v.x ^ v.y ^ v.z ^ v.w; }
}

template <typename F> float time_ms(F f, int reps)
{
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    f();
    hipDeviceSynchronize();
    hipEventRecord(a);
    for (int i = 0; i < reps; ++i) f();
    hipEventRecord(b); hipEventSynchronize(b);
    float ms = 0; hipEventElapsedTime(&ms, a, b);
    return ms / reps;
}

int main()
{
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    printf("device %s CUs %d clock %d kHz\n", p.name, p.multiProcessorCount, p.clockRate);
    const int blocks = p.multiProcessorCount * 8;
    int* out; CK(hipMalloc(&out, (size_t)blocks * 256 * 4));
    struct { const char* name; void (*k)(int*, int, int); } ks[] = {
        {"v_add_u32", k_add}, {"v_mul_lo_u32", k_mullo}, {"v_mul_i32_i24", k_mul24}, {"v_mad_i32_i24", k_mad24},
        {"v_mad_u32_u24", k_madu24}, {"v_dot2_i32_i16", k_dot2}, {"v_perm_b32", k_perm}, {"v_med3_i32", k_med3},
        {"v_ashrrev_i32", k_ashr}, {"v_mov_dpp quad_perm", k_dppq}, {"v_mov_dpp row_shl4 bank", k_dpps},
        {"v_lshl_add_u32", k_lshladd}, {"v_add3_u32", k_add3}, {"v_pk_add_i16", k_pkadd}, {"v_pk_mul_lo_u16", k_pkmul},
        {"v_pk_mad_i16", k_pkmad}, {"v_mad_i32_i16", k_madi16}, {"v_sad_u8", k_sad}, {"v_bfe_i32", k_bfe}, {"v_mul_hi_u32", k_mulhi},
        {"v_fma_f32", k_fma}, {"v_add_f32", k_fadd}, {"v_mul_f32", k_fmul}, {"v_floor_f32", k_floor}, {"v_cvt_f32_i32", k_cvtfi}, {"v_cvt_i32_f32", k_cvtif}, {"v_cvt_f32_ubyte1", k_cvtub}, {"v_cvt_pk_u8_f32", k_cvtpk}, {"v_sat_pk_u8_i16", k_satpk}, {"v_max_i32", k_maxi}, {"v_min_i32", k_mini}, {"v_cndmask_b32", k_cndm}, {"v_and_b32", k_and}, {"v_xor_b32", k_xor}, {"v_sub_u32", k_sub}, {"v_lshlrev_b32", k_lshl}, {"v_lshl_or_b32", k_lshlor}, {"v_and_or_b32", k_andor}, {"v_sad_u32", k_sadu32}, {"v_pk_max_i16", k_pkmax}, {"v_pk_sub_i16", k_pksub}, {"v_alignbyte_b32", k_alignb},
    };
    for (auto& k : ks) {
        float ms = time_ms([&] { hipLaunchKernelGGL(k.k, dim3(blocks), dim3(256), 0, 0, out, 1, 3); }, 5);
        double wave_instr = (double)blocks * 4 * ITERS * 8;          // wave-instructions issued
        double per_cu_clk = wave_instr / p.multiProcessorCount / (ms * 1e-3) / (p.clockRate * 1e3);
        printf("%-26s %8.3f ms  %6.3f wave-instr/clk/CU  (%.2f cycles per wave-instr per SIMD)\n", k.name, ms, per_cu_clk, 4.0 / per_cu_clk);
    }
    if (getenv("QUICK")) return 0;
    const size_t bytes = (size_t)4 << 30;     // 4 GiB per side: far past the 256 MiB Infinity Cache
    void *s, *d; CK(hipMalloc(&s, bytes)); CK(hipMalloc(&d, bytes));
    CK(hipMemset(s, 1, bytes)); CK(hipMemset(d, 2, bytes));
    for (int bpc : {4, 8, 16, 32}) {
        const int g = p.multiProcessorCount * bpc;
        float ms = time_ms([&] { hipLaunchKernelGGL(k_copy16, dim3(g), dim3(256), 0, 0, (const uint4*)s, (uint4*)d, bytes / 16); }, 5);
        printf("copy16   %2d blocks/CU: %7.3f ms  %7.1f GB/s (read+write)\n", bpc, ms, 2.0 * bytes / ms * 1e-6);
    }
    {
        const int g = p.multiProcessorCount * 8;
        float ms = time_ms([&] { hipLaunchKernelGGL(k_expand, dim3(g), dim3(256), 0, 0, (const uint2*)s, (uint4*)d, bytes / 16); }, 5);
        printf("expand 8B->16B: %7.3f ms  %7.1f GB/s\n", ms, 1.5 * bytes / ms * 1e-6);
        ms = time_ms([&] { hipLaunchKernelGGL(k_shrink, dim3(g), dim3(256), 0, 0, (const uint4*)s, (unsigned*)d, bytes / 16); }, 5);
        printf("shrink 16B->4B: %7.3f ms  %7.1f GB/s\n", ms, 1.25 * bytes / ms * 1e-6);
    }
    return 0;
}
