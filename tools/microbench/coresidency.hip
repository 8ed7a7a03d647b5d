// coresidency.hip -- how many workgroups of a given shape (threads, static LDS bytes) does a compute unit of this part hold AT ONCE?
//   hipcc --offload-arch=gfx950 -O2 tools/microbench/coresidency.hip -o tools/bin/coresidency && tools/bin/coresidency
// Every workgroup counts itself in, then waits (bounded) until K x CUs workgroups have done so: if the part holds K of them per CU they all meet;
// if not, the first wave of arrivals times out.  Prints, per shape and K, whether they all met (and how many gave up waiting if not),
// and what hipOccupancyMaxActiveBlocksPerMultiprocessor says.  (Asked in round 6 about k_qoi_pipe: 320 threads, 62 312 bytes of LDS -- two of its
// workgroups on a compute unit take exactly twice one's time.)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
template <int LDS> __global__ void k_meet(unsigned* arrived, unsigned need, unsigned* seen_max, unsigned long long ticks)
{
    __shared__ unsigned char lds[LDS];
    if (threadIdx.x < 64) lds[threadIdx.x * (LDS / 64)] = (unsigned char)threadIdx.x;      // (the array is used)
    __syncthreads();
    if (threadIdx.x == 0) {
        atomicAdd(arrived, 1u);
        const unsigned long long t0 = wall_clock64();
        unsigned s = 0;
        while ((s = atomicAdd(arrived, 0u)) < need && wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
        if (s < need) atomicAdd(seen_max, 1u);                                             // gave up: not everybody was there at once
        if (lds[3 * (LDS / 64)] == 77) arrived[1] = 1;
    }
}
template <int LDS> int run(int threads, int cus)
{
    unsigned* d; CK(hipMalloc(&d, 64));
    int occ = 0; CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k_meet<LDS>, threads, 0));
    printf("%4d threads, %6d bytes of LDS: occupancy query says %d per CU;", threads, LDS, occ);
    for (int K = 1; K <= 4; ++K) {
        CK(hipMemset(d, 0, 64));
        const unsigned need = (unsigned)(K * cus);
        hipLaunchKernelGGL(k_meet<LDS>, dim3(need), dim3(threads), 0, 0, d, need, d + 2, 20000000ull);      // 0.2 s of the 100 MHz clock
        CK(hipDeviceSynchronize());
        unsigned h[4]; CK(hipMemcpy(h, d, 16, hipMemcpyDeviceToHost));
        printf("  %d per CU: %s", K, h[2] == 0 ? "yes" : "NO");
        if (h[2]) { printf(" (%u of %u gave up waiting)", h[2], need); break; }
    }
    printf("\n");
    CK(hipFree(d));
    return 0;
}
int main()
{
    int cus = 0; CK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
    printf("%d compute units\n", cus);
    if (run<62312>(320, cus)) return 1;
    if (run<62312>(256, cus)) return 1;
    if (run<50112>(256, cus)) return 1;
    if (run<50112>(320, cus)) return 1;
    if (run<32768>(320, cus)) return 1;
    if (run<12328>(64, cus)) return 1;
    if (run<148480>(512, cus)) return 1;
    return 0;
}
