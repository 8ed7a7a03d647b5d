// what do v_perm_b32 selectors 8..15 return?  hipcc --offload-arch=gfx950 -o /tmp/perm_probe tools/perm_probe.hip && /tmp/perm_probe
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* out, unsigned s0, unsigned s1)
{
    unsigned sel[8] = { 0x08080808u, 0x09090909u, 0x0a0a0a0au, 0x0b0b0b0bu, 0x0c0c0c0cu, 0x0d0d0d0du, 0x0e0e0e0eu, 0x0f0f0f0fu };
    for (int i = 0; i < 8; ++i) out[i] = __builtin_amdgcn_perm(s0, s1, sel[i] + (threadIdx.x ? 1 : 0));
}
int main()
{
    unsigned* d; hipMalloc(&d, 64);
    // each pattern sets exactly one candidate sign bit
    const unsigned pats[8][2] = { {0, 0x00000080u}, {0, 0x00008000u}, {0, 0x00800000u}, {0, 0x80000000u}, {0x00000080u, 0}, {0x00008000u, 0}, {0x00800000u, 0}, {0x80000000u, 0} };
    for (auto& p : pats) {
        hipLaunchKernelGGL(k, 1, 1, 0, 0, d, p[0], p[1]);
        unsigned h[8]; hipMemcpy(h, d, 32, hipMemcpyDeviceToHost);
        printf("S0=%08x S1=%08x :", p[0], p[1]);
        for (int i = 0; i < 8; ++i) printf(" sel%x=%02x", 8 + i, h[i] & 0xff);
        printf("\n");
    }
    return 0;
}
