// probe: do byte-unaligned global dword / dwordx4 loads and stores work on gfx950 under ROCm's default alignment mode?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
__global__ void k(const unsigned char* src, unsigned char* dst, int off)
{
    const int t = threadIdx.x;
    typedef unsigned v4u __attribute__((ext_vector_type(4)));
    v4u v; unsigned w;
    asm volatile("global_load_dwordx4 %0, %1, off\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(src + off + t * 16) : "memory");
    asm volatile("global_load_dword %0, %1, off\n s_waitcnt vmcnt(0)" : "=v"(w) : "v"(src + off + t * 4) : "memory");
    asm volatile("global_store_dwordx4 %0, %1, off\n s_waitcnt vmcnt(0)" :: "v"(dst + off + t * 16), "v"(v) : "memory");
    asm volatile("global_store_dword %0, %1, off\n s_waitcnt vmcnt(0)" :: "v"(dst + 4096 + off + t * 4), "v"(w) : "memory");
}
int main()
{
    std::vector<unsigned char> h(8192), o(8192);
    for (int i = 0; i < 8192; ++i) h[i] = (unsigned char)(i * 7 + 3);
    unsigned char *s, *d; hipMalloc(&s, 8192); hipMalloc(&d, 8192);
    for (int off : {0, 1, 2, 3, 5, 13}) {
        hipMemcpy(s, h.data(), 8192, hipMemcpyHostToDevice); hipMemset(d, 0, 8192);
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, s, d, off);
        hipError_t e = hipDeviceSynchronize();
        hipMemcpy(o.data(), d, 8192, hipMemcpyDeviceToHost);
        bool ok1 = !memcmp(o.data() + off, h.data() + off, 1024), ok2 = !memcmp(o.data() + 4096 + off, h.data() + off, 256);
        printf("offset %2d: %s  dwordx4 %s  dword %s\n", off, hipGetErrorString(e), ok1 ? "OK" : "WRONG", ok2 ? "OK" : "WRONG");
    }
    return 0;
}
