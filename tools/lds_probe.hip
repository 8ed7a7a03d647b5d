// lds_probe: what do LDS accesses of the PNG ring's shapes cost on gfx950 when their addresses are not multiples of their size?
//   hipcc --offload-arch=gfx950 -O3 tools/lds_probe.hip -o tools/bin/lds_probe && tools/bin/lds_probe
// One workgroup of 8 waves (2 per SIMD, the PNG kernels' occupancy) per CU; every wave issues REP accesses of one kind to its own 18 KB
// of LDS at  lane_row * PITCH + slot * 16 + MIS  (the de-filter kernels' "co_ring + k * 8 * PITCH + offset" pattern: 8 rows x 8
// consecutive 16-byte chunks per instruction), and reports s_memtime cycles per instruction (100 MHz clock -> converted to core
// clocks by the caller's eye: what matters is the ratio to the aligned form of the same instruction).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned u32;
typedef u32 u32x4 __attribute__((ext_vector_type(4)));
typedef u32 u32x2 __attribute__((ext_vector_type(2)));
constexpr int PITCH = 288, REP = 512;
struct __attribute__((packed, aligned(1))) P16 { u32x4 v; };
struct __attribute__((packed, aligned(1))) P8 { u32x2 v; };
struct __attribute__((packed, aligned(1))) P4 { u32 v; };

template <int KIND>
__global__ __launch_bounds__(512) void k(unsigned long long* out, int mis, u32* sink)
{
    __shared__ __attribute__((aligned(16))) unsigned char lds[8][64 * PITCH + 64];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int crow = lane >> 3, cslot = lane & 7;
    unsigned char* base = lds[wave] + crow * PITCH + cslot * 16 + 16 + mis;
    for (int i = lane; i < (64 * PITCH + 64) / 4; i += 64) reinterpret_cast<u32*>(lds[wave])[i] = i;
    __syncthreads();
    u32x4 v = { (u32)lane, 1u, 2u, 3u };
    u32 acc = 0;
    const unsigned long long t0 = __builtin_readcyclecounter();
    #pragma unroll 8
    for (int r = 0; r < REP; ++r) {
        unsigned char* p = base + (r & 7) * 8 * PITCH;
        if constexpr (KIND == 0) reinterpret_cast<P16*>(p)->v = v;                                    // one 16-byte write
        else if constexpr (KIND == 1) { reinterpret_cast<P8*>(p)->v = u32x2{ v.x, v.y }; reinterpret_cast<P8*>(p + 8)->v = u32x2{ v.z, v.w }; }
        else if constexpr (KIND == 2) { reinterpret_cast<P4*>(p)->v = v.x; reinterpret_cast<P4*>(p + 4)->v = v.y; reinterpret_cast<P4*>(p + 8)->v = v.z; reinterpret_cast<P4*>(p + 12)->v = v.w; }
        else if constexpr (KIND == 3) { const u32x4 q = reinterpret_cast<const P16*>(p)->v; acc += q.x ^ q.y ^ q.z ^ q.w; }
        else if constexpr (KIND == 4) { const u32x2 a = reinterpret_cast<const P8*>(p)->v, b = reinterpret_cast<const P8*>(p + 8)->v; acc += a.x ^ a.y ^ b.x ^ b.y; }
        else { acc += reinterpret_cast<const P4*>(p)->v ^ reinterpret_cast<const P4*>(p + 4)->v ^ reinterpret_cast<const P4*>(p + 8)->v ^ reinterpret_cast<const P4*>(p + 12)->v; }
        v.x += acc;
        asm volatile("" ::: "memory");
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (lane == 0) out[blockIdx.x * 8 + wave] = t1 - t0;
    if (acc == 0x12345u) sink[0] = acc + v.x;
}

int main()
{
    int cus = 0; hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    unsigned long long* d; u32* sink; hipMalloc(&d, cus * 8 * 8); hipMalloc(&sink, 4);
    std::vector<unsigned long long> h(cus * 8);
    const char* names[6] = { "write 1 x 16 B", "write 2 x 8 B", "write 4 x 4 B", "read 1 x 16 B", "read 2 x 8 B", "read 4 x 4 B" };
    for (int kind = 0; kind < 6; ++kind)
        for (int mis : { 0, 4, 8, 12, 1, 2, 3, 5, 7, 13 }) {
            for (int rep = 0; rep < 2; ++rep) {
                switch (kind) {
                case 0: hipLaunchKernelGGL(k<0>, dim3(cus), dim3(512), 0, 0, d, mis, sink); break;
                case 1: hipLaunchKernelGGL(k<1>, dim3(cus), dim3(512), 0, 0, d, mis, sink); break;
                case 2: hipLaunchKernelGGL(k<2>, dim3(cus), dim3(512), 0, 0, d, mis, sink); break;
                case 3: hipLaunchKernelGGL(k<3>, dim3(cus), dim3(512), 0, 0, d, mis, sink); break;
                case 4: hipLaunchKernelGGL(k<4>, dim3(cus), dim3(512), 0, 0, d, mis, sink); break;
                default: hipLaunchKernelGGL(k<5>, dim3(cus), dim3(512), 0, 0, d, mis, sink); break;
                }
                hipDeviceSynchronize();
            }
            hipMemcpy(h.data(), d, cus * 8 * 8, hipMemcpyDeviceToHost);
            double s = 0; for (auto x : h) s += (double)x;
            printf("%-16s address %% 16 = %2d : %8.2f clock ticks per 16 bytes x 64 lanes (8 waves per CU at once)\n", names[kind], mis, s / h.size() / REP);
        }
    return 0;
}
