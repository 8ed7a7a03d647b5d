// copy_probe.hip -- what does a streaming copy reach on this part, and with which launch shape?  (the ceiling every kernel of
// the path is measured against: DESIGN.md section 3).  hipcc --offload-arch=gfx950 -O3 tools/copy_probe.hip -o tools/bin/copy_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// grid-stride
template <int T> __global__ __launch_bounds__(T) void k_stride(const u32x4* __restrict__ s, u32x4* __restrict__ d, size_t n)
{
    const size_t stride = (size_t)gridDim.x * T;
    for (size_t i = (size_t)blockIdx.x * T + threadIdx.x; i < n; i += stride) d[i] = s[i];
}
// one shot: a block owns U * T consecutive 16-byte units, all loads issued before the stores.  NT: 0 plain, 1 nt loads, 2 nt stores, 3 both
// XCD: remap the block index so that each XCD (block % 8) walks one contiguous eighth of the buffer
template <int T, int U, int NT, int XCD> __global__ __launch_bounds__(T) void k_shot(const u32x4* __restrict__ s, u32x4* __restrict__ d, size_t n)
{
    size_t b = blockIdx.x;
    if (XCD) b = (b & 7) * (gridDim.x >> 3) + (b >> 3);
    const size_t base = b * (size_t)(U * T) + threadIdx.x;
    u32x4 v[U];
    #pragma unroll
    for (int k = 0; k < U; ++k) { const size_t i = base + (size_t)k * T; if (i < n) v[k] = (NT & 1) ? __builtin_nontemporal_load(s + i) : s[i]; }
    #pragma unroll
    for (int k = 0; k < U; ++k) { const size_t i = base + (size_t)k * T; if (i < n) { if (NT & 2) __builtin_nontemporal_store(v[k], d + i); else d[i] = v[k]; } }
}
// persistent blocks, each iteration U units in flight
template <int T, int U, int NT> __global__ __launch_bounds__(T) void k_persist(const u32x4* __restrict__ s, u32x4* __restrict__ d, size_t n)
{
    const size_t step = (size_t)gridDim.x * (U * T);
    for (size_t base = (size_t)blockIdx.x * (U * T) + threadIdx.x; base < n; base += step) {
        u32x4 v[U];
        #pragma unroll
        for (int k = 0; k < U; ++k) { const size_t i = base + (size_t)k * T; if (i < n) v[k] = (NT & 1) ? __builtin_nontemporal_load(s + i) : s[i]; }
        #pragma unroll
        for (int k = 0; k < U; ++k) { const size_t i = base + (size_t)k * T; if (i < n) { if (NT & 2) __builtin_nontemporal_store(v[k], d + i); else d[i] = v[k]; } }
    }
}
template <int T, int U> __global__ __launch_bounds__(T) void k_read(const u32x4* __restrict__ s, uint32_t* __restrict__ d, size_t n)
{
    const size_t base = (size_t)blockIdx.x * (U * T) + threadIdx.x;
    uint32_t acc = 0;
    #pragma unroll
    for (int k = 0; k < U; ++k) { const size_t i = base + (size_t)k * T; if (i < n) { const u32x4 v = s[i]; acc ^= v.x ^ v.y ^ v.z ^ v.w; } }
    if (acc == 0x12345678u) d[0] = acc;
}
template <int T, int U, int NT> __global__ __launch_bounds__(T) void k_write(u32x4* __restrict__ d, size_t n)
{
    const size_t base = (size_t)blockIdx.x * (U * T) + threadIdx.x;
    const u32x4 v = {1u, 2u, 3u, (uint32_t)threadIdx.x};
    #pragma unroll
    for (int k = 0; k < U; ++k) { const size_t i = base + (size_t)k * T; if (i < n) { if (NT) __builtin_nontemporal_store(v, d + i); else d[i] = v; } }
}

// 16-byte loads, but the stores are SW bytes per lane (4 or 8): the k-th store instruction of a block writes a lane-contiguous
// run of T * SW bytes (what a kernel that produces one pixel per lane per row does)
template <int T, int SW, int NT> __global__ __launch_bounds__(T) void k_narrow_store(const u32x4* __restrict__ s, uint32_t* __restrict__ d, size_t n)
{
    const size_t i = (size_t)blockIdx.x * T + threadIdx.x;
    if (i >= n) return;
    const u32x4 v = s[i];
    uint32_t* o = d + (size_t)blockIdx.x * T * 4;
    if (SW == 4) {
        #pragma unroll
        for (int k = 0; k < 4; ++k) { if (NT) __builtin_nontemporal_store(v[k], o + k * T + threadIdx.x); else o[k * T + threadIdx.x] = v[k]; }
    } else {
        typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
        u32x2* o2 = reinterpret_cast<u32x2*>(o);
        #pragma unroll
        for (int k = 0; k < 2; ++k) { const u32x2 w = { v[2 * k], v[2 * k + 1] }; if (NT) __builtin_nontemporal_store(w, o2 + k * T + threadIdx.x); else o2[k * T + threadIdx.x] = w; }
    }
}

// Tiled stores: the source is read contiguously (tile after tile), a tile is written as ROWS row segments of SEG bytes into
// rows of PITCH bytes (what a block-based image kernel does: the JPEG kernel writes 16 rows x 512 bytes per workgroup)
template <int SEG, int ROWS> __global__ __launch_bounds__(256) void k_tiled_store(const u32x4* __restrict__ s, uint8_t* __restrict__ d, int tiles_per_row, size_t pitch, size_t n_tiles)
{
    constexpr int CH = SEG / 16, PER_TILE = CH * ROWS;          // 16-byte chunks per tile
    const size_t tile = blockIdx.x;
    if (tile >= n_tiles) return;
    const size_t band = tile / tiles_per_row, tx = tile % tiles_per_row;
    for (int c = threadIdx.x; c < PER_TILE; c += 256) {
        const u32x4 v = s[tile * PER_TILE + c];
        const int row = c / CH, col = c % CH;
        *reinterpret_cast<u32x4*>(d + (band * ROWS + row) * pitch + tx * SEG + (size_t)col * 16) = v;
    }
}

template <typename F> float time_ms(F f, int reps)
{
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    f(); hipDeviceSynchronize();
    hipEventRecord(a);
    for (int i = 0; i < reps; ++i) f();
    hipEventRecord(b); hipEventSynchronize(b);
    float ms = 0; hipEventElapsedTime(&ms, a, b);
    return ms / reps;
}

int main()
{
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    const int cus = p.multiProcessorCount;
    const size_t bytes = (size_t)4 << 30, n = bytes / 16;
    void *s, *d; CK(hipMalloc(&s, bytes)); CK(hipMalloc(&d, bytes));
    CK(hipMemset(s, 1, bytes)); CK(hipMemset(d, 2, bytes));
    const int R = 5;
    auto report = [&](const char* name, float ms, double moved) { printf("%-44s %7.3f ms  %7.1f GB/s\n", name, ms, moved / ms * 1e-6); fflush(stdout); };
    char name[96];
    report("hipMemcpyDtoD", time_ms([&] { (void)hipMemcpyAsync(d, s, bytes, hipMemcpyDeviceToDevice, 0); }, R), 2.0 * bytes);
    for (int bpc : {1, 2, 4, 8}) {
        snprintf(name, sizeof name, "grid-stride T=256, %d blocks/CU", bpc);
        report(name, time_ms([&] { hipLaunchKernelGGL(k_stride<256>, dim3(cus * bpc), dim3(256), 0, 0, (const u32x4*)s, (u32x4*)d, n); }, R), 2.0 * bytes);
    }
    for (int bpc : {1, 2}) {
        snprintf(name, sizeof name, "grid-stride T=1024, %d blocks/CU", bpc);
        report(name, time_ms([&] { hipLaunchKernelGGL(k_stride<1024>, dim3(cus * bpc), dim3(1024), 0, 0, (const u32x4*)s, (u32x4*)d, n); }, R), 2.0 * bytes);
    }
#define SHOT(T, U, NT, X) do { snprintf(name, sizeof name, "one-shot T=%d U=%d nt=%d xcd=%d", T, U, NT, X); \
        const unsigned g = (unsigned)((n + (size_t)(T) * (U) - 1) / ((size_t)(T) * (U))); \
        report(name, time_ms([&] { hipLaunchKernelGGL((k_shot<T, U, NT, X>), dim3(g), dim3(T), 0, 0, (const u32x4*)s, (u32x4*)d, n); }, R), 2.0 * bytes); } while (0)
    SHOT(256, 1, 0, 0); SHOT(256, 2, 0, 0); SHOT(256, 4, 0, 0); SHOT(256, 8, 0, 0); SHOT(512, 4, 0, 0); SHOT(1024, 4, 0, 0); SHOT(64, 8, 0, 0);
    SHOT(256, 4, 1, 0); SHOT(256, 4, 2, 0); SHOT(256, 4, 3, 0); SHOT(256, 8, 3, 0);
    SHOT(256, 4, 0, 1); SHOT(256, 4, 3, 1);
#define PERS(T, U, NT, BPC) do { snprintf(name, sizeof name, "persistent T=%d U=%d nt=%d, %d blocks/CU", T, U, NT, BPC); \
        report(name, time_ms([&] { hipLaunchKernelGGL((k_persist<T, U, NT>), dim3(cus * (BPC)), dim3(T), 0, 0, (const u32x4*)s, (u32x4*)d, n); }, R), 2.0 * bytes); } while (0)
    PERS(256, 4, 0, 2); PERS(256, 4, 0, 4); PERS(256, 4, 0, 8); PERS(256, 8, 0, 2); PERS(256, 4, 3, 4); PERS(512, 4, 0, 2); PERS(256, 2, 0, 8);
    {
        const unsigned g = (unsigned)((n + 1023) / 1024);
        report("read only  T=256 U=4", time_ms([&] { hipLaunchKernelGGL((k_read<256, 4>), dim3(g), dim3(256), 0, 0, (const u32x4*)s, (uint32_t*)d, n); }, R), 1.0 * bytes);
        report("write only T=256 U=4", time_ms([&] { hipLaunchKernelGGL((k_write<256, 4, 0>), dim3(g), dim3(256), 0, 0, (u32x4*)d, n); }, R), 1.0 * bytes);
        report("write only T=256 U=4 nt", time_ms([&] { hipLaunchKernelGGL((k_write<256, 4, 1>), dim3(g), dim3(256), 0, 0, (u32x4*)d, n); }, R), 1.0 * bytes);
    }
    {
        const unsigned g = (unsigned)((n + 255) / 256);
        report("one-shot 16-B loads, 4-B stores", time_ms([&] { hipLaunchKernelGGL((k_narrow_store<256, 4, 0>), dim3(g), dim3(256), 0, 0, (const u32x4*)s, (uint32_t*)d, n); }, R), 2.0 * bytes);
        report("one-shot 16-B loads, 4-B stores nt", time_ms([&] { hipLaunchKernelGGL((k_narrow_store<256, 4, 1>), dim3(g), dim3(256), 0, 0, (const u32x4*)s, (uint32_t*)d, n); }, R), 2.0 * bytes);
        report("one-shot 16-B loads, 8-B stores", time_ms([&] { hipLaunchKernelGGL((k_narrow_store<256, 8, 0>), dim3(g), dim3(256), 0, 0, (const u32x4*)s, (uint32_t*)d, n); }, R), 2.0 * bytes);
        report("one-shot 16-B loads, 8-B stores nt", time_ms([&] { hipLaunchKernelGGL((k_narrow_store<256, 8, 1>), dim3(g), dim3(256), 0, 0, (const u32x4*)s, (uint32_t*)d, n); }, R), 2.0 * bytes);
    }
    {
        const size_t pitch = 7680;                                        // 1920 rgba8 pixels
#define TILED(SEG, ROWS) do { const int tpr = (int)(pitch / (SEG)); const size_t nt = (bytes / (pitch * (ROWS)) - 1) * tpr; \
        snprintf(name, sizeof name, "tiled stores: %d rows x %d B per block, pitch 7680", ROWS, SEG); \
        report(name, time_ms([&] { hipLaunchKernelGGL((k_tiled_store<SEG, ROWS>), dim3((unsigned)nt), dim3(256), 0, 0, (const u32x4*)s, (uint8_t*)d, tpr, pitch, nt); }, R), 2.0 * nt * (SEG) * (ROWS)); } while (0)
        TILED(512, 16); TILED(256, 16); TILED(1536, 16); TILED(512, 8); TILED(7680, 1); TILED(3840, 2); TILED(1536, 8); TILED(512, 32);
    }
    return 0;
}
