// rowstream.hip -- what does the MEMORY side of the PNG de-filter's access pattern cost on its own, and how much of it is the placement
// of the buffers?   hipcc --offload-arch=gfx950 -O3 tools/microbench/rowstream.hip -o tools/bin/rowstream && tools/bin/rowstream
//
// The de-filter kernels walk images with lane = row: a wave owns 64 consecutive rows and, per step, moves CH contiguous bytes of EVERY
// one of them (the PNG ring kernel: CH = 128, one aligned line per row and tile), so at any instant the chip has 2 048 waves x 64 row
// streams open, 15 360 bytes apart, all over two 17 GB buffers -- where a linear copy has one sliding window of a few MB.
// tools/png_mode_probe.py showed the heuristic-filter case (HBM-bound) moving between 6.03 and 6.96 ms on fresh PHYSICAL placements of
// the same virtual addresses, while this library's linear copy kernel holds 6.2 TB/s to 1 % on any placement (tools/placement_probe.py).
// This probe is the pattern without the arithmetic: a persistent grid (one workgroup of 8 waves per compute unit), (image, band) units
// drawn from a queue band-major like k_png_defilter_queue, per step and wave 64 rows x CH bytes loaded and stored as 16-byte pieces
// (CH / 16 lanes per row), DEPTH steps of loads in flight.  Per CH: K fresh placements (hipFree + hipMalloc), every one timed.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

struct Args { const uint8_t* src; uint8_t* dst; unsigned* q; unsigned images, bands, rows, pitch_s, pitch_d; size_t stride_s, stride_d; };

// MODE 0: read + write, 1: read only, 2: write only
template <int CH, int DEPTH, int MODE, int NT> __global__ __launch_bounds__(512, 2) void k_rows(Args a)
{
    constexpr int LPR = CH / 16;                 // lanes per row and instruction
    constexpr int RPI = 64 / LPR;                // rows per instruction
    constexpr int NI = 64 / RPI;                 // instructions per step: 64 rows
    const int lane = threadIdx.x & 63;
    const int crow = lane / LPR, cslot = lane % LPR;
    const unsigned total = a.images * a.bands;
    uint32_t acc = 0;
    for (;;) {
        unsigned u = 0;
        if (lane == 0) u = atomicAdd(a.q, 1u);
        u = (unsigned)__builtin_amdgcn_readfirstlane((int)u);
        if (u >= total) break;
        const unsigned band = u / a.images, img = u - band * a.images;       // band-major
        const unsigned row0 = band * 64;
        const uint8_t* s = a.src + (size_t)img * a.stride_s + (size_t)row0 * a.pitch_s;
        uint8_t* d = a.dst + (size_t)img * a.stride_d + (size_t)row0 * a.pitch_d;
        const unsigned steps = a.pitch_d / CH;
        u32x4 v[DEPTH][NI];
        auto load = [&](unsigned st, u32x4 (&r)[NI]) {
            #pragma unroll
            for (int k = 0; k < NI; ++k) {
                const unsigned row = (unsigned)(k * RPI + crow);
                const unsigned rr = row0 + row < a.rows ? row : 0;
                r[k] = *reinterpret_cast<const u32x4*>(s + (size_t)rr * a.pitch_s + (size_t)st * CH + cslot * 16);
            }
        };
        if (MODE != 2) {
            #pragma unroll
            for (int p = 0; p < DEPTH; ++p) load(p < (int)steps ? p : 0, v[p]);
        }
        for (unsigned st = 0; st < steps; st += DEPTH) {
            #pragma unroll
            for (int p = 0; p < DEPTH; ++p) {
                const unsigned cur = st + p;
                if (cur >= steps) break;
                u32x4 w[NI];
                #pragma unroll
                for (int k = 0; k < NI; ++k) w[k] = MODE == 2 ? u32x4{ cur, (uint32_t)lane, 3u, 4u } : v[p][k];
                if (MODE != 2) load(cur + DEPTH < steps ? cur + DEPTH : cur, v[p]);
                #pragma unroll
                for (int k = 0; k < NI; ++k) {
                    const unsigned row = (unsigned)(k * RPI + crow);
                    if (MODE == 1) { acc ^= w[k].x ^ w[k].y ^ w[k].z ^ w[k].w; continue; }
                    if (row0 + row < a.rows) {
                        u32x4* p_ = reinterpret_cast<u32x4*>(d + (size_t)row * a.pitch_d + (size_t)cur * CH + cslot * 16);
                        if (NT) __builtin_nontemporal_store(w[k], p_); else *p_ = w[k];
                    }
                }
            }
        }
    }
    if (acc == 0x12345678u) a.dst[0] = (uint8_t)acc;
}

// the linear reference: same bytes, one sliding window
__global__ __launch_bounds__(256) void k_linear(const u32x4* __restrict__ s, u32x4* __restrict__ d, size_t n)
{
    const size_t base = (size_t)blockIdx.x * (8 * 256) + threadIdx.x;
    u32x4 v[8];
    #pragma unroll
    for (int k = 0; k < 8; ++k) { const size_t i = base + (size_t)k * 256; if (i < n) v[k] = s[i]; }
    #pragma unroll
    for (int k = 0; k < 8; ++k) { const size_t i = base + (size_t)k * 256; if (i < n) __builtin_nontemporal_store(v[k], d + i); }
}

int main(int argc, char** argv)
{
    const int K = argc > 1 ? atoi(argv[1]) : 6;
    const unsigned images = argc > 2 ? atoi(argv[2]) : 512, w = argc > 3 ? atoi(argv[3]) : 3840, h = 2160;
    const unsigned pitch = w * 4, bands = (h + 63) / 64;
    const size_t stride = (size_t)pitch * h, bytes = stride * images;
    int cus = 0; CK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
    unsigned* q; CK(hipMalloc(&q, 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    struct Cfg { const char* name; int id; };
    const Cfg cfgs[] = { { "linear copy (nt stores)", 0 }, { "rows CH=128 depth 2 r+w nt", 1 }, { "rows CH=256 depth 2 r+w nt", 2 }, { "rows CH=512 depth 1 r+w nt", 3 },
                         { "rows CH=1024 depth 1 r+w nt", 4 }, { "rows CH=128 depth 2 read only", 5 }, { "rows CH=128 depth 2 write only nt", 6 },
                         { "rows CH=256 depth 2 read only", 7 }, { "rows CH=256 depth 2 write only nt", 8 }, { "rows CH=128 depth 2 r+w plain stores", 9 },
                         { "rows CH=64 depth 4 r+w nt", 10 } };
    const int NC = sizeof(cfgs) / sizeof(cfgs[0]);
    std::vector<std::vector<float>> ms(NC);
    for (int k = 0; k < K; ++k) {
        uint8_t *s, *d; CK(hipMalloc(&s, bytes + 4096)); CK(hipMalloc(&d, bytes + 4096));
        CK(hipMemset(s, 0x5a, bytes));
        Args a{ s, d, q, images, bands, h, pitch, pitch, stride, stride };
        for (int c = 0; c < NC; ++c) {
            float best = 0;
            for (int rep = 0; rep < 4; ++rep) {
                CK(hipMemsetAsync(q, 0, 4, 0));
                CK(hipEventRecord(e0, 0));
                switch (cfgs[c].id) {
                case 0: { const size_t n = bytes / 16; hipLaunchKernelGGL(k_linear, dim3((unsigned)((n + 2047) / 2048)), dim3(256), 0, 0, (const u32x4*)s, (u32x4*)d, n); break; }
                case 1: hipLaunchKernelGGL((k_rows<128, 2, 0, 1>), dim3(cus), dim3(512), 0, 0, a); break;
                case 2: hipLaunchKernelGGL((k_rows<256, 2, 0, 1>), dim3(cus), dim3(512), 0, 0, a); break;
                case 3: hipLaunchKernelGGL((k_rows<512, 1, 0, 1>), dim3(cus), dim3(512), 0, 0, a); break;
                case 4: hipLaunchKernelGGL((k_rows<1024, 1, 0, 1>), dim3(cus), dim3(512), 0, 0, a); break;
                case 5: hipLaunchKernelGGL((k_rows<128, 2, 1, 1>), dim3(cus), dim3(512), 0, 0, a); break;
                case 6: hipLaunchKernelGGL((k_rows<128, 2, 2, 1>), dim3(cus), dim3(512), 0, 0, a); break;
                case 7: hipLaunchKernelGGL((k_rows<256, 2, 1, 1>), dim3(cus), dim3(512), 0, 0, a); break;
                case 8: hipLaunchKernelGGL((k_rows<256, 2, 2, 1>), dim3(cus), dim3(512), 0, 0, a); break;
                case 9: hipLaunchKernelGGL((k_rows<128, 2, 0, 0>), dim3(cus), dim3(512), 0, 0, a); break;
                default: hipLaunchKernelGGL((k_rows<64, 4, 0, 1>), dim3(cus), dim3(512), 0, 0, a); break;
                }
                CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
                float t; CK(hipEventElapsedTime(&t, e0, e1));
                if (rep >= 1) best = best == 0 ? t : std::min(best, t);
            }
            ms[c].push_back(best);
        }
        CK(hipFree(s)); CK(hipFree(d));
    }
    printf("%u images of %ux%u x 4 bytes (pitch %u), %d placements; ms per pass (min of 3), TB/s counts what the pattern moves (read-only / write-only: half)\n", images, w, h, pitch, K);
    for (int c = 0; c < NC; ++c) {
        std::vector<float> t = ms[c]; std::sort(t.begin(), t.end());
        const double moved = (cfgs[c].id >= 5 && cfgs[c].id <= 8 ? 1.0 : 2.0) * (double)bytes;
        printf("%-38s:", cfgs[c].name);
        for (float x : ms[c]) printf(" %6.3f", x);
        printf("   min %.3f (%.2f TB/s) median %.3f max %.3f  spread %.1f %%\n", t.front(), moved / t.front() / 1e9, t[t.size() / 2], t.back(), 100.0 * (t.back() / t.front() - 1));
    }
    return 0;
}
