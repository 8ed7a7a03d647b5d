// Host -> device copy rate from page-locked memory with 1, 2, 4 concurrent streams (is one copy stream all a PCIe link gives?)
//   hipcc --offload-arch=gfx950 -O2 tools/microbench/h2d_streams.hip -o /tmp/h2d && /tmp/h2d
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
int main()
{
    const size_t total = (size_t)2 << 30;
    char* h = nullptr; char* d = nullptr;
    if (hipHostMalloc((void**)&h, total, hipHostMallocDefault) != hipSuccess || hipMalloc((void**)&d, total) != hipSuccess) { printf("alloc failed\n"); return 1; }
    memset(h, 1, total);
    hipStream_t st[8];
    for (auto& s : st) (void)hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    setvbuf(stdout, nullptr, _IOLBF, 0);
    for (size_t piece : { (size_t)1 << 20, (size_t)8 << 20, (size_t)64 << 20 })
        for (int ns : { 1, 2, 4, 8 }) {
            double best = 1e9;
            for (int rep = 0; rep < 3; ++rep) {
                (void)hipDeviceSynchronize();
                const auto t0 = std::chrono::steady_clock::now();
                size_t k = 0;
                for (size_t off = 0; off < total; off += piece, ++k) (void)hipMemcpyAsync(d + off, h + off, piece, hipMemcpyHostToDevice, st[k % (size_t)ns]);
                (void)hipDeviceSynchronize();
                const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
                if (s < best) best = s;
            }
            printf("pieces of %4zu MiB on %d stream(s): %6.1f GB/s\n", piece >> 20, ns, (double)total / best / 1e9);
        }
    // the pipelines' shape: slice r of every file (files 8 MiB apart) -- as 256 separate 1 MiB copies, or as ONE pitched copy
    for (size_t width : { (size_t)1 << 20, (size_t)512 << 10 }) {
        const size_t pitch = (size_t)8 << 20, rows = total / pitch;
        double best = 1e9;
        for (int rep = 0; rep < 3; ++rep) {
            (void)hipDeviceSynchronize();
            const auto t0 = std::chrono::steady_clock::now();
            for (size_t r = 0; r * width < pitch; ++r) (void)hipMemcpy2DAsync(d + r * width, pitch, h + r * width, pitch, width, rows, hipMemcpyHostToDevice, st[0]);
            (void)hipDeviceSynchronize();
            const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            if (s < best) best = s;
        }
        printf("pitched copies, %zu rows of %4zu KiB, pitch 8 MiB, one stream: %6.1f GB/s\n", rows, width >> 10, (double)total / best / 1e9);
    }
    return 0;
}
