// flip.hip -- Image.flipHorizontal / flipVerticalPhysical (image.d:1475-1509, 1926-1954) on pixels in HBM.
//
// The reference swaps pixel by pixel (rows byte by byte) in place on the host.  Here every thread owns one such swap: pixel x
// with pixel W - 1 - x of a row (x < W / 2), or a 4-byte piece of row y with the same piece of row H - 1 - y (y < H / 2) -- the
// pairs are disjoint, so the flip is in place and needs no scratch.  Rows are addressed with the image's signed pitch, layers
// with its layer offset (layer / scanline, image.d:235-267).  A copy with reversed addressing: HBM-bound, no arithmetic.
#include "common.hpp"

namespace gamut {
namespace {

template <int PS> struct __attribute__((packed, aligned(1))) Px { uint8_t b[PS]; };

template <int PS>
__global__ __launch_bounds__(256) void k_flip_h(uint8_t* data, int64_t pitch, int64_t layer_off, int w, int h)
{
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y, layer = blockIdx.z;
    if (x >= w / 2) return;
    uint8_t* row = data + layer * layer_off + (int64_t)y * pitch;
    Px<PS>* a = reinterpret_cast<Px<PS>*>(row + (int64_t)x * PS);
    Px<PS>* b = reinterpret_cast<Px<PS>*>(row + (int64_t)(w - 1 - x) * PS);
    const Px<PS> va = *a, vb = *b;
    *a = vb; *b = va;
}

// rows y and H - 1 - y, `unit` bytes per thread (4 when rows and pitch allow dwords, else 1)
template <int UNIT>
__global__ __launch_bounds__(256) void k_flip_v(uint8_t* data, int64_t pitch, int64_t layer_off, int scan_units, int h)
{
    const int u = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y, layer = blockIdx.z;
    if (u >= scan_units) return;
    uint8_t* base = data + layer * layer_off;
    Px<UNIT>* a = reinterpret_cast<Px<UNIT>*>(base + (int64_t)y * pitch) + u;
    Px<UNIT>* b = reinterpret_cast<Px<UNIT>*>(base + (int64_t)(h - 1 - y) * pitch) + u;
    const Px<UNIT> va = *a, vb = *b;
    *a = vb; *b = va;
}

} // namespace

int flip_device(int type, void* data, int64_t pitch, int64_t layer_off, int w, int h, int layers, int vertical, hipStream_t st)
{
    if (!valid_type(type) || w < 0 || h < 0 || layers < 0) return set_error(GAMUT_HIP_ERR_INVALID_ARG, "flip: bad arguments");
    if (w == 0 || h == 0 || layers == 0) return GAMUT_HIP_OK;
    if (!data) return set_error(GAMUT_HIP_ERR_INVALID_ARG, "flip: null pointer");
    if (layers > 65535) return set_error(GAMUT_HIP_ERR_UNSUPPORTED, "flip: more than 65535 layers");
    const int ps = kPixelSize[type];
    uint8_t* d = (uint8_t*)data;
    if (vertical) {
        if (h < 2) return GAMUT_HIP_OK;
        const int64_t scan = (int64_t)w * ps;
        const bool dwords = scan % 4 == 0 && pitch % 4 == 0 && layer_off % 4 == 0 && ((uintptr_t)d & 3) == 0;
        const int units = (int)(dwords ? scan / 4 : scan);
        for (int y0 = 0; y0 < h / 2; y0 += 65535) {                      // gridDim.y <= 65535
            const int rows = (h / 2 - y0) < 65535 ? (h / 2 - y0) : 65535;
            // rows y0 .. y0 + rows - 1 against their mirror rows: shift the base so that the kernel's y = 0 is row y0 and its
            // h - 1 - y is row H - 1 - y0 - y
            uint8_t* base = d + (int64_t)y0 * pitch;
            const dim3 grid((unsigned)((units + 255) / 256), (unsigned)rows, (unsigned)layers);
            if (dwords) hipLaunchKernelGGL(k_flip_v<4>, grid, dim3(256), 0, st, base, pitch, layer_off, units, h - 2 * y0);
            else        hipLaunchKernelGGL(k_flip_v<1>, grid, dim3(256), 0, st, base, pitch, layer_off, units, h - 2 * y0);
        }
        return launch_status("flip_vertical");
    }
    if (w < 2) return GAMUT_HIP_OK;
    for (int y0 = 0; y0 < h; y0 += 65535) {
        const int rows = (h - y0) < 65535 ? (h - y0) : 65535;
        const dim3 grid((unsigned)((w / 2 + 255) / 256), (unsigned)rows, (unsigned)layers);
        uint8_t* base = d + (int64_t)y0 * pitch;
        switch (ps) {
#define GAMUT_FLIP_CASE(N) case N: hipLaunchKernelGGL(k_flip_h<N>, grid, dim3(256), 0, st, base, pitch, layer_off, w, rows); break;
        GAMUT_FLIP_CASE(1) GAMUT_FLIP_CASE(2) GAMUT_FLIP_CASE(3) GAMUT_FLIP_CASE(4) GAMUT_FLIP_CASE(6) GAMUT_FLIP_CASE(8) GAMUT_FLIP_CASE(12) GAMUT_FLIP_CASE(16)
#undef GAMUT_FLIP_CASE
        default: return set_error(GAMUT_HIP_ERR_UNSUPPORTED, "flip: pixel size %d", ps);
        }
    }
    return launch_status("flip_horizontal");
}

} // namespace gamut

using namespace gamut;

extern "C" {

int gamut_hip_flip_device(int type, void* data, int64_t pitch, int64_t layerOffset, int width, int height, int layers, int vertical, void* stream)
{
    clear_error();
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1)
        return set_error(GAMUT_HIP_ERR_NO_DEVICE, "no HIP device available (libgamut_hip has no CPU fallback)");
    return flip_device(type, data, pitch, layerOffset, width, height, layers, vertical, pick_stream(stream));
}

// host rows (signed pitch, like Image.scanline): up, flip, down.  Synchronous.
int gamut_hip_flip(int type, uint8_t* data, int pitch, int width, int height, int vertical)
{
    clear_error();
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1)
        return set_error(GAMUT_HIP_ERR_NO_DEVICE, "no HIP device available (libgamut_hip has no CPU fallback)");
    if (!valid_type(type) || width < 0 || height < 0) return set_error(GAMUT_HIP_ERR_INVALID_ARG, "flip: bad arguments");
    if (width == 0 || height == 0) return GAMUT_HIP_OK;
    if (!data) return set_error(GAMUT_HIP_ERR_INVALID_ARG, "flip: null pointer");
    const size_t row = (size_t)width * kPixelSize[type], apitch = (size_t)(pitch < 0 ? -(int64_t)pitch : pitch);
    if (apitch < row && height > 1) return set_error(GAMUT_HIP_ERR_INVALID_ARG, "flip: overlapping scanlines");
    static thread_local PerDevice<DeviceScratch> scratch_pd;
    hipStream_t st = thread_stream();
    uint8_t* d = (uint8_t*)scratch_pd.cur().get(row * (size_t)height + 16, st);
    if (!d) return set_error(GAMUT_HIP_ERR_OUT_OF_MEMORY, "flip: device staging of %zu bytes failed", row * (size_t)height);
    uint8_t* low = pitch < 0 ? data + (int64_t)pitch * (height - 1) : data;          // lowest address; device rows keep the memory order
    GAMUT_HIP_CHECK(hipMemcpy2DAsync(d, row, low, height > 1 ? apitch : row, row, (size_t)height, hipMemcpyHostToDevice, st));
    if (int rc = flip_device(type, d, (int64_t)row, 0, width, height, 1, vertical, st)) return rc;
    GAMUT_HIP_CHECK(hipMemcpy2DAsync(low, height > 1 ? apitch : row, d, row, row, (size_t)height, hipMemcpyDeviceToHost, st));
    GAMUT_HIP_CHECK(hipStreamSynchronize(st));
    return GAMUT_HIP_OK;
}

} // extern "C"
