// stream_host.hip -- the callback-shaped drop-ins: the same arguments as the reference's codec entry points, so the swap
// in plugins/jpeg.d / plugins/png.d is argument for argument.
//
//   decompress_jpeg_image_from_stream(JpegStreamReadFunc, void* user, ...)     jpegload.d:61-70, 3720-3723
//   stbi_load_from_callbacks / stbi_load_16_from_callbacks(stbi_io_callbacks*)  stbdec.d:408-419, 713-735
//   stbi__png_is16 on a callback context                                        stbdec.d:2091-2109 (plugins/png.d:50-62)
//
// The GPU path wants the whole compressed file (entropy decode / inflate are per-file host stages, the kernels start from
// dense coefficients / the inflated stream), so these read the stream to its end through the caller's callbacks -- the way
// the reference's own decoders pull their input, in the same call pattern -- and hand the bytes to the *_from_memory entry
// points.  Host only; no device work of their own.
#include "common.hpp"

namespace gamut {
namespace {

struct Slurp {
    uint8_t* p = nullptr; size_t len = 0, cap = 0;
    ~Slurp() { free(p); }
    bool room(size_t more)
    {
        if (len + more <= cap) return true;
        size_t want = cap ? cap * 2 : (size_t)1 << 16;
        while (want < len + more) want *= 2;
        uint8_t* q = (uint8_t*)realloc(p, want);
        if (!q) return false;
        p = q; cap = want; return true;
    }
};
constexpr size_t kMaxFile = (size_t)1 << 32;           // a compressed file this long is not an image the path can hold anyway

// jpgd's prep_in_buffer (:1980-2003) calls the read function until its 8 KiB buffer is full or *pEOF_flag is set; -1 is an
// error (stop_decoding(JPGD_STREAM_READ)).  Same loop, bigger buffer.
bool slurp_jpeg(gamut_hip_jpeg_stream_read_func rd, void* user, Slurp& s)
{
    const int chunk = 1 << 16;
    int idle = 0;
    for (;;) {
        if (s.len > kMaxFile || !s.room((size_t)chunk)) { set_error(GAMUT_HIP_ERR_OUT_OF_MEMORY, "jpeg: stream too long"); return false; }
        unsigned char eof = 0;
        const int n = rd(s.p + s.len, chunk, &eof, user);
        if (n < 0 || n > chunk) { set_error(GAMUT_HIP_ERR_DECODE, "jpeg: stream read error"); return false; }
        s.len += (size_t)n;
        if (eof) return true;
        // a source that returns 0 bytes without ever raising the flag would spin the reference's loop for ever: give up instead
        idle = n == 0 ? idle + 1 : 0;
        if (idle >= 64) return true;
    }
}

// stb's refill (stbi__refill_buffer :754-770) treats a read of 0 bytes as the end of the data
bool slurp_stb(const gamut_hip_stbi_io_callbacks* c, void* user, Slurp& s, size_t limit = kMaxFile)
{
    const int chunk = 1 << 16;
    while (s.len < limit) {
        const size_t want = limit - s.len < (size_t)chunk ? limit - s.len : (size_t)chunk;
        if (!s.room(want)) { set_error(GAMUT_HIP_ERR_OUT_OF_MEMORY, "png: stream too long"); return false; }
        const int n = c->read(user, (char*)(s.p + s.len), (int)want);
        if (n <= 0) break;
        s.len += (size_t)(n > (int)want ? (int)want : n);
    }
    return true;
}

} // namespace
} // namespace gamut

using namespace gamut;

extern "C" {

uint8_t* gamut_hip_decompress_jpeg_image_from_stream(gamut_hip_jpeg_stream_read_func rfn, void* userData,
        int* width, int* height, int* actual_comps, float* pixelAspectRatio, float* dotsPerInchY, int req_comps)
{
    clear_error();
    if (!rfn || !width || !height || !actual_comps) { set_error(GAMUT_HIP_ERR_INVALID_ARG, "jpeg: null argument"); return nullptr; }   // :3725-3731
    Slurp s;
    if (!slurp_jpeg(rfn, userData, s)) return nullptr;
    float par = -1, dpi = -1;
    uint8_t* px = gamut_hip_decompress_jpeg_image_from_memory(s.p, s.len, width, height, actual_comps, &par, &dpi, req_comps);
    if (pixelAspectRatio) *pixelAspectRatio = par;
    if (dotsPerInchY) *dotsPerInchY = dpi;
    return px;
}

uint8_t* gamut_hip_stbi_load_from_callbacks(const gamut_hip_stbi_io_callbacks* clbk, void* user, int* x, int* y, int* comp, int req_comp,
                                            float* ppmX, float* ppmY, float* pixelRatio)
{
    clear_error();
    if (!clbk || !clbk->read) { set_error(GAMUT_HIP_ERR_INVALID_ARG, "png: null callbacks"); return nullptr; }
    Slurp s;
    if (!slurp_stb(clbk, user, s)) return nullptr;
    return gamut_hip_stbi_load_from_memory(s.p, s.len, x, y, comp, req_comp, ppmX, ppmY, pixelRatio);
}

uint16_t* gamut_hip_stbi_load_16_from_callbacks(const gamut_hip_stbi_io_callbacks* clbk, void* user, int* x, int* y, int* comp, int req_comp,
                                                float* ppmX, float* ppmY, float* pixelRatio)
{
    clear_error();
    if (!clbk || !clbk->read) { set_error(GAMUT_HIP_ERR_INVALID_ARG, "png: null callbacks"); return nullptr; }
    Slurp s;
    if (!slurp_stb(clbk, user, s)) return nullptr;
    return gamut_hip_stbi_load_16_from_memory(s.p, s.len, x, y, comp, req_comp, ppmX, ppmY, pixelRatio);
}

int gamut_hip_stbi_png_is16_from_callbacks(const gamut_hip_stbi_io_callbacks* clbk, void* user)
{
    // stbi__png_is16 parses the header only (signature + IHDR = 33 bytes); the caller rewinds its stream afterwards, as
    // plugins/png.d:50-62 does
    if (!clbk || !clbk->read) return 0;
    Slurp s;
    if (!slurp_stb(clbk, user, s, 64)) return 0;
    return gamut_hip_png_is16(s.p, s.len);
}

} // extern "C"
