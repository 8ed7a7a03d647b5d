// png_host.hip -- host side of the PNG path and its C-ABI entry points.
//
// Host feeder = what stays on the CPU in the reference too:
// the chunk walk of stbi__parse_png_file (stbdec.d:1777-2023: IHDR/PLTE/tRNS/pHYs/CgBI/IDAT,
// missing-IEND tolerance :2008-2012) and the inflate of finalize_decode (:1808-1819; the
// reference calls the third-party `miniz`, here the system zlib in raw mode with the adler32
// unchecked like the reference's trusted_input=true).  Everything after the inflate runs on
// the GPU: de-filter / expand (png.hip), Adam7 scatter, tRNS, palette, channel and depth
// conversion, in the order of finalize_decode (:1821-1857) and stbi__do_png (:2025-2055).
#include "common.hpp"
#include <algorithm>
#include <atomic>
#include <chrono>
#include <map>
#include <mutex>
#include <thread>
#include <tuple>
#include <vector>
#include <zlib.h>

namespace gamut {
namespace {

struct Reader {                       // stb's memory reader: reads past the end yield 0 (stbi__get8)
    const uint8_t* p; const uint8_t* end;
    int      get8()    { return p < end ? *p++ : 0; }
    uint32_t get16be() { uint32_t z = (uint32_t)get8(); return (z << 8) + (uint32_t)get8(); }
    uint32_t get32be() { uint32_t z = get16be(); return (z << 16) + get16be(); }
    bool     eof() const { return p >= end; }
    void     skip(uint32_t n) { if ((size_t)(end - p) < n) p = end; else p += n; }
};

struct PngHeader {
    uint32_t x = 0, y = 0; int depth = 0, color = 0, interlace = 0, img_n = 0, pal_img_n = 0;
    bool has_trans = false, is_iphone = false;
    uint8_t palette[1024] = {}; uint32_t pal_len = 0;     // zero-initialised like the D array (stbdec.d:1779): indices >= pal_len expand to (0,0,0,0)
    uint16_t tc[3] = { 0, 0, 0 };      // tRNS key: 8-bit values already scaled (stbdec.d:1945), 16-bit as-is (:1941)
    float ppmX = -1, ppmY = -1, aspect = -1;
    uint8_t* idata = nullptr; uint32_t ioff = 0;
    // the batch path that inflates on the GPU gathers the IDAT payloads itself, slice by slice: parse() then only notes where they are
    std::vector<std::pair<const uint8_t*, uint32_t>>* idat_segments = nullptr;
    bool seen_idat = false;
    ~PngHeader() { free(idata); }
};

const uint8_t kDepthScale[9] = { 0, 0xff, 0x55, 0, 0x11, 0, 0, 0, 0x01 };
constexpr uint32_t fourcc(char a, char b, char c, char d) { return ((uint32_t)(uint8_t)a << 24) | ((uint32_t)(uint8_t)b << 16) | ((uint32_t)(uint8_t)c << 8) | (uint8_t)d; }

// header_only: stop as stbi__png_info_raw does (SCAN_header), enough for stbi__png_is16
int parse(const uint8_t* data, size_t len, PngHeader& h, bool header_only)
{
    static const uint8_t sig[8] = { 137, 80, 78, 71, 13, 10, 26, 10 };
    Reader s{ data, data + len };
    for (int i = 0; i < 8; ++i) if (s.get8() != sig[i]) return set_error(GAMUT_HIP_ERR_DECODE, "png: bad signature");
    bool first = true; uint32_t cap = 0;
    for (;;) {
        const uint32_t clen = s.get32be(), type = s.get32be();
        switch (type) {
        case fourcc('C','g','B','I'): h.is_iphone = true; s.skip(clen); break;
        case fourcc('p','H','Y','s'):
            h.ppmX = (float)s.get32be(); h.ppmY = (float)s.get32be(); h.aspect = h.ppmX / h.ppmY;
            if (s.get8() != 1) { h.ppmX = -1; h.ppmY = -1; }
            break;
        case fourcc('I','H','D','R'): {
            if (!first || clen != 13) return set_error(GAMUT_HIP_ERR_DECODE, "png: bad IHDR");
            first = false;
            h.x = s.get32be(); h.y = s.get32be();
            if (h.y > (1u << 24) || h.x > (1u << 24)) return set_error(GAMUT_HIP_ERR_DECODE, "png: too large");
            h.depth = s.get8();
            if (h.depth != 1 && h.depth != 2 && h.depth != 4 && h.depth != 8 && h.depth != 16) return set_error(GAMUT_HIP_ERR_DECODE, "png: 1/2/4/8/16-bit only");
            h.color = s.get8();
            if (h.color > 6 || (h.color == 3 && h.depth == 16)) return set_error(GAMUT_HIP_ERR_DECODE, "png: bad ctype");
            if (h.color == 3) h.pal_img_n = 3; else if (h.color & 1) return set_error(GAMUT_HIP_ERR_DECODE, "png: bad ctype");
            if (s.get8()) return set_error(GAMUT_HIP_ERR_DECODE, "png: bad comp method");
            if (s.get8()) return set_error(GAMUT_HIP_ERR_DECODE, "png: bad filter method");
            h.interlace = s.get8(); if (h.interlace > 1) return set_error(GAMUT_HIP_ERR_DECODE, "png: bad interlace method");
            if (!h.x || !h.y) return set_error(GAMUT_HIP_ERR_DECODE, "png: 0-pixel image");
            if (!h.pal_img_n) {
                h.img_n = (h.color & 2 ? 3 : 1) + (h.color & 4 ? 1 : 0);
                if ((1u << 30) / h.x / (uint32_t)h.img_n < h.y) return set_error(GAMUT_HIP_ERR_DECODE, "png: too large");
                if (header_only) return GAMUT_HIP_OK;
            } else {
                h.img_n = 1;
                if ((1u << 30) / h.x / 4 < h.y) return set_error(GAMUT_HIP_ERR_DECODE, "png: too large");
            }
        } break;
        case fourcc('P','L','T','E'):
            if (first || clen > 256 * 3) return set_error(GAMUT_HIP_ERR_DECODE, "png: invalid PLTE");
            h.pal_len = clen / 3;
            if (h.pal_len * 3 != clen) return set_error(GAMUT_HIP_ERR_DECODE, "png: invalid PLTE");
            for (uint32_t i = 0; i < h.pal_len; ++i) {
                h.palette[i*4] = (uint8_t)s.get8(); h.palette[i*4+1] = (uint8_t)s.get8(); h.palette[i*4+2] = (uint8_t)s.get8(); h.palette[i*4+3] = 255;
            }
            break;
        case fourcc('t','R','N','S'):
            if (first) return set_error(GAMUT_HIP_ERR_DECODE, "png: first not IHDR");
            if (h.idata || h.seen_idat) return set_error(GAMUT_HIP_ERR_DECODE, "png: tRNS after IDAT");
            if (h.pal_img_n) {
                if (header_only) { h.img_n = 4; return GAMUT_HIP_OK; }
                if (h.pal_len == 0 || clen > h.pal_len) return set_error(GAMUT_HIP_ERR_DECODE, "png: bad tRNS");
                h.pal_img_n = 4;
                for (uint32_t i = 0; i < clen; ++i) h.palette[i*4+3] = (uint8_t)s.get8();
            } else {
                if (!(h.img_n & 1) || clen != (uint32_t)h.img_n * 2) return set_error(GAMUT_HIP_ERR_DECODE, "png: bad tRNS");
                h.has_trans = true;
                for (int k = 0; k < h.img_n; ++k) {
                    const uint32_t v = s.get16be();
                    h.tc[k] = h.depth == 16 ? (uint16_t)v : (uint16_t)(uint8_t)((uint8_t)(v & 255) * kDepthScale[h.depth]);
                }
            }
            break;
        case fourcc('I','D','A','T'): {
            if (first) return set_error(GAMUT_HIP_ERR_DECODE, "png: first not IHDR");
            if (h.pal_img_n && !h.pal_len) return set_error(GAMUT_HIP_ERR_DECODE, "png: no PLTE");
            if (header_only) { h.img_n = h.pal_img_n ? h.pal_img_n : h.img_n; return GAMUT_HIP_OK; }
            if ((int32_t)(h.ioff + clen) < (int32_t)h.ioff) return set_error(GAMUT_HIP_ERR_DECODE, "png: IDAT overflow");
            h.seen_idat = true;
            if (h.idat_segments) {
                if ((size_t)(s.end - s.p) < clen) return set_error(GAMUT_HIP_ERR_DECODE, "png: out of data");
                if (clen) h.idat_segments->emplace_back(s.p, clen);
                s.p += clen; h.ioff += clen;
                break;
            }
            if (h.ioff + clen > cap) {
                uint32_t ncap = cap ? cap : (clen > 4096 ? clen : 4096);
                while (h.ioff + clen > ncap) ncap *= 2;
                uint8_t* n = (uint8_t*)realloc(h.idata, ncap);
                if (!n) return set_error(GAMUT_HIP_ERR_OUT_OF_MEMORY, "png: out of memory");
                h.idata = n; cap = ncap;
            }
            if ((size_t)(s.end - s.p) < clen) return set_error(GAMUT_HIP_ERR_DECODE, "png: out of data");
            memcpy(h.idata + h.ioff, s.p, clen); s.p += clen; h.ioff += clen;
        } break;
        case fourcc('I','E','N','D'):
            if (first) return set_error(GAMUT_HIP_ERR_DECODE, "png: first not IHDR");
            return GAMUT_HIP_OK;
        default:
            if (first) return set_error(GAMUT_HIP_ERR_DECODE, "png: first not IHDR");
            if (type == 0 && s.eof()) return GAMUT_HIP_OK;                 // Gamut issue #92: no IEND
            if ((type & (1u << 29)) == 0) return set_error(GAMUT_HIP_ERR_DECODE, "png: unknown critical chunk");
            s.skip(clen);
            break;
        }
        s.get32be();      // CRC, not checked
    }
}

// Where zlib writes the inflated stream.  Page-locked (the per-thread staging of the single-image calls: the upload is
// then a plain DMA -- an upload from freshly written pageable memory was seen to take 12-22 ms for 4 MB on some boxes,
// 0.2 ms on others; intentionally not freed at thread exit, the HIP runtime may already be gone) or plain malloc memory
// (the short-lived worker threads of the batch call).
struct HostBuf {
    uint8_t* p = nullptr; size_t cap = 0; bool pinned = false;
    bool reserve(size_t n, size_t keep)
    {
        if (n <= cap) return true;
        const size_t ncap = n + n / 8 + 4096;
        if (pinned) {
            void* np = nullptr;
            if (hipHostMalloc(&np, ncap, hipHostMallocDefault) != hipSuccess) return false;
            if (keep) memcpy(np, p, keep);
            if (p) (void)hipHostFree(p);
            p = (uint8_t*)np;
        } else {
            uint8_t* np = (uint8_t*)realloc(p, ncap);
            if (!np) return false;
            p = np;
        }
        cap = ncap;
        return true;
    }
    void release() { if (p) { if (pinned) (void)hipHostFree(p); else free(p); } p = nullptr; cap = 0; }
};

// stbi_zlib_decode_malloc_guesssize_headerflag (stbdec.d:1267-1321); the result lives in `out` (not to be freed)
uint8_t* inflate_idat(const uint8_t* buf, uint32_t len, size_t guess, uint32_t* outlen, bool parse_header, HostBuf& out)
{
    if (parse_header) {
        if (len < 2 || ((buf[0] * 256 + buf[1]) % 31) != 0 || (buf[1] & 32) || (buf[0] & 15) != 8) { set_error(GAMUT_HIP_ERR_DECODE, "png: bad zlib header"); return nullptr; }
        buf += 2; len -= 2;
    }
    size_t cap = guess ? guess : 1;
    z_stream z; memset(&z, 0, sizeof(z));
    if (!out.reserve(cap, 0) || inflateInit2(&z, -15) != Z_OK) { set_error(GAMUT_HIP_ERR_OUT_OF_MEMORY, "png: inflate init failed"); return nullptr; }
    z.next_in = const_cast<Bytef*>(buf); z.avail_in = len; z.next_out = out.p; z.avail_out = (uInt)cap;
    for (;;) {
        const int r = inflate(&z, Z_NO_FLUSH);
        if (r == Z_STREAM_END) break;
        if ((r == Z_OK || r == Z_BUF_ERROR) && z.avail_out == 0 && cap <= 536870912u) {
            size_t ncap = cap * 2; if (ncap < 32 * 1024) ncap = 32 * 1024;
            if (!out.reserve(ncap, cap)) { inflateEnd(&z); set_error(GAMUT_HIP_ERR_OUT_OF_MEMORY, "png: out of memory"); return nullptr; }
            z.next_out = out.p + cap; z.avail_out = (uInt)(ncap - cap); cap = ncap;
            continue;
        }
        if (r == Z_OK && z.avail_in != 0) continue;
        inflateEnd(&z); set_error(GAMUT_HIP_ERR_DECODE, "png: corrupt zlib stream"); return nullptr;
    }
    *outlen = (uint32_t)z.total_out;
    inflateEnd(&z);
    return out.p;
}

struct Dev {
    void* p = nullptr;
    ~Dev() { if (p) (void)hipFree(p); }
    bool alloc(size_t n) { if (hipMalloc(&p, n ? n : 1) != hipSuccess) { p = nullptr; set_error(GAMUT_HIP_ERR_OUT_OF_MEMORY, "png: hipMalloc(%zu) failed", n); return false; } return true; }
    void swap(Dev& o) { void* t = p; p = o.p; o.p = t; }
};

// parse + inflate of one file (host only)
struct PngJob { PngHeader h; HostBuf raw; uint32_t raw_len = 0; int rc = GAMUT_HIP_OK; char msg[160] = { 0 }; };

int png_prepare(const uint8_t* data, size_t len, PngJob& j)
{
    if (parse(data, len, j.h, false)) return j.rc = GAMUT_HIP_ERR_DECODE;
    if (!j.h.idata) return j.rc = set_error(GAMUT_HIP_ERR_DECODE, "png: no IDAT");
    const uint32_t bpl = (j.h.x * (uint32_t)j.h.depth + 7) / 8;
    if (!inflate_idat(j.h.idata, j.h.ioff, (size_t)bpl * j.h.y * j.h.img_n + j.h.y, &j.raw_len, !j.h.is_iphone, j.raw)) return j.rc = GAMUT_HIP_ERR_DECODE;
    return j.rc = GAMUT_HIP_OK;
}

// The whole of stbi__do_png after the inflate, on the GPU: de-filter (+ Adam7), tRNS, palette, channel-count conversion
// (stbdec.d:1646-1679, 1821-1855, 2038-2045), then the 16 <-> 8 step of stbi__load_and_postprocess_* (:669-707) when
// want_bits (8 / 16; 0 = as decoded) asks for it.  The result is left at d_dst (device) or returned as malloc'd host memory.
uint8_t* png_run(const PngHeader& h, const uint8_t* raw, uint32_t raw_len, int req_comp, int want_bits, uint8_t* d_dst,
                 int* px, int* py, int* pn, int* bits_out, hipStream_t st, const uint8_t* d_raw = nullptr)
{
    const bool trace = getenv("GAMUT_HIP_TRACE") != nullptr;               // stage timings on stderr
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
    int img_n = h.img_n, out_n;
    if ((req_comp == img_n + 1 && req_comp != 3 && !h.pal_img_n) || h.has_trans) out_n = img_n + 1; else out_n = img_n;   // :1821-1824
    int bytes = h.depth == 16 ? 2 : 1;
    const int64_t npx = (int64_t)h.x * h.y;

    const auto t1 = now();
    Dev draw, dimg, dstatus;
    if ((!d_raw && !draw.alloc((size_t)raw_len + 16)) || !dimg.alloc((size_t)npx * out_n * bytes + 16) || !dstatus.alloc(4)) return nullptr;
    const auto t2 = now();
    if ((!d_raw && hipMemcpyAsync(draw.p, raw, raw_len, hipMemcpyHostToDevice, st) != hipSuccess) || hipMemsetAsync(dstatus.p, 0, 4, st) != hipSuccess) {
        set_error(GAMUT_HIP_ERR_HIP, "png: upload failed"); return nullptr;
    }
    const uint8_t* const stream_dev = d_raw ? d_raw : (const uint8_t*)draw.p;      // the inflated stream in HBM
    if (trace) (void)hipStreamSynchronize(st);
    const auto t3 = now();
    if (!h.interlace) {
        if (png_defilter_launch(stream_dev, 0, raw_len, (uint8_t*)dimg.p, 0, h.x, h.y, img_n, out_n, h.depth, h.color, 1, (uint32_t*)dstatus.p, st)) return nullptr;
    } else {                                                              // stbi__create_png_image :1646-1679
        static const int xorig[7] = { 0,4,0,2,0,1,0 }, yorig[7] = { 0,0,4,0,2,0,1 }, xspc[7] = { 8,8,4,4,2,2,1 }, yspc[7] = { 8,8,8,4,4,2,2 };
        Dev dpass;
        if (!dpass.alloc((size_t)npx * out_n * bytes + 16)) return nullptr;
        const uint8_t* rp = stream_dev; uint32_t left = raw_len;
        for (int p = 0; p < 7; ++p) {
            const uint32_t x = (h.x - xorig[p] + xspc[p] - 1) / xspc[p], y = (h.y - yorig[p] + yspc[p] - 1) / yspc[p];
            if (!x || !y) continue;
            const uint32_t img_len = ((((uint32_t)img_n * x * h.depth) + 7) >> 3) * y + y;
            if (png_defilter_launch(rp, 0, left, (uint8_t*)dpass.p, 0, x, y, img_n, out_n, h.depth, h.color, 1, (uint32_t*)dstatus.p, st)) return nullptr;
            if (png_adam7_scatter_launch((const uint8_t*)dpass.p, (uint8_t*)dimg.p, x, y, h.x, out_n * bytes, p, st)) return nullptr;
            rp += img_len; left -= img_len;
        }
        if (hipStreamSynchronize(st) != hipSuccess) { set_error(GAMUT_HIP_ERR_HIP, "png: sync failed"); return nullptr; }   // dpass lifetime
    }
    if (h.has_trans && png_transparency_launch(dimg.p, npx, out_n, h.depth == 16, h.tc, st)) return nullptr;      // :1829-1841
    if (h.pal_img_n) {                                                      // :1843-1851
        img_n = h.pal_img_n; out_n = h.pal_img_n;
        if (req_comp >= 3) out_n = req_comp;
        Dev dpal, dexp;
        if (!dpal.alloc(1024) || !dexp.alloc((size_t)npx * out_n + 16)) return nullptr;
        if (hipMemcpyAsync(dpal.p, h.palette, 1024, hipMemcpyHostToDevice, st) != hipSuccess) { set_error(GAMUT_HIP_ERR_HIP, "png: upload failed"); return nullptr; }
        if (png_palette_launch((const uint8_t*)dimg.p, (uint8_t*)dexp.p, npx, out_n, (const uint8_t*)dpal.p, st)) return nullptr;
        if (hipStreamSynchronize(st) != hipSuccess) { set_error(GAMUT_HIP_ERR_HIP, "png: sync failed"); return nullptr; }   // dpal lifetime
        dimg.swap(dexp);
    } else if (h.has_trans) ++img_n;                                        // :1852-1855
    if (req_comp && req_comp != out_n) {                                    // stbi__do_png :2038-2045
        Dev dconv;
        if (!dconv.alloc((size_t)npx * req_comp * bytes + 16)) return nullptr;
        if (png_convert_format_launch(dimg.p, dconv.p, npx, out_n, req_comp, bytes == 2, st)) return nullptr;
        if (hipStreamSynchronize(st) != hipSuccess) { set_error(GAMUT_HIP_ERR_HIP, "png: sync failed"); return nullptr; }
        dimg.swap(dconv);
        out_n = req_comp;
    }
    int bits = h.depth <= 8 ? 8 : 16;
    if (want_bits && want_bits != bits) {                                   // stbi__convert_16_to_8 / _8_to_16 :635-666
        Dev ddepth;
        const int64_t samples = npx * out_n;
        if (!ddepth.alloc((size_t)samples * (want_bits / 8) + 16)) return nullptr;
        if (png_depth_convert_launch(dimg.p, ddepth.p, samples, want_bits == 16, st)) return nullptr;
        if (hipStreamSynchronize(st) != hipSuccess) { set_error(GAMUT_HIP_ERR_HIP, "png: sync failed"); return nullptr; }
        dimg.swap(ddepth);
        bits = want_bits; bytes = want_bits / 8;
    }
    if (trace) (void)hipStreamSynchronize(st);
    const auto t4 = now();
    const size_t out_bytes = (size_t)npx * out_n * bytes;
    uint8_t* result = d_dst ? d_dst : (uint8_t*)malloc(out_bytes ? out_bytes : 1);
    uint32_t status = 0;
    if (!result) { set_error(GAMUT_HIP_ERR_OUT_OF_MEMORY, "png: out of memory"); return nullptr; }
    if (hipMemcpyAsync(result, dimg.p, out_bytes, d_dst ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, st) != hipSuccess ||
        hipMemcpyAsync(&status, dstatus.p, 4, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) {
        if (!d_dst) free(result);
        set_error(GAMUT_HIP_ERR_HIP, "png: download failed: %s", hipGetErrorString(hipGetLastError())); return nullptr;
    }
    if (trace) fprintf(stderr, "[gamut_hip] png %ux%u: hipMalloc %.2f ms, upload %.2f, kernels %.2f, %s %.2f\n",
                       h.x, h.y, ms(t1, t2), ms(t2, t3), ms(t3, t4), d_dst ? "device copy" : "download", ms(t4, now()));
    if (status) { if (!d_dst) free(result); set_error(GAMUT_HIP_ERR_DECODE, "png: invalid filter"); return nullptr; }
    *px = (int)h.x; *py = (int)h.y; if (pn) *pn = img_n;
    *bits_out = bits;
    return result;
}

// stbi_load_from_memory / stbi_load_16_from_memory on one file: malloc'd host pixels of want_bits bits per sample
uint8_t* png_load(const uint8_t* data, size_t len, int* px, int* py, int* pn, int req_comp, int want_bits,
                  float* ppmX, float* ppmY, float* aspect)
{
    if (req_comp < 0 || req_comp > 4) { set_error(GAMUT_HIP_ERR_INVALID_ARG, "png: bad req_comp"); return nullptr; }
    int dev_count = 0;                                                       // before any work: no GPU, no result
    PngHeader probe;
    if (parse(data, len, probe, true)) return nullptr;                       // a bad signature is reported as such with or without a GPU
    if (hipGetDeviceCount(&dev_count) != hipSuccess || dev_count <= 0) { set_error(GAMUT_HIP_ERR_NO_DEVICE, "no HIP device available (libgamut_hip has no CPU fallback)"); return nullptr; }
    struct Staging { HostBuf b{ nullptr, 0, true }; };
    static thread_local PerDevice<Staging> staging_pd;
    HostBuf& staging = staging_pd.cur().b;
    PngJob j; j.raw = staging;                                               // borrow the per-thread pinned buffer
    const auto t0 = std::chrono::steady_clock::now();
    const int rc = png_prepare(data, len, j);
    staging = j.raw; j.raw = HostBuf{};                                      // hand it back (it may have grown)
    if (rc != GAMUT_HIP_OK) return nullptr;
    if (getenv("GAMUT_HIP_TRACE")) fprintf(stderr, "[gamut_hip] png parse + inflate %.2f ms\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
    if (ppmX) *ppmX = j.h.ppmX;
    if (ppmY) *ppmY = j.h.ppmY;
    if (aspect) *aspect = j.h.aspect;
    int bits = 8;
    return png_run(j.h, staging.p, j.raw_len, req_comp, want_bits, nullptr, px, py, pn, &bits, thread_stream());
}

// Pinned inflate buffers of the batch workers: taken for the duration of a call, kept for the next one (page-locking 30 MB
// costs milliseconds), never freed -- the HIP runtime may be gone at exit.
struct PinnedPool {
    std::mutex m; std::vector<HostBuf> idle;
    HostBuf take() { std::lock_guard<std::mutex> g(m); if (idle.empty()) return HostBuf{ nullptr, 0, true }; HostBuf b = idle.back(); idle.pop_back(); return b; }
    void give(const HostBuf& b) { std::lock_guard<std::mutex> g(m); idle.push_back(b); }
};
PinnedPool& pinned_pool() { static PinnedPool* p = new PinnedPool(); return *p; }

// Batches with more files than host threads inflate on the GPU (one workgroup per stream): a stream takes the kernel 0.8-3 times as long
// as it takes zlib on one host core, but up to 256 of them run side by side -- the host threads need a second round of files from
// threads + 1 files on (16 threads, 4K files: 16 files 64 / 40 ms on the host against 75 / 52 ms on the device, 24 files 112 / 67 against 75 / 51).

struct BatchFile {                    // what a worker leaves behind for one file
    int rc = GAMUT_HIP_OK; char msg[160] = { 0 };
    bool uploaded = false, batched = false;   // inflated stream in the device arena; de-filter goes into a batched launch
    PngHeader h;                      // (idata released)
    uint32_t raw_len = 0, need = 0; int out_n = 0, bits = 8, channels_in_file = 0;
};

} // namespace
} // namespace gamut

using namespace gamut;

extern "C" {

int gamut_hip_png_defilter_batch_device(const uint8_t* raw, int64_t raw_stride, uint32_t raw_len,
                                        uint8_t* out, int64_t out_stride,
                                        uint32_t x, uint32_t y, int img_n, int out_n, int depth, int color,
                                        int count, uint32_t* status, void* stream)
{
    clear_error();
    return png_defilter_launch(raw, raw_stride, raw_len, out, out_stride, x, y, img_n, out_n, depth, color, count, status, pick_stream(stream));
}

int gamut_hip_png_defilter_device(const gamut_hip_png_desc* descs, int count, uint32_t* status, void* stream)
{
    clear_error();
    if (count < 0 || (count > 0 && !descs)) return set_error(GAMUT_HIP_ERR_INVALID_ARG, "png_defilter: bad descriptor array");
    for (int i = 0; i < count; ++i) {
        const gamut_hip_png_desc& d = descs[i];
        if (int rc = png_defilter_launch(d.raw, 0, d.raw_len, d.out, 0, d.x, d.y, d.img_n, d.out_n, d.depth, d.color, 1,
                                         status ? status + i : nullptr, pick_stream(stream)))
            return rc;
    }
    return GAMUT_HIP_OK;
}

uint8_t* gamut_hip_stbi_load_from_memory(const uint8_t* data, size_t len, int* x, int* y, int* comp, int req_comp,
                                         float* ppmX, float* ppmY, float* pixelRatio)
{
    clear_error();
    int n = 0, w = 0, h = 0;
    uint8_t* r = png_load(data, len, &w, &h, &n, req_comp, 8, ppmX, ppmY, pixelRatio);         // 16-bit files: stbi__convert_16_to_8
    if (!r) return nullptr;
    if (x) *x = w;
    if (y) *y = h;
    if (comp) *comp = n;
    return r;
}

uint16_t* gamut_hip_stbi_load_16_from_memory(const uint8_t* data, size_t len, int* x, int* y, int* comp, int req_comp,
                                             float* ppmX, float* ppmY, float* pixelRatio)
{
    clear_error();
    int n = 0, w = 0, h = 0;
    uint8_t* r = png_load(data, len, &w, &h, &n, req_comp, 16, ppmX, ppmY, pixelRatio);        // 8-bit files: stbi__convert_8_to_16
    if (!r) return nullptr;
    if (x) *x = w;
    if (y) *y = h;
    if (comp) *comp = n;
    return (uint16_t*)r;
}

// Batch: chunk walk + inflate on host threads, one file per thread at a time, into pinned memory; the inflated stream goes
// to a device arena at once (copy stream).  Files that need nothing but de-filter + expand -- not interlaced, no palette,
// no tRNS, channel count and depth as asked -- are then de-filtered TOGETHER, one launch per geometry with one workgroup per
// image (a 4K image alone occupies one CU for 4.5 ms; 64 of them side by side take about as long).  The others run the
// whole of stbi__do_png on their worker's own stream right away.
int gamut_hip_png_decode_batch_device(const uint8_t* const* data, const size_t* len, int count, int req_comp, int bits,
                                      const int64_t* out_offset, uint8_t* out, gamut_hip_png_info* info, int* status_host,
                                      int threads, void* stream)
{
    clear_error();
    if (count < 0 || (count > 0 && (!data || !len || !out_offset || !out || !info)) || req_comp < 0 || req_comp > 4 || (bits != 0 && bits != 8 && bits != 16))
        return set_error(GAMUT_HIP_ERR_INVALID_ARG, "png_decode_batch_device: bad arguments");
    if (count == 0) return GAMUT_HIP_OK;
    int ndev = 0, dev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1)
        return set_error(GAMUT_HIP_ERR_NO_DEVICE, "no HIP device available (libgamut_hip has no CPU fallback)");
    (void)hipGetDevice(&dev);
    try {
        if (threads <= 0) threads = host_threads();
        threads = threads < 1 ? 1 : threads > count ? count : threads;
        hipStream_t st = pick_stream(stream);
        static thread_local PerDevice<hipStream_t> copy_stream_pd;
        hipStream_t& copy_stream_tl = copy_stream_pd.cur();
        if (!copy_stream_tl) GAMUT_HIP_CHECK(hipStreamCreateWithFlags(&copy_stream_tl, hipStreamNonBlocking));
        // a plain local: thread_local variables are not captured by the worker lambda below -- every worker thread would
        // read ITS OWN (null) instance and upload on the legacy null stream
        const hipStream_t copy_stream = copy_stream_tl;

        // 0. IHDR of every file: a slot in the device arena for its inflated stream
        std::vector<BatchFile> files((size_t)count);
        std::vector<int64_t> slot((size_t)count, -1), slot_bytes((size_t)count, 0);
        int64_t arena_bytes = 0;
        for (int i = 0; i < count; ++i) {
            PngHeader h;
            if (parse(data[i], len[i], h, true)) continue;                       // the worker reports the error
            const uint64_t wb = ((uint64_t)h.img_n * h.x * (uint32_t)h.depth + 7) >> 3, room = (wb + 8) * h.y + 64;      // Adam7: <= 15/8 y rows, each with a filter byte and a rounded-up last byte
            if (room > 0xFFFFFF00ull) continue;
            slot[(size_t)i] = arena_bytes; slot_bytes[(size_t)i] = (int64_t)room; arena_bytes += (int64_t)((room + 255) & ~(uint64_t)255);
        }
        clear_error();
        static thread_local PerDevice<DeviceScratch> arena_pd;
        DeviceScratch& arena = arena_pd.cur();
        uint8_t* d_arena = arena_bytes ? (uint8_t*)arena.get((size_t)arena_bytes, st) : nullptr;
        if (arena_bytes && !d_arena) return set_error(GAMUT_HIP_ERR_OUT_OF_MEMORY, "png: device arena of %lld bytes failed", (long long)arena_bytes);

        // 1. workers: chunk walk + inflate into the worker's pinned buffer, stream to the arena
        const bool trace = getenv("GAMUT_HIP_TRACE") != nullptr;
        const auto t_begin = std::chrono::steady_clock::now();
        auto ms_since = [](std::chrono::steady_clock::time_point t0) { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); };
        std::atomic<int64_t> us_inflate{ 0 }, us_upload{ 0 };
        std::atomic<int> next{ 0 };
        // Inflate on the GPU (inflate.hip) when the batch is large enough to beat the host threads: the workers then only
        // walk the chunks and gather the IDAT bytes into one pinned upload image (file i's share starts at blob_off[i]; its
        // IDAT bytes cannot outnumber the file's own).  GAMUT_HIP_PNG_INFLATE=host / device overrides the choice.
        bool device_inflate = count > threads;                   // (threads <= count here)
        if (const char* v = getenv("GAMUT_HIP_PNG_INFLATE")) device_inflate = strcmp(v, "device") == 0 ? true : strcmp(v, "host") == 0 ? false : device_inflate;
        std::vector<size_t> blob_off((size_t)count + 1, 0);
        std::vector<uint32_t> idat_len((size_t)count, 0), idat_skip((size_t)count, 0);
        uint8_t* h_blob = nullptr; uint8_t* d_blob = nullptr;
        static thread_local PerDevice<PinnedScratch> blob_pinned_pd;
        static thread_local PerDevice<DeviceScratch> blob_dev_pd;
        PinnedScratch& blob_pinned = blob_pinned_pd.cur(); DeviceScratch& blob_dev = blob_dev_pd.cur();
        // Device inflate, step one: the chunk walk of every file (no byte of IDAT data is touched: parse() notes where the payloads are).
        std::vector<std::vector<std::pair<const uint8_t*, uint32_t>>> segments(device_inflate ? (size_t)count : 0);
        auto gather = [&]() {
            (void)hipSetDevice(dev);
            for (int i; (i = next.fetch_add(1, std::memory_order_relaxed)) < count; ) {
                BatchFile& f = files[(size_t)i];
                PngHeader h;
                h.idat_segments = &segments[(size_t)i];
                const auto t_in = std::chrono::steady_clock::now();
                int rc = parse(data[i], len[i], h, false);
                if (rc == GAMUT_HIP_OK && !h.seen_idat) rc = set_error(GAMUT_HIP_ERR_DECODE, "png: no IDAT");
                uint32_t skip = 0;
                if (rc == GAMUT_HIP_OK && !h.is_iphone) {                       // the zlib header, as inflate_idat checks it (stbdec.d:1281-1290)
                    uint8_t b[2] = { 0, 0 }; uint32_t got = 0;
                    for (const auto& sg : segments[(size_t)i]) for (uint32_t k = 0; k < sg.second && got < 2; ++k) b[got++] = sg.first[k];
                    if (h.ioff < 2 || ((b[0] * 256 + b[1]) % 31) != 0 || (b[1] & 32) || (b[0] & 15) != 8) rc = set_error(GAMUT_HIP_ERR_DECODE, "png: bad zlib header");
                    skip = 2;
                }
                if (rc == GAMUT_HIP_OK && slot[(size_t)i] < 0) rc = set_error(GAMUT_HIP_ERR_DECODE, "png: image too large");
                if (rc != GAMUT_HIP_OK) { f.rc = rc; snprintf(f.msg, sizeof(f.msg), "image %d: %s", i, last_error_buf()); continue; }
                idat_len[(size_t)i] = h.ioff - skip;
                idat_skip[(size_t)i] = skip;
                h.idat_segments = nullptr;
                f.h = h;
                f.out_n = h.img_n;
                if ((req_comp == h.img_n + 1 && req_comp != 3 && !h.pal_img_n) || h.has_trans) f.out_n = h.img_n + 1;      // stbdec.d:1821-1824
                f.bits = h.depth <= 8 ? 8 : 16; f.channels_in_file = h.img_n;
                const uint64_t wb = ((uint64_t)h.img_n * h.x * (uint32_t)h.depth + 7) >> 3;
                f.need = (uint32_t)((wb + 1) * h.y);
                us_inflate += (int64_t)(ms_since(t_in) * 1000);
            }
        };
        auto work = [&]() {
            (void)hipSetDevice(dev);
            HostBuf buf = pinned_pool().take();
            hipEvent_t copied = nullptr;
            (void)hipEventCreateWithFlags(&copied, hipEventDisableTiming);
            for (int i; (i = next.fetch_add(1, std::memory_order_relaxed)) < count; ) {
                BatchFile& f = files[(size_t)i];
                PngJob j; j.raw = buf;                                           // borrow the worker's pinned buffer
                const auto t_in = std::chrono::steady_clock::now();
                const int rc = png_prepare(data[i], len[i], j);
                buf = j.raw; j.raw = HostBuf{};
                us_inflate += (int64_t)(ms_since(t_in) * 1000);
                if (rc != GAMUT_HIP_OK) { f.rc = rc; snprintf(f.msg, sizeof(f.msg), "image %d: %s", i, last_error_buf()); continue; }
                free(j.h.idata); j.h.idata = nullptr;
                f.h = j.h;
                const PngHeader& h = f.h;
                f.out_n = h.img_n;
                if ((req_comp == h.img_n + 1 && req_comp != 3 && !h.pal_img_n) || h.has_trans) f.out_n = h.img_n + 1;      // stbdec.d:1821-1824
                f.bits = h.depth <= 8 ? 8 : 16; f.channels_in_file = h.img_n;
                const uint64_t wb = ((uint64_t)h.img_n * h.x * (uint32_t)h.depth + 7) >> 3, need = (wb + 1) * h.y;
                f.need = (uint32_t)need;
                if (slot[(size_t)i] < 0 || !copied) { f.rc = GAMUT_HIP_ERR_DECODE; snprintf(f.msg, sizeof(f.msg), "image %d: png: image too large", i); continue; }
                f.raw_len = (uint32_t)((int64_t)j.raw_len < slot_bytes[(size_t)i] ? (int64_t)j.raw_len : slot_bytes[(size_t)i]);
                const auto t_upl = std::chrono::steady_clock::now();
                if (hipMemcpyAsync(d_arena + slot[(size_t)i], buf.p, f.raw_len, hipMemcpyHostToDevice, copy_stream) != hipSuccess ||
                    hipEventRecord(copied, copy_stream) != hipSuccess || hipEventSynchronize(copied) != hipSuccess) {
                    f.rc = GAMUT_HIP_ERR_HIP; snprintf(f.msg, sizeof(f.msg), "image %d: png: upload failed", i); continue;
                }
                us_upload += (int64_t)(ms_since(t_upl) * 1000);
                f.uploaded = true;
                f.batched = !h.interlace && !h.has_trans && !h.pal_img_n && (req_comp == 0 || req_comp == f.out_n) && (bits == 0 || bits == f.bits) && f.raw_len >= need;
            }
            if (copied) (void)hipEventDestroy(copied);
            pinned_pool().give(buf);
        };
        {
            std::vector<std::thread> pool;
            if (device_inflate) {
                try { for (int t = 1; t < threads; ++t) pool.emplace_back(gather); } catch (...) {}
                gather();
            } else {
                try { for (int t = 1; t < threads; ++t) pool.emplace_back(work); } catch (...) {}
                work();
            }
            for (std::thread& th : pool) th.join();
        }
        clear_error();
        GAMUT_HIP_CHECK(hipStreamSynchronize(copy_stream));                       // (every worker waited for its own copies already)
        double ms_device_inflate = 0;
        if (device_inflate) {
            // Step two: the IDAT bytes go up SLICE BY SLICE -- slice r of every stream, then slice r + 1 of every stream -- and the inflate
            // kernel is launched once per slice behind them (inflate_sliced_*: a stream stops where the bytes it may read end and goes on
            // in the next launch).  A stream inflates at ~150-600 MB/s, all of them side by side: about what PCIe delivers to all of them
            // together, so the upload hides behind the kernel instead of standing in front of it.
            const auto t_inf = std::chrono::steady_clock::now();
            std::vector<gamut_hip_inflate_desc> descs; std::vector<int> who;
            uint32_t longest = 0;
            for (int i = 0; i < count; ++i) if (files[(size_t)i].rc == GAMUT_HIP_OK) who.push_back(i);
            // longest streams first: a stream is one workgroup for its whole length, and workgroups start in the order of the list -- with
            // more streams than compute units the long ones must not be the ones that start last
            std::stable_sort(who.begin(), who.end(), [&](int a, int b) { return idat_len[(size_t)a] > idat_len[(size_t)b]; });
            for (int i : who) longest = idat_len[(size_t)i] > longest ? idat_len[(size_t)i] : longest;
            if (!who.empty()) {
                const size_t n = who.size();
                static thread_local PerDevice<DeviceScratch> verdict_dev_pd;
                DeviceScratch& verdict_dev = verdict_dev_pd.cur();
                uint32_t* d_verdict = (uint32_t*)verdict_dev.get(n * 8);
                if (!d_verdict) return set_error(GAMUT_HIP_ERR_OUT_OF_MEMORY, "png: verdict table allocation failed");
                std::vector<uint32_t> verdict(n * 2);
                // 1 MiB slices; 512 KiB for up to 256 well-compressed large images, where a stream inflates more slowly than it arrives
                // (measured, 256 files: 4K of 12 MB 100 ms with 1 MiB slices, 127 with 512 KiB; 4K of 3.6 MB 60 / 57 ms; 1080p of 3.1 MB
                // 33 / 36 ms; 1024 files of 1080p 118 / 132 ms); at most 32 launches
                int64_t largest_out = 0;
                for (int i : who) largest_out = std::max(largest_out, slot_bytes[(size_t)i]);
                uint32_t slice = (longest < (8u << 20) && n <= 256 && largest_out >= (16 << 20)) ? 1u << 19 : 1u << 20;
                if (const char* v = getenv("GAMUT_HIP_PNG_SLICE_KB")) { const long kb = atol(v); if (kb >= 64 && kb <= (1 << 20)) slice = (uint32_t)kb << 10; }      // (tuning)
                while ((uint64_t)slice * 32u < longest) slice *= 2;
                const int rounds = longest ? (int)(((uint64_t)longest + slice - 1) / slice) : 1;
                // The staging image.  One copy per (stream, slice) is a megabyte at a time: 37 GB/s on one copy stream where the link does 57
                // (tools/microbench/h2d_streams.hip, profiles/r04_h2d_streams.txt).  With every stream's share `pitch` bytes apart -- the
                // streams in the order of the list, longest first, so that those still running in a round are a prefix -- slice r of all of
                // them is ONE pitched copy (hipMemcpy2DAsync: 57 GB/s even with 512 KiB rows).  Batches of very unequal streams (the pitched
                // image would be more than twice the data) keep their tight layout and piece-wise copies, spread over two copy streams (46 GB/s).
                uint64_t data_bytes = 0;
                for (int i : who) data_bytes += ((uint64_t)idat_len[(size_t)i] + 15) & ~(uint64_t)15;
                const uint64_t pitch = (uint64_t)rounds * slice;
                bool pitched = pitch * n <= 2 * data_bytes + (64u << 20);
                if (const char* v = getenv("GAMUT_HIP_PNG_PITCHED")) pitched = atoi(v) != 0;                           // (measurements)
                const uint64_t blob_bytes = pitched ? pitch * n : data_bytes;
                h_blob = blob_pinned.get(blob_bytes + 16);
                d_blob = (uint8_t*)blob_dev.get(blob_bytes + 16);
                if (!h_blob || !d_blob) return set_error(GAMUT_HIP_ERR_OUT_OF_MEMORY, "png: staging for %llu bytes of IDAT data failed", (unsigned long long)blob_bytes);
                {
                    uint64_t at = 0;
                    for (size_t k = 0; k < n; ++k) { const int i = who[k]; blob_off[(size_t)i] = pitched ? k * pitch : at; at += ((uint64_t)idat_len[(size_t)i] + 15) & ~(uint64_t)15; }
                }
                for (int i : who) descs.push_back(gamut_hip_inflate_desc{ d_blob + blob_off[(size_t)i], d_arena + slot[(size_t)i], idat_len[(size_t)i], (uint32_t)slot_bytes[(size_t)i] });
                static thread_local PerDevice<hipStream_t> copy2_pd;
                hipStream_t& copy_stream2 = copy2_pd.cur();
                if (!copy_stream2 && hipStreamCreateWithFlags(&copy_stream2, hipStreamNonBlocking) != hipSuccess) return set_error(GAMUT_HIP_ERR_HIP, "png: stream creation failed");
                // the units of work, slice-major: (round, stream)
                std::vector<std::pair<int, int>> units;
                std::vector<int> units_in_round((size_t)rounds, 0);
                for (int r = 0; r < rounds; ++r)
                    for (size_t k = 0; k < n; ++k) if ((uint64_t)r * slice < idat_len[(size_t)who[k]] || (r == 0 && idat_len[(size_t)who[k]] == 0)) { units.emplace_back(r, (int)k); ++units_in_round[(size_t)r]; }
                std::vector<std::atomic<int>> round_done((size_t)rounds);
                for (auto& a : round_done) a.store(0);
                std::atomic<size_t> next_unit{ 0 };
                std::atomic<bool> upload_failed{ false };
                auto upload = [&]() {
                    (void)hipSetDevice(dev);
                    for (size_t u; (u = next_unit.fetch_add(1, std::memory_order_relaxed)) < units.size(); ) {
                        const int r = units[u].first, i = who[(size_t)units[u].second];
                        const uint64_t lo = (uint64_t)r * slice, hi = std::min<uint64_t>(lo + slice, idat_len[(size_t)i]);
                        // bytes [lo, hi) of the stream = bytes [lo + skip, hi + skip) of the concatenated IDAT payloads
                        uint8_t* dst = h_blob + blob_off[(size_t)i] + lo;
                        uint64_t at = 0, want_lo = lo + idat_skip[(size_t)i], want_hi = hi + idat_skip[(size_t)i];
                        for (const auto& sg : segments[(size_t)i]) {
                            const uint64_t s_lo = at, s_hi = at + sg.second;
                            at = s_hi;
                            if (s_hi <= want_lo) continue;
                            if (s_lo >= want_hi) break;
                            const uint64_t c_lo = std::max(s_lo, want_lo), c_hi = std::min(s_hi, want_hi);
                            memcpy(dst + (c_lo - want_lo), sg.first + (c_lo - s_lo), (size_t)(c_hi - c_lo));
                        }
                        if (!pitched && hi > lo && hipMemcpyAsync(d_blob + blob_off[(size_t)i] + lo, dst, (size_t)(hi - lo), hipMemcpyHostToDevice, (u & 1) ? copy_stream2 : copy_stream) != hipSuccess) {
                            (void)hipGetLastError(); upload_failed.store(true);
                        }
                        round_done[(size_t)r].fetch_add(1, std::memory_order_release);
                    }
                };
                if (int rc = inflate_sliced_begin(descs.data(), (int)n, st)) return rc;
                std::vector<std::thread> pool;
                try { for (int t = 0; t < threads; ++t) pool.emplace_back(upload); } catch (...) {}
                if (pool.empty()) upload();                                       // (no thread could be started: this one does it all, then launches)
                std::vector<hipEvent_t> up((size_t)rounds, nullptr);
                std::vector<std::vector<uint32_t>> avail((size_t)rounds);          // (one table per launch, alive until the stream has been waited for)
                int launch_rc = GAMUT_HIP_OK;
                for (int r = 0; r < rounds && launch_rc == GAMUT_HIP_OK; ++r) {
                    while (round_done[(size_t)r].load(std::memory_order_acquire) < units_in_round[(size_t)r]) std::this_thread::yield();
                    // slice r of every stream is in the pinned image (pitched: it goes up now, in one copy) / has been queued on the copy streams:
                    // the compute stream waits for it, then inflates what is there
                    if (pitched && units_in_round[(size_t)r] > 0 &&
                        hipMemcpy2DAsync(d_blob + (uint64_t)r * slice, pitch, h_blob + (uint64_t)r * slice, pitch, slice, (size_t)units_in_round[(size_t)r], hipMemcpyHostToDevice, copy_stream) != hipSuccess) {
                        (void)hipGetLastError(); launch_rc = set_error(GAMUT_HIP_ERR_HIP, "png: upload of slice %d failed", r); break;
                    }
                    if (hipEventCreateWithFlags(&up[(size_t)r], hipEventDisableTiming) != hipSuccess || hipEventRecord(up[(size_t)r], copy_stream) != hipSuccess ||
                        hipStreamWaitEvent(st, up[(size_t)r], 0) != hipSuccess) { (void)hipGetLastError(); launch_rc = set_error(GAMUT_HIP_ERR_HIP, "png: event for slice %d failed", r); break; }
                    if (!pitched) {
                        hipEvent_t e2 = nullptr;
                        if (hipEventCreateWithFlags(&e2, hipEventDisableTiming) != hipSuccess || hipEventRecord(e2, copy_stream2) != hipSuccess || hipStreamWaitEvent(st, e2, 0) != hipSuccess) {
                            (void)hipGetLastError(); launch_rc = set_error(GAMUT_HIP_ERR_HIP, "png: event for slice %d failed", r); break; }
                        up.push_back(e2);
                    }
                    avail[(size_t)r].resize(n);
                    for (size_t k = 0; k < n; ++k) avail[(size_t)r][k] = (uint32_t)std::min<uint64_t>((uint64_t)(r + 1) * slice, idat_len[(size_t)who[k]]);
                    launch_rc = inflate_sliced_step((int)n, avail[(size_t)r].data(), d_verdict, d_verdict + n, st);
                }
                for (std::thread& th : pool) th.join();
                if (launch_rc == GAMUT_HIP_OK && upload_failed.load()) launch_rc = set_error(GAMUT_HIP_ERR_HIP, "png: upload of the IDAT data failed");
                if (launch_rc == GAMUT_HIP_OK && hipMemcpyAsync(verdict.data(), d_verdict, n * 8, hipMemcpyDeviceToHost, st) != hipSuccess) launch_rc = set_error(GAMUT_HIP_ERR_HIP, "png: verdict download failed");
                const hipError_t sync_rc = hipStreamSynchronize(st);
                (void)hipStreamSynchronize(copy_stream); (void)hipStreamSynchronize(copy_stream2);
                for (hipEvent_t e : up) if (e) (void)hipEventDestroy(e);
                if (launch_rc != GAMUT_HIP_OK) return launch_rc;
                if (sync_rc != hipSuccess) return set_error(GAMUT_HIP_ERR_HIP, "png: inflate on the device failed: %s", hipGetErrorString(sync_rc));
                for (size_t k = 0; k < n; ++k) {
                    BatchFile& f = files[(size_t)who[k]];
                    if (verdict[n + k]) { f.rc = GAMUT_HIP_ERR_DECODE; snprintf(f.msg, sizeof(f.msg), "image %d: png: corrupt zlib stream", who[k]); continue; }
                    const PngHeader& h = f.h;
                    f.raw_len = verdict[k];
                    f.uploaded = true;
                    f.batched = !h.interlace && !h.has_trans && !h.pal_img_n && (req_comp == 0 || req_comp == f.out_n) && (bits == 0 || bits == f.bits) && f.raw_len >= f.need;
                }
            }
            ms_device_inflate = ms_since(t_inf);
        }
        const double ms_workers = ms_since(t_begin);

        // 2. one de-filter launch per geometry
        std::map<std::tuple<uint32_t, uint32_t, int, int, int, int>, std::vector<int>> groups;
        bool aligned = ((uintptr_t)out % 4) == 0, lines = ((uintptr_t)out % 128) == 0;       // every image on a dword / on a 128-byte line of memory
        int n_batched = 0;
        for (int i = 0; i < count; ++i) {
            const BatchFile& f = files[(size_t)i];
            if (!f.batched) continue;
            groups[std::make_tuple(f.h.x, f.h.y, f.h.img_n, f.out_n, f.h.depth, f.h.color)].push_back(i);
            aligned = aligned && (out_offset[i] % 4) == 0;
            lines = lines && (out_offset[i] % 128) == 0;
            ++n_batched;
        }
        std::vector<uint32_t> status((size_t)n_batched, 0);
        std::vector<int> order; order.reserve((size_t)n_batched);
        int launch_rc = GAMUT_HIP_OK;
        if (n_batched) {
            std::vector<int64_t> offs((size_t)n_batched * 2);
            for (auto& g : groups) for (int i : g.second) { offs[order.size()] = slot[(size_t)i]; offs[(size_t)n_batched + order.size()] = out_offset[i]; order.push_back(i); }
            static thread_local PerDevice<DeviceScratch> tables_pd;
            DeviceScratch& tables = tables_pd.cur();
            const size_t o_status = (size_t)n_batched * 16;
            uint8_t* d_tab = (uint8_t*)tables.get(o_status + (size_t)n_batched * 4);
            if (!d_tab) return set_error(GAMUT_HIP_ERR_OUT_OF_MEMORY, "png: table allocation failed");
            GAMUT_HIP_CHECK(hipMemcpyAsync(d_tab, offs.data(), offs.size() * 8, hipMemcpyHostToDevice, st));
            GAMUT_HIP_CHECK(hipMemsetAsync(d_tab + o_status, 0, (size_t)n_batched * 4, st));
            size_t base = 0;
            for (auto& g : groups) {
                const BatchFile& f = files[(size_t)g.second[0]];
                const int n = (int)g.second.size();
                const int rc = png_defilter_launch(d_arena, 0, f.need, out, 0, f.h.x, f.h.y, f.h.img_n, f.out_n, f.h.depth, f.h.color, n,
                                                   (uint32_t*)(d_tab + o_status) + base, st,
                                                   (const int64_t*)d_tab + base, (const int64_t*)d_tab + n_batched + base, aligned, lines);
                if (rc != GAMUT_HIP_OK && launch_rc == GAMUT_HIP_OK) {
                    launch_rc = rc;
                    for (int i : g.second) { files[(size_t)i].rc = rc; snprintf(files[(size_t)i].msg, sizeof(files[(size_t)i].msg), "image %d: %s", i, last_error_buf()); }
                }
                base += (size_t)n;
            }
            GAMUT_HIP_CHECK(hipMemcpyAsync(status.data(), d_tab + o_status, (size_t)n_batched * 4, hipMemcpyDeviceToHost, st));
            GAMUT_HIP_CHECK(hipStreamSynchronize(st));
            for (size_t k = 0; k < order.size(); ++k)
                if (status[k] && files[(size_t)order[k]].rc == GAMUT_HIP_OK) {
                    files[(size_t)order[k]].rc = GAMUT_HIP_ERR_DECODE; snprintf(files[(size_t)order[k]].msg, sizeof(files[(size_t)order[k]].msg), "image %d: png: invalid filter", order[k]);
                }
        }
        // 3. the others (interlaced, palette, tRNS, channel or depth conversion): the whole of stbi__do_png per file, from the arena
        for (int i = 0; i < count; ++i) {
            BatchFile& f = files[(size_t)i];
            if (f.rc != GAMUT_HIP_OK || !f.uploaded || f.batched) continue;
            int w = 0, hh = 0, n = 0, b = 8;
            if (png_run(f.h, nullptr, f.raw_len, req_comp, bits, out + out_offset[i], &w, &hh, &n, &b, st, d_arena + slot[(size_t)i])) { f.channels_in_file = n; f.bits = b; }
            else { f.rc = GAMUT_HIP_ERR_DECODE; snprintf(f.msg, sizeof(f.msg), "image %d: %s", i, last_error_buf()); }
        }
        clear_error();
        if (trace && device_inflate) fprintf(stderr, "[gamut_hip] png_decode_batch_device: inflate on the device: upload + kernel + verdicts %.1f ms\n", ms_device_inflate);
        if (trace) fprintf(stderr, "[gamut_hip] png_decode_batch_device: %d files on %d threads: workers %.1f ms (per file: chunk walk + inflate %.1f ms, upload + wait %.1f ms), GPU stages %.1f ms (%d files in %d batched launches)\n",
                           count, threads, ms_workers, us_inflate.load() / 1000.0 / count, us_upload.load() / 1000.0 / count, ms_since(t_begin) - ms_workers, n_batched, (int)groups.size());
        // 4. results
        int first = GAMUT_HIP_OK; char first_msg[200] = { 0 };
        for (int i = 0; i < count; ++i) {
            const BatchFile& f = files[(size_t)i];
            memset(&info[i], 0, sizeof(info[i]));
            if (f.rc == GAMUT_HIP_OK) {
                info[i].width = f.h.x; info[i].height = f.h.y; info[i].channels_in_file = f.channels_in_file;
                info[i].channels = req_comp ? req_comp : f.channels_in_file; info[i].bits = f.bits;
                info[i].pixels_per_meter_x = f.h.ppmX; info[i].pixels_per_meter_y = f.h.ppmY; info[i].pixel_aspect_ratio = f.h.aspect;
            }
            if (status_host) status_host[i] = f.rc;
            if (f.rc != GAMUT_HIP_OK && first == GAMUT_HIP_OK) { first = f.rc; snprintf(first_msg, sizeof(first_msg), "%s", f.msg); }
        }
        if (first != GAMUT_HIP_OK) return set_error(first, "%s", first_msg);
        return GAMUT_HIP_OK;
    } catch (...) {
        return set_error(GAMUT_HIP_ERR_OUT_OF_MEMORY, "png_decode_batch_device: out of host memory");
    }
}

int gamut_hip_png_read_header(const uint8_t* data, size_t len, gamut_hip_png_info* info)
{
    clear_error();
    if (!info) return set_error(GAMUT_HIP_ERR_INVALID_ARG, "png_read_header: null info");
    memset(info, 0, sizeof(*info));
    PngHeader h;
    if (int rc = parse(data, len, h, true)) return rc;
    info->width = h.x; info->height = h.y; info->bits = h.depth == 16 ? 16 : 8;
    info->channels_in_file = info->channels = h.color == 3 ? 3 : h.img_n;          // palette images expand to RGB (RGBA with tRNS: known after the chunk walk)
    info->pixels_per_meter_x = info->pixels_per_meter_y = info->pixel_aspect_ratio = -1;
    return GAMUT_HIP_OK;
}

int gamut_hip_png_is16(const uint8_t* data, size_t len)
{
    PngHeader h;
    if (parse(data, len, h, true)) { clear_error(); return 0; }
    return h.depth == 16 ? 1 : 0;
}

} // extern "C"
