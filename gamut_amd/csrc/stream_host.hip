// stream_host.hip -- the callback-shaped drop-ins: the same arguments as the reference's codec entry points, so the swap
// in plugins/jpeg.d / plugins/png.d is argument for argument.
//
//   decompress_jpeg_image_from_stream(JpegStreamReadFunc, void* user, ...)     jpegload.d:61-70, 3720-3723
//   stbi_load_from_callbacks / stbi_load_16_from_callbacks(stbi_io_callbacks*)  stbdec.d:408-419, 713-735
//   stbi__png_is16 on a callback context                                        stbdec.d:2091-2109 (plugins/png.d:50-62)
//
// The GPU path wants the whole compressed image (entropy decode / inflate are per-file stages, the kernels start from
// dense coefficients / the inflated stream), so these gather the image's bytes through the caller's callbacks and hand them
// to the *_from_memory entry points -- but ONLY the image's bytes: an image embedded in a longer stream
// (Image.loadFromStream, image.d:916) must not be read past.
//   PNG: the chunk walk of stbi__parse_png_file (stbdec.d:1777-2023) over the callbacks -- 8-byte chunk headers, payloads of
//        the chunks the parser looks at by `read`, every other ancillary chunk by `skip` (stbi__skip :822-842), the `eof`
//        callback for the missing-IEND case (:2008-2012), and it stops behind IEND's CRC (:1998-2001).  stb itself reads
//        through a 128-byte buffer (:470-471, 780-795) and so leaves the stream up to 127 bytes behind the image; this walk
//        reads exact amounts and leaves it AT the end of the image: a second image may follow at once.
//   JPEG: jpgd pulls JPGD_IN_BUF_SIZE = 8192 bytes per call (prep_in_buffer :1971-2003) and stops decoding at EOI; the
//        gatherer pulls the same 8 KiB pieces, follows the marker structure (segments by their length, entropy-coded data up
//        to the next real marker) and stops calling the reader once EOI has arrived -- the over-read is the reference's
//        own: less than one 8 KiB piece.
// Host only; no device work of their own.
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

// Follows the marker structure of a JPEG file over the bytes gathered so far: `done` once EOI has been seen.
struct JpegWalk {
    enum State { kSoi, kMarker, kLen, kPayload, kScan } st = kSoi;
    size_t pos = 0, skip_to = 0;
    bool done = false;
    void feed(const uint8_t* p, size_t len)
    {
        while (!done) {
            switch (st) {
            case kSoi: {                                     // locate_soi_marker (:1854-1908): FF D8 at once, or within the first 4097 bytes and before any FF D9
                if (len - pos < 2) return;
                if (p[0] == 0xFF && p[1] == 0xD8) { pos = 2; st = kMarker; break; }
                size_t q = 2; bool found = false, never = false;
                for (; q < len && q < 4097 && !found && !never; ++q) {
                    if (p[q - 1] == 0xFF && p[q] == 0xD8) found = true;
                    else if (p[q - 1] == 0xFF && p[q] == 0xD9) never = true;
                }
                if (found) { pos = q; st = kMarker; break; }
                if (never || q >= 4097) { done = true; return; }   // not a JPEG: let the decoder say so
                return;                                      // not decided yet: more bytes
            }
            case kMarker: {                                  // FF (FF)* code, as next_marker skips fill bytes (:1578-1600)
                while (pos < len && p[pos] != 0xFF) ++pos;   // garbage before a marker is skipped there too
                size_t q = pos;
                while (q < len && p[q] == 0xFF) ++q;
                if (q >= len) return;
                const uint8_t code = p[q];
                pos = q + 1;
                if (code == 0xD9) { done = true; return; }
                if (code == 0x00 || code == 0x01 || (code >= 0xD0 && code <= 0xD8)) break;      // stand-alone codes: no length
                sos = code == 0xDA; st = kLen; break;
            }
            case kLen:
                if (len - pos < 2) return;
                skip_to = pos + (((size_t)p[pos] << 8) | p[pos + 1]);
                if (skip_to < pos + 2) skip_to = pos + 2;
                st = kPayload; break;
            case kPayload:
                if (len < skip_to) return;
                pos = skip_to; st = sos ? kScan : kMarker; break;
            case kScan:                                      // entropy-coded bytes up to FF xx, xx not 00 / RSTn / FF
                for (;;) {
                    const uint8_t* f = (const uint8_t*)memchr(p + pos, 0xFF, len - pos);
                    if (!f) { pos = len; return; }
                    pos = (size_t)(f - p);
                    if (pos + 1 >= len) return;
                    const uint8_t c = p[pos + 1];
                    if (c == 0x00 || (c >= 0xD0 && c <= 0xD7)) { pos += 2; continue; }
                    if (c == 0xFF) { pos += 1; continue; }
                    st = kMarker; break;
                }
                break;
            }
        }
    }
    bool sos = false;
};

// jpgd's prep_in_buffer (:1980-2003) calls the read function until its 8 KiB buffer is full or *pEOF_flag is set; -1 is an
// error (stop_decoding(JPGD_STREAM_READ)).  Same pieces; no further call once the image's EOI is in.
bool gather_jpeg(gamut_hip_jpeg_stream_read_func rd, void* user, Slurp& s)
{
    const int chunk = 8192;                                  // JPGD_IN_BUF_SIZE
    int idle = 0;
    JpegWalk walk;
    for (;;) {
        if (s.len > kMaxFile || !s.room((size_t)chunk)) { set_error(GAMUT_HIP_ERR_OUT_OF_MEMORY, "jpeg: stream too long"); return false; }
        unsigned char eof = 0;
        const int n = rd(s.p + s.len, chunk, &eof, user);
        if (n < 0 || n > chunk) { set_error(GAMUT_HIP_ERR_DECODE, "jpeg: stream read error"); return false; }
        s.len += (size_t)n;
        if (eof) return true;
        walk.feed(s.p, s.len);
        if (walk.done) return true;
        // a source that returns 0 bytes without ever raising the flag would spin the reference's loop for ever: give up instead
        idle = n == 0 ? idle + 1 : 0;
        if (idle >= 64) return true;
    }
}

// exact reads over stbi_io_callbacks; a read of 0 bytes is the end of the data (stbi__refill_buffer :780-795)
struct StbSource {
    const gamut_hip_stbi_io_callbacks* c; void* user;
    bool ended = false;
    // appends up to n bytes to s; false on allocation failure.  *got = bytes that arrived
    bool read(Slurp& s, size_t n, size_t* got)
    {
        *got = 0;
        while (*got < n && !ended) {
            const size_t want = n - *got < ((size_t)1 << 20) ? n - *got : (size_t)1 << 20;
            if (!s.room(want)) { set_error(GAMUT_HIP_ERR_OUT_OF_MEMORY, "png: stream too long"); return false; }
            const int r = c->read(user, (char*)(s.p + s.len), (int)want);
            if (r <= 0) { ended = true; break; }
            const size_t k = (size_t)r > want ? want : (size_t)r;
            s.len += k; *got += k;
        }
        return true;
    }
    void skip(size_t n)
    {
        if (!n || ended) return;
        if (c->skip) { c->skip(user, (int)n); return; }      // the reference's stbi__skip beyond its buffer (:836)
        char tmp[4096];                                      // callers without a skip callback: read and drop
        while (n && !ended) { const int r = c->read(user, tmp, (int)(n < sizeof(tmp) ? n : sizeof(tmp))); if (r <= 0) { ended = true; break; } n -= (size_t)r; }
    }
    bool at_eof() { return ended || (c->eof && c->eof(user)); }
};

constexpr uint32_t png_type(char a, char b, char c, char d) { return ((uint32_t)(uint8_t)a << 24) | ((uint32_t)(uint8_t)b << 16) | ((uint32_t)(uint8_t)c << 8) | (uint8_t)d; }

// Gathers one PNG: signature, then chunk by chunk as stbi__parse_png_file visits them.  What arrives in `s` is a PNG file
// the memory parser reads to the same result: ancillary chunks the parser skips are left out, CgBI (skipped too, but its
// presence matters: is_iphone) keeps its header with an empty payload.  A malformed stream ends the walk early; the memory
// parser then reports the error on the bytes that did arrive.  header_only: stop after IHDR (stbi__png_is16).
bool gather_png(StbSource& src, Slurp& s, bool header_only)
{
    size_t got;
    if (!src.read(s, 8, &got)) return false;
    if (got < 8) return true;
    bool first = true;                                           // stbi__parse_png_file's `first`: only IHDR (or CgBI in front of it) may come first
    for (;;) {
        if (s.len > kMaxFile) { set_error(GAMUT_HIP_ERR_OUT_OF_MEMORY, "png: stream too long"); return false; }
        const size_t at = s.len;
        if (!src.read(s, 8, &got)) return false;
        if (got < 8) return true;                                // the stream ends here: a file without IEND (issue #92) or a truncated one
        const uint8_t* h = s.p + at;
        const uint32_t length = ((uint32_t)h[0] << 24) | ((uint32_t)h[1] << 16) | ((uint32_t)h[2] << 8) | h[3];
        const uint32_t type = ((uint32_t)h[4] << 24) | ((uint32_t)h[5] << 16) | ((uint32_t)h[6] << 8) | h[7];
        bool payload;
        switch (type) {
        case png_type('I','H','D','R'): case png_type('P','L','T','E'): case png_type('t','R','N','S'):
        case png_type('I','D','A','T'): case png_type('p','H','Y','s'): case png_type('I','E','N','D'):
            payload = true; break;
        case png_type('C','g','B','I'):
            if (length > 0x7fffffffu) return true;
            payload = false; s.p[at] = s.p[at + 1] = s.p[at + 2] = s.p[at + 3] = 0; break;       // keep the (now empty) chunk
        default:
            if (first) return true;                                                              // "first not IHDR" (:2003-2006): the header stays, the memory parser rejects it
            if (type == 0 && src.at_eof()) { s.len = at; return true; }                          // issue #92: no IEND (:2008-2012)
            if (!(type & (1u << 29))) return true;                                               // unknown critical chunk: the parser rejects it
            if (length > 0x7fffffffu) return true;                                               // the over-long header stays in `s`: no clean end behind IDAT
            payload = false; s.len = at; break;                                                  // ancillary: skipped, left out
        }
        if (type != png_type('C','g','B','I')) first = false;
        if (length > 0x7fffffffu) return true;                                                   // stbi__get_chunk_header's callers reject it
        if (payload) { if (!src.read(s, length, &got)) return false; if (got < length) return true; }
        else src.skip(length);
        if (type == png_type('C','g','B','I') || payload) { if (!src.read(s, 4, &got)) return false; if (got < 4) return true; }   // CRC
        else { Slurp crc; if (!src.read(crc, 4, &got)) return false; if (got < 4) return true; }
        if (type == png_type('I','E','N','D')) return true;
        if (header_only && type == png_type('I','H','D','R')) return true;
    }
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
    if (!gather_jpeg(rfn, userData, s)) return nullptr;
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
    StbSource src{ clbk, user };
    if (!gather_png(src, s, false)) return nullptr;
    return gamut_hip_stbi_load_from_memory(s.p, s.len, x, y, comp, req_comp, ppmX, ppmY, pixelRatio);
}

uint16_t* gamut_hip_stbi_load_16_from_callbacks(const gamut_hip_stbi_io_callbacks* clbk, void* user, int* x, int* y, int* comp, int req_comp,
                                                float* ppmX, float* ppmY, float* pixelRatio)
{
    clear_error();
    if (!clbk || !clbk->read) { set_error(GAMUT_HIP_ERR_INVALID_ARG, "png: null callbacks"); return nullptr; }
    Slurp s;
    StbSource src{ clbk, user };
    if (!gather_png(src, s, false)) return nullptr;
    return gamut_hip_stbi_load_16_from_memory(s.p, s.len, x, y, comp, req_comp, ppmX, ppmY, pixelRatio);
}

int gamut_hip_stbi_png_is16_from_callbacks(const gamut_hip_stbi_io_callbacks* clbk, void* user)
{
    // stbi__png_is16 parses the header only (signature + IHDR = 33 bytes; a CgBI chunk may stand before IHDR); the caller
    // rewinds its stream afterwards, as plugins/png.d:50-62 does
    if (!clbk || !clbk->read) return 0;
    Slurp s;
    StbSource src{ clbk, user };
    if (!gather_png(src, s, true)) return 0;
    return gamut_hip_png_is16(s.p, s.len);
}

} // extern "C"
