/*
 * oracle_png.c -- CPU restatement of Gamut's PNG path (stb_image 2.27 port).
 * TEST INFRASTRUCTURE ONLY (see gamut_oracle.h).
 *
 * Follows /root/reference/source/gamut/codecs/stbdec.d:
 *   16<->8 :635-666; load_and_postprocess :669-707; BYTECAST :895-898;
 *   compute_y / convert_format(16) :911-1199; zlib wrapper :1262-1321;
 *   first_row_filter :1381-1388; paeth :1390-1401; depth scale :1403;
 *   create_png_image_raw :1406-1635; Adam7 :1637-1680; tRNS :1682-1730;
 *   palette :1732-1765; parse_png_file / finalize_decode :1777-2023;
 *   do_png :2025-2055.
 * Inflate: the reference calls the third-party `miniz` dub package (un-vendored,
 * version unpinned: dub.json:10) with adler32 checking off; this file uses the
 * system zlib in raw mode, which yields the same bytes for any valid stream.
 */
#include "gamut_oracle.h"
#include <stdlib.h>
#include <string.h>
#include <zlib.h>

typedef uint8_t stbi_uc;

enum { F_none = 0, F_sub = 1, F_up = 2, F_avg = 3, F_paeth = 4, F_avg_first, F_paeth_first };
static const stbi_uc first_row_filter[5] = { F_none, F_sub, F_none, F_avg_first, F_paeth_first };   /* :1381-1388 */
static const stbi_uc depth_scale_table[9] = { 0, 0xff, 0x55, 0, 0x11, 0, 0, 0, 0x01 };               /* :1403 */

static int paeth(int a, int b, int c)      /* :1390-1401 */
{
    int p = a + b - c;
    int pa = abs(p - a), pb = abs(p - b), pc = abs(p - c);
    if (pa <= pb && pa <= pc) return a;
    if (pb <= pc) return b;
    return c;
}
#define BYTECAST(x) ((stbi_uc)((x) & 255))

/* stbi__create_png_image_raw :1406-1635. `out` is caller-provided (x*y*out_n*bytes). */
int orc_png_create_image_raw(const uint8_t* raw, uint32_t raw_len, int img_n, int out_n,
                             uint32_t x, uint32_t y, int depth, int color, uint8_t* out)
{
    const int bytes = (depth == 16 ? 2 : 1);
    const uint32_t stride = x * (uint32_t)out_n * (uint32_t)bytes;
    const int output_bytes = out_n * bytes;
    int filter_bytes = img_n * bytes;
    int width = (int)x;
    uint32_t i, j;
    int k;

    if (!(out_n == img_n || out_n == img_n + 1)) return 0;
    const uint32_t img_width_bytes = (((uint32_t)img_n * x * (uint32_t)depth) + 7) >> 3;
    const uint64_t img_len = ((uint64_t)img_width_bytes + 1) * y;
    if (raw_len < img_len) return 0;                                  /* :1430 */

    for (j = 0; j < y; ++j) {
        stbi_uc* cur = out + (size_t)stride * j;
        stbi_uc* prior;
        int filter = *raw++;
        if (filter > 4) return 0;                                     /* :1438 */
        if (depth < 8) {
            if (img_width_bytes > x) return 0;
            cur += x * (uint32_t)out_n - img_width_bytes;             /* decode in place at the right end :1443 */
            filter_bytes = 1;
            width = (int)img_width_bytes;
        }
        prior = cur - stride;
        if (j == 0) filter = first_row_filter[filter];

        for (k = 0; k < filter_bytes; ++k) {                          /* first pixel :1453-1465 */
            switch (filter) {
            case F_none: case F_sub: case F_avg_first: case F_paeth_first: cur[k] = raw[k]; break;
            case F_up:    cur[k] = BYTECAST(raw[k] + prior[k]); break;
            case F_avg:   cur[k] = BYTECAST(raw[k] + (prior[k] >> 1)); break;
            case F_paeth: cur[k] = BYTECAST(raw[k] + paeth(0, prior[k], 0)); break;
            }
        }
        if (depth == 8) {
            if (img_n != out_n) cur[img_n] = 255;
            raw += img_n; cur += out_n; prior += out_n;
        } else if (depth == 16) {
            if (img_n != out_n) { cur[filter_bytes] = 255; cur[filter_bytes + 1] = 255; }
            raw += filter_bytes; cur += output_bytes; prior += output_bytes;
        } else { raw += 1; cur += 1; prior += 1; }

        if (depth < 8 || img_n == out_n) {                            /* :1488-1503 */
            const int nk = (width - 1) * filter_bytes;
            switch (filter) {
            case F_none:  memcpy(cur, raw, (size_t)nk); break;
            case F_sub:   for (k = 0; k < nk; ++k) cur[k] = BYTECAST(raw[k] + cur[k - filter_bytes]); break;
            case F_up:    for (k = 0; k < nk; ++k) cur[k] = BYTECAST(raw[k] + prior[k]); break;
            case F_avg:   for (k = 0; k < nk; ++k) cur[k] = BYTECAST(raw[k] + ((prior[k] + cur[k - filter_bytes]) >> 1)); break;
            case F_paeth: for (k = 0; k < nk; ++k) cur[k] = BYTECAST(raw[k] + paeth(cur[k - filter_bytes], prior[k], prior[k - filter_bytes])); break;
            case F_avg_first:   for (k = 0; k < nk; ++k) cur[k] = BYTECAST(raw[k] + (cur[k - filter_bytes] >> 1)); break;
            case F_paeth_first: for (k = 0; k < nk; ++k) cur[k] = BYTECAST(raw[k] + paeth(cur[k - filter_bytes], 0, 0)); break;
            }
            raw += nk;
        } else {                                                      /* alpha-inserting path :1504-1546 */
            for (i = x - 1; i >= 1; --i, cur[filter_bytes] = 255, raw += filter_bytes, cur += output_bytes, prior += output_bytes)
                for (k = 0; k < filter_bytes; ++k) {
                    switch (filter) {
                    case F_none:  cur[k] = raw[k]; break;
                    case F_sub:   cur[k] = BYTECAST(raw[k] + cur[k - output_bytes]); break;
                    case F_up:    cur[k] = BYTECAST(raw[k] + prior[k]); break;
                    case F_avg:   cur[k] = BYTECAST(raw[k] + ((prior[k] + cur[k - output_bytes]) >> 1)); break;
                    case F_paeth: cur[k] = BYTECAST(raw[k] + paeth(cur[k - output_bytes], prior[k], prior[k - output_bytes])); break;
                    case F_avg_first:   cur[k] = BYTECAST(raw[k] + (cur[k - output_bytes] >> 1)); break;
                    case F_paeth_first: cur[k] = BYTECAST(raw[k] + paeth(cur[k - output_bytes], 0, 0)); break;
                    }
                }
            if (depth == 16) {
                cur = out + (size_t)stride * j;
                for (i = 0; i < x; ++i, cur += output_bytes) cur[filter_bytes + 1] = 255;
            }
        }
    }

    if (depth < 8) {                                                  /* :1552-1620 */
        for (j = 0; j < y; ++j) {
            stbi_uc* cur = out + (size_t)stride * j;
            const stbi_uc* in_ = out + (size_t)stride * j + x * (uint32_t)out_n - img_width_bytes;
            const stbi_uc scale = (color == 0) ? depth_scale_table[depth] : 1;
            if (depth == 4) {
                for (k = (int)(x * (uint32_t)img_n); k >= 2; k -= 2, ++in_) {
                    *cur++ = (stbi_uc)(scale * ((*in_ >> 4)));
                    *cur++ = (stbi_uc)(scale * ((*in_) & 0x0f));
                }
                if (k > 0) *cur++ = (stbi_uc)(scale * ((*in_ >> 4)));
            } else if (depth == 2) {
                for (k = (int)(x * (uint32_t)img_n); k >= 4; k -= 4, ++in_) {
                    *cur++ = (stbi_uc)(scale * ((*in_ >> 6)));
                    *cur++ = (stbi_uc)(scale * ((*in_ >> 4) & 0x03));
                    *cur++ = (stbi_uc)(scale * ((*in_ >> 2) & 0x03));
                    *cur++ = (stbi_uc)(scale * ((*in_) & 0x03));
                }
                if (k > 0) *cur++ = (stbi_uc)(scale * ((*in_ >> 6)));
                if (k > 1) *cur++ = (stbi_uc)(scale * ((*in_ >> 4) & 0x03));
                if (k > 2) *cur++ = (stbi_uc)(scale * ((*in_ >> 2) & 0x03));
            } else if (depth == 1) {
                for (k = (int)(x * (uint32_t)img_n); k >= 8; k -= 8, ++in_) {
                    for (int b = 7; b >= 0; --b) *cur++ = (stbi_uc)(scale * ((*in_ >> b) & 0x01));
                }
                for (int b = 0; b < 7; ++b) if (k > b) *cur++ = (stbi_uc)(scale * ((*in_ >> (7 - b)) & 0x01));
            }
            if (img_n != out_n) {
                int q;
                cur = out + (size_t)stride * j;
                if (img_n == 1) {
                    for (q = (int)x - 1; q >= 0; --q) { cur[q*2+1] = 255; cur[q*2+0] = cur[q]; }
                } else {
                    for (q = (int)x - 1; q >= 0; --q) {
                        cur[q*4+3] = 255; cur[q*4+2] = cur[q*3+2]; cur[q*4+1] = cur[q*3+1]; cur[q*4+0] = cur[q*3+0];
                    }
                }
            }
        }
    } else if (depth == 16) {                                          /* :1621-1632 big-endian -> native */
        stbi_uc* cur = out;
        uint16_t* cur16 = (uint16_t*)out;
        for (i = 0; i < x * y * (uint32_t)out_n; ++i, cur16++, cur += 2)
            *cur16 = (uint16_t)((cur[0] << 8) | cur[1]);
    }
    return 1;
}

/* stbi__create_png_image :1637-1680 */
int orc_png_create_image(const uint8_t* raw, uint32_t raw_len, int img_n, int out_n,
                         uint32_t img_x, uint32_t img_y, int depth, int color, int interlaced, uint8_t* out)
{
    const int bytes = (depth == 16 ? 2 : 1);
    const int out_bytes = out_n * bytes;
    if (!interlaced)
        return orc_png_create_image_raw(raw, raw_len, img_n, out_n, img_x, img_y, depth, color, out);
    static const int xorig[7] = { 0,4,0,2,0,1,0 }, yorig[7] = { 0,0,4,0,2,0,1 };
    static const int xspc[7]  = { 8,8,4,4,2,2,1 }, yspc[7]  = { 8,8,8,4,4,2,2 };
    for (int p = 0; p < 7; ++p) {
        const uint32_t x = (img_x - xorig[p] + xspc[p] - 1) / xspc[p];
        const uint32_t y = (img_y - yorig[p] + yspc[p] - 1) / yspc[p];
        if (x && y) {
            const uint32_t img_len = ((((uint32_t)img_n * x * depth) + 7) >> 3) * y + y;
            uint8_t* tmp = (uint8_t*)malloc((size_t)x * y * out_bytes + 16);
            if (!tmp) return 0;
            if (!orc_png_create_image_raw(raw, raw_len, img_n, out_n, x, y, depth, color, tmp)) { free(tmp); return 0; }
            for (uint32_t j = 0; j < y; ++j)
                for (uint32_t i = 0; i < x; ++i) {
                    const uint32_t out_y = j * yspc[p] + yorig[p], out_x = i * xspc[p] + xorig[p];
                    memcpy(out + ((size_t)out_y * img_x + out_x) * out_bytes, tmp + ((size_t)j * x + i) * out_bytes, (size_t)out_bytes);
                }
            free(tmp);
            raw += img_len; raw_len -= img_len;
        }
    }
    return 1;
}

/* ---- memory reader with stb semantics (reads past the end yield 0) ------- */
typedef struct { const uint8_t* p; const uint8_t* end; } rd;
static inline int      get8(rd* s)    { return s->p < s->end ? *s->p++ : 0; }
static inline uint32_t get16be(rd* s) { uint32_t z = (uint32_t)get8(s); return (z << 8) + (uint32_t)get8(s); }
static inline uint32_t get32be(rd* s) { uint32_t z = get16be(s); return (z << 16) + get16be(s); }
static inline int      at_eof(rd* s)  { return s->p >= s->end; }
static inline void     skipn(rd* s, uint32_t n) { if ((size_t)(s->end - s->p) < n) s->p = s->end; else s->p += n; }

/* stbi_zlib_decode_malloc_guesssize_headerflag :1267-1321 (zlib instead of miniz, adler32 unchecked) */
static uint8_t* zlib_decode(const uint8_t* buf, uint32_t len, uint32_t initial, uint32_t* outlen, int parse_header)
{
    if (parse_header) {
        if (len < 2) return NULL;
        const int cmf = buf[0], flg = buf[1];
        if ((cmf * 256 + flg) % 31 != 0 || (flg & 32) || (cmf & 15) != 8) return NULL;
        buf += 2; len -= 2;
    }
    size_t cap = initial ? initial : 1;
    uint8_t* out = (uint8_t*)malloc(cap);
    if (!out) return NULL;
    z_stream z; memset(&z, 0, sizeof(z));
    if (inflateInit2(&z, -15) != Z_OK) { free(out); return NULL; }
    z.next_in = (Bytef*)buf; z.avail_in = len;
    z.next_out = out; z.avail_out = (uInt)cap;
    for (;;) {
        const int r = inflate(&z, Z_NO_FLUSH);
        if (r == Z_STREAM_END) break;
        if (r == Z_OK || r == Z_BUF_ERROR) {
            if (z.avail_out == 0) {
                if (cap > 536870912u) { inflateEnd(&z); free(out); return NULL; }
                size_t ncap = cap * 2; if (ncap < 32 * 1024) ncap = 32 * 1024;
                uint8_t* n = (uint8_t*)realloc(out, ncap);
                if (!n) { inflateEnd(&z); free(out); return NULL; }
                out = n; z.next_out = out + cap; z.avail_out = (uInt)(ncap - cap); cap = ncap;
                continue;
            }
            if (z.avail_in == 0) { inflateEnd(&z); free(out); return NULL; }   /* truncated stream */
            continue;
        }
        inflateEnd(&z); free(out); return NULL;
    }
    *outlen = (uint32_t)z.total_out;
    inflateEnd(&z);
    return out;
}

#define PNG_TYPE(a,b,c,d) (((uint32_t)(a) << 24) + ((uint32_t)(b) << 16) + ((uint32_t)(c) << 8) + (uint32_t)(d))

/* stbi__parse_png_file (SCAN_load) up to and including the inflate of finalize_decode :1777-1819 */
int orc_png_parse(const uint8_t* data, size_t len, orc_png_info* z)
{
    static const uint8_t sig[8] = { 137,80,78,71,13,10,26,10 };
    rd s = { data, data + len };
    memset(z, 0, sizeof(*z));
    z->ppmX = z->ppmY = z->pixelAspectRatio = -1;
    for (int i = 0; i < 8; ++i) if (get8(&s) != sig[i]) return 0;

    uint8_t* idata = NULL; uint32_t ioff = 0, idata_cap = 0;
    int first = 1, ok = 0;
    for (;;) {
        const uint32_t clen = get32be(&s), ctype = get32be(&s);
        switch (ctype) {
        case PNG_TYPE('C','g','B','I'): z->is_iphone = 1; skipn(&s, clen); break;
        case PNG_TYPE('p','H','Y','s'): {
            z->ppmX = (float)get32be(&s); z->ppmY = (float)get32be(&s);
            z->pixelAspectRatio = z->ppmX / z->ppmY;
            if (get8(&s) != 1) { z->ppmX = -1; z->ppmY = -1; }
        } break;
        case PNG_TYPE('I','H','D','R'): {
            if (!first || clen != 13) goto fail;
            first = 0;
            z->width = get32be(&s); z->height = get32be(&s);
            if (z->height > (1u << 24) || z->width > (1u << 24)) goto fail;
            z->depth = get8(&s);
            if (z->depth != 1 && z->depth != 2 && z->depth != 4 && z->depth != 8 && z->depth != 16) goto fail;
            z->color = get8(&s); if (z->color > 6) goto fail;
            if (z->color == 3 && z->depth == 16) goto fail;
            if (z->color == 3) z->pal_img_n = 3; else if (z->color & 1) goto fail;
            if (get8(&s)) goto fail;
            if (get8(&s)) goto fail;
            z->interlace = get8(&s); if (z->interlace > 1) goto fail;
            if (!z->width || !z->height) goto fail;
            if (!z->pal_img_n) {
                z->img_n = (z->color & 2 ? 3 : 1) + (z->color & 4 ? 1 : 0);
                if ((1u << 30) / z->width / (uint32_t)z->img_n < z->height) goto fail;
            } else {
                z->img_n = 1;
                if ((1u << 30) / z->width / 4 < z->height) goto fail;
            }
        } break;
        case PNG_TYPE('P','L','T','E'): {
            if (first || clen > 256 * 3) goto fail;
            z->pal_len = clen / 3;
            if (z->pal_len * 3 != clen) goto fail;
            for (uint32_t i = 0; i < z->pal_len; ++i) {
                z->palette[i*4+0] = (uint8_t)get8(&s); z->palette[i*4+1] = (uint8_t)get8(&s);
                z->palette[i*4+2] = (uint8_t)get8(&s); z->palette[i*4+3] = 255;
            }
        } break;
        case PNG_TYPE('t','R','N','S'): {
            if (first || idata) goto fail;
            if (z->pal_img_n) {
                if (z->pal_len == 0 || clen > z->pal_len) goto fail;
                z->pal_img_n = 4;
                for (uint32_t i = 0; i < clen; ++i) z->palette[i*4+3] = (uint8_t)get8(&s);
            } else {
                if (!(z->img_n & 1) || clen != (uint32_t)z->img_n * 2) goto fail;
                z->has_trans = 1;
                if (z->depth == 16) for (int k = 0; k < z->img_n; ++k) z->tc16[k] = (uint16_t)get16be(&s);
                else for (int k = 0; k < z->img_n; ++k) z->tc[k] = (uint8_t)((uint8_t)(get16be(&s) & 255) * depth_scale_table[z->depth]);
            }
        } break;
        case PNG_TYPE('I','D','A','T'): {
            if (first || (z->pal_img_n && !z->pal_len)) goto fail;
            if ((int32_t)(ioff + clen) < (int32_t)ioff) goto fail;
            if (ioff + clen > idata_cap) {
                uint32_t ncap = idata_cap ? idata_cap : (clen > 4096 ? clen : 4096);
                while (ioff + clen > ncap) ncap *= 2;
                uint8_t* n = (uint8_t*)realloc(idata, ncap);
                if (!n) goto fail;
                idata = n; idata_cap = ncap;
            }
            if ((size_t)(s.end - s.p) < clen) goto fail;              /* stbi__getn: "outofdata" */
            memcpy(idata + ioff, s.p, clen); s.p += clen; ioff += clen;
        } break;
        case PNG_TYPE('I','E','N','D'):
            if (first) goto fail;
            goto finalize;
        default:
            if (first) goto fail;
            if (ctype == 0 && at_eof(&s)) goto finalize;               /* Gamut issue #92 :2008-2012 */
            if ((ctype & (1u << 29)) == 0) goto fail;
            skipn(&s, clen);
            break;
        }
        get32be(&s);   /* CRC, unchecked */
    }
finalize:
    if (!idata) goto fail;
    {
        const uint32_t bpl = (z->width * (uint32_t)z->depth + 7) / 8;
        const uint32_t guess = bpl * z->height * (uint32_t)z->img_n + z->height;
        z->raw = zlib_decode(idata, ioff, guess, &z->raw_len, !z->is_iphone);
        ok = z->raw != NULL;
    }
fail:
    free(idata);
    return ok;
}

void orc_png_info_free(orc_png_info* z) { free(z->raw); z->raw = NULL; }

static uint8_t  compute_y8 (int r, int g, int b) { return (uint8_t) (((r * 77) + (g * 150) + (29 * b)) >> 8); }   /* :911-914 */
static uint16_t compute_y16(int r, int g, int b) { return (uint16_t)(((r * 77) + (g * 150) + (29 * b)) >> 8); }

/* stbi__convert_format :916-1043 (dst caller-provided) */
#define CONVERT_BODY(T, MAXV, CY)                                                         \
    for (uint32_t j = 0; j < y; ++j) {                                                    \
        const T* s = src + (size_t)j * x * img_n; T* d = dst + (size_t)j * x * req_comp;  \
        for (uint32_t i = 0; i < x; ++i, s += img_n, d += req_comp) {                     \
            switch (img_n * 8 + req_comp) {                                               \
            case 1*8+2: d[0] = s[0]; d[1] = MAXV; break;                                  \
            case 1*8+3: d[0] = d[1] = d[2] = s[0]; break;                                 \
            case 1*8+4: d[0] = d[1] = d[2] = s[0]; d[3] = MAXV; break;                    \
            case 2*8+1: d[0] = s[0]; break;                                               \
            case 2*8+3: d[0] = d[1] = d[2] = s[0]; break;                                 \
            case 2*8+4: d[0] = d[1] = d[2] = s[0]; d[3] = s[1]; break;                    \
            case 3*8+4: d[0] = s[0]; d[1] = s[1]; d[2] = s[2]; d[3] = MAXV; break;        \
            case 3*8+1: d[0] = CY(s[0], s[1], s[2]); break;                               \
            case 3*8+2: d[0] = CY(s[0], s[1], s[2]); d[1] = MAXV; break;                  \
            case 4*8+1: d[0] = CY(s[0], s[1], s[2]); break;                               \
            case 4*8+2: d[0] = CY(s[0], s[1], s[2]); d[1] = s[3]; break;                  \
            case 4*8+3: d[0] = s[0]; d[1] = s[1]; d[2] = s[2]; break;                     \
            default: memcpy(d, s, sizeof(T) * (size_t)req_comp); break;                   \
            }                                                                             \
        }                                                                                 \
    }
void orc_png_convert_format8(const uint8_t* src, int img_n, int req_comp, uint32_t x, uint32_t y, uint8_t* dst)
{ CONVERT_BODY(uint8_t, 255, compute_y8) }
void orc_png_convert_format16(const uint16_t* src, int img_n, int req_comp, uint32_t x, uint32_t y, uint16_t* dst)
{ CONVERT_BODY(uint16_t, 0xffff, compute_y16) }

/* stbi__do_png :2025-2055 incl. the tail of finalize_decode :1821-1857.
 * Returns malloc'd pixels; *bits = 8 or 16. */
static uint8_t* png_load(const uint8_t* data, size_t len, int* px, int* py, int* pn, int req_comp, int* bits)
{
    if (req_comp < 0 || req_comp > 4) return NULL;
    orc_png_info z;
    if (!orc_png_parse(data, len, &z)) return NULL;
    int img_n = z.img_n, img_out_n;
    if ((req_comp == img_n + 1 && req_comp != 3 && !z.pal_img_n) || z.has_trans) img_out_n = img_n + 1;   /* :1821-1824 */
    else img_out_n = img_n;
    const int bytes = z.depth == 16 ? 2 : 1;
    const size_t npx = (size_t)z.width * z.height;
    uint8_t* out = (uint8_t*)malloc(npx * img_out_n * bytes + 16);
    if (!out) { orc_png_info_free(&z); return NULL; }
    if (!orc_png_create_image(z.raw, z.raw_len, img_n, img_out_n, z.width, z.height, z.depth, z.color, z.interlace, out)) {
        free(out); orc_png_info_free(&z); return NULL;
    }
    if (z.has_trans) {                                                /* :1682-1730 */
        if (z.depth == 16) {
            uint16_t* p = (uint16_t*)out;
            if (img_out_n == 2) for (size_t i = 0; i < npx; ++i, p += 2) p[1] = (p[0] == z.tc16[0] ? 0 : 65535);
            else for (size_t i = 0; i < npx; ++i, p += 4) if (p[0] == z.tc16[0] && p[1] == z.tc16[1] && p[2] == z.tc16[2]) p[3] = 0;
        } else {
            uint8_t* p = out;
            if (img_out_n == 2) for (size_t i = 0; i < npx; ++i, p += 2) p[1] = (p[0] == z.tc[0] ? 0 : 255);
            else for (size_t i = 0; i < npx; ++i, p += 4) if (p[0] == z.tc[0] && p[1] == z.tc[1] && p[2] == z.tc[2]) p[3] = 0;
        }
    }
    if (z.pal_img_n) {                                                /* :1843-1851, 1732-1765 */
        img_n = z.pal_img_n; img_out_n = z.pal_img_n;
        if (req_comp >= 3) img_out_n = req_comp;
        uint8_t* p = (uint8_t*)malloc(npx * img_out_n + 16);
        if (!p) { free(out); orc_png_info_free(&z); return NULL; }
        for (size_t i = 0; i < npx; ++i) memcpy(p + i * img_out_n, z.palette + out[i] * 4, (size_t)img_out_n);
        free(out); out = p;
    } else if (z.has_trans) ++img_n;
    if (req_comp && req_comp != img_out_n) {                          /* :2038-2045 */
        uint8_t* good = (uint8_t*)malloc(npx * req_comp * bytes + 16);
        if (!good) { free(out); orc_png_info_free(&z); return NULL; }
        if (bytes == 1) orc_png_convert_format8(out, img_out_n, req_comp, z.width, z.height, good);
        else orc_png_convert_format16((uint16_t*)out, img_out_n, req_comp, z.width, z.height, (uint16_t*)good);
        free(out); out = good;
    }
    *px = (int)z.width; *py = (int)z.height; if (pn) *pn = img_n;
    *bits = z.depth <= 8 ? 8 : 16;
    orc_png_info_free(&z);
    return out;
}

uint8_t* orc_stbi_load_from_memory(const uint8_t* data, size_t len, int* x, int* y, int* comp, int req_comp)
{
    int bits = 8, n = 0;
    uint8_t* r = png_load(data, len, x, y, &n, req_comp, &bits);
    if (comp) *comp = n;
    if (!r) return NULL;
    if (bits != 8) {                                                  /* stbi__convert_16_to_8 :635-649 */
        const size_t cnt = (size_t)*x * *y * (req_comp == 0 ? n : req_comp);
        uint8_t* red = (uint8_t*)malloc(cnt + 16);
        if (!red) { free(r); return NULL; }
        for (size_t i = 0; i < cnt; ++i) red[i] = (uint8_t)((((uint16_t*)r)[i] >> 8) & 0xFF);
        free(r); r = red;
    }
    return r;
}

uint16_t* orc_stbi_load_16_from_memory(const uint8_t* data, size_t len, int* x, int* y, int* comp, int req_comp)
{
    int bits = 8, n = 0;
    uint8_t* r = png_load(data, len, x, y, &n, req_comp, &bits);
    if (comp) *comp = n;
    if (!r) return NULL;
    if (bits != 16) {                                                 /* stbi__convert_8_to_16 :651-666 */
        const size_t cnt = (size_t)*x * *y * (req_comp == 0 ? n : req_comp);
        uint16_t* e = (uint16_t*)malloc(cnt * 2 + 16);
        if (!e) { free(r); return NULL; }
        for (size_t i = 0; i < cnt; ++i) e[i] = (uint16_t)((r[i] << 8) + r[i]);
        free(r); return e;
    }
    return (uint16_t*)r;
}
