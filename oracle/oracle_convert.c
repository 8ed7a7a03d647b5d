/*
 * oracle_convert.c -- CPU restatement of Gamut's scanline conversion matrix.
 * TEST INFRASTRUCTURE ONLY (see gamut_oracle.h).
 *
 * Follows /root/reference/source/gamut/scanline.d:
 *   scanlinesInterType :25-31, scanlinesCopy :37-55, scanlinesConvert :70-121,
 *   "to rgba8" rows :160-194, "from rgba8" rows :201-234,
 *   "to rgbaf32" rows :240-529, "from rgbaf32" rows :539-803,
 *   dispatchers :841-930; 8-bit predicate internals/types.d:99-111,144-147.
 *
 * Arithmetic contract restated from the D source: every binary32 operation is
 * rounded individually (x86-64 SSE code generation of LDC/DMD: no FMA, no
 * reassociation).  Build with -ffp-contract=off.  float -> integer casts are
 * x86 cvttss2si (truncate; NaN / out of int32 range -> 0x80000000) followed by
 * taking the low 8 / 16 bits, which is what `cast(ubyte)` / `cast(ushort)` of
 * a float compile to on x86-64.
 */
#include "gamut_oracle.h"
#include <string.h>

static const int k_size[ORC_NUM_TYPES] = { 1,2,4, 2,4,8, 2,4,8, 3,6,12, 4,8,16, 4,8,16 };

int orc_pixel_type_size(int type)
{
    if (type < 0 || type >= ORC_NUM_TYPES) return 0;
    return k_size[type];
}

/* internals/types.d:99-111 : only l8, la8, rgb8, rgba8 (premultiplied 8-bit excluded) */
static int is8bit(int t) { return t == ORC_l8 || t == ORC_la8 || t == ORC_rgb8 || t == ORC_rgba8; }

int orc_scanlines_inter_type(int srcType, int dstType)
{
    if (is8bit(srcType) && is8bit(dstType)) return ORC_rgba8;
    return ORC_rgbaf32;
}

int orc_scanlines_copy(int type, const uint8_t* src, int srcPitch,
                       uint8_t* dst, int dstPitch, int width, int height)
{
    int bytes = orc_pixel_type_size(type) * width;
    for (int y = 0; y < height; ++y) {
        memcpy(dst, src, (size_t)bytes);
        src += srcPitch;
        dst += dstPitch;
    }
    return 1;
}

/* x86 cvttss2si */
static inline int32_t cvtt(float x)
{
    if (!(x >= -2147483648.0f && x < 2147483648.0f)) return INT32_MIN;
    return (int32_t)x;
}
static inline uint8_t  to_u8 (float x) { return (uint8_t) cvtt(x); }
static inline uint16_t to_u16(float x) { return (uint16_t)cvtt(x); }

/* ---- to rgba8 (scanline.d:160-194, 231-234) ---- */
static void to_rgba8(int t, const uint8_t* in, uint8_t* out, int w)
{
    for (int x = 0; x < w; ++x, out += 4) {
        switch (t) {
        case ORC_l8:   out[0] = out[1] = out[2] = in[x]; out[3] = 255; break;
        case ORC_la8:  out[0] = out[1] = out[2] = in[2*x]; out[3] = in[2*x+1]; break;
        case ORC_rgb8: out[0] = in[3*x]; out[1] = in[3*x+1]; out[2] = in[3*x+2]; out[3] = 255; break;
        default:       memcpy(out, in + 4*x, 4); break; /* rgba8 */
        }
    }
}

/* ---- from rgba8 (scanline.d:201-234): l8 keeps R only ---- */
static void from_rgba8(int t, const uint8_t* in, uint8_t* out, int w)
{
    for (int x = 0; x < w; ++x, in += 4) {
        switch (t) {
        case ORC_l8:   out[x] = in[0]; break;
        case ORC_la8:  out[2*x] = in[0]; out[2*x+1] = in[3]; break;
        case ORC_rgb8: out[3*x] = in[0]; out[3*x+1] = in[1]; out[3*x+2] = in[2]; break;
        default:       memcpy(out + 4*x, in, 4); break;
        }
    }
}

/* ---- to rgbaf32 (scanline.d:240-529) ---- */
static void to_rgbaf32(int t, const uint8_t* in, float* out, int w)
{
    const uint16_t* in16 = (const uint16_t*)in;
    const float*    inf  = (const float*)in;
    for (int x = 0; x < w; ++x, out += 4) {
        float r, g, b, a;
        switch (t) {
        case ORC_l8:    r = g = b = in[x] / 255.0f;      a = 1.0f; break;
        case ORC_l16:   r = g = b = in16[x] / 65535.0f;  a = 1.0f; break;
        case ORC_lf32:  r = g = b = inf[x];              a = 1.0f; break;
        case ORC_la8:   r = g = b = in[2*x] / 255.0f;     a = in[2*x+1] / 255.0f; break;
        case ORC_la16:  r = g = b = in16[2*x] / 65535.0f; a = in16[2*x+1] / 65535.0f; break;
        case ORC_laf32: r = g = b = inf[2*x];             a = inf[2*x+1]; break;
        case ORC_lap8:  b = in[2*x] / 255.0f;     a = in[2*x+1] / 255.0f;     if (a != 0) b /= a; r = g = b; break;
        case ORC_lap16: b = in16[2*x] / 65535.0f; a = in16[2*x+1] / 65535.0f; if (a != 0) b /= a; r = g = b; break;
        case ORC_lapf32:b = inf[2*x];             a = inf[2*x+1];             if (a != 0) b /= a; r = g = b; break;
        case ORC_rgb8:  r = in[3*x] / 255.0f; g = in[3*x+1] / 255.0f; b = in[3*x+2] / 255.0f; a = 1.0f; break;
        case ORC_rgb16: r = in16[3*x] / 65535.0f; g = in16[3*x+1] / 65535.0f; b = in16[3*x+2] / 65535.0f; a = 1.0f; break;
        case ORC_rgbf32:r = inf[3*x]; g = inf[3*x+1]; b = inf[3*x+2]; a = 1.0f; break;
        case ORC_rgba8: r = in[4*x] / 255.0f; g = in[4*x+1] / 255.0f; b = in[4*x+2] / 255.0f; a = in[4*x+3] / 255.0f; break;
        case ORC_rgba16:r = in16[4*x] / 65535.0f; g = in16[4*x+1] / 65535.0f; b = in16[4*x+2] / 65535.0f; a = in16[4*x+3] / 65535.0f; break;
        case ORC_rgbap8:
            r = in[4*x] / 255.0f; g = in[4*x+1] / 255.0f; b = in[4*x+2] / 255.0f; a = in[4*x+3] / 255.0f;
            if (a != 0) { r /= a; g /= a; b /= a; } break;
        case ORC_rgbap16:
            r = in16[4*x] / 65535.0f; g = in16[4*x+1] / 65535.0f; b = in16[4*x+2] / 65535.0f; a = in16[4*x+3] / 65535.0f;
            if (a != 0) { r /= a; g /= a; b /= a; } break;
        case ORC_rgbapf32:
            r = inf[4*x]; g = inf[4*x+1]; b = inf[4*x+2]; a = inf[4*x+3];
            if (a != 0) { r /= a; g /= a; b /= a; } break;
        default: /* rgbaf32: memcpy, scanline.d:749-752 */
            memcpy(out, inf + 4*x, 16); continue;
        }
        out[0] = r; out[1] = g; out[2] = b; out[3] = a;
    }
}

/* ---- from rgbaf32 (scanline.d:539-803). Evaluation order is the D source's:
 * grey:  0.5f + (((r+g)+b) * M) / 3.0f ; premul grey: ((((r+g)+b)*a)*M)/3.0f
 * colour: 0.5f + v*M ; premul colour: 0.5f + (v*a)*M                         */
static void from_rgbaf32(int t, const float* in, uint8_t* out, int w)
{
    uint16_t* o16 = (uint16_t*)out;
    float*    of  = (float*)out;
    for (int x = 0; x < w; ++x, in += 4) {
        const float r = in[0], g = in[1], b = in[2], a = in[3];
        switch (t) {
        case ORC_l8:    out[x] = to_u8 (0.5f + (r + g + b) * 255.0f / 3.0f); break;
        case ORC_l16:   o16[x] = to_u16(0.5f + (r + g + b) * 65535.0f / 3.0f); break;
        case ORC_lf32:  of[x]  = (r + g + b) / 3.0f; break;
        case ORC_la8:   out[2*x] = to_u8 (0.5f + (r + g + b) * 255.0f / 3.0f);   out[2*x+1] = to_u8 (0.5f + a * 255.0f); break;
        case ORC_la16:  o16[2*x] = to_u16(0.5f + (r + g + b) * 65535.0f / 3.0f); o16[2*x+1] = to_u16(0.5f + a * 65535.0f); break;
        case ORC_laf32: of[2*x]  = (r + g + b) / 3.0f; of[2*x+1] = a; break;
        case ORC_lap8:  out[2*x] = to_u8 (0.5f + (r + g + b) * a * 255.0f / 3.0f);   out[2*x+1] = to_u8 (0.5f + a * 255.0f); break;
        case ORC_lap16: o16[2*x] = to_u16(0.5f + (r + g + b) * a * 65535.0f / 3.0f); o16[2*x+1] = to_u16(0.5f + a * 65535.0f); break;
        case ORC_lapf32:of[2*x]  = (r + g + b) * a / 3.0f; of[2*x+1] = a; break;
        case ORC_rgb8:  out[3*x] = to_u8 (0.5f + r * 255.0f);   out[3*x+1] = to_u8 (0.5f + g * 255.0f);   out[3*x+2] = to_u8 (0.5f + b * 255.0f); break;
        case ORC_rgb16: o16[3*x] = to_u16(0.5f + r * 65535.0f); o16[3*x+1] = to_u16(0.5f + g * 65535.0f); o16[3*x+2] = to_u16(0.5f + b * 65535.0f); break;
        case ORC_rgbf32:of[3*x] = r; of[3*x+1] = g; of[3*x+2] = b; break;
        case ORC_rgba8:
            out[4*x] = to_u8(0.5f + r * 255.0f); out[4*x+1] = to_u8(0.5f + g * 255.0f);
            out[4*x+2] = to_u8(0.5f + b * 255.0f); out[4*x+3] = to_u8(0.5f + a * 255.0f); break;
        case ORC_rgba16:
            o16[4*x] = to_u16(0.5f + r * 65535.0f); o16[4*x+1] = to_u16(0.5f + g * 65535.0f);
            o16[4*x+2] = to_u16(0.5f + b * 65535.0f); o16[4*x+3] = to_u16(0.5f + a * 65535.0f); break;
        case ORC_rgbap8:
            out[4*x] = to_u8(0.5f + r * a * 255.0f); out[4*x+1] = to_u8(0.5f + g * a * 255.0f);
            out[4*x+2] = to_u8(0.5f + b * a * 255.0f); out[4*x+3] = to_u8(0.5f + a * 255.0f); break;
        case ORC_rgbap16:
            o16[4*x] = to_u16(0.5f + r * a * 65535.0f); o16[4*x+1] = to_u16(0.5f + g * a * 65535.0f);
            o16[4*x+2] = to_u16(0.5f + b * a * 65535.0f); o16[4*x+3] = to_u16(0.5f + a * 65535.0f); break;
        case ORC_rgbapf32:
            of[4*x] = r * a; of[4*x+1] = g * a; of[4*x+2] = b * a; of[4*x+3] = a; break;
        default: /* rgbaf32 */
            memcpy(of + 4*x, in, 16); break;
        }
    }
}

/* scanline.d:841-885 */
static void to_intermediate(int srcType, const uint8_t* src, int interType, uint8_t* dst, int w)
{
    if (interType == ORC_rgba8) to_rgba8(srcType, src, dst, w);
    else                        to_rgbaf32(srcType, src, (float*)dst, w);
}
/* scanline.d:887-930 */
static void from_intermediate(int interType, const uint8_t* src, int dstType, uint8_t* dst, int w)
{
    if (interType == ORC_rgba8) from_rgba8(dstType, src, dst, w);
    else                        from_rgbaf32(dstType, (const float*)src, dst, w);
}

int orc_scanlines_convert(int srcType, const uint8_t* src, int srcPitch,
                          int dstType, uint8_t* dst, int dstPitch,
                          int width, int height, int interType, uint8_t* interBuf)
{
    if (srcType == dstType)
        return orc_scanlines_copy(srcType, src, srcPitch, dst, dstPitch, width, height);
    if (srcType < 0 || dstType < 0 || srcType >= ORC_NUM_TYPES || dstType >= ORC_NUM_TYPES)
        return 0;
    for (int y = 0; y < height; ++y) {
        if (srcType == interType)
            from_intermediate(interType, src, dstType, dst, width);
        else if (dstType == interType)
            to_intermediate(srcType, src, interType, dst, width);
        else {
            to_intermediate(srcType, src, interType, interBuf, width);
            from_intermediate(interType, interBuf, dstType, dst, width);
        }
        src += srcPitch;
        dst += dstPitch;
    }
    return 1;
}
