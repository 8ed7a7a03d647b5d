/* oracle_qoi.c -- CPU restatement of the reference's QOI decoder (source/gamut/codecs/qoi.d:448-550, the "Quite OK Image"
 * reference decoder).  TEST INFRASTRUCTURE ONLY (see gamut_oracle.h): used by tests/, smoke() and bench.py's cpu_baseline. */
#include "gamut_oracle.h"
#include <stdlib.h>
#include <string.h>

enum { OP_INDEX = 0x00, OP_DIFF = 0x40, OP_LUMA = 0x80, OP_RUN = 0xc0, OP_RGB = 0xfe, OP_RGBA = 0xff, MASK_2 = 0xc0 };   /* :230-237 */
#define QOI_MAGIC 0x716F6966u            /* "qoif" :244 */
#define QOI_HEADER_SIZE 14               /* :245 */
#define QOI_PIXELS_MAX 400000000u        /* :251 */
#define QOI_PADDING 8                    /* qoi_padding :268 */

static uint32_t read32(const uint8_t* b, int* p)        /* qoi_read_32 :281-287 */
{
    const uint32_t a = b[(*p)++], c = b[(*p)++], d = b[(*p)++], e = b[(*p)++];
    return a << 24 | c << 16 | d << 8 | e;
}

uint8_t* orc_qoi_decode(const uint8_t* data, int size, orc_qoi_desc* desc, int channels)
{
    typedef struct { uint8_t r, g, b, a; } rgba;
    rgba index[64], px;
    int p = 0, run = 0;

    if ((channels != 0 && channels != 3 && channels != 4) || size < QOI_HEADER_SIZE + QOI_PADDING) return NULL;   /* :458-462 */
    const uint32_t magic = read32(data, &p);
    desc->width = read32(data, &p);
    desc->height = read32(data, &p);
    desc->channels = data[p++];
    desc->colorspace = data[p++];
    if (desc->width == 0 || desc->height == 0 || desc->channels < 3 || desc->channels > 4 || desc->colorspace > 1 ||
        magic != QOI_MAGIC || desc->height >= QOI_PIXELS_MAX / desc->width) return NULL;                           /* :472-480 */
    if (channels == 0) channels = desc->channels;

    const int px_len = (int)(desc->width * desc->height * (uint32_t)channels);
    uint8_t* pixels = (uint8_t*)malloc((size_t)px_len);
    if (!pixels) return NULL;
    memset(index, 0, sizeof(index));
    px.r = px.g = px.b = 0; px.a = 255;

    const int chunks_len = size - QOI_PADDING;
    for (int px_pos = 0; px_pos < px_len; px_pos += channels) {
        if (run > 0) run--;
        else if (p < chunks_len) {
            const int b1 = data[p++];
            if (b1 == OP_RGB)       { px.r = data[p++]; px.g = data[p++]; px.b = data[p++]; }
            else if (b1 == OP_RGBA) { px.r = data[p++]; px.g = data[p++]; px.b = data[p++]; px.a = data[p++]; }
            else if ((b1 & MASK_2) == OP_INDEX) px = index[b1];
            else if ((b1 & MASK_2) == OP_DIFF) {
                px.r = (uint8_t)(px.r + ((b1 >> 4) & 3) - 2);
                px.g = (uint8_t)(px.g + ((b1 >> 2) & 3) - 2);
                px.b = (uint8_t)(px.b + (b1 & 3) - 2);
            } else if ((b1 & MASK_2) == OP_LUMA) {
                const int b2 = data[p++], vg = (b1 & 0x3f) - 32;
                px.r = (uint8_t)(px.r + vg - 8 + ((b2 >> 4) & 0x0f));
                px.g = (uint8_t)(px.g + vg);
                px.b = (uint8_t)(px.b + vg - 8 + (b2 & 0x0f));
            } else if ((b1 & MASK_2) == OP_RUN) run = b1 & 0x3f;
            index[(px.r * 3 + px.g * 5 + px.b * 7 + px.a * 11) % 64] = px;                                         /* QOI_COLOR_HASH :239-242 */
        }
        if (channels == 4) { pixels[px_pos] = px.r; pixels[px_pos + 1] = px.g; pixels[px_pos + 2] = px.b; pixels[px_pos + 3] = px.a; }
        else               { pixels[px_pos] = px.r; pixels[px_pos + 1] = px.g; pixels[px_pos + 2] = px.b; }
    }
    return pixels;
}
